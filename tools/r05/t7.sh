#!/bin/bash
# round 5, call 7: 24-bit dropout hash + packed-f32 conv1/BatchNorm kernels: whole GPU suite, bench, kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/single.json 2>> $O/err
timeout 200 python bench.py $B --dp-hooks > $O/dp.json 2>> $O/err
echo "single $(grep -o '"ms_per_step": [0-9.]*' $O/single.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/dp.json)"
bash tools/prof_quick.sh r5_t7/prof_M > $O/prof.txt 2>&1
head -42 $O/prof.txt | cut -c1-170
