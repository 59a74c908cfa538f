#!/bin/bash
# round 5, call 28: software-pipelined forward attention (no extra LDS) - parity tests, step, kernel times
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t28
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_attn_bwdq_t_gpu.py tests/test_model_gpu.py tests/test_parity_baseline_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/new1.json 2>> $O/err
echo "new: $(grep -o '"ms_per_step": [0-9.]*' $O/new1.json | head -1)"
bash tools/prof_quick.sh r5_t28/prof > $O/prof.txt 2>&1
grep -i "relattn" $O/prof.txt | cut -c1-40,100-170 | head
