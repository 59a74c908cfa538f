#!/bin/bash
# round 5, call 46: key-per-lane attention backward (opt-in) - comparison test, kernel time, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t46
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu -s -k key_per_lane 2>&1 | grep -v "^$" | tail -8
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/old1.json 2>> $O/err
TFASR_ATTN_BWDK_T=1 timeout 200 python bench.py $B > $O/new1.json 2>> $O/err
for f in old1 new1; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
TFASR_ATTN_BWDK_T=1 bash tools/prof_quick.sh r5_t46/prof > $O/prof.txt 2>&1
grep -i "relattn" $O/prof.txt | cut -c1-44,100-170 | head -5
