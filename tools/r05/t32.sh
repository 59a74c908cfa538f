#!/bin/bash
# round 5, call 32: key-side attention backward without the per-iteration bias-row rewrite and its barrier - parity tests, kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t32
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_attn_bwdq_t_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/prof_quick.sh r5_t32/prof > $O/prof.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/prof/trace.log | head -1
grep -i "relattn" $O/prof.txt | cut -c1-40,100-170 | head
