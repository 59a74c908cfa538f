#!/bin/bash
# round 5, call 48: query backward with the score read-back as four aligned 4-byte LDS reads - test, kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t48
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu -k "transposed" 2>&1 | tail -1
bash tools/prof_quick.sh r5_t48/prof > $O/prof.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/prof/trace.log | head -1
grep -i "relattn" $O/prof.txt | cut -c1-44,100-170 | head -4
