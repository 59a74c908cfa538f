#!/bin/bash
# round 5, call 38: the lean build of the transposed query backward (168 registers, three workgroups per CU) against the pipelined one
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t38
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu 2>&1 | tail -2
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
TFASR_ATTN_BWDQ_T=1 timeout 200 python bench.py $B > $O/pipe$i.json 2>> $O/err
TFASR_ATTN_BWDQ_T=2 timeout 200 python bench.py $B > $O/lean$i.json 2>> $O/err
done
for f in pipe1 lean1 pipe2 lean2; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
TFASR_ATTN_BWDQ_T=2 bash tools/prof_quick.sh r5_t38/prof > $O/prof.txt 2>&1
grep -i "relattn" $O/prof.txt | cut -c1-44,100-170 | head -5
