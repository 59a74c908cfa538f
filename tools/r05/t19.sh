#!/bin/bash
# round 5, call 19: transposed query-side attention backward - correctness against the row-oriented kernel, the attention / model parity
# tests, then A/B of the step and the kernel's own time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t19
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_attn_bwdq_t_gpu.py tests/test_model_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -15
timeout 600 python -m pytest tests/test_parity_baseline_gpu.py tests/test_block_hoist_gpu.py -x -q -m gpu 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
TFASR_ATTN_BWDQ_T=0 timeout 200 python bench.py $B > $O/old$i.json 2>> $O/err
TFASR_ATTN_BWDQ_T=1 timeout 200 python bench.py $B > $O/new$i.json 2>> $O/err
done
for f in old1 new1 old2 new2; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
bash tools/prof_quick.sh r5_t19/prof > $O/prof.txt 2>&1
grep -i "relattn\|Name" $O/prof.txt | cut -c1-170 | head
