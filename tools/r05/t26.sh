#!/bin/bash
# round 5, call 26: transposed query backward with three workgroups per CU (168 VGPRs + spills, image over the strip)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t26
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu 2>&1 | tail -2
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 300 python tools/attn_timing.py 2>&1 | tail -8
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/new1.json 2>> $O/err
echo "new: $(grep -o '"ms_per_step": [0-9.]*' $O/new1.json | head -1)"
bash tools/prof_quick.sh r5_t26/prof > $O/prof.txt 2>&1
grep -i "relattn" $O/prof.txt | cut -c1-40,100-170 | head
