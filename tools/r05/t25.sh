#!/bin/bash
# round 5, call 25: transposed query backward, second version - comparison test, phase clocks, step A/B, kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t25
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_attn_bwdq_t_gpu.py "tests/test_model_gpu.py::test_streaming_conformer_step_matches_oracle" tests/test_model_gpu.py::test_bf16_step_close_to_oracle -x -q -m gpu 2>&1 | tail -3
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 300 python tools/attn_timing.py 2>&1 | tail -8
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1; do
TFASR_ATTN_BWDQ_T=0 timeout 200 python bench.py $B > $O/old$i.json 2>> $O/err
TFASR_ATTN_BWDQ_T=1 timeout 200 python bench.py $B > $O/new$i.json 2>> $O/err
done
for f in old1 new1; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
bash tools/prof_quick.sh r5_t25/prof > $O/prof.txt 2>&1
grep -i "relattn" $O/prof.txt | cut -c1-40,100-170 | head
