#!/bin/bash
# round 5, call 41: depthwise conv blocks of 32 instead of 64 time steps (768 instead of 384 blocks; probe lib -DTFASR_DW_TG=16)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t41
mkdir -p $O
cd $R
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "dwconv or conv or step_matches or bf16_step" 2>&1 | tail -1
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
timeout 200 python bench.py $B > $O/tg32_$i.json 2>> $O/err
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 200 python bench.py $B > $O/tg16_$i.json 2>> $O/err
done
for f in tg32_1 tg16_1 tg32_2 tg16_2; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so bash tools/prof_quick.sh r5_t41/prof16 > $O/prof16.txt 2>&1
grep -i "dwconv" $O/prof16.txt | cut -c1-50,100-170 | head -6
