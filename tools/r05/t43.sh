#!/bin/bash
# round 5, call 43: depthwise conv data kernels in 16-step blocks (probe lib) against the 32-step default
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t43
mkdir -p $O
cd $R
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dwconv or conv" 2>&1 | tail -1
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
timeout 200 python bench.py $B > $O/tgd16_$i.json 2>> $O/err
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 200 python bench.py $B > $O/tgd8_$i.json 2>> $O/err
done
for f in tgd16_1 tgd8_1 tgd16_2 tgd8_2; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so bash tools/prof_quick.sh r5_t43/prof8 > $O/prof8.txt 2>&1
grep -i "dwconv_tile" $O/prof8.txt | cut -c1-50,100-170 | head -3
