#!/bin/bash
# round 5, call 39: query backward with the epilogue's bias-row loads requested before the loop; dS stores non-temporal (probe lib) or not
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t39
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu 2>&1 | tail -1
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 300 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu -k "mode1 or 1-" 2>&1 | tail -1
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
timeout 200 python bench.py $B > $O/plain$i.json 2>> $O/err
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 200 python bench.py $B > $O/nt$i.json 2>> $O/err
done
for f in plain1 nt1 plain2 nt2; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
bash tools/prof_quick.sh r5_t39/prof > $O/prof.txt 2>&1
grep -i "relattn" $O/prof.txt | cut -c1-44,100-170 | head -5
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so bash tools/prof_quick.sh r5_t39/prof_nt > $O/prof_nt.txt 2>&1
grep -i "relattn" $O/prof_nt.txt | cut -c1-44,100-170 | head -5
