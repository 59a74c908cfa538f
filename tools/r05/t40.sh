#!/bin/bash
# round 5, call 40: non-temporal stores for the joint projection's [cells, V] logits (gemm_big TR epilogue) against plain stores (probe lib)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t40
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_rnnt_gpu.py tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -1
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
timeout 200 python bench.py $B > $O/nt$i.json 2>> $O/err
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 200 python bench.py $B > $O/plain$i.json 2>> $O/err
done
for f in nt1 plain1 nt2 plain2; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1) joint fwd $(grep -o '"ms_per_launch": [0-9.]*' $O/$f.json | head -1) $(grep -o '"frac": [0-9.]*' $O/$f.json | head -1)"; done
