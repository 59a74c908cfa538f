#!/bin/bash
# round 5, call 31: the subsampling's weight gradients on the side stream (TFASR_SUB_WGRAD_SIDE) - parity tests, A/B, data-parallel route
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t31
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_block_hoist_gpu.py tests/test_parity_baseline_gpu.py -x -q -m gpu 2>&1 | tail -2
B="--steps 30 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
TFASR_SUB_WGRAD_SIDE=0 timeout 200 python bench.py $B > $O/old$i.json 2>> $O/err
TFASR_SUB_WGRAD_SIDE=1 timeout 200 python bench.py $B > $O/new$i.json 2>> $O/err
done
TFASR_SUB_WGRAD_SIDE=1 timeout 200 python bench.py $B --dp-hooks > $O/dp.json 2>> $O/err
TFASR_SUB_WGRAD_SIDE=1 timeout 200 python bench.py $B --model S > $O/S.json 2>> $O/err
TFASR_SUB_WGRAD_SIDE=0 timeout 200 python bench.py $B --model S > $O/S_old.json 2>> $O/err
for f in old1 new1 old2 new2 dp S_old S; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
