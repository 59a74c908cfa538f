#!/bin/bash
# round 5, call 12: deferred block gradients on the side stream beside the subsampling's backward; vocabulary weight gradient on the side stream
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_block_hoist_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_ctc_model_gpu.py tests/test_lstm_persist_gpu.py tests/test_contextnet_gpu.py -m gpu -x -q 2>&1 | tail -4
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for v in "TFASR_DEFER_SIDE=0" "TFASR_DEFER_SIDE=1" "TFASR_JOINT_WGRAD_AUX=1" "TFASR_DEFER_SIDE=0" "TFASR_DEFER_SIDE=1" "TFASR_JOINT_WGRAD_AUX=1"; do
  env $v timeout 200 python bench.py $B > $O/s.json 2>> $O/err
  env $v timeout 200 python bench.py $B --dp-hooks > $O/d.json 2>> $O/err
  echo "$v: single $(grep -o '"ms_per_step": [0-9.]*' $O/s.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/d.json)"
done
tail -2 $O/err
