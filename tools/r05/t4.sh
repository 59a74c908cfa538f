#!/bin/bash
# round 5, call 4: a data-parallel rank on the hoisted three-stream step (deferred gradient region, one side stream, prediction stream at high priority)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_block_hoist_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_checkpoint.py -m gpu -x -q 2>&1 | tail -4
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for q in 2 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py $B > $O/single_q$q.json 2>> $O/err
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py $B --dp-hooks > $O/dp_q$q.json 2>> $O/err
  GPU_MAX_HW_QUEUES=$q TFASR_DP_FORCE_SPLIT=1 timeout 200 python bench.py $B --dp-hooks > $O/dp_split_q$q.json 2>> $O/err
  echo "q=$q single $(grep -o '"ms_per_step": [0-9.]*' $O/single_q$q.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/dp_q$q.json) | dp+split $(grep -o '"ms_per_step": [0-9.]*' $O/dp_split_q$q.json)"
done
for v in "TFASR_PRED_PRIO=0" "TFASR_ONE_SIDE_STREAM=0" "TFASR_PRED_PRIO=0 TFASR_ONE_SIDE_STREAM=0" "TFASR_LSTM_PERSIST=1" "TFASR_BLOCK_HOIST=0 TFASR_WGRAD_STREAM=0"; do
  n=$(echo $v | tr ' =' '__')
  env $v timeout 200 python bench.py $B > $O/single_$n.json 2>> $O/err
  env $v timeout 200 python bench.py $B --dp-hooks > $O/dp_$n.json 2>> $O/err
  echo "$v: single $(grep -o '"ms_per_step": [0-9.]*' $O/single_$n.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/dp_$n.json)"
done
tail -5 $O/err
