#!/bin/bash
# round 5, call 5: the prediction network queued in slices between the encoder blocks (host starvation of the main queue)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py tests/test_contextnet_gpu.py tests/test_parity_baseline_gpu.py -m gpu -x -q 2>&1 | tail -4
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
export GPU_MAX_HW_QUEUES=4
for n in 0 8 4 16 0 8; do
  TFASR_PRED_SLICES=$n TFASR_BENCH_HOST=1 timeout 200 python bench.py $B > $O/single_s$n.json 2> $O/single_s$n.err
  TFASR_PRED_SLICES=$n timeout 200 python bench.py $B --dp-hooks > $O/dp_s$n.json 2>> $O/err
  echo "slices=$n single $(grep -o '"ms_per_step": [0-9.]*' $O/single_s$n.json) $(grep host $O/single_s$n.err) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/dp_s$n.json)"
done
for n in 0 8; do
  TFASR_PRED_SLICES=$n timeout 200 python bench.py $B --model S > $O/S_s$n.json 2>> $O/err
  TFASR_PRED_SLICES=$n timeout 300 python bench.py --model contextnet --alpha 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/cn_s$n.json 2>> $O/err
  echo "slices=$n S $(grep -o '"ms_per_step": [0-9.]*' $O/S_s$n.json) | contextnet $(grep -o '"ms_per_step": [0-9.]*' $O/cn_s$n.json)"
done
ENV="TFASR_PRED_SLICES=8" bash tools/prof_streams.sh > $O/streams_s8.txt 2>&1
ENV="TFASR_PRED_SLICES=0" bash tools/prof_streams.sh > $O/streams_s0.txt 2>&1
head -24 $O/streams_s8.txt | cut -c1-200; head -24 $O/streams_s0.txt | cut -c1-200
tail -3 $O/err
