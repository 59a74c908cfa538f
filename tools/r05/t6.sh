#!/bin/bash
# round 5, call 6: pure host cost of a step (tiny batch: the GPU is never the limit), default queue count
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t6
mkdir -p $O
cd $R
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
TFASR_BENCH_HOST=1 timeout 200 python bench.py $B --batch 2 > $O/b2.json 2> $O/b2.err
echo "batch 2: $(grep -o '"ms_per_step": [0-9.]*' $O/b2.json) $(grep host $O/b2.err)"
TFASR_BENCH_HOST=1 timeout 200 python bench.py $B --batch 2 --dp-hooks > $O/b2dp.json 2> $O/b2dp.err
echo "batch 2 dp: $(grep -o '"ms_per_step": [0-9.]*' $O/b2dp.json) $(grep host $O/b2dp.err)"
for q in "" 4 8; do
  if [ -z "$q" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 200 python bench.py $B > $O/single_q$q.json 2>> $O/err
  timeout 200 python bench.py $B --dp-hooks > $O/dp_q$q.json 2>> $O/err
  echo "q=$q single $(grep -o '"ms_per_step": [0-9.]*' $O/single_q$q.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/dp_q$q.json)"
done
