#!/bin/bash
# round 5, call 3: hardware-queue sweep of the 38-ms case and of the single-GPU line; bench.py's new keys
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t3
mkdir -p $O
cd $R
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for q in 1 2 3 4 5 6 8 12; do
  GPU_MAX_HW_QUEUES=$q TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1 timeout 200 python bench.py $B --dp-hooks > $O/dp_both_q$q.json 2>> $O/dp.err
  echo "dp both q=$q: $(grep -o '"ms_per_step": [0-9.]*' $O/dp_both_q$q.json)"
done
for q in 2 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$q TFASR_BENCH_HOST=1 timeout 200 python bench.py $B > $O/single_q$q.json 2> $O/single_q$q.err
  echo "single q=$q: $(grep -o '"ms_per_step": [0-9.]*' $O/single_q$q.json) $(grep host $O/single_q$q.err)"
done
for q in 2 4; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py $B --dp-hooks > $O/dp_q$q.json 2>> $O/dp.err
  echo "dp default q=$q: $(grep -o '"ms_per_step": [0-9.]*' $O/dp_q$q.json)"
done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python -c "
import json,sys
d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('ms_per_step','launches_per_step','dp_route_ms','dp_route_over_single')})
print(d.get('roofline_by_time'))"
tail -3 $O/bench_full.err
