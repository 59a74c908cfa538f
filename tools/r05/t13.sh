#!/bin/bash
# round 5, call 13: front end of the next step beside the previous step's tail (prefetched inputs); whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t13
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for v in 0 1 0 1; do
  TFASR_FRONT_EARLY=$v timeout 200 python bench.py $B > $O/s.json 2>> $O/err
  TFASR_FRONT_EARLY=$v timeout 200 python bench.py $B --dp-hooks > $O/d.json 2>> $O/err
  TFASR_FRONT_EARLY=$v timeout 200 python bench.py $B --model S > $O/S.json 2>> $O/err
  echo "front early=$v: single $(grep -o '"ms_per_step": [0-9.]*' $O/s.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/d.json) | S $(grep -o '"ms_per_step": [0-9.]*' $O/S.json)"
done
tail -2 $O/err
