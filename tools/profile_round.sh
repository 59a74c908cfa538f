#!/bin/bash
# Round-end profile of the default bench command: rocprofv3 kernel trace (summary -> gpurun_out/final/stats.md) and the two
# PMC passes for HBM traffic (-> gpurun_out/final/pmc.json).  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/final
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/final/trace.log 2>&1
DB=$(find $R/gpurun_out/final/trace -name "*.db" | head -1)
python $R/tools/prof_summary.py "$DB" $R/gpurun_out/final/stats.md > /dev/null 2>&1
tail -1 $R/gpurun_out/final/trace.log | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/final/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/final/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py "$(find $R/gpurun_out/final/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $R/gpurun_out/final/pmc_WRITE_SIZE -name '*.db' | head -1)" $R/gpurun_out/final/pmc.json 50 2>&1 | head -5
rm -rf $R/gpurun_out/final/trace $R/gpurun_out/final/pmc_FETCH_SIZE $R/gpurun_out/final/pmc_WRITE_SIZE
head -12 $R/gpurun_out/final/stats.md
