#!/bin/bash
# kernel tables of the default bench under two environments in ONE gpurun call (same box): tools/ab_profile.sh "<env A>" "<env B>" [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${3:-ab}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
for v in A B; do
  if [ $v = A ]; then E="$1"; else E="$2"; fi
  env $E timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace_$v -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/$TAG/log_$v.txt 2>&1
  python $R/tools/prof_summary.py "$(find $R/gpurun_out/$TAG/trace_$v -name '*.db' | head -1)" $R/gpurun_out/$TAG/stats_$v.md > /dev/null 2>&1
  rm -rf $R/gpurun_out/$TAG/trace_$v
  tail -1 $R/gpurun_out/$TAG/log_$v.txt | cut -c1-20,150-190
done
