#!/bin/bash
# Kernel-trace profile of the default bench command -> gpurun_out/$1/stats.md (usage: tools/prof_quick.sh <tag> [bench args])
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-quick}; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -- python $R/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/$TAG/trace.log 2>&1
DB=$(find $R/gpurun_out/$TAG/trace -name "*.db" | head -1)
python $R/tools/prof_summary.py "$DB" $R/gpurun_out/$TAG/stats.md > /dev/null 2>&1
tail -1 $R/gpurun_out/$TAG/trace.log | cut -c1-200
rm -rf $R/gpurun_out/$TAG/trace
head -50 $R/gpurun_out/$TAG/stats.md
