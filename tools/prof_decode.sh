#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/dec
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dec/trace -- python $R/bench.py --mode decode --model M --steps 3 --warmup 1 > $R/gpurun_out/dec/trace.log 2>&1
DB=$(find $R/gpurun_out/dec/trace -name "*.db" | head -1)
python $R/tools/prof_summary.py "$DB" $R/gpurun_out/dec/stats.md > /dev/null 2>&1
rm -rf $R/gpurun_out/dec/trace
head -24 $R/gpurun_out/dec/stats.md
