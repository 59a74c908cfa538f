#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/gaps
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/gaps/trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras $PROF_GAPS_ARGS > $R/gpurun_out/gaps/trace.log 2>&1
DB=$(find $R/gpurun_out/gaps/trace -name "*.db" | head -1)
python $R/tools/prof_gaps.py "$DB" > $R/gpurun_out/gaps.txt 2>&1
rm -rf $R/gpurun_out/gaps
cat $R/gpurun_out/gaps.txt
