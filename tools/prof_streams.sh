#!/bin/bash
# Per-queue kernel timeline of a bench command (usage through gpurun: ENV="TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1" bash tools/prof_streams.sh --dp-hooks)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/streams
env $ENV timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/streams/trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/streams/trace.log 2>&1
DB=$(find $R/gpurun_out/streams/trace -name "*.db" | head -1)
python $R/tools/prof_streams.py "$DB" | tee $R/gpurun_out/streams.txt
rm -rf $R/gpurun_out/streams
