#!/bin/bash
# round 5, call 29: which layer shapes hide inside the per-name averages of the GEMM kernels (per grid size and queue)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t29
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $R/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/prof_bygrid.py "$DB" "gemm|ffn|wgrad|ln_|relattn" $O/bygrid.md > /dev/null
python $R/tools/prof_streams.py "$DB" > $O/streams.txt 2>&1
rm -rf $O/trace
head -64 $O/bygrid.md | cut -c1-160
head -12 $O/streams.txt | cut -c1-200
