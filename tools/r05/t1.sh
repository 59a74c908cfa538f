#!/bin/bash
# round 5, call 1: baseline line, data-parallel route (default / both streams forced), per-queue trace of the forced case
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t1
mkdir -p $O
cd $R
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/base.json 2> $O/base.err
timeout 200 python bench.py $B --dp-hooks > $O/dp.json 2> $O/dp.err
TFASR_WGRAD_STREAM=1 timeout 200 python bench.py $B --dp-hooks > $O/dp_wgrad.json 2>> $O/dp.err
TFASR_BLOCK_HOIST=1 timeout 200 python bench.py $B --dp-hooks > $O/dp_hoist.json 2>> $O/dp.err
TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1 timeout 200 python bench.py $B --dp-hooks > $O/dp_both.json 2>> $O/dp.err
ENV="TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1" bash tools/prof_streams.sh --dp-hooks > $O/streams_both.txt 2>&1
ENV="TFASR_WGRAD_STREAM=1" bash tools/prof_streams.sh --dp-hooks > $O/streams_wgrad.txt 2>&1
for f in base dp dp_wgrad dp_hoist dp_both; do echo "== $f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json)"; done
cat $O/streams_both.txt; cat $O/streams_wgrad.txt
