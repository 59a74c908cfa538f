#!/bin/bash
# round 5, call 2: where the main queue waits in the 38-ms case (wgrad stream + auxiliary stream under a process group), priority A/B, GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t2
mkdir -p $O
cd $R
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1 TFASR_WGRAD_STREAM_PRIO=0 timeout 200 python bench.py $B --dp-hooks > $O/dp_both_prio0.json 2> $O/dp.err
TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1 TFASR_NO_PRED_STREAM=1 timeout 200 python bench.py $B --dp-hooks > $O/dp_both_nopred.json 2>> $O/dp.err
TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1 GPU_MAX_HW_QUEUES=2 timeout 200 python bench.py $B --dp-hooks > $O/dp_both_q2.json 2>> $O/dp.err
ENV="TFASR_WGRAD_STREAM=1 TFASR_BLOCK_HOIST=1" bash tools/prof_streams.sh --dp-hooks > $O/streams_both.txt 2>&1
ENV="TFASR_WGRAD_STREAM=1" bash tools/prof_streams.sh --dp-hooks > $O/streams_wgrad.txt 2>&1
for f in dp_both_prio0 dp_both_nopred dp_both_q2; do echo "== $f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json)"; done
cat $O/streams_both.txt; cat $O/streams_wgrad.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
