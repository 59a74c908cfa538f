#!/bin/bash
# round 5, call 30: host queues the main stream's subsampling before the other streams' first launches (TFASR_MAIN_FIRST) - tests, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t30
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_block_hoist_gpu.py tests/test_model_gpu.py tests/test_lstm_persist_gpu.py -x -q -m gpu 2>&1 | tail -2
B="--steps 30 --warmup 5 --no-cpu-baseline --no-extras"
for i in 1 2; do
TFASR_MAIN_FIRST=0 timeout 200 python bench.py $B > $O/old$i.json 2>> $O/err
TFASR_MAIN_FIRST=1 timeout 200 python bench.py $B > $O/new$i.json 2>> $O/err
done
TFASR_MAIN_FIRST=1 timeout 200 python bench.py $B --dp-hooks > $O/dp.json 2>> $O/err
for f in old1 new1 old2 new2 dp; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
