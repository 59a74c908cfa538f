#!/bin/bash
# flakiness hunt: the hoist test many times (fresh process each), failure text kept
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in $(seq 1 14); do
timeout 120 python -m pytest tests/test_attn_bwdq_t_gpu.py tests/test_block_hoist_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|Mismatch|Max abs|Max rel|err_msg|^E  " | head -12
done
