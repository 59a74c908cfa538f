#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_attn_bwdq_t_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -12
