#!/bin/bash
# round 5, call 36: the default bench line again with the recompute leg first of the extra legs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_final2
mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_M.json 2> $O/bench_M.err
cut -c1-300 $O/bench_M.json; grep -o '"joint_recompute_variant": {"ms_per_step": [0-9.]*' $O/bench_M.json; grep -o '"reference_padding": {"ms_per_step": [0-9.]*' $O/bench_M.json; grep -o '"dp_route_ms": [0-9.]*' $O/bench_M.json
