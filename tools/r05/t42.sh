#!/bin/bash
# round 5, call 42: depthwise conv forward / data gradient in 32-step blocks (weight gradients unchanged) - tests, kernel table, ContextNet
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t42
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_contextnet_gpu.py -x -q -m gpu 2>&1 | tail -1
bash tools/prof_quick.sh r5_t42/prof > $O/prof.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/prof/trace.log | head -1
grep -i "dwconv" $O/prof.txt | cut -c1-50,100-170 | head -6
timeout 300 python bench.py --model contextnet --alpha 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/contextnet_L.json 2> $O/contextnet.err
grep -o '"ms_per_step": [0-9.]*' $O/contextnet_L.json | head -1
