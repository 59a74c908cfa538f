#!/bin/bash
# round 5, call 37: the subsampling linear layer's weight gradient through the grouped weight-gradient kernel (row bands) vs the split-K GEMM
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t37
mkdir -p $O
cd $R
TFASR_LINEAR_WGRAD_GROUP=2 timeout 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "step_matches_oracle or bf16_step" 2>&1 | tail -2
for nh in 0 2 4; do
TFASR_LINEAR_WGRAD_GROUP=$nh bash tools/prof_quick.sh r5_t37/p$nh > $O/p$nh.txt 2>&1
echo "bands=$nh: $(grep -o '"ms_per_step": [0-9.]*' $O/p$nh/trace.log | head -1)"
grep 'gemm_fast_kernel<true, false, 128, 32>\|wgrad_group_kernel' $O/p$nh.txt | sed 's/(.*`//' | cut -c1-120
done
