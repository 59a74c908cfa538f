#!/bin/bash
# round 5, call 35: k-split of the subsampling linear layer's weight gradient ([5120 x 256] from 23.8 k rows: 0.44 ms at the default 16)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t35
mkdir -p $O
cd $R
for sp in 16 8 4 2 32; do
TFASR_LINEAR_WGRAD_SPLIT=$sp bash tools/prof_quick.sh r5_t35/p$sp > $O/p$sp.txt 2>&1
echo "split=$sp: $(grep -o '"ms_per_step": [0-9.]*' $O/p$sp/trace.log | head -1) | $(grep 'gemm_fast_kernel<true, false, 128, 32>' $O/p$sp.txt | cut -c100-150)"
done
