#!/bin/bash
# round 5, call 22 (diagnosis): the two query-backward kernels without the side stream's company
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t22
mkdir -p $O
cd $R
export TFASR_BLOCK_HOIST=0 TFASR_WGRAD_STREAM=0
for v in "1 0" "1 511" "0 0"; do
set -- $v
TFASR_ATTN_BWDQ_T=$1 TFASR_QT_DBG=$2 bash tools/prof_quick.sh r5_t22/p$1_$2 > $O/p$1_$2.txt 2>&1
echo "T=$1 dbg=$2: $(grep -i 'bwd_q' $O/p$1_$2.txt | cut -c1-30,100-170)"
done
