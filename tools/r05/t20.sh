#!/bin/bash
# round 5 (diagnosis, temporary debug mask in relattn_fused_bwd_qT_kernel): which part of the key-block loop costs the time
# bits: 1 dS store, 2 -, 4 exp/dS/scatter, 8 dq_v products, 16 score readback, 32 S/dP products, 64 G tiles, 128 dq_u products, 256 DMA
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t20
mkdir -p $O
cd $R
for d in 0 256 1 224 232 20 511; do
TFASR_QT_DBG=$d bash tools/prof_quick.sh r5_t20/p$d > $O/p$d.txt 2>&1
echo "dbg=$d: $(grep -i 'bwd_qT' $O/p$d.txt | cut -c100-170)"
done
