#!/bin/bash
# round 5, call 20 (diagnosis, temporary debug mask in relattn_fused_bwd_qT_kernel): which phase of the key-block loop costs the time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t20
mkdir -p $O
cd $R
for d in 0 63 127 255 511 287 256; do
TFASR_QT_DBG=$d bash tools/prof_quick.sh r5_t20/p$d > $O/p$d.txt 2>&1
echo "dbg=$d: $(grep -i 'bwd_qT' $O/p$d.txt | cut -c100-170)"
done
