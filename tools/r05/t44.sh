#!/bin/bash
# round 5, call 44: block count of the fat row-reduction kernels (BatchNorm statistics): TFASR_RED_GRID cap sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t44
mkdir -p $O
cd $R
for gcap in 192 256 512 768; do
TFASR_RED_GRID=$gcap bash tools/prof_quick.sh r5_t44/p$gcap > $O/p$gcap.txt 2>&1
echo "cap=$gcap: $(grep -o '"ms_per_step": [0-9.]*' $O/p$gcap/trace.log | head -1)"
grep "bn_stats_vec_kernel\|ln_bwd_vec" $O/p$gcap.txt | sed 's/(.*`//' | cut -c1-110
done
