#!/bin/bash
# round 5, call 21: LDS conflict and wait-share counters of the attention kernels, transposed vs row-oriented query backward
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t21
mkdir -p $O
cd $R
for t in 1 0; do
export TFASR_ATTN_BWDQ_T=$t
bash tools/pmc_lds.sh > /dev/null 2>&1; cp gpurun_out/lds_conflicts.txt $O/lds_T$t.txt
bash tools/pmc_sqwait.sh > /dev/null 2>&1; cp gpurun_out/sq_wait.txt $O/wait_T$t.txt
echo "== T=$t"; head -1 $O/lds_T$t.txt | cut -c1-200; grep relattn $O/lds_T$t.txt | cut -c1-200; head -1 $O/wait_T$t.txt; grep relattn $O/wait_T$t.txt
done
