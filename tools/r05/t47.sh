#!/bin/bash
# round 5, call 47: LDS conflict / wait counters and the per-grid table of the FINAL build
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t47
mkdir -p $O
cd $R
bash tools/pmc_lds.sh > /dev/null 2>&1; cp gpurun_out/lds_conflicts.txt $O/lds.txt
bash tools/pmc_sqwait.sh > /dev/null 2>&1; cp gpurun_out/sq_wait.txt $O/wait.txt
head -1 $O/lds.txt | cut -c1-160; grep "relattn\|dwconv_tile" $O/lds.txt | cut -c1-160
head -1 $O/wait.txt; grep "relattn" $O/wait.txt
