#!/bin/bash
# round 5, call 18: why is the joint-recompute leg of the default bench line 3.4 ms slower than the headline?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t18
mkdir -p $O
cd $R
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/a.json 2>> $O/err
TFASR_JOINT_RECOMPUTE=1 timeout 200 python bench.py $B > $O/b.json 2>> $O/err
TFASR_JOINT_RECOMPUTE=1 TFASR_BIG_TR=0 timeout 200 python bench.py $B > $O/c.json 2>> $O/err
TFASR_JOINT_RECOMPUTE=1 TFASR_DEFER_SIDE=0 timeout 200 python bench.py $B > $O/d.json 2>> $O/err
TFASR_JOINT_RECOMPUTE=1 TFASR_FRONT_EARLY=0 timeout 200 python bench.py $B > $O/e.json 2>> $O/err
for f in a b c d e; do echo "$f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1)"; done
TFASR_JOINT_RECOMPUTE=1 bash tools/prof_quick.sh r5_t18/prof > $O/prof.txt 2>&1
head -16 $O/prof.txt | cut -c1-150
