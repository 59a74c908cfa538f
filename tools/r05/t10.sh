#!/bin/bash
# round 5, call 10: would two half-batch chains side by side fill the chip? (two independent processes, 16 utterances each, one GPU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t10
mkdir -p $O
cd $R
B="--steps 30 --warmup 8 --no-cpu-baseline --no-extras"
timeout 200 python bench.py $B > $O/b32.json 2>> $O/err
timeout 200 python bench.py $B --batch 16 > $O/b16.json 2>> $O/err
echo "alone: B=32 $(grep -o '"ms_per_step": [0-9.]*' $O/b32.json) | B=16 $(grep -o '"ms_per_step": [0-9.]*' $O/b16.json)"
timeout 300 python bench.py $B --batch 16 > $O/c1.json 2>> $O/err &
P1=$!
timeout 300 python bench.py $B --batch 16 > $O/c2.json 2>> $O/err &
P2=$!
wait $P1 $P2
echo "two B=16 processes side by side: $(grep -o '"ms_per_step": [0-9.]*' $O/c1.json) | $(grep -o '"ms_per_step": [0-9.]*' $O/c2.json)"
timeout 300 python bench.py $B --batch 16 > $O/d1.json 2>> $O/err &
P1=$!
timeout 300 python bench.py $B --batch 16 > $O/d2.json 2>> $O/err &
P2=$!
wait $P1 $P2
echo "again: $(grep -o '"ms_per_step": [0-9.]*' $O/d1.json) | $(grep -o '"ms_per_step": [0-9.]*' $O/d2.json)"
tail -2 $O/err
