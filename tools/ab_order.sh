#!/bin/bash
# A/B of two libraries with the order swapped every pair (rules out "first run of a pair is faster"): tools/ab_order.sh <prev.so> [pairs]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$1; N=${2:-4}
run() { env $1 timeout 200 python bench.py --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'], '$2')"; }
for i in $(seq 1 $N); do
  if [ $((i % 2)) = 1 ]; then run TFASR_NOP=1 new; run TFASR_LIB=$P prev; else run TFASR_LIB=$P prev; run TFASR_NOP=1 new; fi
done
