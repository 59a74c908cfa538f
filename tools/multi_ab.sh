#!/bin/bash
# several environments round-robin on one box: tools/multi_ab.sh tag rounds "ENV1=.." "ENV2=.." ...   (prints ms/step per run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ROUNDS=$2; shift; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
: > $O/results.txt
for i in $(seq 1 $ROUNDS); do
  for e in "$@"; do
    ms=$(env $e timeout 200 python bench.py --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-extras $BENCH_ARGS 2>>$O/err.txt | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "${ms:-FAILED}  $e" | tee -a $O/results.txt
  done
done
