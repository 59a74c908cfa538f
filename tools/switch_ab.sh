#!/bin/bash
# One switch against the default, same box, interleaved: tools/switch_ab.sh "<ENV=VALUE ...>" [tag] [bench args...]
# prints `ms/step  <environment>` per run (3 pairs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
SW="$1"; TAG=${2:-switch_ab}; shift; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() {
  local ms
  ms=$(env $1 timeout 200 python bench.py --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-extras "${@:2}" 2>>$O/err.txt | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "${ms:-FAILED}  $1" | tee -a $O/results.txt
}
: > $O/results.txt
for i in $(seq 1 ${PAIRS:-3}); do run "$SW" "$@"; run "TFASR_NOP=1" "$@"; done
