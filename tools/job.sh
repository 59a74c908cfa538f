#!/bin/bash
# One parameterised gpurun job (replaces the 49 one-off scripts of round 5): gpurun -- 'bash tools/job.sh <job> <tag> [args]'
#   tests  <tag> [pytest args]          GPU suite (or a selection) -> gpurun_out/<tag>/gpu_tests.txt
#   repeat <tag> <n> <pytest args>      the same selection n times in fresh processes (flakiness hunt), failure text kept
#   bench  <tag> [bench args]           the driver's command (--steps 20 --warmup 5) -> gpurun_out/<tag>/bench.json
#   ab     <tag> "<ENV=V ...>" [args]   one environment against the default, interleaved, same box (tools/switch_ab.sh)
#   multi  <tag> <rounds> "<env1>" "<env2>" ...   several environments round-robin (tools/multi_ab.sh)
#   prof   <tag> [bench args]           rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/<tag>/stats.md
#   pmc    <tag> mfma|lds|sqwait|l2hit|traffic    one counter pass over the bench command (never combined with API tracing)
R=${GRAFT_REPO_ROOT:-$(pwd)}
JOB=${1:?job}; TAG=${2:?tag}; shift; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
case $JOB in
  tests)  (time timeout 900 python -m pytest tests -m gpu -x -q --durations=5 "$@") > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt ;;
  repeat) N=$1; shift; for i in $(seq 1 $N); do timeout 300 python -m pytest -q -m gpu -x "$@" 2>&1 | grep -E "passed|failed|Error|Mismatch|Max abs|Max rel|^E  " | head -12; done | tee $O/repeat.txt ;;
  bench)  (time timeout 600 python bench.py --steps 20 --warmup 5 "$@") > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json ;;
  ab)     SW="$1"; shift; PAIRS=${PAIRS:-3} bash tools/switch_ab.sh "$SW" $TAG "$@" ;;
  multi)  RO=$1; shift; bash tools/multi_ab.sh $TAG $RO "$@" ;;
  prof)   bash tools/prof_quick.sh $TAG "$@" ;;
  pmc)    case $1 in mfma) bash tools/pmc_mfma.sh ;; lds) bash tools/pmc_lds.sh ;; sqwait) bash tools/pmc_sqwait.sh ;; l2hit) bash tools/pmc_l2hit.sh ;;
            traffic) cd /tmp && export TMPDIR=/tmp
                     for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_$c.log 2>&1; done
                     python $R/tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*.db' | head -1)" $O/pmc_traffic.json 30 > $O/pmc_top.txt 2>&1
                     rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE; head -20 $O/pmc_top.txt ;; *) echo "pmc: mfma|lds|sqwait|l2hit|traffic"; exit 2 ;; esac ;;
  *) echo "unknown job $JOB"; exit 2 ;;
esac
