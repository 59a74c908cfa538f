#!/bin/bash
# HIP / ROCr runtime switches against the default environment, same box, interleaved (usage through gpurun: bash tools/env_ab.sh [tag]).
# Every line is `ms/step  <environment>`; the default environment is measured first, in the middle and last (box drift).
# What is being asked: a step is ~1000 dependent launches, so anything the command processor does per dispatch (kernel-argument fetch,
# cache flush scope at kernel boundaries, scratch re-allocation, completion signalling) is paid a thousand times per step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-env_ab}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() {  # run "<env assignments>"
  local ms
  ms=$(env $1 timeout 200 python bench.py --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-extras 2>>$O/err.txt | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "${ms:-FAILED}  ${1:-<default>}" | tee -a $O/results.txt
}
: > $O/results.txt
run "TFASR_NOP=1"
for e in "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_NO_SCRATCH_RECLAIM=1" "AMD_OPT_FLUSH=0" "DEBUG_HIP_KERNARG_COPY_OPT=0"; do run "$e"; done
run "TFASR_NOP=1"
for e in "HSA_ENABLE_INTERRUPT=0" "GPU_FLUSH_ON_EXECUTION=1" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1"; do run "$e"; done
run "TFASR_NOP=1"
if [ -n "$TFASR_LIB_B" ]; then  # a second build of the library (tools/build_probe_lib.sh or a copy of an older .so)
  for i in 1 2 3; do run "TFASR_LIB=$TFASR_LIB_B"; run "TFASR_NOP=1"; done
fi
