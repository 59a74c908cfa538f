#!/bin/bash
# Probe build of the library: ONE source compiled with a timing define, linked with the production objects into
# tools/hwprobe/libtfasr_probe.so (load it with TFASR_LIB=<path>).  Run in the build container after `python -m tensorflowasr_amd.build`.
#   tools/build_probe_lib.sh decode_step.hip -DTFASR_DECODE_TIMING      (tools/decode_timing.sh)
#   tools/build_probe_lib.sh gemm_fast.hip   -DTFASR_FFN_TIMING         (tools/ffn_timing.sh, tools/ffn_bwd_timing.sh)
set -e
SRC=${1:?source file under tensorflowasr_amd/csrc}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$R/tensorflowasr_amd/build
BASE=$(basename "$SRC" .hip)
EXTRA=""
[ "$BASE" = logmel ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $EXTRA -Wno-unused-result "$@" -I$R/include -I$R/tensorflowasr_amd/csrc \
  -c $R/tensorflowasr_amd/csrc/$BASE.hip -o /tmp/${BASE}_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/hwprobe/libtfasr_probe.so /tmp/${BASE}_probe.o $(ls $OBJ/*.o | grep -v "/$BASE.o")
echo "built $R/tools/hwprobe/libtfasr_probe.so ($BASE with $*)"
