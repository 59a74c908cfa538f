#!/bin/bash
# per-phase clocks of the three search-step kernels (probe build: csrc/decode_step.hip with -DTFASR_DECODE_TIMING linked into
# tools/hwprobe/libtfasr_probe.so, see tools/README.md); run through gpurun from the repo root
TFASR_LIB=$PWD/tools/hwprobe/libtfasr_probe.so TFASR_DECODE_DBG_DUMP=1 timeout 300 python bench.py --mode decode --model M --steps 1 --warmup 1 2>&1 | grep decode_timing | tail -8
