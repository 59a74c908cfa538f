#!/bin/bash
# round 5, call 24: phase clocks of the transposed query-side attention backward inside the M step (probe build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so timeout 300 python tools/attn_timing.py 2>&1 | tail -12
