#!/bin/bash
# Matrix-pipe busy cycles per kernel of the default bench command: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass)
# -> gpurun_out/mfma_busy.txt.  utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); calibrated against a kernel of known
# MFMA count (joint data gradient: flop / 16384 MFMAs x 16 clocks = the counter).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/mfma
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/mfma/run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/mfma/run.log 2>&1
DB=$(find $R/gpurun_out/mfma/run -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/mfma_busy.txt <<'PY'
import sqlite3, sys, re, collections
con = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
for name, cname, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name).split("(")[0]
    a = acc[name]
    if "MFMA" in cname: a[0] += val; a[2] += 1; a[3] += dur
    elif "GUI" in cname: a[1] += val
rows = sorted(acc.items(), key=lambda kv: -kv[1][0])
print("kernel | launches | SQ_VALU_MFMA_BUSY_CYCLES (M) | GRBM_GUI_ACTIVE / 8 (M) | MFMA utilisation | total ms under pmc")
for k, (m, g, n, d) in rows[:24]:
    print(f"{k[:90]} | {n} | {m/1e6:.1f} | {g/8e6:.2f} | {m/max(g/8*1024,1):.3f} | {d/1e6:.2f}")
PY
rm -rf $R/gpurun_out/mfma
head -26 $R/gpurun_out/mfma_busy.txt
