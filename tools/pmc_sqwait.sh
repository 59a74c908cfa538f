#!/bin/bash
# Where the waves of each kernel spend their cycles: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
# SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY over the default bench command -> gpurun_out/sq_wait.txt (shares of SQ_WAVE_CYCLES)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/sqw
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY -d $R/gpurun_out/sqw/run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/sqw/run.log 2>&1
DB=$(find $R/gpurun_out/sqw/run -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/sq_wait.txt <<'PY'
import sqlite3, sys, re, collections
con = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for name, cname, val, d in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name).split("(")[0]
    acc[name][cname] += val
    if cname == "SQ_WAVE_CYCLES": dur[name] += d
rows = sorted(acc.items(), key=lambda kv: -dur[kv[0]])
cols = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"]
print("shares of SQ_WAVE_CYCLES | kernel | ms under pmc | " + " | ".join(c.replace("SQ_", "") for c in cols))
for k, c in rows[:28]:
    w = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print(f"{k[:80]} | {dur[k]/1e6:.2f} | " + " | ".join(f"{c.get(x, 0.0)/w:.2f}" for x in cols))
PY
rm -rf $R/gpurun_out/sqw
head -30 $R/gpurun_out/sq_wait.txt
