#!/bin/bash
# L2 hit rate per kernel of the default bench command: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (one pass) -> gpurun_out/l2hit.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/l2hit
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/l2hit/run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/l2hit/run.log 2>&1
DB=$(find $R/gpurun_out/l2hit/run -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/l2hit.txt <<'PY'
import sqlite3, sys, re, collections
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
q = "select kernel_name, counter_name, value, duration from counters_collection"
for name, cname, val, dur in con.execute(q):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name).split("(")[0]
    a = acc[name]
    if "HIT" in cname: a[0] += val; a[2] += 1; a[3] += dur
    elif "MISS" in cname: a[1] += val
rows = sorted(acc.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))
print("kernel | launches | hits (M) | misses (M) | hit rate | total ms under pmc")
for k, (h, m, n, d) in rows[:40]:
    print(f"{k[:90]} | {n} | {h/1e6:.1f} | {m/1e6:.1f} | {h/max(h+m,1):.3f} | {d/1e6:.2f}")
PY
rm -rf $R/gpurun_out/l2hit
head -30 $R/gpurun_out/l2hit.txt
