#!/bin/bash
# LDS bank conflicts per kernel of the default bench command: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
# SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS (one pass) -> gpurun_out/lds_conflicts.txt.  conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
# (extra LDS-array cycles over all LDS-array cycles, MI355X_MICROARCH LDS section).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/ldsq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS -d $R/gpurun_out/ldsq/run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/ldsq/run.log 2>&1
DB=$(find $R/gpurun_out/ldsq/run -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/lds_conflicts.txt <<'PY'
import sqlite3, sys, re, collections
con = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for name, cname, val, d in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name).split("(")[0]
    acc[name][cname] += val
    if cname == "SQ_LDS_IDX_ACTIVE": dur[name] += d
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0.0))
print("kernel | ms under pmc | LDS_IDX_ACTIVE (M) | BANK_CONFLICT (M) | conflict share | ADDR_CONFLICT (M) | UNALIGNED_STALL (M) | INSTS_LDS (M) | array cycles per LDS instruction")
for k, c in rows[:30]:
    a = max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    print(f"{k[:80]} | {dur[k]/1e6:.2f} | {a/1e6:.1f} | {c.get('SQ_LDS_BANK_CONFLICT',0)/1e6:.1f} | {c.get('SQ_LDS_BANK_CONFLICT',0)/a:.3f} | "
          f"{c.get('SQ_LDS_ADDR_CONFLICT',0)/1e6:.1f} | {c.get('SQ_LDS_UNALIGNED_STALL',0)/1e6:.1f} | {c.get('SQ_INSTS_LDS',0)/1e6:.1f} | {a/max(c.get('SQ_INSTS_LDS',0),1):.2f}")
PY
tail -3 $R/gpurun_out/ldsq/run.log | cut -c1-200 >> $R/gpurun_out/lds_conflicts.txt
rm -rf $R/gpurun_out/ldsq
head -34 $R/gpurun_out/lds_conflicts.txt
