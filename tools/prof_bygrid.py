"""Per (kernel, grid size, queue) view of a rocprofv3 kernel trace (rocpd sqlite): the same kernel template serves several layer shapes,
and a bad shape hides inside the per-name average of prof_summary.py.  Usage: prof_bygrid.py trace.db [name-regex] [out.md]"""
import collections
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    gcols = [c for c in cols if "grid" in c.lower()]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    sel = ", ".join(["name", "start", "end"] + gcols + ([qcol] if qcol else []))
    rows = con.execute(f"select {sel} from kernels order by start").fetchall()
    rows = rows[int(len(rows) * 0.4):]  # timed steps, not warm-up
    agg = collections.defaultdict(lambda: [0, 0])
    for row in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", row[0]))
        if not pat.search(name):
            continue
        key = (name.split("(")[0][:70],) + tuple(row[3:])
        agg[key][0] += 1
        agg[key][1] += row[2] - row[1]
    lines = [f"columns: {gcols} {qcol}", "| kernel | " + " | ".join(gcols + ([qcol] if qcol else [])) + " | calls | total ms | avg us |", "|---|" + "---:|" * (len(gcols) + (1 if qcol else 0) + 3)]
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        lines.append(f"| `{k[0]}` | " + " | ".join(str(x) for x in k[1:]) + f" | {n} | {t / 1e6:.3f} | {t / n / 1e3:.1f} |")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
