"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / share.  Usage: prof_summary.py db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += (e - s)
    tot = sum(v[1] for v in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        lines.append(f"| `{k}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.1f} | {100.0 * t / tot:.1f} |")
    lines.append(f"\ntotal kernel time {tot / 1e6:.2f} ms over {sum(v[0] for v in agg.values())} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
