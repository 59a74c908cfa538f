"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd sqlite): union of busy intervals over all streams vs the
span of each step-sized window; histogram of the gaps and the kernels that precede the largest ones.  Usage: prof_gaps.py db"""
import collections
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    # keep the last 60 % of the trace (timed steps, not warm-up / setup)
    rows = rows[int(len(rows) * 0.4):]
    span = rows[-1][2] - rows[0][1]
    busy, gaps = 0, []
    cur_s, cur_e = rows[0][1], rows[0][2]
    prev_name = rows[0][0]
    for name, s, e in rows[1:]:
        if s <= cur_e:
            if e > cur_e:
                cur_e, prev_name = e, name
        else:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev_name, name))
            cur_s, cur_e, prev_name = s, e, name
    busy += cur_e - cur_s
    print(f"{len(rows)} dispatches over {span / 1e6:.2f} ms: GPU busy (union over streams) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %, idle {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps")
    h = collections.Counter()
    for g, _, _ in gaps:
        h[min(int(g / 1000), 20)] += 1
    print("gap histogram (us: count):", dict(sorted(h.items())))
    by = collections.defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        k = re.sub(r"\(anonymous namespace\)::", "", a).split("(")[0][:50] + " -> " + re.sub(r"\(anonymous namespace\)::", "", b).split("(")[0][:50]
        by[k][0] += g
        by[k][1] += 1
    for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{t / 1e3:9.1f} us in {n:4d} gaps (avg {t / n / 1e3:5.2f})  {k}")
    # context of the largest gaps: the dispatches around them in global time order (all streams)
    def nm(x):
        return re.sub(r"\(anonymous namespace\)::", "", x).split("(")[0][:60]
    big = sorted(range(1, len(rows)), key=lambda i: -(rows[i][1] - max(r[2] for r in rows[max(0, i - 8):i])))[:3]
    for i in sorted(big):
        prev_end = max(r[2] for r in rows[max(0, i - 8):i])
        print(f"--- gap of {(rows[i][1] - prev_end) / 1e3:.1f} us before dispatch {i}:")
        for j in range(max(0, i - 6), min(len(rows), i + 5)):
            print(f"   {'>>' if j == i else '  '} +{(rows[j][1] - rows[i][1]) / 1e3:9.1f} us  {(rows[j][2] - rows[j][1]) / 1e3:7.1f} us  {nm(rows[j][0])}")


if __name__ == "__main__":
    main()
