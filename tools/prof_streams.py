"""Per-queue view of a rocprofv3 kernel trace (rocpd sqlite): which kernels ran on which hardware queue / stream, how busy each queue was and
how much of it overlapped the busiest one - the question behind the data-parallel regression of round 4 (weight-gradient stream + auxiliary
stream under a process group: 38 instead of 22.7 ms).  Usage: prof_streams.py trace.db"""
import collections
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    if qcol is None:
        print("no queue / stream column in `kernels`:", cols)
        return
    rows = con.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
    rows = rows[int(len(rows) * 0.4):]  # timed steps, not warm-up
    span = rows[-1][2] - rows[0][1]
    by_q = collections.defaultdict(list)
    for name, s, e, q in rows:
        by_q[q].append((s, e, name))
    main_q = max(by_q, key=lambda q: sum(e - s for s, e, _ in by_q[q]))
    main_iv = sorted((s, e) for s, e, _ in by_q[main_q])
    print(f"{len(rows)} dispatches over {span / 1e6:.2f} ms on {len(by_q)} queues ({qcol}); busiest = {main_q}")
    for q, ks in sorted(by_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        ov, j = 0, 0
        for s, e, _ in sorted(ks):  # overlap with the busiest queue's kernels
            while j < len(main_iv) and main_iv[j][1] <= s:
                j += 1
            k = j
            while k < len(main_iv) and main_iv[k][0] < e:
                ov += min(e, main_iv[k][1]) - max(s, main_iv[k][0])
                k += 1
        top = collections.Counter()
        for s, e, n in ks:
            top[n.split("(")[0][:60]] += e - s
        names = ", ".join(f"{n} {t / 1e6:.2f} ms" for n, t in top.most_common(3))
        print(f"  queue {q}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.0f} % of the span), "
              f"{100.0 * ov / max(busy, 1):.0f} % of it beside queue {main_q}'s kernels | {names}")
    # where the busiest queue WAITS: its idle gaps by (kernel before -> kernel after), and what the other queues ran inside the largest ones
    def nm(x):
        import re
        return re.sub(r"\(anonymous namespace\)::", "", x).split("(")[0].replace("void ", "")[:48]
    mk = sorted(by_q[main_q])
    gaps = [(mk[i + 1][0] - mk[i][1], i) for i in range(len(mk) - 1) if mk[i + 1][0] > mk[i][1]]
    tot = sum(g for g, _ in gaps)
    print(f"queue {main_q} idle {tot / 1e6:.2f} ms in {len(gaps)} gaps; by (before -> after), gaps >= 20 us:")
    agg = collections.defaultdict(lambda: [0, 0])
    for g, i in gaps:
        if g >= 20000:
            k = nm(mk[i][2]) + " -> " + nm(mk[i + 1][2])
            agg[k][0] += g
            agg[k][1] += 1
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"   {t / 1e3:9.1f} us in {n:3d} gaps (avg {t / n / 1e3:7.1f})  {k}")
    others = sorted((s, e, n, q) for q, ks in by_q.items() if q != main_q for s, e, n in ks)
    for g, i in sorted(gaps, reverse=True)[:4]:
        lo, hi = mk[i][1], mk[i + 1][0]
        inside = [(s, e, n, q) for s, e, n, q in others if e > lo and s < hi]
        print(f"--- {g / 1e3:.1f} us gap after {nm(mk[i][2])} before {nm(mk[i + 1][2])}: {len(inside)} kernels of other queues inside")
        for s, e, n, q in inside[:3] + ([("...",) * 4] if len(inside) > 6 else []) + inside[-3:] if len(inside) > 6 else inside:
            if s == "...":
                print("      ...")
            else:
                print(f"      q{q} +{(s - lo) / 1e3:8.1f} us .. +{(e - lo) / 1e3:8.1f} us  {nm(n)}")


if __name__ == "__main__":
    main()
