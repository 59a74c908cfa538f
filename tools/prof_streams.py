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


if __name__ == "__main__":
    main()
