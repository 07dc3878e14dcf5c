#!/usr/bin/env python3
"""How much do the kernels of a rocprofv3 kernel trace overlap?  For the dispatches between bench.py's markers (else the whole trace): wall span, union of the
kernel intervals (GPU busy), sum of kernel durations, time with >= 2 kernels in flight, per queue: busy time and dispatches.  usage: trace_overlap.py results.db"""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    kd = "rocpd_kernel_dispatch" if "rocpd_kernel_dispatch" in tabs else [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = "rocpd_info_kernel_symbol" if "rocpd_info_kernel_symbol" in tabs else [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute(f"select d.start, d.end, s.kernel_name{', d.' + qcol if qcol else ''} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    marks = [r for r in rows if "ks265_marker_kernel" in r[2]]
    if len(marks) >= 2:
        lo, hi = marks[0][1], marks[-1][0]
        rows = [r for r in rows if r[0] >= lo and r[1] <= hi and "ks265_marker_kernel" not in r[2]]
    if not rows:
        print("no dispatches"); return
    span = rows[-1][1] - rows[0][0] if len(rows) > 1 else 1
    span = max(r[1] for r in rows) - rows[0][0]
    ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
    busy = over = 0; depth = 0; last = ev[0][0]
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    tot = sum(r[1] - r[0] for r in rows)
    print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms  GPU busy (union) {busy / 1e6:.3f} ms = {busy / span:.3f}  sum of kernel durations {tot / 1e6:.3f} ms  >= 2 kernels in flight {over / 1e6:.3f} ms = {over / span:.3f}")
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if nw:                                                           # the same per window of the span: where the GPU idles, where queues overlap
        t0 = rows[0][0]; w = span / nw
        acc = [[0, 0, {}] for _ in range(nw)]
        pics = [0] * nw                                              # pictures per window: one SAO launch each
        depth = 0; last = ev[0][0]
        def add(a, b, k):
            while a < b:
                i = min(nw - 1, int((a - t0) / w)); e = min(b, t0 + (i + 1) * w) if i < nw - 1 else b
                if e <= a: e = b if i >= nw - 1 else min(b, a + 1)      # (rounding at a window boundary: always advance)
                acc[i][k] += e - a; a = e
        for t, d in ev:
            if depth >= 1: add(last, t, 0)
            if depth >= 2: add(last, t, 1)
            depth += d; last = t
        if qcol:
            for r in rows:
                i = min(nw - 1, int((r[0] - t0) / w)); acc[i][2][r[3]] = acc[i][2].get(r[3], 0) + r[1] - r[0]
        for r in rows:
            if "sao_ctu_kernel" in r[2]: pics[min(nw - 1, int((r[1] - t0) / w))] += 1
        for i, (b, o, q) in enumerate(acc):
            print(f"  window {i:2d} [{i * w / 1e6:8.1f} ms]: busy {b / w:.2f}  >=2 {o / w:.2f}  pictures {pics[i]:4d} = {pics[i] / (w / 1e9):7.1f} /s  " + "  ".join(f"q{k}: {v / w:.2f}" for k, v in sorted(q.items())))
    if qcol:
        q = {}
        for r in rows:
            a = q.setdefault(r[3], [0, 0]); a[0] += 1; a[1] += r[1] - r[0]
        for k, (n, t) in sorted(q.items(), key=lambda kv: -kv[1][1]):
            print(f"  {qcol} {k}: {n} dispatches, {t / 1e6:.3f} ms of kernels")


if __name__ == "__main__":
    main(sys.argv[1])
