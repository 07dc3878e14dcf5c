#!/usr/bin/env python3
"""Print a window of a rocprofv3 rocpd kernel trace as a timeline: per queue the busy time, then every dispatch (start offset, duration, queue, short name).
usage: rocpd_timeline.py results.db [start_fraction=0.6] [window_ms=30]"""
import re
import sqlite3
import sys


def main(db, frac=0.6, win_ms=30.0):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    kd = "rocpd_kernel_dispatch" if "rocpd_kernel_dispatch" in tabs else [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = "rocpd_info_kernel_symbol" if "rocpd_info_kernel_symbol" in tabs else [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    q = "d.stream_id" if "stream_id" in cols else "d.queue_id"
    t0, t1 = c.execute(f"select min(start), max(end) from {kd}").fetchone()
    a = t0 + (t1 - t0) * frac
    b = a + win_ms * 1e6
    rows = c.execute(f"select d.start, d.end, {q}, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id where d.start >= {a} and d.start < {b} order by d.start").fetchall()
    busy = {}
    for s, e, qi, _ in rows:
        busy[qi] = busy.get(qi, 0) + (e - s)
    ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
    depth, last, union = 0, a, 0
    for t, d in ev:
        if depth > 0:
            union += t - last
        depth += d; last = t
    print(f"# window {win_ms} ms at {frac:.2f} of the run ({q}); {len(rows)} dispatches; some kernel running {union / 1e6:.2f} ms; busy ms per queue: " + ", ".join(f"{k}: {v / 1e6:.2f}" for k, v in sorted(busy.items())))
    for s, e, qi, name in rows:
        m = re.match(r"_Z\d+([A-Za-z0-9_]+?)(?:I|\d|P|v|$)", name)
        short = name[:40]
        mm = re.match(r"_Z(\d+)", name)
        if mm:
            n = int(mm.group(1)); short = name[2 + len(mm.group(1)):2 + len(mm.group(1)) + n]
        print(f"{(s - a) / 1e3:10.1f} {(e - s) / 1e3:9.1f} q{qi:<4} {short}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.6, float(sys.argv[3]) if len(sys.argv) > 3 else 30.0)
