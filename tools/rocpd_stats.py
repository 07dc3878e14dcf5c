#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into a per-kernel stats table
(the same columns as `--stats` CSV: calls, total, average, min, max, percentage).
If the trace holds ks265_marker_kernel dispatches (bench.py brackets its timed region with them), only the dispatches
between the first and the last marker are counted, so the averages are those of the timed region itself."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    kd = "rocpd_kernel_dispatch" if "rocpd_kernel_dispatch" in tabs else [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = "rocpd_info_kernel_symbol" if "rocpd_info_kernel_symbol" in tabs else [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    marks = c.execute(f"select d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like '%ks265_marker_kernel%' order by d.start").fetchall()
    where, note = "", "whole run"
    if len(marks) >= 2:
        where = f" where d.start >= {marks[0][1]} and d.end <= {marks[-1][0]} "
        note = f"timed region only: {len(marks)} markers, {(marks[-1][0] - marks[0][1]) / 1e6:.3f} ms between the first and the last"
    rows = c.execute(
        f"select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id {where} group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"# {note}", f"{'kernel':<72} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}"]
    for name, n, tot, avg, mn, mx in rows:
        lines.append(f"{name[:72]:<72} {n:>7} {tot / 1e3:>12.1f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100.0 * tot / total:>6.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
