#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc SQ_* pass (csv): per-wave averages, for diagnosis (what a kernel waits on)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(40), "launches", " ".join(n.replace("SQ_", "").rjust(16) for n in names), "  (per wave, except WAVES = per launch)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if not cnt[k] or not v.get("SQ_WAVES"):
        continue
    w = v["SQ_WAVES"]
    print(k.ljust(40), str(cnt[k]).rjust(8), " ".join((f"{v[n] / cnt[k]:16.0f}" if n == "SQ_WAVES" else f"{v[n] / w:16.1f}") for n in names))
