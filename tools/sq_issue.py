#!/usr/bin/env python3
"""profiles/sq_counters.json from a rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM ... pass (csv) of `bench.py --leg hot --streams 1`:
per stage of a P picture the waves and the wave-instructions per launch by class.  bench.py turns SQ_INSTS_VALU into the VALU-issue fraction of a kernel
(wave-instructions / 1.23e12 per second / its duration), the second roofline figure next to the HBM one.
usage: sq_issue.py counter_collection.csv 3840x2160 profiles/sq_counters.json"""
import collections
import csv
import json
import sys

STAGE_OF = {"me_int_kernel": "me_integer", "me_subpel_kernel": "me_subpel", "intra_decide_kernel": "intra_candidates", "cu_decide_kernel": "cu_decide", "merge_pass_kernel": "merge_pass",
            "reconstruct_kernel": "reconstruct", "intra_recon_kernel<true>": "intra_pass", "intra_recon_kernel<false>": "key_picture_intra_pass", "deblock_kernel": "deblock", "sao_ctu_kernel": "sao"}


def main(path, res, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("intra_recon_kernel"):
            name = name.split("<")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[name] += 1
    data = {}
    for kname, stage in STAGE_OF.items():
        if not cnt[kname]:
            continue
        v, n = agg[kname], cnt[kname]
        launches = 2 if stage == "deblock" else 1
        data[stage] = {"kernel": kname, "samples": n, "waves_per_launch": round(v["SQ_WAVES"] / n * launches)}
        for c in sorted(v):
            if c.startswith("SQ_INSTS_"):
                data[stage][c.lower().replace("sq_", "") + "_per_launch"] = round(v[c] / n * launches)
    try:
        allres = json.load(open(out))
    except Exception:
        allres = {}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ks265codec_amd.build import source_sha
    data["_stamp"] = {"kernel_src_sha": source_sha(), "git_head": os.environ.get("KS265_GIT_HEAD", "")}      # what these counters were measured on
    allres[res] = data
    json.dump(allres, open(out, "w"), indent=1)
    for k, d in data.items():
        print(k.ljust(24), d)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
