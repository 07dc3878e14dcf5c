#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md §HBM prescribes) into
profiles/hbm_traffic.json: HBM bytes per launch for every kernel of the bench.

Units / corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB-like units of 1024 B? No - rocprofv3 derives
FETCH_SIZE = TCC_EA0_RDREQ x 64 B expressed in KB; on gfx950 a wide coalesced streaming read is tallied at half its bytes
(128-B requests counted as 64 B), so the read side is doubled ("x2 gfx950 correction"); WRITE_SIZE is uncalibrated and used
as reported.  Both raw and corrected numbers are stored."""
import collections
import csv
import json
import sys

STAGE_OF = {"me_int_kernel": "me_integer", "me_subpel_kernel": "me_subpel", "intra_decide_kernel": "intra_candidates", "cu_decide_kernel": "cu_decide", "merge_pass_kernel": "merge_pass",
            "reconstruct_kernel": "reconstruct", "intra_recon_kernel<true>": "intra_pass", "intra_recon_kernel<false>": "key_picture_intra_pass", "deblock_kernel": "deblock", "sao_ctu_kernel": "sao"}


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("intra_recon_kernel"):             # (its two instantiations are different stages)
            name = name.split("<")[0]
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main(fetch_csv, write_csv, res, out):
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    w, _ = per_kernel(write_csv, "WRITE_SIZE")
    data = {}
    for kname, stage in STAGE_OF.items():
        if kname not in f:
            continue
        launches = 2 if stage == "deblock" else 1            # the deblock stage is two launches (vertical, horizontal)
        fetch_kb, write_kb = f[kname] * launches, w.get(kname, 0.0) * launches
        data[stage] = {"kernel": kname, "fetch_size_kb_raw": round(fetch_kb, 1), "write_size_kb_raw": round(write_kb, 1),
                       "bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024), "samples": nf[kname],
                       "note": "bytes = (2 x FETCH_SIZE [gfx950 half-count correction] + WRITE_SIZE) x 1024"}
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
    for k, v in data.items():
        if k.startswith("_"):
            continue
        print(f"{k:12s} fetch_raw {v['fetch_size_kb_raw'] / 1024:9.1f} MiB  write_raw {v['write_size_kb_raw'] / 1024:9.1f} MiB  -> {v['bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4])
