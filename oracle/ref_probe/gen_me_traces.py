#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): record the reference's integer-pel search functions on real encodes and write
tests/golden/me_search.npz (pin interMeDia enc@0x48fbe0 / interMeHex enc@0x48fde0 / interMeUMH enc@0x4907b0).

Pass 1 runs `appencoder` under me_trace_shim.so in log mode (one text line per call), pass 2 re-runs it (the encoder is deterministic
at -threads 1) and dumps the selected calls: per call the source block, the mvd-cost table slices, start point / cost / merange / limits
and the function's result; per reference picture the plane region the calls can reach.  The fixture holds DATA only (pixels of synthetic
clips and of the encoder's reconstructions, motion vectors, costs).
"""
from __future__ import annotations

import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/ubuntu_x64/appencoder"

# (name, clip kwargs, frames, encoder args).  Clips with fast / irregular motion and noise so that the long paths of the searches are taken.
RUNS = [
    ("dia", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 3, ["-preset", "slow", "-me", "0", "-qp", "27"]),
    ("dia_fast", dict(seed=77, abc=(9, 11, 5), pan=(13, 9)), 3, ["-preset", "veryfast", "-me", "0", "-qp", "37"]),
    ("hex", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 3, ["-preset", "veryfast", "-me", "1", "-qp", "32"]),
    ("hex_fast", dict(seed=78, abc=(9, 11, 5), pan=(14, 7)), 3, ["-preset", "slow", "-me", "1", "-qp", "27"]),
    ("umh", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 3, ["-preset", "slow", "-me", "2", "-qp", "27"]),
    ("umh_fast", dict(seed=79, abc=(9, 11, 5), pan=(15, 10)), 4, ["-preset", "slow", "-me", "2", "-qp", "32"]),
    ("umh_b", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 5, ["-preset", "slow", "-me", "2", "-qp", "27", "-bframes", "3"]),
]
W, H = 416, 240
PER_RUN = 140          # calls kept per run (all UMH calls are kept first)


def parse_log(path):
    rows = []
    for line in open(path):
        t = line.split()
        rows.append(dict(idx=int(t[0]), m=int(t[1][1:]), w=int(t[2][1:]), h=int(t[3][1:]), pux=int(t[5]), puy=int(t[6]), sx=int(t[12]), sy=int(t[13]),
                         lim=[int(t[19]), int(t[20]), int(t[21]), int(t[22])], ox=int(t[34]), oy=int(t[35])))
    return rows


def main():
    from ks265codec_amd.synth import make_clip
    tmp = tempfile.mkdtemp(prefix="ks265me_")
    try:
        shutil.copy(REF, tmp)
        enc = os.path.join(tmp, "appencoder")
        os.chmod(enc, 0o755)
        shim = os.path.join(tmp, "me_trace.so")
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "me_trace_shim.c")])
        planes, plane_meta, calls, fenc_blob, cm_blob = [], [], [], [], []
        rng = np.random.default_rng(1)
        for name, ckw, frames, args in RUNS:
            clip = make_clip(W, H, frames, **ckw)
            yuv = os.path.join(tmp, "in.yuv")
            clip.tofile(yuv)
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-rc", "0", "-iper", "128", "-threads", "1", "-b", os.path.join(tmp, "o.265"), *args]
            if "-bframes" not in args:
                cmd += ["-bframes", "0"]
            log = os.path.join(tmp, "log.txt")
            subprocess.run(cmd, env=dict(os.environ, LD_PRELOAD=shim, KS265_ME_LOG=log), capture_output=True, check=True, cwd=tmp)
            rows = parse_log(log)
            # selection: every UMH call, every call that moved more than 2 pels, then a random fill
            moved = [r for r in rows if abs(r["ox"] - r["sx"]) + abs(r["oy"] - r["sy"]) > 2]
            umh = [r for r in rows if r["m"] == 2]
            pick = {r["idx"] for r in umh[:PER_RUN]}
            for r in moved:
                if len(pick) < PER_RUN:
                    pick.add(r["idx"])
            rest = [r["idx"] for r in rows if r["idx"] not in pick]
            rng.shuffle(rest)
            for i in rest:
                if len(pick) >= PER_RUN:
                    break
                pick.add(i)
            sel = np.zeros(len(rows), np.uint8)
            sel[list(pick)] = 1
            selp = os.path.join(tmp, "sel.bin")
            sel.tofile(selp)
            x0 = min(r["pux"] + min(r["lim"][0], r["sx"]) for r in rows) - 6
            x1 = max(r["pux"] + max(r["lim"][1], r["sx"]) + r["w"] for r in rows) + 6
            y0 = min(r["puy"] + min(r["lim"][2], r["sy"]) for r in rows) - 6
            y1 = max(r["puy"] + max(r["lim"][3], r["sy"]) + r["h"] for r in rows) + 6
            x0, y0 = max(x0, -90), max(y0, -76)             # stay inside the padded allocation (stride 608 = 416 + 2 * 96)
            x1, y1 = min(x1, W + 90), min(y1, H + 76)
            dump = os.path.join(tmp, "dump.bin")
            subprocess.run(cmd, env=dict(os.environ, LD_PRELOAD=shim, KS265_ME_DUMP=dump, KS265_ME_SELECT=selp, KS265_ME_REGION=f"{x0} {y0} {x1} {y1}"),
                           capture_output=True, check=True, cwd=tmp)
            data = open(dump, "rb").read()
            pos, epoch_map = 0, {}
            while pos < len(data):
                magic = struct.unpack_from("<I", data, pos)[0]
                if magic == 0x4E414C50:
                    _, ep, rx0, ry0, rw, rh, stride, _ = struct.unpack_from("<IiiiiiiI", data, pos)
                    pos += 32
                    pl = np.frombuffer(data, np.uint8, rw * rh, pos).reshape(rh, rw)
                    pos += rw * rh
                    for k, q in enumerate(planes):          # the first P picture's reference (the I picture) is shared between runs
                        if q.shape == pl.shape and plane_meta[k] == (rx0, ry0) and (q == pl).all():
                            epoch_map[ep] = k
                            break
                    else:
                        epoch_map[ep] = len(planes)
                        planes.append(pl.copy()); plane_meta.append((rx0, ry0))
                else:
                    assert magic == 0x4C4C4143, hex(magic)
                    h = struct.unpack_from("<32i", data, pos)
                    pos += 128
                    (_, idx, method, l2w, l2h, pux, puy, stride, sx, sy, cost0, merange, l0, l1, l2, l3, f3, shift, fin, cur_off, ep, xlo, xhi, ylo, yhi,
                     ox, oy, ocost, oflag, out_off, use_had, _) = h
                    nx, ny = xhi - xlo + 1, yhi - ylo + 1
                    cm = np.frombuffer(data, np.uint16, nx + ny, pos); pos += 2 * (nx + ny)
                    fe = np.frombuffer(data, np.uint8, (1 << l2w) << l2h, pos); pos += (1 << l2w) << l2h
                    assert cur_off == sy * stride + sx + puy * stride + pux, "tME+0x40 does not point at the start position"
                    calls.append((method, l2w, l2h, pux, puy, sx, sy, cost0 & 0xFFFFFFFF, merange, l0, l1, l2, l3, f3, shift, use_had, epoch_map[ep], xlo, xhi, ylo, yhi,
                                  sum(len(c) for c in cm_blob), sum(len(f) for f in fenc_blob), ox, oy, ocost & 0xFFFFFFFF, oflag))
                    cm_blob.append(cm.copy()); fenc_blob.append(fe.copy())
            print(name, len(rows), "calls,", int(sel.sum()), "recorded; planes so far", len(planes))
        names = "method l2w l2h pux puy sx sy cost0 merange xmin xmax ymin ymax skip_cross range_shift use_had plane xlo xhi ylo yhi cm_off fenc_off out_x out_y out_cost out_flag".split()
        arr = np.array(calls, dtype=np.int64)
        out = {"call_fields": np.array(names), "calls": arr, "cm": np.concatenate(cm_blob), "fenc": np.concatenate(fenc_blob),
               "plane_org": np.array(plane_meta, np.int32)}
        for k, p in enumerate(planes):
            out[f"plane{k}"] = p
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "me_search.npz"), **out)
        by = {m: int((arr[:, 0] == m).sum()) for m in (0, 1, 2)}
        print("cases per method (DIA, HEX, UMH):", by, "file", os.path.getsize(os.path.join(ROOT, "tests", "golden", "me_search.npz")), "bytes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
