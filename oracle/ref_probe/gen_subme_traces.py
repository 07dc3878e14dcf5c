#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): record the reference's sub-pel refinement on real encodes and write tests/golden/subme.npz
(pin getMvResolution enc@0x483ca0, subMeSquare enc@0x4b5660, subMeHpel_RealInterp enc@0x4b4e90 and the four
subMeQpel_8Sad_*_RealInterp enc@0x4b2bc0-0x4b43a0).

Every run encodes a synthetic clip with `appencoder -threads 1` twice - plain and under subme_shim.so - and requires the two streams to be
byte-identical (the hooks must not disturb the encoder).  From the dump a subset of the calls is kept: every kind of outcome (no refinement,
half / quarter step winners of every index, all four quarter functions, the fast and the full candidate sets, SAD and Hadamard, square and
rectangular PUs, the tME+0x65 rate path) and a random fill.  The fixture holds DATA only: pixels of synthetic clips and of the encoder's
reconstruction, vectors, costs and the configuration words the functions read.

usage: python oracle/ref_probe/gen_subme_traces.py [--check]     (--check: replay EVERY call of every run against the oracle, write nothing)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("KS265_REF_ENCODER_SRC", "/root/reference/ubuntu_x64/appencoder")
W, H = 416, 240
PER_RUN = 230

# (name, clip kwargs, frames, encoder args): three clips, the presets that differ in the refinement's flags
RUNS = [
    ("slow", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 4, ["-preset", "slow", "-qp", "27"]),
    ("slow_fastclip", dict(seed=79, abc=(9, 11, 5), pan=(15, 10)), 4, ["-preset", "slow", "-qp", "32"]),
    ("medium", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 4, ["-preset", "medium", "-qp", "30"]),
    ("veryfast", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 4, ["-preset", "veryfast", "-qp", "32"]),
    ("veryslow", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 3, ["-preset", "veryslow", "-qp", "27"]),
    ("veryslow_fastclip", dict(seed=79, abc=(9, 11, 5), pan=(15, 10)), 3, ["-preset", "veryslow", "-qp", "30"]),
    ("slow_subme2", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 4, ["-preset", "slow", "-subme", "2", "-qp", "27"]),
    ("slow_b3", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 6, ["-preset", "slow", "-qp", "27", "-bframes", "3"]),
]


def parse(path):
    data = open(path, "rb").read()
    p, sq, res = 0, [], []
    while p < len(data):
        h = np.frombuffer(data, np.int32, 64, p).copy(); p += 256
        assert h[0] == 0x53554232, hex(int(h[0]))
        w, hh = int(h[3]), int(h[4])
        if h[1] == 0:
            fe = reg = cm = None
            if h[42]:
                fe = np.frombuffer(data, np.uint8, w * hh, p); p += w * hh
                reg = np.frombuffer(data, np.uint8, (w + 16) * (hh + 16), p); p += (w + 16) * (hh + 16)
                cm = np.frombuffer(data, np.uint16, 34, p); p += 68
            sq.append((h, fe, reg, cm))
        else:
            fe = reg = None
            if h[42]:
                fe = np.frombuffer(data, np.uint8, w * hh, p); p += w * hh
                reg = np.frombuffer(data, np.uint8, (w + 2) * (hh + 2), p); p += (w + 2) * (hh + 2)
            res.append((h, fe, reg))
    return sq, res


def select(sq, rng):
    """indices of the calls kept from one run"""
    n = len(sq)
    moved = lambda h: (h[32], h[33]) != (h[5], h[6])
    big = lambda h: h[3] * h[4] >= 1024
    pick, nbig = [], 0

    def take(cands, limit):
        nonlocal nbig
        cands = list(cands); rng.shuffle(cands)
        k = 0
        for i in cands:
            if k >= limit or len(pick) >= PER_RUN:
                break
            if i in seen:
                continue
            h = sq[i][0]
            if big(h):
                if nbig >= 22 or (h[3] * h[4] == 4096 and sum(1 for j in pick if sq[j][0][3] * sq[j][0][4] == 4096) >= 6):
                    continue
                nbig += 1
            seen.add(i); pick.append(i); k += 1
    seen = set()
    idx = range(n)
    for dx in (-3, -2, -1, 1, 2, 3):                                   # every kind of result offset
        take([i for i in idx if sq[i][0][32] - sq[i][0][5] == dx], 9)
        take([i for i in idx if sq[i][0][33] - sq[i][0][6] == dx], 9)
    take([i for i in idx if moved(sq[i][0]) and sq[i][0][32] - sq[i][0][5] in (2, 3) and sq[i][0][10]], 25)    # half step moved right (the overlapping-buffer path)
    take([i for i in idx if sq[i][0][3] != sq[i][0][4] and moved(sq[i][0])], 35)                                 # rectangular PUs
    take([i for i in idx if not sq[i][0][10]], 15)                       # no refinement
    take([i for i in idx if sq[i][0][9]], 5)                             # tME+0x65
    take([i for i in idx if big(sq[i][0]) and moved(sq[i][0])], 12)
    take([i for i in idx if moved(sq[i][0])], 60)
    take(idx, PER_RUN)
    return sorted(pick)


def replay(o, ptr, sqrec, resrec):
    import ctypes as C
    h, fe, reg, cm = sqrec
    out = (C.c_int32 * 9)()
    o.kso_subme_replay(ptr(h), ptr(np.ascontiguousarray(fe)), ptr(np.ascontiguousarray(reg)), ptr(np.ascontiguousarray(cm)), out)
    ok = tuple(out[:6]) == tuple(int(h[k]) for k in range(32, 38))
    hr, fr, rr = resrec
    o2 = (C.c_int32 * 2)()
    o.kso_mvres_replay(ptr(hr), ptr(np.ascontiguousarray(fr)) if fr is not None else None, ptr(np.ascontiguousarray(rr)) if rr is not None else None, o2)
    return ok, o2[0] == hr[32], out[6]


def main():
    from ks265codec_amd.synth import make_clip
    from oracle_lib import lib, ptr
    check_all = "--check" in sys.argv
    o = lib()
    tmp = tempfile.mkdtemp(prefix="ks265sm_")
    rng = np.random.default_rng(4)
    try:
        shutil.copy(REF, tmp)
        enc = os.path.join(tmp, "appencoder")
        os.chmod(enc, 0o755)
        shim = os.path.join(tmp, "subme.so")
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "subme_shim.c")])
        sq_hdr, res_hdr, cms, run_of, fenc, region, rfenc, rregion, offs = [], [], [], [], [], [], [], [], []
        fo = ro = rfo = rro = 0
        for r, (name, ckw, frames, args) in enumerate(RUNS):
            clip = make_clip(W, H, frames, **ckw)
            yuv = os.path.join(tmp, "in.yuv")
            clip.tofile(yuv)
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-rc", "0", "-iper", "128", "-threads", "1", *args]
            if "-bframes" not in args:
                cmd += ["-bframes", "0"]
            dump = os.path.join(tmp, "dump.bin")
            subprocess.run(cmd + ["-b", os.path.join(tmp, "plain.265")], capture_output=True, check=True, cwd=tmp)
            subprocess.run(cmd + ["-b", os.path.join(tmp, "hook.265")], env=dict(os.environ, LD_PRELOAD=shim, KS265_SM_DUMP=dump), capture_output=True, check=True, cwd=tmp)
            assert open(os.path.join(tmp, "plain.265"), "rb").read() == open(os.path.join(tmp, "hook.265"), "rb").read(), f"{name}: the hooks changed the stream"
            sq, res = parse(dump)
            assert len(sq) == len(res) and all(int(res[i][0][22]) == i for i in range(len(res))), "getMvResolution / subMeSquare calls are not paired"
            keep = range(len(sq)) if check_all else select(sq, rng)
            bad = 0
            for i in keep:
                ok, ok2, _ = replay(o, ptr, sq[i], res[i])
                bad += (not ok) + (not ok2)
            print(f"{name}: {len(sq)} calls, {len(keep)} {'replayed' if check_all else 'kept'}, {bad} differ from the oracle", flush=True)
            if check_all:
                continue
            for i in keep:
                h, fe, reg, cm = sq[i]
                hr, fr, rr = res[i]
                sq_hdr.append(h); res_hdr.append(hr); cms.append(cm); run_of.append(r)
                offs.append((fo, ro, rfo if fr is not None else -1, rro if rr is not None else -1))
                fenc.append(fe); region.append(reg); fo += len(fe); ro += len(reg)
                if fr is not None:
                    rfenc.append(fr); rregion.append(rr); rfo += len(fr); rro += len(rr)
        if check_all:
            return
        path = os.path.join(ROOT, "tests", "golden", "subme.npz")
        np.savez_compressed(path, runs=np.array([f"{n}: {' '.join(a)}" for n, _, _, a in RUNS]), run_of=np.array(run_of, np.int32),
                            sq_hdr=np.array(sq_hdr, np.int32), res_hdr=np.array(res_hdr, np.int32), cm=np.array(cms, np.uint16), offs=np.array(offs, np.int64),
                            fenc=np.concatenate(fenc), region=np.concatenate(region),
                            res_fenc=np.concatenate(rfenc) if rfenc else np.zeros(0, np.uint8), res_region=np.concatenate(rregion) if rregion else np.zeros(0, np.uint8))
        print("calls", len(sq_hdr), "file", os.path.getsize(path), "bytes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
