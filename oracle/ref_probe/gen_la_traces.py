#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): record the reference's lookahead decisions on real encodes and write tests/golden/lookahead_ref.npz
(pin calcFrameAdaptQuant enc@0x4653c0, cuTreePropagate enc@0x47d460, scenecut enc@0x47e9d0).

Every run encodes a synthetic clip (scene changes and flat pictures included) with `appencoder -threads 1` twice - plain and under la_shim.so - and
requires the two streams to be byte-identical.  The fixture holds DATA only: pixels of synthetic clips, cost planes, vectors, the words the functions read,
and what they returned.

usage: python oracle/ref_probe/gen_la_traces.py [--check]     (--check: replay EVERY call of every run against the oracle, write nothing)
"""
from __future__ import annotations

import ctypes as C
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

# (name, width, height, encoder args, AQ calls kept, cuTree calls kept)
RUNS = [
    ("crf_b3", 416, 240, ["-preset", "slow", "-rc", "3", "-crf", "26", "-bframes", "3", "-aq", "1", "-aqs", "1.0", "-cutree", "1", "-scenecut", "40", "-lookahead", "10"], 2, 30),
    ("crf_b7_small", 208, 128, ["-preset", "medium", "-rc", "3", "-crf", "30", "-bframes", "7", "-aq", "1", "-aqs", "1.5", "-cutree", "1", "-scenecut", "60", "-lookahead", "20"], 8, 40),
    ("cbr_p", 208, 128, ["-preset", "slow", "-rc", "1", "-br", "300", "-bframes", "0", "-aq", "1", "-aqs", "0.6", "-scenecut", "30", "-iper", "24"], 6, 20),
    ("abr_b3", 416, 240, ["-preset", "veryfast", "-rc", "2", "-br", "500", "-bframes", "3", "-aq", "1", "-cutree", "1", "-scenecut", "40"], 1, 20),
]


def clip_with_cuts(W, H):
    """44 pictures: a panning scene, a hard cut to a brighter, busier one, three flat pictures, the first scene again, a dark still"""
    from ks265codec_amd.synth import make_clip
    a = make_clip(W, H, 12, seed=1234, abc=(17, 23, 9), pan=(5, 3))
    b = make_clip(W, H, 10, seed=99, abc=(5, 7, 3), pan=(2, 6)).astype(np.int32)
    b[:, :W * H] = np.clip((b[:, :W * H] - 128) * 2 + 170, 0, 255)
    flat = np.full((3, W * H * 3 // 2), 128, np.uint8)
    dark = np.repeat((a[:1].astype(np.int32) // 4 + 16).astype(np.uint8), 7, axis=0)
    return np.concatenate([a, b.astype(np.uint8), flat, a[::-1], dark])


def parse(path):
    data = open(path, "rb").read()
    p, out = 0, {1: [], 2: [], 3: []}
    while p < len(data):
        h = np.frombuffer(data, np.int32, 32, p).copy(); p += 128
        assert h[0] == 0x4c4f4f4b, hex(int(h[0]))
        pay = data[p:p + int(h[2])]; p += int(h[2])
        kind = int(h[1])
        if kind == 1:
            nx, ny, cnt = int(h[3]), int(h[4]), int(h[5])
            n = nx * ny
            o = 0
            Y = np.frombuffer(pay, np.uint8, n * 256, o); o += n * 256
            U = np.frombuffer(pay, np.uint8, n * 64, o); o += n * 64
            V = np.frombuffer(pay, np.uint8, n * 64, o); o += n * 64
            off = np.frombuffer(pay, np.float64, cnt, o); o += cnt * 8
            off2 = np.frombuffer(pay, np.float64, cnt, o); o += cnt * 8
            inv = np.frombuffer(pay, np.uint16, cnt, o)
            assert (off == off2).all()
            out[1].append(dict(h=h, Y=Y, U=U, V=V, off=off, inv=inv, strength=float(np.frombuffer(h[8:10].tobytes(), np.float64)[0])))
        elif kind == 2:
            n = int(h[3]) * int(h[4])
            o = 0
            r = dict(h=h)
            for name, dt, cnt in (("intra", np.uint16, n), ("invq", np.uint16, n), ("own", np.uint16, n), ("inter", np.uint16, n), ("bits", np.uint8, (n + 3) // 4), ("mv0", np.int32, n),
                                  ("mv1", np.int32, n), ("bef0", np.uint16, n), ("bef1", np.uint16, n), ("aft0", np.uint16, n), ("aft1", np.uint16, n)):
                r[name] = np.frombuffer(pay, dt, cnt, o).copy(); o += cnt * np.dtype(dt).itemsize
            out[2].append(r)
        else:
            out[3].append(h)
    return out


def replay_aq(o, ptr, r):
    h = r["h"]
    nx, ny, cnt = int(h[3]), int(h[4]), int(h[5])
    off = np.zeros(nx * ny, np.float64); inv = np.zeros(nx * ny, np.uint16)
    o.kso_ref_frame_adapt_quant(ptr(np.ascontiguousarray(r["Y"])), ptr(np.ascontiguousarray(r["U"])), ptr(np.ascontiguousarray(r["V"])), nx, ny, cnt, C.c_double(r["strength"]), ptr(off), ptr(inv))
    return bool((off[:cnt] == r["off"]).all() and (inv[:cnt] == r["inv"]).all())


def replay_ct(o, ptr, r):
    h = r["h"]
    p0, p1, b = int(h[6]), int(h[7]), int(h[8])
    r0 = r["bef0"].copy()
    r1 = r0 if p0 == p1 else r["bef1"].copy()                           # one plane when both references are one picture
    o.kso_ref_cutree_propagate(int(h[5]), int(h[3]), int(h[4]), ptr(r["intra"]), ptr(r["invq"]), ptr(r["own"]), ptr(r["inter"]), ptr(r["bits"]), ptr(r["mv0"]), ptr(r["mv1"]), ptr(r0), ptr(r1))
    return bool((r0 == r["aft0"]).all() and (r1 == r["aft1"]).all())


def replay_sc(o, h):
    return o.kso_ref_scenecut(int(h[6]), int(h[7]), int(h[8]), int(h[9]) * int(h[10]), int(h[11]), int(h[12]), int(h[13]), int(h[14]), int(h[15])) == (int(h[3]) & 0xff)


def main():
    from oracle_lib import lib, ptr
    check_all = "--check" in sys.argv
    o = lib()
    tmp = tempfile.mkdtemp(prefix="ks265la_")
    rng = np.random.default_rng(6)
    aq, ct, sc = [], [], []
    try:
        shutil.copy(REF, tmp)
        enc = os.path.join(tmp, "appencoder")
        os.chmod(enc, 0o755)
        shim = os.path.join(tmp, "la.so")
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "la_shim.c")])
        for r, (name, W, H, args, keep_aq, keep_ct) in enumerate(RUNS):
            yuv = os.path.join(tmp, "in.yuv")
            clip_with_cuts(W, H).tofile(yuv)
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-threads", "1", *args]
            dump = os.path.join(tmp, "dump.bin")
            subprocess.run(cmd + ["-b", os.path.join(tmp, "plain.265")], capture_output=True, check=True, cwd=tmp)
            subprocess.run(cmd + ["-b", os.path.join(tmp, "hook.265")], env=dict(os.environ, LD_PRELOAD=shim, KS265_LA_DUMP=dump), capture_output=True, check=True, cwd=tmp)
            assert open(os.path.join(tmp, "plain.265"), "rb").read() == open(os.path.join(tmp, "hook.265"), "rb").read(), f"{name}: the hooks changed the stream"
            rec = parse(dump)
            ka = range(len(rec[1])) if check_all else sorted(rng.choice(len(rec[1]), min(keep_aq, len(rec[1])), replace=False))
            moved = [i for i, c in enumerate(rec[2]) if c["mv0"].any() or c["mv1"].any()]
            kc = range(len(rec[2])) if check_all else sorted(set(rng.choice(moved, min(keep_ct * 2 // 3, len(moved)), replace=False)) | set(rng.choice(len(rec[2]), min(keep_ct // 3, len(rec[2])), replace=False))) if rec[2] else []
            bad = [sum(not replay_aq(o, ptr, rec[1][i]) for i in ka), sum(not replay_ct(o, ptr, rec[2][i]) for i in kc), sum(not replay_sc(o, h) for h in rec[3])]
            print(f"{name}: calcFrameAdaptQuant {len(rec[1])} calls ({len(ka)} {'replayed' if check_all else 'kept'}, {bad[0]} differ), cuTreePropagate {len(rec[2])} ({len(kc)}, {bad[1]} differ), "
                  f"scenecut {len(rec[3])} ({sum(1 for h in rec[3] if h[3] & 0xff)} cuts, {bad[2]} differ)", flush=True)
            if not check_all:
                aq += [dict(rec[1][i], run=r) for i in ka]; ct += [dict(rec[2][i], run=r) for i in kc]; sc += [np.concatenate([h, [r]]) for h in rec[3]]
        if check_all:
            return
        cat = lambda rows, k: np.concatenate([x[k] for x in rows])
        path = os.path.join(ROOT, "tests", "golden", "lookahead_ref.npz")
        np.savez_compressed(path, runs=np.array([f"{n} {W}x{H}: {' '.join(a)}" for n, W, H, a, _, _ in RUNS]),
                            aq_hdr=np.array([x["h"] for x in aq], np.int32), aq_run=np.array([x["run"] for x in aq], np.int32), aq_strength=np.array([x["strength"] for x in aq]),
                            aq_y=cat(aq, "Y"), aq_u=cat(aq, "U"), aq_v=cat(aq, "V"), aq_off=cat(aq, "off"), aq_inv=cat(aq, "inv"),
                            ct_hdr=np.array([x["h"] for x in ct], np.int32), ct_run=np.array([x["run"] for x in ct], np.int32),
                            **{"ct_" + k: cat(ct, k) for k in ("intra", "invq", "own", "inter", "bits", "mv0", "mv1", "bef0", "bef1", "aft0", "aft1")},
                            sc=np.array(sc, np.int32))
        print("fixture", os.path.getsize(path), "bytes:", len(aq), "AQ,", len(ct), "cuTree,", len(sc), "scenecut calls")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
