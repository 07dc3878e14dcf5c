#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): record the reference's lookahead cost function on real encodes and write tests/golden/calc_frame_cost.npz
(pin calcFrameCost enc@0x4a7410 and the cuTree finish inlined in CInputPicManage::updateQueue enc@0x480964..0x480a54).

Every run encodes a synthetic clip with `appencoder -threads 1` twice - plain and under cfc_shim.so - and requires the two streams to be byte-identical.
The fixture holds DATA only: half-size pictures of synthetic clips (each distinct plane once), the per-block arrays and sums the function read and what it left.

usage: python oracle/ref_probe/gen_cfc_traces.py [--check]     (--check: replay EVERY call of every run against the oracle, write nothing)
"""
from __future__ import annotations

import ctypes as C
import hashlib
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
sys.path.insert(0, HERE)
REF = os.environ.get("KS265_REF_ENCODER_SRC", "/root/reference/ubuntu_x64/appencoder")
ADD_PROLOGUE = bytes.fromhex("41 57 31 c0 b9 09 00 00 00 41 56 41 55".replace(" ", ""))     # addPicTobeEncoded enc@0x47f9a0: push r15; xor eax,eax; mov ecx,9; push r14; push r13

# (name, width, height, encoder args, calcFrameCost calls kept, finish records kept)
RUNS = [
    ("crf_b3", 208, 128, ["-preset", "slow", "-rc", "3", "-crf", "26", "-bframes", "3", "-aq", "1", "-aqs", "1.0", "-cutree", "1", "-scenecut", "40", "-lookahead", "10"], 28, 10),
    ("crf_b7", 208, 128, ["-preset", "medium", "-rc", "3", "-crf", "30", "-bframes", "7", "-aq", "1", "-aqs", "1.5", "-cutree", "1", "-scenecut", "60", "-lookahead", "20"], 28, 10),
    ("cbr_p", 208, 128, ["-preset", "slow", "-rc", "1", "-br", "300", "-bframes", "0", "-aq", "1", "-aqs", "0.6", "-scenecut", "30", "-iper", "24"], 14, 4),
    ("crf_default", 416, 240, ["-preset", "slow", "-rc", "3", "-crf", "24"], 8, 4),
    ("abr_veryfast", 208, 128, ["-preset", "veryfast", "-rc", "2", "-br", "400", "-bframes", "3", "-cutree", "1", "-scenecut", "40"], 14, 4),
    ("crf_ultrafast", 208, 128, ["-preset", "ultrafast", "-rc", "3", "-crf", "28", "-scenecut", "0"], 12, 2),       # 16x16 blocks, fast intra, two intra modes under the flat threshold
    ("crf_superfast", 256, 144, ["-preset", "superfast", "-rc", "3", "-crf", "28", "-bframes", "3"], 10, 2),
    ("crf_nocutree", 208, 128, ["-preset", "slow", "-rc", "3", "-crf", "26", "-bframes", "3", "-cutree", "0"], 12, 2),  # B pictures without the intra comparison
    ("crf_480p", 832, 480, ["-preset", "slow", "-rc", "3", "-crf", "26", "-frms", "14"], 3, 2),                      # the large-picture branch of the motion threshold
]
from cfc_cases import ARR, KsoCfc, replay_call, replay_finish, set_bits_table   # tests/cfc_cases.py: the struct and the replay the CPU test uses too


# whole runs (name, width, height, pictures, encoder args, the mirror's arguments): clips without cuts, a length of 4 k + 1 so that the pyramid of 4 has no partial mini-GOP
CLIP_ARGS = dict(seed=11, abc=(17, 23, 9), pan=(3, 2))
WHOLE_RUNS = [
    ("crf_b3_33", 208, 128, 33, ["-preset", "slow", "-rc", "3", "-crf", "26", "-bframes", "3", "-iper", "64"], dict(preset=5, gop_b=3, hier=True, iper=64)),
    ("crf_b3_aq_25", 208, 128, 25, ["-preset", "slow", "-rc", "3", "-crf", "24", "-bframes", "3", "-iper", "64", "-aq", "1", "-aqs", "1.2"], dict(preset=5, gop_b=3, hier=True, iper=64, aq_strength=1.2)),
    ("crf_b3_la20_41", 208, 128, 41, ["-preset", "slow", "-rc", "3", "-crf", "26", "-bframes", "3", "-iper", "64", "-lookahead", "20"], dict(preset=5, gop_b=3, hier=True, iper=64, lookahead=20)),
    ("crf_b7_33", 256, 144, 33, ["-preset", "slow", "-rc", "3", "-crf", "28", "-bframes", "7", "-iper", "64"], dict(preset=5, gop_b=7, hier=True, iper=64)),
    ("crf_veryfast_b3_17", 256, 144, 17, ["-preset", "veryfast", "-rc", "3", "-crf", "28", "-bframes", "3", "-iper", "64"], dict(preset=2, gop_b=3, hier=True, iper=64)),
]


def parse(path):
    data = open(path, "rb").read()
    p, calls, fin = 0, [], []
    while p < len(data):
        h = np.frombuffer(data, np.int32, 64, p).copy(); p += 256
        assert h[0] == 0x43464331, hex(int(h[0]))
        pay = data[p:p + int(h[2])]; p += int(h[2])
        if h[1] == 4:
            w, hh, nx, ny = (int(v) for v in h[7:11]); n = nx * ny
            mx, my = int(h[34]), int(h[35]); psz = (w + 2 * mx) * (hh + 2 * my)
            o = 0
            r = dict(h=h, lam=np.frombuffer(pay, np.uint16, 52, o).copy(), tab=np.frombuffer(pay, np.uint16, 2049, o + 104).copy()); o += 104 + 4098
            r["cur"] = np.frombuffer(pay, np.uint8, psz, o).copy(); o += psz
            for k, given in (("ref0", h[54]), ("ref1", h[55])):
                r[k] = np.frombuffer(pay, np.uint8, psz if given else 1, o).copy(); o += psz if given else 1
            for tag in ("b_", "a_"):
                for name, dt, per in ARR:
                    cnt = n if per else (n + 3) // 4
                    r[tag + name] = np.frombuffer(pay, dt, cnt, o).copy(); o += cnt * np.dtype(dt).itemsize
            assert o == len(pay)
            calls.append(r)
        else:
            n, cnt = int(h[9]) * int(h[10]), int(h[11]); o = 0
            r = dict(h=h)
            for name, dt, c in (("intra", np.uint16, n), ("invq", np.uint16, n), ("prop", np.uint16, n), ("aq", np.float64, cnt), ("out", np.float64, cnt)):
                r[name] = np.frombuffer(pay, dt, c, o).copy(); o += c * np.dtype(dt).itemsize
            fin.append(r)
    return calls, fin


def main():
    from oracle_lib import lib
    check_all = "--check" in sys.argv
    o = lib()
    o.kso_mvd_bits.restype = C.c_int
    set_bits_table(o)
    tmp = tempfile.mkdtemp(prefix="ks265cfc_")
    rng = np.random.default_rng(11)
    kept_calls, kept_fin = [], []
    try:
        shutil.copy(REF, tmp)
        enc = os.path.join(tmp, "appencoder")
        os.chmod(enc, 0o755)
        with open(enc, "rb") as f:
            f.seek(0x47f9a0 - 0x400000)
            assert f.read(len(ADD_PROLOGUE)) == ADD_PROLOGUE, "addPicTobeEncoded's prologue is not the one the shim displaces"
        shim = os.path.join(tmp, "cfc.so")
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "cfc_shim.c")])
        from gen_la_traces import clip_with_cuts
        for ri, (name, W, H, args, keep_c, keep_f) in enumerate(RUNS):
            yuv = os.path.join(tmp, "in.yuv")
            clip_with_cuts(W, H).tofile(yuv)
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-threads", "1", *args]
            dump = os.path.join(tmp, "dump.bin")
            subprocess.run(cmd + ["-b", os.path.join(tmp, "plain.265")], capture_output=True, check=True, cwd=tmp)
            subprocess.run(cmd + ["-b", os.path.join(tmp, "hook.265")], env=dict(os.environ, LD_PRELOAD=shim, KS265_CFC_DUMP=dump, KS265_CFC_ADDLEN=str(len(ADD_PROLOGUE))), capture_output=True, check=True, cwd=tmp)
            assert open(os.path.join(tmp, "plain.265"), "rb").read() == open(os.path.join(tmp, "hook.265"), "rb").read(), f"{name}: the hooks changed the stream"
            calls, fin = parse(dump)
            kc = range(len(calls)) if check_all else sorted(rng.choice(len(calls), min(keep_c, len(calls)), replace=False))
            refd = [i for i, r in enumerate(fin) if r["h"][5]]
            kf = range(len(fin)) if check_all else sorted(set(rng.choice(refd, min(keep_f, len(refd)), replace=False)) | set(rng.choice(len(fin), min(2, len(fin)), replace=False))) if fin else []
            nbad = 0
            for i in kc:
                bad = replay_call(o, calls[i])
                if [b for b in bad if not b.startswith("(")]:
                    nbad += 1
                    if nbad <= 5:
                        h = calls[i]["h"]
                        print(f"  call {i} poc {h[14]} d0 {h[3]} d1 {h[4]}: {bad}")
            fbad = sum(not replay_finish(o, fin[i]) for i in kf)
            kinds = {}
            for r in calls:
                k = (int(r["h"][3]) > 0, int(r["h"][4]) > 0); kinds[k] = kinds.get(k, 0) + 1
            print(f"{name}: calcFrameCost {len(calls)} computing calls (intra {kinds.get((False, False), 0)}, P {kinds.get((True, False), 0)}, B {kinds.get((True, True), 0)}; "
                  f"{len(kc)} {'replayed' if check_all else 'kept'}, {nbad} differ), cuTree finish {len(fin)} pictures ({len(kf)}, {fbad} differ)", flush=True)
            if not check_all:
                kept_calls += [dict(calls[i], run=ri) for i in kc]; kept_fin += [dict(fin[i], run=ri) for i in kf]
        # a WHOLE run: the offsets the reference left for every picture of a clip without cuts - what the composition of the pinned pieces (the host's cuTree pass, restated in
        # tests/cutree_mirror.py) has to reproduce picture for picture
        from ks265codec_amd.synth import make_clip
        from cutree_mirror import CuTree
        whole = {}
        for name, W, H, nfr, args, kw in WHOLE_RUNS:
            clip = make_clip(W, H, nfr, **CLIP_ARGS)
            yuv = os.path.join(tmp, "w.yuv"); clip.tofile(yuv)
            dump = os.path.join(tmp, "w.bin")
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-threads", "1", *args]
            subprocess.run(cmd + ["-b", os.path.join(tmp, "plain.265")], capture_output=True, check=True, cwd=tmp)
            subprocess.run(cmd + ["-b", os.path.join(tmp, "hook.265")], env=dict(os.environ, LD_PRELOAD=shim, KS265_CFC_DUMP=dump, KS265_CFC_ADDLEN=str(len(ADD_PROLOGUE))), capture_output=True, check=True, cwd=tmp)
            assert open(os.path.join(tmp, "plain.265"), "rb").read() == open(os.path.join(tmp, "hook.265"), "rb").read(), f"{name}: the hooks changed the stream"
            calls, fin = parse(dump)
            assert sorted(int(f["h"][3]) for f in fin) == list(range(nfr))
            ct = CuTree(clip, W, H, **kw); ct.run()
            same = sum(bool((ct.maps_qoff[int(f["h"][3])] == f["out"]).all()) for f in fin)
            print(f"{name}: {len(calls)} calcFrameCost calls, {len(fin)} pictures handed on; the mirror of the host's pass leaves the reference's offsets in {same} of {len(fin)} pictures", flush=True)
            by = {int(f["h"][3]): f for f in fin}
            whole[name] = dict(same=np.array([bool((ct.maps_qoff[t] == by[t]["out"]).all()) for t in range(nfr)]), args=np.array(" ".join(args)), size=np.array([W, H, nfr], np.int32),
                               kw=np.array(repr(kw)), off=np.array([by[t]["out"] for t in range(nfr)]), kind=np.array([by[t]["h"][4] for t in range(nfr)], np.int32), isref=np.array([by[t]["h"][5] for t in range(nfr)], np.int32),
                               order=np.array([[c["h"][14], c["h"][15], c["h"][16]] for c in calls], np.int32))
        if check_all:
            return
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cutree_run.npz"), names=np.array(list(whole)), **{f"{n}_{k}": v for n, d in whole.items() for k, v in d.items()})
        # every distinct plane once
        planes, index = [], {}
        def pid(a):
            if a.size <= 1:
                return -1
            k = hashlib.sha1(a.tobytes()).hexdigest()
            if k not in index:
                index[k] = len(planes); planes.append(a)
            return index[k]
        hdr = np.array([x["h"] for x in kept_calls], np.int32)
        pl = np.array([[pid(x["cur"]), pid(x["ref0"]), pid(x["ref1"])] for x in kept_calls], np.int32)
        cat = lambda rows, k: np.concatenate([x[k] for x in rows])
        path = os.path.join(ROOT, "tests", "golden", "calc_frame_cost.npz")
        np.savez_compressed(path, runs=np.array([f"{n} {W}x{H}: {' '.join(a)}" for n, W, H, a, _, _ in RUNS]),
                            hdr=hdr, run=np.array([x["run"] for x in kept_calls], np.int32), plane_of=pl,
                            plane_len=np.array([p.size for p in planes], np.int64), plane_data=np.concatenate(planes), tab=np.array([x["tab"] for x in kept_calls]), lam=np.array([x["lam"] for x in kept_calls]),
                            **{t + name: cat(kept_calls, t + name) for t in ("b_", "a_") for name, _, _ in ARR},
                            fin_hdr=np.array([x["h"] for x in kept_fin], np.int32), **{"fin_" + k: cat(kept_fin, k) for k in ("intra", "invq", "prop", "aq", "out")})
        print("fixture", os.path.getsize(path), "bytes:", len(kept_calls), "calcFrameCost calls,", len(planes), "planes,", len(kept_fin), "finish records")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
