#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): record the reference's rate-distortion optimised quantisation on real encodes and write
tests/golden/rdoq.npz (pin rdoQuant enc@0x4aac50).

Every run encodes a synthetic clip with `appencoder -threads 1` twice - plain and under rdoq_shim.so - and requires the two streams to be
byte-identical.  From the dump a subset of the calls is kept: every block size and colour component, the three scans, calls that change
levels / move the last position / drop sub-blocks / apply sign-data hiding in both directions, and a random fill.  The fixture holds DATA
only: levels in and out, transform coefficients (of synthetic clips), the bit table estBitRdoq built, the words of the structures the
function reads and writes.

usage: python oracle/ref_probe/gen_rdoq_traces.py [--check]     (--check: replay EVERY call of every run against the oracle, write nothing)
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
W, H = 416, 240
PER_RUN = 260
MAX_BIG = 10           # 32x32 calls per run (6 KB each)

RUNS = [
    ("slow_qp27", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 4, ["-preset", "slow", "-qp", "27"]),
    ("slow_qp37_b3", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 6, ["-preset", "slow", "-qp", "37", "-bframes", "3"]),
    ("veryslow_qp22", dict(seed=79, abc=(9, 11, 5), pan=(15, 10)), 3, ["-preset", "veryslow", "-qp", "22"]),
    ("medium_qp32", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 4, ["-preset", "medium", "-qp", "32"]),
    ("slow_nosbh", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 3, ["-preset", "slow", "-qp", "30", "-sbh", "0"]),
    ("slow_weights", dict(seed=79, abc=(9, 11, 5), pan=(15, 10)), 3, ["-preset", "slow", "-qp", "30", "-rdoql", "160", "-rdoqc", "200", "-rdoqls", "300", "-rdoqcs", "60"]),
]


def parse(path):
    """one record: 64 int32 header, levels in / out and coefficients (N x N s16 each), the 180-word bit table, 0x1e8 bytes of the TTransUnit before and after"""
    data = open(path, "rb").read()
    p, recs = 0, []
    while p < len(data):
        h = np.frombuffer(data, np.int32, 64, p).copy(); p += 256
        assert h[0] == 0x52444f51, hex(int(h[0]))
        nn = 1 << (2 * int(h[2]))
        b, a, c = (np.frombuffer(data, np.int16, nn, p + 2 * nn * k).copy() for k in range(3)); p += 6 * nn
        tab = np.frombuffer(data, np.int32, 180, p).copy(); p += 720
        tu, tu2 = (np.frombuffer(data, np.uint8, 0x1e8, p + 0x1e8 * k).copy() for k in range(2)); p += 2 * 0x1e8
        recs.append(pack(h, b, a, c, tab, tu, tu2))
    return recs


def pack(h, b, a, c, tab, tu, tu2):
    """the words the function reads and writes, as the fixture keeps them: meta = log2, scan, comp, dq, per, tu5, last_in, flag_a4c0, sdh, ret, last_out, nz_in;
    lam = the two integer lambdas; masks in / out (64 u16), hidden (u64)"""
    comp = int(h[5])
    lamtab = np.frombuffer((h[20:22] if comp == 0 else h[31:33]).tobytes(), np.float64)[0]
    m_main, m_sdh = (int(h[9]), int(h[8])) if comp == 0 else (int(h[25]), int(h[33]))
    i32 = lambda buf, off: int(np.frombuffer(buf[off:off + 4].tobytes(), np.int32)[0])
    meta = np.array([h[2], h[4], comp, h[14], h[16], int(tu[5]), i32(tu, 0x40 + 4 * comp), h[10], h[34], h[6], i32(tu2, 0x40 + 4 * comp), h[3], h[7] if comp == 0 else h[27], m_main, m_sdh, 0], np.int32)
    lam = np.array([int(m_main * lamtab + 0.5), int(m_sdh * lamtab + 0.5)], np.int64)       # cvttsd2si(mult x table[qp] + 0.5), enc@0x4aacdc..0x4aad09
    mk = lambda buf: np.frombuffer(buf[0x68 + comp * 0x80:0xe8 + comp * 0x80].tobytes(), np.uint16).copy()
    hidden = np.frombuffer(tu2[0x50 + 8 * comp:0x58 + 8 * comp].tobytes(), np.uint64)[0]
    return dict(meta=meta, lam=lam, lvl_in=b, lvl_out=a, coef=c, tab=tab, mask_in=mk(tu), mask_out=mk(tu2), hidden=hidden)


def replay(o, ptr, r):
    m = r["meta"]
    lvl, mask = r["lvl_in"].copy(), r["mask_in"].copy()
    ol, oh = C.c_int32(0), C.c_uint64(0)
    ret = o.kso_ref_rdo_quant(ptr(lvl), ptr(r["coef"]), int(m[0]), int(m[1]), int(m[2]), int(m[3]), int(m[4]), C.c_int64(int(r["lam"][0])), C.c_int64(int(r["lam"][1])), ptr(r["tab"]),
                              int(m[5]), int(m[6]), ptr(mask), int(m[7]), int(m[8]), C.byref(ol), C.byref(oh))
    ncg = max(1, (1 << (2 * int(m[0]))) // 16)
    return dict(lvl=(lvl == r["lvl_out"]).all(), ret=ret == m[9], last=ol.value == m[10], hidden=oh.value == int(r["hidden"]), mask=(mask[:ncg] == r["mask_out"][:ncg]).all())


def select(recs, rng):
    n = len(recs)
    pick, seen, nbig = [], set(), 0

    def take(cands, limit):
        nonlocal nbig
        cands = list(cands); rng.shuffle(cands)
        k = 0
        for i in cands:
            if k >= limit or len(pick) >= PER_RUN:
                break
            if i in seen:
                continue
            if recs[i]["meta"][0] == 5:
                if nbig >= MAX_BIG:
                    continue
                nbig += 1
            seen.add(i); pick.append(i); k += 1
    idx = range(n)
    m = lambda i: recs[i]["meta"]
    nzo = lambda i: int(m(i)[9])
    changed = lambda i: not (np.abs(recs[i]["lvl_in"]) == np.abs(recs[i]["lvl_out"])).all()
    grew = lambda i: bool((np.abs(recs[i]["lvl_out"].astype(np.int32)) > np.abs(recs[i]["lvl_in"].astype(np.int32))).any())      # only sign hiding raises a level
    for log2 in (2, 3, 4, 5):
        for comp in (0, 1, 2):
            take([i for i in idx if m(i)[0] == log2 and m(i)[2] == comp and changed(i)], 8)
            take([i for i in idx if m(i)[0] == log2 and m(i)[2] == comp], 4)
    for scan in (1, 2):
        take([i for i in idx if m(i)[1] == scan and changed(i)], 12)
    take([i for i in idx if grew(i)], 40)
    take([i for i in idx if nzo(i) == 0], 15)                                             # everything dropped
    take([i for i in idx if m(i)[10] + 1 < m(i)[6] and nzo(i)], 30)                        # last position moved
    take([i for i in idx if recs[i]["hidden"] and changed(i)], 30)
    take([i for i in idx if m(i)[5] != 0], 20)                                             # deeper transform levels
    take([i for i in idx if not changed(i)], 15)
    take(idx, PER_RUN)
    return sorted(pick)


def main():
    from ks265codec_amd.synth import make_clip
    from oracle_lib import lib, ptr
    check_all = "--check" in sys.argv
    o = lib()
    tmp = tempfile.mkdtemp(prefix="ks265rq_")
    rng = np.random.default_rng(5)
    keep_all, run_of = [], []
    try:
        shutil.copy(REF, tmp)
        enc = os.path.join(tmp, "appencoder")
        os.chmod(enc, 0o755)
        shim = os.path.join(tmp, "rdoq.so")
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "rdoq_shim.c")])
        for r, (name, ckw, frames, args) in enumerate(RUNS):
            clip = make_clip(W, H, frames, **ckw)
            yuv = os.path.join(tmp, "in.yuv")
            clip.tofile(yuv)
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-rc", "0", "-iper", "128", "-threads", "1", *args]
            if "-bframes" not in args:
                cmd += ["-bframes", "0"]
            dump = os.path.join(tmp, "dump.bin")
            subprocess.run(cmd + ["-b", os.path.join(tmp, "plain.265")], capture_output=True, check=True, cwd=tmp)
            subprocess.run(cmd + ["-b", os.path.join(tmp, "hook.265")], env=dict(os.environ, LD_PRELOAD=shim, KS265_RQ_DUMP=dump), capture_output=True, check=True, cwd=tmp)
            assert open(os.path.join(tmp, "plain.265"), "rb").read() == open(os.path.join(tmp, "hook.265"), "rb").read(), f"{name}: the hooks changed the stream"
            recs = parse(dump)
            keep = range(len(recs)) if check_all else select(recs, rng)
            bad = sum(not all(replay(o, ptr, recs[i]).values()) for i in keep)
            print(f"{name}: {len(recs)} calls, {len(keep)} {'replayed' if check_all else 'kept'}, {bad} differ from the oracle", flush=True)
            if not check_all:
                keep_all += [recs[i] for i in keep]; run_of += [r] * len(keep)
        if check_all:
            return
        offs = np.cumsum([0] + [len(r["coef"]) for r in keep_all]).astype(np.int64)
        path = os.path.join(ROOT, "tests", "golden", "rdoq.npz")
        np.savez_compressed(path, runs=np.array([f"{n}: {' '.join(a)}" for n, _, _, a in RUNS]), run_of=np.array(run_of, np.int32), offs=offs,
                            meta=np.array([r["meta"] for r in keep_all], np.int32), lam=np.array([r["lam"] for r in keep_all], np.int64),
                            tab=np.array([r["tab"] for r in keep_all], np.int32), mask_in=np.array([r["mask_in"] for r in keep_all], np.uint16),
                            mask_out=np.array([r["mask_out"] for r in keep_all], np.uint16), hidden=np.array([r["hidden"] for r in keep_all], np.uint64),
                            lvl_in=np.concatenate([r["lvl_in"] for r in keep_all]), lvl_out=np.concatenate([r["lvl_out"] for r in keep_all]),
                            coef=np.concatenate([r["coef"] for r in keep_all]))
        print("calls", len(keep_all), "file", os.path.getsize(path), "bytes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
