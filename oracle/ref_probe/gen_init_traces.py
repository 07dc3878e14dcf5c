#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): record the reference's choice of the integer search's start point on real encodes and write
tests/golden/me_init.npz (pin meInitPoint enc@0x48af50 and checkLayerMv enc@0x48ad80).

Every run encodes a synthetic clip with `appencoder -threads 1` twice - plain and under init_shim.so - and requires the two streams to be
byte-identical.  One record per call: the words of TPredUnit / tME the function reads (AMVP candidates, mv limits, merange, lambda, index
costs, the extra vectors, the look-ahead's vector), every block comparison it made as (plane offset, value), the words it wrote, and the
mvd cost table base[-256..256].  The table is checked against its closed form (lambda x exp-Golomb length, oracle kso_mvd_bits) for every
call and kept in the fixture only as lambda.  The fixture holds DATA only.

usage: python oracle/ref_probe/gen_init_traces.py [--check]     (--check: replay EVERY call of every run against the oracle, write nothing)
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
PER_RUN = 220
REC = 256 + 1026

RUNS = [
    ("dia_slow", dict(seed=1234, abc=(17, 23, 9), pan=(5, 3)), 4, ["-preset", "slow", "-me", "0", "-qp", "27", "-bframes", "0"]),
    ("hex_fast_pan", dict(seed=77, abc=(9, 11, 5), pan=(13, 9)), 4, ["-preset", "veryfast", "-me", "1", "-qp", "37", "-bframes", "0"]),
    ("umh_b3", dict(seed=80, abc=(13, 7, 5), pan=(10, 12)), 6, ["-preset", "slow", "-me", "2", "-qp", "32", "-bframes", "3"]),
    ("range4", dict(seed=79, abc=(9, 11, 5), pan=(15, 10)), 4, ["-preset", "medium", "-me", "1", "-qp", "30", "-bframes", "0", "-merange", "4"]),
    ("range4_b", dict(seed=78, abc=(9, 11, 5), pan=(14, 7)), 6, ["-preset", "slow", "-me", "0", "-qp", "24", "-bframes", "2", "-merange", "4"]),
    ("ref4", dict(seed=81, abc=(11, 5, 7), pan=(7, 6)), 6, ["-preset", "veryslow", "-me", "1", "-qp", "30", "-bframes", "0", "-ref", "4"]),
    ("rc_lookahead", dict(seed=82, abc=(7, 9, 11), pan=(9, 4)), 8, ["-preset", "medium", "-me", "1", "-rc", "1", "-br", "400", "-bframes", "3"]),
]


def parse(path):
    d = np.fromfile(path, np.uint8)
    n = len(d) // REC
    assert n * REC == len(d) and n
    d = d.reshape(n, REC)
    h = np.ascontiguousarray(d[:, :256]).view(np.int32)
    t = np.ascontiguousarray(d[:, 256:]).view(np.uint16)
    assert (h[:, 0] == 0x54504e49).all()
    return h, t


EXPECT = [34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 60, 61]     # record words the replay's out[0..17] must equal


def table(o, lam):
    return np.array([(lam * o.kso_mvd_bits(d)) & 0xffff for d in range(-256, 257)], np.uint16)


def replay(o, ptr, h, tab):
    out = (C.c_int32 * 20)()
    hh = np.ascontiguousarray(h)
    rc = o.kso_me_init_replay(ptr(hh), ptr(tab), out)
    exp = [int(h[k]) for k in EXPECT]
    if not h[46]:
        exp[13] = exp[14] = None                     # the stored look-ahead vector means something only when its flag is set
    got = [g if e is not None else None for g, e in zip(list(out)[:18], exp)]
    return rc == 0 and got == exp


def main():
    check = "--check" in sys.argv
    from ks265codec_amd.synth import make_clip
    from oracle_lib import lib, ptr
    o = lib()
    tmp = tempfile.mkdtemp(prefix="ks265ip_")
    keep_h, keep_run, stats = [], [], []
    rng = np.random.default_rng(4)
    try:
        enc = os.path.join(tmp, "appencoder")
        shutil.copy(REF, enc)
        os.chmod(enc, 0o755)
        shim = os.path.join(tmp, "init.so")
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "init_shim.c")])
        tabs = {}
        for ri, (name, ckw, frames, args) in enumerate(RUNS):
            yuv = os.path.join(tmp, "in.yuv")
            make_clip(W, H, frames, **ckw).tofile(yuv)
            cmd = [enc, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-iper", "128", "-threads", "1", *args]
            if "-rc" not in args:
                cmd += ["-rc", "0"]
            dump = os.path.join(tmp, "dump.bin")
            subprocess.run(cmd + ["-b", os.path.join(tmp, "a.265")], capture_output=True, check=True, cwd=tmp)
            subprocess.run(cmd + ["-b", os.path.join(tmp, "b.265")], env=dict(os.environ, LD_PRELOAD=shim, KS265_IP_DUMP=dump), capture_output=True, check=True, cwd=tmp)
            a, b = open(os.path.join(tmp, "a.265"), "rb").read(), open(os.path.join(tmp, "b.265"), "rb").read()
            assert a == b and len(a) > 100, f"{name}: the hooks changed the stream"
            h, t = parse(dump)
            bad_tab = 0
            for i in range(len(h)):
                lam = int(h[i, 18])
                if lam not in tabs:
                    tabs[lam] = table(o, lam)
                span = min(256, 4 * int(h[i, 17]) + 16)                # createMvdCostTable enc@0x48b850 fills |d| <= 4 merange + 16
                bad_tab += int((tabs[lam][256 - span:257 + span] != t[i][256 - span:257 + span]).any())
            assert bad_tab == 0, f"{name}: {bad_tab} calls whose mvd cost table is not lambda x bits"
            # with a small merange a start point outside the window makes the function read base[d] past the table's end (|d| <= 256 is all it checks):
            # what it adds then is whatever the heap holds.  Those calls replay on the recorded memory but cannot be part of a closed-form fixture.
            ok_mem = np.array([replay(o, ptr, h[i], np.ascontiguousarray(t[i])) for i in range(len(h))])
            ok = np.array([replay(o, ptr, h[i], tabs[int(h[i, 18])]) for i in range(len(h))])
            assert ok_mem.all(), f"{name}: {int((~ok_mem).sum())} calls differ"
            st = dict(run=name, calls=len(h), list1=int((h[:, 8] == 1).sum()), layer=int((h[:, 28] == 1).sum()), outside=int((h[:, 40] == 1).sum()),
                      same_cand=int((h[:, 49] >= 1).sum() - ((h[:, 9] + 2 >> 2 != h[:, 11] + 2 >> 2) | (h[:, 10] + 2 >> 2 != h[:, 12] + 2 >> 2)).sum()),
                      five=int((h[:, 49] == 5).sum()), moved=int(((h[:, 49] > 2) & (h[:, 45] != h[:, 50]) & (h[:, 45] != h[:, 52])).sum()))
            st["replayed"] = int(ok_mem.sum()); st["past_table_end"] = int((~ok).sum())
            stats.append(st)
            print(st, flush=True)
            assert (h[~ok, 40] == 1).all() and (4 * h[~ok, 17] + 16 < 256).all()
            if check:
                continue
            h = h[ok]
            # selection: calls where an extra vector won, calls outside the window, all five comparisons, list 1, then a random fill
            score = (h[:, 40] == 1) * 8 + (h[:, 49] == 5) * 4 + ((h[:, 49] > 2) & (h[:, 45] != h[:, 50]) & (h[:, 45] != h[:, 52])) * 2 + (h[:, 8] == 1) + rng.random(len(h)) * 1.5
            pick = np.sort(np.argsort(-score)[:PER_RUN])
            keep_h.append(h[pick]); keep_run += [ri] * len(pick)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if check:
        print("all calls of all runs replayed:", sum(s["calls"] for s in stats))
        return
    out = os.path.join(ROOT, "tests", "golden", "me_init.npz")
    np.savez_compressed(out, calls=np.concatenate(keep_h), run=np.array(keep_run, np.int16), run_names=np.array([r[0] for r in RUNS]),
                        expect_words=np.array(EXPECT, np.int16), stats=np.array([str(s) for s in stats]))
    print("wrote", out, os.path.getsize(out), "bytes,", len(keep_run), "calls")


if __name__ == "__main__":
    main()
