"""The oracle's restatement of the reference's START POINT for the integer search (oracle/ks265_me_ref.c: meInitPoint enc@0x48af50 with
checkLayerMv enc@0x48ad80) replayed on calls recorded inside the reference binary (tests/golden/me_init.npz, written by
oracle/ref_probe/gen_init_traces.py: real `appencoder` runs with the function hooked and the PU's distortion pointer logged).  Every
recorded call must choose the reference's predictor index, start vector, cost, window and make exactly the reference's block comparisons
in the reference's order."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle_lib import lib, ptr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "me_init.npz")


def test_start_point_matches_reference_traces():
    z = np.load(GOLD)
    calls, words = z["calls"], [int(w) for w in z["expect_words"]]
    o = lib()
    tabs = {}
    assert len(calls) > 1200
    seen = dict(list1=0, layer=0, outside=0, five=0, extra_won=0, one_cmp=0)
    for h in calls:
        lam = int(h[18])
        if lam not in tabs:        # the mvd cost table: lambda x signed exp-Golomb length (createMvdCostTable enc@0x48b850, pinned in test_me_search.py)
            tabs[lam] = np.array([(lam * o.kso_mvd_bits(d)) & 0xffff for d in range(-256, 257)], np.uint16)
        out = (C.c_int32 * 20)()
        hh = np.ascontiguousarray(h)
        assert o.kso_me_init_replay(ptr(hh), ptr(tabs[lam]), out) == 0, f"call {int(h[1])}: a block comparison the reference did not make (or one missing)"
        exp = [int(h[k]) for k in words]
        got = list(out)[:len(words)]
        if not h[46]:
            got[13] = got[14] = exp[13] = exp[14] = 0          # the stored look-ahead vector is read only when its flag is set
        assert got == exp, f"call {int(h[1])}: {got} != {exp}"
        seen["list1"] += int(h[8] == 1); seen["layer"] += int(h[28] == 1); seen["outside"] += int(h[40] == 1); seen["five"] += int(h[49] == 5)
        seen["extra_won"] += int(h[49] > 2 and h[45] != h[50] and h[45] != h[52]); seen["one_cmp"] += int(h[49] == 1)
    assert all(v > 0 for v in seen.values()), seen          # every branch of the function is in the fixture


def test_mvd_cost_beyond_the_table():
    """|d| > 256: the function's loop (enc@0x48b22c..0x48b24e) counts 3 + 2 floor(log2 |d|) - the signed exp-Golomb length the table holds, continued past its end"""
    o = lib()
    for d in (257, 300, 511, 512, 1000, 4095):
        assert 3 + 2 * (d.bit_length() - 1) == o.kso_mvd_bits(d) == o.kso_mvd_bits(-d)
