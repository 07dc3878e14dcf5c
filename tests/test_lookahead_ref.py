"""The oracle's restatement of three of the reference's LOOKAHEAD decisions (oracle/ks265_lookahead_ref.c: calcFrameAdaptQuant enc@0x4653c0, cuTreePropagate enc@0x47d460,
scenecut enc@0x47e9d0) replayed on calls recorded inside the reference binary (tests/golden/lookahead_ref.npz, written by
oracle/ref_probe/gen_la_traces.py: real `appencoder` runs with -aq 1, -cutree 1, -scenecut N under -rc 1 / 2 / 3, -bframes 0 / 3 / 7, on a clip with hard cuts, flat
pictures and a still; the stream checked to be unchanged by the hooks).  Bit-exact: the QP offsets are doubles and must be equal, not close."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle_lib import lib, ptr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lookahead_ref.npz")


def test_frame_adapt_quant_matches_reference_traces():
    z = np.load(GOLD)
    o = lib()
    hdr = z["aq_hdr"]
    assert len(hdr) >= 15
    py = pc = po = 0
    Y, U, V, OFF, INV = (np.ascontiguousarray(z[k]) for k in ("aq_y", "aq_u", "aq_v", "aq_off", "aq_inv"))
    strengths = set()
    for i, h in enumerate(hdr):
        nx, ny, cnt = int(h[3]), int(h[4]), int(h[5])
        n = nx * ny
        off, inv = np.zeros(n, np.float64), np.zeros(n, np.uint16)
        o.kso_ref_frame_adapt_quant(ptr(Y, py), ptr(U, pc), ptr(V, pc), nx, ny, cnt, C.c_double(float(z["aq_strength"][i])), ptr(off), ptr(inv))
        assert (off[:cnt] == OFF[po:po + cnt]).all(), f"call {i}: QP offsets differ (max {np.abs(off[:cnt] - OFF[po:po + cnt]).max()})"
        assert (inv[:cnt] == INV[po:po + cnt]).all(), f"call {i}: inverse qscale factors differ"
        strengths.add(float(z["aq_strength"][i]))
        py += n * 256; pc += n * 64; po += cnt
    assert len(strengths) >= 3 and len({(int(h[3]), int(h[4])) for h in hdr}) >= 2, "several strengths, two picture sizes"
    assert np.ptp(OFF) > 1.0 and (INV == 256).sum() < INV.size // 2, "the offsets are not trivial"


def test_log2_and_exp2fix8_forms():
    """the closed forms behind _log2 enc@0x4c3c20 / qy265_exp2fix8 enc@0x4c3c50 (their tables were checked against the file when the oracle was written): spot values"""
    o = lib()
    o.kso_ref_log2.restype = C.c_double
    assert o.kso_ref_log2(C.c_uint32(1)) == 0.0 and o.kso_ref_log2(C.c_uint32(1 << 20)) == 20.0
    assert o.kso_ref_log2(C.c_uint32(129 << 10)) == 17.0 + 0.01123
    assert [o.kso_ref_exp2fix8(C.c_double(x)) for x in (0.0, 6.0, -6.0, 100.0, -100.0)] == [256, 128, 512, 0, 65535]


def test_cutree_propagate_matches_reference_traces():
    z = np.load(GOLD)
    o = lib()
    hdr = z["ct_hdr"]
    assert len(hdr) >= 60
    arr = {k: np.ascontiguousarray(z["ct_" + k]) for k in ("intra", "invq", "own", "inter", "bits", "mv0", "mv1", "bef0", "bef1", "aft0", "aft1")}
    p = pb = 0
    two = moved = edge = same = 0
    for i, h in enumerate(hdr):
        nx, ny = int(h[3]), int(h[4])
        n = nx * ny
        sl = slice(p, p + n)
        r0 = arr["bef0"][sl].copy()
        r1 = r0 if h[6] == h[7] else arr["bef1"][sl].copy()
        bits = arr["bits"][pb:pb + (n + 3) // 4].copy()
        mv0, mv1 = arr["mv0"][sl].copy(), arr["mv1"][sl].copy()
        o.kso_ref_cutree_propagate(int(h[5]), nx, ny, ptr(arr["intra"], 2 * p), ptr(arr["invq"], 2 * p), ptr(arr["own"], 2 * p), ptr(arr["inter"], 2 * p), ptr(bits), ptr(mv0), ptr(mv1), ptr(r0), ptr(r1))
        assert (r0 == arr["aft0"][sl]).all() and (r1 == arr["aft1"][sl]).all(), f"call {i} (p0 {h[6]} p1 {h[7]} b {h[8]}): propagated costs differ"
        lists = (np.repeat(bits, 4)[:n] >> (2 * (np.arange(n) & 3))) & 3
        two += int((lists == 3).any()); moved += int(mv0.any() or mv1.any()); same += int(h[6] == h[7])
        x = (mv0.astype(np.int16).astype(np.int32) >> (int(h[5]) + 2)) + np.arange(n) % nx
        edge += int(((x < 0) | (x >= nx - 1)).any())
        p += n; pb += (n + 3) // 4
    assert two >= 20 and moved >= 40 and edge >= 10, (two, moved, edge)
    assert (arr["aft0"] != arr["bef0"]).any() and (arr["aft0"] == 0xffff).sum() >= 0


def test_scenecut_matches_reference_traces():
    z = np.load(GOLD)
    o = lib()
    sc = z["sc"]
    assert len(sc) >= 60
    cuts = flat = 0
    for h in sc:
        got = o.kso_ref_scenecut(int(h[6]), int(h[7]), int(h[8]), int(h[9]) * int(h[10]), int(h[11]), int(h[12]), int(h[13]), int(h[14]), int(h[15]))
        assert got == (int(h[3]) & 0xff), f"scenecut differs: header {h[:16]}"
        T = (int(h[9]) * int(h[10])) << (2 * int(h[11]) - 4)
        cuts += got
        flat += int(h[8] != -1 and (h[8] < T) != (h[7] < T))
    assert cuts >= 6 and len({int(h[12]) for h in sc}) >= 3, "cuts were found; several thresholds"
    print("scenecut calls", len(sc), "cuts", cuts, "flatness changes", flat)
