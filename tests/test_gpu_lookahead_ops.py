"""GPU: the reference's adaptive quantisation and cuTree propagation as device operators (csrc/lookahead_ops.hip: ks265_frame_adapt_quant = calcFrameAdaptQuant
enc@0x4653c0, ks265_cutree_propagate = cuTreePropagate enc@0x47d460) against the oracle restatement that tests/test_lookahead_ref.py pins on recorded calls of the
reference binary - bit for bit, the doubles included: (1) on the recorded calls themselves (tests/golden/lookahead_ref.npz: the device reproduces what the REFERENCE
returned), (2) on seeded inputs at 2160p's block counts against the oracle."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
import torch  # noqa: E402
torch.cuda.is_available()
from oracle_lib import lib as olib, ptr  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lookahead_ref.npz")


@pytest.fixture(scope="module")
def ks():
    from ks265codec_amd.lib import KsContext
    return KsContext(0)


def test_adapt_quant_reproduces_the_reference_calls(ks):
    z = np.load(GOLD)
    py = pc = po = 0
    for i, h in enumerate(z["aq_hdr"]):
        nx, ny, cnt = int(h[3]), int(h[4]), int(h[5])
        n = nx * ny
        Y, U, V = z["aq_y"][py:py + n * 256], z["aq_u"][pc:pc + n * 64], z["aq_v"][pc:pc + n * 64]
        off, inv = ks.frame_adapt_quant(ks.dev(Y), nx * 16, ks.dev(U), ks.dev(V), nx * 8, nx, ny, float(z["aq_strength"][i]), cnt)
        assert (off.ravel()[:cnt] == z["aq_off"][po:po + cnt]).all(), f"call {i}: QP offsets differ from the reference's"
        assert (inv.ravel()[:cnt] == z["aq_inv"][po:po + cnt]).all(), f"call {i}: inverse qscale factors differ from the reference's"
        py += n * 256; pc += n * 64; po += cnt


def test_adapt_quant_at_2160p_matches_oracle(ks):
    from ks265codec_amd.synth import make_clip
    W, H = 3840, 2160
    fr = make_clip(W, H, 1, seed=7, abc=(67, 91, 33), pan=(8, 5))[0]
    nx, ny = W // 16, H // 16
    Y = np.ascontiguousarray(fr[:W * H].reshape(H, W)[:ny * 16]); U = np.ascontiguousarray(fr[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)[:ny * 8]); V = np.ascontiguousarray(fr[W * H * 5 // 4:].reshape(H // 2, W // 2)[:ny * 8])
    o = olib()
    want_off, want_inv = np.zeros(nx * ny), np.zeros(nx * ny, np.uint16)
    for s in (0.4, 1.0, 2.3):
        o.kso_ref_frame_adapt_quant(ptr(Y), ptr(U), ptr(V), nx, ny, nx * ny, C.c_double(s), ptr(want_off), ptr(want_inv))
        off, inv = ks.frame_adapt_quant(ks.dev(Y), W, ks.dev(U), ks.dev(V), W // 2, nx, ny, s)
        assert (off.ravel() == want_off).all() and (inv.ravel() == want_inv).all(), s
    assert len(set(want_inv.tolist())) > 50


def _ct_device(ks, lg, nx, ny, a, same):
    n = nx * ny
    d = {k: ks.dev(np.ascontiguousarray(v)) for k, v in a.items()}
    acc = ks.zeros(16 * n)
    r0 = d["bef0"]; r1 = r0 if same else d["bef1"]
    ks.cutree_propagate(lg, nx, ny, d["intra"], d["invq"], d["own"], d["inter"], d["bits"], d["mv0"], d["mv1"], r0, r1, acc)
    assert not ks.host(acc, np.uint64).any(), "the accumulators are left zero"
    return ks.host(r0, np.uint16), ks.host(r1, np.uint16)


def test_cutree_reproduces_the_reference_calls(ks):
    z = np.load(GOLD)
    names = ("intra", "invq", "own", "inter", "bits", "mv0", "mv1", "bef0", "bef1", "aft0", "aft1")
    p = pb = 0
    for i, h in enumerate(z["ct_hdr"]):
        nx, ny = int(h[3]), int(h[4]); n = nx * ny
        a = {k: (z["ct_" + k][pb:pb + (n + 3) // 4] if k == "bits" else z["ct_" + k][p:p + n]) for k in names}
        g0, g1 = _ct_device(ks, int(h[5]), nx, ny, a, h[6] == h[7])
        assert (g0 == a["aft0"]).all() and (g1 == a["aft1"]).all(), f"call {i} (p0 {h[6]} p1 {h[7]} b {h[8]})"
        p += n; pb += (n + 3) // 4


def test_cutree_at_2160p_block_counts_matches_oracle(ks):
    rng = np.random.default_rng(11)
    nx, ny = 240, 135                                             # 2160p: the lookahead's 16 x 16 blocks
    n = nx * ny
    o = olib()
    for trial in range(3):
        intra = rng.integers(1, 16000, n).astype(np.uint16)
        if trial == 0:
            intra[rng.random(n) < 0.01] = 0                       # a zero intra cost (the reference never stores one): skipped by operator and oracle alike
        a = dict(intra=intra, invq=rng.integers(100, 700, n).astype(np.uint16), own=rng.integers(0, 60000 if trial == 2 else 3000, n).astype(np.uint16),
                 inter=np.minimum(intra, rng.integers(0, 16000, n)).astype(np.uint16), bits=rng.integers(0, 256, (n + 3) // 4).astype(np.uint8),
                 mv0=((rng.integers(-200, 200, n) & 0xffff) | (rng.integers(-200, 200, n) << 16)).astype(np.int32),
                 mv1=((rng.integers(-40, 40, n) & 0xffff) | (rng.integers(-40, 40, n) << 16)).astype(np.int32),
                 bef0=rng.integers(0, 65536 if trial == 2 else 2000, n).astype(np.uint16), bef1=rng.integers(0, 2000, n).astype(np.uint16))
        a["mv0"][rng.random(n) < 0.3] = 0
        same = trial == 1
        w0 = a["bef0"].copy(); w1 = w0 if same else a["bef1"].copy()
        o.kso_ref_cutree_propagate(3, nx, ny, ptr(a["intra"]), ptr(a["invq"]), ptr(a["own"]), ptr(a["inter"]), ptr(a["bits"]), ptr(a["mv0"]), ptr(a["mv1"]), ptr(w0), ptr(w1))
        g0, g1 = _ct_device(ks, 3, nx, ny, a, same)
        assert (g0 == w0).all() and (g1 == w1).all(), trial
        assert (w0 != a["bef0"]).sum() > n // 4
        if trial == 2:
            assert (w0 == 0xffff).sum() > 100, "saturation is exercised"


# ---- calcFrameCost enc@0x4a7410 and the cuTree finish on the device (csrc/lookahead_cost.hip) ---------------------------------------------------------------------------------
def test_calc_frame_cost_reproduces_the_reference_calls(ks):
    """every recorded call of tests/golden/calc_frame_cost.npz: the device returns what the REFERENCE left - vectors, list costs, list bits, intra cost / mode, inter cost, sums, statistics"""
    from cfc_cases import ARR, CFG_WORDS, device_run, load_fixture, repad
    runs, calls, _ = load_fixture()
    for i, r in enumerate(calls):
        h = r["h"]
        w, hh, nx, ny = (int(v) for v in h[7:11]); d0, d1 = int(h[3]), int(h[4]); mx, my = int(h[34]), int(h[35])
        cfgw = {n: int(h[18 + k]) for k, n in enumerate(CFG_WORDS)}
        pl = {k: (repad(r[k], w, hh, mx, my) if r[k].size > 1 else None) for k in ("cur", "ref0", "ref1")}
        arrays = {name: r["b_" + name].copy() for name, _, _ in ARR}
        got, sums, stats, ret, done = device_run(ks, w, hh, nx, ny, cfgw, r["lam"], pl["cur"], pl["ref0"], pl["ref1"], d0, d1, int(h[5]), int(h[17]), (int(h[32]), int(h[33])), int(h[12]),
                                                 arrays, [int(v) for v in h[36:41]], [int(v) for v in h[46:50]], cnt=int(h[11]))
        where = f"call {i} ({runs[r['run']]}; poc {h[14]}, d0 {d0}, d1 {d1})"
        for name, _, _ in ARR:
            if name != "invq":
                assert (got[name] == r["a_" + name]).all(), f"{where}: {name} differs in {int((got[name] != r['a_' + name]).sum())} entries"
        assert sums == [int(v) for v in h[41:46]], f"{where}: sums {sums} != {[int(v) for v in h[41:46]]}"
        assert stats == [int(v) for v in h[50:54]] and ret == int(h[6]) and done == int(h[13]), where


def _lowres(o, frame, W, H):
    from cfc_cases import PAD
    w, h = W // 2, H // 2
    out = np.zeros((h, w), np.uint8)
    o.ks265o_downsample(ptr(out), ptr(np.ascontiguousarray(frame[:W * H])), w, W, w, h)
    return np.ascontiguousarray(np.pad(out, PAD, mode="edge"))


@pytest.mark.parametrize("W,H,lg,preset", [(3840, 2160, 4, 5), (1920, 1080, 3, 5), (1280, 720, 4, 2)])
def test_calc_frame_cost_at_full_sizes_matches_oracle(ks, W, H, lg, preset):
    """2160p's / 1080p's lookahead pictures (the reference uses 16 x 16 blocks from 1080p up): an intra pass, a P pass, a B pass that searches list 1 and reuses list 0, with
    adaptive-quantisation weights; oracle == device on every array and sum"""
    from cfc_cases import ARR, CFG_WORDS, device_run, oracle_run
    from ks265codec_amd.synth import make_clip
    o = olib()
    clip = make_clip(W, H, 3, seed=7, abc=(67, 91, 33), pan=(8, 5))
    p0, cur, p1 = (_lowres(o, clip[t], W, H) for t in range(3))
    w, h = W // 2, H // 2
    nx, ny = (w + (1 << lg) - 1) >> lg, (h + (1 << lg) - 1) >> lg
    n = nx * ny
    rng = np.random.default_rng(3)
    cfgw = dict(zip(CFG_WORDS, (64, lg, 4 if preset <= 2 else 0, 1 if preset <= 2 else 0, 30, preset, 0, 1, 1, 13, 1, 0, 1)))
    lam = np.array([max(1, int(round(0.85 * 2 ** ((q - 12) / 6.0)))) for q in range(52)], np.uint16)
    st = dict(intra=np.zeros(n, np.uint16), imode=np.zeros(n, np.uint8), invq=rng.integers(150, 500, n).astype(np.uint16), inter=np.zeros(n, np.uint16), bits=np.zeros((n + 3) // 4, np.uint8),
              mv0=np.full(n, 0x7fff, np.int32), c0=np.zeros(n, np.int32), mv1=np.full(n, 0x7fff, np.int32), c1=np.zeros(n, np.int32))
    sums, stats, done = [0, -1, -1, -1, -1], [0, 0, 0, 0], 0
    moved = 0
    for (d0, d1, r0, r1, dl) in ((0, 0, cur, cur, (0, 0)), (1, 0, p0, None, (1, 0)), (1, 1, p0, p1, (0, 1))):
        a_o = {k: v.copy() for k, v in st.items()}
        so, to, ro, do = oracle_run(o, w, h, nx, ny, cfgw, lam, cur, r0 if d0 else None, r1 if d1 else None, d0, d1, 0, 0, dl, done, a_o, sums if d0 + d1 == 0 else [sums[0], sums[1], sums[2], -1, -1], stats)
        got, sg, tg, rg, dg = device_run(ks, w, h, nx, ny, cfgw, lam, cur, r0 if d0 else None, r1 if d1 else None, d0, d1, 0, 0, dl, done, st, sums if d0 + d1 == 0 else [sums[0], sums[1], sums[2], -1, -1], stats)
        for name, _, _ in ARR:
            assert (got[name] == a_o[name]).all(), f"({d0}, {d1}): {name} differs in {int((got[name] != a_o[name]).sum())} of {a_o[name].size}"
        assert (sg, tg, rg, dg) == (so, to, ro, do), f"({d0}, {d1}): {(sg, tg, rg, dg)} != {(so, to, ro, do)}"
        st = a_o; done = do; sums = [so[0], so[1], so[2], -1, -1]; stats = to
        moved += int((a_o["mv0"] != 0).sum())
    assert moved > n // 2 and (st["bits"] != 0).any() and len(set(st["imode"].tolist())) > 4


def test_cutree_finish_reproduces_the_reference(ks):
    from cfc_cases import load_fixture
    _, _, fin = load_fixture()
    done = 0
    for i, r in enumerate(fin):
        h = r["h"]
        if not h[5]:
            continue
        cnt = int(h[11])
        out = ks.dev(np.full(cnt, -99.0))
        ks.cutree_finish(cnt, ks.dev(r["intra"]), ks.dev(r["invq"]), ks.dev(r["prop"]), ks.dev(r["aq"]), int(bool(h[8]) and h[6] == 0), out)
        g = ks.host(out, np.float64)
        m = g != -99.0
        assert (g[m] == r["out"][m]).all(), f"finish record {i} (poc {h[3]})"
        want = (((r["intra"].astype(np.int64) * r["invq"] + 128) >> 8) != 0)[:cnt]       # blocks without a weighted intra cost (no AQ plane: inverse qscale 0) keep what they hold
        assert (m == want).all()
        done += int(m.sum() > cnt // 2)
    assert done >= 20
