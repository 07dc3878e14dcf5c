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
