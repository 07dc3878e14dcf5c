"""GPU: the reference's rate-distortion optimised quantisation as a device operator (csrc/rdoq_ops.hip: ks265_rdoq_batch = h265_codec::rdoQuant enc@0x4aac50, SURVEY.md 8(f)
rank 3).  (1) The calls recorded inside real `appencoder` runs (tests/golden/rdoq.npz: -preset medium / slow / veryslow, QP 22..37, -bframes 3, -sbh 0,
non-default lambda weights; each with the bit table estBitRdoq had built for it): the MI355X returns the REFERENCE's levels (signs included), count, last position,
significance masks and hidden-sign mask.  (2) Transform blocks of THIS pipeline (a P picture's and a key picture's residuals through the pinned forward transform and
quantiser, every block size, luma and chroma) under tables taken from the fixture: device == the oracle restatement (oracle/ks265_rdoq_ref.c, pinned on the same fixture)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
import torch  # noqa: E402
torch.cuda.is_available()
from oracle_lib import lib as olib, ptr  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdoq.npz")


@pytest.fixture(scope="module")
def ks():
    from ks265codec_amd.lib import KsContext
    return KsContext(0)


def test_rdoq_reproduces_the_reference_calls(ks):
    from ks265codec_amd.lib import RDOQ_TU
    z = np.load(GOLD)
    meta, lam, offs = z["meta"], z["lam"], z["offs"]
    n = len(meta)
    tus = np.zeros(n, RDOQ_TU)
    tus["off"] = offs[:-1]; tus["tab"] = np.arange(n); tus["dq"] = meta[:, 3]; tus["last_pos"] = meta[:, 6]
    tus["lam"] = lam[:, 0]; tus["lam_sdh"] = lam[:, 1]
    tus["log2"] = meta[:, 0]; tus["scan_idx"] = meta[:, 1]; tus["comp"] = meta[:, 2]; tus["per"] = meta[:, 4]; tus["tu5"] = meta[:, 5]; tus["flag_a4c0"] = meta[:, 7]; tus["sdh"] = meta[:, 8]
    lvl, mask, out, hid = ks.rdoq(tus, z["lvl_in"], z["coef"], z["tab"], z["mask_in"])
    bad = []
    for i in range(n):
        a, b = int(offs[i]), int(offs[i + 1])
        ncg = max(1, (b - a) // 16)
        ok = (lvl[a:b] == z["lvl_out"][a:b]).all() and out[i, 0] == meta[i, 9] and out[i, 1] == meta[i, 10] and int(hid[i]) == int(z["hidden"][i]) and (mask[i, :ncg] == z["mask_out"][i][:ncg]).all()
        if not ok:
            bad.append((i, meta[i].tolist(), bool((lvl[a:b] == z["lvl_out"][a:b]).all()), out[i].tolist(), int(hid[i]), int(z["hidden"][i])))
    assert not bad, f"{len(bad)} of {n} recorded calls differ on the device; first: {bad[:3]}"
    assert n >= 1200


def _pipeline_blocks(rng):
    """transform blocks of this pipeline: residuals of a moving textured clip against its previous picture (inter) and against a flat prediction (intra-like: large
    low-frequency content), through the pinned forward transform and the quantiser that feeds rdoQuant (rounding at 1/2), all block sizes"""
    from ks265codec_amd.synth import make_clip
    o = olib()
    W, H = 416, 240
    clip = make_clip(W, H, 3, seed=5, abc=(17, 23, 9))
    Y = [c[:W * H].reshape(H, W).astype(np.int16) for c in clip]
    U = [c[W * H:W * H * 5 // 4].reshape(H // 2, W // 2).astype(np.int16) for c in clip]
    blocks = []
    for log2 in (2, 3, 4, 5):
        N = 1 << log2
        for comp, planes in ((0, Y), (1, U)):
            if comp and log2 == 5:
                continue
            hh, ww = planes[0].shape
            for t in range(24):
                y0, x0 = int(rng.integers(0, hh - N)), int(rng.integers(0, ww - N))
                cur = planes[2][y0:y0 + N, x0:x0 + N]
                pred = planes[1][y0:y0 + N, x0:x0 + N] if t % 3 else np.full((N, N), int(cur.mean()), np.int16)
                res = np.ascontiguousarray(cur - pred, np.int16)
                coef, tmp = np.zeros((N, N), np.int16), np.zeros((N, N), np.int16)
                o.ks265o_fwd_transform(log2 - 1, ptr(res), ptr(coef), N, N, ptr(tmp))
                blocks.append((log2, comp, coef))
    return blocks


def test_rdoq_on_pipeline_blocks_matches_oracle(ks):
    from ks265codec_amd.lib import RDOQ_TU
    z = np.load(GOLD)
    rng = np.random.default_rng(3)
    o = olib()
    blocks = _pipeline_blocks(rng)
    kScale, kInv = [26214, 23302, 20560, 18396, 16384, 14564], [40, 45, 51, 57, 64, 72]
    tus, lv_all, cf_all, masks, tabs, want = [], [], [], [], [], []
    off = 0
    for bi, (log2, comp, coef) in enumerate(blocks):
        N = 1 << log2
        qp = int(rng.integers(22, 38))
        per, scale, dq = qp // 6, kScale[qp % 6], kInv[qp % 6] << (qp // 6)
        qbits = 21 + per - log2
        a = np.abs(coef.astype(np.int64))
        q = np.minimum(32767, (a * scale + (1 << (qbits - 1))) >> qbits)
        lvl = np.where(coef < 0, -q, q).astype(np.int16).ravel()
        scan_idx = int(rng.integers(0, 3)) if (log2 <= 3 and (comp == 0 or log2 == 2)) else 0
        sm = np.zeros(64, np.uint16)
        last = o.kso_rdoq_scan_flags(ptr(lvl), log2, scan_idx, ptr(sm))
        if last < 0:
            continue
        m = np.flatnonzero((z["meta"][:, 0] == log2) & ((z["meta"][:, 2] > 0) == bool(comp)))      # a table the reference built for this size and component
        T = np.ascontiguousarray(z["tab"][m[int(rng.integers(0, len(m)))]], np.int32)
        lam0 = 0.85 * 2.0 ** ((qp - 12) / 3.0)
        lam, lam_sdh = int((90 if comp else 256) * lam0 + 0.5), int((90 if comp else 256) * lam0 + 0.5)
        sdh, tu5, fa = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        tus.append((off, len(tabs), dq, last, lam, lam_sdh, log2, scan_idx, comp, per, tu5, fa, sdh, 0))
        lv_all.append(lvl.copy()); cf_all.append(coef.ravel().copy()); masks.append(sm.copy()); tabs.append(T)
        l2, sm2 = lvl.copy(), sm.copy()
        ol, oh = C.c_int32(0), C.c_uint64(0)
        cf = np.ascontiguousarray(coef.ravel())
        ret = o.kso_ref_rdo_quant(ptr(l2), ptr(cf), log2, scan_idx, comp, dq, per, C.c_int64(lam), C.c_int64(lam_sdh), ptr(T), tu5, last, ptr(sm2), fa, sdh, C.byref(ol), C.byref(oh))
        want.append((l2, sm2, ret, ol.value, oh.value))
        off += N * N
    tus = np.array(tus, dtype=RDOQ_TU)
    lvl, mask, out, hid = ks.rdoq(tus, np.concatenate(lv_all), np.concatenate(cf_all), np.stack(tabs), np.stack(masks))
    assert len(tus) >= 120
    changed = 0
    for i, (l2, sm2, ret, last, hidden) in enumerate(want):
        a = int(tus["off"][i]); n2 = len(l2); ncg = max(1, n2 // 16)
        assert (lvl[a:a + n2] == l2).all() and out[i, 0] == ret and out[i, 1] == last and int(hid[i]) == hidden and (mask[i, :ncg] == sm2[:ncg]).all(), \
            f"block {i} (log2 {tus['log2'][i]}, comp {tus['comp'][i]}, scan {tus['scan_idx'][i]}, sdh {tus['sdh'][i]}): device != oracle"
        changed += bool((np.abs(l2) != np.abs(lv_all[i])).any())
    assert changed >= len(want) // 3, "rdoQuant changed too few of the pipeline's blocks for this to be a test"
    assert {int(t) for t in tus["log2"]} == {2, 3, 4, 5}
