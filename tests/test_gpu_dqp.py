"""GPU: QP per CTU (ks265_frame_set_qp_map: cu_qp_delta with the quantisation group = the CTU) - the HIP stages against the oracle pipeline with the same random maps, key,
P and B pictures, bit for bit (reconstruction incl. the deblocking at the decoder's QpY, levels, CU map).  That the oracle's pictures are what a decoder makes of the
stream is tests/test_dqp.py (the reference's decoder)."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ks():
    from ks265codec_amd.lib import KsContext
    c = KsContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("W,H,spread,seed", [(416, 240, 6, 5), (200, 136, 12, 6), (1920, 1080, 5, 7)])
def test_qp_per_ctu_matches_oracle(ks, W, H, spread, seed):
    from ks265codec_amd.lib import CU8, KsFrame
    from ks265codec_amd.synth import ENCODER_TOOLS, lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    n = 5 if W < 1000 else 3
    clip = make_clip(W, H, n, seed=seed, abc=(17, 23, 9), pan=(5, 3))
    rng = np.random.default_rng(seed)
    cols, rows = (W + 63) // 64, (H + 63) // 64
    tools = dict(ENCODER_TOOLS, bframes=1, bi_refine=2)                  # (with the joint refinement, as from -preset slower on: the B pictures' QP maps reach it too)
    o = OraclePipeline(W, H, 30, lambda_q4(30), **{k: v for k, v in tools.items() if k != "bframes"})
    with KsFrame(ks, W, H, 30, lambda_q4(30), **tools) as f:
        src = f.new_pic()
        dev, ora = {}, {}
        order = [(0, "I"), (2, "P"), (1, "B")] + ([(4, "P"), (3, "B")] if n >= 5 else [])
        several = 0
        for d, kind in order:
            q = 30 if kind == "I" else 31 + (kind == "B")
            qmap = np.clip(q + rng.integers(-spread, spread + 1, cols * rows), 10, 51).astype(np.int8)
            qmap[rng.random(cols * rows) < 0.3] = q
            lam = lambda_q4(q, inter=kind != "I")
            o.set_qp(q, lam); f.set_qp(q, lam)
            o.set_qp_map(qmap)
            dmap = ks.dev(qmap.view(np.uint8))
            f.set_qp_map(dmap)
            out = f.new_pic()
            f.load_i420(ks.dev(clip[d]), src)
            if kind == "I":
                ora[d] = o.encode(clip[d], "I"); f.encode_picture(src, src, True, out)
            elif kind == "P":
                ora[d] = o.encode(clip[d], "P", ora[d - 2]); f.encode_picture(src, dev[d - 2], False, out)
            else:
                ora[d] = o.encode(clip[d], "B", ora[d - 1], ora[d + 1]); f.encode_picture_b(src, dev[d - 1], dev[d + 1], out)
            dev[d] = out
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(ora[d])
            assert (got == exp).all(), f"{W}x{H} picture {d} ({kind}): {int((got != exp).sum())} reconstructed samples differ with a QP per CTU"
            for c in range(3):
                lv = f.ws_read("levels", o.lvl[c].size * 2, c).view(np.int16)
                assert (lv == o.lvl[c]).all(), f"picture {d}: levels of component {c} differ"
            gc = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
            assert (gc == o.cu8).all(), f"picture {d}: CU map differs"
            several += len(set(o.effective_qp().tolist())) > 2
        assert several >= 2
        f.set_qp_map(None); o.set_qp_map(None)
