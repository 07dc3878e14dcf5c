"""GPU parity, whole-frame stages: every stage of include/ks265_hip.h §3 (through the C ABI) must equal the CPU
restatement oracle/ks265_pipeline_oracle.c bit for bit on the same seeded inputs, stage by stage and end to end."""
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


def _cmp_region(name, a, b, stride, org, w, h, margin=0):
    """compare the [-margin, w+margin) x [-margin, h+margin) window of two padded planes"""
    A = a.reshape(-1, stride)
    B = b.reshape(-1, stride)
    oy, ox = divmod(org, stride)
    sa = A[oy - margin:oy + h + margin, ox - margin:ox + w + margin]
    sb = B[oy - margin:oy + h + margin, ox - margin:ox + w + margin]
    bad = np.argwhere(sa != sb)
    assert len(bad) == 0, f"{name}: {len(bad)} mismatches, first at (y,x)={bad[0] - margin} gpu={sa[tuple(bad[0])]} oracle={sb[tuple(bad[0])]}"


@pytest.mark.parametrize("W,H,seed,me", [(416, 240, 1234, 0), (200, 136, 5, 0), (1280, 720, 43, 0), (416, 240, 77, 1), (1280, 720, 9, 1), (416, 240, 31, 2), (200, 136, 8, 2), (1280, 720, 21, 2)])
def test_stages_match_oracle(ks, W, H, seed, me):
    from ks265codec_amd.lib import CU8, PU, SAO_PARAM, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline

    nfr = 3 if W <= 416 else 2
    clip = make_clip(W, H, nfr, seed=seed, abc=(17, 23, 9))
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=me)
    f = KsFrame(ks, W, H, 27, lambda_q4(27), me_method=me)
    g = f.geom
    org_y, org_c = g.pad_y * g.stride_y + g.pad_y, g.pad_c * g.stride_c + g.pad_c
    src, ref, deb, dst = f.new_pic(), f.new_pic(), f.new_pic(), f.new_pic()
    planes = ks.zeros(16 * g.bytes_y)
    pu = [ks.zeros(g.bytes_pu), ks.zeros(g.bytes_pu)]
    cu8, sao = ks.zeros(g.bytes_cu8), ks.zeros(g.bytes_sao)
    lvl = [ks.zeros(W * H * 2), ks.zeros(W * H // 2), ks.zeros(W * H // 2)]
    have_prev = False
    for t in range(nfr):
        qp = 27 if t == 0 else 28
        o.set_qp(qp, lambda_q4(qp)); f.set_qp(qp, lambda_q4(qp))
        key = t == 0
        o_ref_before = o.ref
        o.encode_picture(clip[t], key)
        f.load_i420(ks.dev(clip[t]), src)
        _cmp_region("src.y", ks.host(src.y, np.uint8), o.src.y, g.stride_y, org_y, W, H, margin=g.pad_y)
        _cmp_region("src.u", ks.host(src.u, np.uint8), o.src.u, g.stride_c, org_c, W // 2, H // 2, margin=g.pad_c)
        if key:
            f.cu_flat_intra(cu8)
        else:
            f.ref_planes(ref, planes)
            P = ks.host(planes, np.uint8).reshape(16, -1)
            OP = o.planes.reshape(16, -1)
            for k in range(16):
                _cmp_region(f"plane{k}", P[k], OP[k], g.stride_y, org_y, W, H, margin=72)
            f.me_integer(src, ref, pu[1] if have_prev else None, pu[0])
            got = ks.host(pu[0], PU)
            assert (got == o.pu_int).all(), f"integer ME: {int((got != o.pu_int).sum())} PU records differ (frame {t})"
            f.me_subpel(src, planes, pu[0])
            got = ks.host(pu[0], PU)
            exp = o.prev_pu  # the oracle swapped its buffers after the picture
            assert (got == exp).all(), f"sub-pel ME: {int((got != exp).sum())} PU records differ (frame {t})"
            f.cu_decide(pu[0], cu8)
        f.reconstruct(src, ref, planes, cu8, lvl, deb)
        gc = ks.host(cu8, CU8)
        assert (gc == o.cu8).all(), f"cu8 map differs in {int((gc != o.cu8).sum())} blocks (frame {t})"
        for c in range(3):
            assert (ks.host(lvl[c], np.int16) == o.lvl[c]).all(), f"levels comp {c} differ (frame {t})"
        _cmp_region("recon.y", ks.host(deb.y, np.uint8), o.rec_pre[0], g.stride_y, org_y, W, H)
        _cmp_region("recon.u", ks.host(deb.u, np.uint8), o.rec_pre[1], g.stride_c, org_c, W // 2, H // 2)
        _cmp_region("recon.v", ks.host(deb.v, np.uint8), o.rec_pre[2], g.stride_c, org_c, W // 2, H // 2)
        f.deblock(cu8, deb)
        _cmp_region("deblock.y", ks.host(deb.y, np.uint8), o.rec.y, g.stride_y, org_y, W, H)
        _cmp_region("deblock.u", ks.host(deb.u, np.uint8), o.rec.u, g.stride_c, org_c, W // 2, H // 2)
        _cmp_region("deblock.v", ks.host(deb.v, np.uint8), o.rec.v, g.stride_c, org_c, W // 2, H // 2)
        f.sao(src, deb, sao, dst)
        gs = ks.host(sao, SAO_PARAM)
        assert (gs == o.sao).all(), f"SAO params differ for {int((gs != o.sao).sum())} CTU components (frame {t})"
        _cmp_region("final.y", ks.host(dst.y, np.uint8), o.ref.y, g.stride_y, org_y, W, H, margin=g.pad_y)
        _cmp_region("final.u", ks.host(dst.u, np.uint8), o.ref.u, g.stride_c, org_c, W // 2, H // 2, margin=g.pad_c)
        _cmp_region("final.v", ks.host(dst.v, np.uint8), o.ref.v, g.stride_c, org_c, W // 2, H // 2, margin=g.pad_c)
        ref, dst = dst, ref
        if not key:
            pu.reverse()
            have_prev = True
    f.close()


@pytest.mark.parametrize("W,H", [(416, 240), (1920, 1080)])
def test_encode_picture_end_to_end(ks, W, H):
    """ks265_encode_picture (the bench path) == oracle pipeline over a short GOP, via recon I420 + PSNR sanity"""
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip, psnr
    from oracle_lib import OraclePipeline

    nfr = 4 if W <= 416 else 2
    clip = make_clip(W, H, nfr, seed=42)
    o = OraclePipeline(W, H, 27, lambda_q4(27))
    f = KsFrame(ks, W, H, 27, lambda_q4(27))
    src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
    for t in range(nfr):
        qp = 27 if t == 0 else 28
        o.set_qp(qp, lambda_q4(qp)); f.set_qp(qp, lambda_q4(qp))
        exp = o.encode_picture(clip[t], t == 0)
        f.load_i420(ks.dev(clip[t]), src)
        f.encode_picture(src, a, t == 0, b)
        got = ks.host(f.store_i420(b), np.uint8)
        assert (got == exp).all(), f"frame {t}: {int((got != exp).sum())} recon bytes differ"
        sse = f.sse_picture(src, b)
        y0, y1 = clip[t][:W * H].astype(np.int64), got[:W * H].astype(np.int64)
        assert int(sse[0]) == int(((y0 - y1) ** 2).sum())
        assert psnr(clip[t][:W * H], got[:W * H]) > 30.0
        a, b = b, a
    f.close()
