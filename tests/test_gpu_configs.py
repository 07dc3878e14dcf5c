"""GPU parity on the configurations BASELINE.json names (VERDICT r1 "configs_untested"): what bench.py times is what is checked.

  config 1  1280x720  veryfast (-me 1 HEX, qp 32)          -> test_config1_720p_hex_qp32
  config 2  1920x1080 slow (-me 2 UMH, qp 27)              -> test_config2_1080p_umh
  config 3  3840x2160 slow (-me 2 UMH, qp 27) = bench.py   -> test_config3_2160p_umh (2 pictures vs the OpenMP oracle)
  config 4  -bframes 3 pixel path with UMH at 720p         -> test_config4_bframes3_umh_720p
  bounded randomised sweep (40 configurations) collected by pytest, log kept under gpurun_out/ -> test_fuzz_bounded
"""
from __future__ import annotations

import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ks():
    from ks265codec_amd.lib import KsContext
    c = KsContext(0)
    yield c
    c.close()


# the tool set the C host (ks265_enc.c) and bench.py switch on for -preset slow: what is timed is what is checked (VERDICT r2 "weak 1a")
from ks265codec_amd.synth import ENCODER_TOOLS as _HOST_TOOLS      # the ONE dict bench.py, smoke() and rd_eval.py --host use too
ENCODER_TOOLS = {k: v for k, v in _HOST_TOOLS.items() if k not in ("me_method", "me_hex_thr")}     # the search method travels separately in these tests


def _ippp(ks, W, H, qp, me, nfr, seed, abc=(37, 53, 19), pan=(5, 3), hidden_offset=True, hex_thr=0, **tools):
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip, psnr
    from oracle_lib import OraclePipeline

    clip = make_clip(W, H, nfr, seed=seed, abc=abc, pan=pan)
    o = OraclePipeline(W, H, qp, lambda_q4(qp), me_method=me, me_hex_thr=hex_thr, **tools)
    with KsFrame(ks, W, H, qp, lambda_q4(qp), me_method=me, me_hex_thr=hex_thr, **tools) as f:
        src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
        for t in range(nfr):
            q = qp + (1 if (t > 0 and hidden_offset) else 0)          # the reference's hidden hierarchy offset: I = Q, P = Q+1
            lam = lambda_q4(q, inter=t > 0 and bool(tools.get("rdo")))      # with the encoder's tool set: the host's lambda table of P pictures
            o.set_qp(q, lam); f.set_qp(q, lam)
            exp = o.encode_picture(clip[t], t == 0)
            f.load_i420(ks.dev(clip[t]), src)
            f.encode_picture(src, a, t == 0, b)
            got = ks.host(f.store_i420(b), np.uint8)
            assert (got == exp).all(), f"{W}x{H} me={me} qp={q} picture {t}: {int((got != exp).sum())} recon bytes differ"
            assert psnr(clip[t][:W * H], got[:W * H]) > 28.0
            a, b = b, a


def test_config1_720p_hex_qp32(ks):
    _ippp(ks, 1280, 720, 32, 1, 3, seed=43)


def test_config2_1080p_umh(ks):
    """-preset slow = -me 2 with the interMeHex shortcut below 16 SAD units per sample (tME+0x368 = 16)"""
    _ippp(ks, 1920, 1080, 27, 2, 3, seed=42, hex_thr=16)


def test_config5_tool_set_1080p(ks):
    """BASELINE config 5 = -preset veryslow: the tool set the reference resolves it to - UMH always (tME+0x368 = 0), -subme 2 judged by Hadamard, -part 1 (2NxN / Nx2N
    partitions) - on top of the encoder host's tools, 1920x1080 key picture + two P pictures == oracle"""
    from ks265codec_amd.synth import subme_knobs
    tools = dict(ENCODER_TOOLS, part=1, **subme_knobs("veryslow"))
    _ippp(ks, 1920, 1080, 27, 2, 3, seed=42, hex_thr=0, **tools)


def test_config5_tool_set_2160p(ks):
    """the same tool set at config 5's own size, 3840x2160 (key picture + one P picture == oracle)"""
    from ks265codec_amd.synth import subme_knobs
    tools = dict(ENCODER_TOOLS, part=1, **subme_knobs("veryslow"))
    _ippp(ks, 3840, 2160, 27, 2, 2, seed=7, abc=(67, 91, 33), pan=(8, 5), hex_thr=0, **tools)


def test_config5_1080p_umh_always(ks):
    """-preset veryslow resolves tME+0x368 to 0: interMeUMH for every PU"""
    _ippp(ks, 1920, 1080, 27, 2, 3, seed=42, hex_thr=0)


def test_config3_2160p_umh(ks):
    """the exact bench workload (3840x2160, UMH, qp 27/28, deblock + SAO): key picture + one P picture against the OpenMP oracle"""
    _ippp(ks, 3840, 2160, 27, 2, 2, seed=7, abc=(67, 91, 33), pan=(8, 5), hex_thr=16)


@pytest.mark.parametrize("W,H,me,rng_,pan", [(64, 48, 0, 64, (9, 6)), (416, 240, 1, 64, (9, 6)), (1280, 720, 1, 64, (70, -0)), (1920, 1080, 2, 64, (100, 75)), (600, 344, 2, 16, (9, 6)), (3840, 2160, 2, 64, (9, 6)),
                                              (1280, 720, 2, 64, (120, 90))])
def test_presearch_field(ks, W, H, me, rng_, pan):
    """stage A0: the pre-search vector field (pyramid of downsample_c pictures: exhaustive L2 search, +-2 at L1, +-1 at full resolution) equals
    the oracle's, incl. pictures whose size is no multiple of 16 / 32 (partial blocks) and a short search range"""
    import ctypes as C
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import HostPic, OraclePipeline, ptr
    clip = make_clip(W, H, 2, seed=W + H, abc=(37, 53, 19), pan=pan)          # pans beyond the search range: the window offsets must follow
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=me, me_range=rng_, pre_search=1)
    a, b = HostPic(o.geom), HostPic(o.geom)
    o.load(a, clip[0]); o.load(b, clip[1])
    nbx, nby = (W + 15) // 16, (H + 15) // 16
    exp = np.zeros((nby, nbx, 2), np.int16)
    exp_off = np.zeros((o.geom.ctu_rows, o.geom.ctu_cols, 2), np.int16)
    o.o.kso_presearch(C.byref(o.cfg), b.c(), a.c(), ptr(exp), ptr(exp_off))
    with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=me, me_range=rng_, pre_search=1) as f:
        src, ref = f.new_pic(), f.new_pic()
        f.load_i420(ks.dev(clip[1]), src); f.load_i420(ks.dev(clip[0]), ref)
        got, got_off = f.presearch(src, ref)
    assert (got_off == exp_off).all(), f"{int((got_off != exp_off).any(-1).sum())} window offsets differ"
    assert (got == exp).all(), f"{int((got != exp).any(-1).sum())} of {nbx * nby} vectors differ"
    assert (np.abs(exp) <= 3 * rng_).all() and (exp != 0).any()


def test_config1_720p_hex_qp32_presearch(ks):
    _ippp(ks, 1280, 720, 32, 1, 4, seed=43, pre_search=1, merge=1)


def test_fast_pan_window_offset(ks):
    """a pan of (72, 44) samples per picture is out of reach of a +-64 window around zero: the 1/8-resolution vote moves every CTU's window (ctu_off),
    stage A searches around it, vectors beyond +-64 reach the sub-pel stage, the merge pass and the reconstruction"""
    _ippp(ks, 1280, 720, 30, 1, 3, seed=5, pan=(72, 44), pre_search=1, merge=1)
    _ippp(ks, 832, 480, 27, 2, 3, seed=6, pan=(-0, 90), hex_thr=16, pre_search=1, merge=1)


def test_config3_2160p_umh_presearch(ks):
    """the bench workload with the pre-search candidates on"""
    _ippp(ks, 3840, 2160, 27, 2, 3, seed=7, abc=(67, 91, 33), pan=(8, 5), hex_thr=16, pre_search=1, merge=1)


@pytest.mark.parametrize("W,H,abc,pan,seed", [(1920, 1080, (37, 53, 19), (5, 3), 42), (3840, 2160, (67, 91, 33), (8, 5), 7)])
def test_anchor_with_three_past_anchors_encoder_tools(ks, W, H, abc, pan, seed):
    """round 6 (-ref0 3 = what -preset slow resolves to, VERDICT r5 missing 3): an anchor of the pyramid as a multi-reference P picture with the host's whole tool set
    (pre-search, propagation, intra candidates against the two-list records, merge pass on the records' pictures, intra CUs, group pruning): key picture, then anchors 8
    pictures apart with 1, 2 and 3 pictures in list 0 == oracle, CU records included.  One of the pictures repeats an older anchor (the ping-pong clips of the same-clip
    tables): its blocks go to the identical picture"""
    from ks265codec_amd.lib import CU8, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    base = make_clip(W, H, 17, seed=seed, abc=abc, pan=pan)
    clip = [base[0], base[8], base[16], base[8]]                    # anchors of a ping-pong clip: 0, 8, 16, 24 (= 8 again)
    tools = dict(ENCODER_TOOLS)
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, **tools)
    with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, bframes=7, refs=3, **tools) as f:
        src = f.new_pic()
        dg, do = [], []
        for t in range(4):
            q = 27 if t == 0 else 28
            lam = lambda_q4(q, inter=t > 0)
            o.set_qp(q, lam); f.set_qp(q, lam)
            f.load_i420(ks.dev(clip[t]), src)
            out = f.new_pic()
            if t == 0:
                eo = o.encode(clip[0], "I"); f.encode_picture(src, out, True, out)
            else:
                eo = o.encode_mref(clip[t], do[:3]); f.encode_picture_mref(src, dg[:3], out)
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(eo)
            assert (got == exp).all(), f"{W}x{H} anchor {t}: {int((got != exp).sum())} recon bytes differ"
            cu = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
            assert (cu.view(np.uint8) == o.cu8.view(np.uint8)).all(), f"anchor {t}: CU records differ"
            if t == 3:
                inter = cu["pred_mode"] == 0
                assert ((cu["inter_dir"][inter] >> 4) & 3 == 1).mean() > 0.5, "the repeated picture is the second entry of list 0"
                assert (cu["cbf"][inter] == 0).mean() > 0.5
            dg.insert(0, out); do.insert(0, eo)


@pytest.mark.parametrize("W,H,abc,pan,seed", [(1920, 1080, (37, 53, 19), (5, 3), 42), (3840, 2160, (67, 91, 33), (8, 5), 7), (416, 240, (17, 23, 9), (3, 2), 5), (200, 136, (17, 23, 9), (2, 1), 6)])
def test_skip_pass(ks, W, H, abc, pan, seed):
    """round 6 (VERDICT r5 missing 2: the decision on the coded distortion): cfg.skip_rd - after the reconstruction, an inter CU with residual whose merge candidate without residual is
    the cheaper coding (SSE of the real reconstruction + the bits of its levels against SSE of the candidate's prediction) drops its residual and takes the candidate's motion.
    Key picture, P, P, then the B pictures of a pyramid of 4 between them == oracle: reconstruction, CU records, all three level planes (skip_rd 2: P pictures too); the pass
    really acts (more blocks without residual than a run without it)"""
    from ks265codec_amd.lib import CU8, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    clip = make_clip(W, H, 9, seed=seed, abc=abc, pan=pan)
    order = [(0, "I", None, None, 0), (4, "P", 0, None, 1), (8, "P", 4, None, 1), (2, "B", 0, 4, 2), (1, "B", 0, 2, 4), (3, "B", 2, 4, 4), (6, "B", 4, 8, 2)]
    stats = {}
    for skip in (2, 0):
        tools = dict(ENCODER_TOOLS, skip_rd=skip)
        o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, **tools)
        with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, bframes=3, **tools) as f:
            src = f.new_pic()
            dg, do = {}, {}
            for (d, kind, r0, r1, dq) in order:
                q = 27 + dq
                lam = lambda_q4(q, inter=kind != "I")
                o.set_qp(q, lam); f.set_qp(q, lam)
                f.load_i420(ks.dev(clip[d]), src)
                out = f.new_pic()
                if kind == "I":
                    eo = o.encode(clip[d], "I"); f.encode_picture(src, out, True, out)
                elif kind == "P":
                    eo = o.encode(clip[d], "P", do[r0]); f.encode_picture(src, dg[r0], False, out)
                else:
                    eo = o.encode(clip[d], "B", do[r0], do[r1]); f.encode_picture_b(src, dg[r0], dg[r1], out)
                got, exp = ks.host(f.store_i420(out), np.uint8), o.store(eo)
                cu = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
                assert (cu.view(np.uint8) == o.cu8.view(np.uint8)).all(), f"skip_rd {skip} picture {d} ({kind}): {int((cu.view(np.uint8) != o.cu8.view(np.uint8)).sum())} CU record bytes differ"
                for comp, n in ((0, W * H), (1, W * H // 4), (2, W * H // 4)):
                    lv = f.ws_read("levels", n * 2, comp).view(np.int16)
                    assert (lv == o.lvl[comp]).all(), f"skip_rd {skip} picture {d} ({kind}): {int((lv != o.lvl[comp]).sum())} levels of component {comp} differ"
                assert (got == exp).all(), f"skip_rd {skip} picture {d} ({kind}): {int((got != exp).sum())} recon bytes differ"
                if kind != "I":
                    inter = cu["pred_mode"] == 0
                    stats.setdefault(skip, []).append((cu["cbf"][inter] == 0).mean())
                dg[d], do[d] = out, eo
    assert np.mean(stats[2]) > np.mean(stats[0]), (stats[2], stats[0])


@pytest.mark.parametrize("W,H,abc,pan,seed", [(1920, 1080, (37, 53, 19), (5, 3), 42), (3840, 2160, (67, 91, 33), (8, 5), 7), (200, 136, (17, 23, 9), (2, 1), 6)])
def test_tools_per_picture(ks, W, H, abc, pan, seed):
    """round 6: ks265_frame_set_picture_tools - on ONE frame object created with the host's tool set, the B pictures nothing predicts from are coded without intra candidates,
    without the joint refinement and without SAO (reconstructed and deblocked straight in the output picture), the others with everything: every picture == the oracle pipeline
    given the same per-picture settings - reconstruction, CU records, levels, SAO records ("off" for the lean pictures); values the workspace was not made for are refused"""
    from ks265codec_amd.lib import CU8, SAO_PARAM, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    clip = make_clip(W, H, 9, seed=seed, abc=abc, pan=pan)
    order = [(0, "I", None, None, 0, False), (4, "P", 0, None, 1, False), (2, "B", 0, 4, 2, False), (1, "B", 0, 2, 4, True), (3, "B", 2, 4, 4, True), (8, "P", 4, None, 1, False), (6, "B", 4, 8, 2, False), (5, "B", 4, 6, 4, True)]
    tools = dict(ENCODER_TOOLS, bi_refine=2)                            # (the joint refinement as from -preset slower on: switched per picture here)
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, **tools)
    with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, bframes=3, **tools) as f:
        with pytest.raises(Exception):
            f.set_picture_tools(sao=2)                                    # the frame object was created with cfg.sao = 1
        src = f.new_pic()
        dg, do = {}, {}
        for (d, kind, r0, r1, dq, lean) in order:
            q = 27 + dq
            lam = lambda_q4(q, inter=kind != "I")
            o.set_qp(q, lam); f.set_qp(q, lam)
            t = ((0, 0, 0, 1) if W == 1920 else (0, 0, 0)) if lean else (-1, -1, -1)       # (1080p: the lean pictures also search with interMeHex instead of interMeUMH)
            o.set_picture_tools(*t); f.set_picture_tools(*t)
            f.load_i420(ks.dev(clip[d]), src)
            out = f.new_pic()
            if kind == "I":
                eo = o.encode(clip[d], "I"); f.encode_picture(src, out, True, out)
            elif kind == "P":
                eo = o.encode(clip[d], "P", do[r0]); f.encode_picture(src, dg[r0], False, out)
            else:
                eo = o.encode(clip[d], "B", do[r0], do[r1]); f.encode_picture_b(src, dg[r0], dg[r1], out)
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(eo)
            cu = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
            assert (cu.view(np.uint8) == o.cu8.view(np.uint8)).all(), f"picture {d} ({kind}, lean {lean}): {int((cu.view(np.uint8) != o.cu8.view(np.uint8)).sum())} CU record bytes differ"
            for comp, n in ((0, W * H), (1, W * H // 4), (2, W * H // 4)):
                lv = f.ws_read("levels", n * 2, comp).view(np.int16)
                assert (lv == o.lvl[comp]).all(), f"picture {d} ({kind}, lean {lean}): {int((lv != o.lvl[comp]).sum())} levels of component {comp} differ"
            rec = f.ws_read("sao", f.geom.bytes_sao).view(SAO_PARAM)
            assert (rec.view(np.uint8) == o.sao.view(np.uint8)).all(), f"picture {d} ({kind}, lean {lean}): SAO records differ"
            assert (got == exp).all(), f"picture {d} ({kind}, lean {lean}): {int((got != exp).sum())} recon bytes differ"
            if lean:
                assert (rec["type"] == -1).all() and (cu["pred_mode"] == 0).all()
            elif kind == "B":
                assert (rec["type"] != -1).any()
            dg[d], do[d] = out, eo


@pytest.mark.parametrize("W,H", [(1920, 1080), (416, 240), (200, 136)])
def test_reference_sao_decision(ks, W, H):
    """round 6 (VERDICT r5 missing 5): cfg.sao = 2 (-sao 3) - the decision of CEncSao::modeDecisionCtu enc@0x4af690 on its -sao 4 path (EO class 0, EO class 1, band offset per
    component group; the reference's pinned estimation functions, rates and lambda table) on the device == oracle, SAO records included; the records differ from the default rule's"""
    from ks265codec_amd.lib import SAO_PARAM, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    clip = make_clip(W, H, 3, seed=W, abc=(37, 53, 19), pan=(5, 3))
    tools = dict(ENCODER_TOOLS)
    recs = {}
    for sao in (2, 1):
        o = OraclePipeline(W, H, 29, lambda_q4(29), me_method=2, me_hex_thr=16, sao=sao, **tools)
        with KsFrame(ks, W, H, 29, lambda_q4(29), me_method=2, me_hex_thr=16, sao=sao, **tools) as f:
            src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
            for t in range(3):
                q = 29 + (t > 0)
                lam = lambda_q4(q, inter=t > 0)
                o.set_qp(q, lam); f.set_qp(q, lam)
                exp = o.encode_picture(clip[t], t == 0)
                f.load_i420(ks.dev(clip[t]), src)
                f.encode_picture(src, a, t == 0, b)
                got = ks.host(f.store_i420(b), np.uint8)
                rec = f.ws_read("sao", f.geom.bytes_sao).view(SAO_PARAM)
                assert (rec.view(np.uint8) == o.sao.view(np.uint8)).all(), f"sao {sao} picture {t}: SAO records differ"
                assert (got == exp).all(), f"sao {sao} picture {t}: {int((got != exp).sum())} recon bytes differ"
                a, b = b, a
            recs[sao] = rec.copy()
    assert set(np.unique(recs[2]["type"])) <= {-1, 0, 1, 2} and (recs[2]["type"] >= 0).any()
    assert (recs[1].view(np.uint8) != recs[2].view(np.uint8)).any()


@pytest.mark.parametrize("W,H", [(1920, 1080), (416, 240), (200, 136)])
def test_rdoq_in_the_pixel_path(ks, W, H):
    """round 6 (VERDICT r5 missing 4, cfg.rdoq): with tables set (ks265_frame_set_rdoq) the luma transform blocks of inter CUs go through the reference's rdoQuant between the
    two halves of the reconstruction (front: transform + levels rounded at 1 / 2; rdoq_prep / rdoq / unpack kernels; back: dequantisation + inverse transform) - P pictures with one
    and several reference pictures and B pictures == the oracle pipeline with the pinned rdoQuant restatement at its seam (mode: inter CUs only), level planes and CU records
    included; the levels differ from the default seam's; -intertu's four transform units and per-CTU QPs ride along"""
    import ctypes as C
    from ks265codec_amd import stream as S
    from ks265codec_amd.lib import CU8, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline, lib as olib
    clip = make_clip(W, H, 6, seed=W + 1, abc=(37, 53, 19), pan=(5, 3))
    tools = dict(ENCODER_TOOLS, tu_inter=1)
    w = S.StreamWriter(W, H)
    lam = np.array([int(256 * (0.85 * 2.0 ** ((q - 12) / 3.0)) + 0.5) for q in range(52)], np.int64)
    o = OraclePipeline(W, H, 28, lambda_q4(28), me_method=2, me_hex_thr=16, **tools)
    qmap = (28 + (np.arange(o.nctu) % 5) - 2).astype(np.int8)
    levels = {}
    for rdoq in (1, 0):
        o = OraclePipeline(W, H, 28, lambda_q4(28), me_method=2, me_hex_thr=16, **tools)
        with KsFrame(ks, W, H, 28, lambda_q4(28), me_method=2, me_hex_thr=16, bframes=3, refs=2, **tools) as f:
            src = f.new_pic()
            dg, do = {}, {}
            # coding order: I0 P4(0) B2(0,4) P5(4,0: two pictures in list 0) P3 with a QP per CTU
            for step, (d, kind, refs) in enumerate([(0, "I", []), (4, "P", [0]), (2, "B", [0, 4]), (5, "P", [4, 0]), (3, "P", [5])]):
                q = 28 if kind == "I" else 29 + (kind == "B")
                lq = lambda_q4(q, inter=kind != "I")
                o.set_qp(q, lq); f.set_qp(q, lq)
                use_map = step == 4
                o.set_qp_map(qmap if use_map else None); f.set_qp_map(ks.dev(qmap) if use_map else None)
                tab = w.rdoq_tables(None, S.SLICE_P if kind == "P" else S.SLICE_B, q)
                if rdoq and kind != "I":
                    olib().kso_experiment_rdoq(tab.ctypes.data_as(C.c_void_p), 1 | 8, None); f.set_rdoq(tab, lam, lam)
                else:
                    olib().kso_experiment_rdoq(None, 0, None); f.set_rdoq(None)
                f.load_i420(ks.dev(clip[d]), src)
                out = f.new_pic()
                try:
                    if kind == "I":
                        eo = o.encode(clip[d], "I"); f.encode_picture(src, out, True, out)
                    elif kind == "B":
                        eo = o.encode(clip[d], "B", do[refs[0]], do[refs[1]]); f.encode_picture_b(src, dg[refs[0]], dg[refs[1]], out)
                    elif len(refs) > 1:
                        eo = o.encode_mref(clip[d], [do[r] for r in refs]); f.encode_picture_mref(src, [dg[r] for r in refs], out)
                    else:
                        eo = o.encode(clip[d], "P", do[refs[0]]); f.encode_picture(src, dg[refs[0]], False, out)
                finally:
                    olib().kso_experiment_rdoq(None, 0, None)
                got, exp = ks.host(f.store_i420(out), np.uint8), o.store(eo)
                ly = f.ws_read("levels", W * H * 2, 0).view(np.int16)
                assert (ly == o.lvl[0]).all(), f"rdoq {rdoq} picture {d} ({kind}): {int((ly != o.lvl[0]).sum())} luma levels differ"
                cu = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
                assert (cu.view(np.uint8) == o.cu8.view(np.uint8)).all(), f"rdoq {rdoq} picture {d} ({kind}): CU records differ"
                assert (got == exp).all(), f"rdoq {rdoq} picture {d} ({kind}): {int((got != exp).sum())} recon bytes differ"
                dg[d], do[d] = out, eo
                levels[(rdoq, d)] = ly.copy()
            o.set_qp_map(None)
    assert any((levels[(1, d)] != levels[(0, d)]).any() for d in (4, 2, 5, 3)), "rdoQuant left every level where the default seam leaves it"


def test_config3_2160p_encoder_tools(ks):
    """3840x2160 -preset slow with EXACTLY the tool set bench.py and the C host run (ENCODER_TOOLS): key picture + two P pictures == oracle"""
    _ippp(ks, 3840, 2160, 27, 2, 3, seed=7, abc=(67, 91, 33), pan=(8, 5), hex_thr=16, **ENCODER_TOOLS)


def test_config2_1080p_encoder_tools(ks):
    """1920x1080 -preset slow with the encoder's tool set"""
    _ippp(ks, 1920, 1080, 27, 2, 3, seed=42, hex_thr=16, **ENCODER_TOOLS)


def test_config1_720p_encoder_tools(ks):
    """1280x720 veryfast (HEX, qp 32) with the encoder's tool set"""
    _ippp(ks, 1280, 720, 32, 1, 3, seed=43, **ENCODER_TOOLS)


def test_config4_bframes3_umh_720p(ks):
    """-bframes 3 with UMH: coding order I0 P4 B1 B2 B3 P8 B5 B6 B7 through ks265_encode_picture_b"""
    from ks265codec_amd.gop import coding_order
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline

    W, H = 1280, 720
    clip = make_clip(W, H, 9, seed=44)
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, **ENCODER_TOOLS)        # the encoder's tool set (what ks265enc -bframes 3 runs)
    with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=2, bframes=3, me_hex_thr=16, **ENCODER_TOOLS) as f:
        src = f.new_pic()
        dg, do = {}, {}
        prev_anchor = {}
        last = None
        for d, kind in itertools.islice(coding_order(3, 128), 9):
            if kind == "B":
                r0, r1 = prev_anchor[last], last
            else:
                r0, r1 = last, None
                prev_anchor[d], last = last, d
            q = {"I": 27, "P": 28, "B": 30}[kind]
            lam = lambda_q4(q, inter=kind != "I")
            o.set_qp(q, lam); f.set_qp(q, lam)
            do[d] = o.encode(clip[d], kind, do.get(r0), do.get(r1))
            f.load_i420(ks.dev(clip[d]), src)
            out = f.new_pic()
            if kind == "B":
                f.encode_picture_b(src, dg[r0], dg[r1], out)
            else:
                f.encode_picture(src, dg[r0] if r0 is not None else out, kind == "I", out)
            dg[d] = out
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(do[d])
            assert (got == exp).all(), f"picture {d} ({kind}): {int((got != exp).sum())} bytes differ"


def test_hier_b8_1080p_encoder_tools(ks):
    """the SDK's default GOP (hierarchical B, 8) at 1920x1080 with the encoder's tool set: anchors 8 pictures apart (intra CUs where the motion uncovers),
    three layers of B pictures with the joint refinement, coefficient-group pruning everywhere - every picture == oracle"""
    from ks265codec_amd.gop import hier_order
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline

    W, H, G = 1920, 1080, 8
    clip = make_clip(W, H, G + 1, seed=45)
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, **ENCODER_TOOLS)
    nintra = 0
    with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=2, bframes=3, me_hex_thr=16, **ENCODER_TOOLS) as f:
        src = f.new_pic()
        dg, do = [f.new_pic() for _ in range(G + 1)], {}
        for d, kind, r0, r1, layer in itertools.islice(hier_order(G, 128), G + 1):
            q = 27 if kind == "I" else 28 + layer
            lam = lambda_q4(q, inter=kind != "I")
            o.set_qp(q, lam); f.set_qp(q, lam)
            do[d] = o.encode(clip[d], kind, do.get(r0), do.get(r1))
            f.load_i420(ks.dev(clip[d]), src)
            out = dg[d % (G + 1)]
            if kind == "B":
                f.encode_picture_b(src, dg[r0 % (G + 1)], dg[r1 % (G + 1)], out)
            else:
                f.encode_picture(src, dg[r0 % (G + 1)] if r0 is not None else out, kind == "I", out)
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(do[d])
            assert (got == exp).all(), f"picture {d} ({kind}, layer {layer}): {int((got != exp).sum())} bytes differ"
            if kind != "I":
                nintra += int((o.cu8["pred_mode"] == 2).sum())
    assert nintra > 0, "the clip should make some CU of a P / B picture intra"


@pytest.mark.parametrize("W,H,G,seed", [(1920, 1080, 4, 46), (3840, 2160, 2, 7)])
def test_config5_hierarchical_b_with_partitions(ks, W, H, G, seed):
    """round 5: config 5's tool set IN ITS B PICTURES - -preset veryslow codes a hierarchical-B GOP with -part 1: UMH always, -subme 2 judged by Hadamard,
    2NxN / Nx2N partitions whose halves take direction and vectors of their own (ks265_cu_decide_part_b) - every picture of a mini-GOP == oracle, and B pictures do hold
    partitioned CUs of every direction"""
    from ks265codec_amd.gop import hier_order
    from ks265codec_amd.lib import CU8, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip, subme_knobs
    from oracle_lib import OraclePipeline
    tools = dict(ENCODER_TOOLS, me_hex_thr=0, part=1, tu_inter=1, **subme_knobs("veryslow"))
    clip = make_clip(W, H, G + 1, seed=seed, abc=(37, 53, 19) if W < 3000 else (67, 91, 33), pan=(5, 3) if W < 3000 else (8, 5))
    o = OraclePipeline(W, H, 27, lambda_q4(27), **tools)
    dirs = np.zeros(4, np.int64)
    with KsFrame(ks, W, H, 27, lambda_q4(27), bframes=3, **tools) as f:
        src = f.new_pic()
        dg, do = [f.new_pic() for _ in range(G + 1)], {}
        for d, kind, r0, r1, layer in itertools.islice(hier_order(G, 128), G + 1):
            q = 27 if kind == "I" else 28 + layer
            lam = lambda_q4(q, inter=kind != "I")
            o.set_qp(q, lam); f.set_qp(q, lam)
            do[d] = o.encode(clip[d], kind, do.get(r0), do.get(r1))
            f.load_i420(ks.dev(clip[d]), src)
            out = dg[d % (G + 1)]
            if kind == "B":
                f.encode_picture_b(src, dg[r0 % (G + 1)], dg[r1 % (G + 1)], out)
            else:
                f.encode_picture(src, dg[r0 % (G + 1)] if r0 is not None else out, kind == "I", out)
            cu = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
            assert (cu == o.cu8.view(CU8).ravel()).all(), f"picture {d} ({kind}): CU records differ"
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(do[d])
            assert (got == exp).all(), f"picture {d} ({kind}, layer {layer}): {int((got != exp).sum())} bytes differ"
            if kind == "B":
                part = o.cu8["log2_cu"] >> 4
                dirs += np.bincount(o.cu8["inter_dir"][part > 0] & 3, minlength=4)
    assert dirs[1] > 0 and dirs[2] > 0 and dirs[3] > 0, f"partitioned CUs of B pictures by direction: {dirs.tolist()}"


def test_config5_b_pictures_with_several_references_per_list(ks):
    """round 5: -ref N in B pictures - config 5's tool set (UMH always, -subme 2 by Hadamard, -part 1) at 1920x1080 on a pyramid whose B pictures search two
    pictures per list (ks265_encode_picture_b_mref: one search per picture, ks265_ref_pick per list, every later stage takes a block's pictures from its record): CU records and
    reconstruction of every picture == oracle, and both lists' second pictures are used"""
    from ks265codec_amd.lib import CU8, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip, subme_knobs
    from oracle_lib import OraclePipeline
    from stream_cases import schedule
    W, H = 1920, 1080
    tools = dict(ENCODER_TOOLS, me_hex_thr=0, part=1, tu_inter=1, **subme_knobs("veryslow"))
    sched = schedule("hiermr", 4)[:8]                                          # I0 P4 B2 B1 B3 P8 B6 B5 (B5: two pictures in BOTH lists - both index rates in a pair's cost)
    clip = make_clip(W, H, 9, seed=47, abc=(37, 53, 19), pan=(5, 3))
    o = OraclePipeline(W, H, 27, lambda_q4(27), **tools)
    used = np.zeros(2, np.int64)
    with KsFrame(ks, W, H, 27, lambda_q4(27), bframes=3, refs=2, **tools) as f:
        src = f.new_pic()
        dg, do = {}, {}
        for d, kind, l0, l1, dq, rps, isref in sched:
            q = 27 + dq
            lam = lambda_q4(q, inter=kind != "I")
            o.set_qp(q, lam); f.set_qp(q, lam)
            f.load_i420(ks.dev(clip[d]), src)
            out = f.new_pic()
            if kind == "B" and (len(l0) > 1 or len(l1) > 1):
                do[d] = o.encode_b_mref(clip[d], [do[r] for r in l0], [do[r] for r in l1])
                f.encode_picture_b_mref(src, [dg[r] for r in l0], [dg[r] for r in l1], out)
            elif kind == "B":
                do[d] = o.encode(clip[d], "B", do[l0[0]], do[l1[0]])
                f.encode_picture_b(src, dg[l0[0]], dg[l1[0]], out)
            else:
                do[d] = o.encode(clip[d], kind, do.get(l0[0]) if l0 else None, None)
                f.encode_picture(src, dg[l0[0]] if l0 else out, kind == "I", out)
            dg[d] = out
            cu = f.ws_read("cu8", f.geom.bytes_cu8).view(CU8)
            assert (cu == o.cu8.view(CU8).ravel()).all(), f"picture {d} ({kind}, lists {l0} {l1}): CU records differ"
            got, exp = ks.host(f.store_i420(out), np.uint8), o.store(do[d])
            assert (got == exp).all(), f"picture {d} ({kind}, lists {l0} {l1}): {int((got != exp).sum())} bytes differ"
            if kind == "B":
                c, inter = o.cu8, o.cu8["pred_mode"] == 0
                used[0] += int((inter & ((c["inter_dir"] & 1) > 0) & (((c["inter_dir"] >> 4) & 3) == 1)).sum())
                used[1] += int((inter & ((c["inter_dir"] & 2) > 0) & (((c["inter_dir"] >> 6) & 3) == 1)).sum())
    assert used[0] > 0 and used[1] > 0, f"blocks predicting from the second picture of list 0 / list 1: {used.tolist()}"


def test_full_size_properties_2160p_umh(ks):
    """3840x2160 with the bench's search method: run-to-run determinism and PSNR sanity over a 4-picture GOP head"""
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip, psnr

    W, H = 3840, 2160
    clip = make_clip(W, H, 4, seed=7, abc=(67, 91, 33), pan=(8, 5))
    outs = []
    for rep in range(2):
        with KsFrame(ks, W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16) as f:
            src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
            recs = []
            for t in range(4):
                q = 27 + (t > 0)
                f.set_qp(q, lambda_q4(q))
                f.load_i420(ks.dev(clip[t]), src)
                f.encode_picture(src, a, t == 0, b)
                recs.append(ks.host(f.store_i420(b), np.uint8))
                a, b = b, a
            outs.append(recs)
    for t in range(4):
        assert (outs[0][t] == outs[1][t]).all(), f"picture {t} not deterministic"
        assert psnr(clip[t][:W * H], outs[0][t][:W * H]) > 31.0


def test_fuzz_bounded(ks, tmp_path):
    """40 random configurations (sizes 8..472 x 8..312, QP 0..51, DIA/HEX/UMH, range, sub-pel / deblock / SAO switches, IPPP / multi-reference /
    hierarchical B): every reconstructed picture equals the oracle's.  The per-case log goes to $KS265_FUZZ_LOG (else pytest's tmp_path)."""
    from ks265codec_amd.gop import hier_order
    from ks265codec_amd.lib import KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline

    log = open(os.environ.get("KS265_FUZZ_LOG") or os.path.join(str(tmp_path), "fuzz_bounded.log"), "w")
    rng = np.random.default_rng(20260927)
    fails = []
    for it in range(40):
        W, H = int(rng.integers(1, 60)) * 8, int(rng.integers(1, 40)) * 8
        qp, me = int(rng.integers(0, 52)), int(rng.integers(0, 3))
        kw = dict(me_range=int(rng.choice([8, 16, 32, 64])), subme=int(rng.integers(0, 2)), deblock=int(rng.integers(0, 2)), sao=int(rng.integers(0, 2)), me_method=me, me_hex_thr=int(rng.choice([0, 0, 16, 40])), sdh=int(rng.integers(0, 2)), pre_search=int(rng.integers(0, 2)), merge=int(rng.integers(0, 2)), bi_refine=int(rng.integers(0, 2)), decimate=int(rng.integers(0, 4)), rdo=int(rng.choice([0, 0, 2, 4, 9])), intra_inter=int(rng.integers(0, 3)), propagate=int(rng.integers(0, 3)))
        mode = str(rng.choice(["ippp", "mref", "hier"]))
        clip = make_clip(W, H, 9, seed=int(rng.integers(0, 10000)), noisy=bool(rng.integers(0, 2)), pan=(int(rng.integers(0, 100)), int(rng.integers(0, 60))) if it % 3 == 0 else (5, 3))
        o = OraclePipeline(W, H, qp, lambda_q4(qp), **kw)
        tag = f"{it} {W}x{H} qp{qp} {kw} {mode}"
        try:
            with KsFrame(ks, W, H, qp, lambda_q4(qp), bframes=3, refs=3, **kw) as f:
                src = f.new_pic()
                if mode == "hier":
                    G = 4
                    dg, do = [f.new_pic() for _ in range(G + 1)], {}
                    for d, kind, r0, r1, layer in itertools.islice(hier_order(G, 128), 2 * G + 1):
                        q = min(51, qp if kind == "I" else qp + 1 + layer)
                        o.set_qp(q, lambda_q4(q)); f.set_qp(q, lambda_q4(q))
                        do[d] = o.encode(clip[d], kind, do.get(r0), do.get(r1))
                        f.load_i420(ks.dev(clip[d]), src)
                        out = dg[d % (G + 1)]
                        if kind == "B":
                            f.encode_picture_b(src, dg[r0 % (G + 1)], dg[r1 % (G + 1)], out)
                        else:
                            f.encode_picture(src, dg[r0 % (G + 1)] if r0 is not None else out, kind == "I", out)
                        got, exp = ks.host(f.store_i420(out), np.uint8), o.store(do[d])
                        assert (got == exp).all(), (d, kind, int((got != exp).sum()))
                else:
                    dpo, dpg = [], []
                    for t in range(6):
                        f.load_i420(ks.dev(clip[t]), src)
                        out = f.new_pic()
                        if t == 0:
                            eo = o.encode(clip[0], "I"); f.encode_picture(src, out, True, out)
                        elif mode == "mref":
                            eo = o.encode_mref(clip[t], dpo[:3]); f.encode_picture_mref(src, dpg[:3], out)
                        else:
                            eo = o.encode(clip[t], "P", dpo[0]); f.encode_picture(src, dpg[0], False, out)
                        got, exp = ks.host(f.store_i420(out), np.uint8), o.store(eo)
                        assert (got == exp).all(), (t, int((got != exp).sum()))
                        dpo.insert(0, eo); dpg.insert(0, out)
            log.write("ok   " + tag + "\n")
        except AssertionError as e:
            fails.append(tag)
            log.write(f"FAIL {tag} {e}\n")
        log.flush()
    log.write(f"failures {len(fails)} of 40\n")
    log.close()
    assert not fails, fails


def test_wavefront_timeout_is_reported(ks):
    """the intra wavefront's bounded wait must not time out silently.  With the poll budget forced to 0 every CTU row
    below the first gives up at once -> the device error word is set, ks265_synchronize returns KS265_FAIL and names the condition; after
    restoring the budget the same picture encodes correctly again (== oracle)."""
    from ks265codec_amd.lib import KsFrame, Ks265Error
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline

    W, H = 416, 240
    clip = make_clip(W, H, 1, seed=3)
    exp = OraclePipeline(W, H, 30, lambda_q4(30)).encode_picture(clip[0], True)
    with KsFrame(ks, W, H, 30, lambda_q4(30)) as f:
        src, out = f.new_pic(), f.new_pic()
        f.load_i420(ks.dev(clip[0]), src)
        ks.sync()
        ks.debug_set(1, 0)
        try:
            f.encode_picture(src, out, True, out)
            with pytest.raises(Ks265Error, match="wavefront"):
                ks.sync()
        finally:
            ks.debug_set(1, -1)
        ks.sync()                                    # the error word was cleared by the failing call
        f.encode_picture(src, out, True, out)
        ks.sync()
        assert (ks.host(f.store_i420(out), np.uint8) == exp).all()
