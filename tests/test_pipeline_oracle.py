"""CPU: properties of the frame-stage oracle (oracle/ks265_pipeline_oracle.c) that do not need a GPU:
geometry identical to the product's, planes == block-wise normative interpolation, motion found on a known pan,
decoder-side consistency of the reconstruct stage (recon == pred + inverse(dequant(levels))), idempotent padding."""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle_lib import I, OraclePipeline, lib, ptr
from ks265codec_amd.synth import lambda_q4, make_clip, psnr


def _run(W, H, n, seed=3, **kw):
    clip = make_clip(W, H, n, seed=seed, abc=(17, 23, 9))
    o = OraclePipeline(W, H, 27, lambda_q4(27), **kw)
    recs = []
    for t in range(n):
        qp = 27 if t == 0 else 28
        o.set_qp(qp, lambda_q4(qp))
        recs.append(o.encode_picture(clip[t], t == 0))
    return clip, o, recs


def test_geometry_matches_product_library():
    from ks265codec_amd import build as kb
    kb.build()
    from ks265codec_amd.lib import FrameCfg, FrameGeom, load_library
    l = load_library()
    for (w, h) in [(416, 240), (1280, 720), (1920, 1080), (3840, 2160), (200, 136)]:
        o = OraclePipeline(w, h, 27, 80)
        g = FrameGeom()
        assert l.ks265_frame_geometry(C.byref(FrameCfg(w, h, 27, 80, 64, 0, 1, 1, 1, 0, 0, 0)), C.byref(g)) == 0
        for name, _ in FrameGeom._fields_:
            assert getattr(g, name) == getattr(o.geom, name), name


def test_pan_is_found_and_quality_is_sane():
    clip, o, recs = _run(256, 192, 3)
    W, H = 256, 192
    cu = o.cu8.reshape(H // 8, W // 8)
    assert np.median(cu["mvx"]) == 20 and np.median(cu["mvy"]) == 12      # global pan (5, 3) pixels in quarter-pel units
    for t in range(3):
        assert psnr(clip[t][:W * H], recs[t][:W * H]) > 32.0
    assert set(np.unique(cu["log2_cu"])) <= {3, 4, 5, 6}


def test_planes_equal_blockwise_interpolation():
    """plane[fy*4+fx] must equal the pinned block interpolators applied at an arbitrary block"""
    clip, o, _ = _run(128, 72, 2)
    g, ol = o.geom, lib()
    P = o.planes.reshape(16, -1, g.stride_y)[:, :g.rows_y]
    # o.planes were built from the reference of picture 1 = reconstruction of picture 0: rebuild that picture
    o2 = OraclePipeline(128, 72, 27, lambda_q4(27))
    o2.encode_picture(clip[0], True)
    R = o2.ref.y.reshape(-1, g.stride_y)[:g.rows_y]
    x0, y0, w, h = 37, 11, 24, 16
    for fy in range(4):
        for fx in range(4):
            if not fx and not fy:
                continue
            dst = np.zeros((h, w), np.uint8)
            base = (g.pad_y + y0) * g.stride_y + g.pad_y + x0
            if not fy:
                ol.ks265o_interp_luma_hor_8to8(ptr(dst), I(w), ptr(R, base), I(g.stride_y), I(w), I(h), I(fx))
            elif not fx:
                ol.ks265o_interp_luma_ver_8to8(ptr(dst), I(w), ptr(R, base), I(g.stride_y), I(w), I(h), I(fy))
            else:
                tmp = np.zeros((h + 7, w), np.int16)
                ol.ks265o_interp_luma_hor_8to16(ptr(tmp), I(w), ptr(R, base - 3 * g.stride_y), I(g.stride_y), I(w), I(h + 7), I(fx))
                ol.ks265o_interp_luma_ver_16to8(ptr(dst), I(w), ptr(tmp, 3 * w * 2), I(w), I(w), I(h), I(fy))
            got = P[fy * 4 + fx][g.pad_y + y0:g.pad_y + y0 + h, g.pad_y + x0:g.pad_y + x0 + w]
            assert (got == dst).all(), (fx, fy)


def test_padding_is_idempotent_and_replicates_edges():
    _, o, _ = _run(64, 40, 1)
    g = o.geom
    before = o.ref.y.copy()
    o.o.kso_pad_picture(C.byref(o.cfg), o.ref.c())
    assert (before == o.ref.y).all()
    Y = o.ref.y.reshape(-1, g.stride_y)[:g.rows_y]
    assert (Y[:g.pad_y, g.pad_y:g.pad_y + 64] == Y[g.pad_y, g.pad_y:g.pad_y + 64]).all()
    assert (Y[g.pad_y:g.pad_y + 40, :g.pad_y] == Y[g.pad_y:g.pad_y + 40, g.pad_y:g.pad_y + 1]).all()


def test_levels_reproduce_the_reconstruction():
    """decoder-side property: pred + IDCT(dequant(levels)) == the pre-deblock reconstruction, for every luma TU"""
    clip, o, _ = _run(128, 72, 2)
    W, H, g, ol = 128, 72, o.geom, lib()
    cu = o.cu8.reshape(H // 8, W // 8)
    P = o.planes.reshape(16, -1, g.stride_y)[:, :g.rows_y]
    rec = o.rec_pre[0].reshape(-1, g.stride_y)[:g.rows_y]
    lv = o.lvl[0].reshape(H, W)
    qp = o.cfg.qp
    inv = [40, 45, 51, 57, 64, 72]
    checked = 0
    for by in range(H // 8):
        for bx in range(W // 8):
            c = cu[by, bx]
            t8 = min(1 << (int(c["log2_cu"]) - 3), 4)
            if bx % t8 or by % t8:
                continue
            n, x0, y0 = t8 * 8, bx * 8, by * 8
            log2n = {8: 3, 16: 4, 32: 5}[n]
            mvx, mvy = int(c["mvx"]), int(c["mvy"])
            pl = P[(mvy & 3) * 4 + (mvx & 3)]
            pred = np.ascontiguousarray(pl[g.pad_y + y0 + (mvy >> 2):g.pad_y + y0 + (mvy >> 2) + n, g.pad_y + x0 + (mvx >> 2):g.pad_y + x0 + (mvx >> 2) + n])
            lvl = np.ascontiguousarray(lv[y0:y0 + n, x0:x0 + n])
            out = np.zeros((n, n), np.uint8)
            if (lvl != 0).any():
                dq = np.zeros((n, n), np.int16)
                shift = log2n - 1
                ol.ks265o_dequant(ptr(lvl), ptr(dq), I(n), I(inv[qp % 6] << (qp // 6)), I(1 << (shift - 1)), I(shift), I(n - 1), I(n - 1))
                tmp = np.zeros((n, n), np.int16)
                ol.ks265o_inv_transform(I(log2n - 1), ptr(dq), ptr(out), ptr(pred), I(n), I(n), I(n), ptr(tmp), I(n - 1), I(n - 1))
            else:
                out = pred
            assert (out == rec[g.pad_y + y0:g.pad_y + y0 + n, g.pad_y + x0:g.pad_y + x0 + n]).all(), (bx, by)
            checked += 1
    assert checked > 10


def test_ragged_picture_sizes_code_every_block():
    """widths/heights that are multiples of 8 but not of 64: every 8x8 block gets a CU, invalid PUs are marked"""
    for (w, h) in [(72, 40), (200, 136), (8, 8)]:
        _, o, recs = _run(w, h, 2)
        cu = o.cu8.reshape(h // 8, w // 8)
        assert (cu["log2_cu"] >= 3).all() and (cu["log2_cu"] <= 6).all()
        pu = o.prev_pu.reshape(-1, 85)
        assert (pu["cost"][:, 0] == 0xFFFFFFFF).any() or (w % 64 == 0 and h % 64 == 0)


def test_intra_picture_properties():
    """SURVEY.md §8(f) rank 1 on the CPU side: the intra picture of the oracle is a legal HEVC intra decision (modes 0..34, CU 8..32, no
    64x64), beats the flat stand-in on rate at the same quantiser, and a smooth ramp is coded almost for free by planar / angular modes"""
    W, H = 200, 136
    clip = make_clip(W, H, 1, seed=5)
    res = {}
    for intra in (False, True):
        o = OraclePipeline(W, H, 27, lambda_q4(27), intra=intra)
        rec = o.encode_picture(clip[0], True)
        res[intra] = (psnr(clip[0][:W * H], rec[:W * H]), sum(int(np.abs(l.astype(np.int64)).sum()) for l in o.lvl), o.cu8.copy())
    cu = res[True][2]
    assert (cu["pred_mode"] == 2).all() and cu["mvx"].min() >= 0 and cu["mvx"].max() <= 34
    assert set(np.unique(cu["log2_cu"])) <= {3, 4, 5}
    assert res[True][1] < 0.7 * res[False][1] and res[True][0] > res[False][0] - 0.5          # much less residual energy at about the same PSNR
    # a pure diagonal ramp: every block is predicted (almost) exactly from its neighbours -> hardly any coefficient survives
    yy, xx = np.mgrid[0:H, 0:W]
    ramp = np.clip(40 + xx + yy // 2, 0, 255).astype(np.uint8)
    pic = np.concatenate([ramp.reshape(-1), np.full(W * H // 2, 128, np.uint8)])
    o = OraclePipeline(W, H, 27, lambda_q4(27), intra=True)
    rec = o.encode_picture(pic, True)
    nzl = int((o.lvl[0] != 0).sum())
    assert psnr(pic[:W * H], rec[:W * H]) > 40.0 and nzl < W * H // 40, nzl


def test_bi_refinement_and_decimation_properties():
    """the two late tools of round 2 at the oracle level (the GPU tests compare the kernels with exactly these functions):
      * cfg.bi_refine never raises a PU's cost, only touches bi-predictive PUs' vectors, and keeps them inside the planes' margin;
      * cfg.decimate only removes levels: where a luma TU is dropped its levels are all zero and the CU's luma cbf is clear, every other level is untouched when the
        motion is the same (first P picture after the key picture), chroma levels are never touched."""
    W, H = 200, 136
    clip = make_clip(W, H, 5, seed=6, abc=(17, 23, 9))
    pubs = {}
    for r in (0, 1, 2):
        o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=1, bi_refine=r)
        d = {0: o.encode(clip[0], "I")}
        o.set_qp(28, lambda_q4(28)); d[4] = o.encode(clip[4], "P", d[0])
        o.set_qp(30, lambda_q4(30)); o.encode(clip[2], "B", d[0], d[4])
        pubs[r] = (o.pub.copy(), o.pu.copy(), o.pu1.copy())
    a, b = pubs[0][0], pubs[1][0]
    valid = a["cost"] != 0xFFFFFFFF
    assert (b["cost"][valid] <= a["cost"][valid]).all() and (b["cost"][valid] < a["cost"][valid]).any()
    moved = valid & ((a["mvx"] != b["mvx"]) | (a["mvy"] != b["mvy"]) | (a["mv1x"] != b["mv1x"]) | (a["mv1y"] != b["mv1y"]))
    assert moved.any() and (b["inter_dir"][moved] == 3).all()
    for k in ("mvx", "mv1x"):
        assert (np.abs(b[k][valid].astype(int)) <= 4 * (W + 80)).all()
    # cfg.bi_refine = 2 (round 5): the same refinement after the CU decision - only records of PUs that are CUs of the final tree change, each into what mode 1 makes of it
    # (same inputs: the lists' records and the unrefined pair), and the CU records carry the refined motion
    c = pubs[2][0]
    late = valid & (c["cost"] != a["cost"])
    assert late.any() and (c[late] == b[late]).all() and (c[~late] == a[~late]).all() and late.sum() < (b["cost"] != a["cost"]).sum()
    # decimation: same key picture, same motion search on the first P picture -> only levels of dropped luma TUs differ
    lv = {}
    for k in (0, 2):
        o = OraclePipeline(W, H, 30, lambda_q4(30), me_method=1, decimate=k)
        d0 = o.encode(clip[0], "I")
        o.set_qp(31, lambda_q4(31)); o.encode(clip[1], "P", d0)
        lv[k] = ([x.copy() for x in o.lvl], o.cu8.copy())
    y0, y2 = lv[0][0][0].reshape(H, W), lv[2][0][0].reshape(H, W)
    changed = y0 != y2
    assert changed.any() and (y2[changed] == 0).all() and (np.abs(y0[changed]) == 1).all()
    assert (lv[0][0][1] == lv[2][0][1]).all() and (lv[0][0][2] == lv[2][0][2]).all()
    cu0, cu2 = lv[0][1].reshape(H // 8, W // 8), lv[2][1].reshape(H // 8, W // 8)
    blk = changed.reshape(H // 8, 8, W // 8, 8).any(axis=(1, 3))
    assert ((cu2["cbf"][blk] & 1) == 0).all() and ((cu0["cbf"][blk] & 1) == 1).all()
    diff = cu0["cbf"] != cu2["cbf"]                                  # a dropped 16x16 / 32x32 TU clears the flag of all its 8x8 blocks
    assert blk[diff | blk].any() and ((cu2["cbf"][diff] & 1) == 0).all() and ((cu0["cbf"][diff] & 1) == 1).all()
    assert ((cu0["cbf"] & 6) == (cu2["cbf"] & 6)).all()


def test_vector_propagation_properties():
    """stage A2 (cfg.propagate): no PU's cost rises, every adopted vector is the integer vector of a same-size neighbour before the round and lies inside the CTU's
    limits, the rate term is the PU's own (predictor unchanged), invalid PUs pass through; a small object in front of a panning background is what it is for: the
    coded picture gets smaller at the same QP without losing quality"""
    from oracle_lib import PU
    W, H = 200, 136                                        # ragged: partial CTUs on both sides
    clip = make_clip(W, H, 2, seed=5, abc=(17, 23, 9))
    o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, pre_search=1, propagate=1)
    ref = o.encode(clip[0], "I")
    o.load(o.src, clip[1])
    cfg, g = C.byref(o.cfg), o.geom
    pu, off = np.zeros(o.nctu * 85, PU), np.zeros(2 * o.nctu, np.int16)
    o.o.kso_me_integer_ex(cfg, o.src.c(), ref.c(), None, ptr(pu), ptr(off))
    plain = np.zeros_like(pu)
    o.o.kso_me_integer(cfg, o.src.c(), ref.c(), None, ptr(plain))
    assert (plain == pu).all()                             # the _ex form is the same search
    out = np.zeros_like(pu)
    o.o.kso_me_propagate(cfg, o.src.c(), ref.c(), ptr(off), ptr(pu), ptr(out))
    inv = pu["cost"] == 0xFFFFFFFF
    assert inv.any() and (out[inv] == pu[inv]).all()
    assert (out["cost"] <= pu["cost"]).all() and (out["mvpx"] == pu["mvpx"]).all() and (out["mvpy"] == pu["mvpy"]).all()
    moved = np.nonzero((out["mvx"] != pu["mvx"]) | (out["mvy"] != pu["mvy"]))[0]
    assert len(moved) > 0
    base = [0, 1, 5, 21]
    for i in moved:
        ctu, idx = divmod(int(i), 85)
        cx, cy = ctu % g.ctu_cols, ctu // g.ctu_cols
        l = 0 if idx < 1 else 1 if idx < 5 else 2 if idx < 21 else 3
        n, px, py = 1 << l, (idx - base[l]) & ((1 << l) - 1), (idx - base[l]) >> l
        nb = set()
        for dx, dy in ((-1, 0), (0, -1), (1, 0), (0, 1)):
            gx, gy = cx * n + px + dx, cy * n + py + dy
            if 0 <= gx < g.ctu_cols * n and 0 <= gy < g.ctu_rows * n:
                q = pu[((gy >> l) * g.ctu_cols + (gx >> l)) * 85 + base[l] + (gy & (n - 1)) * n + (gx & (n - 1))]
                if q["cost"] != 0xFFFFFFFF:
                    nb.add((int(q["mvx"]) >> 2, int(q["mvy"]) >> 2))
        assert (int(out[i]["mvx"]) >> 2, int(out[i]["mvy"]) >> 2) in nb and out[i]["mvx"] % 4 == 0 and out[i]["mvy"] % 4 == 0
        lim = (C.c_int * 4)()
        o.o.kso_ctu_mv_limits(cfg, cx, cy, int(off[2 * ctu]), int(off[2 * ctu + 1]), lim)
        assert lim[0] <= int(out[i]["mvx"]) >> 2 <= lim[1] and lim[2] <= int(out[i]["mvy"]) >> 2 <= lim[3]
    # effect on a coded picture: the bench-style clip (squares in front of a pan), same QP
    W, H = 416, 240
    clip = make_clip(W, H, 3, seed=7, pan=(8, 5))
    size = {}
    for p in (0, 1):
        o = OraclePipeline(W, H, 27, lambda_q4(27), me_method=2, me_hex_thr=16, pre_search=1, merge=1, rdo=4, intra_inter=1, propagate=p)
        r = o.encode(clip[0], "I")
        nz, ps = 0, []
        for t in (1, 2):
            o.set_qp(28, lambda_q4(28))
            r = o.encode(clip[t], "P", r)
            nz += sum(int((a != 0).sum()) for a in o.lvl)
            ps.append(psnr(clip[t][:W * H], o.store(r)[:W * H]))
        size[p] = (nz, min(ps))
    assert size[1][0] < 0.9 * size[0][0] and size[1][1] > size[0][1] - 0.1, size


def test_tools_per_picture_carry_no_state():
    """round 6 (lean B pictures): OraclePipeline.set_picture_tools - the checker's side of ks265_frame_set_picture_tools - switches tools for the pictures coded from there on and
    nothing else: a B picture coded lean on a pipeline created with every tool == the same picture on a pipeline CREATED with the lean set fed the same reference pictures, a full
    picture after it == one on a pipeline that never left the full set, and a lean picture holds no intra CU, no refined pair and only "off" SAO records"""
    from ks265codec_amd.synth import ENCODER_TOOLS
    W, H = 200, 136
    clip = make_clip(W, H, 5, seed=6, abc=(17, 23, 9), pan=(2, 1))
    full = dict(ENCODER_TOOLS, bi_refine=2)
    lean = dict(full, intra_inter=0, bi_refine=0, sao=0)

    def run(tools, switch):
        o = OraclePipeline(W, H, 27, lambda_q4(27), **tools)
        i0 = o.encode(clip[0], "I")
        o.set_qp(28, lambda_q4(28, inter=True)); p4 = o.encode(clip[4], "P", i0)
        o.set_qp(29, lambda_q4(29, inter=True)); b2 = o.encode(clip[2], "B", i0, p4)
        recs = [o.store(b2).copy()]
        if switch:
            o.set_picture_tools(0, 0, 0)
        o.set_qp(31, lambda_q4(31, inter=True)); b1 = o.encode(clip[1], "B", i0, b2)
        recs.append((o.store(b1).copy(), o.cu8.copy(), [a.copy() for a in o.lvl], o.sao.copy()))
        if switch:
            o.set_picture_tools(-1, -1, -1)
        b3 = o.encode(clip[3], "B", b2, p4)
        recs.append((o.store(b3).copy(), o.cu8.copy(), o.sao.copy()))
        return i0, p4, b2, recs

    i0, p4, b2, sw = run(full, True)
    _, _, _, never = run(full, False)
    # the lean picture on a pipeline created lean, fed the switched pipeline's reference pictures
    ol = OraclePipeline(W, H, 31, lambda_q4(31, inter=True), **lean)
    ol.set_qp(31, lambda_q4(31, inter=True))
    b1 = ol.encode(clip[1], "B", i0, b2)
    assert (ol.store(b1) == sw[1][0]).all() and (ol.cu8.view(np.uint8) == sw[1][1].view(np.uint8)).all() and all((a == b).all() for a, b in zip(ol.lvl, sw[1][2]))
    assert (sw[1][3]["type"] == -1).all() and (sw[1][1]["pred_mode"] == 0).all()
    assert (sw[0] == never[0]).all()
    assert not (sw[1][0] == never[1][0]).all(), "the lean picture differs from the full one (SAO alone changes samples)"
    # the picture after the switch back: the full set again (its references differ from the never-switched run's only through b1, which it does not use)
    assert (sw[2][0] == never[2][0]).all() and (sw[2][1].view(np.uint8) == never[2][1].view(np.uint8)).all() and (sw[2][2].view(np.uint8) == never[2][2].view(np.uint8)).all()
    assert (sw[2][2]["type"] != -1).any()
