#!/usr/bin/env python3
"""Coding efficiency on the CPU: the ORACLE pipeline (the specification the HIP stages equal bit for bit) + the host's stream writer on a bench-style clip,
next to the reference encoder `appencoder` on the same clip - bitrate and PSNR-Y per QP, and the bitrate ratio at the reference's PSNR-Y.
Builder container only (needs oracle/_ref/appencoder or /root/reference); a design aid for the RD tools of DESIGN.md, not part of the product.

usage: tools/rd_eval.py [--size 832x480] [--frames 17] [--gop ippp|hier] [--qps 27,29,31] [--tools k=v,...] [--no-ref] [--tag text]
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ks265codec_amd.synth import ENCODER_TOOLS as HOST_TOOLS                                                # the host encoder's tool set (--host)
ENCODER_TOOLS = dict(me_method=2, me_hex_thr=16, sdh=1, pre_search=1, merge=1, bi_refine=1, decimate=2)   # round 2's set (without --host)


def stats_lib():
    """the stream writer built with -DKS265_BIT_STATS: information content of the slice data per syntax category"""
    import ctypes as C
    from ks265codec_amd import stream as S
    so = os.path.join(tempfile.gettempdir(), "libks265_stats.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-DKS265_BIT_STATS", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so,
                           os.path.join(ROOT, "ks265codec_amd/host/ks265_stream.c"), "-lm"])
    S._lib = C.CDLL(so)
    for n in ("ks265_write_vps", "ks265_write_sps", "ks265_write_pps", "ks265_write_slice"):
        getattr(S._lib, n).restype = C.c_long
    S._lib.ks265_slice_scratch_bytes.restype = C.c_size_t
    return S._lib


STAT_NAMES = ["cu", "merge", "motion", "coef", "sao", "intra"]


def encode_ours(clip, W, H, qp, gop, tools, verbose=False, stats=None, pdelta=1, cascade=None, lam_scale=1.0, rdo_layers=None, b_lam=None, layer_qp=None, seq=None):
    from ks265codec_amd import stream as S
    from ks265codec_amd.gop import hier_order
    from ks265codec_amd.synth import lambda_q4, psnr
    from oracle_lib import OraclePipeline
    n = len(clip)
    o = OraclePipeline(W, H, qp, lambda_q4(qp), **tools)
    G = int(os.environ.get("RD_G", "8").replace("plain", ""))
    w = S.StreamWriter(W, H, max_dec_pic_buffering=10 if gop == "hier" else 2, max_num_reorder=7 if gop == "hier" else 0, sdh=tools.get("sdh", 0), wpp=0 if stats else 1, list_mod=1 if os.environ.get('RD_GPB2') else 0, tu_inter=tools.get("tu_inter", 0))
    import ctypes as C
    st = (C.c_double * 6)()
    agg = {}
    bs = w.headers()
    per = []
    ps = []
    pc = []
    if seq is not None:
        pass
    elif gop == "ippp":
        seq = [(t, "I" if t == 0 else "P", t - 1 if t else None, None, 0) for t in range(n)]
    else:
        seq = [s for s in itertools.islice(hier_order(G, 1 << 20), n) if s[0] < n]
        if os.environ.get('RD_GPB_SAME'):                             # experiment: generalised B at the P positions, both lists = the previous anchor
            seq = [(d, 'B', r0, r0, 0) if k == 'P' else (d, k, r0, r1, l) for (d, k, r0, r1, l) in seq]
        if os.environ.get('RD_GPB2'):                                 # generalised B at the P positions: list 0 = the previous anchor, list 1 = the one before it
            seq = [(d, 'B', r0, r0 - G if r0 >= G else r0, 0) if k == 'P' else (d, k, r0, r1, l) for (d, k, r0, r1, l) in seq]
    encode_ours._cfg0 = {nm: getattr(o.cfg, nm) for nm in ('sao', 'bi_refine', 'propagate', 'intra_inter', 'merge', 'rdo')}
    dpb = {}
    # -ref0 (round 6; the host's default under --host: 3 = what -preset slow resolves to): the anchors of the hierarchy search the last RD_MREF anchors of their GOP, nearest first
    nmref = int(os.environ.get('RD_MREF', '3' if lam_scale == -1 else '1'))
    mrs, hist = [], []
    for (d, kind, r0, r1, layer) in seq:
        if kind == 'I':
            hist = [d]; mrs.append([])
        elif kind == 'P':
            mrs.append(hist[:nmref] if (gop == 'hier' and nmref > 1 and hist and hist[0] == r0) else [])
            hist = [d] + hist
        else:
            mrs.append([])
    for i, (d, kind, r0, r1, layer) in enumerate(seq):
        lq = 1 if layer < 0 else layer_qp[min(len(layer_qp) - 1, layer)] if (layer_qp and kind == 'B') else layer      # B layers are numbered 1 (referenced most) .. 3; -1: a plain B picture outside a pyramid (+ 2)
        q = min(51, qp if kind == "I" else qp + pdelta + lq + (cascade[d % len(cascade)] if cascade and gop == "ippp" else 0))
        ls = lam_scale if lam_scale > 0 else (min(4.0, max(2.0, (q - 12) / 6.0))) ** 0.5      # <= 0: HM's factor for non-key pictures, clip(2, 4, (qp - 12) / 6) on lambda_mode; -1 = --host
        if kind == 'B' and b_lam:
            ls *= b_lam[min(len(b_lam) - 1, layer)]
        o.set_qp(q, lambda_q4(q) if kind == "I" else (lambda_q4(q, inter=True) if lam_scale == -1 else int(round(lambda_q4(q) * ls))))     # lam_scale -1: the encoder host's integer table (kLambdaInterQ4)
        if os.environ.get("RD_SPLIT_BITS_B") and kind == 'B':
            sb = [int(x) for x in os.environ["RD_SPLIT_BITS_B"].split(",")]
            o.o.kso_experiment_split_bits_b(sb[min(len(sb) - 1, max(layer, 0))])
        if rdo_layers:
            o.cfg.rdo = rdo_layers[0] if kind == 'P' else rdo_layers[min(len(rdo_layers) - 1, layer)] if kind == 'B' else tools.get('rdo', 0)
        if os.environ.get('RD_II_LAYERS') and kind == 'B':            # experiment: intra CUs per B layer (1 .. 3), e.g. 1,1,0 = none in the top layer
            iil = [int(x) for x in os.environ['RD_II_LAYERS'].split(',')]
            o.cfg.intra_inter = iil[min(len(iil) - 1, max(layer, 1) - 1)]
        elif os.environ.get('RD_II_LAYERS'):
            o.cfg.intra_inter = tools.get('intra_inter', 0)
        if lam_scale == -1 and not os.environ.get('RD_NO_LEAN_B'):     # --host: the host's lean B pictures (ks265_enc.c submit) - a B picture nothing predicts from runs without intra candidates, joint refinement, SAO
            lean = kind == 'B' and not any(d in (a, b) for (_, _, a, b, _) in seq[i + 1:])
            near = kind == 'B' and not lean and d - r0 <= 2 and r1 - d <= 2 and not os.environ.get('RD_LEAN_NONREF_ONLY')      # a reference B picture with both references at most two pictures away: no intra candidates, no SAO
            o.set_picture_tools(*((0, 0, 0) if lean else (0, -1, 0) if near else (-1, -1, -1)))
        if os.environ.get('RD_LAYER_TOOLS'):                          # experiment: cfg fields per B layer (1 .. 3), e.g. sao=1,1,0;bi_refine=2,2,0 (the other pictures keep the tool set's values)
            for spec in os.environ['RD_LAYER_TOOLS'].split(';'):
                nm, vals = spec.split('='); vals = [int(x) for x in vals.split(',')]
                setattr(o.cfg, nm, vals[min(len(vals) - 1, max(layer, 1) - 1)] if kind == 'B' else tools.get(nm, getattr(encode_ours, '_cfg0', {}).get(nm, 0)))
        mr = mrs[i]
        if getattr(encode_ours, "rdoq_select", None):
            encode_ours.rdoq_select(kind)
        if len(mr) > 1:
            dpb[d] = o.encode_mref(clip[d], [dpb[r] for r in mr])
        else:
            dpb[d] = o.encode(clip[d], kind, dpb.get(r0), dpb.get(r1))
        rec = o.store(dpb[d])
        later = seq[i + 1:]
        needed = {r for j, (dd, kk, a, b, _) in enumerate(later) for r in [a, b] + mrs[i + 1 + j] if r is not None and r in dpb and r != d}
        cur = {r for r in (r0, r1) if r is not None} | set(mr)
        rps = [(p, p in cur) for p in sorted(needed | cur)]
        isref = any(d in (a, b) for (_, _, a, b, _) in later)
        if stats:
            stats.ks265_bit_stats(st, 1)
        if kind == "I":
            b = w.slice(S.NAL_IDR_W_RADL, S.SLICE_I, 0, q, o.cu8, o.lvl, o.sao)
        elif kind == "P":
            b = w.slice(S.NAL_TRAIL_R, S.SLICE_P, d, q, o.cu8, o.lvl, o.sao, rps=rps, l0=mr if len(mr) > 1 else [r0])
        else:
            b = w.slice(S.NAL_TRAIL_R if isref else S.NAL_TRAIL_N, S.SLICE_B, d, q, o.cu8, o.lvl, o.sao if o.cfg.sao else None, rps=rps, l0=[r0], l1=[r1])
        bs += b
        if getattr(encode_ours, "rdoq_adaptive", None):                  # --rdoq-adaptive: the tables of the NEXT picture of this kind come from this slice's final context states
            encode_ours.rdoq_adaptive(kind, w)
        if stats:
            stats.ks265_bit_stats(st, 1)
            cu = o.cu8.reshape(H // 8, W // 8)
            t = agg.setdefault(f"{kind}{layer if kind == 'B' else ''}", dict(n=0, bits=np.zeros(6), cus={}))
            t["n"] += 1; t["bits"] += np.array(st[:]) / 8
            if kind != "I":
                for lg in (3, 4, 5, 6):
                    m = cu["log2_cu"] == lg
                    ncu = int(m.sum()) >> (2 * (lg - 3))
                    coded = int((m & (cu["cbf"] != 0)).sum()) >> (2 * (lg - 3))
                    intra = int((m & (cu["pred_mode"] != 0)).sum()) >> (2 * (lg - 3))
                    c = t["cus"].setdefault(lg, [0, 0, 0]); c[0] += ncu; c[1] += coded; c[2] += intra
        p = psnr(clip[d][:W * H], rec[:W * H])
        per.append((d, kind, layer, len(b), p))
        ps.append((clip[d][:W * H].astype(np.float64) - rec[:W * H]) ** 2)
        pc.append((clip[d][W * H:].astype(np.float64) - rec[W * H:]) ** 2)
        if verbose:
            print(f"   {d:3d} {kind} L{layer} {len(b):7d} B  {p:.2f} dB", flush=True)
        for k in [k for k in dpb if k not in needed and k != d and k not in cur]:
            del dpb[k]
    mse = float(np.mean([x.mean() for x in ps]))
    encode_ours.chroma_psnr = 10 * np.log10(255.0 ** 2 / float(np.mean([x.mean() for x in pc])))
    if stats:
        for k, t in sorted(agg.items()):
            print(f"      {k}: " + "  ".join(f"{nm} {t['bits'][i] / t['n']:.0f}" for i, nm in enumerate(STAT_NAMES)) + "   CUs(n coded intra) " +
                  "  ".join(f"{1 << lg}: {c[0] / t['n']:.0f} {c[1] / t['n']:.0f} {c[2] / t['n']:.0f}" for lg, c in sorted(t["cus"].items())))
    return bs, 10 * np.log10(255.0 ** 2 / mse), per


def mini_gop(lo, hi):
    """coding order of the mini-GOP (lo, hi] as the host codes it: the anchor, then - a power of two apart - the B pictures breadth first (code_hier), else plain
    non-reference B pictures between the two anchors at QP + 2 (the flush of a clip that does not end on the grid); entries of encode_ours' seq"""
    if (hi - lo) & (hi - lo - 1):
        return [(hi, "P", lo, None, 0)] + [(b, "B", lo, hi, -1) for b in range(lo + 1, hi)]
    out, cur, layer = [(hi, "P", lo, None, 0)], [(lo, hi)], 1
    while cur:
        nxt = []
        for a, b in cur:
            if b - a >= 2:
                mid = (a + b) // 2
                out.append((mid, "B", a, b, layer)); nxt += [(a, mid), (mid, b)]
        cur, layer = nxt, layer + 1
    return out


def adaptive_seq(clip, W, H, qp, decide=True):
    """the host's slice-type decision (-lookahead N, ks265_enc.c lane_put): per block of 8 pictures inter(8) x 12 > (inter(4) + inter(4)') x 13 -> 4 + 4; the inter sums
    are those of the frame-cost kernels on half-size pictures (here: their oracle restatement, which the GPU tests hold equal)"""
    import ctypes as C
    from ks265codec_amd.lib import PU
    from ks265codec_amd.synth import lambda_q4
    from oracle_lib import I, HostPic, OraclePipeline, lib as olib, ptr
    ol, n = olib(), len(clip)
    w, h = (W // 2) & ~7, (H // 2) & ~7
    o_full, o_low = OraclePipeline(W, H, qp, lambda_q4(qp)), OraclePipeline(w, h, qp, lambda_q4(qp), me_range=32, me_method=1, subme=0)
    gf, gl = o_full.geom, o_low.geom
    of, olo = gf.pad_y * gf.stride_y + gf.pad_y, gl.pad_y * gl.stride_y + gl.pad_y
    low = {}

    def half(t):
        if t not in low:
            o_full.load(o_full.src, clip[t])
            lo = HostPic(gl)
            ol.ks265o_downsample(ptr(lo.y, olo), ptr(o_full.src.y, of), I(gl.stride_y), I(gf.stride_y), I(w), I(h))
            lo.u[:] = 0; lo.v[:] = 0
            ol.kso_pad_picture(C.byref(o_low.cfg), lo.c())
            low[t] = lo
        return low[t]

    def inter(cur, ref):
        cost, pu, out = np.zeros(o_low.nctu * 85, np.uint32), np.zeros(o_low.nctu * 85, PU), np.zeros(4, np.uint64)
        ol.kso_intra_decide_ex(C.byref(o_low.cfg), half(cur).c(), ptr(o_low.cu8), ptr(cost))
        ol.kso_me_integer(C.byref(o_low.cfg), half(cur).c(), half(ref).c(), None, ptr(pu))
        ol.kso_lookahead_reduce(C.byref(o_low.cfg), ptr(cost), ptr(pu), ptr(out))
        return int(out[1])

    seq, d, four = [(0, "I", None, None, 0)], 0, 0
    if os.environ.get("RD_G", "8") == "4plain":                    # the host's -bframes 3 until the end of round 4: P + three plain (non-reference) B pictures at Q + 2
        while d + 1 < n:
            a = min(d + 4, n - 1)
            seq += [(a, "P", d, None, 0)] + [(t, "B", d, a, -1) for t in range(d + 1, a)]
            d = a
        return seq, 0
    if os.environ.get("RD_G", "8") == "4":                         # -bframes 3: the host's (and the reference's) pyramid of 4; run with --layer-qp 0,1,2 (its ladder + 2 / + 3)
        while d + 4 < n:
            seq += mini_gop(d, d + 4); d += 4
        if d < n - 1:
            seq += mini_gop(d, n - 1)
        return seq, 0
    while d + 8 < n:
        c4a, c4b, c8 = (inter(d + 4, d), inter(d + 8, d + 4), inter(d + 8, d)) if decide else (1, 1, 0)
        if c8 * 12 > (c4a + c4b) * 13:
            seq += mini_gop(d, d + 4) + mini_gop(d + 4, d + 8); four += 1
        else:
            seq += mini_gop(d, d + 8)
        d += 8
        for k in [k for k in low if k < d]:
            del low[k]
    if d < n - 1:                                              # the flush: what is left, as one short mini-GOP (a power of two: pyramid, else the host codes plain B pictures - not mirrored)
        seq += mini_gop(d, n - 1)
    return seq, four


def encode_ref(yuv_path, W, H, qp, gop, n, threads=4):
    enc = os.path.join(ROOT, "oracle", "_ref", "appencoder")
    if not os.path.exists(enc):
        enc = "/root/reference/ubuntu_x64/appencoder"
    d = tempfile.mkdtemp(prefix="rdref")
    exe = os.path.join(d, "appencoder")
    subprocess.check_call(["cp", enc, exe]); os.chmod(exe, 0o755)
    args = [exe, "-i", yuv_path, "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-preset", "slow", "-rc", "0", "-qp", str(qp), "-iper", "128", "-threads", str(threads), "-psnr", "2",
            "-b", os.path.join(d, "o.265")]
    if gop == "ippp":
        args += ["-bframes", "0"]
    elif os.environ.get("RD_G", "8") in ("4", "4plain"):
        args += ["-bframes", "3"]                                    # RD_G=4 --gop hier: the pyramid of 4 = what both encoders make of -bframes 3 (BASELINE config 4's GOP)
    r = subprocess.run(args, capture_output=True, text=True, cwd=d)
    m = re.search(r"bitrate, psnr:\s*([\d.]+)\s+([\d.]+)", r.stdout)
    size = os.path.getsize(os.path.join(d, "o.265"))
    per = [(int(a), b, int(c) // 8, float(e)) for a, b, c, e in re.findall(r"^(\d+)\t([IPB])\t(\d+)\t([\d.]+)\t", r.stdout, re.M)]
    subprocess.call(["rm", "-rf", d])
    return size, float(m.group(2)), per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="832x480")
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--gop", default="ippp")
    ap.add_argument("--qps", default="27,29,31")
    ap.add_argument("--ref-qp", type=int, default=27)
    ap.add_argument("--tools", default="")
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--tag", default="")
    ap.add_argument("-v", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--pdelta", type=int, default=1)
    ap.add_argument("--cascade", default="")
    ap.add_argument("--lam-scale", type=float, default=1.0)
    ap.add_argument("--host", action="store_true", help="exactly what the encoder host does: its tool set (intra_inter=1, rdo=4, propagate=1 on top of ENCODER_TOOLS without the decimation), its P / B lambda table, its QP ladder")
    ap.add_argument("--adaptive", action="store_true", help="--gop hier: the host's slice-type decision of -lookahead N (blocks of 8 pictures as 8 or 4 + 4)")
    ap.add_argument("--pingpong", type=int, default=0, metavar="K", help="the clip of the same-clip tables: K pictures of the generator played forth and back, --frames pictures in all")
    ap.add_argument("--pan", default="", help="pan of the synthetic clip in samples per picture, e.g. 8,5")
    ap.add_argument("--rdoq", type=int, default=0, metavar="MODE", help="experiment: the reference's rdoQuant (oracle/ks265_rdoq_ref.c) at the seam with static bit tables (medians of tests/golden/rdoq.npz); 1 = luma of P / B pictures, +2 chroma, +4 key pictures")
    ap.add_argument("--rdoq-adaptive", action="store_true", help="with --rdoq: the bit tables follow the stream - built (pinned estBitRdoq) from the context states the writer ended the previous picture of the same kind with")
    ap.add_argument("--rdoq-mult", default="", help="the four lambda multipliers (luma sign-hiding, luma, chroma sign-hiding, chroma; reference: 256,256,90,90)")
    ap.add_argument("--split-bits-b", default="", help="experiment: what a split is taken to cost in a B picture's CU decision (1/16 bit), indexed by B layer (entry 0 unused; pipeline: 80 everywhere)")
    ap.add_argument("--merge-bits-b", type=int, default=0, metavar="Q4", help="experiment: what explicit motion is taken to cost (1/16 bit) when a B picture's CU weighs a merge candidate (pipeline: 32)")
    ap.add_argument("--rdo-layers", default="", help="cfg.rdo of P pictures and of the B layers 1, 2, 3 (comma separated)")
    ap.add_argument("--b-lam", default="", help="extra lambda factors, indexed by B layer (entry 0 unused)")
    ap.add_argument("--layer-qp", default="", help="QP offsets on top of the P offset, indexed by B layer (entry 0 unused; default = the layer number)")
    a = ap.parse_args()
    from ks265codec_amd.synth import make_clip
    W, H = (int(x) for x in a.size.split("x"))
    big = W >= 3000
    pan = tuple(int(x) for x in a.pan.split(',')) if a.pan else ((8, 5) if big else (5, 3))
    clip = make_clip(W, H, a.pingpong or a.frames, seed=a.seed, abc=(67, 91, 33) if big else (37, 53, 19), pan=pan)
    if a.pingpong:
        order = list(range(a.pingpong)) + list(range(a.pingpong - 2, 0, -1))
        clip = clip[[order[t % len(order)] for t in range(a.frames)]]
    tools = dict(ENCODER_TOOLS)
    if a.host:
        tools = dict(HOST_TOOLS); a.lam_scale = -1.0
        if a.gop == "ippp" and not a.cascade:
            a.cascade = "0,2,1,2"                                   # ks265_enc.c kIpppCascade
        if a.gop == "hier" and not a.layer_qp:
            a.layer_qp = "0,1,3,3"                                  # ks265_enc.c kHierLayerQp
    for kv in filter(None, a.tools.split(",")):
        k, v = kv.split("=")
        tools[k] = int(v)
    if a.rdoq:
        import ctypes as C
        from oracle_lib import lib as olib
        z = np.load(os.path.join(ROOT, "tests", "golden", "rdoq.npz"))
        T = np.zeros((4, 2, 180), np.int32)
        for lg in range(2, 6):
            for ch in (0, 1):
                m = (z["meta"][:, 0] == lg) & ((z["meta"][:, 2] > 0) == bool(ch)) & (z["run_of"] < 2)       # the two -preset slow runs with default weights
                if m.any():
                    T[lg - 2, ch] = np.median(z["tab"][m], axis=0).astype(np.int32)
        main._rq_T = T                                                 # keep alive
        mult = (C.c_int * 4)(*[int(x) for x in a.rdoq_mult.split(",")]) if a.rdoq_mult else None
        olib().kso_experiment_rdoq(T.ctypes.data_as(C.c_void_p), a.rdoq, mult)
        if a.rdoq_adaptive:
            # ADAPTIVE tables - what a device-side rdoQuant could be fed: after every slice the writer's final context states (ks265_slice_final_contexts) go
            # through the pinned estBitRdoq (oracle) into the eight tables (4 sizes x luma / chroma) the next picture of the same kind is quantised with; the first picture of a
            # kind starts from the static medians.  The entropy values are the reference's (tests/golden/estbits.npz `table` cases = HM's table).
            ez = np.load(os.path.join(ROOT, "tests", "golden", "estbits.npz"))
            ent = np.zeros(128, np.int32)
            for i in range(int(ez["__n__"])):
                if str(ez[f"c{i}__kind"]) == "table":
                    ent[int(ez[f"c{i}__ctx"][0])] = ez[f"c{i}__exp"][0]
            assert ent[0] == 0x7B23 and (ent > 0).all()
            tabs = {k: T.copy() for k in "IPB"}
            main._rq_tabs = tabs

            def to_ref_layout(st, lay):
                cbf_l, cbf_c, csbf, sig, lx, ly, g1, g2, root, _ = lay
                c = np.zeros(256, np.uint8)
                c[0x0d:0x0d + 2] = st[cbf_l:cbf_l + 2]; c[0x12:0x12 + 4] = st[cbf_c:cbf_c + 4]
                c[0x1d:0x1d + 4] = st[csbf:csbf + 4]
                c[0x21:0x21 + 42] = st[sig:sig + 42]
                c[0x4b:0x4b + 18] = st[lx:lx + 18]; c[0x69:0x69 + 18] = st[ly:ly + 18]
                c[0x87:0x87 + 24] = st[g1:g1 + 24]; c[0x9f:0x9f + 6] = st[g2:g2 + 6]
                c[0xaa] = st[root]
                return c

            def adapt(kind, w):
                st, lay = w.final_contexts()
                c = to_ref_layout(st, lay)
                for lg in range(2, 6):
                    for ch in (0, 1):
                        out = tabs[kind][lg - 2, ch]                      # (words the function does not write keep the static values)
                        olib().ks265o_est_bit_rdoq(out.ctypes.data_as(C.c_void_p), lg, int(not ch), c.ctypes.data_as(C.c_void_p), ent.ctypes.data_as(C.c_void_p))
            encode_ours.rdoq_adaptive = adapt
            encode_ours.rdoq_select = lambda kind: olib().kso_experiment_rdoq(tabs[kind].ctypes.data_as(C.c_void_p), a.rdoq, mult)
    if a.split_bits_b:
        os.environ["RD_SPLIT_BITS_B"] = a.split_bits_b
    if a.merge_bits_b:
        from oracle_lib import lib as olib2
        olib2().kso_experiment_merge_bits_b(a.merge_bits_b)
    ref = None
    if not a.no_ref:
        with tempfile.NamedTemporaryFile(suffix=".yuv", delete=False) as f:
            clip.tofile(f)
        size, p, per = encode_ref(f.name, W, H, a.ref_qp, a.gop, a.frames)
        os.unlink(f.name)
        ref = (size, p)
        bykind = {}
        for d, k, b, e in per:
            bykind.setdefault(k, []).append(b)
        print(f"reference  qp {a.ref_qp}: {size:8d} B  {p:.3f} dB   " + "  ".join(f"{k}: {int(np.mean(v))} B x{len(v)}" for k, v in bykind.items()), flush=True)
        if a.v:
            for d, k, b, e in sorted(per):
                print(f"   ref {d:3d} {k} {b:7d} B  {e:.2f} dB")
    pts = []
    for qp in (int(x) for x in a.qps.split(",")):
        t0 = time.time()
        seq = None
        if (a.adaptive or a.host) and a.gop == "hier":               # --host without --adaptive: the host's own layout (blocks of 8, its flush), no decision
            seq, four = adaptive_seq(clip, W, H, qp, decide=a.adaptive)
            if a.adaptive:
                print(f"   slice types at qp {qp}: {four} of {(len(clip) - 1) // 8} blocks of 8 as 4 + 4", flush=True)
        bs, p, per = encode_ours(clip, W, H, qp, a.gop, tools, a.v, stats_lib() if a.stats else None, a.pdelta, [int(x) for x in a.cascade.split(',')] if a.cascade else None, a.lam_scale,
                                 [int(x) for x in a.rdo_layers.split(',')] if a.rdo_layers else None, [float(x) for x in a.b_lam.split(',')] if a.b_lam else None,
                                 [int(x) for x in a.layer_qp.split(',')] if a.layer_qp else None, seq=seq)
        bykind, pk = {}, {}
        for d, k, l, b, e in per:
            bykind.setdefault(f"{k}{l if k == 'B' else ''}", []).append(b)
            pk.setdefault(f"{k}{l if k == 'B' else ''}", []).append(e)
        pts.append((len(bs), p))
        print(f"ours {a.tag} qp {qp}: {len(bs):8d} B  {p:.3f} dB (chroma {encode_ours.chroma_psnr:.2f})   " + "  ".join(f"{k}: {int(np.mean(v))} B x{len(v)} {np.mean(pk[k]):.2f}dB" for k, v in bykind.items()) + f"   ({time.time() - t0:.0f} s)", flush=True)
    if ref and len(pts) >= 2:
        pts.sort(key=lambda t: t[1])
        lo = [t for t in pts if t[1] <= ref[1]]
        hi = [t for t in pts if t[1] >= ref[1]]
        if lo and hi:
            (b0, p0), (b1, p1) = lo[-1], hi[0]
            b = b0 if p1 == p0 else np.exp(np.log(b0) + (ref[1] - p0) / (p1 - p0) * (np.log(b1) - np.log(b0)))
            print(f"==> at the reference's PSNR-Y {ref[1]:.2f} dB: ours {b:.0f} B vs {ref[0]} B = {b / ref[0]:.3f} x   {a.tag}")
        else:
            print(f"==> reference PSNR {ref[1]:.2f} outside our range {pts[0][1]:.2f}..{pts[-1][1]:.2f}")
    print(json.dumps({"size": a.size, "gop": a.gop, "frames": a.frames, "tools": tools, "ref": ref, "ours": pts, "tag": a.tag}))


if __name__ == "__main__":
    main()
