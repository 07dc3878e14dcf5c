"""Shared by tests/test_stream.py, tests/test_gpu_stream.py and tests/golden/gen_stream_golden.py: the stream-level parity cases.
Each case encodes a short synthetic GOP with a per-picture `encode(kind, display index, refs)` callback supplied by the caller (CPU
oracle pipeline or HIP pipeline) and writes the Annex-B stream with the host bitstream writer."""
from __future__ import annotations

import itertools

import numpy as np

CASES = {
    # name: (W, H, qp, me_method, me_hex_thr, sao, deblock, gop kind, frames / parameter)
    "ippp_416x240_umh": (416, 240, 27, 2, 16, 1, 1, "ippp", 4),
    "ippp_200x136_hex_qp37": (200, 136, 37, 1, 0, 1, 1, "ippp", 4),
    "ippp_136x72_dia_qp12": (136, 72, 12, 0, 0, 1, 1, "ippp", 3),
    "ippp_72x200_nodeblock_qp45": (72, 200, 45, 1, 0, 1, 0, "ippp", 3),
    "ippp_8x8": (8, 8, 30, 1, 0, 1, 1, "ippp", 3),
    "ippp_64x64_nosao_qp5": (64, 64, 5, 2, 0, 0, 1, "ippp", 3),
    "mref3_416x240": (416, 240, 27, 2, 16, 1, 1, "mref", 3),
    "hierb4_416x240": (416, 240, 27, 1, 0, 1, 1, "hier", 4),
    "ippp_1280x720_umh": (1280, 720, 27, 2, 16, 1, 1, "ippp", 3),
    # sign-data hiding on (the reference's PPS setting): levels changed by signBitHidingHDQ in the pixel path, hidden signs in the stream
    "sdh_ippp_416x240_umh": (416, 240, 27, 2, 16, 1, 1, "ippp", 4),
    "sdh_hierb4_416x240": (416, 240, 30, 1, 0, 1, 1, "hier", 4),
    "sdh_ippp_200x136_qp12": (200, 136, 12, 1, 0, 1, 1, "ippp", 3),
    # what the C host (ks265_enc.c) runs: sign-data hiding + the pre-search start candidates of stage A
    "ps_ippp_416x240_umh": (416, 240, 27, 2, 16, 1, 1, "ippp", 4),
    "ps_hierb4_416x240": (416, 240, 30, 1, 0, 1, 1, "hier", 4),
    "ps_mref3_200x136": (200, 136, 33, 2, 16, 1, 1, "mref", 3),
    # + entropy_coding_sync (CTU rows as substreams with entry points): the C host's streams
    "wpp_ippp_416x240_umh": (416, 240, 27, 2, 16, 1, 1, "ippp", 4),
    "wpp_hierb4_416x240": (416, 240, 30, 1, 0, 1, 1, "hier", 4),
    "wpp_ippp_56x200_qp12": (56, 200, 12, 1, 0, 1, 1, "ippp", 3),      # one CTU per row: no context hand-over, every row starts from the initial contexts
    "wpp_ippp_1280x720_umh": (1280, 720, 27, 2, 16, 1, 1, "ippp", 3),
    # everything the C host (ks265_enc.c) switches on: sign-data hiding, pre-search, merge pass, WPP substreams
    "enc_ippp_416x240_umh": (416, 240, 27, 2, 16, 1, 1, "ippph", 4),      # ippph: IPPP with the encoder host's QP ladder (P pictures at + 1 + {0, 2, 1, 2}[position & 3])
    "enc_hierb4_416x240": (416, 240, 30, 1, 0, 1, 1, "hier", 4),
    "enc_ippp_1280x720_qp32": (1280, 720, 32, 1, 0, 1, 1, "ippph", 3),
    # -part 1 (slower / veryslow / placebo): CUs of 64 / 32 / 16 in two 2NxN or Nx2N prediction units with four transform units; the rest as the C host runs it
    "part_ippp_416x240_umh": (416, 240, 27, 2, 0, 1, 1, "ippph", 4),
    "part_ippp_200x136_qp34": (200, 136, 34, 1, 0, 1, 1, "ippph", 4),
    # round 5: -part 1 in B pictures (config 5 = -preset veryslow codes hierarchical B with part 1): the halves of a CU differ in direction and vectors
    "part_hierb4_416x240": (416, 240, 28, 2, 0, 1, 1, "hier", 4),
    "part_hierb4_200x136_qp34": (200, 136, 34, 1, 0, 1, 1, "hier", 4),
    # round 5: several reference pictures per list in B pictures (config 5: -ref 4): list 0 = the nearest past pictures, list 1 = the nearest future ones, ref_idx_l0 / _l1 per PU
    # round 5: -intertu 1 (tuInter: veryslow, placebo): 2Nx2N inter CUs of 32 / 16 with four transform units where the residual sits in part of the CU; split_transform_flag in the stream
    "rqt_ippp_416x240_umh": (416, 240, 27, 2, 0, 1, 1, "ippph", 4),
    "rqt_part_hierb4_416x240": (416, 240, 29, 2, 0, 1, 1, "hier", 4),
    "rqt_part_hiermr4_200x136": (200, 136, 31, 1, 0, 1, 1, "hiermr", 4),
    "enc_hiermr4_416x240": (416, 240, 29, 2, 16, 1, 1, "hiermr", 4),
    "part_hiermr4_200x136_qp33": (200, 136, 33, 1, 0, 1, 1, "hiermr", 4),
    # round 5: cfg.bi_refine = 2 (what the C host runs from round 5 on): the joint refinement of a bi-predictive pair after the CU decision, for the CUs it chose
    "enc_hierb4_416x240_bir2": (416, 240, 30, 1, 0, 1, 1, "hier", 4),
    "enc_hiermr4_416x240_bir2": (416, 240, 29, 2, 16, 1, 1, "hiermr", 4),
    "part_hierb4_200x136_qp34_bir2": (200, 136, 34, 1, 0, 1, 1, "hier", 4),
    "rqt_part_hiermr4_200x136_bir2": (200, 136, 31, 1, 0, 1, 1, "hiermr", 4),
    # round 6: -ref0 3 (every preset from superfast up): the anchors of the pyramid search the last three anchors of their GOP - multi-reference P pictures with the host's whole
    # tool set (intra candidates against the two-list records, merge pass on the records' pictures, intra CUs), three mini-GOPs so that the last anchor has three pictures
    "enc_hiera3_416x240_bir2": (416, 240, 29, 2, 16, 1, 1, "hiera", 4),
    "enc_hiera3_200x136_qp34_bir2": (200, 136, 34, 1, 0, 1, 1, "hiera", 4),
    # round 6: cfg.skip_rd (stage D2, what the C host runs from round 6 on): after the reconstruction, nodes whose merge candidate without residual is the cheaper coding become one CU -
    # P pictures, B pictures, multi-reference anchors, multi-reference B pictures with -part 1 / -intertu 1 around them
    "enc_ippp_416x240_umh_skip": (416, 240, 27, 2, 16, 1, 1, "ippph", 4),
    "enc_hierb4_416x240_skip_bir2": (416, 240, 30, 1, 0, 1, 1, "hier", 4),
    "enc_hiera3_416x240_skip_bir2": (416, 240, 29, 2, 16, 1, 1, "hiera", 4),
    "enc_hiera3_200x136_qp34_skip_bir2": (200, 136, 34, 1, 0, 1, 1, "hiera", 4),
    "rqt_part_hiermr4_200x136_skip_bir2": (200, 136, 31, 1, 0, 1, 1, "hiermr", 4),
}


def case_part(name: str) -> int:
    return 1 if name.startswith("part_") or name.startswith("rqt_part_") else 0


def case_rqt(name: str) -> int:
    return 1 if name.startswith("rqt_") else 0


def case_skip(name: str) -> int:
    return 2 if "_skip" in name else 0          # (2: P pictures too - the C host runs 1 = B pictures only)


def case_sdh(name: str) -> int:
    return 1 if name.startswith(("sdh_", "ps_", "wpp_", "enc_", "part_", "rqt_")) else 0


def case_ps(name: str) -> int:
    return 1 if name.startswith(("ps_", "wpp_", "enc_", "part_", "rqt_")) else 0


def case_wpp(name: str) -> int:
    return 1 if name.startswith(("wpp_", "enc_", "part_", "rqt_")) else 0


def case_merge(name: str) -> int:
    return 1 if name.startswith(("enc_", "part_", "rqt_")) else 0


def case_bir(name: str) -> int:
    return 2 if name.endswith("_bir2") else 1 if name.startswith(("enc_", "part_", "rqt_")) else 0


def case_dec(name: str) -> int:
    return 0


def case_rdo(name: str) -> int:
    return 4 if name.startswith(("enc_", "part_", "rqt_")) else 0            # the C host: coefficient-group pruning (superseded the coefficient decimation of round 2)


def case_ii(name: str) -> int:
    return (2 if "1280x720" in name else 1) if name.startswith(("enc_", "part_", "rqt_")) else 0   # the C host: intra CUs in P / B pictures (one case with 2 = 16x16 and 32x32 only)


def case_prop(name: str) -> int:
    return 1 if name.startswith(("enc_", "part_", "rqt_")) else 0            # the C host: one round of vector propagation after every integer search (stage A2)


def case_subme(name: str) -> dict:
    """the sub-pel knobs: the C host's (-preset slow: fast candidate sets judged by SAD) for the enc_ cases, -preset veryslow's (all 8 + 8 candidates judged by
    Hadamard) for the wpp_ cases, -preset medium's for the hierarchical ones, veryfast's for the rest"""
    from ks265codec_amd.synth import subme_knobs
    return subme_knobs("veryslow" if name.startswith("rqt_") else "slower" if name.startswith("part_") else "slow" if name.startswith("enc_") else "veryslow" if name.startswith("wpp_") else "medium" if "hier" in name else "veryfast")


def case_lambda(name: str, q: int, kind: str) -> int:
    """the C host prices P / B pictures with the inter table (HM's factor for pictures that are not key pictures)"""
    from ks265codec_amd.synth import lambda_q4
    return lambda_q4(q, inter=name.startswith(("enc_", "part_", "rqt_")) and kind != "I")


from ks265codec_amd.synth import HOST_IPPP_CASCADE      # ks265_enc.c kIpppCascade: the QP of an IPPP P picture is the key picture's + 1 + this, by its position in the GOP (the reference's 30 / 29 / 30 / 28 at -qp 27)


def schedule(kind: str, par: int):
    """list of (display index, picture kind, list-0 display indices, list-1 display indices, qp offset, rps [(display index, used)], is reference)"""
    from ks265codec_amd.gop import hier_order
    out = []
    if kind in ("ippp", "ippph"):
        for t in range(par):
            out.append((t, "I" if t == 0 else "P", [t - 1] if t else [], [], 0 if t == 0 else 1 + (HOST_IPPP_CASCADE[t & 3] if kind == "ippph" else 0), [(t - 1, True)] if t else [], True))
    elif kind == "mref":
        for t in range(par + 3):
            refs = [t - 1 - i for i in range(min(par, t))]
            out.append((t, "I" if t == 0 else "P", refs, [], 0 if t == 0 else 1, [(p, True) for p in refs], True))
    elif kind == "hiera":
        # the hierarchy of "hier" over three mini-GOPs, the anchors (P) with the last three anchors in list 0, nearest first (ks265_enc.c: -ref0 3); the older anchors stay in the
        # reference picture sets of the B pictures in between
        G = par
        seq = list(itertools.islice(hier_order(G, 128), 3 * G + 1))
        lists, anchors = [], []
        for d, k, r0, r1, layer in seq:
            lists.append((anchors[:3], []) if k == "P" else ([r0], [r1]) if k == "B" else ([], []))
            if k != "B":
                anchors.insert(0, d)
        for i, (d, k, r0, r1, layer) in enumerate(seq):
            l0, l1 = lists[i]
            later = lists[i + 1:]
            coded = {s[0] for s in seq[:i]}
            needed = {r for (a, b) in later for r in a + b if r in coded}
            cur = set(l0) | set(l1)
            rps = [(p, p in cur) for p in sorted(needed | cur)]
            isref = any(d in a + b for (a, b) in later)
            out.append((d, k, l0, l1, 0 if k == "I" else 1 + layer, rps, isref))
    elif kind == "hiermr":
        # the hierarchy of "hier", B pictures with up to two pictures per list: of the pictures coded so far that are not older than the previous mini-GOP's first anchor,
        # list 0 = the nearest two before the picture, list 1 = the nearest two after it; anchors keep their one reference.  The RPS of a picture = every picture a later one uses.
        G = par
        seq = list(itertools.islice(hier_order(G, 128), 2 * G + 1))
        lists, done = [], []
        for d, k, r0, r1, layer in seq:
            if k == "B":
                lo = ((d - 1) // G) * G - G
                av = [p for p in done if p >= lo]
                l0 = sorted([p for p in av if p < d], reverse=True)[:2]
                l1 = sorted([p for p in av if p > d])[:2]
            else:
                l0, l1 = ([r0] if r0 is not None else []), []
            lists.append((l0, l1))
            done.append(d)
        for i, (d, k, r0, r1, layer) in enumerate(seq):
            l0, l1 = lists[i]
            later = lists[i + 1:]
            coded = {s[0] for s in seq[:i]}
            needed = {r for (a, b) in later for r in a + b if r in coded}
            cur = set(l0) | set(l1)
            rps = [(p, p in cur) for p in sorted(needed | cur)]
            isref = any(d in a + b for (a, b) in later)
            out.append((d, k, l0, l1, 0 if k == "I" else 1 + layer, rps, isref))
    else:
        G = par
        seq = list(itertools.islice(hier_order(G, 128), 2 * G + 1))
        done = set()
        for i, (d, k, r0, r1, layer) in enumerate(seq):
            later = seq[i + 1:]
            needed = {r for (_, _, a, b, _) in later for r in (a, b) if r is not None and r in done}
            cur = {r for r in (r0, r1) if r is not None}
            rps = [(p, p in cur) for p in sorted(needed | cur)]
            isref = any(d in (a, b) for (_, _, a, b, _) in later)
            out.append((d, k, [r0] if r0 is not None else [], [r1] if r1 is not None else [], 0 if k == "I" else 1 + layer, rps, isref))
            done.add(d)
    return out


def make_stream(name: str, encode):
    """encode(display index, kind, l0 display indices, l1 display indices, qp) -> (cu8, [lvl_y, lvl_u, lvl_v], sao records or None, recon I420);
    returns (stream bytes, {display index: recon I420})"""
    from ks265codec_amd import stream as S
    W, H, qp, me, thr, sao, df, kind, par = CASES[name]
    sched = schedule(kind, par)
    nref = max([len(s[2]) + len(s[3]) for s in sched] + [1])
    reorder = par if kind in ("hier", "hiermr", "hiera") else 0
    w = S.StreamWriter(W, H, sao=sao, deblock=df, max_dec_pic_buffering=(par + 4) if kind in ("hiermr", "hiera") else (par + 2) if kind == "hier" else nref + 1, max_num_reorder=reorder, sdh=case_sdh(name), wpp=case_wpp(name), tu_inter=case_rqt(name))      # the C host's rule (ks265_enc.c)
    bs = w.headers()
    recs = {}
    for d, k, l0, l1, dq, rps, isref in sched:
        cu8, lvl, saop, rec = encode(d, k, l0, l1, min(51, qp + dq))
        recs[d] = rec
        if k == "I":
            bs += w.slice(S.NAL_IDR_W_RADL, S.SLICE_I, 0, min(51, qp + dq), cu8, lvl, saop if sao else None)
        else:
            nal = S.NAL_TRAIL_R if isref else S.NAL_TRAIL_N
            bs += w.slice(nal, S.SLICE_P if k == "P" else S.SLICE_B, d, min(51, qp + dq), cu8, lvl, saop if sao else None, rps=rps, l0=l0, l1=l1)
    return bs, recs


def oracle_encoder(name: str):
    """the CPU oracle pipeline as the per-picture encoder of make_stream"""
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    W, H, qp, me, thr, sao, df, kind, par = CASES[name]
    n = 1 + max(s[0] for s in schedule(kind, par))
    clip = make_clip(W, H, n, seed=len(name) * 7 + W, abc=(17, 23, 9))
    o = OraclePipeline(W, H, qp, lambda_q4(qp), me_method=me, me_hex_thr=thr, sao=sao, deblock=df, sdh=case_sdh(name), pre_search=case_ps(name), merge=case_merge(name), bi_refine=case_bir(name), decimate=case_dec(name), rdo=case_rdo(name), intra_inter=case_ii(name), propagate=case_prop(name), part=case_part(name), tu_inter=case_rqt(name), skip_rd=case_skip(name), **case_subme(name))
    dpb = {}

    def encode(d, k, l0, l1, q):
        o.set_qp(q, case_lambda(name, q, k))
        if k == "P" and len(l0) > 1:
            dpb[d] = o.encode_mref(clip[d], [dpb[r] for r in l0])
        elif k == "B" and (len(l0) > 1 or len(l1) > 1):
            dpb[d] = o.encode_b_mref(clip[d], [dpb[r] for r in l0], [dpb[r] for r in l1])
        else:
            dpb[d] = o.encode(clip[d], k, dpb.get(l0[0]) if l0 else None, dpb.get(l1[0]) if l1 else None)
        return o.cu8.copy(), [a.copy() for a in o.lvl], o.sao.copy(), o.store(dpb[d])
    return encode
