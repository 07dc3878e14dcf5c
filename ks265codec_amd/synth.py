"""Synthetic I420 clips of SURVEY.md §8(d): smooth sinusoid luma + static texture, global pan, moving textured
squares, optional fresh per-frame noise; slow sinusoid chroma.  Deterministic (seeded numpy)."""
from __future__ import annotations

import numpy as np


def make_clip(width: int, height: int, frames: int, seed: int = 42, noisy: bool = True, abc=(37, 53, 19), pan=(5, 3)) -> np.ndarray:
    """returns uint8 array [frames, width*height*3/2] (planar I420, frames concatenated, no header)"""
    rng = np.random.default_rng(seed)
    a, b, c = abc
    big_w, big_h = width + pan[0] * frames + 16, height + pan[1] * frames + 16
    yy, xx = np.mgrid[0:big_h, 0:big_w].astype(np.float64)
    base = 128 + 60 * np.sin(xx / a) + 50 * np.cos(yy / b) + 20 * np.sin((xx + yy) / c) + rng.integers(-10, 11, (big_h, big_w))
    nsq = 6 + int(rng.integers(0, 3))
    squares = []
    for _ in range(nsq):
        s = int(rng.integers(max(16, height // 12), max(32, height // 4)))
        s = max(1, min(s, width // 2, height // 2))                 # tiny test pictures: keep the squares inside
        squares.append(dict(size=s, x=float(rng.integers(0, max(1, width - s))), y=float(rng.integers(0, max(1, height - s))),
                            vx=float(rng.integers(-9, 10)), vy=float(rng.integers(-6, 7)),
                            tex=np.clip(rng.integers(40, 216) + rng.integers(-25, 26, (s, s)), 0, 255)))
    cy, cx = np.mgrid[0:height // 2, 0:width // 2].astype(np.float64)
    out = np.empty((frames, width * height * 3 // 2), np.uint8)
    for t in range(frames):
        ox, oy = pan[0] * t, pan[1] * t
        y = base[oy:oy + height, ox:ox + width].copy()
        for q in squares:
            px = int(round(q["x"] + q["vx"] * t)) % max(1, width - q["size"])
            py = int(round(q["y"] + q["vy"] * t)) % max(1, height - q["size"])
            y[py:py + q["size"], px:px + q["size"]] = q["tex"]
        if noisy:
            y = y + rng.integers(-2, 3, (height, width))
        u = 128 + 30 * np.sin((cx + 2 * t) / 41.0) + 20 * np.cos(cy / 29.0)
        v = 128 + 25 * np.cos((cx - t) / 33.0) + 25 * np.sin((cy + t) / 47.0)
        out[t] = np.concatenate([np.clip(np.rint(p), 0, 255).astype(np.uint8).reshape(-1) for p in (y, u, v)])
    return out


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


# lambda of P / B pictures: HM's factor for pictures that are not key pictures, clip(2, 4, (qp - 12) / 6) on lambda_mode (the reference's own integer motion lambda,
# read from its cost tables - 8 at qp 27, 9 at 28, 16 at 33, 18 at 34 - is within 10 percent of this); a table so that every host uses identical integers (ks265_enc.c kLambdaInterQ4)
LAMBDA_INTER_Q4 = [4, 5, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 17, 19, 22, 24, 27, 30, 34, 38, 43, 48, 54, 61, 68, 80, 93, 108, 125, 145, 167, 193, 222, 256, 294, 337, 387, 434, 487, 547, 614, 689, 773, 868, 974, 1093, 1227, 1378, 1546, 1736, 1948, 2187]


def lambda_q4(qp: int, inter: bool = False) -> int:
    """motion lambda in Q4 (host-side float setup, HM-style sqrt(0.57 * 2^((qp-12)/3))); inter: the P / B picture table"""
    if inter:
        return LAMBDA_INTER_Q4[qp]
    return int(round(16.0 * (0.57 * 2.0 ** ((qp - 12) / 3.0)) ** 0.5))


# What the reference's nine presets (ultrafast .. placebo) put into the configuration words its sub-pel refinement reads, measured inside real `appencoder` runs
# (oracle/ref_probe/subme_shim.c): (subme, sub_satd, sub_thr, sub_flat, sub_cap, sub_cap_step, sub_diag_fast).  host/ks265_enc.c holds the same table.
SUBME_PRESET = {
    "ultrafast": (1, 0, 80, 40, 6, 6, 1), "superfast": (1, 0, 76, 36, 6, 6, 1), "veryfast": (1, 0, 68, 16, 12, 6, 0), "fast": (1, 0, 56, 14, 0, 0, 0),
    "medium": (1, 0, 40, 10, 0, 0, 0), "slow": (1, 0, 24, 8, 0, 0, 0), "slower": (1, 0, 24, 8, 0, 0, 0), "veryslow": (2, 1, 0, 8, 0, 0, 0), "placebo": (2, 1, 0, 8, 0, 0, 0),
}


def subme_knobs(preset: str) -> dict:
    return dict(zip(("subme", "sub_satd", "sub_thr", "sub_flat", "sub_cap", "sub_cap_step", "sub_diag_fast"), SUBME_PRESET[preset]))


# THE tool set of the encoder host at -preset slow (host/ks265_enc.c: encoder_open) - bench.py, __graft_entry__.smoke(), the GPU tests and tools/rd_eval.py --host all
# import this one dict, so that what is timed is what is tested.
ENCODER_TOOLS = dict(me_method=2, me_hex_thr=16, sdh=1, pre_search=1, merge=1, bi_refine=0, rdo=4, intra_inter=1, propagate=1, skip_rd=1, **subme_knobs("slow"))     # (bi_refine: 2 from -preset slower on)


# the encoder host's QP ladders at -rc 0 (host/ks265_enc.c kIpppCascade / kHierLayerQp = the reference's own, read from its -psnr 2 lines): the QP of a picture is the
# key picture's + host_qp_offset(...).  bench.py's hot-path leg, tests/stream_cases.py and tools/rd_eval.py --host use these.
HOST_IPPP_CASCADE = (0, 2, 1, 2)
HOST_HIER_LAYER_QP = (0, 1, 3, 3)


def host_qp_offset(kind: str, pos_in_gop: int = 0, layer: int = 0, hier: bool = False) -> int:
    """kind 'I' / 'P' / 'B'; IPPP: pos_in_gop = the P picture's position behind its key picture; hierarchy: layer 0 = anchors, 1..3 = B layers; plain B: + 2"""
    if kind == "I":
        return 0
    if hier:
        return 1 + HOST_HIER_LAYER_QP[min(layer, 3)]
    return 2 if kind == "B" else 1 + HOST_IPPP_CASCADE[pos_in_gop & 3]
