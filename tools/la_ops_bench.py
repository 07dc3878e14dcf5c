#!/usr/bin/env python3
"""GPU: the lookahead operators of csrc/lookahead_ops.hip timed at 2160p's block counts (240 x 135 blocks of 16 x 16): HIP-event time per call, and the algorithmic bytes
each call moves (ks265_frame_adapt_quant: the picture, 12.4 MB; ks265_cutree_propagate: 21 bytes per block in, 4 out).  Run under rocprofv3 --kernel-trace --stats for the
per-kernel table (tools/r4_profiles.sh)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ks265codec_amd.lib import KsContext
from ks265codec_amd.synth import make_clip

ks = KsContext(0)
W, H = 3840, 2160
nx, ny = W // 16, H // 16
n = nx * ny
fr = make_clip(W, H, 1, seed=7, abc=(67, 91, 33), pan=(8, 5))[0]
Y, U, V = ks.dev(fr[:W * H]), ks.dev(fr[W * H:W * H * 5 // 4]), ks.dev(fr[W * H * 5 // 4:])
rng = np.random.default_rng(3)
intra = rng.integers(1, 16000, n).astype(np.uint16)
d = dict(intra=intra, invq=rng.integers(100, 700, n).astype(np.uint16), own=rng.integers(0, 3000, n).astype(np.uint16), inter=np.minimum(intra, rng.integers(0, 16000, n)).astype(np.uint16),
         bits=rng.integers(0, 256, (n + 3) // 4).astype(np.uint8), mv0=((rng.integers(-200, 200, n) & 0xffff) | (rng.integers(-200, 200, n) << 16)).astype(np.int32),
         mv1=((rng.integers(-40, 40, n) & 0xffff) | (rng.integers(-40, 40, n) << 16)).astype(np.int32), r0=rng.integers(0, 2000, n).astype(np.uint16), r1=rng.integers(0, 2000, n).astype(np.uint16))
dv = {k: ks.dev(v) for k, v in d.items()}
acc = ks.zeros(16 * n)


def timed(fn, reps=50):
    fn(); ks.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ks.sync()
    return (time.perf_counter() - t) / reps * 1e3


aq = timed(lambda: ks.frame_adapt_quant(Y, W, U, V, W // 2, nx, ny, 1.0))
ct = timed(lambda: ks.cutree_propagate(3, nx, ny, dv["intra"], dv["invq"], dv["own"], dv["inter"], dv["bits"], dv["mv0"], dv["mv1"], dv["r0"], dv["r1"], acc))
print(f"ks265_frame_adapt_quant 2160p: {aq:.3f} ms per call incl. the copy-back of the {n} offsets ({W * H * 3 // 2 / 1e6:.1f} MB read)")
print(f"ks265_cutree_propagate {nx} x {ny} blocks: {ct:.3f} ms per call ({n * 25 / 1e6:.2f} MB)")
