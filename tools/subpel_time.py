"""me_subpel / me_int / stage times of P pictures with a given library: python tools/subpel_time.py [lib.so]"""
import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ks265codec_amd.lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.abspath(sys.argv[1])
from ks265codec_amd.lib import KsContext, KsFrame
from ks265codec_amd.synth import make_clip, lambda_q4, ENCODER_TOOLS
W, H = 3840, 2160
clip = make_clip(W, H, 10, seed=7, abc=(67, 91, 33), pan=(8, 5))
ks = KsContext(0)
f = KsFrame(ks, W, H, 27, lambda_q4(27), **ENCODER_TOOLS)
f.set_profiling(True)
src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
tot, n = {}, 0
for t in range(10):
    q = 27 + (t > 0); f.set_qp(q, lambda_q4(q, inter=t > 0))
    f.load_i420(ks.dev(clip[t]), src)
    f.encode_picture(src, a, t == 0, b); a, b = b, a
    ks.sync()
    if t >= 3:
        ms = f.stage_ms(); ms["me_int_kernel"] = f.me_int_ms()
        for k, v in ms.items(): tot[k] = tot.get(k, 0) + v
        n += 1
print(os.path.basename(L.LIB_PATH), {k: round(v / n, 4) for k, v in tot.items()})
