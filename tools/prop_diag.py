"""GPU: dump the records of the propagation stage (device and oracle) for offline comparison"""
import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ks265codec_amd.lib import PU, KsFrame, KsContext
from ks265codec_amd.synth import lambda_q4, make_clip
from oracle_lib import OraclePipeline, ptr
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r3final5'
ks = KsContext(0)
for (W, H, seed, me, pre) in ((200, 136, 5, 0, 0), (416, 240, 31, 2, 1)):
    clip = make_clip(W, H, 3, seed=seed, pan=(8, 5))
    kw = dict(me_method=me, me_hex_thr=16 if me == 2 else 0, pre_search=pre, propagate=2)
    o = OraclePipeline(W, H, 27, lambda_q4(27), **kw)
    with KsFrame(ks, W, H, 27, lambda_q4(27), **kw) as f:
        g = f.geom
        src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
        o.encode_picture(clip[0], True)
        f.load_i420(ks.dev(clip[0]), src); f.encode_picture(src, a, True, b)
        o.load(o.src, clip[1])
        opu, off = np.zeros(o.nctu * 85, PU), np.zeros(2 * o.nctu, np.int16)
        o.o.kso_me_integer_ex(C.byref(o.cfg), o.src.c(), o.ref.c(), None, ptr(opu), ptr(off))
        onext = np.zeros_like(opu)
        o.o.kso_me_propagate(C.byref(o.cfg), o.src.c(), o.ref.c(), ptr(off), ptr(opu), ptr(onext))
        f.load_i420(ks.dev(clip[1]), src)
        pu = [ks.zeros(g.bytes_pu), ks.zeros(g.bytes_pu)]
        f.me_integer(src, b, None, pu[0])
        g0 = ks.host(pu[0], PU).copy()
        f.me_propagate(src, b, pu[0], pu[1])
        g1 = ks.host(pu[1], PU).copy()
        np.savez_compressed(os.path.join(out, f'prop_{W}x{H}.npz'), opu=opu, onext=onext, off=off, g0=g0, g1=g1)
        bad = np.nonzero(g1 != onext)[0]
        print(W, H, 'integer records differ:', int((g0 != opu).sum()), ' propagated differ:', len(bad), ' changed by oracle round:', int((onext != opu).sum()))
        for i in bad[:12]:
            print('  ctu', i // 85, 'pu', i % 85, 'in', opu[i], 'dev', g1[i], 'oracle', onext[i])
