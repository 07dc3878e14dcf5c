"""per-work-group timeline of me_int_kernel (experiment build -DKS_EXP_ME_TRACE): python tools/me_trace.py [lib.so]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ks265codec_amd.lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.abspath(sys.argv[1])
from ks265codec_amd.lib import KsContext, KsFrame
from ks265codec_amd.synth import make_clip, lambda_q4, ENCODER_TOOLS
W, H = 3840, 2160
clip = make_clip(W, H, 6, seed=7, abc=(67, 91, 33), pan=(8, 5))
ks = KsContext(0)
f = KsFrame(ks, W, H, 27, lambda_q4(27), **ENCODER_TOOLS)
f.set_profiling(True)
src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
n = 2040
buf = (C.c_ulonglong * (8 * n))()
for t in range(6):
    q = 27 + (t > 0); f.set_qp(q, lambda_q4(q, inter=t > 0))
    f.load_i420(ks.dev(clip[t]), src)
    f.encode_picture(src, a, t == 0, b); a, b = b, a
    ks.sync()
    if t < 3: continue
    ks.lib.ks265_me_trace(buf, 8 * n)
    d = np.array(list(buf), dtype=np.float64).reshape(n, 8)
    t0 = d[:, 0].min()
    st, pro, l0 = (d[:, 0] - t0) / 100.0, (d[:, 1] - d[:, 0]) / 100.0, (d[:, 2] - d[:, 1]) / 100.0          # us
    ends = (d[:, 3:7] - t0) / 100.0
    end = ends.max(axis=1); life = end - st
    lv = ends - ((d[:, 2] - t0) / 100.0)[:, None]
    hw = d[:, 7].astype(np.uint64)
    print(f"picture {t}: me_int_ms {f.me_int_ms():.4f}  span {end.max():.1f} us; WG life mean {life.mean():.1f} p50 {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f}; "
          f"prologue mean {pro.mean():.1f} max {pro.max():.1f}; level0 mean {l0.mean():.1f} max {l0.max():.1f}; levels1-3 per wave mean {lv.mean():.1f} p90 {np.percentile(lv, 90):.1f} max {lv.max():.1f}; "
          f"wave imbalance inside a WG (max - mean of the 4 waves) mean {(lv.max(axis=1) - lv.mean(axis=1)).mean():.1f}")
    # concurrency over time
    T = np.linspace(0, end.max(), 30)
    conc = [(int(((st <= x) & (end > x)).sum())) for x in T]
    print("   running WGs over time:", conc)
    print(f"   sum of WG lifetimes / 768 slots = {life.sum() / 768:.1f} us; last start {st.max():.1f} us; WGs started after 50 % of the span: {(st > end.max() / 2).sum()}")
    order = np.argsort(st)
    print("   lifetimes of the last 12 WGs to start:", np.round(life[order[-12:]], 1), " of the 12 longest:", np.round(np.sort(life)[-12:], 1))
