"""experiment: phase clocks of the intra chain (variant library built with -DKS_INTRA_CLOCK).  usage on the GPU box: KS_VARIANT=build/variants/libks265hip_clk.so (profiles/scripts/build_variant.sh clk -DKS_INTRA_CLOCK) python tools/intra_clock.py"""
import ctypes as C, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
var = os.environ.get("KS_VARIANT")
if var:
    shutil.copy(os.path.join(ROOT, var), os.path.join(ROOT, "ks265codec_amd", "libks265hip.so"))
import numpy as np, torch
from ks265codec_amd.lib import KsContext, KsFrame, load_library
from ks265codec_amd.synth import ENCODER_TOOLS, lambda_q4, make_clip
W, H = 3840, 2160
clip = make_clip(W, H, 17, seed=7, abc=(67, 91, 33), pan=(8, 5))
ks = KsContext(0)
lib = load_library()
names = ["setup", "wait-nbr", "window", "z-walk", "cu32", "mask", "gather", "smooth/dc", "tu8", "tu16", "z-end", "fence", "", "", "", "", "n8", "n16", "n32", "ctus"]
def dump(tag):
    buf = (C.c_ulonglong * 64)()
    assert lib.ks265_debug_clock_read(buf, 1) == 0
    for m, nm in ((0, "key picture"), (1, "P / B pictures")):
        a = [buf[m * 32 + i] for i in range(32)]
        tot = sum(a[:12])
        if not tot: continue
        print(f"{tag} {nm}: CUs 8/16/32 = {a[16]}/{a[17]}/{a[18]}, CTUs {a[19]}; wave-0 cycles by phase (share of {tot/1e6:.1f} M; per CU where it applies):")
        ncu = max(1, a[16] + a[17] + a[18])
        for i in range(12):
            per = {4: a[18], 8: a[16], 9: a[17]}.get(i, ncu if i in (3, 5, 6, 7) else max(1, a[19]))
            print(f"   {names[i]:10s} {100.0*a[i]/tot:5.1f} %   {a[i]/max(1,per):8.0f} cycles per {'CU' if i in (3,4,5,6,7,8,9) else 'CTU'}")
        tn = ["pred+res", "fwd1", "fwd2+quant", "rdo", "nz count", "sdh", "dequant+lvl", "inv1", "inv2+store"]
        small = max(1, a[16] + a[17])
        print("   inside the small luma TU pipeline (cycles per CU): " + "  ".join(f"{tn[i]} {a[20+i]/small:.0f}" for i in range(9)))
with KsFrame(ks, W, H, 27, lambda_q4(27), bframes=7, refs=3, **ENCODER_TOOLS) as f:
    src, a, b = f.new_pic(), f.new_pic(), f.new_pic()
    f.set_qp(27, lambda_q4(27)); f.load_i420(ks.dev(clip[0]), src)
    for _ in range(2):
        f.encode_picture(src, a, True, a); ks.sync()
    dump("")
    f.set_qp(28, lambda_q4(28, inter=True))
    for t in (8, 16):
        f.load_i420(ks.dev(clip[t]), src)
        f.encode_picture(src, a, False, b); ks.sync()
        a, b = b, a
    dump("anchors 8 apart")
