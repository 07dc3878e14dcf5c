"""diagnostic (scratch): information content of the slice data by syntax category, per picture, for the GPU pipeline's records at 2160p hier-B 8"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
out = os.path.join(ROOT, "gpurun_out", "bitstats"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libstats.so")
subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-DKS265_BIT_STATS", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so,
                       os.path.join(ROOT, "ks265codec_amd/host/ks265_stream.c"), "-lm"])
from ks265codec_amd import stream as S
S._lib = C.CDLL(so)
for n in ("ks265_write_vps", "ks265_write_sps", "ks265_write_pps", "ks265_write_slice"):
    getattr(S._lib, n).restype = C.c_long
S._lib.ks265_slice_scratch_bytes.restype = C.c_size_t
import stream_cases as sc
from ks265codec_amd.lib import CU8, SAO_PARAM, KsContext, KsFrame
from ks265codec_amd.synth import lambda_q4, make_clip, psnr
W, H, qp, G = int(sys.argv[1]), int(sys.argv[2]), 27, 8
kind = sys.argv[3] if len(sys.argv) > 3 else "hier"
sched = sc.schedule(kind, G if kind == "hier" else 9)
n = 1 + max(s[0] for s in sched)
clip = make_clip(W, H, n, seed=7, abc=(67, 91, 33), pan=(8, 5))
ks = KsContext(0)
f = KsFrame(ks, W, H, qp, lambda_q4(qp), me_method=2, me_hex_thr=16, bframes=3, sdh=1, pre_search=1, merge=1, bi_refine=int(os.environ.get("BIR", "1")), rdo=4, intra_inter=1, propagate=1)      # the host's tool set (round 3)
g = f.geom
src = f.new_pic()
dpb = {}
w = S.StreamWriter(W, H, sao=1, deblock=1, max_dec_pic_buffering=G + 2, max_num_reorder=G if kind == "hier" else 0, sdh=1, wpp=0)
bs = w.headers()
names = ["cu", "merge", "motion", "coef", "sao", "intra"]
st = (C.c_double * 6)()
tot = {}
for d, k, l0, l1, dq, rps, isref in sched:
    q = min(51, qp + dq)
    f.set_qp(q, lambda_q4(q, inter=k != "I"))                  # P / B pictures: the inter lambda table, as the host
    f.load_i420(ks.dev(clip[d]), src)
    o = f.new_pic()
    if k == "B": f.encode_picture_b(src, dpb[l0[0]], dpb[l1[0]], o)
    else: f.encode_picture(src, dpb[l0[0]] if l0 else o, k == "I", o)
    dpb[d] = o
    cu8 = f.ws_read("cu8", g.bytes_cu8).view(CU8)
    lvl = [f.ws_read("levels", W * H * 2, 0).view(np.int16), f.ws_read("levels", W * H // 2, 1).view(np.int16), f.ws_read("levels", W * H // 2, 2).view(np.int16)]
    saop = f.ws_read("sao", g.bytes_sao).view(SAO_PARAM)
    S._lib.ks265_bit_stats(st, 1)
    if k == "I": b = w.slice(S.NAL_IDR_W_RADL, S.SLICE_I, 0, q, cu8, lvl, saop)
    else: b = w.slice(S.NAL_TRAIL_R if isref else S.NAL_TRAIL_N, S.SLICE_P if k == "P" else S.SLICE_B, d, q, cu8, lvl, saop, rps=rps, l0=l0, l1=l1)
    S._lib.ks265_bit_stats(st, 1)
    bs += b
    rec = ks.host(f.store_i420(o), np.uint8)
    inter = cu8["pred_mode"] == 0 if "pred_mode" in cu8.dtype.names else None
    dirs = np.bincount(cu8["inter_dir"] & 3, minlength=4) if k != "I" else None
    nz = [int((a != 0).sum()) for a in lvl]
    print(f"d{d:3d} {k} qp{q} {len(b):8d} B  psnr {psnr(clip[d][:W*H], rec[:W*H]):.2f}  " + "  ".join(f"{nm} {st[i]/8:9.0f}" for i, nm in enumerate(names)) + f"  nz {nz} dirs {dirs}", flush=True)
    key = (k, dq)
    t = tot.setdefault(key, [0, 0] + [0.0] * 6)
    t[0] += 1; t[1] += len(b)
    for i in range(6): t[2 + i] += st[i] / 8
for key, t in sorted(tot.items()):
    print(key, f"n {t[0]} bytes/pic {t[1]/t[0]:.0f}  " + "  ".join(f"{nm} {t[2+i]/t[0]:.0f}" for i, nm in enumerate(names)))
open(os.path.join(out, f"s_{kind}.265"), "wb").write(bs)
