"""ctypes binding of the CPU oracle (oracle/libks265_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None


def build_oracle() -> str:
    so = os.path.join(ORACLE_DIR, "libks265_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ks265_oracle.c", "ks265_intra_oracle.c", "ks265_pipeline_oracle.c", "ks265_oracle.h", "ks265_pipeline_oracle.h", "Makefile")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return so


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        _lib.ks265o_sad.restype = C.c_uint32
        _lib.ks265o_had.restype = C.c_uint32
        _lib.ks265o_sse.restype = C.c_uint32
        _lib.ks265o_quant.restype = C.c_int
    return _lib


def ptr(a: np.ndarray, byte_off: int = 0) -> C.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data + int(byte_off))


L = C.c_long
I = C.c_int


# ------------------------------------------------------------------ pipeline oracle (ks265_pipeline_oracle.h)
class OFrameCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "qp", "lambda_q4", "me_range", "me_method", "subme", "deblock", "sao",
                                          "beta_offset_div2", "tc_offset_div2", "bframes", "refs", "me_hex_thr", "sdh", "pre_search", "merge", "bi_refine", "decimate", "rdo", "intra_inter", "propagate", "sub_satd", "sub_thr", "sub_flat", "sub_cap", "sub_cap_step", "sub_diag_fast", "part", "tu_inter", "skip_rd")]


class OFrameGeom(C.Structure):
    _fields_ = [("pad_y", C.c_int32), ("pad_c", C.c_int32), ("stride_y", C.c_int32), ("stride_c", C.c_int32),
                ("rows_y", C.c_int32), ("rows_c", C.c_int32), ("bytes_y", C.c_int64), ("bytes_c", C.c_int64),
                ("ctu_cols", C.c_int32), ("ctu_rows", C.c_int32), ("pu_per_ctu", C.c_int32), ("bytes_pu", C.c_int64),
                ("bytes_cu8", C.c_int64), ("bytes_sao", C.c_int64)]


class OPic(C.Structure):
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p)]


class OMref(C.Structure):                   # kso_mref
    _fields_ = [("n0", C.c_int), ("n1", C.c_int), ("planes0", C.c_void_p * 4), ("planes1", C.c_void_p * 4), ("pic0", OPic * 4), ("pic1", OPic * 4), ("idx0", C.c_void_p), ("idx1", C.c_void_p)]


PU = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("mvpx", "<i2"), ("mvpy", "<i2"), ("cost", "<u4"), ("dist", "<u4")])
CU8 = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2"), ("log2_cu", "u1"), ("cbf", "u1"), ("pred_mode", "u1"), ("inter_dir", "u1")])
PU_B = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2"), ("cost", "<u4"), ("inter_dir", "<u4")])
SAO_PARAM = np.dtype([("type", "i1"), ("band", "i1"), ("offset", "i1", 4), ("rsv", "i1", 2)])


class HostPic:
    """padded picture in host memory"""

    def __init__(self, geom: OFrameGeom):
        self.y = np.zeros(geom.bytes_y, np.uint8)
        self.u = np.zeros(geom.bytes_c, np.uint8)
        self.v = np.zeros(geom.bytes_c, np.uint8)

    def c(self) -> OPic:
        return OPic(self.y.ctypes.data, self.u.ctypes.data, self.v.ctypes.data)


class OraclePipeline:
    """CPU restatement of the frame stages (test checker / cpu_baseline 'port')."""

    def __init__(self, width, height, qp, lambda_q4, me_range=64, subme=1, deblock=1, sao=1, me_method=0, intra=True, me_hex_thr=0, sdh=0, pre_search=0, merge=0, bi_refine=0, decimate=0, rdo=0, intra_inter=0, propagate=0,
                 sub_satd=0, sub_thr=24, sub_flat=8, sub_cap=0, sub_cap_step=0, sub_diag_fast=0, part=0, tu_inter=0, skip_rd=0):
        self.o = lib()
        self.intra = intra                      # key pictures: real intra prediction (True) or the flat stand-in
        self.skip_rd = int(os.environ.get("RD_SKIP", skip_rd))         # (RD_SKIP / RD_SKIP_PARAMS: experiment hooks of tools/rd_eval.py)
        if os.environ.get("RD_SKIP_PARAMS"):
            self.o.kso_experiment_skip((C.c_int * 8)(*([int(x) for x in os.environ["RD_SKIP_PARAMS"].split(",")] + [0] * 8)[:8]))
        self.cfg = OFrameCfg(width, height, qp, lambda_q4, me_range, me_method, subme, deblock, sao, 0, 0, 1, 4, me_hex_thr, sdh, pre_search, merge, bi_refine, decimate, rdo, intra_inter, propagate, sub_satd, sub_thr, sub_flat, sub_cap, sub_cap_step, sub_diag_fast, part, tu_inter, self.skip_rd)
        self.geom = OFrameGeom()
        assert self.o.kso_frame_geometry(C.byref(self.cfg), C.byref(self.geom)) == 0
        g = self.geom
        self.nctu = g.ctu_cols * g.ctu_rows
        self.planes = np.zeros(16 * g.bytes_y, np.uint8)
        self.pu = np.zeros(self.nctu * 85, PU)
        self.prev_pu = np.zeros(self.nctu * 85, PU)
        self.have_prev = False
        self.cu8 = np.zeros((height // 8) * (width // 8), CU8)
        self.sao = np.zeros(self.nctu * 3, SAO_PARAM)
        self.lvl = [np.zeros(width * height, np.int16), np.zeros(width * height // 4, np.int16), np.zeros(width * height // 4, np.int16)]
        self.src, self.rec, self.deb = HostPic(g), HostPic(g), HostPic(g)
        self.ref = HostPic(g)

    def set_picture_tools(self, intra_inter: int = -1, bi_refine: int = -1, sao: int = -1, me_method: int = -1) -> None:
        """ks265_frame_set_picture_tools: the tools of the pictures coded from here on (-1 = as created)"""
        if not hasattr(self, "_tools0"):
            self._tools0 = (self.cfg.intra_inter, self.cfg.bi_refine, self.cfg.sao, self.cfg.me_method)
        self.cfg.intra_inter, self.cfg.bi_refine, self.cfg.sao, self.cfg.me_method = (t0 if v < 0 else v for v, t0 in zip((intra_inter, bi_refine, sao, me_method), self._tools0))

    def skip_pass(self, r0: OPic, r1: OPic) -> None:
        """stage D2 (round 6): after the reconstruction of the inter CUs - nodes whose merge candidate without residual is the cheaper coding become one CU (kso_skip_pass)"""
        tmp = self.cu8.copy()
        self.cu_pre_skip = tmp
        self.o.kso_skip_pass(C.byref(self.cfg), self.src.c(), r0, r1, ptr(tmp), ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
        if os.environ.get("RD_SKIP_DEBUG"):
            w8, h8 = self.cfg.width // 8, self.cfg.height // 8
            a, b = tmp.reshape(h8, w8), self.cu8.reshape(h8, w8)
            n = m = 0
            for y in range(h8):
                for x in range(w8):
                    c = b[y, x]
                    if c == a[y, x] or c["pred_mode"] != 0:
                        continue
                    n8 = 1 << (int(c["log2_cu"] & 15) - 3)
                    if (x & (n8 - 1)) or (y & (n8 - 1)):
                        continue
                    n += 1
                    key = lambda q: (int(q["mvx"]), int(q["mvy"]), int(q["mv1x"]), int(q["mv1y"]), int(q["inter_dir"]))
                    cands = []
                    if x > 0: cands.append(b[y + n8 - 1, x - 1])
                    if y > 0: cands.append(b[y - 1, x + n8 - 1])
                    if y > 0 and x + n8 < w8 and ((x + n8) % 8 != 0 or True): cands.append(b[y - 1, x + n8])
                    if x > 0 and y > 0: cands.append(b[y - 1, x - 1])
                    ok = any(q["pred_mode"] == 0 and key(q) == key(c) for q in cands) or key(c)[:4] == (0, 0, 0, 0)
                    m += ok
            print(f"      skip pass: {n} union CUs, {m} of them with a final-field neighbour (A1/B1/B0/B2) of the same motion or zero", flush=True)

    def set_qp(self, qp, lambda_q4):
        self.cfg.qp, self.cfg.lambda_q4 = qp, lambda_q4

    def set_qp_map(self, qp_map: "np.ndarray | None"):
        """one QP per CTU (raster) for the pictures coded from here on (None: the slice QP everywhere); the oracle keeps the pointer: the array is held here"""
        self._qp_map = None if qp_map is None else np.ascontiguousarray(qp_map, dtype=np.int8)
        self.o.kso_set_qp_map(ptr(self._qp_map) if self._qp_map is not None else None)

    def effective_qp(self) -> np.ndarray:
        """QpY of every 8x8 block as the decoder derives it from the last coded picture's CU map"""
        eff = np.zeros((self.cfg.height // 8) * (self.cfg.width // 8), np.uint8)
        self.o.kso_effective_qp(C.byref(self.cfg), ptr(self.cu8), ptr(eff))
        return eff

    def load(self, pic: HostPic, i420: np.ndarray):
        self.o.kso_load_i420(C.byref(self.cfg), ptr(np.ascontiguousarray(i420)), pic.c())

    def store(self, pic: HostPic) -> np.ndarray:
        out = np.zeros(self.cfg.width * self.cfg.height * 3 // 2, np.uint8)
        self.o.kso_store_i420(C.byref(self.cfg), pic.c(), ptr(out))
        return out

    def search(self, ref_c: OPic, prev: "np.ndarray | None", pu: np.ndarray) -> None:
        """stage A (+ A2): the integer search of self.src in one reference picture, then cfg.propagate rounds of vector propagation between neighbouring PUs"""
        o, cfg = self.o, C.byref(self.cfg)
        if not self.cfg.propagate:
            o.kso_me_integer(cfg, self.src.c(), ref_c, ptr(prev) if prev is not None else None, ptr(pu))
            return
        off = np.zeros(2 * self.nctu, np.int16)
        o.kso_me_integer_ex(cfg, self.src.c(), ref_c, ptr(prev) if prev is not None else None, ptr(pu), ptr(off))
        self.pu_search = pu.copy()                # before the propagation (stage tests)
        for _ in range(self.cfg.propagate):
            out = np.zeros_like(pu)
            o.kso_me_propagate(cfg, self.src.c(), ref_c, ptr(off), ptr(pu), ptr(out))
            pu[:] = out

    def encode(self, i420: np.ndarray, kind: str, ref0: "HostPic | None" = None, ref1: "HostPic | None" = None) -> "HostPic":
        """one picture through all stages; kind 'I' (flat key picture), 'P' (ref0) or 'B' (ref0 = L0 past, ref1 = L1 future);
        returns the reconstructed padded picture.  Intermediate results stay in self.* for stage-by-stage comparison."""
        o, cfg = self.o, C.byref(self.cfg)
        self.load(self.src, i420)
        null = OPic(None, None, None)
        r0 = ref0.c() if ref0 is not None else null
        r1 = ref1.c() if ref1 is not None else null
        p1 = None
        if kind == "I":
            if self.intra:
                o.kso_intra_decide(cfg, self.src.c(), ptr(self.cu8))
            else:
                o.kso_cu_flat_intra(cfg, ptr(self.cu8))
            self.have_prev = False
        else:
            o.kso_ref_planes(cfg, r0, ptr(self.planes))
            self.search(r0, self.prev_pu if (self.have_prev and kind == "P") else None, self.pu)
            self.pu_int = self.pu.copy()
            if self.cfg.subme:
                o.kso_me_subpel(cfg, self.src.c(), ptr(self.planes), ptr(self.pu))
            ii = self.cfg.intra_inter
            if ii and not hasattr(self, "icost"):
                self.icost, self.imode = np.zeros(self.nctu * 85, np.uint32), np.zeros(self.nctu * 85, np.uint8)
            if ii and kind == "P":                       # intra candidates of this picture: cost + mode of every block from source neighbours, gated by the inter costs
                o.kso_intra_candidates(cfg, self.src.c(), ptr(self.pu), ptr(self.icost), ptr(self.imode))
            if kind == "P":
                if self.cfg.part:                        # -part 1: the CU decision also prices the 2NxN / Nx2N halves (needs the pixels)
                    o.kso_cu_decide_part(cfg, self.src.c(), ptr(self.planes), ptr(self.pu), ptr(self.icost) if ii else None, ptr(self.imode) if ii else None, ptr(self.cu8))
                elif ii:
                    o.kso_cu_decide_ii(cfg, ptr(self.pu), ptr(self.icost), ptr(self.imode), ptr(self.cu8))
                else:
                    o.kso_cu_decide(cfg, ptr(self.pu), ptr(self.cu8))
                if self.cfg.merge:
                    tmp = self.cu8.copy()
                    o.kso_merge_pass(cfg, self.src.c(), ptr(self.planes), None, ptr(self.pu), None, ptr(tmp), ptr(self.cu8))
            else:
                if not hasattr(self, "planes1"):
                    self.planes1 = np.zeros(16 * self.geom.bytes_y, np.uint8)
                    self.pu1 = np.zeros(self.nctu * 85, PU)
                    self.pub = np.zeros(self.nctu * 85, PU_B)
                o.kso_ref_planes(cfg, r1, ptr(self.planes1))
                self.search(r1, None, self.pu1)
                self.pu1_int = self.pu1.copy()
                if self.cfg.subme:
                    o.kso_me_subpel(cfg, self.src.c(), ptr(self.planes1), ptr(self.pu1))
                o.kso_bi_decide(cfg, self.src.c(), ptr(self.planes), ptr(self.planes1), ptr(self.pu), ptr(self.pu1), ptr(self.pub))
                if ii:
                    o.kso_intra_candidates(cfg, self.src.c(), ptr(self.pub), ptr(self.icost), ptr(self.imode))
                if self.cfg.part:                        # -part 1 in B pictures: the halves take the motions of the CU and of its quarters
                    o.kso_cu_decide_part_b(cfg, self.src.c(), ptr(self.planes), ptr(self.planes1), ptr(self.pu), ptr(self.pu1), ptr(self.pub),
                                           ptr(self.icost) if ii else None, ptr(self.imode) if ii else None, ptr(self.cu8))
                elif ii:
                    o.kso_cu_decide_b_ii(cfg, ptr(self.pub), ptr(self.icost), ptr(self.imode), ptr(self.cu8))
                else:
                    o.kso_cu_decide_b(cfg, ptr(self.pub), ptr(self.cu8))
                if self.cfg.bi_refine == 2:              # the joint refinement for the CUs the decision chose (round 5)
                    o.kso_bi_refine_chosen(cfg, self.src.c(), ptr(self.planes), ptr(self.planes1), ptr(self.pu), ptr(self.pu1), ptr(self.pub), ptr(self.cu8))
                if self.cfg.merge:
                    for _ in range(int(os.environ.get("RD_MERGE_ROUNDS", "1"))):     # (experiment hook of tools/rd_eval.py; the pipeline runs one round)
                        tmp = self.cu8.copy()
                        o.kso_merge_pass(cfg, self.src.c(), ptr(self.planes), ptr(self.planes1), None, ptr(self.pub), ptr(tmp), ptr(self.cu8))
                p1 = ptr(self.planes1)
        if kind == "I" and self.intra:
            o.kso_intra_reconstruct(cfg, self.src.c(), ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
        else:
            o.kso_reconstruct(cfg, self.src.c(), r0, ptr(self.planes), r1, p1, ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]),
                              self.rec.c())
            if (self.skip_rd and kind == "B") or (self.skip_rd >= 2 and kind == "P"):     # 1: B pictures only (where the pass pays: P pictures gain nothing measurable), 2: P pictures too
                self.skip_pass(r0, r1)
            if self.cfg.intra_inter:
                o.kso_intra_inter_reconstruct(cfg, self.src.c(), ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
        self.rec_pre = [self.rec.y.copy(), self.rec.u.copy(), self.rec.v.copy()]
        if self.cfg.deblock:
            o.kso_deblock(cfg, ptr(self.cu8), self.rec.c())
        out = HostPic(self.geom)
        o.kso_sao(cfg, self.src.c(), self.rec.c(), ptr(self.sao), out.c())
        if kind == "P":
            self.prev_pu, self.pu = self.pu, self.prev_pu
            self.have_prev = True
        return out

    def encode_b_mref(self, i420: np.ndarray, refs0: "list[HostPic]", refs1: "list[HostPic]") -> "HostPic":
        """B picture with several pictures per list (round 5; list 0 = past pictures nearest first, list 1 = future ones nearest first, no picture in both): one search per
        picture, per PU and list the cheapest picture (kso_ref_pick), then the stages of encode('B') taking every block's pictures from its record (kso_set_mref)"""
        o, cfg = self.o, C.byref(self.cfg)
        if len(refs0) == 1 and len(refs1) == 1:
            return self.encode(i420, "B", refs0[0], refs1[0])
        self.load(self.src, i420)
        lists = (refs0, refs1)
        if not hasattr(self, "mr_planes"):
            self.mr_planes = [[np.zeros(16 * self.geom.bytes_y, np.uint8) for _ in range(4)] for _ in range(2)]
            self.mr_pu = [[np.zeros(self.nctu * 85, PU) for _ in range(4)] for _ in range(2)]
            self.mr_idx = [np.zeros(self.nctu * 85, np.uint8) for _ in range(2)]
        if not hasattr(self, "pub"):
            self.pu1 = np.zeros(self.nctu * 85, PU)
            self.pub = np.zeros(self.nctu * 85, PU_B)
        if not hasattr(self, "pu1"):
            self.pu1 = np.zeros(self.nctu * 85, PU)
        best = (self.pu, self.pu1)
        for L in range(2):
            for i, r in enumerate(lists[L]):
                o.kso_ref_planes(cfg, r.c(), ptr(self.mr_planes[L][i]))
                self.search(r.c(), None, self.mr_pu[L][i])
                if self.cfg.subme:
                    o.kso_me_subpel(cfg, self.src.c(), ptr(self.mr_planes[L][i]), ptr(self.mr_pu[L][i]))
            n = len(lists[L])
            arr = (C.c_void_p * n)(*[p.ctypes.data for p in self.mr_pu[L][:n]])
            o.kso_ref_pick(cfg, C.c_int(n), arr, ptr(best[L]), ptr(self.mr_idx[L]))
        mr = OMref()
        mr.n0, mr.n1 = len(refs0), len(refs1)
        for i in range(4):
            mr.planes0[i] = self.mr_planes[0][min(i, mr.n0 - 1)].ctypes.data; mr.planes1[i] = self.mr_planes[1][min(i, mr.n1 - 1)].ctypes.data
            mr.pic0[i] = refs0[min(i, mr.n0 - 1)].c(); mr.pic1[i] = refs1[min(i, mr.n1 - 1)].c()
        mr.idx0, mr.idx1 = self.mr_idx[0].ctypes.data, self.mr_idx[1].ctypes.data
        o.kso_set_mref(C.byref(mr))
        try:
            pl0, pl1 = ptr(self.mr_planes[0][0]), ptr(self.mr_planes[1][0])
            o.kso_bi_decide(cfg, self.src.c(), pl0, pl1, ptr(self.pu), ptr(self.pu1), ptr(self.pub))
            ii = self.cfg.intra_inter
            if ii and not hasattr(self, "icost"):
                self.icost, self.imode = np.zeros(self.nctu * 85, np.uint32), np.zeros(self.nctu * 85, np.uint8)
            if ii:
                o.kso_intra_candidates(cfg, self.src.c(), ptr(self.pub), ptr(self.icost), ptr(self.imode))
            if self.cfg.part:
                o.kso_cu_decide_part_b(cfg, self.src.c(), pl0, pl1, ptr(self.pu), ptr(self.pu1), ptr(self.pub), ptr(self.icost) if ii else None, ptr(self.imode) if ii else None, ptr(self.cu8))
            elif ii:
                o.kso_cu_decide_b_ii(cfg, ptr(self.pub), ptr(self.icost), ptr(self.imode), ptr(self.cu8))
            else:
                o.kso_cu_decide_b(cfg, ptr(self.pub), ptr(self.cu8))
            if self.cfg.bi_refine == 2:
                o.kso_bi_refine_chosen(cfg, self.src.c(), pl0, pl1, ptr(self.pu), ptr(self.pu1), ptr(self.pub), ptr(self.cu8))
            if self.cfg.merge:
                tmp = self.cu8.copy()
                o.kso_merge_pass(cfg, self.src.c(), pl0, pl1, None, ptr(self.pub), ptr(tmp), ptr(self.cu8))
            ref_arr = (OPic * mr.n0)(*[r.c() for r in refs0])
            pl_arr = (C.c_void_p * mr.n0)(*[p.ctypes.data for p in self.mr_planes[0][:mr.n0]])
            o.kso_reconstruct_mref(cfg, self.src.c(), C.c_int(mr.n0), ref_arr, pl_arr, ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
            if self.skip_rd:
                self.skip_pass(refs0[0].c(), refs1[0].c())
            if self.cfg.intra_inter:
                o.kso_intra_inter_reconstruct(cfg, self.src.c(), ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
        finally:
            o.kso_set_mref(None)
        self.rec_pre = [self.rec.y.copy(), self.rec.u.copy(), self.rec.v.copy()]
        if self.cfg.deblock:
            o.kso_deblock(cfg, ptr(self.cu8), self.rec.c())
        out = HostPic(self.geom)
        o.kso_sao(cfg, self.src.c(), self.rec.c(), ptr(self.sao), out.c())
        return out

    def encode_mref(self, i420: np.ndarray, refs: "list[HostPic]") -> "HostPic":
        """P picture searching several list-0 pictures (nearest first): one search per picture, per-PU choice, CU tree, reconstruction"""
        o, cfg = self.o, C.byref(self.cfg)
        if len(refs) == 1:
            return self.encode(i420, "P", refs[0])
        self.load(self.src, i420)
        n = len(refs)
        if not hasattr(self, "planes_x"):
            self.planes_x = [np.zeros(16 * self.geom.bytes_y, np.uint8) for _ in range(3)]
            self.pu_x = [np.zeros(self.nctu * 85, PU) for _ in range(3)]
        if not hasattr(self, "pub"):
            self.pub = np.zeros(self.nctu * 85, PU_B)
        planes = [self.planes] + self.planes_x[:n - 1]
        pus = [self.pu] + self.pu_x[:n - 1]
        for i, r in enumerate(refs):
            o.kso_ref_planes(cfg, r.c(), ptr(planes[i]))
            self.search(r.c(), self.prev_pu if (i == 0 and self.have_prev) else None, pus[i])
            if self.cfg.subme:
                o.kso_me_subpel(cfg, self.src.c(), ptr(planes[i]), ptr(pus[i]))
        pu_arr = (C.c_void_p * n)(*[p.ctypes.data for p in pus])
        o.kso_ref_decide(cfg, C.c_int(n), pu_arr, ptr(self.pub))
        # round 6 (-ref0: the anchors of the pyramid GOPs search several past anchors): the two-list records go through the stages a one-reference P picture has - intra
        # candidates against them (cfg.intra_inter), the CU tree, the merge pass on the records' pictures (cfg.merge; a context without list-1 pictures), the intra CUs' pass
        ii = self.cfg.intra_inter
        if ii and not hasattr(self, "icost"):
            self.icost, self.imode = np.zeros(self.nctu * 85, np.uint32), np.zeros(self.nctu * 85, np.uint8)
        if ii:
            o.kso_intra_candidates(cfg, self.src.c(), ptr(self.pub), ptr(self.icost), ptr(self.imode))
            o.kso_cu_decide_b_ii(cfg, ptr(self.pub), ptr(self.icost), ptr(self.imode), ptr(self.cu8))
        else:
            o.kso_cu_decide_b(cfg, ptr(self.pub), ptr(self.cu8))
        if self.cfg.merge:
            mr = OMref()
            mr.n0, mr.n1 = n, 0
            for i in range(4):
                mr.planes0[i] = planes[min(i, n - 1)].ctypes.data; mr.planes1[i] = planes[0].ctypes.data
                mr.pic0[i] = refs[min(i, n - 1)].c(); mr.pic1[i] = refs[0].c()
            o.kso_set_mref(C.byref(mr))
            try:
                tmp = self.cu8.copy()
                o.kso_merge_pass(cfg, self.src.c(), ptr(planes[0]), None, None, ptr(self.pub), ptr(tmp), ptr(self.cu8))
            finally:
                o.kso_set_mref(None)
        ref_arr = (OPic * n)(*[r.c() for r in refs])
        pl_arr = (C.c_void_p * n)(*[p.ctypes.data for p in planes])
        o.kso_reconstruct_mref(cfg, self.src.c(), C.c_int(n), ref_arr, pl_arr, ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
        if self.skip_rd >= 2:
            mr = OMref()
            mr.n0, mr.n1 = n, 0
            for i in range(4):
                mr.planes0[i] = planes[min(i, n - 1)].ctypes.data; mr.planes1[i] = planes[0].ctypes.data
                mr.pic0[i] = refs[min(i, n - 1)].c(); mr.pic1[i] = refs[0].c()
            o.kso_set_mref(C.byref(mr))
            try:
                self.skip_pass(refs[0].c(), OPic(None, None, None))
            finally:
                o.kso_set_mref(None)
        if ii:
            o.kso_intra_inter_reconstruct(cfg, self.src.c(), ptr(self.cu8), ptr(self.lvl[0]), ptr(self.lvl[1]), ptr(self.lvl[2]), self.rec.c())
        self.rec_pre = [self.rec.y.copy(), self.rec.u.copy(), self.rec.v.copy()]
        if self.cfg.deblock:
            o.kso_deblock(cfg, ptr(self.cu8), self.rec.c())
        out = HostPic(self.geom)
        o.kso_sao(cfg, self.src.c(), self.rec.c(), ptr(self.sao), out.c())
        self.prev_pu, self.pu = self.pu, self.prev_pu
        self.have_prev = True
        return out

    def encode_picture(self, i420: np.ndarray, is_key: bool) -> np.ndarray:
        """IPPP convenience: self.ref is replaced by the new reconstructed picture; returns recon I420"""
        self.ref = self.encode(i420, "I" if is_key else "P", None if is_key else self.ref)
        return self.store(self.ref)
