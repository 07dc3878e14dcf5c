"""ctypes binding of include/ks265_hip.h.  Device buffers are torch CUDA(HIP) tensors; only their
data_ptr() crosses the C ABI."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libks265hip.so")

# descriptor layouts of include/ks265_hip.h
BLK = np.dtype([("a_off", "<i4"), ("b_off", "<i4"), ("w", "<i4"), ("h", "<i4")])
BLK3 = np.dtype([("a_off", "<i4"), ("b_off", "<i4", 3), ("w", "<i4"), ("h", "<i4")])
EDGE = np.dtype([("pix_off", "<i4"), ("beta", "<i2"), ("tc", "<i2"), ("length", "<i2"), ("dir", "u1"), ("flags", "u1")])
SAO_RECT = np.dtype([("org_off", "<i4"), ("rec_off", "<i4"), ("w", "<i4"), ("h", "<i4")])
INTRA_BLK = np.dtype([("ref_off", "<i4"), ("dst_off", "<i4"), ("dst_stride", "<i2"), ("mode", "u1"), ("log2", "u1"), ("edge_filter", "u1"), ("rsv", "u1", (3,))])
INTRA_REF = np.dtype([("src_off", "<i4"), ("dst_off", "<i4"), ("size", "<i4"), ("strong_enabled", "<i4")])
PU = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("mvpx", "<i2"), ("mvpy", "<i2"), ("cost", "<u4"), ("dist", "<u4")])
CU8 = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2"), ("log2_cu", "u1"), ("cbf", "u1"), ("pred_mode", "u1"), ("inter_dir", "u1")])
PU_B = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("mv1x", "<i2"), ("mv1y", "<i2"), ("cost", "<u4"), ("inter_dir", "<u4")])
RDOQ_TU = np.dtype([("off", "<i4"), ("tab", "<i4"), ("dq", "<i4"), ("last_pos", "<i4"), ("lam", "<i8"), ("lam_sdh", "<i8"), ("log2", "i1"), ("scan_idx", "i1"), ("comp", "i1"), ("per", "i1"),
                    ("tu5", "i1"), ("flag_a4c0", "i1"), ("sdh", "i1"), ("rsv", "i1")])      # ks265_rdoq_tu
SAO_PARAM = np.dtype([("type", "i1"), ("band", "i1"), ("offset", "i1", 4), ("rsv", "i1", 2)])


class FrameCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "qp", "lambda_q4", "me_range", "me_method", "subme", "deblock", "sao",
                                          "beta_offset_div2", "tc_offset_div2", "bframes", "refs", "me_hex_thr", "sdh", "pre_search", "merge", "bi_refine", "decimate", "rdo", "intra_inter", "propagate", "sub_satd", "sub_thr", "sub_flat", "sub_cap", "sub_cap_step", "sub_diag_fast", "part", "tu_inter", "skip_rd")]


class FrameGeom(C.Structure):
    _fields_ = [("pad_y", C.c_int32), ("pad_c", C.c_int32), ("stride_y", C.c_int32), ("stride_c", C.c_int32),
                ("rows_y", C.c_int32), ("rows_c", C.c_int32), ("bytes_y", C.c_int64), ("bytes_c", C.c_int64),
                ("ctu_cols", C.c_int32), ("ctu_rows", C.c_int32), ("pu_per_ctu", C.c_int32), ("bytes_pu", C.c_int64),
                ("bytes_cu8", C.c_int64), ("bytes_sao", C.c_int64)]


class Pic(C.Structure):
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p)]


class Ks265Error(RuntimeError):
    pass


_lib = None

# every symbol include/ks265_hip.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "ks265_create", "ks265_create_prio", "ks265_destroy", "ks265_set_stream", "ks265_synchronize", "ks265_take_device_error", "ks265_last_error", "ks265_version",
    "ks265_timer_start", "ks265_timer_stop_ms", "ks265_marker", "ks265_debug_set",
    "ks265_dev_malloc", "ks265_dev_free", "ks265_host_malloc", "ks265_host_free", "ks265_memcpy_h2d_async", "ks265_memcpy_d2h_async", "ks265_memcpy_d2d_async", "ks265_copy_out_async", "ks265_memset_async",
    "ks265_event_create", "ks265_event_record", "ks265_event_wait", "ks265_event_query", "ks265_stream_wait_event", "ks265_event_destroy",
    "ks265_sad_batch", "ks265_sad4_batch", "ks265_sad3_batch", "ks265_sad4blk_8x8_batch", "ks265_sse_batch", "ks265_had_batch",
    "ks265_residual_batch", "ks265_fwd_transform_batch", "ks265_quant_batch", "ks265_sign_hiding_batch", "ks265_rdoq_batch", "ks265_dequant_batch", "ks265_dequant_rect_batch", "ks265_inv_transform_batch",
    "ks265_edge_filter_luma_batch", "ks265_edge_filter_chroma_batch", "ks265_interp_rect",
    "ks265_sao_apply_bo_rect", "ks265_sao_apply_eo_rect", "ks265_sao_stats_batch", "ks265_intra_pred_batch", "ks265_intra_filter_ref_batch",
    "ks265_downsample_rect", "ks265_downsample_from_host", "ks265_weight_bi_sad_batch", "ks265_ac_energy_batch", "ks265_ac_energy_map",
    "ks265_frame_reset_prediction", "ks265_frame_records_layout", "ks265_frame_pack_records", "ks265_frame_compact_layout", "ks265_frame_pack_compact", "ks265_frame_pack_compact_on", "ks265_frame_adapt_quant", "ks265_aq_ctu_map", "ks265_cutree_propagate", "ks265_calc_frame_cost", "ks265_calc_frame_cost_workspace", "ks265_cutree_finish", "ks265_host_register", "ks265_memcpy_h2d_sync", "ks265_host_unregister", "ks265_pad_plane", "ks265_fill_u16", "ks265_qoff_ctu_map", "ks265_frame_set_records_fence", "ks265_load_i420_on", "ks265_sse_picture_on", "ks265_copy_out_compact_async", "ks265_copy_out_compact_dma_async", "ks265_frame_geometry", "ks265_frame_create", "ks265_frame_destroy", "ks265_frame_set_qp", "ks265_frame_set_picture_tools", "ks265_frame_set_qp_map", "ks265_frame_set_rdoq", "ks265_pad_picture",
    "ks265_load_i420", "ks265_store_i420", "ks265_presearch", "ks265_me_integer", "ks265_me_propagate", "ks265_me_subpel", "ks265_cu_decide_part", "ks265_cu_decide_part_b", "ks265_merge_pass", "ks265_skip_pass", "ks265_cu_decide",
    "ks265_cu_flat_intra", "ks265_intra_decide", "ks265_intra_decide_ex", "ks265_lookahead_reduce", "ks265_lookahead_picture", "ks265_lookahead_inter", "ks265_intra_reconstruct", "ks265_reconstruct", "ks265_reconstruct_b", "ks265_bi_decide", "ks265_bi_refine_chosen", "ks265_bi_full_batch", "ks265_capture_begin", "ks265_capture_end", "ks265_graph_launch", "ks265_graph_destroy", "ks265_frame_p_state", "ks265_frame_p_advance", "ks265_frame_p_restore", "ks265_cu_decide_b", "ks265_deblock", "ks265_sao",
    "ks265_encode_picture", "ks265_encode_picture_b", "ks265_encode_picture_mref", "ks265_encode_picture_b_mref", "ks265_ref_pick", "ks265_ref_decide", "ks265_reconstruct_mref",
    "ks265_intra_candidates", "ks265_cu_decide_ii", "ks265_cu_decide_b_ii", "ks265_intra_inter_reconstruct", "ks265_frame_set_profiling", "ks265_frame_stage_ms", "ks265_frame_me_int_ms", "ks265_frame_levels", "ks265_frame_pu", "ks265_frame_cu8", "ks265_frame_ibest", "ks265_frame_sao", "ks265_sse_picture",
]


def load_library() -> C.CDLL:
    """Load libks265hip.so; fail loudly if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Ks265Error(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        _lib = C.CDLL(LIB_PATH)
        _lib.ks265_last_error.restype = C.c_char_p
        _lib.ks265_version.restype = C.c_char_p
        for n in ("ks265_frame_levels", "ks265_frame_pu", "ks265_frame_cu8", "ks265_frame_ibest", "ks265_frame_sao"):
            if hasattr(_lib, n):
                getattr(_lib, n).restype = C.c_void_p
    return _lib


def _p(t) -> C.c_void_p:
    """device pointer of a torch tensor (or None)"""
    return C.c_void_p(0 if t is None else t.data_ptr())


class KsContext:
    """Owns a ks265_ctx bound to torch's current HIP stream on `device`."""

    def __init__(self, device: int = 0):
        import torch

        self.lib = load_library()
        if not torch.cuda.is_available():
            raise Ks265Error("no HIP device visible: the ks265 pixel path has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        self._chk(self.lib.ks265_create(C.byref(h), C.c_int(device)), None)
        self.h = h
        torch.cuda.set_device(device)
        self._chk(self.lib.ks265_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))

    def _chk(self, rc: int, h="self"):
        if rc != 0:
            msg = self.lib.ks265_last_error(self.h).decode() if h == "self" and getattr(self, "h", None) else ""
            raise Ks265Error(f"ks265 call failed rc={rc} {msg}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.ks265_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def dev(self, arr: np.ndarray):
        """host numpy (any dtype incl. structured) -> device byte-exact tensor of the same dtype family"""
        t = self.torch
        a = np.ascontiguousarray(arr)
        raw = t.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self.device)
        return raw

    def zeros(self, nbytes: int):
        return self.torch.zeros(int(nbytes), dtype=self.torch.uint8, device=self.device)

    def host(self, t, dtype, shape=None) -> np.ndarray:
        a = t.cpu().numpy().view(dtype)
        return a.reshape(shape) if shape is not None else a

    def sync(self):
        self._chk(self.lib.ks265_synchronize(self.h))

    def debug_set(self, what: int, value: int):
        self._chk(self.lib.ks265_debug_set(self.h, C.c_int(what), C.c_int(value)))

    def marker(self, ident: int = 0):
        self._chk(self.lib.ks265_marker(self.h, C.c_int(ident)))

    # ---- section 2: batched operator tables (names follow the reference tables)
    def _dist(self, fn, a, sa, b, sb, blks: np.ndarray, k: int) -> np.ndarray:
        n = len(blks)
        out = self.zeros(4 * n * k)
        self._chk(fn(self.h, _p(a), C.c_int(sa), _p(b), C.c_int(sb), _p(self.dev(blks)), C.c_int(n), _p(out)))
        return self.host(out, np.uint32, (n, k) if k > 1 else (n,))

    def sad(self, a, sa, b, sb, blks): return self._dist(self.lib.ks265_sad_batch, a, sa, b, sb, blks, 1)
    def sse(self, a, sa, b, sb, blks): return self._dist(self.lib.ks265_sse_batch, a, sa, b, sb, blks, 1)
    def had(self, a, sa, b, sb, blks): return self._dist(self.lib.ks265_had_batch, a, sa, b, sb, blks, 1)
    def sad4(self, a, sa, b, sb, blks): return self._dist(self.lib.ks265_sad4_batch, a, sa, b, sb, blks, 4)
    def sad3(self, a, sa, b, sb, blks): return self._dist(self.lib.ks265_sad3_batch, a, sa, b, sb, blks, 3)
    def sad4blk_8x8(self, a, sa, b, sb, blks): return self._dist(self.lib.ks265_sad4blk_8x8_batch, a, sa, b, sb, blks, 4)

    def residual(self, org, so, pred, sp, blks: np.ndarray) -> np.ndarray:
        total = int((blks["w"].astype(np.int64) ** 2).sum())
        out = self.zeros(2 * total)
        self._chk(self.lib.ks265_residual_batch(self.h, _p(org), C.c_int(so), _p(pred), C.c_int(sp), _p(self.dev(blks)), C.c_int(len(blks)), _p(out)))
        return self.host(out, np.int16)

    def fwd_transform(self, idx: int, src: np.ndarray) -> np.ndarray:
        nblk = src.shape[0]
        d, out = self.dev(src.astype(np.int16)), self.zeros(src.size * 2)
        self._chk(self.lib.ks265_fwd_transform_batch(self.h, C.c_int(idx), _p(d), _p(out), C.c_int(nblk)))
        return self.host(out, np.int16, src.shape)

    def inv_transform(self, idx: int, coef: np.ndarray, pred: np.ndarray) -> np.ndarray:
        nblk = coef.shape[0]
        dc, dp, out = self.dev(coef.astype(np.int16)), self.dev(pred.astype(np.uint8)), self.zeros(pred.size)
        self._chk(self.lib.ks265_inv_transform_batch(self.h, C.c_int(idx), _p(dc), _p(dp), _p(out), C.c_int(nblk)))
        return self.host(out, np.uint8, pred.shape)

    def quant(self, n: int, coef: np.ndarray, scale: int, off: int, qbits: int):
        nblk = coef.shape[0]
        dc = self.dev(coef.astype(np.int16))
        lvl, du, nz = self.zeros(coef.size * 2), self.zeros(coef.size * 2), self.zeros(4 * nblk)
        self._chk(self.lib.ks265_quant_batch(self.h, C.c_int(n), _p(dc), _p(lvl), _p(du), _p(nz), C.c_int(scale), C.c_int(off), C.c_int(qbits), C.c_int(nblk)))
        return self.host(lvl, np.int16, coef.shape), self.host(du, np.int16, coef.shape), self.host(nz, np.int32)

    def sign_hiding(self, n: int, scan_idx: int, lvl: np.ndarray, coef: np.ndarray, delta_u: np.ndarray) -> np.ndarray:
        dl, dc, du = self.dev(lvl.astype(np.int16)), self.dev(coef.astype(np.int16)), self.dev(delta_u.astype(np.int16))
        self._chk(self.lib.ks265_sign_hiding_batch(self.h, C.c_int(n), C.c_int(scan_idx), _p(dl), _p(dc), _p(du), C.c_int(lvl.shape[0])))
        return self.host(dl, np.int16, lvl.shape)

    def rdoq(self, tus: np.ndarray, lvl: np.ndarray, coef: np.ndarray, tables: np.ndarray, sigmask: np.ndarray):
        """ks265_rdoq_batch (rdoQuant enc@0x4aac50): tus = RDOQ_TU records; lvl / coef flat int16; tables [ntab, 180] int32; sigmask [n, 64] uint16.
        Returns (levels, sigmask, out [n, 2] = (non-zero levels, last scan position), hidden [n])"""
        n = len(tus)
        dl, dc, dt, dm = self.dev(lvl.astype(np.int16)), self.dev(coef.astype(np.int16)), self.dev(np.ascontiguousarray(tables, np.int32)), self.dev(np.ascontiguousarray(sigmask, np.uint16))
        out, hid = self.zeros(8 * n), self.zeros(8 * n)
        self._chk(self.lib.ks265_rdoq_batch(self.h, _p(self.dev(tus)), C.c_int(n), _p(dl), _p(dc), _p(dt), _p(dm), _p(out), _p(hid)))
        return self.host(dl, np.int16), self.host(dm, np.uint16, (n, 64)), self.host(out, np.int32, (n, 2)), self.host(hid, np.uint64)

    def dequant(self, n: int, lvl: np.ndarray, scale: int, add: int, shift: int) -> np.ndarray:
        nblk = lvl.shape[0]
        dl, out = self.dev(lvl.astype(np.int16)), self.zeros(lvl.size * 2)
        self._chk(self.lib.ks265_dequant_batch(self.h, C.c_int(n), _p(dl), _p(out), C.c_int(scale), C.c_int(add), C.c_int(shift), C.c_int(nblk)))
        return self.host(out, np.int16, lvl.shape)

    def dequant_rect(self, n: int, lvl: np.ndarray, coef_init: np.ndarray, scale: int, add: int, shift: int, last_x: int, last_y: int) -> np.ndarray:
        dl, out = self.dev(lvl.astype(np.int16)), self.dev(coef_init.astype(np.int16))
        self._chk(self.lib.ks265_dequant_rect_batch(self.h, C.c_int(n), C.c_int(n), _p(dl), _p(out), C.c_int(scale), C.c_int(add), C.c_int(shift),
                                                    C.c_int(last_x), C.c_int(last_y), C.c_int(lvl.shape[0])))
        return self.host(out, np.int16, lvl.shape)

    def edge_filter(self, plane, stride: int, edges: np.ndarray, chroma: bool = False):
        fn = self.lib.ks265_edge_filter_chroma_batch if chroma else self.lib.ks265_edge_filter_luma_batch
        self._chk(fn(self.h, _p(plane), C.c_int(stride), _p(self.dev(edges)), C.c_int(len(edges))))

    def interp_rect(self, kind: int, dst, ds: int, src, src_byte_off: int, ss: int, w: int, h: int, frac: int):
        self._chk(self.lib.ks265_interp_rect(self.h, C.c_int(kind), _p(dst), C.c_int(ds), C.c_void_p(src.data_ptr() + src_byte_off),
                                             C.c_int(ss), C.c_int(w), C.c_int(h), C.c_int(frac)))

    def sao_apply_bo(self, offsets: np.ndarray, rec, stride: int, h: int, w: int, band: int):
        o = (C.c_int8 * 4)(*[int(x) for x in offsets])
        self._chk(self.lib.ks265_sao_apply_bo_rect(self.h, o, _p(rec), C.c_int(stride), C.c_int(h), C.c_int(w), C.c_int(band)))

    def sao_apply_eo(self, cls: int, offsets: np.ndarray, src, dst, byte_off: int, stride: int, h: int, w: int):
        o = (C.c_int8 * 5)(*[int(x) for x in offsets])
        self._chk(self.lib.ks265_sao_apply_eo_rect(self.h, C.c_int(cls), o, C.c_void_p(src.data_ptr() + byte_off),
                                                   C.c_void_p(dst.data_ptr() + byte_off), C.c_int(stride), C.c_int(h), C.c_int(w)))

    def sao_stats(self, org, os_: int, rec, rs: int, rects: np.ndarray, row_step: int) -> np.ndarray:
        out = self.zeros(4 * 96 * len(rects))
        self._chk(self.lib.ks265_sao_stats_batch(self.h, _p(org), C.c_int(os_), _p(rec), C.c_int(rs), _p(self.dev(rects)), C.c_int(len(rects)),
                                                 C.c_int(row_step), _p(out)))
        return self.host(out, np.int32, (len(rects), 96))

    def downsample(self, src, ss: int, w: int, h: int, ds: int):
        dst = self.zeros(ds * h)
        self._chk(self.lib.ks265_downsample_rect(self.h, _p(src), C.c_int(ss), _p(dst), C.c_int(ds), C.c_int(w), C.c_int(h)))
        return self.host(dst, np.uint8, (h, ds))

    def downsample_from_host(self, src: np.ndarray, ss: int, w: int, h: int, ds: int):
        """the same with the source in pinned host memory of the library (ks265_host_malloc): the kernel reads it over PCIe"""
        hp = C.c_void_p()
        self._chk(self.lib.ks265_host_malloc(self.h, C.byref(hp), C.c_size_t(src.size)))
        try:
            C.memmove(hp, src.ctypes.data, src.size)
            dst = self.zeros(ds * h)
            self._chk(self.lib.ks265_downsample_from_host(self.h, hp, C.c_int(ss), _p(dst), C.c_int(ds), C.c_int(w), C.c_int(h)))
            self.sync()
            return self.host(dst, np.uint8, (h, ds))
        finally:
            self._chk(self.lib.ks265_host_free(self.h, hp))

    def weight_bi_sad(self, org, so: int, r0, s0: int, r1, s1: int, blks: np.ndarray) -> np.ndarray:
        out = self.zeros(4 * len(blks))
        self._chk(self.lib.ks265_weight_bi_sad_batch(self.h, _p(org), C.c_int(so), _p(r0), C.c_int(s0), _p(r1), C.c_int(s1), _p(self.dev(blks)), C.c_int(len(blks)), _p(out)))
        return self.host(out, np.uint32)

    def bi_full(self, use_had: int, org, so: int, ref, sr: int, blks: np.ndarray, mvcost: np.ndarray) -> np.ndarray:
        out = self.zeros(8 * len(blks))
        d_blks, d_mvc = self.dev(blks), self.dev(np.ascontiguousarray(mvcost, np.uint16))      # both alive across the call (a temporary's block would be reused)
        self._chk(self.lib.ks265_bi_full_batch(self.h, C.c_int(int(use_had)), _p(org), C.c_int(so), _p(ref), C.c_int(sr), _p(d_blks), _p(d_mvc), C.c_int(len(blks)), _p(out)))
        return self.host(out, np.uint32, (len(blks), 2))

    def ac_energy(self, src, stride: int, log2: int, offs: np.ndarray) -> np.ndarray:
        out = self.zeros(4 * len(offs))
        self._chk(self.lib.ks265_ac_energy_batch(self.h, _p(src), C.c_int(stride), C.c_int(log2), _p(self.dev(np.asarray(offs, np.int32))), C.c_int(len(offs)), _p(out)))
        return self.host(out, np.uint32)

    def ac_energy_map(self, plane, stride: int, w: int, h: int, log2: int) -> np.ndarray:
        n = (w >> log2) * (h >> log2)
        out = self.zeros(4 * n)
        self._chk(self.lib.ks265_ac_energy_map(self.h, _p(plane), C.c_int(stride), C.c_int(w), C.c_int(h), C.c_int(log2), _p(out)))
        return self.host(out, np.uint32, (h >> log2, w >> log2))

    def frame_adapt_quant(self, y, stride_y: int, u, v, stride_c: int, nx: int, ny: int, strength: float, count: int | None = None):
        """calcFrameAdaptQuant enc@0x4653c0: (QP offsets float64 [ny, nx], inverse qscale factors uint16 [ny, nx]) of a picture's 16x16 blocks; y / u / v: device planes"""
        n = nx * ny
        off, inv, scal = self.zeros(8 * n), self.zeros(2 * n), self.zeros(16)
        self._chk(self.lib.ks265_frame_adapt_quant(self.h, _p(y), C.c_int(stride_y), _p(u), _p(v), C.c_int(stride_c), C.c_int(nx), C.c_int(ny), C.c_int(count or n), C.c_double(strength),
                                                   _p(off), _p(inv), _p(scal)))
        return self.host(off, np.float64, (ny, nx)), self.host(inv, np.uint16, (ny, nx))

    def cutree_propagate(self, lg: int, nx: int, ny: int, intra, invq, own, inter, bits, mv0, mv1, ref0, ref1, acc):
        """cuTreePropagate enc@0x47d460 on device arrays (ref0 / ref1 updated in place; acc: 2 * nx * ny zeroed uint64, left zero)"""
        self._chk(self.lib.ks265_cutree_propagate(self.h, C.c_int(lg), C.c_int(nx), C.c_int(ny), _p(intra), _p(invq), _p(own), _p(inter), _p(bits), _p(mv0), _p(mv1), _p(ref0), _p(ref1), _p(acc)))

    def calc_frame_cost(self, prm: "CfcParams", cur, ref0, ref1, arr: dict, sums: np.ndarray):
        """calcFrameCost enc@0x4a7410 on device data.  cur / ref0 / ref1: (tensor, byte offset of sample (0, 0)) or None; arr: device tensors intra (u16), imode (u8), invq (u16),
        inter (u16), bits (u8), mv0 / c0 / mv1 / c1 (i32) - updated in place; sums: 11 int32 (ks265_cfc_sums) in, the updated words out"""
        at = lambda t: C.c_void_p(t[0].data_ptr() + int(t[1])) if t is not None else None
        self.lib.ks265_calc_frame_cost_workspace.restype = C.c_size_t
        ws = self.zeros(int(self.lib.ks265_calc_frame_cost_workspace(C.c_int(prm.nx), C.c_int(prm.ny))))
        ds = self.dev(np.ascontiguousarray(sums, np.int32))
        g = lambda k: _p(arr[k]) if arr.get(k) is not None else None
        self._chk(self.lib.ks265_calc_frame_cost(self.h, C.byref(prm), at(cur), at(ref0), at(ref1), g("intra"), g("imode"), g("invq"), g("inter"), g("bits"), g("mv0"), g("c0"), g("mv1"), g("c1"),
                                                 _p(ds), _p(ws)))
        return self.host(ds, np.int32)

    def cutree_finish(self, cnt: int, intra, invq, prop, aq_off, dbl: int, out):
        """the cuTree finish (enc@0x480964..0x480a54) on device arrays; out (float64) updated in place"""
        self._chk(self.lib.ks265_cutree_finish(self.h, C.c_int(cnt), _p(intra), _p(invq), _p(prop), _p(aq_off), C.c_int(dbl), _p(out)))

    def intra_pred(self, ref, dst, blks: np.ndarray):
        """g_IntraPredFunction: predict every described block from dev `ref` into dev `dst` (in place)"""
        self._chk(self.lib.ks265_intra_pred_batch(self.h, _p(ref), _p(dst), _p(self.dev(blks)), C.c_int(len(blks))))

    def intra_filter_ref(self, src, dst, refs: np.ndarray):
        self._chk(self.lib.ks265_intra_filter_ref_batch(self.h, _p(src), _p(dst), _p(self.dev(refs)), C.c_int(len(refs))))


class CfcParams(C.Structure):                 # ks265_cfc_params (include/ks265_hip.h)
    _fields_ = [*[(n, C.c_int32) for n in ("w", "h", "nx", "ny", "cnt", "stride", "d0", "d1", "flag", "slice_type", "merange", "lg", "zero_thr", "fast_intra", "scenecut", "preset", "p8",
                                             "aq", "b_intra", "f3a8", "f36c", "f538", "f3b4")], ("do_list", C.c_int32 * 2), ("intra_done", C.c_int32), ("lambda_tab", C.c_uint16 * 52)]


class DevPic:
    """A padded YUV 4:2:0 picture in HBM (three torch byte tensors)."""

    def __init__(self, ks: "KsContext", geom: FrameGeom):
        self.y, self.u, self.v = ks.zeros(geom.bytes_y), ks.zeros(geom.bytes_c), ks.zeros(geom.bytes_c)

    def c(self) -> Pic:
        return Pic(self.y.data_ptr(), self.u.data_ptr(), self.v.data_ptr())


class KsFrame:
    """Whole-frame stages of include/ks265_hip.h section 3 (one picture size, one stream)."""

    def __init__(self, ks: KsContext, width: int, height: int, qp: int, lambda_q4: int, me_range: int = 64, subme: int = 1,
                 deblock: int = 1, sao: int = 1, me_method: int = 0, bframes: int = 0, refs: int = 1, me_hex_thr: int = 0, sdh: int = 0, pre_search: int = 0, merge: int = 0, bi_refine: int = 0, decimate: int = 0, rdo: int = 0, intra_inter: int = 0, propagate: int = 0,
                 sub_satd: int = 0, sub_thr: int = 24, sub_flat: int = 8, sub_cap: int = 0, sub_cap_step: int = 0, sub_diag_fast: int = 0, part: int = 0, tu_inter: int = 0, skip_rd: int = 0):     # the sub-pel knobs of -preset slow (synth.SUBME_PRESET)
        self.ks, self.lib = ks, ks.lib
        self.cfg = FrameCfg(width, height, qp, lambda_q4, me_range, me_method, subme, deblock, sao, 0, 0, bframes, refs, me_hex_thr, sdh, pre_search, merge, bi_refine, decimate, rdo, intra_inter, propagate, sub_satd, sub_thr, sub_flat, sub_cap, sub_cap_step, sub_diag_fast, part, tu_inter, skip_rd)
        self.geom = FrameGeom()
        ks._chk(self.lib.ks265_frame_geometry(C.byref(self.cfg), C.byref(self.geom)))
        h = C.c_void_p()
        ks._chk(self.lib.ks265_frame_create(ks.h, C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.width, self.height = width, height
        self.nctu = self.geom.ctu_cols * self.geom.ctu_rows

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if getattr(self, "h", None):
            self.lib.ks265_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def new_pic(self) -> DevPic:
        return DevPic(self.ks, self.geom)

    def set_qp(self, qp: int, lambda_q4: int):
        self.cfg.qp, self.cfg.lambda_q4 = qp, lambda_q4
        self.ks._chk(self.lib.ks265_frame_set_qp(self.h, C.c_int(qp), C.c_int(lambda_q4)))

    def set_picture_tools(self, intra_inter: int = -1, bi_refine: int = -1, sao: int = -1, me_method: int = -1):
        """tools of the pictures coded from here on (-1 = as created, else 0 or the created value): ks265_frame_set_picture_tools"""
        self.ks._chk(self.lib.ks265_frame_set_picture_tools(self.h, C.c_int(intra_inter), C.c_int(bi_refine), C.c_int(sao), C.c_int(me_method)))

    def set_qp_map(self, dev_map):
        """one QP per CTU (device int8 array, raster; None = off) for the pictures coded from here on; the caller keeps the array alive"""
        self._qp_map = dev_map
        self.ks._chk(self.lib.ks265_frame_set_qp_map(self.h, _p(dev_map) if dev_map is not None else None))

    def set_rdoq(self, tables: "np.ndarray | None", lam: "np.ndarray | None" = None, lam_sdh: "np.ndarray | None" = None):
        """cfg.rdoq (-rdoq 1): the reference's rdoQuant for the luma transform blocks of inter CUs from the next picture on - tables int32 [4][2][180] (estBitRdoq), the two lambdas
        int64 [52] by QP; None = back to the default seam.  The copies are enqueued: the arrays are kept alive here"""
        if tables is None:
            self.ks._chk(self.lib.ks265_frame_set_rdoq(self.h, None, None, None)); return
        self._rq = (np.ascontiguousarray(tables, np.int32), np.ascontiguousarray(lam, np.int64), np.ascontiguousarray(lam_sdh if lam_sdh is not None else lam, np.int64))
        assert self._rq[0].size == 1440 and self._rq[1].size == 52 and self._rq[2].size == 52
        self.ks._chk(self.lib.ks265_frame_set_rdoq(self.h, *(a.ctypes.data_as(C.c_void_p) for a in self._rq)))
        self.ks.sync()

    def load_i420(self, dev_i420, pic: DevPic):
        self.ks._chk(self.lib.ks265_load_i420(self.h, _p(dev_i420), pic.c()))

    def store_i420(self, pic: DevPic):
        out = self.ks.zeros(self.width * self.height * 3 // 2)
        self.ks._chk(self.lib.ks265_store_i420(self.h, pic.c(), _p(out)))
        return out

    def pad(self, pic: DevPic):
        self.ks._chk(self.lib.ks265_pad_picture(self.h, pic.c()))

    def presearch(self, src: DevPic, ref: DevPic) -> np.ndarray:
        """stage A0: the pre-search vector field, [ceil(H/16), ceil(W/16), 2] int16 (integer pel), and the CTUs' window offsets [rows, cols, 2]"""
        nbx, nby = (self.cfg.width + 15) // 16, (self.cfg.height + 15) // 16
        out, off = self.ks.zeros(nbx * nby * 4), self.ks.zeros(self.geom.ctu_cols * self.geom.ctu_rows * 4)
        self.ks._chk(self.lib.ks265_presearch(self.h, src.c(), ref.c(), _p(out), _p(off)))
        return self.ks.host(out, np.int16, (nby, nbx, 2)), self.ks.host(off, np.int16, (self.geom.ctu_rows, self.geom.ctu_cols, 2))

    def me_integer(self, src: DevPic, ref: DevPic, prev_pu, pu):
        self.ks._chk(self.lib.ks265_me_integer(self.h, src.c(), ref.c(), _p(prev_pu), _p(pu)))

    def me_propagate(self, src: DevPic, ref: DevPic, pu_in, pu_out):
        self.ks._chk(self.lib.ks265_me_propagate(self.h, src.c(), ref.c(), _p(pu_in), _p(pu_out)))

    def me_subpel(self, src: DevPic, ref: DevPic, pu):
        self.ks._chk(self.lib.ks265_me_subpel(self.h, src.c(), ref.c(), _p(pu)))

    def intra_decide(self, src: DevPic, cu8):
        self.ks._chk(self.lib.ks265_intra_decide(self.h, src.c(), _p(cu8)))

    def lookahead_picture(self, cur: DevPic, ref: DevPic) -> np.ndarray:
        """low-resolution frame cost: returns [sum intra, sum inter, sum min, blocks | intra-cheaper << 32] (this frame object = half size)"""
        nctu = self.geom.ctu_cols * self.geom.ctu_rows
        ws, out = self.ks.zeros(4 * 85 * nctu), self.ks.zeros(32)
        self.ks._chk(self.lib.ks265_lookahead_picture(self.h, cur.c(), ref.c(), _p(ws), _p(out)))
        self._la_ws = ws
        return self.ks.host(out, np.uint64)

    def lookahead_inter(self, cur: DevPic, ref: DevPic) -> np.ndarray:
        """cur against another reference, with the intra costs of the lookahead_picture call before (same cur)"""
        out = self.ks.zeros(32)
        self.ks._chk(self.lib.ks265_lookahead_inter(self.h, cur.c(), ref.c(), _p(self._la_ws), _p(out)))
        return self.ks.host(out, np.uint64)

    def intra_reconstruct(self, src: DevPic, cu8, lvl, recon: DevPic):
        self.ks._chk(self.lib.ks265_intra_reconstruct(self.h, src.c(), _p(cu8), _p(lvl[0]), _p(lvl[1]), _p(lvl[2]), recon.c()))

    def cu_decide(self, pu, cu8):
        self.ks._chk(self.lib.ks265_cu_decide(self.h, _p(pu), _p(cu8)))

    def cu_flat_intra(self, cu8):
        self.ks._chk(self.lib.ks265_cu_flat_intra(self.h, _p(cu8)))

    def reconstruct(self, src: DevPic, ref: DevPic, cu8, lvl, recon: DevPic):
        self.ks._chk(self.lib.ks265_reconstruct(self.h, src.c(), ref.c(), _p(cu8), _p(lvl[0]), _p(lvl[1]), _p(lvl[2]), recon.c()))

    def reconstruct_b(self, src: DevPic, ref0: DevPic, ref1: DevPic, cu8, lvl, recon: DevPic):
        self.ks._chk(self.lib.ks265_reconstruct_b(self.h, src.c(), ref0.c(), ref1.c(), _p(cu8), _p(lvl[0]), _p(lvl[1]), _p(lvl[2]), recon.c()))

    def skip_pass(self, src: DevPic, ref0: DevPic, ref1: "DevPic | None", cu8, lvl, recon: DevPic):
        """stage D2 after reconstruct[_b]: cu8 / lvl / recon updated in place (ref1 None: a P picture)"""
        self.ks._chk(self.lib.ks265_skip_pass(self.h, src.c(), ref0.c(), ref1.c() if ref1 is not None else Pic(None, None, None), _p(cu8), _p(lvl[0]), _p(lvl[1]), _p(lvl[2]), recon.c()))

    def bi_decide(self, src: DevPic, ref0: DevPic, ref1: DevPic, pu0, pu1, pub):
        self.ks._chk(self.lib.ks265_bi_decide(self.h, src.c(), ref0.c(), ref1.c(), _p(pu0), _p(pu1), _p(pub)))

    def bi_refine_chosen(self, src: DevPic, ref0: DevPic, ref1: DevPic, pu0, pu1, pub, cu8):
        self.ks._chk(self.lib.ks265_bi_refine_chosen(self.h, src.c(), ref0.c(), ref1.c(), _p(pu0), _p(pu1), _p(pub), _p(cu8)))

    def cu_decide_b(self, pub, cu8):
        self.ks._chk(self.lib.ks265_cu_decide_b(self.h, _p(pub), _p(cu8)))

    def encode_picture_mref(self, src: DevPic, refs: list, out: DevPic):
        """P picture searching len(refs) list-0 pictures (nearest first); needs KsFrame(refs >= len(refs))"""
        arr = (Pic * len(refs))(*[r.c() for r in refs])
        self.ks._chk(self.lib.ks265_encode_picture_mref(self.h, src.c(), arr, C.c_int(len(refs)), out.c()))

    def encode_picture_b_mref(self, src: DevPic, refs0: list, refs1: list, out: DevPic):
        """B picture with several pictures per list (refs0: before it, nearest first; refs1: after it, nearest first); needs KsFrame(refs >= both lengths, bframes > 0)"""
        a0, a1 = (Pic * len(refs0))(*[r.c() for r in refs0]), (Pic * len(refs1))(*[r.c() for r in refs1])
        self.ks._chk(self.lib.ks265_encode_picture_b_mref(self.h, src.c(), a0, C.c_int(len(refs0)), a1, C.c_int(len(refs1)), out.c()))

    def ref_decide(self, pus: list, pub):
        arr = (C.c_void_p * len(pus))(*[p.data_ptr() for p in pus])
        self.ks._chk(self.lib.ks265_ref_decide(self.h, C.c_int(len(pus)), arr, _p(pub)))

    def reconstruct_mref(self, src: DevPic, refs: list, cu8, lvl, recon: DevPic):
        ra = (Pic * len(refs))(*[r.c() for r in refs])
        self.ks._chk(self.lib.ks265_reconstruct_mref(self.h, src.c(), C.c_int(len(refs)), ra, _p(cu8), _p(lvl[0]), _p(lvl[1]), _p(lvl[2]), recon.c()))

    def encode_picture_b(self, src: DevPic, ref0: DevPic, ref1: DevPic, recon_out: DevPic):
        self.ks._chk(self.lib.ks265_encode_picture_b(self.h, src.c(), ref0.c(), ref1.c(), recon_out.c()))

    def deblock(self, cu8, recon: DevPic):
        self.ks._chk(self.lib.ks265_deblock(self.h, _p(cu8), recon.c()))

    def sao(self, src: DevPic, deb: DevPic, sao, dst: DevPic):
        self.ks._chk(self.lib.ks265_sao(self.h, src.c(), deb.c(), _p(sao), dst.c()))

    def encode_picture(self, src: DevPic, ref: DevPic, is_key: bool, recon_out: DevPic):
        self.ks._chk(self.lib.ks265_encode_picture(self.h, src.c(), ref.c(), C.c_int(1 if is_key else 0), recon_out.c()))

    STAGES = ("me_integer", "me_subpel", "intra_candidates", "cu_decide", "reconstruct", "intra_pass", "deblock", "sao")

    def set_profiling(self, on: bool):
        self.ks._chk(self.lib.ks265_frame_set_profiling(self.h, C.c_int(1 if on else 0)))

    def me_int_ms(self) -> float:
        """duration of the last me_int_kernel launch alone (profiling on)"""
        v = C.c_float(-1.0)
        self.ks._chk(self.lib.ks265_frame_me_int_ms(self.h, C.byref(v)))
        return float(v.value)

    def stage_ms(self) -> dict:
        ms = (C.c_float * 8)()
        self.ks._chk(self.lib.ks265_frame_stage_ms(self.h, ms))
        return {n: float(ms[i]) for i, n in enumerate(self.STAGES)}

    def sse_picture(self, a: DevPic, b: DevPic) -> np.ndarray:
        out = self.ks.zeros(24)
        self.ks._chk(self.lib.ks265_sse_picture(self.h, a.c(), b.c(), _p(out)))
        return self.ks.host(out, np.uint64)

    # internal workspace views (device pointers wrapped as ctypes addresses)
    def ws_ptr(self, name: str, comp: int = 0) -> int:
        fn = getattr(self.lib, "ks265_frame_" + name)
        return fn(self.h, C.c_int(comp)) if name == "levels" else fn(self.h)

    def ws_read(self, name: str, nbytes: int, comp: int = 0) -> np.ndarray:
        """copy `nbytes` of an internal workspace buffer to the host"""
        t = self.ks.torch
        out = t.empty(nbytes, dtype=t.uint8, device=self.ks.device)
        src = self.ws_ptr(name, comp)
        hip = C.CDLL("libamdhip64.so")
        self.ks.sync()
        rc = hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(src), C.c_size_t(nbytes), C.c_int(3))  # device to device
        if rc != 0:
            raise Ks265Error(f"hipMemcpy rc={rc}")
        return out.cpu().numpy()
