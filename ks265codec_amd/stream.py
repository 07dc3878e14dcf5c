"""ctypes binding of the host bitstream writer (ks265codec_amd/host/ks265_stream.c -> libks265enc.so; include/ks265_stream.h).
Plain C on the CPU: no GPU, no torch needed.  The writer consumes the records the HIP stages leave (CU map, levels, SAO parameters)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "host", "ks265_stream.c")]
LIB = os.path.join(HERE, "libks265enc.so")
INC = os.path.join(os.path.dirname(HERE), "include")
_lib = None

SLICE_B, SLICE_P, SLICE_I = 0, 1, 2
NAL_TRAIL_N, NAL_TRAIL_R, NAL_IDR_W_RADL, NAL_IDR_N_LP = 0, 1, 19, 20


class StreamCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "fps_num", "fps_den", "sao", "deblock", "beta_offset_div2", "tc_offset_div2",
                                          "max_dec_pic_buffering", "max_num_reorder", "log2_max_poc_lsb", "sdh", "wpp", "list_mod", "cu_qp_delta", "tu_inter")]


class SliceIn(C.Structure):
    _fields_ = [("nal_type", C.c_int32), ("slice_type", C.c_int32), ("poc", C.c_int32), ("qp", C.c_int32), ("num_rps", C.c_int32),
                ("rps_poc", C.c_int32 * 16), ("rps_used", C.c_uint8 * 16), ("num_l0", C.c_int32), ("num_l1", C.c_int32),
                ("l0_poc", C.c_int32 * 4), ("l1_poc", C.c_int32 * 4), ("cu8", C.c_void_p), ("lvl", C.c_void_p * 3), ("sao", C.c_void_p), ("qp_map", C.c_void_p)]


ENC_SRC = [os.path.join(HERE, "host", "ks265_enc.c")]
CLI_SRC = os.path.join(HERE, "host", "ks265_cli.c")
CLI = os.path.join(HERE, "ks265enc")
HIPLIB = os.path.join(HERE, "libks265hip.so")


def build(force: bool = False) -> str:
    """libks265enc.so = bitstream writer + the SDK-compatible encoder API (links libks265hip.so when that has been built; without it only
    the writer is available, which is all the CPU tests need); ks265enc = the appencoder-compatible CLI"""
    deps = SRC + ENC_SRC + [CLI_SRC] + [os.path.join(INC, h) for h in ("ks265_stream.h", "ks265_hip.h", "ks265_enc.h")]
    have_hip = os.path.exists(HIPLIB)
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps) or (have_hip and os.path.getmtime(HIPLIB) > os.path.getmtime(LIB)):
        flags = ["gcc", "-O2", "-std=c11", "-fPIC", "-Wall", "-Wextra", "-I", INC]
        if have_hip:
            subprocess.check_call([*flags, "-shared", "-o", LIB, *SRC, *ENC_SRC, "-L", HERE, "-lks265hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"])
            subprocess.check_call([*flags, "-o", CLI, CLI_SRC, "-L", HERE, "-lks265enc", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + HERE])
        else:
            subprocess.check_call([*flags, "-shared", "-o", LIB, *SRC])
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for n in ("ks265_write_vps", "ks265_write_sps", "ks265_write_pps", "ks265_write_slice"):
            getattr(_lib, n).restype = C.c_long
        _lib.ks265_slice_scratch_bytes.restype = C.c_size_t
    return _lib


class StreamWriter:
    """Annex-B HEVC stream from per-picture records (host numpy arrays with the dtypes of ks265codec_amd.lib)."""

    def __init__(self, width: int, height: int, sao: int = 1, deblock: int = 1, beta_offset_div2: int = 0, tc_offset_div2: int = 0,
                 max_dec_pic_buffering: int = 2, max_num_reorder: int = 0, sdh: int = 0, wpp: int = 0, list_mod: int = 0, cu_qp_delta: int = 0, tu_inter: int = 0):
        self.l = lib()
        self.cfg = StreamCfg(width, height, 0, 0, sao, deblock, beta_offset_div2, tc_offset_div2, max_dec_pic_buffering, max_num_reorder, 16, sdh, wpp, list_mod, cu_qp_delta, tu_inter)
        self.scratch = np.zeros(self.l.ks265_slice_scratch_bytes(C.byref(self.cfg)), np.uint8)
        self.out = np.zeros(width * height * 4 + 65536, np.uint8)

    def _take(self, n: int) -> bytes:
        if n < 0:
            raise RuntimeError(f"ks265 stream writer failed rc={n}")
        return self.out[:n].tobytes()

    def headers(self) -> bytes:
        o = self.out.ctypes.data_as(C.c_void_p)
        return b"".join(self._take(f(C.byref(self.cfg), o, C.c_size_t(self.out.size))) for f in (self.l.ks265_write_vps, self.l.ks265_write_sps, self.l.ks265_write_pps))

    def rdoq_tables(self, states: "np.ndarray | None", slice_type: int = 1, qp: int = 27) -> np.ndarray:
        """ks265_rdoq_tables: the eight bit tables [4][2][180] of the reference's rdoQuant from the context states a slice ended with (final_contexts()[0]) or, None, from the
        initial states of a slice of slice_type at qp"""
        out = np.zeros(1440, np.int32)
        st = None if states is None else np.ascontiguousarray(states, np.uint8)
        rc = self.l.ks265_rdoq_tables(C.byref(self.cfg), st.ctypes.data_as(C.c_void_p) if st is not None else None, C.c_int(slice_type), C.c_int(qp), out.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError(f"ks265_rdoq_tables rc={rc}")
        return out

    def final_contexts(self):
        """(states, layout) of the slice written last: ks265_slice_final_contexts"""
        st, lay = np.zeros(256, np.uint8), (C.c_int * 10)()
        n = self.l.ks265_slice_final_contexts(C.byref(self.cfg), self.scratch.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), C.c_int(256), lay)
        if n < 0:
            raise RuntimeError(f"ks265_slice_final_contexts rc={n}")
        return st[:n].copy(), list(lay)

    def slice(self, nal_type: int, slice_type: int, poc: int, qp: int, cu8: np.ndarray, lvl: "list[np.ndarray]", sao: "np.ndarray | None",
              rps: "list[tuple[int, bool]]" = (), l0: "list[int]" = (), l1: "list[int]" = (), qp_map: "np.ndarray | None" = None) -> bytes:
        s = SliceIn()
        s.nal_type, s.slice_type, s.poc, s.qp = nal_type, slice_type, poc, qp
        s.num_rps = len(rps)
        for i, (p, u) in enumerate(rps):
            s.rps_poc[i], s.rps_used[i] = p, int(u)
        s.num_l0, s.num_l1 = len(l0), len(l1)
        for i, p in enumerate(l0):
            s.l0_poc[i] = p
        for i, p in enumerate(l1):
            s.l1_poc[i] = p
        keep = [np.ascontiguousarray(cu8)] + [np.ascontiguousarray(a, dtype=np.int16) for a in lvl]
        s.cu8 = keep[0].ctypes.data
        for i in range(3):
            s.lvl[i] = keep[1 + i].ctypes.data
        if sao is not None:
            keep.append(np.ascontiguousarray(sao))
            s.sao = keep[-1].ctypes.data
        if qp_map is not None:                                            # cfg.cu_qp_delta: one QP per CTU, raster order
            keep.append(np.ascontiguousarray(qp_map, dtype=np.int8))
            s.qp_map = keep[-1].ctypes.data
        return self._take(self.l.ks265_write_slice(C.byref(self.cfg), C.byref(s), self.scratch.ctypes.data_as(C.c_void_p),
                                                   self.out.ctypes.data_as(C.c_void_p), C.c_size_t(self.out.size)))
