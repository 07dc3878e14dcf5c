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
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ks265_oracle.c", "ks265_pipeline_oracle.c", "ks265_oracle.h", "ks265_pipeline_oracle.h")]
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
