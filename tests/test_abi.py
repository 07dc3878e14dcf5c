"""CPU: the C-ABI library loads and exports every symbol include/ks265_hip.h declares; no compute calls."""
from __future__ import annotations

import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ks265_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ks265_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from ks265codec_amd import build as kb
    kb.build()
    from ks265codec_amd.lib import load_library
    return load_library()


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_export_list_matches_header():
    from ks265codec_amd.lib import EXPORTS
    assert sorted(EXPORTS) == _declared()


def test_struct_layouts_match_header(lib):
    from ks265codec_amd import lib as L
    assert L.BLK.itemsize == 16 and L.BLK3.itemsize == 24 and L.EDGE.itemsize == 12 and L.SAO_RECT.itemsize == 16
    assert L.PU.itemsize == 16 and L.CU8.itemsize == 12 and L.PU_B.itemsize == 16 and L.SAO_PARAM.itemsize == 8
    assert L.INTRA_BLK.itemsize == 16 and L.INTRA_REF.itemsize == 16
    assert C.sizeof(L.FrameCfg) == 124 and C.sizeof(L.FrameGeom) == 80         # 31 x int32 (..., rdo, intra_inter, propagate, sub_satd, sub_thr, sub_flat, sub_cap, sub_cap_step, sub_diag_fast, part, tu_inter, skip_rd)


def test_geometry_and_argument_errors_without_gpu(lib):
    from ks265codec_amd.lib import FrameCfg, FrameGeom
    g = FrameGeom()
    cfg = FrameCfg(3840, 2160, 27, 80, 64, 0, 1, 1, 1, 0, 0, 0)
    assert lib.ks265_frame_geometry(C.byref(cfg), C.byref(g)) == 0
    assert (g.stride_y, g.rows_y, g.ctu_cols, g.ctu_rows, g.pu_per_ctu) == (4096, 2320, 60, 34, 85)
    assert g.stride_y % 128 == 0 and (g.pad_y * g.stride_y + g.pad_y) % 16 == 0
    bad = FrameCfg(3841, 2160, 27, 80, 64, 0, 1, 1, 1, 0, 0, 0)
    assert lib.ks265_frame_geometry(C.byref(bad), C.byref(g)) == -4          # KS265_NOTSUPPORTED
    assert lib.ks265_frame_geometry(None, C.byref(g)) == -3                  # KS265_POINTER
    assert lib.ks265_sad_batch(None, None, 0, None, 0, None, 0, None) == -3
    assert b"ks265hip" in lib.ks265_version()


def test_no_gpu_means_no_context(lib):
    """On a box without a HIP device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.ks265_create(C.byref(h), 0) == -5                             # KS265_NO_DEVICE
    from ks265codec_amd.lib import Ks265Error, KsContext
    with pytest.raises(Ks265Error):
        KsContext(0)


def test_product_does_not_touch_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    pkg = os.path.join(ROOT, "ks265codec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "libks265_oracle" not in text and "ks265o_" not in text and "kso_" not in text, f


def test_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/ks265_hip.h must compile as C99 (-pedantic) with the documented struct sizes"""
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "ks265_hip.h"\nint main(void) { ks265_frame_cfg c = {0}; (void)c;\n'
                   '  return sizeof(ks265_intra_blk) == 16 && sizeof(ks265_cu8) == 12 && sizeof(ks265_pu) == 16 && sizeof(ks265_frame_cfg) == 124 ? 0 : 1; }\n')
    exe = tmp_path / "h"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0
