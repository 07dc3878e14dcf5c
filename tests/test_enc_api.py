"""The library boundary B2 (include/ks265_enc.h): layout compatibility with the SDK's qy265enc.h, configuration functions, error behaviour.
CPU only: nothing here encodes (that needs the MI355X: tests/test_gpu_enc_api.py)."""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SDK = "/root/reference/Android_demo/prebuilt/include"


def _layout(tmp_path, hdr, inc):
    exe = tmp_path / "off"
    subprocess.check_call(["gcc", f'-DHDR="{hdr}"', "-I", inc, os.path.join(HERE, "golden", "qy265_layout_probe.c"), "-o", str(exe)])
    return json.loads(subprocess.check_output([str(exe)]))


def test_struct_layout_is_the_sdks(tmp_path):
    """offsets of every field group of QY265EncConfig / QY265YUV / QY265Picture / QY265Nal equal those computed from the SDK's own header
    (tests/golden/qy265_layout.json, written in the builder container from /root/reference/.../qy265enc.h)"""
    gold = json.load(open(os.path.join(HERE, "golden", "qy265_layout.json")))
    assert _layout(tmp_path, "ks265_enc.h", os.path.join(ROOT, "include")) == gold
    if os.path.exists(os.path.join(SDK, "qy265enc.h")):
        assert _layout(tmp_path, "qy265enc.h", SDK) == gold


class Cfg(C.Structure):
    pass


def _lib():
    from ks265codec_amd import stream
    so = stream.build()
    if not os.path.exists(stream.HIPLIB):
        pytest.skip("libks265hip.so not built")
    return C.CDLL(so)


def test_config_functions_and_error_codes():
    lib = _lib()
    gold = json.load(open(os.path.join(HERE, "golden", "qy265_layout.json")))
    buf = (C.c_uint8 * gold["sizeof_config"])()

    def i32(name): return C.c_int32.from_buffer(buf, gold[name]).value
    assert lib.QY265ConfigDefaultPreset(buf, b"slow", None, b"default") == 0
    assert (i32("preset"), i32("me"), i32("subme"), i32("refnum"), i32("sao"), i32("rdoq"), i32("searchrange"), i32("bframes")) == (5, 2, 1, 1, 4, 1, 64, -1)
    assert lib.QY265ConfigDefaultPreset(buf, b"veryfast", b"default", b"zerolatency") == 0
    assert (i32("preset"), i32("latency"), i32("me"), i32("sao"), i32("rdoq")) == (2, 0, 1, 3, 0)
    assert lib.QY265ConfigDefaultPreset(buf, b"warp9", None, None) != 0
    assert lib.QY265ConfigParse(buf, b"qp", b"32") == 0 and i32("qp") == 32
    assert lib.QY265ConfigParse(buf, b"qp", b"99") == -2                 # QY265_PARAM_BAD_VALUE
    assert lib.QY265ConfigParse(buf, b"nosuchflag", b"1") == -1          # QY265_PARAM_BAD_NAME
    assert lib.QY265ConfigParse(buf, b"preset", b"slow") == 0 and i32("preset") == 5
    # tools asked for BY NAME that select the reference's own restated functions are stored apart from the presets' values (the presets' streams stay this build's)
    assert lib.QY265ConfigParse(buf, b"sao", b"3") == 0 and i32("sao") == 5 and lib.QY265ConfigParse(buf, b"sao", b"4") == 0 and i32("sao") == 4 and lib.QY265ConfigParse(buf, b"sao", b"5") == -2
    assert lib.QY265ConfigParse(buf, b"rdoq", b"1") == 0 and i32("rdoq") == 2
    err = C.c_int(0)
    lib.QY265EncoderOpen.restype = C.c_void_p
    assert lib.QY265EncoderOpen(None, C.byref(err)) is None and (err.value & 0xFFFFFFFF) == 0x80000003      # QY_POINTER
    assert lib.QY265ConfigParse(buf, b"wdt", b"416") == 0 and lib.QY265ConfigParse(buf, b"hgt", b"241") == 0
    assert lib.QY265EncoderOpen(buf, C.byref(err)) is None and (err.value & 0xFFFFFFFF) == 0x80000004       # QY_NOTSUPPORTED: height not a multiple of 8
    assert (C.c_char * 8).in_dll(lib, "strLibQy265Version").value.startswith(b"ks265enc")
    assert lib.QY265EncoderDelayedFrames(None) == 0


def test_exports_the_sdk_entry_points():
    lib = _lib()
    for n in ("QY265EncoderOpen", "QY265EncoderClose", "QY265EncoderReconfig", "QY265EncoderEncodeHeaders", "QY265EncoderEncodeFrame", "QY265EncoderKeyFrameRequest",
              "QY265EncoderDelayedFrames", "QY265ConfigDefault", "QY265ConfigDefaultPreset", "QY265ConfigParse", "QY265SetLogPrintf", "strLibQy265Version",
              "ks265_write_vps", "ks265_write_sps", "ks265_write_pps", "ks265_write_slice", "ks265_slice_scratch_bytes", "ks265_enc_get_stats"):
        assert hasattr(lib, n), n
