"""Stream-level parity (SURVEY.md §7.1 "stream level": conformance).  The host bitstream writer (ks265codec_amd/host/ks265_stream.c) turns
the pipeline's records into an HEVC stream; the reference's OWN decoder must decode it to exactly the pipeline's reconstruction.

  * tests/golden/stream_md5.json was written by tests/golden/gen_stream_golden.py in the builder container, where every case was decoded by
    /root/reference/ubuntu_x64/appdecoder and compared picture by picture; it holds the MD5 of each decoder-verified stream.
  * here (any machine): the CPU oracle pipeline + writer must reproduce those streams byte for byte;
  * where the reference decoder exists, one case is also decoded live."""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from stream_cases import CASES, make_stream, oracle_encoder

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_md5.json")))
DEC = "/root/reference/ubuntu_x64/appdecoder"


@pytest.mark.parametrize("name", [n for n in CASES if "1280" not in n])
def test_oracle_pipeline_stream_is_the_decoder_verified_one(name):
    bs, recs = make_stream(name, oracle_encoder(name))
    assert hashlib.md5(bs).hexdigest() == GOLD[name]["stream_md5"], f"{name}: stream differs from the decoder-verified fixture ({len(bs)} vs {GOLD[name]['stream_bytes']} bytes)"
    assert [hashlib.md5(recs[d].tobytes()).hexdigest() for d in sorted(recs)] == GOLD[name]["recon_md5"]


def test_headers_and_argument_errors():
    import ctypes as C
    from ks265codec_amd import stream as S
    w = S.StreamWriter(416, 240)
    h = w.headers()
    nal_types = [(h[i + 4] >> 1) & 63 for i in range(len(h) - 4) if h[i:i + 4] == b"\x00\x00\x00\x01"]
    assert nal_types == [32, 33, 34]                               # VPS, SPS, PPS
    assert b"\x00\x00\x00" not in h.replace(b"\x00\x00\x00\x01", b"")          # emulation prevention
    bad = S.StreamCfg(417, 240, 0, 0, 1, 1, 0, 0, 2, 0, 16, 0)
    out = np.zeros(256, np.uint8)
    assert w.l.ks265_write_sps(C.byref(bad), out.ctypes.data_as(C.c_void_p), C.c_size_t(256)) == -4      # KS265_NOTSUPPORTED
    assert w.l.ks265_write_sps(None, out.ctypes.data_as(C.c_void_p), C.c_size_t(256)) == -3              # KS265_POINTER
    assert w.l.ks265_write_sps(C.byref(w.cfg), out.ctypes.data_as(C.c_void_p), C.c_size_t(8)) == -4       # buffer too small


@pytest.mark.skipif(not os.path.exists(DEC), reason="reference decoder only exists in the builder container")
def test_reference_decoder_reproduces_the_reconstruction_live():
    name = "hierb4_416x240"
    W, H = CASES[name][0], CASES[name][1]
    bs, recs = make_stream(name, oracle_encoder(name))
    tmp = tempfile.mkdtemp(prefix="ks265dec_")
    try:
        shutil.copy(DEC, tmp); os.chmod(os.path.join(tmp, "appdecoder"), 0o755)
        open(os.path.join(tmp, "t.265"), "wb").write(bs)
        r = subprocess.run([os.path.join(tmp, "appdecoder"), "-b", "t.265", "-o", "t.yuv", "-threads", "1"], capture_output=True, text=True, cwd=tmp)
        assert "decoder passed" in r.stdout, r.stdout[-300:]
        dec = np.fromfile(os.path.join(tmp, "t.yuv"), np.uint8).reshape(-1, W * H * 3 // 2)
        assert len(dec) == len(recs)
        for d in sorted(recs):
            assert (dec[d] == recs[d]).all(), f"decoded picture {d} differs"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
