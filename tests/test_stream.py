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


@pytest.mark.skipif(not os.path.exists(DEC), reason="reference decoder only exists in the builder container")
def test_reference_lists_in_any_order_decode():
    """cfg.list_mod (lists_modification_present_flag, 7.3.6.2): a B picture whose two lists hold two PAST pictures (the generalised B pictures the reference codes at
    the P positions of its hierarchy) and a P picture that predicts from the farther of two pictures - lists the default construction does not give; the
    reference's decoder must reproduce the oracle pipeline's reconstruction.  Without list_mod the writer refuses such lists."""
    from ks265codec_amd import stream as S
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    W, H = 416, 240
    clip = make_clip(W, H, 4, seed=11, pan=(5, 3))
    o = OraclePipeline(W, H, 30, lambda_q4(30), me_method=1, sdh=1, pre_search=1, merge=1, bi_refine=1, propagate=1)
    w = S.StreamWriter(W, H, max_dec_pic_buffering=4, max_num_reorder=0, sdh=1, wpp=1, list_mod=1)
    plain = S.StreamWriter(W, H, max_dec_pic_buffering=4, max_num_reorder=0, sdh=1, wpp=1)
    bs, recs, dpb = w.headers(), {}, {}
    sched = [(0, "I", None, None, [], S.NAL_IDR_W_RADL), (1, "P", 0, None, [(0, True)], S.NAL_TRAIL_R),
             (2, "B", 1, 0, [(1, True), (0, True)], S.NAL_TRAIL_R),                 # list 1 = the FARTHER past picture: default construction would give picture 1 twice
             (3, "P", 0, None, [(2, True), (0, True)], S.NAL_TRAIL_R)]              # list 0 = [0] although picture 2 is nearer
    for d, kind, r0, r1, rps, nal in sched:
        dpb[d] = o.encode(clip[d], kind, dpb.get(r0), dpb.get(r1))
        recs[d] = o.store(dpb[d])
        st = {"I": S.SLICE_I, "P": S.SLICE_P, "B": S.SLICE_B}[kind]
        kw = dict(rps=rps, l0=[r0] if r0 is not None else [], l1=[r1] if r1 is not None else [])
        bs += w.slice(nal, st, d, 30, o.cu8, o.lvl, o.sao, **kw)
        if d >= 2:
            with pytest.raises(RuntimeError):
                plain.slice(nal, st, d, 30, o.cu8, o.lvl, o.sao, **kw)
    tmp = tempfile.mkdtemp(prefix="ks265dec_")
    try:
        shutil.copy(DEC, tmp); os.chmod(os.path.join(tmp, "appdecoder"), 0o755)
        open(os.path.join(tmp, "t.265"), "wb").write(bs)
        r = subprocess.run([os.path.join(tmp, "appdecoder"), "-b", "t.265", "-o", "t.yuv", "-threads", "1"], capture_output=True, text=True, cwd=tmp)
        assert "decoder passed" in r.stdout, r.stdout[-300:]
        dec = np.fromfile(os.path.join(tmp, "t.yuv"), np.uint8).reshape(-1, W * H * 3 // 2)
        assert len(dec) == 4
        for d in range(4):
            assert (dec[d] == recs[d]).all(), f"decoded picture {d} differs in {int((dec[d] != recs[d]).sum())} samples"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_host_mirror_writes_the_encoder_fixture():
    """tools/rd_eval.py --host (the encoder host mirrored on the oracle pipeline + writer: its tools, lambda table, QP ladder) writes, for the clip of stream case
    enc_ippp_416x240_umh, the decoder-verified fixture byte for byte - the stream `ks265enc` writes on the GPU (tests/test_gpu_enc_api.py, profiles/r03_pytest_gpu.txt):
    what the mirror computes on the CPU is what the product does"""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rd_eval as R
    from ks265codec_amd.synth import make_clip
    from stream_cases import HOST_IPPP_CASCADE
    name = "enc_ippp_416x240_umh"
    clip = make_clip(416, 240, 4, seed=len(name) * 7 + 416, abc=(17, 23, 9))
    tools = dict(R.ENCODER_TOOLS, decimate=0, intra_inter=1, rdo=4, propagate=1)
    bs, _, per = R.encode_ours(clip, 416, 240, 27, "ippp", tools, cascade=list(HOST_IPPP_CASCADE), lam_scale=-1.0)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "stream_md5.json")))[name]
    assert len(bs) == g["stream_bytes"] and hashlib.md5(bs).hexdigest() == g["stream_md5"]


@pytest.mark.skipif(not os.path.exists(DEC), reason="reference decoder only exists in the builder container")
def test_reference_sao_decision_decodes():
    """cfg.sao = 2 (round 6, -sao 3): the decision of CEncSao::modeDecisionCtu enc@0x4af690 on its -sao 4 path - band offset and the two axis-aligned edge classes, Cb and Cr with a
    band position each - in the oracle pipeline; the stream written from its records decodes with the reference's decoder to the pipeline's reconstruction, and the decision uses
    every type it can"""
    from ks265codec_amd import stream as S
    from ks265codec_amd.synth import lambda_q4, make_clip
    from oracle_lib import OraclePipeline
    W, H, qp = 416, 240, 31
    clip = make_clip(W, H, 4, seed=5, pan=(5, 3))
    o = OraclePipeline(W, H, qp, lambda_q4(qp), me_method=1, sdh=1, pre_search=1, merge=1, sao=2)
    w = S.StreamWriter(W, H, max_dec_pic_buffering=2, max_num_reorder=0, sdh=1, wpp=1)
    bs, recs, ref, types = w.headers(), {}, None, set()
    for d in range(4):
        ref = o.encode(clip[d], "I" if d == 0 else "P", ref, None)
        recs[d] = o.store(ref)
        types |= {(c, int(t)) for c, t in zip(np.arange(len(o.sao)) % 3 > 0, o.sao["type"])}
        bs += w.slice(S.NAL_IDR_W_RADL if d == 0 else S.NAL_TRAIL_R, S.SLICE_I if d == 0 else S.SLICE_P, d, qp, o.cu8, o.lvl, o.sao, rps=[(d - 1, True)] if d else [], l0=[d - 1] if d else [], l1=[])
    assert {t for _, t in types} <= {-1, 0, 1, 2} and {t for c, t in types if not c} >= {0, 1, 2}, sorted(types)
    tmp = tempfile.mkdtemp(prefix="ks265dec_")
    try:
        shutil.copy(DEC, tmp); os.chmod(os.path.join(tmp, "appdecoder"), 0o755)
        open(os.path.join(tmp, "t.265"), "wb").write(bs)
        r = subprocess.run([os.path.join(tmp, "appdecoder"), "-b", "t.265", "-o", "t.yuv", "-threads", "1"], capture_output=True, text=True, cwd=tmp)
        assert "decoder passed" in r.stdout, r.stdout[-300:]
        dec = np.fromfile(os.path.join(tmp, "t.yuv"), np.uint8).reshape(-1, W * H * 3 // 2)
        assert len(dec) == 4
        for d in range(4):
            assert (dec[d] == recs[d]).all(), f"decoded picture {d} differs"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
