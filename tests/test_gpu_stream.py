"""GPU: the records of the HIP pipeline, copied to the host and written by the host bitstream writer, give byte for byte the streams that
the reference's decoder verified in the builder container (tests/golden/stream_md5.json) - i.e. what the MI355X path produces is a
conforming HEVC stream that decodes to its own reconstruction."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import pytest

from stream_cases import CASES, case_bir, case_dec, case_ii, case_lambda, case_merge, case_prop, case_ps, case_rdo, case_part, case_rqt, case_sdh, case_skip, case_subme, make_stream, schedule

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_md5.json")))


@pytest.fixture(scope="module")
def ks():
    from ks265codec_amd.lib import KsContext
    c = KsContext(0)
    yield c
    c.close()


def hip_encoder(ks, name):
    from ks265codec_amd.lib import CU8, SAO_PARAM, KsFrame
    from ks265codec_amd.synth import lambda_q4, make_clip
    W, H, qp, me, thr, sao, df, kind, par = CASES[name]
    n = 1 + max(s[0] for s in schedule(kind, par))
    clip = make_clip(W, H, n, seed=len(name) * 7 + W, abc=(17, 23, 9))
    f = KsFrame(ks, W, H, qp, lambda_q4(qp), me_method=me, me_hex_thr=thr, sao=sao, deblock=df, bframes=3 if kind in ("hier", "hiermr", "hiera") else 0, refs=par if kind == "mref" else 2 if kind == "hiermr" else 3 if kind == "hiera" else 1,
                sdh=case_sdh(name), pre_search=case_ps(name), merge=case_merge(name), bi_refine=case_bir(name), decimate=case_dec(name), rdo=case_rdo(name), intra_inter=case_ii(name), propagate=case_prop(name), part=case_part(name), tu_inter=case_rqt(name), skip_rd=case_skip(name), **case_subme(name))
    g = f.geom
    src = f.new_pic()
    dpb = {}

    def encode(d, k, l0, l1, q):
        f.set_qp(q, case_lambda(name, q, k))
        f.load_i420(ks.dev(clip[d]), src)
        out = f.new_pic()
        if k == "B" and (len(l0) > 1 or len(l1) > 1):
            f.encode_picture_b_mref(src, [dpb[r] for r in l0], [dpb[r] for r in l1], out)
        elif k == "B":
            f.encode_picture_b(src, dpb[l0[0]], dpb[l1[0]], out)
        elif k == "P" and len(l0) > 1:
            f.encode_picture_mref(src, [dpb[r] for r in l0], out)
        else:
            f.encode_picture(src, dpb[l0[0]] if l0 else out, k == "I", out)
        dpb[d] = out
        cu8 = f.ws_read("cu8", g.bytes_cu8).view(CU8)
        lvl = [f.ws_read("levels", W * H * 2, 0).view(np.int16), f.ws_read("levels", W * H // 2, 1).view(np.int16), f.ws_read("levels", W * H // 2, 2).view(np.int16)]
        saop = f.ws_read("sao", g.bytes_sao).view(SAO_PARAM)
        return cu8, lvl, saop, ks.host(f.store_i420(out), np.uint8)
    return encode, f


@pytest.mark.parametrize("name", list(CASES))
def test_hip_records_give_the_decoder_verified_stream(ks, name):
    enc, f = hip_encoder(ks, name)
    try:
        bs, recs = make_stream(name, enc)
    finally:
        f.close()
    assert [hashlib.md5(recs[d].tobytes()).hexdigest() for d in sorted(recs)] == GOLD[name]["recon_md5"], f"{name}: reconstruction differs"
    assert hashlib.md5(bs).hexdigest() == GOLD[name]["stream_md5"], f"{name}: stream differs from the decoder-verified fixture ({len(bs)} vs {GOLD[name]['stream_bytes']} bytes)"
