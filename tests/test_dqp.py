"""QP per CTU (cu_qp_delta, quantisation group = the CTU; round 4, for adaptive quantisation): the oracle pipeline quantises every CTU with its map entry and deblocks with
the QpY the DECODER derives (H.265 8.6.1: the delta arrives with a CTU's first coded residual, the CUs in front of it keep the previous CTU's QP; every CTU row starts from
the slice QP under entropy_coding_sync), the writer sends cu_qp_delta_abs / sign.  Checked the only way that proves all of it: the reference's own decoder reconstructs the
stream to exactly the oracle's pictures.  Without the decoder (oracle/_ref not staged) the stream's MD5 is held against the value recorded when the decoder was there."""
from __future__ import annotations

import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from ks265codec_amd import stream as S
from ks265codec_amd.synth import ENCODER_TOOLS, lambda_q4, make_clip
from oracle_lib import OraclePipeline

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DEC = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "appdecoder")
GOLD = os.path.join(HERE, "golden", "dqp_md5.json")


FIXTURE_TOOLS = dict(ENCODER_TOOLS, bi_refine=2)              # the tool set the decoder-verified fixtures were written with (the host's up to the end of round 6; from -preset slower on since)


def encode(W, H, n, qp, seed, spread, kinds="IPPP", tools=FIXTURE_TOOLS):
    clip = make_clip(W, H, n, seed=seed, abc=(17, 23, 9), pan=(5, 3))
    rng = np.random.default_rng(seed)
    cols, rows = (W + 63) // 64, (H + 63) // 64
    o = OraclePipeline(W, H, qp, lambda_q4(qp), **tools)
    w = S.StreamWriter(W, H, sdh=tools.get("sdh", 0), wpp=1, cu_qp_delta=1, max_dec_pic_buffering=3, max_num_reorder=1 if "B" in kinds else 0)
    bs = w.headers()
    recs, dpb, effs = [], {}, []
    order = list(range(n))
    if "B" in kinds:                                                # I P(2) B(1) P(4) B(3) ...
        order = [0] + [x for t in range(2, n, 2) for x in (t, t - 1)]
    for d in order:
        kind = "I" if d == 0 else ("B" if ("B" in kinds and d % 2 == 1) else "P")
        q = qp if kind == "I" else qp + 1 + (kind == "B")
        qmap = np.clip(q + rng.integers(-spread, spread + 1, cols * rows), 10, 51).astype(np.int8)
        qmap[rng.random(cols * rows) < 0.3] = q                    # runs of equal QPs: zero deltas
        o.set_qp(q, lambda_q4(q, inter=kind != "I"))
        o.set_qp_map(qmap)
        if kind == "I":
            dpb[d] = o.encode(clip[d], "I")
        elif kind == "P":
            r0 = max(p for p in dpb if p < d and (p % 2 == 0 or "B" not in kinds))
            dpb[d] = o.encode(clip[d], "P", dpb[r0])
        else:
            dpb[d] = o.encode(clip[d], "B", dpb[d - 1], dpb[d + 1])
        effs.append(o.effective_qp().copy())
        recs.append((d, o.store(dpb[d])))
        if kind == "I":
            bs += w.slice(S.NAL_IDR_W_RADL, S.SLICE_I, 0, q, o.cu8, o.lvl, o.sao, qp_map=qmap)
        elif kind == "P":
            bs += w.slice(S.NAL_TRAIL_R, S.SLICE_P, d, q, o.cu8, o.lvl, o.sao, rps=[(r0, True)] + ([(d - 2, False)] if False else []), l0=[r0], qp_map=qmap)
        else:
            bs += w.slice(S.NAL_TRAIL_N, S.SLICE_B, d, q, o.cu8, o.lvl, o.sao, rps=[(d - 1, True), (d + 1, True)], l0=[d - 1], l1=[d + 1], qp_map=qmap)
        keep = {d} | ({d - 1, d + 1} if kind == "B" else set())
        for k in [k for k in dpb if k < d - 2 and k not in keep]:
            del dpb[k]
    o.set_qp_map(None)
    return bs, dict(recs), effs


CASES = {"ippp_416x240_spread6": dict(W=416, H=240, n=4, qp=30, seed=5, spread=6), "ippp_200x136_spread12": dict(W=200, H=136, n=4, qp=27, seed=6, spread=12),
         "ipb_416x240_spread4": dict(W=416, H=240, n=5, qp=32, seed=7, spread=4, kinds="IPB")}


@pytest.mark.parametrize("name", sorted(CASES))
def test_qp_per_ctu_decodes_to_the_oracle_reconstruction(tmp_path, name):
    kw = CASES[name]
    bs, recs, effs = encode(**kw)
    W, H = kw["W"], kw["H"]
    assert any(len(set(e.tolist())) > 2 for e in effs), "the pictures really carry several QPs"
    md5 = hashlib.md5(bs).hexdigest()
    gold = json.load(open(GOLD)) if os.path.exists(GOLD) else {}
    if os.path.exists(REF_DEC):
        (tmp_path / "s.265").write_bytes(bs)
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "s.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "1"], capture_output=True, text=True, cwd=tmp_path)
        assert "decoder passed" in d.stdout, d.stdout[-300:] + d.stderr[-300:]
        dec = np.fromfile(tmp_path / "d.yuv", np.uint8)
        fsz = W * H * 3 // 2
        assert dec.size == len(recs) * fsz, (dec.size, len(recs))
        for t in sorted(recs):
            assert (dec[t * fsz:(t + 1) * fsz] == recs[t]).all(), f"{name}: picture {t} decodes differently from the oracle's reconstruction"
        if gold.get(name) != md5 and os.environ.get("KS265_WRITE_GOLDEN"):
            gold[name] = md5
            json.dump(gold, open(GOLD, "w"), indent=1, sort_keys=True)
    assert gold.get(name) == md5, f"{name}: stream MD5 {md5} differs from the decoder-verified one {gold.get(name)}"
