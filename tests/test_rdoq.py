"""The oracle's restatement of the reference's RATE-DISTORTION OPTIMISED QUANTISATION (oracle/ks265_rdoq_ref.c: h265_codec::rdoQuant enc@0x4aac50)
replayed on calls recorded inside the reference binary (tests/golden/rdoq.npz, written by oracle/ref_probe/gen_rdoq_traces.py: real `appencoder` runs at -preset medium /
slow / veryslow, QP 22..37, -bframes 3, -sbh 0 and non-default -rdoql/-rdoqc/-rdoqls/-rdoqcs weights, with rdoQuant and estBitRdoq hooked and the stream checked to be
unchanged).  Every recorded call must come out with the reference's levels (signs included), return value (number of non-zero levels), last scan position
(TTransUnit+0x40), significance masks (TTransUnit+0x68) and the mask of sub-blocks that hide a sign (TTransUnit+0x50)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle_lib import lib, ptr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdoq.npz")


def replay_all():
    z = np.load(GOLD)
    meta, lam, tab, offs = z["meta"], z["lam"], np.ascontiguousarray(z["tab"]), z["offs"]
    lvl_in, lvl_out, coef = z["lvl_in"], z["lvl_out"], np.ascontiguousarray(z["coef"])
    o = lib()
    rows = []
    for i in range(len(meta)):
        m = meta[i]
        a, b = int(offs[i]), int(offs[i + 1])
        lvl, mask = lvl_in[a:b].copy(), z["mask_in"][i].copy()
        ol, oh = C.c_int32(0), C.c_uint64(0)
        ret = o.kso_ref_rdo_quant(ptr(lvl), ptr(coef, 2 * a), int(m[0]), int(m[1]), int(m[2]), int(m[3]), int(m[4]), C.c_int64(int(lam[i][0])), C.c_int64(int(lam[i][1])), ptr(tab, 720 * i),
                                  int(m[5]), int(m[6]), ptr(mask), int(m[7]), int(m[8]), C.byref(ol), C.byref(oh))
        ncg = max(1, (b - a) // 16)
        rows.append(dict(run=int(z["run_of"][i]), m=m, lvl_ok=bool((lvl == lvl_out[a:b]).all()), ret=ret, last=ol.value, hidden=oh.value, hidden_exp=int(z["hidden"][i]),
                         mask_ok=bool((mask[:ncg] == z["mask_out"][i][:ncg]).all()), lin=lvl_in[a:b].astype(np.int32), lout=lvl_out[a:b].astype(np.int32)))
    return [str(s) for s in z["runs"]], rows


def test_rdo_quant_matches_reference_traces():
    runs, rows = replay_all()
    assert len(rows) >= 1200 and len(runs) >= 6
    bad = [r for r in rows if not (r["lvl_ok"] and r["ret"] == r["m"][9] and r["last"] == r["m"][10] and r["hidden"] == r["hidden_exp"] and r["mask_ok"])]
    assert not bad, f"{len(bad)} of {len(rows)} calls differ, first: run {runs[bad[0]['run']]} meta {bad[0]['m']} levels {bad[0]['lvl_ok']} ret {bad[0]['ret']} last {bad[0]['last']}"
    # the fixture covers what it claims to cover
    assert {(int(r["m"][0]), int(r["m"][2])) for r in rows} >= {(l, c) for l in (2, 3, 4) for c in (0, 1, 2)} | {(5, 0)}, "every block size of every component"
    assert {int(r["m"][1]) for r in rows} == {0, 1, 2}, "the three scans"
    changed = [r for r in rows if (np.abs(r["lin"]) != np.abs(r["lout"])).any()]
    assert len(changed) >= 600
    assert sum(1 for r in rows if (np.abs(r["lout"]) > np.abs(r["lin"])).any()) >= 100, "sign hiding raised a level"
    assert sum(1 for r in rows if r["m"][9] == 0) >= 40, "blocks dropped whole"
    assert sum(1 for r in rows if r["m"][10] + 1 < r["m"][6] and r["m"][9]) >= 100, "the last position moved"
    assert sum(1 for r in rows if r["m"][8] == 0) >= 100 and all(r["hidden_exp"] == 0 for r in rows if r["m"][8] == 0), "-sbh 0 runs hide nothing"
    assert sum(1 for r in rows if r["hidden_exp"]) >= 400
    assert len({(int(r["m"][13]), int(r["m"][14])) for r in rows}) >= 4, "default and non-default lambda weights, luma and chroma"
    assert len({int(r["m"][12]) for r in rows}) >= 10, "QPs"
