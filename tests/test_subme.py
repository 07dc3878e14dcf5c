"""The oracle's restatement of the reference's SUB-PEL REFINEMENT CONTROL (oracle/ks265_subme_ref.c: getMvResolution enc@0x483ca0, subMeSquare enc@0x4b5660,
subMeHpel_RealInterp enc@0x4b4e90, subMeQpel_8Sad_v{0,2}h{0,2}_RealInterp enc@0x4b2bc0-0x4b43a0) replayed on calls recorded inside the reference binary
(tests/golden/subme.npz, written by oracle/ref_probe/gen_subme_traces.py: real `appencoder` runs at -preset veryfast / medium / slow / veryslow, -subme 2 and
-bframes 3 on three clips, with the two outer functions hooked and the stream checked to be unchanged).  Every recorded call must come out with the reference's
vector, cost (tME+0x90), rate (tME+0x94), distortion (tME+0x98) and predictor index (tME+0x58); every getMvResolution call with its verdict (tME+0x3bc)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle_lib import lib, ptr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subme.npz")


def replay_all():
    z = np.load(GOLD)
    sq, rs, cm, offs = z["sq_hdr"], z["res_hdr"], np.ascontiguousarray(z["cm"]), z["offs"]
    fenc, region, rfenc, rregion = (np.ascontiguousarray(z[k]) for k in ("fenc", "region", "res_fenc", "res_region"))
    o = lib()
    rows = []
    for i in range(len(sq)):
        h, hr = np.ascontiguousarray(sq[i]), np.ascontiguousarray(rs[i])
        fo, ro, rfo, rro = (int(v) for v in offs[i])
        out = (C.c_int32 * 9)()
        o.kso_subme_replay(ptr(h), ptr(fenc, fo), ptr(region, ro), ptr(cm, 68 * i), out)
        o2 = (C.c_int32 * 2)()
        o.kso_mvres_replay(ptr(hr), ptr(rfenc, rfo) if rfo >= 0 else None, ptr(rregion, rro) if rro >= 0 else None, o2)
        rows.append(dict(run=int(z["run_of"][i]), h=h, got=tuple(out[:6]), exp=tuple(int(h[k]) for k in range(32, 38)), overlap=bool(out[6]), hpel=out[7], qpel=out[8],
                         res_got=o2[0], res_exp=int(hr[32]), res_pixels=rfo >= 0))
    return [str(s) for s in z["runs"]], rows


def test_subpel_refinement_matches_reference_traces():
    runs, rows = replay_all()
    assert len(rows) >= 1000 and len(runs) >= 6
    bad = [r for r in rows if r["got"] != r["exp"]]
    assert not bad, f"{len(bad)} of {len(rows)} calls differ, first: run {runs[bad[0]['run']]} got {bad[0]['got']} reference {bad[0]['exp']} header {bad[0]['h'][:30]}"
    # the fixture covers what it claims to cover
    moved = [r for r in rows if r["exp"][:2] != (int(r["h"][5]), int(r["h"][6]))]
    assert len(moved) >= 500
    assert {r["hpel"] for r in rows} >= set(range(-1, 8)) and {r["qpel"] for r in rows} >= set(range(-1, 8)), "every half / quarter candidate wins somewhere"
    assert sum(r["overlap"] for r in rows) >= 60, "the overlapping-buffer path of the reference (half step moved right) is in the fixture and reproduced"
    assert sum(1 for r in rows if r["h"][26] == 0) >= 300, "Hadamard calls (satdInter: veryslow)"
    assert sum(1 for r in rows if r["h"][11] == 2 and r["h"][26] == 1) >= 100, "-subme 2 with SAD"
    assert sum(1 for r in rows if r["h"][3] != r["h"][4]) >= 50, "rectangular PUs (-part 1)"
    assert sum(1 for r in rows if r["h"][10] == 0) >= 50 and sum(1 for r in rows if r["h"][9]) >= 5, "calls without refinement, calls on the tME+0x65 rate path"
    # quarter step skipped by the flat-surface verdict (cfg+0x464 != 0): half-grid results of refined calls
    assert sum(1 for r in rows if r["h"][10] and r["h"][12] and r["qpel"] == -1) >= 100


def test_mv_resolution_matches_reference_traces():
    _, rows = replay_all()
    bad = [r for r in rows if r["res_got"] != r["res_exp"]]
    assert not bad, f"{len(bad)} getMvResolution verdicts differ"
    assert sum(r["res_pixels"] for r in rows) >= 100, "calls in which the function computes the four neighbour SADs itself"
    assert sum(1 for r in rows if r["res_exp"] == 0) >= 50 and sum(1 for r in rows if r["res_exp"] == 1) >= 500


if __name__ == "__main__":
    runs, rows = replay_all()
    for k, n in enumerate(runs):
        mine = [r for r in rows if r["run"] == k]
        print(n, "calls", len(mine), "mismatch", sum(r["got"] != r["exp"] for r in mine), "moved", sum(r["exp"][:2] != (int(r["h"][5]), int(r["h"][6])) for r in mine),
              "overlap path", sum(r["overlap"] for r in mine), "mvres mismatch", sum(r["res_got"] != r["res_exp"] for r in mine))
