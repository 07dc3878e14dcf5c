"""The oracle's restatement of the reference's SEARCH CONTROL (oracle/ks265_me_ref.c: interMeDia enc@0x48fbe0, interMeHex enc@0x48fde0,
interMeUMH enc@0x4907b0) replayed on traces recorded from the reference binary itself (tests/golden/me_search.npz, written by
oracle/ref_probe/gen_me_traces.py: real `appencoder` runs with the three functions hooked).  Every recorded call must come out with the
reference's motion vector, cost and convergence flag."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle_lib import lib, ptr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "me_search.npz")


def replay_all():
    z = np.load(GOLD)
    f = {n: i for i, n in enumerate(z["call_fields"])}
    calls, cm, fenc, org = z["calls"], np.ascontiguousarray(z["cm"]), np.ascontiguousarray(z["fenc"]), z["plane_org"]
    planes = [np.ascontiguousarray(z[f"plane{k}"]) for k in range(len(org))]
    o = lib()
    res = []
    for c in calls:
        g = lambda n: int(c[f[n]])
        pl = planes[g("plane")]
        nx, ny = g("xhi") - g("xlo") + 1, g("yhi") - g("ylo") + 1
        cmx, cmy = cm[g("cm_off"):g("cm_off") + nx], cm[g("cm_off") + nx:g("cm_off") + nx + ny]
        n = (1 << g("l2w")) << g("l2h")
        fe = fenc[g("fenc_off"):g("fenc_off") + n]
        lim = (C.c_int * 4)(g("xmin"), g("xmax"), g("ymin"), g("ymax"))
        out = (C.c_int32 * 4)()
        rc = o.kso_me_replay(g("method"), ptr(fe), g("l2w"), g("l2h"), ptr(pl), int(org[g("plane")][0]), int(org[g("plane")][1]), pl.shape[1], pl.shape[0],
                             g("pux"), g("puy"), ptr(cmx), g("xlo"), g("xhi"), ptr(cmy), g("ylo"), g("yhi"), g("merange"), g("range_shift"), lim,
                             g("skip_cross"), g("use_had"), g("sx"), g("sy"), C.c_uint32(g("cost0")), out)
        exp = (g("out_x"), g("out_y"), g("out_cost"), g("out_flag"))
        res.append((g("method"), rc, tuple(out), exp, (g("sx"), g("sy"))))
    return res


def test_search_control_matches_reference_traces():
    res = replay_all()
    for m, name in enumerate(("interMeDia", "interMeHex", "interMeUMH")):
        mine = [r for r in res if r[0] == m]
        ok = [r for r in mine if r[1] == 0]
        bad = [r for r in ok if r[2] != r[3]]
        moved = [r for r in ok if r[3][:2] != r[4]]
        assert len(ok) >= 200, f"{name}: only {len(ok)} replayable cases"
        assert len(moved) >= 40, f"{name}: only {len(moved)} cases in which the search moved"
        assert not bad, f"{name}: {len(bad)} of {len(ok)} cases differ, first (got, reference, start) = {bad[0][2:]}"


if __name__ == "__main__":
    res = replay_all()
    for m in range(3):
        mine = [r for r in res if r[0] == m]
        print(m, "cases", len(mine), "replayable", sum(r[1] == 0 for r in mine), "mismatch", sum(r[1] == 0 and r[2] != r[3] for r in mine),
              "moved", sum(r[3][:2] != r[4] for r in mine))
        for r in [r for r in mine if r[1] == 0 and r[2] != r[3]][:5]:
            print("   ", r)


def test_mvd_cost_table_matches_reference_slices():
    """createMvdCostTable enc@0x48b850: every p_cost_mvx / p_cost_mvy slice the trace shim recorded inside the reference (1 960 slices, 7 lambdas) is
    lambda x kso_mvd_bits(4 x - mvp) for one integer lambda and one quarter-pel predictor, over the table's own span |4 x - mvp| <= 4 * 64 + 16
    (beyond it the shim recorded the neighbouring table).  The pipeline oracle's vector rate is this function (se_bits -> kso_mvd_bits)."""
    z = np.load(GOLD)
    f = {n: i for i, n in enumerate(z["call_fields"])}
    cm = np.ascontiguousarray(z["cm"])
    o = lib()
    assert [o.kso_mvd_bits(d) for d in (0, 1, -1, 2, -2, 3, 4, -4, 272, -272)] == [1, 3, 3, 5, 5, 5, 7, 7, 19, 19]
    lambdas, fractional, entries = set(), 0, 0
    for c in z["calls"]:
        g = lambda n: int(c[f[n]])
        nx, ny = g("xhi") - g("xlo") + 1, g("yhi") - g("ylo") + 1
        for sl, lo in ((cm[g("cm_off"):g("cm_off") + nx], g("xlo")), (cm[g("cm_off") + nx:g("cm_off") + nx + ny], g("ylo"))):
            k, mn = int(sl.argmin()), int(sl.min())
            hit = None
            pos = 4 * (lo + np.arange(len(sl)))

            def fits(lam, mvp):
                mine = np.zeros(len(sl), np.uint16)
                o.kso_mvd_cost_slice(lam, mvp, lo, lo + len(sl) - 1, ptr(mine))
                inside = np.abs(pos - mvp) <= 4 * 64 + 16
                return (lam, mvp, int(inside.sum())) if inside.sum() >= 16 and (mine[inside] == sl[inside]).all() else None

            for b in range(1, 21, 2):                               # the smallest entry is lambda x bits(d) for some odd code length
                if mn % b:
                    continue
                near = range(4 * (lo + k) - 3, 4 * (lo + k) + 4) if b <= 5 else range(4 * lo - 300, 4 * (lo + len(sl)) + 300)   # predictor inside / outside the slice
                for mvp in near:
                    hit = fits(mn // b, mvp)
                    if hit:
                        break
                if hit:
                    break
            assert hit, f"slice of call {g('method')} at {lo}: no (lambda, mvp) reproduces it"
            lambdas.add(hit[0]); fractional += (hit[1] & 3) != 0; entries += hit[2]
    assert len(lambdas) >= 5 and fractional >= 100 and entries > 200000, (lambdas, fractional, entries)
