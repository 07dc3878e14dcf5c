"""The oracle's restatement of the reference's lookahead cost function (oracle/ks265_lookahead_ref.c: calcFrameCost enc@0x4a7410 - SURVEY.md 8(f) rank 2, the core of row f2 - and
the cuTree finish inlined in CInputPicManage::updateQueue enc@0x480964..0x480a54) replayed on calls recorded inside the reference binary (tests/golden/calc_frame_cost.npz, written by
oracle/ref_probe/gen_cfc_traces.py: nine real `appencoder` runs - ultrafast .. slow, -rc 1 / 2 / 3, -bframes 0 / 3 / 7, cuTree on and off, 208x128 .. 832x480, a clip with hard cuts,
flat pictures and a still; the stream checked to be unchanged by the hooks).  Bit-exact: every per-block array (vectors, list costs, list bits, intra cost and mode, inter cost), the
picture sums, the motion statistics and the return value; the QP offsets of the finish are doubles and must be equal, not close."""
from __future__ import annotations

import os

import numpy as np

from cfc_cases import load_fixture, replay_call, replay_finish, set_bits_table
from oracle_lib import lib


def test_calc_frame_cost_matches_reference_traces():
    o = lib(); set_bits_table(o)
    runs, calls, _ = load_fixture()
    assert len(runs) >= 9 and len(calls) >= 120
    kinds = set()
    for i, r in enumerate(calls):
        bad = [b for b in replay_call(o, r) if not b.startswith("(")]
        assert not bad, f"call {i} ({runs[r['run']]}; poc {r['h'][14]}, d0 {r['h'][3]}, d1 {r['h'][4]}): {bad}"
        h = r["h"]
        kinds.add((int(h[3]) > 0, int(h[4]) > 0, int(h[19]), int(h[21]) != 0, int(h[26]) != 0))
    # intra-only, P and B calls; 8x8 and 16x16 blocks; full and fast intra; with and without the B pictures' intra comparison
    assert {k[:2] for k in kinds} >= {(False, False), (True, False), (True, True)}
    assert {k[2] for k in kinds} == {3, 4} and {k[3] for k in kinds} == {False, True} and {k[4] for k in kinds} == {False, True}
    moved = sum(int((r["a_mv0"] != 0).sum()) for r in calls)
    assert moved > 2000, "the recorded searches move"


def test_calc_frame_cost_reuses_stored_vectors():
    """a call whose list-0 vectors exist already (same distance, other list-1 picture) must read them, not search again: the fixture holds such calls"""
    _, calls, _ = load_fixture()
    reuse = [r for r in calls if r["h"][3] > 0 and not r["h"][32]]
    assert len(reuse) >= 3
    for r in reuse:
        assert (r["b_mv0"] == r["a_mv0"]).all() and (r["b_c0"] == r["a_c0"]).all()


def test_cutree_finish_matches_reference_traces():
    o = lib()
    _, _, fin = load_fixture()
    assert len(fin) >= 40
    refd = 0
    for i, r in enumerate(fin):
        assert replay_finish(o, r), f"finish record {i} (poc {r['h'][3]})"
        refd += bool(r["h"][5])
    assert refd >= 25
    spread = max(float(np.ptp(r["out"])) for r in fin if r["h"][5])
    assert spread > 3.0, "the offsets are not trivial"


def test_cutree_pass_reproduces_whole_reference_runs():
    """the COMPOSITION: tests/cutree_mirror.py - the host's cuTree pass (ks265_enc.c ct_run: which picture is costed against which, the window behind a mini-GOP, propagation in reverse coding
    order, the finish and its end-of-window doubling) written with the oracle's pinned pieces - against the offsets the REFERENCE left for every picture of whole CLI runs
    (tests/golden/cutree_run.npz: L+0x9b0 of each picture when it leaves the lookahead, recorded by oracle/ref_probe/gen_cfc_traces.py; the clips are make_clip's, regenerated here).
    Exact (doubles equal) for config 4's GOP - -preset slow -rc 3 -bframes 3, with and without -aq, whole-clip and depth-limited (-lookahead 20) windows; for -bframes 7 the reference's
    adaptive anchor placement and for veryfast two pictures differ: held to what the fixture recorded"""
    from cutree_mirror import CuTree
    from ks265codec_amd.synth import make_clip
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cutree_run.npz"))
    names = [str(n) for n in z["names"]]
    assert {"crf_b3_33", "crf_b3_aq_25", "crf_b3_la20_41"} <= set(names)
    for name in names:
        W, H, n = (int(v) for v in z[name + "_size"])
        kw = eval(str(z[name + "_kw"]), {"dict": dict, "True": True, "False": False})
        clip = make_clip(W, H, n, seed=11, abc=(17, 23, 9), pan=(3, 2))
        ct = CuTree(clip, W, H, **kw); ct.run()
        off, rec = z[name + "_off"], z[name + "_same"]
        same = np.array([bool((ct.maps_qoff[t] == off[t]).all()) for t in range(n)])
        if name.startswith("crf_b3"):
            assert same.all(), f"{name} ({z[name + '_args']}): pictures {np.nonzero(~same)[0].tolist()} differ from the reference's offsets"
            assert float(np.abs(off).max()) > 3.0 and (off[z[name + "_isref"] == 0] == 0).all() == ("aq" not in name)
        else:
            assert (same | ~rec).all(), f"{name}: pictures {np.nonzero(~same & rec)[0].tolist()} matched when the fixture was written and do not now"
