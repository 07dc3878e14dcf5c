"""CPU: the encoder host (ks265codec_amd/host/ks265_enc.c + ks265_stream.c) linked against tests/hip_stub.c, a stand-in for the device library that hands out fixed,
valid records whose content depends on WHICH pictures met in WHICH order.  No pixel is encoded here; what is tested is everything around the GPU: input slots and
back-pressure, the scheduler / dispatcher / writer threads, rings that fill up, GOP structures, flush, GOP lanes (output order, full rings, wake-ups), the graph cache
(replay == launch by launch, capture failures), and that what the writer emits is a stream the reference decoder accepts.  Every run is a process under a time limit."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_DEC = os.path.join(ROOT, "oracle", "_ref", "appdecoder")


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    from oracle_lib import build_oracle
    build_oracle()
    d = tmp_path_factory.mktemp("stubenc")
    so = str(d / "libks265enc_stub.so")
    host = os.path.join(ROOT, "ks265codec_amd", "host")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so,
                           os.path.join(host, "ks265_enc.c"), os.path.join(host, "ks265_stream.c"), os.path.join(HERE, "hip_stub.c"),
                           "-L", os.path.join(ROOT, "oracle"), "-lks265_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-lm"])
    return so


def run(stub_lib, n, iper, bframes, W=128, H=72, out=None, timeout=120, **env):
    e = dict(os.environ, KS265_STUB_LIB=stub_lib, **{k: str(v) for k, v in env.items()})
    args = [sys.executable, os.path.join(HERE, "host_driver.py"), ROOT, str(n), str(iper), str(bframes), str(W), str(H)] + ([str(out)] if out else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-1200:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("bframes", [0, -1, 3])
def test_lanes_hand_out_the_one_lane_stream(stub_lib, tmp_path, bframes):
    """IPPP, hierarchical B (8), P + 3 B: the scheduler closes every GOP in front of the next key picture, so GOPs can be coded on different lanes"""
    res = {L: run(stub_lib, 150, 32, bframes, out=tmp_path / f"l{L}.265", KS265_GOP_LANES=L) for L in (1, 2, 3)}
    for L, r in res.items():
        assert r["lanes"] == L and r["vcl"] == 150 and r["idr"] == 5 and sorted(r["pts"]) == list(range(150)), (L, r["vcl"], r["idr"])
        assert (r["pts"] == list(range(150))) == (bframes == 0)
    assert res[1]["md5"] == res[2]["md5"] == res[3]["md5"]
    if bframes == 0:
        assert res[2]["maxdelay"] > 32                                       # the second GOP is held back until the first one has left
    if os.path.exists(REF_DEC):                                              # the records of the stand-in make a stream the reference decoder accepts
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "l2.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
        assert d.returncode == 0 and os.path.getsize(tmp_path / "d.yuv") == 150 * 128 * 72 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]


@pytest.mark.parametrize("n,iper", [(700, 300), (420, 140)])
def test_lanes_with_gops_longer_than_the_ring(stub_lib, n, iper):
    """64x64 pictures: the ring holds 128 pictures, a GOP more - a lane that may not hand out yet fills up completely (scheduler waiting for ring space, input waiting
    for slots) and must come back to life when its turn comes"""
    res = {L: run(stub_lib, n, iper, 0, W=64, H=64, KS265_GOP_LANES=L) for L in (1, 2, 3)}
    assert res[1]["md5"] == res[2]["md5"] == res[3]["md5"] and all(r["pts"] == list(range(n)) for r in res.values())


@pytest.mark.parametrize("bframes", [0, -1, 3])
def test_graph_replay_equals_launch_by_launch(stub_lib, bframes):
    """IPPP, hierarchical B (8) and P + 3 B: replay of captured pictures, plain launches, a runtime without capture and one whose instantiation fails all write the same
    stream (the last two fall back for good after the first attempt)"""
    iper = 64
    a = run(stub_lib, 170, iper, bframes)                                   # (graphs are opt-in since round 4: KS265_GRAPH=1)
    assert a["vcl"] == 170 and sorted(a["pts"]) == list(range(170))
    for env in ({"KS265_GRAPH": 1}, {"KS265_GRAPH": 1, "KS265_STUB_NO_CAPTURE": 1}, {"KS265_GRAPH": 1, "KS265_STUB_NO_INSTANTIATE": 1}):   # the last: completion by a word in pinned memory
        b = run(stub_lib, 170, iper, bframes, **env)
        assert b["md5"] == a["md5"], env
    if bframes:
        assert a["pts"] != list(range(170)) and a["maxdelay"] >= 2           # coding order differs from display order


@pytest.mark.parametrize("bframes,iper,n", [(-1, 64, 170), (-1, 20, 90), (3, 48, 150), (-1, 0, 75)])
def test_anchor_lane_writes_the_same_stream(stub_lib, bframes, iper, n):
    """pyramid GOPs: the anchors' P chain on a stream, frame object and DPB slots of its own (round 5) or on the main stream - the same stream, with long and short intra
    periods (key pictures on their own stream / on the main stream), mini-GOPs cut short in front of key pictures and at the flush, and key-picture requests in between"""
    for extra in ({}, {"KS_TEST_KEYREQ": 1}, {"KS265_GOP_LANES": 2}):
        on = run(stub_lib, n, iper, bframes, KS265_STUB_B_STATELESS=1, KS265_ANCHOR_LANE=1, **extra)
        off = run(stub_lib, n, iper, bframes, KS265_STUB_B_STATELESS=1, **extra)
        assert on["vcl"] == off["vcl"] == n and on["md5"] == off["md5"], (bframes, iper, extra)
        assert sorted(on["pts"]) == list(range(n))


def test_key_frame_requests(stub_lib):
    """QY265EncoderKeyFrameRequest in the middle of GOPs: one lane codes the next scheduled picture as a key picture, lanes open a new GOP on the next lane; all
    pictures come out, in order, with the extra key pictures"""
    res = {L: run(stub_lib, 100, 32, 0, KS265_GOP_LANES=L, KS_TEST_KEYREQ=1) for L in (1, 2, 3)}
    for L, r in res.items():
        assert r["vcl"] == 100 and r["pts"] == list(range(100)) and r["idr"] == 5, (L, r["idr"])    # 0, the requested 18, 19, 41, and 73 = 41 + the period
    assert res[1]["md5"] == res[2]["md5"] == res[3]["md5"]                  # the request travels with the next picture: no dependence on the scheduler's lag
    for bframes in (-1, 3):                                                 # with B pictures the mini-GOP in front of the requested key picture is shortened
        a, b = run(stub_lib, 100, 32, bframes, KS_TEST_KEYREQ=1), run(stub_lib, 100, 32, bframes, KS_TEST_KEYREQ=1, KS265_GRAPH=1)
        assert a["vcl"] == 100 and sorted(a["pts"]) == list(range(100)) and a["idr"] == 5 and a["md5"] == b["md5"], (bframes, a["idr"])
        c = run(stub_lib, 100, 32, bframes, KS_TEST_KEYREQ=1, KS265_GOP_LANES=2)     # lanes: the GOP that ends early is told so (its lane schedules what it has)
        assert c["md5"] == a["md5"]


def test_no_device_means_no_encoder(stub_lib):
    """the host has no CPU path of its own: when the device library reports no device, QY265EncoderOpen fails"""
    e = dict(os.environ, KS265_STUB_LIB=stub_lib, KS265_STUB_NO_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "host_driver.py"), ROOT, "3", "32", "0", "64", "64"], capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode != 0 and "AssertionError" in r.stderr


def test_strided_input_and_reconfigure(stub_lib):
    """planes with padded rows are copied row by row (same stream as packed planes); QY265EncoderReconfig changes the QP in the middle of the stream: new graph keys,
    all pictures still come out"""
    a, b = run(stub_lib, 60, 32, 0), run(stub_lib, 60, 32, 0, KS_TEST_STRIDE=1)
    assert a["md5"] == b["md5"]
    c = run(stub_lib, 130, 64, 0, KS_TEST_RECONFIG=1)
    d = run(stub_lib, 130, 64, 0, KS_TEST_RECONFIG=1, KS265_GRAPH=1)
    assert c["vcl"] == 130 and c["md5"] == d["md5"] and c["md5"] != run(stub_lib, 130, 64, 0)["md5"]


@pytest.mark.parametrize("lanes", [1, 2])
@pytest.mark.parametrize("at", [0, 1, 5])
def test_device_failure_is_reported_not_hung(stub_lib, lanes, at):
    """the at-th picture fails on the device (a launch error): QY265EncoderEncodeFrame returns QY_FAIL within a few calls - also when the caller had run 17 pictures
    ahead of the scheduler thread and was waiting for it - and QY265EncoderClose comes back"""
    r = run(stub_lib, 100, 32, 0, timeout=60, KS265_GOP_LANES=lanes, KS265_STUB_FAIL_AT=at, KS_TEST_EXPECT_ERROR=1)
    # (GOP lanes buffer a whole GOP of input per lane beyond the pictures in flight: the caller may have fed all of this short clip before the failed picture is through)
    assert r.get("error") == 0x80000001 and r["at"] <= (at + 40 if lanes == 1 else 100), r


@pytest.mark.parametrize("bframes,lanes", [(0, 1), (-1, 1), (0, 2), (-1, 2), (-1, 3)])
def test_device_error_word_sticks_until_the_next_key_picture(stub_lib, tmp_path, bframes, lanes):
    """ks265_take_device_error (a wavefront time-out on the GPU) does not say which picture raised it, and every picture predicted from a broken one is broken: after the
    error is seen NO picture goes out until a key picture submitted later starts a clean GOP; a caller that keeps feeding pictures after QY_FAIL gets the
    failure for every such picture, then a stream that starts again with an IDR and decodes"""
    out = tmp_path / "e.265"
    # (round 5: with GOP lanes - two by default for the pyramid GOPs - the failed GOP's pictures leave their lane without payload and are counted, the other lanes' GOPs go on:
    #  before, the handle waited for ever for the pictures that were never written)
    clean = run(stub_lib, 200, 32, bframes, KS265_GOP_LANES=lanes)
    r = run(stub_lib, 200, 32, bframes, out=out, KS265_STUB_DEVERR_AT=9, KS_TEST_CONTINUE_ON_ERROR=1, KS265_GOP_LANES=lanes)
    assert r["errors"] >= 1
    order = clean["pts"]                                         # coding order of the clean run
    lost = [p for p in order if p not in set(r["pts"])]
    assert lost, "the failed picture did not go out"
    first = order.index(lost[0])
    # from the first lost picture on (coding order) nothing goes out until a key picture, and from that key picture on everything does (how far the caller had run
    # ahead when the error was seen decides WHICH key picture: all pictures submitted by then fail)
    resumed = [p for p in order[first:] if p in set(r["pts"])]
    if lanes > 1:
        # GOPs are dealt to the lanes in turn and every lane recovers at ITS next key picture: a GOP goes out whole, or up to the picture where its lane saw the error and
        # nothing after it (how far the caller had run ahead decides how many of that lane's GOPs are hit); GOPs of the other lanes are not touched; output stays in stream order
        got = r["pts"]
        by_gop = {}
        for p in order:
            by_gop.setdefault(p // 32, []).append(p)
        pos = 0
        for g in sorted(by_gop):
            mine = [p for p in got if p // 32 == g]
            assert mine == by_gop[g][:len(mine)], (g, mine[:6], by_gop[g][:6])
            assert got[pos:pos + len(mine)] == mine, "GOPs out of stream order"
            pos += len(mine)
        hit = [g for g in by_gop if len([p for p in got if p // 32 == g]) < len(by_gop[g])]
        assert hit and all(g % lanes == hit[0] % lanes for g in hit), hit                   # one lane's GOPs only
        assert len([p for p in got if p // 32 == max(by_gop)]) == len(by_gop[max(by_gop)]) or max(by_gop) % lanes == hit[0] % lanes
    elif resumed:
        assert resumed[0] % 32 == 0, resumed[:4]
        k = order.index(resumed[0])
        assert order[k:] == r["pts"][len(r["pts"]) - len(order[k:]):], "the stream is not whole after the restart"
        assert all(p not in set(r["pts"]) for p in order[first:k])
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(out), "-o", str(tmp_path / "d.yuv"), "-threads", "1"], capture_output=True, text=True, cwd=tmp_path)
        assert "decoder passed" in d.stdout, d.stdout[-300:]


@pytest.mark.parametrize("lanes,bframes", [(1, 0), (2, 0), (1, -1)])
def test_close_without_flush(stub_lib, lanes, bframes):
    """QY265EncoderClose while pictures are waiting for the scheduler, in flight and half written: every thread is joined, nothing hangs"""
    r = run(stub_lib, 90, 32, bframes, timeout=60, KS265_GOP_LANES=lanes, KS_TEST_CLOSE_EARLY=1)
    assert r["closed_early"] and r["vcl"] < 90


def test_two_encoders_in_one_process(stub_lib):
    """two handles driven from two threads at once (the API calls release the GIL): no shared state between handles - both write the stream a lone encoder writes"""
    code = (
        "import sys, os, threading, subprocess, json\n"
        "sys.argv = ['d'] + sys.argv[1:]\n"
        "src = open(os.path.join(sys.argv[1], 'tests', 'host_driver.py')).read()\n"
        "import io, contextlib\n"
        "outs = [io.StringIO(), io.StringIO()]\n"
        "def work(k):\n"
        "    g = {'__name__': 'drv%d' % k}\n"
        "    import builtins\n"
        "    g['print'] = lambda *a, **kw: outs[k].write(' '.join(str(x) for x in a) + '\\n')\n"
        "    exec(compile(src, 'host_driver.py', 'exec'), g)\n"
        "ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n"
        "print(outs[0].getvalue().strip().splitlines()[-1]); print(outs[1].getvalue().strip().splitlines()[-1])\n")
    e = dict(os.environ, KS265_STUB_LIB=stub_lib)
    r = subprocess.run([sys.executable, "-c", code, ROOT, "120", "32", "0", "128", "72"], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-1200:]
    a, b = (json.loads(l) for l in r.stdout.strip().splitlines()[-2:])
    lone = run(stub_lib, 120, 32, 0)
    assert a["md5"] == b["md5"] == lone["md5"]


@pytest.mark.parametrize("bframes", [1, 4, 7, 15])
def test_every_bframes_value_opens_and_decodes(stub_lib, tmp_path, bframes):
    """-bframes 4..15 used to fail in QY265EncoderOpen (reorder depth >= DPB size in the SPS).  P + n non-reference B has a reorder depth of ONE whatever n
    is; the stream of every n decodes with the reference decoder, all pictures in display order"""
    r = run(stub_lib, 70, 48, bframes, out=tmp_path / "b.265")
    assert r["vcl"] == 70 and sorted(r["pts"]) == list(range(70)) and r["idr"] == 2
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "b.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
        assert d.returncode == 0 and os.path.getsize(tmp_path / "d.yuv") == 70 * 128 * 72 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]


@pytest.mark.parametrize("bframes,lean32,near32", [(-1, 16, 8), (3, 16, 8), (2, 1, 0), (0, 0, 0)])
def test_b_pictures_nothing_predicts_from_are_coded_lean(stub_lib, tmp_path, bframes, lean32, near32):
    """round 6: the host lowers the tools (ks265_frame_set_picture_tools) exactly for the B pictures nothing predicts from - half the pictures of a pyramid of 8: no intra
    candidates, no joint refinement, no SAO - and, keeping the refinement, for the reference B pictures whose own references are at most two pictures away (the layer above);
    every other picture runs the full set; a picture without SAO carries slice_sao_luma_flag = slice_sao_chroma_flag = 0 (the writer gets no SAO records) and the stream still
    decodes; KS265_LEAN_B=3 lowers the non-reference pictures alone, KS265_LEAN_B=0 none
    (run with the joint refinement on, as from -preset slower on, so that the two rules show apart in the stand-in's log)"""
    log = tmp_path / "tools.txt"
    r = run(stub_lib, 33, 128, bframes, out=tmp_path / "l.265", KS265_STUB_TOOLS_LOG=log, KS265_BI_REFINE=2)
    lines = [ln.split() for ln in open(log).read().splitlines()]
    assert len(lines) == 33 and r["vcl"] == 33
    full = [ln for ln in lines if ln[1] != "0"]
    tools0 = full[0][1:]
    lean = [ln for ln in lines if ln[1:4] == ["0", "0", "0"]]
    near = [ln for ln in lines if ln[1:4] == ["0", tools0[1], "0"]]
    assert len(full) + len(lean) + len(near) == 33 and all(k == "B" for k, *_ in lean + near), lines[:12]
    assert (len(lean), len(near)) == (lean32, near32) if bframes != 2 else len(lean) > 0 and not near, (len(lean), len(near), lines[:12])
    assert len({tuple(ln[1:]) for ln in full}) == 1 and tools0[2] != "0", full[:4]                   # everything else: the one full tool set, SAO on
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "l.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
        assert d.returncode == 0 and os.path.getsize(tmp_path / "d.yuv") == 33 * 128 * 72 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]
    os.remove(log)
    r3 = run(stub_lib, 33, 128, bframes, out=tmp_path / "n.265", KS265_STUB_TOOLS_LOG=log, KS265_LEAN_B=3, KS265_BI_REFINE=2)
    l3 = [ln.split() for ln in open(log).read().splitlines()]
    assert sum(ln[1:4] == ["0", "0", "0"] for ln in l3) == len(lean) and sum(ln[1:] == tools0 for ln in l3) == 33 - len(lean)
    assert (r3["md5"] != r["md5"]) == (len(near) > 0)
    os.remove(log)
    r0 = run(stub_lib, 33, 128, bframes, out=tmp_path / "f.265", KS265_STUB_TOOLS_LOG=log, KS265_LEAN_B=0, KS265_BI_REFINE=2)
    assert all(ln.split()[1:] == tools0 for ln in open(log).read().splitlines())
    assert (r0["md5"] != r["md5"]) == (len(lean) > 0)                                               # the slice headers differ where SAO is off


@pytest.mark.parametrize("bframes,ref", [(-1, 2), (-1, 4), (3, 3)])
def test_pyramid_b_pictures_with_several_references_per_list(stub_lib, tmp_path, bframes, ref):
    """round 5: -ref N with the pyramid GOPs - the B pictures' lists hold up to N of the pictures the mini-GOP keeps anyway (list 0 before, list 1 after the picture, nearest
    first): slice headers with num_ref_idx_active > 1 on both lists, reference picture sets, default list construction - the reference decoder takes the stream, every
    picture comes out, over several GOPs and a shortened last mini-GOP"""
    r = run(stub_lib, 75, 32, bframes, out=tmp_path / "m.265", KS_TEST_REF=ref, KS_TEST_LOOKAHEAD=0)
    assert r["vcl"] == 75 and sorted(r["pts"]) == list(range(75)) and r["idr"] == 3
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "m.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
        assert d.returncode == 0 and "decoder passed" in d.stdout and os.path.getsize(tmp_path / "d.yuv") == 75 * 128 * 72 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]


@pytest.mark.parametrize("bframes", [-1, 3])
def test_anchors_of_the_pyramid_search_ref0_past_anchors(stub_lib, tmp_path, bframes):
    """round 6 (VERDICT r5 missing 3): -ref0 N (qy265enc.h:142; every preset from superfast up resolves to 3) - an anchor of a pyramid GOP searches the last N anchors of its
    GOP: P slices with num_ref_idx_l0_active > 1, the older anchors kept in the reference picture sets of the B pictures in between.  The stand-in device predicts a
    multi-reference P picture from its FARTHEST picture, so a decoder that reproduces the pictures proves lists and sets; -ref0 1 is round 5's stream"""
    md = {}
    for ref0 in (1, 3, 4):
        r = run(stub_lib, 75, 40, bframes, out=tmp_path / f"a{ref0}.265", KS_TEST_REF0=ref0, KS_TEST_LOOKAHEAD=0)
        assert r["vcl"] == 75 and sorted(r["pts"]) == list(range(75)) and r["idr"] == 2
        md[ref0] = r["md5"]
        if os.path.exists(REF_DEC):
            d = subprocess.run([REF_DEC, "-b", str(tmp_path / f"a{ref0}.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
            assert d.returncode == 0 and "decoder passed" in d.stdout and os.path.getsize(tmp_path / "d.yuv") == 75 * 128 * 72 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]
    assert len(set(md.values())) == 3                                  # the lists differ, so do the stand-in's pictures
    assert run(stub_lib, 75, 40, bframes, KS_TEST_REF0=3, KS_TEST_LOOKAHEAD=0, KS265_GOP_LANES=2)["md5"] == md[3]      # lane-count invariant


@pytest.mark.parametrize("bframes", [0, -1])
def test_rdoq_tables_follow_the_stream_and_not_the_threads(stub_lib, bframes):
    """round 6 (VERDICT r5 missing 4): rdoq set by name ("rdoq" "1" = -rdoq 1) - every P / B picture gets the bit tables built from the context states of the latest picture of its
    kind coded at least 17 pictures earlier (else from the initial states of its slice): the stand-in device stamps what it was handed into the picture, so (1) the stream differs
    from the default's, (2) it does not depend on writer threads, GOP lanes or enqueue timing, (3) "rdoq" "0" and the presets' own rdoq = 1 are the default stream"""
    base = run(stub_lib, 90, 40, bframes, KS_TEST_LOOKAHEAD=0)
    assert run(stub_lib, 90, 40, bframes, KS_TEST_LOOKAHEAD=0, KS_TEST_RDOQ=0)["md5"] == base["md5"]
    a = run(stub_lib, 90, 40, bframes, KS_TEST_LOOKAHEAD=0, KS_TEST_RDOQ=1)
    assert a["md5"] != base["md5"] and a["vcl"] == 90
    assert run(stub_lib, 90, 40, bframes, KS_TEST_LOOKAHEAD=0, KS_TEST_RDOQ=1, KS265_GOP_LANES=2)["md5"] == a["md5"]
    assert run(stub_lib, 90, 40, bframes, KS_TEST_LOOKAHEAD=0, KS_TEST_RDOQ=1, KS265_STUB_EVENT_LAG=3)["md5"] == a["md5"]


@pytest.mark.parametrize("rc,bframes", [(2, 0), (1, -1)])
def test_rate_control_does_not_depend_on_thread_timing(stub_lib, rc, bframes):
    """the frame-level controller (rc 1 / 2 / 4) decides the QP offset of a mini-GOP from exactly the pictures coded RC_LAG earlier in coding order (the
    scheduler waits for those), so two runs - one of them with a single writer thread's worth of jitter (KS265_GRAPH changes the enqueue timing) - write the same
    bytes; a budget far below / above what the records cost moves the QP (the stand-in's records depend on the QP)"""
    a = run(stub_lib, 150, 64, bframes, KS_TEST_RC=rc, KS_TEST_BR=40)
    b = run(stub_lib, 150, 64, bframes, KS_TEST_RC=rc, KS_TEST_BR=40, KS265_GRAPH=1)
    c = run(stub_lib, 150, 64, bframes, KS_TEST_RC=rc, KS_TEST_BR=40)
    assert a["vcl"] == 150 and a["md5"] == b["md5"] == c["md5"]
    hi = run(stub_lib, 150, 64, bframes, KS_TEST_RC=rc, KS_TEST_BR=400000)
    assert hi["md5"] != a["md5"]                                             # the controller acts: another budget, another stream


def test_encode_headers_returns_three_parameter_sets(stub_lib):
    """QY265EncoderEncodeHeaders hands out VPS, SPS and PPS as three entries (naltype 32, 33, 34), each starting with its own start code"""
    r = run(stub_lib, 3, 32, 0, KS_TEST_HEADERS=1)
    assert [t for t, _, _ in r["hdr"]] == [32, 33, 34]
    assert all(size > 6 and head.startswith("00000001") for _, size, head in r["hdr"])
    assert [int(head[8:10], 16) >> 1 for _, _, head in r["hdr"]] == [32, 33, 34]     # nal_unit_type in the NAL header


def test_cli_prints_the_reference_per_picture_and_md5_lines(stub_lib, tmp_path):
    """B1 output contract (README.md:61-66, the lines appencoder -psnr 2 -md5 1 prints): header `poc slice bits psnr qp`, one tab-separated line per picture,
    `POC n MD5 y,u,v` in display order, the two `Total Frames:` lines and `bitrate, psnr:` (the line encoderwrapper.c:249-277 parses)"""
    import re
    import numpy as np
    host = os.path.join(ROOT, "ks265codec_amd", "host")
    exe = str(tmp_path / "ks265enc_stub")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(host, "ks265_cli.c"),
                           os.path.join(host, "ks265_enc.c"), os.path.join(host, "ks265_stream.c"), os.path.join(HERE, "hip_stub.c"),
                           "-L", os.path.join(ROOT, "oracle"), "-lks265_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-lm"])
    W, H, N = 128, 72, 12
    np.random.default_rng(5).integers(0, 256, N * W * H * 3 // 2, dtype=np.uint8).tofile(tmp_path / "in.yuv")
    (tmp_path / "enc.cfg").write_text("# options of the run\nqp 30\n-iper = 64\nbframes : 3\n")
    r = subprocess.run([exe, "-i", str(tmp_path / "in.yuv"), "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-preset", "veryfast", "-rc", "0", "-c", str(tmp_path / "enc.cfg"),
                        "-psnr", "2", "-md5", "1", "-fixqp", "1", "-df", "0", "-threads", "3", "-b", str(tmp_path / "o.265")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "H265 encoder passed!!!" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
    lines = r.stdout.splitlines()
    assert "poc\tslice\tbits\tpsnr\t\t\tqp" in lines
    pics = [re.fullmatch(r"(\d+)\t([IPB])\t(\d+)\t([\d.]+)\t([\d.]+)\t([\d.]+)\t(\d+)", ln) for ln in lines]
    pics = [m for m in pics if m]
    assert len(pics) == N and sorted(int(m.group(1)) for m in pics) == list(range(N))
    assert {m.group(2) for m in pics} == {"I", "P", "B"}                      # -bframes 3 came from the -c file
    assert {int(m.group(7)) for m in pics} == {30}                           # -fixqp 1: no per-layer offsets; qp 30 from the -c file
    md5s = [re.fullmatch(r"POC (\d+) MD5 ([0-9a-f]{32}),([0-9a-f]{32}),([0-9a-f]{32})", ln) for ln in lines]
    md5s = [m for m in md5s if m]
    assert [int(m.group(1)) for m in md5s] == list(range(N))                 # display order
    assert any(re.fullmatch(r"Total Frames: 12, test time: \d+ms, FPS: [\d.]+", ln) for ln in lines)
    assert any(re.fullmatch(r"Total Frames: 12, pure encoding time: \d+ms, [\d.]+ fps", ln) for ln in lines)
    assert any(re.fullmatch(r"bitrate, psnr: [\d.]+\t[\d.]+\t[\d.]+\t[\d.]+", ln) for ln in lines), [ln for ln in lines if "psnr" in ln]


def test_qp_ladders_of_the_host(stub_lib, tmp_path):
    """the QP every picture is coded with (-psnr 2 prints it): IPPP = the reference's cascade (key picture Q, P pictures Q + 1 + {0, 2, 1, 2}[position in the GOP & 3]:
    appencoder -bframes 0 -qp 27 codes 27 / 30 / 29 / 30 / 28 ..), the pyramid of the default GOP = Q / Q + 1 for the anchors, + 2 / + 4 / + 4 by B layer (the reference's 29 / 31 / 31), the pyramid of 4 of -bframes 3 + 2 / + 3, plain B
    pictures (-bframes 2) Q + 2; the GPU fixtures (tests/stream_cases.py HOST_IPPP_CASCADE, tools/rd_eval.py --host) mirror exactly this"""
    import re
    import numpy as np
    from stream_cases import HOST_IPPP_CASCADE
    host = os.path.join(ROOT, "ks265codec_amd", "host")
    exe = str(tmp_path / "ks265enc_stub")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(host, "ks265_cli.c"),
                           os.path.join(host, "ks265_enc.c"), os.path.join(host, "ks265_stream.c"), os.path.join(HERE, "hip_stub.c"),
                           "-L", os.path.join(ROOT, "oracle"), "-lks265_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-lm"])
    W, H, N = 128, 72, 21
    np.random.default_rng(5).integers(0, 256, N * W * H * 3 // 2, dtype=np.uint8).tofile(tmp_path / "in.yuv")

    def qps(*extra):
        r = subprocess.run([exe, "-i", str(tmp_path / "in.yuv"), "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-preset", "slow", "-rc", "0", "-qp", "27", "-iper", "12",
                            "-psnr", "2", "-threads", "3", "-b", str(tmp_path / "o.265"), *extra], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "H265 encoder passed!!!" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
        got = {int(a): (k, int(q)) for a, k, q in re.findall(r"^(\d+)\t([IPB])\t\d+\t[\d.]+\t[\d.]+\t[\d.]+\t(\d+)$", r.stdout, re.M)}
        assert sorted(got) == list(range(N))
        return got
    ippp = qps("-bframes", "0")
    for t in range(N):
        p = t % 12                                                           # position in the GOP (key period 12)
        assert ippp[t] == (("I", 27) if p == 0 else ("P", 28 + HOST_IPPP_CASCADE[p & 3])), (t, ippp[t])
    assert [ippp[t][1] for t in range(1, 9)] == [30, 29, 30, 28, 30, 29, 30, 28]
    hier = qps()                                                             # the default GOP: pyramid of 8 (the GOP of 12 ends with a mini-GOP of 3 = anchor + two plain B pictures)
    assert [hier[t] for t in range(0, 9)] == [("I", 27), ("B", 31), ("B", 31), ("B", 31), ("B", 29), ("B", 31), ("B", 31), ("B", 31), ("P", 28)], [hier[t] for t in range(9)]
    assert [hier[t] for t in (9, 10, 11)] == [("B", 29), ("B", 29), ("P", 28)]
    # -bframes 3: a pyramid of 4 like the reference's (appencoder -bframes 3 -qp 27 -psnr 2: 28 / 29 / 30 / 30 - anchor, the middle picture a reference B, the outer ones);
    # -bframes 2: P + 2 plain B pictures at Q + 2
    pyr4 = qps("-bframes", "3")
    assert [pyr4[t] for t in range(0, 9)] == [("I", 27), ("B", 30), ("B", 29), ("B", 30), ("P", 28), ("B", 30), ("B", 29), ("B", 30), ("P", 28)]
    flat = qps("-bframes", "2")
    assert [flat[t] for t in range(0, 4)] == [("I", 27), ("B", 29), ("B", 29), ("P", 28)]


def test_gop_structure_and_qp_ladders_are_the_reference_s(tmp_path):
    """the scheduler's coding order and QP per picture at constant QP against what the REFERENCE does (tests/golden/ref_gop_structure.json: its own -psnr 2 lines on a 26-picture clip,
    written by oracle/ref_probe/gen_gop_structure.py): IPPP - order and the cascade 30 / 29 / 30 / 28; -bframes 3 - the pyramid of 4, order 4 2 1 3 and 28 / 29 / 30 / 30; -bframes 1;
    the default GOP and -bframes 7 - the pyramid of 8 with 28 / 29 / 31 / 31 per layer (the reference codes a block depth first, ours breadth first: QP per picture compared; after its
    first block the reference's adaptive decision takes over, ours is switched off here)"""
    import json
    import re
    import numpy as np
    ref = json.load(open(os.path.join(HERE, "golden", "ref_gop_structure.json")))
    host = os.path.join(ROOT, "ks265codec_amd", "host")
    exe = str(tmp_path / "ks265enc_stub")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(host, "ks265_cli.c"),
                           os.path.join(host, "ks265_enc.c"), os.path.join(host, "ks265_stream.c"), os.path.join(HERE, "hip_stub.c"),
                           "-L", os.path.join(ROOT, "oracle"), "-lks265_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-lm"])
    W, H, N = 128, 72, ref["clip"]["pictures"]
    np.random.default_rng(5).integers(0, 256, N * W * H * 3 // 2, dtype=np.uint8).tofile(tmp_path / "in.yuv")

    def ours(extra):
        r = subprocess.run([exe, "-i", str(tmp_path / "in.yuv"), "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-preset", ref["preset"], "-rc", "0", "-qp", str(ref["qp"]), "-iper", "128",
                            "-psnr", "2", "-threads", "3", "-lookahead", "0", "-b", str(tmp_path / "o.265"), *extra], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "H265 encoder passed!!!" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
        return [(int(a), k, int(q)) for a, k, q in re.findall(r"^(\d+)\t([IPB])\t\d+\t[\d.]+\t[\d.]+\t[\d.]+\t(\d+)$", r.stdout, re.M)]
    for name, full_until, same_order in (("ippp", N, True), ("bframes3", 24, True), ("bframes1", 24, True), ("default", 8, False), ("bframes7", 8, False)):
        want = [(a, q) for a, _, q in ref["cases"][name]["coding_order"] if a <= full_until]
        got = [(a, q) for a, _, q in ours(ref["cases"][name]["args"]) if a <= full_until]
        assert dict(got) == dict(want), (name, sorted(set(got) ^ set(want)))
        if same_order:
            assert got == want, (name, got[:12], want[:12])
    kinds = {a: k for a, k, _ in ours(["-bframes", "3"])}
    assert [kinds[a] for a in range(9)] == ["I", "B", "B", "B", "P", "B", "B", "B", "P"]      # (the reference's anchors are generalised B pictures: 'B' in its lines)


def test_gops_dealt_to_several_gpus_behind_one_handle(stub_lib, tmp_path):
    """one handle, N GPUs - KS265_GPUS = N (the CLI's -gpus N) or KS265_DEVICES = list makes every GPU a GOP lane (closed GOPs, no data-path
    collective: SURVEY.md 8e); the stream is byte for byte the one-GPU stream, every listed device gets a context, a device the box does not have fails the open"""
    one = run(stub_lib, 200, 32, 0, out=tmp_path / "g1.265")
    for env in ({"KS265_GPUS": 2}, {"KS265_GPUS": 4}, {"KS265_DEVICES": "3,5"}, {"KS265_GPUS": 2, "KS265_GOP_LANES": 2}):
        e = dict(os.environ, KS265_STUB_LIB=stub_lib, KS265_STUB_LOG_DEVICES="1", **{k: str(v) for k, v in env.items()})
        r = subprocess.run([sys.executable, os.path.join(HERE, "host_driver.py"), ROOT, "200", "32", "0", "128", "72"], capture_output=True, text=True, timeout=120, env=e)
        assert r.returncode == 0, r.stdout[-400:] + r.stderr[-800:]
        res = json.loads(r.stdout.strip().splitlines()[-1])
        devs = {int(ln.rsplit(" ", 1)[1]) for ln in r.stderr.splitlines() if ln.startswith("stub: context on device")}
        want = {3, 5} if "KS265_DEVICES" in env else set(range(env["KS265_GPUS"]))
        assert devs == want, (env, devs)
        assert res["lanes"] == len(want) * int(env.get("KS265_GOP_LANES", 1)) and res["md5"] == one["md5"] and res["pts"] == list(range(200)), env
    for bframes in (-1, 3):                                                  # B-picture GOPs shard the same way
        a, b = run(stub_lib, 150, 32, bframes), run(stub_lib, 150, 32, bframes, KS265_GPUS=3)
        assert b["lanes"] == 3 and a["md5"] == b["md5"]
    e = dict(os.environ, KS265_STUB_LIB=stub_lib, KS265_GPUS="2", KS265_STUB_DEVICES="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "host_driver.py"), ROOT, "40", "32", "0", "128", "72"], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["lanes"] == 1      # the second GPU is not there: the handle goes on with the lanes it has


def test_rate_controlled_jobs_over_several_gpus(stub_lib, tmp_path):
    """VERDICT r3 #9 - BASELINE configs 4 / 5 name 8 GPUs with -rc 3 / -rc 1.  -rc 3 (CRF: a constant QP ladder on crf, nothing carried across GOPs) over 8 GPUs is byte for
    byte the one-GPU stream; -rc 1 (a bitrate target) deals the GOPs to lanes that each run their own controller on the same per-picture budget (a host-side
    bit-budget split, no collective): deterministic for a lane count, decodes, every picture there in display order.  8 lanes also exercise the writer-thread budget
    (threads / lanes, at least 2 per lane)."""
    one = run(stub_lib, 300, 32, -1, KS_TEST_RC=3)
    for env in ({"KS265_GPUS": 8}, {"KS265_GPUS": 2, "KS265_GOP_LANES": 2}):
        r = run(stub_lib, 300, 32, -1, KS_TEST_RC=3, **env)
        assert r["lanes"] == env["KS265_GPUS"] * env.get("KS265_GOP_LANES", 1) and r["md5"] == one["md5"], env
    a = run(stub_lib, 300, 32, 0, out=tmp_path / "rc1.265", KS_TEST_RC=1, KS_TEST_BR=300, KS265_GPUS=8)
    b = run(stub_lib, 300, 32, 0, KS_TEST_RC=1, KS_TEST_BR=300, KS265_GPUS=8)
    assert a["lanes"] == 8 and a["md5"] == b["md5"] and sorted(a["pts"]) == list(range(300))
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "rc1.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "1"], capture_output=True, text=True, cwd=tmp_path)
        assert "decoder passed" in d.stdout and os.path.getsize(tmp_path / "d.yuv") == 300 * 128 * 72 * 3 // 2


@pytest.mark.parametrize("bframes", [0, -1, 3])
def test_scene_cut_starts_a_closed_gop(stub_lib, bframes):
    """-lookahead N: every input picture is compared with its predecessor on a stream of its own before the scheduler sees it; where prediction is not clearly cheaper
    than intra coding the picture becomes a key picture (the mini-GOP in front of it is shortened), elsewhere nothing changes"""
    plain = run(stub_lib, 60, 128, bframes, W=128, H=96, KS_TEST_CUTS="23,41")
    la = run(stub_lib, 60, 128, bframes, W=128, H=96, KS_TEST_CUTS="23,41", KS_TEST_LOOKAHEAD=8)
    assert plain["idr"] == 1 and la["idr"] == 3 and sorted(la["pts"]) == list(range(60)), (plain["idr"], la["idr"])
    assert la["md5"] == run(stub_lib, 60, 128, bframes, W=128, H=96, KS_TEST_CUTS="23,41", KS_TEST_LOOKAHEAD=8, KS265_STUB_EVENT_LAG=3)["md5"]      # results that arrive late: same verdicts
    calm = run(stub_lib, 60, 128, bframes, W=128, H=96, KS_TEST_CUTS="1000", KS_TEST_LOOKAHEAD=8)                 # one scene: the analysis runs and finds nothing
    assert calm["idr"] == 1 and calm["md5"] == run(stub_lib, 60, 128, bframes, W=128, H=96, KS_TEST_CUTS="1000")["md5"]
    close = run(stub_lib, 60, 128, 0, W=128, H=96, KS_TEST_CUTS="20,23,26,40", KS_TEST_LOOKAHEAD=8)               # cuts closer than eight pictures to the last key picture are not key pictures
    assert close["idr"] == 3, close["idr"]


@pytest.mark.parametrize("bframes", [0, 3])
def test_adaptive_quantisation_streams_decode(stub_lib, tmp_path, bframes):
    """-aq 1 on the stand-in: every picture gets a QP per CTU (the oracle's restatement of calcFrameAdaptQuant on the source picture, the CTU rule), the parameter sets
    switch cu_qp_delta on, the slices carry the deltas - and the reference's decoder takes the stream (the stand-in's pictures are not real: what is checked here is the
    host's plumbing and the syntax; the pixels are tests/test_dqp.py and tests/test_gpu_rc.py)"""
    out = tmp_path / "aq.265"
    r = run(stub_lib, 40, 16, bframes, W=128, H=72, KS_TEST_AQ=1, out=out)
    plain = run(stub_lib, 40, 16, bframes, W=128, H=72)
    assert sorted(r["pts"]) == list(range(40)) and r["md5"] != plain["md5"]
    if os.path.exists(REF_DEC) and out.exists():
        d = subprocess.run([REF_DEC, "-b", str(out), "-o", str(tmp_path / "d.yuv"), "-threads", "1"], capture_output=True, text=True, cwd=tmp_path)
        assert "decoder passed" in d.stdout and os.path.getsize(tmp_path / "d.yuv") == 40 * 128 * 72 * 3 // 2, d.stdout[-300:]
    if bframes == 0:                                                       # GOP lanes (several GPUs per handle): every picture's map is its own - the one-lane stream byte for byte
        one = run(stub_lib, 100, 32, 0, W=128, H=72, KS_TEST_AQ=1)
        assert one["md5"] == run(stub_lib, 100, 32, 0, W=128, H=72, KS_TEST_AQ=1, KS265_GOP_LANES=3)["md5"]


def test_zero_latency_hands_out_every_picture_at_once(stub_lib):
    """-latency zerolatency (QY265LATENCY_ZERO, the SDK's live mode): every QY265EncoderEncodeFrame call returns with its own picture, nothing is delayed, no flush is needed;
    the stream is the one the pipelined modes write for the same GOP (IPPP)"""
    z = run(stub_lib, 60, 16, 0, KS_TEST_LATENCY="zerolatency")
    assert z["pts"] == list(range(60)) and z["maxdelay"] == 0 and z["flush_calls"] == 0, (z["maxdelay"], z["flush_calls"])
    assert z["md5"] == run(stub_lib, 60, 16, 0, KS_TEST_LATENCY="lowdelay")["md5"]


def test_scenecut_flag_runs_the_reference_rule(stub_lib):
    """-scenecut N (the reference's hidden flag): the scene-cut verdict is the rule of scenecut enc@0x47e9d0 (pinned: tests/test_lookahead_ref.py) on this lookahead's frame
    costs - no distance guard of its own: cuts three pictures apart are all key pictures; a calm clip has none"""
    la = run(stub_lib, 60, 128, 0, W=128, H=96, KS_TEST_CUTS="23,41", KS_TEST_LOOKAHEAD=8, KS_TEST_SCENECUT=40)
    assert la["idr"] == 3 and sorted(la["pts"]) == list(range(60)), la["idr"]
    close = run(stub_lib, 60, 128, 0, W=128, H=96, KS_TEST_CUTS="20,23,26,40", KS_TEST_LOOKAHEAD=8, KS_TEST_SCENECUT=40)
    assert close["idr"] == 5, close["idr"]
    calm = run(stub_lib, 60, 128, 0, W=128, H=96, KS_TEST_CUTS="1000", KS_TEST_LOOKAHEAD=8, KS_TEST_SCENECUT=40)
    assert calm["idr"] == 1 and calm["md5"] == run(stub_lib, 60, 128, 0, W=128, H=96, KS_TEST_CUTS="1000")["md5"]


@pytest.mark.parametrize("lanes", [1, 2])
def test_zero_copy_input_writes_the_same_stream(stub_lib, lanes):
    """ks265_enc_acquire_input: the caller produces every picture into one of the encoder's pinned buffers and hands that pointer in - nothing is copied, same stream"""
    a = run(stub_lib, 150, 32, 0, KS265_GOP_LANES=lanes)
    b = run(stub_lib, 150, 32, 0, KS265_GOP_LANES=lanes, KS_TEST_ZEROCOPY=1)
    assert b["zero_copy"] >= 140 and a["md5"] == b["md5"] and b["pts"] == list(range(150)), (b["zero_copy"], a["md5"], b["md5"])


def test_scene_cuts_with_gop_lanes(stub_lib):
    """a scene cut restarts the key period, so the GOP boundaries are not known when pictures are dealt to lanes: with -lookahead the handle runs one lane (and says so
    through ks265_enc_lanes), the stream is the one-lane stream"""
    kw = dict(W=128, H=96, KS_TEST_CUTS="23,41,77,100", KS_TEST_LOOKAHEAD=8)
    one = run(stub_lib, 150, 32, 0, KS265_GOP_LANES=1, **kw)
    two = run(stub_lib, 150, 32, 0, KS265_GOP_LANES=2, **kw)
    assert two["lanes"] == 1 and one["idr"] == two["idr"] == 6 and one["md5"] == two["md5"], (one["idr"], two["idr"], two["lanes"])   # keys at 0, 23, 41, 73, 100, 132


def _hier_order(d, a):
    """coding order of the mini-GOP (d, a]: the anchor, then the B pictures breadth first (code_hier)"""
    out, cur = [a], [(d, a)]
    while cur:
        nxt = []
        for lo, hi in cur:
            if hi - lo >= 2:
                mid = (lo + hi) // 2
                out.append(mid); nxt += [(lo, mid), (mid, hi)]
        cur = nxt
    return out


def test_slice_type_decision_codes_blocks_of_eight_as_four_plus_four(stub_lib, tmp_path):
    """-lookahead N with the hierarchical GOP (the reference's adaptive BiPredFrames): a block of 8 pictures is coded with anchors 4 apart when predicting its last
    picture from 8 back costs more than the two steps of 4 together (+ 1/12).  Clip: one scene, still in [0, 32), brightening by 3 per picture in [32, 64), still again:
    the blocks inside the ramp are 4 + 4, the blocks outside are 8; coding order, determinism, graph replay, and the stream decodes with the reference decoder"""
    kw = dict(W=128, H=96, KS_TEST_RAMP="32:64:3")
    plain = run(stub_lib, 100, 128, -1, KS_TEST_LOOKAHEAD=0, **kw)
    la = run(stub_lib, 100, 128, -1, out=tmp_path / "la.265", KS_TEST_LOOKAHEAD=8, **kw)
    # no -lookahead given: with the SDK's default GOP the decision runs by itself (pictures on the GOP's grid of 4 only, no scene cuts) - the same slice types, the same stream;
    # and, unlike -lookahead N, in GOP lanes
    auto = run(stub_lib, 100, 128, -1, **kw)
    assert auto["md5"] == la["md5"] and auto["pts"] == la["pts"]
    # the caller never waits for an analysis as long as fewer than six pictures are queued at the input: results that arrive late (the stand-in reports an event done at its
    # n-th query) change nothing but the moment a picture reaches the scheduler
    for lag in (2, 7, 1000):
        assert run(stub_lib, 100, 128, -1, KS265_STUB_EVENT_LAG=lag, **kw)["md5"] == la["md5"] == run(stub_lib, 100, 128, -1, KS_TEST_LOOKAHEAD=8, KS265_STUB_EVENT_LAG=lag, **kw)["md5"]
    for seed in (-1, -2, -3, -4, -5, -6):                                # ... or at random moments (a seed: every look is a coin toss) - with the scheduler thread looking as well
        assert run(stub_lib, 100, 128, -1, KS265_STUB_EVENT_LAG=seed, **kw)["md5"] == la["md5"]
        assert run(stub_lib, 100, 128, -1, KS_TEST_LOOKAHEAD=8, KS265_STUB_EVENT_LAG=seed, **kw)["md5"] == la["md5"]
    lanes = {L: run(stub_lib, 150, 48, -1, KS265_GOP_LANES=L, **kw) for L in (1, 3)}
    assert lanes[1]["md5"] == run(stub_lib, 150, 48, -1, KS265_GOP_LANES=2, KS265_STUB_EVENT_LAG=5, **kw)["md5"] == run(stub_lib, 150, 48, -1, KS265_GOP_LANES=3, KS265_STUB_EVENT_LAG=-9, **kw)["md5"]
    assert lanes[1]["lanes"] == 1 and lanes[3]["lanes"] == 3 and lanes[1]["md5"] == lanes[3]["md5"] and lanes[1]["md5"] != run(stub_lib, 150, 48, -1, KS_TEST_LOOKAHEAD=0, **kw)["md5"]
    assert plain["idr"] == la["idr"] == 1 and sorted(la["pts"]) == list(range(100)) and la["vcl"] == 100
    expect_plain, expect_la = [0], [0]
    for d in range(0, 96, 8):
        expect_plain += _hier_order(d, d + 8)
        expect_la += (_hier_order(d, d + 4) + _hier_order(d + 4, d + 8)) if 32 <= d < 64 else _hier_order(d, d + 8)
    tail = [99, 97, 98]                                      # the flush: 96 + 3 pictures, not a power of two: an anchor and two plain B pictures
    assert plain["pts"][:97] == expect_plain, plain["pts"][:40]
    assert la["pts"][:97] == expect_la, la["pts"][:80]
    assert sorted(la["pts"][97:]) == sorted(tail)
    assert la["md5"] != plain["md5"]
    assert la["md5"] == run(stub_lib, 100, 128, -1, KS_TEST_LOOKAHEAD=8, **kw)["md5"] == run(stub_lib, 100, 128, -1, KS_TEST_LOOKAHEAD=8, KS265_GRAPH=1, **kw)["md5"]
    still = run(stub_lib, 100, 128, -1, KS_TEST_LOOKAHEAD=8, W=128, H=96, KS_TEST_RAMP="1000:1001:3")       # nothing moves: the analysis runs, every block stays 8
    assert still["pts"][:97] == expect_plain
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "la.265"), "-o", str(tmp_path / "d.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
        assert d.returncode == 0 and os.path.getsize(tmp_path / "d.yuv") == 100 * 128 * 96 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]
    # key pictures inside: the period cuts blocks short, a requested key picture too - the grid of 8 restarts at every key picture
    per = run(stub_lib, 100, 44, -1, out=tmp_path / "p.265", KS_TEST_LOOKAHEAD=8, KS_TEST_KEYREQ=1, **kw)
    assert sorted(per["pts"]) == list(range(100)) and per["idr"] >= 3
    if os.path.exists(REF_DEC):
        d = subprocess.run([REF_DEC, "-b", str(tmp_path / "p.265"), "-o", str(tmp_path / "d2.yuv"), "-threads", "2"], capture_output=True, text=True, cwd=tmp_path)
        assert d.returncode == 0 and os.path.getsize(tmp_path / "d2.yuv") == 100 * 128 * 96 * 3 // 2, d.stdout[-300:] + d.stderr[-300:]


@pytest.mark.parametrize("opts,n,W,H", [(["-bframes", "3"], 26, 192, 128), (["-bframes", "3", "-iper", "100"], 90, 128, 128), (["-bframes", "3", "-lookahead", "24", "-aq", "1", "-aqs", "1.0"], 40, 192, 128), (["-bframes", "3", "-iper", "10"], 23, 128, 128)])
def test_crf_runs_the_cutree_pass_of_the_lookahead(tmp_path, opts, n, W, H):
    """-rc 3 (config 4's rate control): the host's cuTree pass - calcFrameCost over the lookahead window, propagation in reverse coding order, the finish, one QP per CTU - linked against
    the stand-in whose operators ARE the oracle's pinned restatements, against tests/cutree_mirror.py (the same pass written independently in Python): the same QP per CTU for every
    picture; the maps are not flat, reference pictures carry the tree's (negative) offsets, non-reference B pictures stay at their picture QP"""
    import re
    import numpy as np
    from ks265codec_amd.synth import make_clip
    from cutree_mirror import CuTree, read_qpmap_dump
    host = os.path.join(ROOT, "ks265codec_amd", "host")
    exe = str(tmp_path / "ks265enc_stub")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(host, "ks265_cli.c"),
                           os.path.join(host, "ks265_enc.c"), os.path.join(host, "ks265_stream.c"), os.path.join(HERE, "hip_stub.c"),
                           "-L", os.path.join(ROOT, "oracle"), "-lks265_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-lm"])
    clip = make_clip(W, H, n, seed=11, abc=(17, 23, 9), pan=(3, 2))
    clip.tofile(tmp_path / "in.yuv")
    dump = tmp_path / "maps.bin"
    args = [exe, "-i", str(tmp_path / "in.yuv"), "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-preset", "slow", "-rc", "3", "-crf", "26", "-iper", "64", "-psnr", "2", "-threads", "3", "-b", str(tmp_path / "o.265"), *opts]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, KS265_DUMP_QPMAP=str(dump)))
    assert r.returncode == 0 and "H265 encoder passed!!!" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
    assert "cuTree over a lookahead" in r.stdout + r.stderr
    got = read_qpmap_dump(dump)
    assert sorted(got) == list(range(n))
    kw = dict(zip(opts[::2], opts[1::2]))
    gop_b = int(kw["-bframes"]); iper = int(kw.get("-iper", 64))
    keys = [d for d in got if got[d][0] == "I" and d > 0]                     # (-lookahead N also runs the scene-cut lookahead: the stand-in's costs cut every 8 pictures)
    assert bool(keys) == ("-lookahead" in kw or iper < n)
    ct = CuTree(clip, W, H, preset=5, gop_b=gop_b, hier=gop_b == 3, iper=iper, lookahead=int(kw.get("-lookahead", -1)), aq_strength=float(kw.get("-aqs", 0)) if "-aq" in kw else 0.0, keys=keys)
    ct.run()
    spread = 0
    for d in range(n):
        kind, qp, m = got[d]
        want = ct.ctu_map(d, qp)
        assert (m == want).all(), f"picture {d} ({kind}, qp {qp}): {m.tolist()} != {want.tolist()}"
        spread = max(spread, int(m.max()) - int(m.min()))
        if "-aq" not in kw:
            assert (m <= qp).all(), "the tree only lowers QPs"
            if kind == "B" and d % 4 != 2 and iper >= n:
                assert (m == qp).all(), "non-reference B pictures keep their picture QP"
    assert spread >= (2 if iper >= n else 1), "the maps are not flat"
    # -cutree 0: the ladder alone, no cu_qp_delta
    r0 = subprocess.run(args + ["-cutree", "0"], capture_output=True, text=True, timeout=300, env=dict(os.environ, KS265_DUMP_QPMAP=str(tmp_path / "none.bin")))
    assert r0.returncode == 0 and "cuTree" not in r0.stdout + r0.stderr
    if "-aq" not in kw:
        assert not os.path.exists(tmp_path / "none.bin") or os.path.getsize(tmp_path / "none.bin") == 0


@pytest.mark.parametrize("lanes,bframes", [(1, 0), (2, -1)])
def test_input_straight_from_the_callers_buffers(stub_lib, lanes, bframes):
    """pictures of 1 MB and more are uploaded from the caller's own planes (pinned in place once, remembered by address) instead of being copied first: the same stream as the
    copying path (KS265_INPUT_COPY=1), with distinct buffers, with ONE buffer refilled (and scribbled over) after every call as the SDK's demo callers do, with buffers the runtime
    refuses to pin (fallback to the copy), and without waiting for the upload (KS265_INPUT_HOLD=1: the caller keeps its buffers)"""
    W, H, n = 1024, 704, 70
    env = {"KS265_GOP_LANES": lanes} if lanes > 1 else {}
    ref = run(stub_lib, n, 32, bframes, W, H, KS265_INPUT_COPY=1, **env)
    assert ref["vcl"] == n and ref["lanes"] == lanes
    for extra in ({}, {"KS_TEST_ONE_BUFFER": 1}, {"KS265_STUB_NO_REGISTER": 1}, {"KS265_INPUT_HOLD": 1}, {"KS_TEST_ONE_BUFFER": 1, "KS265_INPUT_COPY": 1}):
        got = run(stub_lib, n, 32, bframes, W, H, **env, **extra)
        assert got["md5"] == ref["md5"] and got["pts"] == ref["pts"], extra
