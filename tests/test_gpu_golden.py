"""GPU parity, batched operator tables: the HIP kernels (through the C ABI) must reproduce, bit for bit,
the outputs of the reference binary's own `_c` kernels recorded in tests/golden/ — and agree with the
CPU oracle on the same inputs.  Run on the MI355X box: pytest -m gpu."""
from __future__ import annotations

import numpy as np
import pytest

from golden_io import load_cases

pytestmark = pytest.mark.gpu

SIZES = [4, 4, 8, 16, 32]


@pytest.fixture(scope="module")
def ks():
    from ks265codec_amd.lib import KsContext
    c = KsContext(0)
    yield c
    c.close()


def _planes(ks, cases, ka, kb):
    """Concatenate the per-case planes into two big device byte buffers; return per-case base offsets."""
    offs_a, offs_b, pa, pb = [], [], [], []
    na = nb = 0
    for c in cases:
        offs_a.append(na); offs_b.append(nb)
        pa.append(c[ka].reshape(-1)); pb.append(c[kb].reshape(-1))
        na += c[ka].size; nb += c[kb].size
    return ks.dev(np.concatenate(pa)), ks.dev(np.concatenate(pb)), offs_a, offs_b


def _run_dist(ks, fn, cases, groups):
    """strides are per-call scalars in the ABI (as in the reference tables) -> one batch per (sa, sb)"""
    from ks265codec_amd.lib import BLK
    A, B, oa, ob = _planes(ks, cases, "a", "b")
    got = {}
    by_stride = {}
    for i, c in enumerate(cases):
        by_stride.setdefault((int(c["sa"]), int(c["sb"])), []).append(i)
    for (sa, sb), idxs in by_stride.items():
        blks = np.zeros(len(idxs), BLK)
        for j, i in enumerate(idxs):
            w, h = groups(cases[i])
            blks[j] = (oa[i], ob[i], w, h)
        r = fn(A, sa, B, sb, blks)
        for j, i in enumerate(idxs):
            got[i] = r[j]
    return got


def test_sad(ks):
    cases = load_cases("sad")
    got = _run_dist(ks, ks.sad, cases, lambda c: (c["w"], c["h"]))
    for i, c in enumerate(cases):
        assert got[i] == c["exp"], (c["h"], c["w"])


def test_sse(ks):
    cases = load_cases("sse")
    got = _run_dist(ks, ks.sse, cases, lambda c: (c["n"], c["n"]))
    for i, c in enumerate(cases):
        assert got[i] == c["exp"]


def test_had(ks):
    cases = load_cases("had")
    got = _run_dist(ks, ks.had, cases, lambda c: (c["w"], c["h"]))
    for i, c in enumerate(cases):
        assert got[i] == c["exp"], (c["h"], c["w"])


def test_bi_full_window(ks):
    """interMeBiFull_c enc@0x4896d0 / interMeBiHadFull_c enc@0x4897e0 against the reference's own outputs (tests/golden/bifull.npz), ties included"""
    from ks265codec_amd.lib import BLK
    cases = load_cases("bifull")
    A, B, oa, ob = _planes(ks, cases, "org", "ref")
    groups = {}
    for i, c in enumerate(cases):
        groups.setdefault((int(c["had"]), int(c["so"]), int(c["sr"])), []).append(i)
    for (had, so, sr), idxs in groups.items():
        blks = np.zeros(len(idxs), BLK)
        for j, i in enumerate(idxs):
            blks[j] = (oa[i], ob[i], cases[i]["w"], cases[i]["h"])
        got = ks.bi_full(had, A, so, B, sr, blks, np.stack([cases[i]["mvcost"] for i in idxs]))
        for j, i in enumerate(idxs):
            c = cases[i]
            assert got[j, 0] == c["exp_cost"] and got[j, 1] == np.uint32(c["exp_best"][0]), (had, int(c["w"]), int(c["h"]), got[j], int(c["exp_cost"]), int(c["exp_best"][0]))


def test_sad4blk(ks):
    """sad4blk_8x8_c enc@0x4cee30: the four 8x8 quadrant SADs of a 16x16 block"""
    cases = load_cases("sad4blk")
    got = _run_dist(ks, ks.sad4blk_8x8, cases, lambda c: (16, 16))
    for i, c in enumerate(cases):
        assert (got[i] == c["exp"]).all()


def test_sad4(ks):
    from ks265codec_amd.lib import BLK
    for c in load_cases("sad4"):
        F, R = ks.dev(c["fenc"]), ks.dev(c["ref"])
        blks = np.zeros(1, BLK); blks[0] = (0, c["ref_off"], c["w"], c["h"])
        got = ks.sad4(F, int(c["sf"]), R, int(c["sr"]), blks)
        assert (got[0] == c["exp"]).all(), (c["w"], c["h"])


def test_sad3(ks):
    from ks265codec_amd.lib import BLK3
    for c in load_cases("sad3"):
        F, R = ks.dev(c["fenc"]), ks.dev(c["ref"])
        blks = np.zeros(1, BLK3)
        blks[0]["a_off"] = 0; blks[0]["b_off"] = c["offs"]; blks[0]["w"] = c["w"]; blks[0]["h"] = c["h"]
        got = ks.sad3(F, int(c["sf"]), R, int(c["sr"]), blks)
        assert (got[0] == c["exp"]).all()


def test_residual(ks):
    from ks265codec_amd.lib import BLK
    for c in load_cases("residual"):
        n = c["n"]
        blks = np.zeros(1, BLK); blks[0] = (0, 0, n, n)
        got = ks.residual(ks.dev(c["org"]), int(c["so"]), ks.dev(c["pred"]), int(c["sp"]), blks)
        assert (got.reshape(n, n) == c["exp"]).all()


def test_fwd_transform(ks):
    cases = load_cases("fwd_transform")
    for idx in range(5):
        n = SIZES[idx]
        sel = [c for c in cases if c["idx"] == idx]
        src = np.stack([c["src"][:, :n] for c in sel])
        got = ks.fwd_transform(idx, src)
        for g, c in zip(got, sel):
            assert (g == c["exp"][:, :n]).all(), idx


def test_inv_transform(ks):
    cases = load_cases("inv_transform")
    for idx in range(5):
        n = SIZES[idx]
        sel = [c for c in cases if c["idx"] == idx]
        coef = np.stack([c["coef"][:, :n] for c in sel])
        pred = np.stack([c["pred"][:, :n] for c in sel])
        got = ks.inv_transform(idx, coef, pred)
        for g, c in zip(got, sel):
            assert (g == c["exp"]).all(), (idx, c["variant"])


def test_quant(ks):
    for c in load_cases("quant"):
        if c.get("kind") == "base_param":
            continue
        n = c["n"]
        lvl, du, nz = ks.quant(n, c["coef"][None], int(c["scale"]), int(c["off"]), int(c["qbits"]))
        assert (lvl[0] == c["exp_lvl"]).all() and (du[0] == c["exp_du"]).all() and nz[0] == c["exp_nz"], (n, c["qp"])


def test_dequant(ks):
    for c in load_cases("dequant"):
        n = c["n"]
        if c["lx"] != n - 1 or c["ly"] != n - 1:        # lastX / lastY sub-rectangle: untouched coefficients keep the fill value of the fixture
            init = np.full((n, n), c["fill"], np.int16)
            got = ks.dequant_rect(n, c["lvl"][None], init[None], int(c["scale"]), int(c["add"]), int(c["shift"]), int(c["lx"]), int(c["ly"]))
            assert (got[0] == c["exp"]).all(), (n, int(c["lx"]), int(c["ly"]))
            continue
        got = ks.dequant(n, c["lvl"][None], int(c["scale"]), int(c["add"]), int(c["shift"]))
        assert (got[0] == c["exp"]).all()


@pytest.mark.parametrize("fam,chroma", [("deblock_luma", False), ("deblock_chroma", True)])
def test_deblock_edges(ks, fam, chroma):
    from ks265codec_amd.lib import EDGE
    cases = load_cases(fam)
    for c in cases:
        img = ks.dev(c["img"])
        e = np.zeros(1, EDGE)
        e[0]["pix_off"] = c["off"]; e[0]["beta"] = c.get("beta", 0); e[0]["tc"] = c["tc"]; e[0]["length"] = c["length"]
        e[0]["dir"] = 0 if c["orient"] == "ver" else 1
        e[0]["flags"] = int(c["fp"]) | (int(c["fq"]) << 1)
        ks.edge_filter(img, int(c["stride"]), e, chroma=chroma)
        got = ks.host(img, np.uint8, c["img"].shape)
        assert (got == c["exp"]).all()


def test_interp(ks):
    io_code = {"8to8": 0, "8to16": 1, "16to8": 2, "16to16": 3}
    for c in load_cases("interp"):
        name = str(c["name"])
        comp, direc, io = name.split("_")
        kind = (1 if comp == "chroma" else 0) | (2 if direc == "ver" else 0) | (io_code[io] << 2)
        ddt = np.uint8 if io.endswith("to8") else np.int16
        src = ks.dev(c["src"])
        dst = ks.zeros(c["h"] * c["ds"] * np.dtype(ddt).itemsize)
        ks.interp_rect(kind, dst, int(c["ds"]), src, int(c["off"]) * c["src"].itemsize, int(c["ss"]), int(c["w"]), int(c["h"]), int(c["frac"]))
        got = ks.host(dst, ddt, (c["h"], c["ds"]))[:, :c["w"]]
        assert (got == c["exp"]).all(), (name, c["frac"])


def test_sao_apply(ks):
    for c in load_cases("sao_apply"):
        kind = str(c["kind"])
        rec = ks.dev(c["rec"])
        if kind == "bo":
            ks.sao_apply_bo(c["offs"], rec, int(c["stride"]), int(c["h"]), int(c["w"]), int(c["band"]))
            got = ks.host(rec, np.uint8, c["rec"].shape)
        else:
            dst = ks.dev(c["rec"])
            ks.sao_apply_eo(int(kind[2]), c["offs"], rec, dst, int(c["stride"]) + 1, int(c["stride"]), int(c["h"]), int(c["w"]))
            got = ks.host(dst, np.uint8, c["rec"].shape)
        assert (got == c["exp"]).all(), kind


def test_sao_stats(ks):
    from ks265codec_amd.lib import SAO_RECT
    for c in load_cases("sao_stats"):
        r = np.zeros(1, SAO_RECT); r[0] = (0, c["rs"] + 1, c["w"], c["h"])
        got = ks.sao_stats(ks.dev(c["org"]), int(c["os"]), ks.dev(c["rec"]), int(c["rs"]), r, int(c["step"]))
        assert (got[0, :64] == c["exp_eo"]).all() and (got[0, 64:] == c["exp_bo"]).all()


def test_intra(ks):
    """§8(f) rank 1: g_IntraPredFunction / IntraPredFilterRef_c through the C ABI against the reference's own outputs"""
    from ks265codec_amd.lib import INTRA_BLK, INTRA_REF
    cases = load_cases("intra")
    pred = [c for c in cases if str(c["kind"]) == "pred"]
    filt = [c for c in cases if str(c["kind"]) == "filter"]
    # one ref arena + one dst arena, all blocks in ONE launch
    ref_off, dst_off, refs, blks = 0, 0, [], np.zeros(len(pred), INTRA_BLK)
    for i, c in enumerate(pred):
        n = 1 << c["log2"]
        edge = 0 if str(c["func"]) == "chroma_dc" else c["filt"]
        blks[i] = (ref_off + c["corner"], dst_off, c["ds"], c["mode"], c["log2"], edge, (0, 0, 0))
        refs.append(c["ref"]); ref_off += len(c["ref"]); dst_off += n * c["ds"]
    dst = ks.dev(np.full(dst_off, 7, np.uint8))
    ks.intra_pred(ks.dev(np.concatenate(refs)), dst, blks)
    got = ks.host(dst, np.uint8)
    for b, c in zip(blks, pred):
        n = 1 << c["log2"]
        g = got[b["dst_off"]:b["dst_off"] + n * c["ds"]].reshape(n, c["ds"])
        assert (g == c["exp"]).all(), (str(c["func"]), c["mode"], n, c["filt"])
    off, srcs, rr = 0, [], np.zeros(len(filt), INTRA_REF)
    for i, c in enumerate(filt):
        rr[i] = (off + c["corner"], off + c["corner"], c["size"], c["flag"])
        srcs.append(c["src"]); off += len(c["src"])
    dst = ks.dev(np.full(off, 9, np.uint8))
    ks.intra_filter_ref(ks.dev(np.concatenate(srcs)), dst, rr)
    got = ks.host(dst, np.uint8)
    off = 0
    for c in filt:
        assert (got[off:off + len(c["src"])] == c["exp"]).all(), (c["size"], c["flag"])
        off += len(c["src"])


def test_downsample_from_pinned_host_memory(ks):
    """ks265_downsample_from_host (the encoder host's lookahead reads its pinned input picture over PCIe: 64 work-groups, a wave per row) == ks265_downsample_rect == downsample_c,
    at 2160p, at a width that leaves a partial group of four at the row end, and at a width below one wave iteration"""
    rng = np.random.default_rng(5)
    for W, H in ((3840, 2160), (1928, 1080), (136, 72)):
        src = rng.integers(0, 256, (H, W), dtype=np.uint8)
        w, h = (W // 2 - 2 if W == 1928 else W // 2), H // 2       # (962: a partial group of four at the row end)
        ds = (w + 3) & ~3
        exp = ks.downsample(ks.dev(src), W, w, h, ds)
        got = ks.downsample_from_host(src, W, w, h, ds)
        assert (got[:, :w] == exp[:, :w]).all(), (W, H)
        a = src.astype(np.int32)
        ref = (((a[0::2, 0::2] + a[1::2, 0::2] + 1) >> 1) + ((a[0::2, 1::2] + a[1::2, 1::2] + 1) >> 1) + 1) >> 1
        assert (got[:, :w] == ref[:h, :w]).all()


def test_lookahead_kernels(ks):
    """§8(f) rank 2 leaf kernels through the C ABI: downsample_c, weightBi_sad_c, acEnergyPlane_c (+ the whole-plane map)"""
    from ks265codec_amd.lib import BLK3
    for c in load_cases("lookahead"):
        kind = str(c["kind"])
        if kind == "down":
            got = ks.downsample(ks.dev(c["src"]), int(c["ss"]), int(c["w"]), int(c["h"]), int(c["ds"]))
            assert (got[:, :c["w"]] == c["exp"][:, :c["w"]]).all(), (c["w"], c["h"])
        elif kind == "wbsad":
            b = np.zeros(1, BLK3); b[0] = (0, (0, 0, 0), c["w"], c["h"])
            got = ks.weight_bi_sad(ks.dev(c["org"]), int(c["so"]), ks.dev(c["r0"]), int(c["s0"]), ks.dev(c["r1"]), int(c["s1"]), b)
            assert got[0] == c["ret"], (c["w"], c["h"])
        else:
            got = ks.ac_energy(ks.dev(c["src"]), int(c["st"]), int(c["log2"]), np.zeros(1, np.int32))
            assert got[0] == c["ret"], c["log2"]
    # the map variant = the batch variant at every aligned block
    rng = np.random.default_rng(3)
    plane = rng.integers(0, 256, (72, 136), dtype=np.uint8)
    for log2 in (3, 4):
        n = 1 << log2
        m = ks.ac_energy_map(ks.dev(plane), 136, 128, 64, log2)
        offs = np.array([y * n * 136 + x * n for y in range(64 // n) for x in range(128 // n)], np.int32)
        assert (m.reshape(-1) == ks.ac_energy(ks.dev(plane), 136, log2, offs)).all()


def test_sign_hiding(ks):
    """ks265_sign_hiding_batch == signBitHidingHDQ enc@0x4aa150 as executed inside the reference binary (tests/golden/sbh.npz)"""
    cases = load_cases("sbh")
    for n in (4, 8, 16, 32):
        for scan in (0, 1, 2):
            sel = [c for c in cases if int(c["n"]) == n and int(c["scan"]) == scan]
            if not sel:
                continue
            got = ks.sign_hiding(n, scan, np.stack([c["lvl"] for c in sel]), np.stack([c["coef"] for c in sel]), np.stack([c["deltaU"] for c in sel]))
            for g, c in zip(got, sel):
                exp = c["exp_lvl"] if int(c["nz"]) > 1 else c["lvl"]
                assert (g == exp).all(), (n, scan, int(c["qp"]))
