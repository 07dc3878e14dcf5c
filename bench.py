#!/usr/bin/env python3
"""bench.py — throughput of the HEVC encode pixel-kernel hot path on MI355X.

One "step" = one picture (3840x2160 4:2:0, BASELINE.json configs[2]) of every GOP shard of the rank through the whole hot path
(integer ME (UMH by default, as -preset slow) -> sub-pel SATD -> CU decision -> residual/DCT/quant/dequant/IDCT/recon ->
deblock -> SAO -> border padding; key pictures: intra mode pre-selection + wavefront reconstruction), all inputs and outputs resident in
HBM.  Each rank (one per GPU) encodes --streams (default 3) independent GOP shards, each on its own HIP stream: GOPs are
independent units (frames/GOPs shard, no data-path collective) and the search kernels are latency bound, so the
kernels of different shards overlap (+53 % pictures/s over one stream on a whole -iper 128 GOP: the intra wavefront of a key
picture keeps only 34 of 256 CUs busy and runs underneath the other shards' P pictures).  Weak scaling; value = pictures of all ranks / max-over-ranks time.  Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the CPU oracle port on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

# ALGORITHMIC bytes per picture, in units of P = luma samples (DESIGN.md §5 states each derivation)
ALGO_BYTES_P = {
    "me_integer": 2.2,       # source P + padded reference ~1.1 P + PU records (the propagation round of stage A2 re-reads source and candidates from L2: + 13 us, not counted)
    "me_subpel": 2.2,
    "intra_candidates": 0.25,  # the source samples of the CTUs the gate lets through (16 - 20 % of them) + their PU records + the packed candidates
    "cu_decide": 1.6,        # CU decision (PU records) + merge pass: source 1 P + prediction tiles of the candidates ~0.5 P + maps
    "reconstruct": 7.5,      # org 1.5 P + pred 1.5 P + levels 3 P + recon 1.5 P
    "intra_pass": 0.05,      # the intra CUs of a P picture (under 1 % of its blocks): source, levels, recon of those CUs - a dependency chain, not a streaming kernel
    "deblock": 3.1,
    "sao": 6.0,              # statistics 3 P + apply 3 P (fused in one launch)
}


USER_LANES = None


def hot_tools(args, me_method):
    """the hot-path leg's tool set = the encoder host's at -preset slow (ks265codec_amd.synth.ENCODER_TOOLS: the one dict the GPU tests, smoke() and
    tools/rd_eval.py --host use too) with this run's command-line overrides"""
    from ks265codec_amd.synth import ENCODER_TOOLS, subme_knobs
    t = dict(ENCODER_TOOLS)
    t.update(me_method=me_method, me_hex_thr=(args.me_hex_thr if me_method == 2 else 0), pre_search=0 if args.no_pre_search else 1, propagate=args.propagate)
    if args.subme_preset:
        t.update(subme_knobs(args.subme_preset))
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128, help="timed steps; the default covers one whole -iper 128 GOP per shard, key picture included")
    ap.add_argument("--lookahead", type=int, default=-1, metavar="N", help="-lookahead N of the encoder host: -1 = its default (IPPP: off; hierarchical-B 8: the slice-type decision runs by itself), 0 = off, "
                    "N > 0 = scene cuts + slice types, every picture analysed")
    ap.add_argument("--hier-b", type=int, default=-1, metavar="G", help="hierarchical-B mini-GOPs of G pictures (power of two). Default (-1): 8 = the GOP the reference codes for this command line "
                    "(-latency offline default; SURVEY.md 5), with the IPPP variant (-bframes 0) run beside it and reported as `ippp`; 0 = IPPP only (or --bframes N: P + N plain B)")
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=27)
    ap.add_argument("--zero-copy", action="store_true", help="encoded leg: produce every picture straight into one of the encoder's pinned input buffers (ks265_enc_acquire_input) instead of handing in a buffer that is copied")
    ap.add_argument("--iper", type=int, default=128)
    ap.add_argument("--clip-frames", type=int, default=33, help="distinct pictures of the synthetic clip (ping-ponged: 2 n - 2 pictures per period); SURVEY.md 8(d) asks for clips, not a handful of pictures - "
                    "the search kernels' early exits are data dependent")
    ap.add_argument("--me", choices=["dia", "hex", "umh"], default="umh", help="integer search: -preset slow resolves to -me 2 (UMH), SURVEY.md §5")
    ap.add_argument("--me-hex-thr", type=int, default=16, help="tME+0x368 of the reference: with -me 2 a PU whose start-point SAD is below this many units per sample runs "
                    "interMeHex instead of interMeUMH; -preset slow resolves to 16, veryslow to 0 (always UMH)")
    ap.add_argument("--bframes", type=int, default=0, help="-bframes: B pictures between anchors (coding order P b b b); 0 = IPPP")
    ap.add_argument("--b-spread", action="store_true", help="config-5 style: anchor chain on rank 0, RCCL broadcast of every reconstructed anchor, "
                    "B pictures dealt to the other ranks (needs --bframes > 0); default = one GOP shard per rank, no collective")
    ap.add_argument("--streams", type=int, default=3, help="independent GOP shards in flight per GPU, each on its own HIP stream (their kernels overlap: the search kernels are latency bound)")
    ap.add_argument("--refs", type=int, default=1, help="list-0 reference pictures a P picture searches (-ref / -ref0; -preset slow resolves to 1 / 3: three for the first picture of a mini-GOP); IPPP only")
    ap.add_argument("--ref0", type=int, default=3, help="list-0 pictures an ANCHOR of the pyramid GOPs searches (-ref0; -preset slow resolves to 3: the encoder host's default) - the hot-path leg's --hier-b schedule follows it")
    ap.add_argument("--subme-preset", default="", help="hot-path leg: the sub-pel refinement's knobs of another preset (ultrafast .. placebo; default: slow's)")
    ap.add_argument("--propagate", type=int, default=1, help="hot-path leg: rounds of vector propagation between neighbouring PUs after every integer search (stage A2; the encoder runs 1)")
    ap.add_argument("--no-pre-search", action="store_true", help="hot-path leg: stage A without the pyramid pre-search start candidates (the encoder always runs them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-lanes", action="store_true", help="also run the encoded leg with KS265_GOP_LANES=2 in a process of its own (reported beside value, never as it)")
    ap.add_argument("--leg", choices=["both", "encoded", "hot"], default="both", help="encoded: the whole encoder through the SDK-compatible C API (host pictures in, "
                    "Annex-B NAL units out: H2D, pixel path, D2H, CABAC) = the headline value; hot: the device-resident pixel path only (roofline leg)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak (default): every rank encodes --steps pictures of its own clip.  strong: ONE fixed job of --job-frames pictures "
                    "(closed GOPs of -iper pictures) is split over the ranks GOP by GOP, rank 0 gathers the ranks' NAL units in stream order; value = job frames / wall time (encoded leg only)")
    ap.add_argument("--job-frames", type=int, default=1024, help="--scaling strong: pictures of the fixed job")
    ap.add_argument("--out", default=None, help="--scaling strong: rank 0 writes the gathered Annex-B stream here")
    ap.add_argument("--host-threads", type=int, default=0, help="slice-writer threads of the encoded leg per rank (0 = min(32, host cores / ranks))")
    args = ap.parse_args()
    # round 5: the headline is the GOP the reference itself codes for `-preset slow -rc 0 -qp 27 -iper 128` = hierarchical-B 8; IPPP (`-bframes 0`) runs beside it
    args.both_gops = args.hier_b < 0 and not args.bframes and args.refs <= 1 and not args.b_spread and args.scaling == "weak"
    if args.hier_b < 0:
        args.hier_b = 8 if args.both_gops else 0

    # GOP lanes (two closed GOPs side by side on the one GPU) need more than the runtime's four hardware queues - with four, two lanes' streams share queues and code 561
    # pictures/s where eight give 700.  The runtime reads the variable once, when it starts: set before torch (and, with several ranks, RCCL) touches it.  A value the user set stays.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    global USER_LANES
    USER_LANES = os.environ.get("KS265_GOP_LANES")       # round 6: the library defaults to ONE lane and never touches the environment; two lanes for the pyramid GOPs are this application's decision (run_encoded), as in the CLI
    import torch
    from ks265codec_amd.lib import KsContext, KsFrame
    from ks265codec_amd.synth import host_qp_offset, lambda_q4, make_clip
    from ks265codec_amd import gop

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # functional test of the N > 1 logic on a box with fewer GPUs than ranks: KS265_BENCH_BACKEND=gloo KS265_BENCH_ONE_DEVICE=1
    # (every rank on device 0, host-side collectives).  The driver's runs use neither: one rank per GPU over RCCL.
    backend = os.environ.get("KS265_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("KS265_BENCH_ONE_DEVICE") else local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W, H, qp = args.width, args.height, args.qp
    me_method = {"dia": 0, "hex": 1, "umh": 2}[args.me]
    nb = args.bframes
    nstreams = 1 if args.b_spread else max(1, args.streams)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    encoded = None
    shared_clip = None
    if args.scaling == "strong":
        args.leg = "encoded"
        if args.b_spread:
            raise SystemExit("--scaling strong splits closed GOPs over the ranks; --b-spread is the other sharding")
    if args.leg != "hot" and not args.b_spread:
        import copy
        shared_clip = None if args.scaling == "strong" else make_clip(W, H, args.clip_frames, seed=7 + rank, abc=(67, 91, 33), pan=(8, 5))
        encoded = encoded_leg(args, torch, dist, rank, world, dev_index, backend, sync_all, shared_clip)
        if args.both_gops:
            a2 = copy.copy(args); a2.hier_b = 0; a2.bframes = 0
            e2 = encoded_leg(a2, torch, dist, rank, world, dev_index, backend, sync_all, shared_clip)
            encoded["ippp"] = {"value": round(e2["fps"], 2), "unit": "frames/s", "psnr_y": round(float(e2["psnr_y"]), 3), "bytes_per_picture": int(e2["bytes_per_picture"]),
                               "kbps_at_50fps": round(e2["bytes_per_picture"] * 8 * 50 / 1000.0, 1), "windows": e2["windows"], "caller_ms_per_picture": e2.get("caller_ms_per_picture"),
                               "what": "the same encoder, same clip, same run, with -bframes 0 (IPPP): a VARIANT of the metric's command line (BASELINE.md 2: the reference codes the "
                                       "command line as hierarchical-B 8, which is `value`); rounds 1 - 4 reported this figure as `value`"}
        # round 4: the two-lane leg is gone from the default run (opt in with --two-lanes).  Measured: two GOP lanes code 1.00x, two encoder processes 1.15x
        # what one lane does on the one GPU - every large kernel of the picture fills the CUs on its own - so lanes are for handles that span several GPUs
        encoded["two_lanes"] = None
        if args.two_lanes and args.leg == "both" and args.scaling == "weak" and world == 1 and USER_LANES is None and encoded.get("gop_lanes", 1) == 1 and not args.hier_b and not args.bframes:
            import subprocess                                     # a process of its own: the library asks for eight hardware queues, which only a fresh runtime honours
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "encoded", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
                                    "--width", str(args.width), "--height", str(args.height), "--qp", str(args.qp), "--iper", str(args.iper), "--me", args.me],
                                   env=dict(os.environ, KS265_GOP_LANES="2"), capture_output=True, text=True, timeout=240)
                e2 = json.loads(r.stdout.strip().splitlines()[-1])
                encoded["two_lanes"] = {"value": e2["value"], "unit": "frames/s", "gop_lanes": e2["config"].get("gop_lanes"), "psnr_y": e2["psnr_y"], "windows": e2["config"].get("windows"),
                                        "what": "the same encoder with KS265_GOP_LANES=2 (opt-in): two closed GOPs in flight on the one GPU, same stream; each window is a closed piece of "
                                                "work (flush + synchronize on both sides), so it includes starting from and draining to an empty pipeline"}
            except Exception as ex:                               # the opt-in leg must never cost the default line
                encoded["two_lanes"] = {"error": repr(ex)[:200]}
        if args.leg == "encoded":
            if rank == 0:
                print(json.dumps(encoded_line(args, encoded, world, None, None)), flush=True)
            if dist is not None:
                dist.barrier(); dist.destroy_process_group()
            return

    # the device-resident leg (roofline / stage times): with the default command it stays on IPPP - every picture but the key picture is a P picture, whose kernels are the ones the
    # stage events time; the GOP of `value` is the encoded leg's business.  --hier-b G / --bframes N given explicitly run here too
    args.ref_hier_b = args.hier_b
    if args.both_gops:
        args.hier_b = 0
    shared = shared_clip

    def make_shard(sidx):
        """one independent GOP shard: its own context (= HIP stream), frame object, clip, decoded-picture buffer and schedule"""
        # shard 0 lives on torch's default stream; every further shard gets a stream of its own (KsContext adopts the torch stream
        # that is current when it is created), so that kernels of different shards can overlap on the GPU
        import contextlib
        tstream = None if sidx == 0 else torch.cuda.Stream(device=dev_index)
        with (torch.cuda.stream(tstream) if tstream is not None else contextlib.nullcontext()):
            ks = KsContext(dev_index)
            fr = KsFrame(ks, W, H, qp, lambda_q4(qp), bframes=max(args.bframes, args.hier_b - 1 if args.hier_b else 0), refs=max(1, args.refs, args.ref0 if args.hier_b else 1), **hot_tools(args, me_method))
            # synthetic clip of SURVEY.md §8(d), one GOP shard per stream (different seed per shard = different content)
            # one clip per rank (the encoded leg's), every shard at its own phase of it (round 5: 33 distinct pictures instead of 5 - one clip per shard would be 1.2 GB of host memory and 3 x the set-up time)
            clip = shared if shared is not None else make_clip(W, H, args.clip_frames, seed=7 + (0 if args.b_spread else rank), abc=(67, 91, 33), pan=(8, 5))
            phase = (sidx * (2 * len(clip) - 2)) // max(1, nstreams)
            dev_clip = [ks.dev(c) for c in clip]
            srcs = [fr.new_pic() for _ in clip]
            for d, s in zip(dev_clip, srcs):
                fr.load_i420(d, s)
            order = list(range(len(clip))) + list(range(len(clip) - 2, 0, -1))   # ping-pong keeps the motion continuous
            # decoded-picture buffer: two anchors (previous / next I-or-P picture) + one scratch output for non-reference B pictures
            anchors = [fr.new_pic(), fr.new_pic()]
            bout = fr.new_pic()

            sched = gop.hier_order(args.hier_b, args.iper) if args.hier_b else gop.coding_order(nb, args.iper)
            R0 = max(1, min(4, args.ref0))
            dpb = [fr.new_pic() for _ in range(R0 * args.hier_b + 1)] if args.hier_b else []      # slot = display index mod (ref0 x G + 1): an anchor stays until the anchor ref0 mini-GOPs later is coded
            state = {"n": 0, "cur": 0, "last": None, "since_key": 0, "hist": []}

            def src_of(d):
                return srcs[order[(d + phase) % len(order)]]

            def step_hier():
                d, kind, r0, r1, layer = next(sched)
                G1 = len(dpb)
                q = qp + host_qp_offset(kind, layer=layer, hier=True)   # the encoder host's ladder (= the reference's): anchors Q + 1, B layers + 2 / + 4 / + 4
                fr.set_qp(q, lambda_q4(q, inter=kind != "I"))       # P / B pictures: the encoder host's inter table (ks265_enc.c kLambdaInterQ4)
                lean = kind == "B" and (1 << layer) == args.hier_b and not os.environ.get("KS265_LEAN_B") == "0"     # the host's lean B pictures: the top layer (nothing predicts from it) without intra candidates, joint refinement, SAO
                near = kind == "B" and (2 << layer) == args.hier_b and os.environ.get("KS265_LEAN_B") not in ("0", "3")      # ... the layer above it (references two pictures away): without intra candidates and SAO
                fr.set_picture_tools(*((0, 0, 0, 1 if os.environ.get("KS265_LEAN_B") == "2" and me_method == 2 else -1) if lean else (0, -1, 0, -1) if near else (-1, -1, -1, -1)))
                out = dpb[d % G1]
                hist = state["hist"]                                # the GOP's anchors so far, nearest first (-ref0: an anchor searches the last ref0 of them, as the encoder host schedules it)
                multi = kind == "P" and R0 > 1 and len(hist) > 1 and hist[0] == r0
                if kind == "B":
                    fr.encode_picture_b(src_of(d), dpb[r0 % G1], dpb[r1 % G1], out)
                elif multi:
                    fr.encode_picture_mref(src_of(d), [dpb[p % G1] for p in hist[:R0]], out)
                else:
                    fr.encode_picture(src_of(d), dpb[r0 % G1] if r0 is not None else out, kind == "I", out)
                if kind == "I":
                    hist[:] = [d]
                elif kind == "P":
                    hist.insert(0, d); del hist[4:]
                state["last"] = (d, out)
                if multi:
                    state["ref0"] = dpb[r0 % G1]
                state["kind"] = "Pm" if multi else kind               # "Pm": several references; ks265_encode_picture_mref records no stage events
                state["n"] += 1

            ring = [fr.new_pic() for _ in range(args.refs + 1)] if args.refs > 1 else []   # multi-reference IPPP: the picture being written + the most recent ones

            def step_mref():
                d, kind = next(sched)
                R, cur = len(ring), state["cur"]
                nxt = (cur + 1) % R
                q = qp if kind == "I" else qp + 1
                fr.set_qp(q, lambda_q4(q, inter=kind != "I"))
                avail = 0 if kind == "I" else min(args.refs, state["since_key"])
                if avail <= 1:
                    fr.encode_picture(src_of(d), ring[cur], kind == "I", ring[nxt])
                else:
                    fr.encode_picture_mref(src_of(d), [ring[(cur - i) % R] for i in range(avail)], ring[nxt])
                state["since_key"] = 1 if kind == "I" else state["since_key"] + 1
                state["cur"] = nxt
                state["last"] = (d, ring[nxt])
                state["ref0"] = ring[cur]
                state["kind"] = kind if avail <= 1 else "Pm"     # "Pm": several references; ks265_encode_picture_mref records no stage events
                state["n"] += 1

            def step():
                if args.hier_b:
                    return step_hier()
                if args.refs > 1:
                    return step_mref()
                d, kind = next(sched)
                cur = state["cur"]
                if kind == "B":
                    q = qp + 2                                  # the reference's hidden hierarchy offsets: I = Q, P = Q+1, B = Q+2..
                    fr.set_qp(q, lambda_q4(q, inter=True))
                    fr.encode_picture_b(src_of(d), anchors[cur ^ 1], anchors[cur], bout)   # list 0 = previous anchor, list 1 = the anchor just coded
                    state["last"] = (d, bout)
                else:
                    state["pos"] = 0 if kind == "I" else state.get("pos", 0) + 1
                    q = qp + host_qp_offset(kind, state["pos"])      # the encoder host's IPPP ladder (= the reference's cascade): P = Q + 1 + {0, 2, 1, 2}[position & 3]
                    fr.set_qp(q, lambda_q4(q, inter=kind != "I"))
                    fr.encode_picture(src_of(d), anchors[cur], kind == "I", anchors[cur ^ 1])
                    state["cur"] = cur ^ 1
                    state["last"] = (d, anchors[cur ^ 1])
                state["kind"] = kind
                state["n"] += 1
        torch.cuda.synchronize()
        import types
        return types.SimpleNamespace(ks=ks, fr=fr, clip=clip, srcs=srcs, order=order, anchors=anchors, bout=bout, state=state, step=step, src_of=src_of)

    shards = [make_shard(i) for i in range(nstreams)]
    # de-phase the shards: shard i starts i/nstreams of a GOP ahead, so that the key pictures (intra wavefront: 34 of 256 CUs busy)
    # of different shards do not fall on the same step but run underneath the other shards' P pictures - as independent
    # streams in a real deployment would
    if not args.b_spread and not args.hier_b and nstreams > 1:
        for i, sh in enumerate(shards):
            for _ in range((i * args.iper) // nstreams):
                sh.step()
        torch.cuda.synchronize()
    sh0 = shards[0]
    clip = sh0.clip
    ks, fr, srcs, order, anchors, bout, state, src_of = sh0.ks, sh0.fr, sh0.srcs, sh0.order, sh0.anchors, sh0.bout, sh0.state, sh0.src_of

    def step():                                         # one step = one picture on every stream of this rank (kernels of different shards overlap)
        for sh in shards:
            sh.step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.b_spread:
        if nb <= 0:
            raise SystemExit("--b-spread needs --bframes > 0")
        slots = [fr.new_pic(), fr.new_pic(), fr.new_pic()]

        def enc_anchor(d, kind, prev, out):
            q = qp if kind == "I" else qp + 1
            fr.set_qp(q, lambda_q4(q, inter=kind != "I"))
            fr.encode_picture(src_of(d), slots[prev] if prev is not None else slots[out], kind == "I", slots[out])

        def enc_b(d, s0, s1):
            fr.set_qp(qp + 2, lambda_q4(qp + 2, inter=True))
            fr.encode_picture_b(src_of(d), slots[s0], slots[s1], bout)

        def bcast(slot, src):
            if dist is not None:
                for t in (slots[slot].y, slots[slot].u, slots[slot].v):
                    dist.broadcast(t, src=src)        # RCCL over xGMI: the only exchange step of the path

        per = nb + 1
        n_mg = max(1, (args.steps * world) // per)
        gop.spread_b(rank, world, max(1, args.warmup // per), nb, enc_anchor, enc_b, bcast)
        barrier()
        t0 = time.perf_counter()
        gop.spread_b(rank, world, n_mg, nb, enc_anchor, enc_b, bcast)
        barrier()
        dt = time.perf_counter() - t0
        total_pictures = 1 + n_mg * per
    else:
        for _ in range(args.warmup):
            step()
        # two marker kernels bracket the timed region in a kernel trace (tools/rocpd_stats.py then reports exactly these launches)
        barrier()
        ks.marker(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        ks.marker(2)
        total_pictures = world * args.steps * nstreams
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=ks.device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fps = total_pictures / dt

    if rank == 0:
        # ---- PSNR-Y of the reconstructed pictures (untimed pass; CPSNR_I420::calcPSNR enc@0x4c4060)
        sse = 0
        npic = min(8, len(order))
        for i in range(npic):
            sh0.step()
            d, pic = state["last"]
            s = fr.sse_picture(src_of(d), pic)
            sse += int(s[0])
        mse = sse / (npic * W * H)
        psnr_y = 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)

        # ---- per-stage HIP-event timing IN SITU: events recorded on the kernels' own stream between the stages of
        #      ks265_encode_picture while the real pipeline runs -> roofline of the dominant kernel
        P = float(W * H)
        fr.set_profiling(True)

        probe_out = fr.new_pic()

        def stage_pass(all_shards):
            acc, nacc = {}, 0
            for _ in range(24):
                (step if all_shards else sh0.step)()
                if state["kind"] == "Pm":              # multi-reference run: time the stages on an extra single-reference picture (same kernels per launch)
                    fr.encode_picture(src_of(state["last"][0]), state["ref0"], False, probe_out)
                elif state["kind"] != "P":             # stage events are recorded by ks265_encode_picture (I / P pictures)
                    continue
                ms = fr.stage_ms()
                ms["me_int_kernel"] = fr.me_int_ms()      # the SAD kernel alone (HIP events around its launch on the frame's stream): stage me_integer also holds pre-search + propagation
                for k, v in ms.items():
                    acc[k] = acc.get(k, 0.0) + v
                nacc += 1
            torch.cuda.synchronize()
            return {k: v / nacc for k, v in acc.items()}

        torch.cuda.synchronize()
        stage_alone = stage_pass(False)                 # shard 0 alone on the GPU: the kernels' own durations (= a --streams 1 run)
        stage_run = stage_pass(True) if nstreams > 1 else stage_alone   # event-to-event intervals on shard 0's stream while the other shards run
        # one key picture on its own (not part of the schedule): intra mode pre-selection + wavefront reconstruction + loop filters
        fr.set_qp(qp, lambda_q4(qp))
        kout = fr.new_pic()
        torch.cuda.synchronize()
        fr.encode_picture(srcs[0], kout, True, kout)
        torch.cuda.synchronize()
        key_ms = {k: round(v, 3) for k, v in fr.stage_ms().items() if v > 0}
        fr.set_profiling(False)
        me_int_kernel_ms = stage_alone.pop("me_int_kernel"); stage_run.pop("me_int_kernel", None)
        # round 6: the roofline's kernel ON THE GOP OF `value`.  With the default command `value` is the hierarchical-B 8 GOP; there me_int_kernel is launched once
        # per P picture and twice per B picture (list 1's search runs beside list 0's on the frame's side stream), so a launch takes longer than alone in an IPPP chain.
        # A device-resident pyramid of 8 on a frame object of its own (one stream), HIP events around the MAIN chain's launch of every picture (ks265_frame_me_int_ms)
        hier_me_ms = None
        if getattr(args, "ref_hier_b", 0) == 8 and not args.hier_b:
            R0 = max(1, min(4, args.ref0))
            frh = KsFrame(ks, W, H, qp, lambda_q4(qp), bframes=7, refs=R0, **hot_tools(args, me_method))
            frh.set_profiling(True)
            NH = 8 * R0 + 1
            hd = [frh.new_pic() for _ in range(NH)]
            hs = gop.hier_order(8, args.iper)
            acc_p, acc_b, n_p, n_b = 0.0, 0.0, 0, 0
            hist = []
            for i in range(1 + 8 * 6):
                d, kind, r0, r1, layer = next(hs)
                q = qp + host_qp_offset(kind, layer=layer, hier=True)
                frh.set_qp(q, lambda_q4(q, inter=kind != "I"))
                frh.set_picture_tools(*((0, 0, 0, 1 if os.environ.get("KS265_LEAN_B") == "2" and me_method == 2 else -1) if kind == "B" and layer == 3 and not os.environ.get("KS265_LEAN_B") == "0" else (0, -1, 0, -1) if kind == "B" and layer == 2 and os.environ.get("KS265_LEAN_B") not in ("0", "3") else (-1, -1, -1, -1)))      # the host's lean B pictures
                out = hd[d % NH]
                if kind == "B":
                    frh.encode_picture_b(src_of(d), hd[r0 % NH], hd[r1 % NH], out)
                elif kind == "P" and R0 > 1 and len(hist) > 1 and hist[0] == r0:      # -ref0: the anchor searches the last ref0 anchors (one me_int_kernel launch per picture searched; the events time the last one)
                    frh.encode_picture_mref(src_of(d), [hd[p % NH] for p in hist[:R0]], out)
                else:
                    frh.encode_picture(src_of(d), hd[r0 % NH] if r0 is not None else out, kind == "I", out)
                if kind == "I":
                    hist = [d]
                elif kind == "P":
                    hist = [d] + hist[:3]
                torch.cuda.synchronize()
                if i > 8 and kind != "I":                       # (the first mini-GOP warms the frame object up)
                    v = frh.me_int_ms()
                    if v > 0:
                        if kind == "B": acc_b += v; n_b += 1
                        else: acc_p += v; n_p += 1
            frh.set_profiling(False); frh.close()
            if n_p and n_b:
                tp, tb = acc_p / n_p, acc_b / n_b
                hier_me_ms = {"p_picture": round(tp, 4), "b_picture_list0": round(tb, 4), "per_launch": round((R0 * tp + 14 * tb) / (R0 + 14), 4),
                              "launches_per_mini_gop": R0 + 14, "what": f"me_int_kernel by HIP events in a device-resident pyramid of 8 on one stream: {R0} launches per anchor (-ref0 {R0}: one per picture searched), 2 per B picture (the two lists' searches side by side: the event pair times list 0's launch while list 1's runs on the side stream)"}
        stage_ms = stage_alone
        dom_stage = max(stage_ms, key=stage_ms.get)
        # the roofline figure is the SAD kernel's (north_star; VERDICT r3: the kernel alone, not the stage): algorithmic bytes / its own launch duration
        dom = "me_integer"
        algo_bytes = ALGO_BYTES_P[dom] * P
        if not me_int_kernel_ms or me_int_kernel_ms <= 0:      # ks265_frame_me_int_ms had no valid event pair: fall back to the stage's interval, which contains the kernel
            me_int_kernel_ms = stage_ms["me_integer"]
        achieved = algo_bytes / (me_int_kernel_ms * 1e-3) / 1e9
        # counters under profiles/ are inputs measured by a rocprofv3 --pmc run of an EARLIER invocation: they count only if they were taken on this very kernel source
        from ks265codec_amd.build import source_sha
        my_sha, stale = source_sha(), []
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/hbm_traffic.py)
        if os.path.exists(tfile):
            tall = json.load(open(tfile)).get(f"{W}x{H}", {})
            if tall.get("_stamp", {}).get("kernel_src_sha") != my_sha:
                stale.append("hbm_traffic.json")
            elif tall.get(dom):
                traffic = tall[dom]["bytes_per_launch"]
        # second figure per stage: the VALU-issue fraction = wave-level VALU instructions per launch (SQ_INSTS_VALU of a rocprofv3 --pmc pass,
        # profiles/sq_counters.json by tools/sq_issue.py) / 1.23e12 per second / the stage's duration - the kernels of this path are bound by latency and issue, not bytes
        valu = {}
        sfile = os.path.join(ROOT, "profiles", "sq_counters.json")
        if os.path.exists(sfile):
            sq = json.load(open(sfile)).get(f"{W}x{H}", {})
            if sq.get("_stamp", {}).get("kernel_src_sha") != my_sha:
                stale.append("sq_counters.json"); sq = {}
            for k, v in stage_ms.items():
                n = sum(sq.get(kk, {}).get("insts_valu_per_launch", 0) for kk in ((k, "merge_pass") if k == "cu_decide" else (k,)))
                if n and v > 0:
                    valu[k] = round(n / 1.23e12 / (v * 1e-3), 4)
        ippp_fig = {"achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5), "avg_launch_ms": round(me_int_kernel_ms, 4), "gop": "IPPP (device-resident leg)"}
        if hier_me_ms:                                            # the headline's GOP: this is the figure of `roofline`; the IPPP one stays beside it
            me_int_kernel_ms = hier_me_ms["per_launch"]
            achieved = algo_bytes / (me_int_kernel_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "me_int_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "gop": "hierarchical-B 8 (the GOP of value)" if hier_me_ms else "the device-resident leg's",
                    "in_the_gop_of_value": hier_me_ms, "roofline_ippp": ippp_fig if hier_me_ms else None,
                    "counters_stale": stale or None, "kernel_src_sha": my_sha,
                    "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(me_int_kernel_ms, 4), "longest_stage": dom_stage,
                    "stages_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                    "streams_in_run": nstreams,
                    "stage_intervals_ms_in_run": {k: round(v, 4) for k, v in stage_run.items()},
                    "note": "avg_launch_ms / stages_ms / frac: HIP events around each stage with ONE shard on the GPU = the kernel's own duration; it agrees with the rocprofv3 kernel trace of `bench.py --streams 1` (profiles/). With the run's --streams shards in flight the kernels of different shards overlap: stage_intervals_ms_in_run are event-to-event intervals on one shard's stream under that load (queueing behind the other shards' kernels included), the kernel durations of that condition are in the rocprofv3 trace of the default command (profiles/).",
                    "stages_frac": {k: round(ALGO_BYTES_P[k] * P / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) for k, v in stage_ms.items()},
                    "valu_issue_frac": (round(valu["me_integer"] * stage_ms["me_integer"] / ippp_fig["avg_launch_ms"], 4) if valu.get("me_integer") else None), "stages_valu_issue_frac": valu or None,
                    "bound_note": "no stage of a P picture is limited by HBM bytes: each is a dependency chain per CTU (search state machines, the intra CUs' wavefront) or issue-bound "
                                  "arithmetic on L2-resident samples; valu_issue_frac = VALU wave-instructions / 1.23e12 per s / duration says how busy the vector ALUs are"}

        # ---- CPU baseline on this box's host cores: the reference's own CLI encoder when it is staged (oracle/_ref/appencoder or $KS265_REF_ENCODER,
        #      SURVEY.md §8d-iii), else the oracle port; a bounded sample of the same workload either way
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = reference_leg(args, clip, order) or port_leg(args, clip, order, me_method)

        bf_desc = f"{args.hier_b - 1} (hierarchical GOP {args.hier_b}, B-ref)" if args.hier_b else str(args.bframes)
        line = {
            "metric": "encoded frames/sec + PSNR-Y, 2160p -preset slow -qp 27, 1/2/4/8 GPU",
            "value": round(fps, 2), "unit": "frames/s", "psnr_y": round(float(psnr_y), 3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} 4:2:0 8-bit, hot path only (ME + transform/quant/recon + deblock + SAO; CABAC/RC not included), "
                                   f"-rc 0 -qp {qp} (the encoder host's ladders = the reference's: I = Q; IPPP P = Q + 1 + 0 / 2 / 1 / 2 over four pictures; pyramid anchors Q + 1, B layers + 2 / + 4 / + 4; plain B = Q + 2) -iper {args.iper}, -bframes {bf_desc}, -ref {max(1, args.refs)} -ref0 {max(1, args.ref0) if args.hier_b else max(1, args.refs)} ({'the anchors of the pyramid search the last ref0 anchors of their GOP, B pictures one picture per list' if args.hier_b else 'one reference picture per list' if args.refs <= 1 else 'every P picture searches that many list-0 pictures'}), -me {me_method} ({args.me.upper()}{', interMeHex below ' + str(args.me_hex_thr) + ' SAD/sample as at -preset slow' if me_method == 2 and args.me_hex_thr else ''}) range 64, -subme 1 as the reference runs it at -preset slow (fast candidate sets judged by SAD + rate; DESIGN.md 5e), sao on, df on",
                       "pictures_per_step": nstreams, "streams_per_gpu": nstreams,
                       "key_picture_ms": {"intra_decide": key_ms.get("intra_candidates"), "intra_reconstruct": key_ms.get("intra_pass"), "total": round(sum(key_ms.values()), 3),
                                          "note": f"one intra picture per -iper {args.iper} pictures; it is in the timed region whenever the schedule puts one there"},
                       "sharding": "anchor chain rotating over the ranks + RCCL broadcast of every reconstructed anchor from its owner, B pictures spread over the ranks not coding an anchor" if args.b_spread else f"{nstreams} GOP shard(s) in flight per GPU on separate HIP streams, no data-path collective"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if encoded is not None:
            line = encoded_line(args, encoded, world, line, cpu)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for sh in shards:
        sh.fr.close()
    ks.close()


def host_cores():
    try:
        return int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def port_leg(args, clip, order, me_method):
    """cpu_baseline kind "port": oracle/ks265_pipeline_oracle.c (OpenMP over CTUs) on a few pictures of the bench clip"""
    from oracle_lib import OraclePipeline
    from ks265codec_amd.synth import lambda_q4
    W, H, qp = args.width, args.height, args.qp
    o = OraclePipeline(W, H, qp, lambda_q4(qp), **hot_tools(args, me_method))
    nbase = 3
    tc0 = time.perf_counter()
    if args.bframes == 0:
        t = 0
        while t < 3 or (time.perf_counter() - tc0 < 8.0 and t < 24):     # 1 key + P pictures until ~10 s of wall time (bounded)
            q = qp if t == 0 else qp + 1
            o.set_qp(q, lambda_q4(q))
            o.encode_picture(clip[order[t % len(order)]], t == 0)
            t += 1
        nbase = t
    else:                                       # I0, P2, B1 of the same clip
        o.set_qp(qp, lambda_q4(qp)); i0 = o.encode(clip[0], "I")
        o.set_qp(qp + 1, lambda_q4(qp + 1)); p2 = o.encode(clip[2], "P", i0)
        o.set_qp(qp + 2, lambda_q4(qp + 2)); o.encode(clip[1], "B", i0, p2)
    tc = time.perf_counter() - tc0
    ncores = host_cores()
    return {"value": round(nbase / tc, 4), "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": f"{nbase} pictures (1 key + {nbase - 1} {'P' if args.bframes == 0 else 'P/B'}) of the same {W}x{H} clip, oracle/ks265_pipeline_oracle.c, "
                      f"OpenMP over CTUs on {ncores} host threads (the intra wavefront of the key picture is sequential), {tc:.1f} s; pixel path only (no CABAC)"}


def reference_leg(args, clip, order):
    """cpu_baseline kind "reference": the reference's CLI encoder (appencoder V2.6.1.3, /root/reference/ubuntu_x64, staged by __graft_entry__.build() as
    oracle/_ref/appencoder; $KS265_REF_ENCODER overrides) on a bounded run of the bench clip, on this box's host cores.  It is the WHOLE reference encoder
    (lookahead, RDO, CABAC) at the preset the metric names: the same job the `encoded` value measures.  None when no executable is there."""
    import re, subprocess, tempfile
    exe = os.environ.get("KS265_REF_ENCODER") or os.path.join(ROOT, "oracle", "_ref", "appencoder")
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        return None
    W, H = args.width, args.height
    cores = host_cores()
    threads = int(os.environ.get("KS265_REF_THREADS", "0")) or min(cores, 64)
    n = int(os.environ.get("KS265_REF_FRAMES", "0")) or int(min(256, max(128, 128 * (3840 * 2160) // (W * H))))     # round 6: a whole -iper 128 GOP at 2160p - the reference's frame-parallel pipeline needs that many pictures to fill (64 pictures: 45 pictures/s, 128: 55 - 59)
    preset = "slow" if args.me == "umh" and args.me_hex_thr == 16 else "veryslow" if args.me == "umh" else "medium"
    with tempfile.TemporaryDirectory(prefix="ks265_ref_") as td:
        yuv = os.path.join(td, "clip.yuv")
        with open(yuv, "wb") as f:
            for t in range(n):
                f.write(clip[order[t % len(order)]].tobytes())
        hb = getattr(args, "ref_hier_b", args.hier_b)                # the GOP of `value` (the hot leg may have switched args.hier_b off for itself)
        bf = [] if hb == 8 else ["-bframes", str(hb - 1 if hb else args.bframes)]
        cmd = [exe, "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "50", "-preset", preset, "-rc", "0", "-qp", str(args.qp), "-iper", str(args.iper),
               "-threads", str(threads), "-psnr", "1", "-b", os.path.join(td, "o.265"), *bf]
        t0 = time.perf_counter()
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, cwd=td, timeout=600).stdout
        except Exception as e:                                   # a broken executable must not take the bench line down
            print(f"[bench] reference leg failed: {e!r}", file=sys.stderr)
            return None
        wall = time.perf_counter() - t0
    m = re.search(r"FPS:\s*([0-9.]+)", out)
    b = re.search(r"bitrate, psnr:\s*([0-9.]+)\s+([0-9.]+)", out)
    if not m:
        print("[bench] reference leg: no FPS line in appencoder's output", file=sys.stderr)
        return None
    return {"value": float(m.group(1)), "unit": "frames/s", "cores": threads, "kind": "reference",
            "kbps_at_50fps": float(b.group(1)) if b else None, "psnr_y": float(b.group(2)) if b else None,
            "sample": f"appencoder V2.6.1.3 -preset {preset} -rc 0 -qp {args.qp} -iper {args.iper} {' '.join(bf) or '(default hierarchical-B GOP 8)'} -threads {threads} on {n} pictures of the "
                      f"same {W}x{H} clip (ping-pong over {len(clip)} pictures), {wall:.1f} s wall incl. reading the .yuv; box has {cores} host threads; FPS as the encoder prints it"}


def encoded_leg(args, torch, dist, rank, world, dev_index, backend, sync_all, shared_clip=None):
    """The whole encoder on this rank's GPU through the SDK-compatible C API (libks265enc.so, include/ks265_enc.h): per step one host picture goes
    in (QY265EncoderEncodeFrame: copy to pinned memory, H2D, pixel path, D2H of the records, CABAC on the writer threads), NAL units come out.
    Timed: `steps` pictures incl. the flush that drains the pipeline, bracketed by barriers; the maximum over ranks counts."""
    import ctypes as C
    from ks265codec_amd import stream
    from ks265codec_amd.synth import make_clip
    lay = json.load(open(os.path.join(ROOT, "tests", "golden", "qy265_layout.json")))
    lib = C.CDLL(stream.build())
    lib.QY265EncoderOpen.restype = C.c_void_p

    class YUV(C.Structure):
        _fields_ = [("iWidth", C.c_int), ("iHeight", C.c_int), ("pData", C.POINTER(C.c_ubyte) * 3), ("iStride", C.c_int * 3)]

    class Picture(C.Structure):
        _fields_ = [("iSliceType", C.c_int), ("poc", C.c_int), ("pts", C.c_longlong), ("dts", C.c_longlong), ("yuv", C.POINTER(YUV))]

    class Nal(C.Structure):
        _fields_ = [("naltype", C.c_int), ("tid", C.c_int), ("iSize", C.c_int), ("pts", C.c_longlong), ("pPayload", C.POINTER(C.c_ubyte))]

    class Stats(C.Structure):
        _fields_ = [("frames", C.c_long), ("bytes", C.c_longlong), ("sse", C.c_double * 3), ("gpu_ms", C.c_double), ("host_write_ms", C.c_double),
                    ("in_copy_ms", C.c_double), ("submit_ms", C.c_double), ("output_ms", C.c_double), ("lat_gpu_ms", C.c_double), ("lat_queue_ms", C.c_double), ("key_wall_ms", C.c_double), ("key_cpu_ms", C.c_double), ("keys", C.c_long),
                    ("occ_samples", C.c_long), ("occ_ring", C.c_long), ("occ_gpu", C.c_long), ("occ_ready", C.c_long), ("submit_wait_ms", C.c_double)]

    W, H = args.width, args.height
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 8
    threads = args.host_threads or max(2, min(32, cores // max(1, world)))        # slice writers of this rank's encoder (shared out over its GOP lanes, if any)
    os.environ["KS265_DEVICE"] = str(dev_index)            # one process per GPU (the driver's launch contract): this rank's encoder handle owns exactly its GPU
    os.environ.pop("KS265_GPUS", None); os.environ.pop("KS265_DEVICES", None)
    cfg = (C.c_uint8 * lay["sizeof_config"])()
    preset = b"slow" if args.me == "umh" and args.me_hex_thr == 16 else b"veryslow" if args.me == "umh" else b"medium"
    assert lib.QY265ConfigDefaultPreset(cfg, preset, None, b"default") == 0
    gop_b = (args.hier_b - 1) if args.hier_b else args.bframes
    for k, v in (("wdt", W), ("hgt", H), ("fr", 50), ("rc", 0), ("qp", args.qp), ("iper", args.iper), ("bframes", -1 if args.hier_b == 8 else gop_b), ("threads", threads),
                 ("psnr", 1), ("log", 3), ("lookahead", args.lookahead), ("me", {"dia": 0, "hex": 1, "umh": 2}[args.me]), ("subme", 1), ("ref", max(1, args.refs))):
        assert lib.QY265ConfigParse(cfg, k.encode(), str(v).encode()) == 0, k
    clip = shared_clip if shared_clip is not None else make_clip(W, H, args.clip_frames, seed=7 + rank, abc=(67, 91, 33), pan=(8, 5))
    order = list(range(len(clip))) + list(range(len(clip) - 2, 0, -1))          # ping-pong keeps the motion continuous
    strong = args.scaling == "strong"
    if strong:
        clip = make_clip(W, H, args.clip_frames, seed=7, abc=(67, 91, 33), pan=(8, 5))      # ONE job: every rank reads the same clip, its own GOPs of it

    def open_encoder():
        err = C.c_int(0)
        if USER_LANES is None:                           # the application's choice, as ks265enc makes it: two closed GOPs side by side for the pyramid GOPs, one lane for IPPP / plain B
            os.environ["KS265_GOP_LANES"] = "2" if (args.hier_b == 8 or gop_b == 3) and args.iper >= 32 and not strong else "1"
        hh = C.c_void_p(lib.QY265EncoderOpen(cfg, C.byref(err)))
        if not hh.value:
            raise SystemExit(f"QY265EncoderOpen failed: 0x{err.value & 0xFFFFFFFF:08x} (the encoder needs the MI355X: there is no CPU fallback)")
        return hh

    h = open_encoder()
    nal, nn, pic, outp, yuv = C.POINTER(Nal)(), C.c_int(0), Picture(), Picture(), YUV()
    yuv.iWidth, yuv.iHeight = W, H
    yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W, W // 2, W // 2
    pic.yuv = C.pointer(yuv)
    state = {"t": 0, "bytes": 0, "nals": 0, "pics": 0, "keep": None}
    lib.ks265_enc_lanes.restype = C.c_int
    lanes = lib.ks265_enc_lanes(h)

    def collect():
        state["nals"] += nn.value
        for i in range(nn.value):
            state["bytes"] += nal[i].iSize
            state["pics"] += nal[i].naltype < 32                 # VCL NAL units = coded pictures (one slice each)
            if state["keep"] is not None:
                state["keep"].append(C.string_at(nal[i].pPayload, nal[i].iSize))

    def feed(n):
        for _ in range(n):
            fr = clip[order[state["t"] % len(order)]]
            if args.zero_copy and lib.ks265_enc_acquire_input(h, C.byref(yuv)) == 0:
                C.memmove(yuv.pData[0], fr.ctypes.data, W * H * 3 // 2)          # the application's own work: producing the picture where the encoder reads it
            else:
                yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W, W // 2, W // 2
                for k, off in enumerate((0, W * H, W * H * 5 // 4)):
                    yuv.pData[k] = C.cast(fr.ctypes.data + off, C.POINTER(C.c_ubyte))
            pic.pts = state["t"]
            state["t"] += 1
            rc = lib.QY265EncoderEncodeFrame(h, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(outp), 0)
            assert rc == 0, hex(rc & 0xFFFFFFFF)
            collect()

    def feed_until(npics):
        """keep feeding until `npics` coded pictures have come OUT of the encoder (GOP lanes: input runs ahead of the output in bursts, the output side is the clock)"""
        while state["pics"] < npics:
            feed(1)

    def flush():
        while lib.QY265EncoderDelayedFrames(h):
            rc = lib.QY265EncoderEncodeFrame(h, C.byref(nal), C.byref(nn), None, C.byref(outp), 0)
            assert rc == 0, hex(rc & 0xFFFFFFFF)
            collect()

    job = None
    win = None
    if strong:
        feed(args.warmup); flush()
        # the fixed job: pictures 0 .. F-1 of the clip in closed GOPs of -iper pictures; rank r takes a contiguous run of GOPs (ks265codec_amd/gop.py shard_gops),
        # on a fresh encoder (the warm-up one is closed: its pictures are not part of the job); the ranks' streams are gathered on rank 0 in GOP order
        from ks265codec_amd import gop
        F, per = args.job_frames, (args.iper if args.iper > 0 else args.job_frames)
        mine = gop.shard_gops(F, per, world, rank, contiguous=True)
        lib.QY265EncoderClose(h)
        h = open_encoder()
        state.update(bytes=0, nals=0, keep=[])
    if strong:
        sync_all()
        t0 = time.perf_counter()
        for a, b in mine:
            state["t"] = a
            feed(b - a)
        flush()
        blob = b"".join(state["keep"])
        if dist is not None:
            dv = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
            sizes = [torch.zeros(1, dtype=torch.int64, device=dv) for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([len(blob)], dtype=torch.int64, device=dv))
            cap = int(max(int(x.item()) for x in sizes))
            buf = torch.zeros(cap, dtype=torch.uint8)
            buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            buf = buf.to(dv)
            parts = [torch.empty(cap, dtype=torch.uint8, device=dv) for _ in range(world)] if rank == 0 else None
            dist.gather(buf, parts, dst=0)              # the only exchange step of the strong-scaling job: coded bytes to rank 0 (RCCL when backend = nccl)
            if rank == 0:
                blob = b"".join(bytes(p[:int(n.item())].cpu().numpy().tobytes()) for p, n in zip(parts, sizes))
        job = {"frames": F, "gops": -(-F // per), "bytes": len(blob) if rank == 0 else 0, "frames_this_rank": sum(b - a for a, b in mine)}
        if rank == 0 and args.out:
            open(args.out, "wb").write(blob)
        sync_all()
        dt = time.perf_counter() - t0
    else:
        # Weak mode: STEADY-STATE throughput of the asynchronous encoder (output lags input by the SDK's contract).  Untimed: the W warm-up pictures and
        # as many more as it takes to fill the pipeline and to stand just behind a key picture (the ring of pictures in flight is then full and every
        # EncodeFrame call returns only when a picture has left the encoder: back-pressure = one picture in, one picture out).  Timed window A: exactly
        # K pictures (no key picture among them when K < iper).  Timed window B (unless A spans four GOPs itself): exactly FOUR whole GOPs of -iper
        # pictures each = iper - 1 P/B pictures + ONE key picture (one GOP when -iper > 256).  `value` is the whole-GOP rate (the key picture's share included); A is reported beside it.
        iper = args.iper if args.iper > 0 else 1 << 30
        if lanes > 1:
            # GOP lanes: `lanes` closed GOPs are coded at once and handed out in GOP order; every lane buffers a whole GOP of input beyond the one it is coding, and
            # while the caller sits in a device synchronize the lanes go on coding from those buffers - "pictures in" or "pictures out" between two synchronizes
            # is then not the number of pictures the GPU coded between them.  So each window is a closed piece of work: the encoder is EMPTY on both sides
            # (flush + barrier + device synchronize), the pictures fed in between are exactly the pictures coded in between.  Untimed: the warm-up and two whole
            # rounds of `lanes` GOPs (graphs captured, buffers touched).  Window A = --steps pictures.  Window B = four whole rounds (4 x `lanes` GOPs, as many key
            # pictures) = `value`; it includes starting from and draining to an empty pipeline, which a long-running encoder pays once (about 5 % of the window).
            rnd = lanes * iper
            feed(-(-(args.warmup + 2 * rnd) // rnd) * rnd)
            flush(); sync_all()
            t0 = time.perf_counter()
            feed(args.steps); flush()
            sync_all()
            dt_a = time.perf_counter() - t0
            win = {"A": {"pictures": args.steps, "seconds": round(dt_a, 5), "clock": "fed, coded and flushed between two synchronizes"}}
            sync_all()
            o0, t0 = state["pics"], time.perf_counter()
            feed(4 * rnd)
            t_in = time.perf_counter() - t0
            flush()
            sync_all()
            dt = time.perf_counter() - t0
            npic = 4 * rnd
            assert state["pics"] - o0 == npic, (state["pics"] - o0, npic)
            win["B"] = {"pictures": npic, "seconds": round(dt, 5), "key_pictures": 4 * lanes, "clock": "fed, coded and flushed between two synchronizes", "gop_lanes": lanes,
                        "seconds_feeding": round(t_in, 5)}
        else:
            fill = args.warmup + 132
            if iper < 1 << 20:
                fill = -(-fill // iper) * iper + 1                # first picture of window A = the one right after a key picture
            feed(fill)
            sync_all()
            t0 = time.perf_counter()
            feed(args.steps)
            sync_all()                                            # barrier + device synchronize on both sides (all ranks)
            dt_a = time.perf_counter() - t0
            dt, npic = dt_a, args.steps
            win = {"A": {"pictures": args.steps, "seconds": round(dt_a, 5), "key_pictures": (fill + args.steps - 1) // iper - (fill - 1) // iper}}
            if args.steps < 4 * iper and iper < 1 << 20:               # (A itself is the value only when it spans four GOPs or more)
                pos = fill + args.steps
                feed(-pos % iper + 1 if pos % iper != 1 else 0)   # untimed: up to the picture right after the next key picture
                ngop = 4 if iper <= 256 else 1                     # round 4: four whole GOPs, not one - 0.14 s windows moved by 12 % when one key picture was not hidden under its predecessor GOP
                sync_all()
                t0 = time.perf_counter()
                feed(ngop * iper)
                sync_all()
                dt = time.perf_counter() - t0
                npic = ngop * iper
                win["B"] = {"pictures": npic, "seconds": round(dt, 5), "key_pictures": ngop}
        flush()
    st = Stats()
    lib.ks265_enc_get_stats(h, C.byref(st))
    lib.QY265EncoderClose(h)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", dev_index) if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    mse = st.sse[0] / max(1, st.frames) / (W * H)
    if strong:
        import hashlib
        return {"fps": job["frames"] / dt, "dt": dt, "host_threads": threads, "host_cores": cores, "bytes_per_picture": job["bytes"] / job["frames"], "job": job,
                "md5": hashlib.md5(blob).hexdigest() if rank == 0 else None, "gop_lanes": lanes,
                "psnr_y": 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse), "slice_write_ms_per_picture": st.host_write_ms / max(1, st.frames), "preset": preset.decode(),
                "gop": f"hierarchical-B {args.hier_b}" if args.hier_b else (f"P + {gop_b} B" if gop_b else "IPPP")}
    if dist is not None:
        ta = torch.tensor([win["A"]["seconds"]], dtype=torch.float64, device=torch.device("cuda", dev_index) if backend == "nccl" else "cpu")
        dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        win["A"]["seconds"] = round(float(ta.item()), 5)
    win["A"]["fps_all_ranks"] = round(world * win["A"]["pictures"] / win["A"]["seconds"], 2)
    return {"fps": world * npic / dt, "dt": dt * args.steps / npic, "windows": win, "gop_lanes": lanes, "host_threads": threads, "host_cores": cores, "bytes_per_picture": st.bytes / max(1, st.frames),
            "psnr_y": 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse), "slice_write_ms_per_picture": st.host_write_ms / max(1, st.frames), "preset": preset.decode(),
            "caller_ms_per_picture": {"input_copy": round(st.in_copy_ms / max(1, st.frames), 3), "enqueue": round((st.submit_ms - st.submit_wait_ms) / max(1, st.frames), 3), "enqueue_waiting_for_ring_space": round(st.submit_wait_ms / max(1, st.frames), 3), "output": round(st.output_ms / max(1, st.frames), 3),
                                      "latency_enqueue_to_records_on_host": round(st.lat_gpu_ms / max(1, st.frames), 2), "latency_enqueue_to_writer_pickup": round(st.lat_queue_ms / max(1, st.frames), 2),
                                      "key_picture_slice_wall_ms": round(st.key_wall_ms / max(1, st.keys), 2), "key_picture_slice_thread_ms": round(st.key_cpu_ms / max(1, st.keys), 2),
                                      "ring_occupancy_at_submission": {"in_ring": round(st.occ_ring / max(1, st.occ_samples), 1), "not_through_gpu": round(st.occ_gpu / max(1, st.occ_samples), 1),
                                                                       "waiting_for_a_writer": round(st.occ_ready / max(1, st.occ_samples), 1)}},
            "lookahead": ("off" if args.lookahead == 0 else f"-lookahead {args.lookahead}: scene cuts + slice types" if args.lookahead > 0 else
                          "default: slice-type decision (blocks of 8 coded as 8 or 4 + 4) from the half-size pictures on the GOP's grid of 4" if args.hier_b == 8 else "default: off"),
            "gop": f"hierarchical-B {args.hier_b}" if args.hier_b else (f"P + {gop_b} B" if gop_b else "IPPP")}


def encoded_line(args, enc, world, hot, cpu):
    """the bench line: value = encoded frames/s (the whole encoder); the device-resident pixel-path leg (if run) supplies roofline / stage times"""
    W, H = args.width, args.height
    strong = args.scaling == "strong"
    steps = enc["job"]["frames_this_rank"] if strong else args.steps          # strong: pictures rank 0 coded of the fixed job
    line = {
        "metric": "encoded frames/sec + PSNR-Y, 2160p -preset slow -qp 27, 1/2/4/8 GPU",
        "value": round(enc["fps"], 2), "unit": "frames/s", "psnr_y": round(float(enc["psnr_y"]), 3),
        "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * enc["dt"] / steps, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{W}x{H} 4:2:0 8-bit synthetic clip, ENCODED end to end through the SDK-compatible C API (QY265EncoderEncodeFrame): host I420 in -> pinned copy -> H2D -> "
                               f"pixel path on the MI355X (-preset {enc['preset']}: -me {args.me}, subme 1, deblock + SAO) -> D2H of CU map / levels / SAO -> CABAC slice data on "
                               f"{enc['host_threads']} host threads -> Annex-B NAL units out; -rc 0 -qp {args.qp} (I = Q; IPPP: P = Q + 1 + the reference's cascade 2 / 1 / 2 / 0 over four pictures; pyramid: anchors Q + 1, B layers + 2 / + 4 / + 4 as in the reference) -iper {args.iper}, {enc['gop']}, "
                               f"-ref {max(1, args.refs)} -ref0 3 (the preset's: an anchor of a pyramid GOP searches the last three anchors of its GOP); the stream decodes with the reference's appdecoder to the encoder's reconstruction (tests/test_stream.py)",
                   "timed": None if strong else (("the asynchronous encoder with %d GOP lanes (closed GOPs coded concurrently on the one GPU, output in stream order).  A lane buffers a whole GOP "
                             "of input and goes on coding while the caller sits in a synchronize, so each window is a closed piece of work: flush + barrier + device synchronize on both sides; "
                             "A = --steps pictures, B = four whole rounds of lanes x iper pictures (one key picture per GOP) fed, coded and flushed in between.  value = pictures / seconds of B "
                             "(starting from and draining to an empty pipeline included); ms_per_step = 1000 / value per GPU" % enc["gop_lanes"]) if enc.get("gop_lanes", 1) > 1 else
                             "steady state of the asynchronous encoder (pipeline full before and after, back-pressure: one picture in = one picture out), barrier + "
                             "device synchronize on both sides of each window.  A = exactly --steps pictures; B = four whole GOPs of -iper pictures incl. their key pictures "
                             "(run unless A spans four GOPs itself).  value = pictures / seconds of B (of A in that case); ms_per_step = 1000 / value per GPU"),
                   "gop_lanes": enc.get("gop_lanes", 1),
                   "windows": enc.get("windows"),
                   "pictures_per_step": 1, "host_threads_per_gpu": enc["host_threads"], "host_cores": enc["host_cores"],
                   "bytes_per_picture": int(enc["bytes_per_picture"]), "kbps_at_50fps": round(enc["bytes_per_picture"] * 8 * 50 / 1000.0, 1),
                   "bytes_per_picture_note": "all pictures this encoder coded in the run (warm-up, fill and the timed windows; key pictures in their proportion)",
                   "slice_write_ms_per_picture_per_thread": round(enc["slice_write_ms_per_picture"], 2),
                   "caller_ms_per_picture": enc.get("caller_ms_per_picture"),
                   "in_the_path": "pyramid pre-search, merge pass (merge / skip decided on SATD + rate, signalled where the motion equals a merge candidate), AMVP with the better of the two "
                                  "predictors, vector propagation between neighbouring PUs (stage A2, one round), bi-prediction judged at 31/32 (the joint refinement of the pairs runs from -preset slower on: at -preset slow it cost 0.5 - 0.7 % bytes at equal PSNR-Y and 134 us per B picture, measured), the anchors of a pyramid searching the last three anchors (-ref0 3), intra CUs in P / B pictures, the skip pass over B pictures (a CU with residual against its first two merge candidates without residual: SSE of the real reconstruction + level bits), lean B pictures (the B pictures nothing predicts from - half of a pyramid of 8 - without intra candidates, joint refinement and SAO: ks265_frame_set_picture_tools; bytes at equal PSNR-Y unchanged), coefficient-group pruning (luma) and sign-data hiding "
                                  "(signBitHidingHDQ) at the postQuant seam, P / B lambda table; fractional samples interpolated on the fly; the picture's drain (SSE, packing of the records) and the next source picture's unpack on side streams (DESIGN.md 6a)",
                   "not_in_the_path": "the reference's rdoQuant (an option: -rdoq 1; the presets' rdoq = 1 runs this build's seam), CU size judged with the residual's cost, generalised B pictures; "
                                      "at equal PSNR-Y the stream is 0.99x (2160p) / 1.02x (1080p) the size of appencoder's for IPPP and 1.19x / 1.00x for the default GOP on the same (ping-pong) clips, "
                                      "1.14x / 1.14x and 1.08x / 1.12x on clips without repeats (measured on the MI355X in round 6: profiles/r06_same_clips_equal_psnr.txt, r06_straight_clips_equal_psnr.txt)",
                   "lookahead": enc.get("lookahead"),
                   "sharding": (f"ONE job of {enc['job']['frames']} pictures = {enc['job']['gops']} closed GOPs dealt to the ranks in contiguous runs; every rank encodes its GOPs, the coded bytes are "
                                f"gathered on rank 0 in stream order (the only exchange step; inside the timed region); stream md5 {enc['md5']}") if strong else
                               "one encoder per GPU (rank), each on its own synthetic clip = its own closed GOPs; no data-path collective"},
        "roofline": None, "cpu_baseline": cpu,
    }
    if enc.get("ippp"):
        line["ippp"] = enc["ippp"]
    if enc.get("two_lanes"):
        line["two_gop_lanes"] = enc["two_lanes"]
    if strong:
        line["config"]["job"] = {"frames": enc["job"]["frames"], "gops": enc["job"]["gops"], "stream_bytes": enc["job"]["bytes"], "stream_md5": enc["md5"]}
    if hot is not None:
        line["roofline"] = hot["roofline"]
        line["hot_path"] = {"value": hot["value"], "unit": "frames/s", "psnr_y": hot["psnr_y"], "ms_per_step": hot["ms_per_step"],
                            "what": "device-resident pixel path only (inputs and outputs stay in HBM, no host), IPPP unless --hier-b / --bframes is given - the leg the stage events and the "
                                    "roofline of the SAD kernel are taken on (every picture but the key picture is a P picture): " + hot["config"]["workload"],
                            "streams_per_gpu": hot["config"]["streams_per_gpu"], "key_picture_ms": hot["config"]["key_picture_ms"]}
    return line


if __name__ == "__main__":
    main()
