"""CPU, world_size 2 over gloo: the N>1 paths of bench.py.
  * default: GOPs sharded across ranks with NO data-path collective - only a barrier and a MAX all-reduce of the
    elapsed time cross ranks;
  * --b-spread (config 5): anchor chain on rank 0, every reconstructed anchor BROADCAST (RCCL on the GPU box, gloo here), the B
    pictures dealt to the other ranks.
The per-rank work here is the CPU oracle (no GPU in this container); the scheduling code (ks265codec_amd/gop.py) is the one bench.py runs."""
from __future__ import annotations

import itertools
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from ks265codec_amd.gop import b_owner, coding_order, shard_gops, spread_b  # noqa: E402


def _setup(rank, world, port):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _shard_worker(rank, world, port, q):
    _setup(rank, world, port)
    from oracle_lib import OraclePipeline
    from ks265codec_amd.synth import lambda_q4, make_clip
    W, H, iper, total = 64, 48, 2, 8
    clip = make_clip(W, H, total, seed=11, abc=(17, 23, 9))
    o = OraclePipeline(W, H, 27, lambda_q4(27))
    mine = {}
    for (a, b) in shard_gops(total, iper, world, rank):
        for t in range(a, b):
            mine[t] = o.encode_picture(clip[t], t == a)          # every shard starts with a key picture
    dist.barrier()
    elapsed = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    count = torch.tensor([len(mine)], dtype=torch.int64)
    dist.all_reduce(count, op=dist.ReduceOp.SUM)
    q.put((rank, sorted(mine), {t: int(v.astype(np.int64).sum()) for t, v in mine.items()}, float(elapsed.item()), int(count.item())))
    dist.barrier()
    dist.destroy_process_group()


def _run(worker, world, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_gop_sharding_two_ranks():
    res = _run(_shard_worker, 2, 29611)
    frames = sorted(sum((r[1] for r in res), []))
    assert frames == list(range(8))                                # every frame coded exactly once
    assert res[0][1] == [0, 1, 4, 5] and res[1][1] == [2, 3, 6, 7]
    assert all(abs(r[3] - 0.2) < 1e-9 for r in res)                # MAX over ranks
    assert all(r[4] == 8 for r in res)                             # whole-job picture count
    # a shard's result equals a 1-rank run on the same chunk (parity contract of SURVEY.md §8e)
    sys.path.insert(0, HERE)
    from oracle_lib import OraclePipeline
    from ks265codec_amd.synth import lambda_q4, make_clip
    clip = make_clip(64, 48, 8, seed=11, abc=(17, 23, 9))
    o = OraclePipeline(64, 48, 27, lambda_q4(27))
    for t in range(2, 4):
        rec = o.encode_picture(clip[t], t == 2)
        assert int(rec.astype(np.int64).sum()) == res[1][2][t]


def _spread_run(rank, world, nmg, nb, bcast):
    """the oracle behind gop.spread_b; returns {display index: checksum of the recon}"""
    sys.path.insert(0, HERE)
    from oracle_lib import HostPic, OraclePipeline
    from ks265codec_amd.synth import lambda_q4, make_clip
    W, H = 64, 48
    clip = make_clip(W, H, 1 + nmg * (nb + 1), seed=5, abc=(17, 23, 9))
    o = OraclePipeline(W, H, 27, lambda_q4(27))
    slots = [HostPic(o.geom) for _ in range(3)]
    sums = {}

    def put(slot, pic):
        for a, b in ((slots[slot].y, pic.y), (slots[slot].u, pic.u), (slots[slot].v, pic.v)):
            a[:] = b

    def enc_anchor(d, kind, prev, out):
        o.set_qp(27 if kind == "I" else 28, lambda_q4(28))
        pic = o.encode(clip[d], kind, slots[prev] if prev is not None else None)
        put(out, pic)
        sums[d] = int(o.store(pic).astype(np.int64).sum())

    def enc_b(d, s0, s1):
        o.set_qp(29, lambda_q4(29))
        sums[d] = int(o.store(o.encode(clip[d], "B", slots[s0], slots[s1])).astype(np.int64).sum())

    mine = spread_b(rank, world, nmg, nb, enc_anchor, enc_b, lambda s, src: bcast(slots[s], src))
    return mine, sums


def _spread_worker(rank, world, port, q):
    _setup(rank, world, port)

    def bcast(pic, src):
        for arr in (pic.y, pic.u, pic.v):
            t = torch.from_numpy(arr)                              # shares memory with the numpy plane
            dist.broadcast(t, src=src)

    mine, sums = _spread_run(rank, world, 4, 3, bcast)
    q.put((rank, mine, sums))
    dist.barrier()
    dist.destroy_process_group()


def test_b_spread_two_ranks_matches_single_process():
    """anchor chain rotating over the two ranks + broadcast from the owner, B pictures on the rank that is not coding the next anchor:
    every picture coded once and identical to the 1-process run"""
    res = _run(_spread_worker, 2, 29613)
    (_, mine0, sums0), (_, mine1, sums1) = res
    assert [d for d, k in mine0 if k != "B"] == [0, 8, 16] and [d for d, k in mine1 if k != "B"] == [4, 12]
    assert sum(k == "B" for _, k in mine0) == 6 and sum(k == "B" for _, k in mine1) == 6
    single_mine, single = _spread_run(0, 1, 4, 3, lambda pic, src: None)
    assert sorted(d for d, _ in mine0 + mine1) == sorted(d for d, _ in single_mine) == list(range(17))
    merged = dict(sums0); merged.update(sums1)
    assert merged == single


def test_schedules_cover_everything():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += [t for a, b in shard_gops(1000, 128, world, r) for t in range(a, b)]
        assert sorted(seen) == list(range(1000))
        # B pictures: never on the rank that codes the next anchor of the chain, and over many mini-GOPs every rank gets its share
        from ks265codec_amd.gop import anchor_owner
        load = [0] * world
        for k in range(8 * world):
            for j in range(3):
                r = b_owner(j, world, k, 3)
                assert world == 1 or r != anchor_owner(k + 2, world)
                load[r] += 1
        assert max(load) - min(load) <= 3, load
        assert sorted(anchor_owner(k, world) for k in range(world)) == list(range(world))
    it = coding_order(3, 8)
    assert [next(it) for _ in range(10)] == [(0, "I"), (4, "P"), (1, "B"), (2, "B"), (3, "B"), (8, "I"), (5, "B"), (6, "B"), (7, "B"), (12, "P")]
    # iper not a multiple of bframes + 1 -> the mini-GOP before the boundary is shortened, a key picture on EVERY multiple of iper
    seq = list(itertools.islice(coding_order(2, 8), 40))
    assert sorted(d for d, _ in seq) == list(range(len(seq))) or sorted(d for d, _ in seq)[:30] == list(range(30))
    keys = [d for d, k in seq if k == "I"]
    assert keys[:4] == [0, 8, 16, 24], keys
    assert [x for x in seq[:8]] == [(0, "I"), (3, "P"), (1, "B"), (2, "B"), (6, "P"), (4, "B"), (5, "B"), (8, "I")]


def test_hierarchical_b_order_is_decodable():
    """hier_order: every picture once, references coded before use, list 0 in the past and list 1 in the future, DPB slot scheme of
    bench.py (display index mod G+1) never overwrites a picture that is still needed"""
    import itertools
    from ks265codec_amd.gop import hier_order
    for G, iper in ((8, 128), (4, 8), (2, 128), (1, 4)):
        seq = list(itertools.islice(hier_order(G, iper), 1 + 5 * G))
        coded, slot = set(), {}
        for d, kind, r0, r1, layer in seq:
            assert d not in coded
            for r in (r0, r1):
                if r is not None:
                    assert r in coded and slot[r % (G + 1)] == r, (G, d, r)      # coded, and still in its DPB slot
            if kind == "B":
                assert r0 < d < r1 and layer >= 1
            elif kind == "P":
                assert r0 == d - G and r1 is None and layer == 0
            else:
                assert d % iper == 0 and r0 is None
            coded.add(d)
            slot[d % (G + 1)] = d
        assert sorted(coded) == list(range(5 * G + 1))
    assert [x[0] for x in itertools.islice(hier_order(8, 128), 9)] == [0, 8, 4, 2, 6, 1, 3, 5, 7]
