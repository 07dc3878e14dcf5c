"""CPU, world_size 2 over gloo: the N>1 path of bench.py shards GOPs across ranks with no data-path collective
(SURVEY.md §8e); only a barrier and a MAX all-reduce of the elapsed time cross ranks.  The per-rank work here is the
CPU oracle (no GPU in this container); the sharding / reduction logic is the same code shape as bench.py."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def shard_gops(n_frames: int, iper: int, world: int, rank: int):
    """frames [k*iper, (k+1)*iper) go to rank k % world — closed GOP per shard"""
    return [(k * iper, min((k + 1) * iper, n_frames)) for k in range((n_frames + iper - 1) // iper) if k % world == rank]


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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


def test_gop_sharding_two_ranks():
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
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
    for (a, b) in [(2, 4)]:
        for t in range(a, b):
            rec = o.encode_picture(clip[t], t == a)
            assert int(rec.astype(np.int64).sum()) == res[1][2][t]


def test_shard_gops_covers_everything():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += [t for a, b in shard_gops(1000, 128, world, r) for t in range(a, b)]
        assert sorted(seen) == list(range(1000))
