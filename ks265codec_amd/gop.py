"""GOP scheduling shared by bench.py and the CPU (gloo) tests — pure host logic, no device code.

Two ways to use N GPUs:
  * shard_gops: closed GOPs round-robin over ranks, NO data-path collective (the default; weak scaling);
  * spread_b:   config 5 style — rank 0 codes the anchor chain (I / P pictures) and broadcasts every reconstructed anchor
                (RCCL broadcast over xGMI: 3 padded planes, ~14 MB at 2160p); the non-reference B pictures between two
                anchors are dealt round-robin to the other ranks, which code them while rank 0 already works on the next
                anchor.  The only exchange step of the whole path.
"""
from __future__ import annotations

from typing import Callable, Iterator


def shard_gops(n_frames: int, iper: int, world: int, rank: int) -> list[tuple[int, int]]:
    """frames [k*iper, (k+1)*iper) go to rank k % world — one closed GOP per shard"""
    return [(k * iper, min((k + 1) * iper, n_frames)) for k in range((n_frames + iper - 1) // iper) if k % world == rank]


def coding_order(bframes: int, iper: int) -> Iterator[tuple[int, str]]:
    """(display index, kind) in coding order: I, then per mini-GOP the anchor followed by its B pictures.  A key picture falls on EVERY
    multiple of iper: when iper is not a multiple of bframes + 1 the mini-GOP in front of the boundary is shortened (-iper 128
    -bframes 2 -> ... 126 P, 128 I with one B picture in between), never skipped."""
    d = 0
    yield d, "I"
    while True:
        a = min(d + bframes + 1, (d // iper + 1) * iper)
        yield a, ("I" if a % iper == 0 else "P")
        for b in range(d + 1, a):
            yield b, "B"
        d = a


def hier_order(gop_size: int, iper: int) -> Iterator[tuple[int, str, int | None, int | None, int]]:
    """Hierarchical-B coding order (the reference's default at `-latency offline`: bframes -1 -> GOP 8):
    yields (display index, kind, ref0, ref1, temporal layer); ref0 / ref1 are DISPLAY indices of list-0 (past) / list-1 (future)
    references, all already coded.  Per mini-GOP: the anchor (layer 0), then the middle B picture of every open interval, breadth
    first - GOP 8: 8, 4, 2, 6, 1, 3, 5, 7.  B pictures of all layers but the last are themselves references (B-ref)."""
    assert gop_size >= 1 and gop_size & (gop_size - 1) == 0, "power of two"
    assert iper % gop_size == 0, "key pictures must fall on mini-GOP boundaries (iper % gop_size == 0)"
    yield 0, "I", None, None, 0
    d = 0
    while True:
        a = d + gop_size
        yield a, ("I" if a % iper == 0 else "P"), (None if a % iper == 0 else d), None, 0
        level, layer = [(d, a)], 1
        while level:
            nxt = []
            for lo, hi in level:
                if hi - lo < 2:
                    continue
                mid = (lo + hi) // 2
                yield mid, "B", lo, hi, layer
                nxt += [(lo, mid), (mid, hi)]
            level, layer = nxt, layer + 1
        d = a


def b_owner(j: int, world: int) -> int:
    """rank that codes the j-th B picture of a mini-GOP: ranks 1..world-1 round-robin (rank 0 when alone)"""
    return 0 if world == 1 else 1 + j % (world - 1)


def spread_b(rank: int, world: int, n_minigops: int, bframes: int, encode_anchor: Callable, encode_b: Callable, broadcast: Callable) -> list[tuple[int, str]]:
    """Run `n_minigops` mini-GOPs with the anchor chain on rank 0 and the B pictures spread over the other ranks.

    encode_anchor(d, kind, prev_slot, out_slot)   code picture d ('I'/'P') from the anchor in prev_slot into out_slot   [rank 0 only]
    encode_b(d, slot0, slot1)                      code B picture d between the anchors held in slot0 (past) / slot1 (future)
    broadcast(slot)                                make rank 0's picture in `slot` visible in `slot` on every rank (collective)
    Anchors live in three rotating slots: while the B pictures between anchors k and k+1 are coded from two slots, rank 0 already
    writes anchor k+2 into the third.
    Returns the (display index, kind) pictures THIS rank coded, in order.
    """
    mine: list[tuple[int, str]] = []
    if rank == 0:
        encode_anchor(0, "I", None, 0)
        mine.append((0, "I"))
    broadcast(0)
    for k in range(n_minigops):
        d0, d1 = k * (bframes + 1), (k + 1) * (bframes + 1)
        s0, s1 = k % 3, (k + 1) % 3
        if rank == 0:
            encode_anchor(d1, "P", s0, s1)
            mine.append((d1, "P"))
        broadcast(s1)
        for j, d in enumerate(range(d0 + 1, d1)):
            if b_owner(j, world) == rank:
                encode_b(d, s0, s1)
                mine.append((d, "B"))
    return mine
