"""GOP scheduling shared by bench.py and the CPU (gloo) tests — pure host logic, no device code.

Two ways to use N GPUs:
  * shard_gops: closed GOPs round-robin over ranks, NO data-path collective (the default; weak scaling);
  * spread_b:   config 5 style — the anchor chain (I / P pictures) rotates over the ranks, its owner broadcasts every reconstructed
                anchor (RCCL broadcast over xGMI: 3 padded planes, ~14 MB at 2160p); the non-reference B pictures between two
                anchors are dealt to the ranks that are not coding the next anchor at that moment.  The only exchange step of the
                whole path.
"""
from __future__ import annotations

from typing import Callable, Iterator


def shard_gops(n_frames: int, iper: int, world: int, rank: int, contiguous: bool = False) -> list[tuple[int, int]]:
    """frames [k*iper, (k+1)*iper) go to rank k % world — one closed GOP per shard.  contiguous=True: rank r takes GOPs
    [r*n/world, (r+1)*n/world) instead, so that the ranks' streams concatenate in rank order (bench.py --scaling strong)"""
    n = (n_frames + iper - 1) // iper
    own = (lambda k: r_of(k, n, world) == rank) if contiguous else (lambda k: k % world == rank)
    return [(k * iper, min((k + 1) * iper, n_frames)) for k in range(n) if own(k)]


def r_of(k: int, n: int, world: int) -> int:
    """owner of GOP k when n GOPs are dealt to `world` ranks in contiguous runs whose lengths differ by at most one"""
    base, extra = divmod(n, world)
    edge = extra * (base + 1)
    return k // (base + 1) if k < edge else extra + (k - edge) // max(1, base)


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


def anchor_owner(k: int, world: int) -> int:
    """rank that codes anchor k (k = 0: the key picture): the chain rotates over the ranks, so that no rank carries all of it"""
    return k % world


def b_owner(j: int, world: int, k: int = 0, bframes: int = 0) -> int:
    """rank that codes the j-th B picture of mini-GOP k (between anchors k and k+1).  The rank that is coding anchor k+2 at that moment
    (the serial chain) is left alone; the B pictures rotate over the other ranks, continuing from where mini-GOP k-1 stopped."""
    if world == 1:
        return 0
    busy = anchor_owner(k + 2, world)
    ranks = [r for r in range(world) if r != busy]
    return ranks[(k * bframes + j) % (world - 1)]


def spread_b(rank: int, world: int, n_minigops: int, bframes: int, encode_anchor: Callable, encode_b: Callable, broadcast: Callable) -> list[tuple[int, str]]:
    """Run `n_minigops` mini-GOPs with the anchor chain rotating over the ranks and the B pictures spread over the ranks that are not
    coding an anchor at that moment.

    encode_anchor(d, kind, prev_slot, out_slot)   code picture d ('I'/'P') from the anchor in prev_slot into out_slot   [its owner only]
    encode_b(d, slot0, slot1)                      code B picture d between the anchors held in slot0 (past) / slot1 (future)
    broadcast(slot, src)                           make rank src's picture in `slot` visible in `slot` on every rank (collective)
    Anchors live in three rotating slots: while the B pictures between anchors k and k+1 are coded from two slots, anchor k+2 is already
    written into the third.  The chain itself stays serial (anchor k+1 predicts from anchor k), but every rank carries 1/world of it and
    of the B pictures instead of rank 0 carrying the whole chain.
    Returns the (display index, kind) pictures THIS rank coded, in order.
    """
    mine: list[tuple[int, str]] = []
    if rank == anchor_owner(0, world):
        encode_anchor(0, "I", None, 0)
        mine.append((0, "I"))
    broadcast(0, anchor_owner(0, world))
    for k in range(n_minigops):
        d0, d1 = k * (bframes + 1), (k + 1) * (bframes + 1)
        s0, s1 = k % 3, (k + 1) % 3
        own = anchor_owner(k + 1, world)
        if rank == own:
            encode_anchor(d1, "P", s0, s1)
            mine.append((d1, "P"))
        broadcast(s1, own)
        for j, d in enumerate(range(d0 + 1, d1)):
            if b_owner(j, world, k, bframes) == rank:
                encode_b(d, s0, s1)
                mine.append((d, "B"))
    return mine
