"""The host's cuTree pass (ks265codec_amd/host/ks265_enc.c ct_*) restated on the CPU with the oracle's pinned pieces - calcFrameCost enc@0x4a7410, cuTreePropagate enc@0x47d460,
the finish enc@0x480964, calcFrameAdaptQuant enc@0x4653c0, downsample_c enc@0x4a6a60 (oracle/ks265_lookahead_ref.c, ks265_intra_oracle.c) - so that the QP per CTU the encoder
codes a picture with can be computed independently of it.  TEST INFRASTRUCTURE (tests only)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from cfc_cases import ARR, CFG_WORDS, PAD, oracle_run
from oracle_lib import lib, ptr

# the words calcFrameCost reads, by preset (host: kCtPreset): zero_thr, fast_intra, f36c, f538, f3b4, 16x16 blocks whatever the size
PRESET = {0: (4, 4, 0, 120, 2, 1), 1: (4, 4, 0, 120, 2, 1), 2: (4, 1, 0, 0, 1, 1), 3: (0, 0, 1, 0, 1, 0), 4: (0, 0, 1, 0, 1, 0), 5: (0, 0, 1, 0, 1, 0), 6: (0, 0, 1, 0, 1, 0), 7: (0, 0, 1, 0, 1, 0), 8: (0, 0, 0, 0, 1, 0)}
LAMBDA = np.array([1] * 19 + [2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 7, 8, 9, 11, 12, 13, 15, 17, 19, 21, 24, 27, 30, 34, 38, 42, 47, 53, 60, 67], np.uint16)


class CuTree:
    def __init__(self, clip, W, H, preset=5, gop_b=3, hier=True, iper=128, lookahead=-1, aq_strength=0.0, scenecut=30, keys=()):
        self.o = lib(); self.o.kso_mvd_bits.restype = C.c_int
        self.clip, self.W, self.H, self.n = clip, W, H, len(clip)
        L = lookahead if lookahead > 0 else (12 if preset <= 2 else 16 if preset <= 4 else 64)      # the host's rule (ks265_enc.c lane_open: measured inside the reference)
        span = gop_b + 1
        self.gop_b, self.hier, self.iper, self.depth, self.aq = gop_b, hier, iper, max(((L - 1) // span) * span, span), aq_strength
        zt, fi, f36c, f538, f3b4, lg4 = PRESET[preset]
        self.lg = 4 if (lg4 or W * H >= 1280 * 720) else 3
        bs = 1 << self.lg
        self.w, self.h = W // 2, H // 2
        self.nx, self.ny = (self.w + bs - 1) // bs, (self.h + bs - 1) // bs
        self.cfgw = dict(zip(CFG_WORDS, (64, self.lg, zt, fi, scenecut, preset, 0, int(aq_strength > 0), 1, 13, f36c, f538, f3b4)))
        self.keys = set(keys)                                          # pictures the host made key pictures on its own (scene cuts, requests): they close the GOP in front of them
        self.pic = {}
        self.maps_qoff = {}

    # ---- per picture state
    def _get(self, d):
        if d in self.pic:
            return self.pic[d]
        o, W, H, w, h, n = self.o, self.W, self.H, self.w, self.h, self.nx * self.ny
        fr = np.ascontiguousarray(self.clip[d])
        low = np.zeros((h, w), np.uint8)
        o.ks265o_downsample(ptr(low), ptr(fr), w, W, w, h)
        pad = PAD + (1 << self.lg)
        st = dict(low=np.ascontiguousarray(np.pad(low, pad, mode="edge")), pad=pad, p0=None, p1=None, intra_done=0,
                  arr=dict(intra=np.zeros(n, np.uint16), imode=np.zeros(n, np.uint8), invq=np.full(n, 256, np.uint16), inter=np.zeros(n, np.uint16), bits=np.zeros((n + 3) // 4, np.uint8),
                           mv0=np.zeros(n, np.int32), c0=np.zeros(n, np.int32), mv1=np.zeros(n, np.int32), c1=np.zeros(n, np.int32)),
                  prop=np.zeros(n, np.uint16), aq_off=np.zeros(n, np.float64), qoff=np.zeros(n, np.float64))
        if self.aq > 0:                                                        # calcFrameAdaptQuant on the lookahead's grid, 16 x 16 blocks of the full-size picture (as the reference calls it)
            Y = fr[:W * H].reshape(H, W); U = fr[W * H:W * H * 5 // 4].reshape(H // 2, W // 2); V = fr[W * H * 5 // 4:].reshape(H // 2, W // 2)
            nx, ny = self.nx, self.ny
            Yp = np.ascontiguousarray(Y[:ny * 16, :nx * 16]); Up = np.ascontiguousarray(U[:ny * 8, :nx * 8]); Vp = np.ascontiguousarray(V[:ny * 8, :nx * 8])
            assert Yp.shape == (ny * 16, nx * 16), "the mirror handles pictures whose AQ grid lies inside them"
            o.kso_ref_frame_adapt_quant(ptr(Yp), ptr(Up), ptr(Vp), nx, ny, n, C.c_double(self.aq), ptr(st["aq_off"]), ptr(st["arr"]["invq"]))
        self.pic[d] = st
        return st

    def _plane(self, st):
        """the PAD-margin view oracle_run expects, cut out of the wider padded plane"""
        e = st["pad"] - PAD
        a = st["low"]
        return np.ascontiguousarray(a[e:a.shape[0] - e, e:a.shape[1] - e]) if e else a

    def _cost(self, b, p0, p1):
        c = self._get(b)
        if p0 is None:
            if c["intra_done"]:
                return
        elif c["p0"] == p0 and c["p1"] == p1:
            return
        d0 = 0 if p0 is None else b - p0
        d1 = 0 if (p0 is None or p1 is None) else p1 - b
        st = 2 if p0 is None else (0 if d1 else 1)
        oracle_run(self.o, self.w, self.h, self.nx, self.ny, self.cfgw, LAMBDA, self._plane(c), self._plane(self._get(p0)) if d0 else None, self._plane(self._get(p1)) if d1 else None,
                   d0, d1, 0, st, (int(d0 > 0), int(d1 > 0)), c["intra_done"], c["arr"], [-1, -1, -1, -1, -1], [-1, -1, -1, -1])
        if p0 is not None:
            c["p0"], c["p1"] = p0, p1
        if d1 == 0:
            c["intra_done"] = 1

    def _structure(self, lo, hi, end):
        out, span = [], self.gop_b + 1
        while lo < end:
            hi = min(hi, end)
            out.append((hi, lo, None, 1))
            if hi - lo > 1:
                if self.hier and ((hi - lo) & (hi - lo - 1)) == 0:
                    cur = [(lo, hi)]
                    while cur:
                        nxt = []
                        for a, b in cur:
                            if b - a < 2:
                                continue
                            mid = (a + b) // 2
                            out.append((mid, a, b, int(mid - a >= 2 or b - mid >= 2)))
                            nxt += [(a, mid), (mid, b)]
                        cur = nxt
                else:
                    out += [(b, lo, hi, 0) for b in range(lo + 1, hi)]
            lo, hi = hi, hi + span
        return out

    def _run(self, key, d, a, end):
        nodes = self._structure(key, key + self.gop_b + 1, end) if key is not None else self._structure(d, a, end)
        first = key if key is not None else d
        for k in range(first, end + 1):
            self._get(k)["prop"][:] = 0
        if key is not None:
            self._cost(key, None, None)
        o = self.o
        for b, p0, p1, _ in reversed(nodes):
            self._cost(b, p0, p1)
            c, r0 = self._get(b), self._get(p0)
            r1 = self._get(p1) if p1 is not None else r0
            A = c["arr"]
            o.kso_ref_cutree_propagate(self.lg, self.nx, self.ny, ptr(A["intra"]), ptr(A["invq"]), ptr(c["prop"]), ptr(A["inter"]), ptr(A["bits"]), ptr(A["mv0"]),
                                       ptr(A["mv1"] if p1 is not None else A["mv0"]), ptr(r0["prop"]), ptr(r1["prop"]))
        fin = [(key, 1, 0)] if key is not None else [(b, ref, int(p1 is None and end == a)) for b, _, p1, ref in nodes if b <= a]
        for b, is_ref, dbl in fin:                                             # dbl: nothing of the window lies behind this mini-GOP - its anchor's propagated cost counts twice (enc@0x480c6a, 0x4809d2)
            c = self._get(b)
            q = c["aq_off"].copy()
            if is_ref:
                o.kso_ref_cutree_finish(self.nx * self.ny, ptr(c["arr"]["intra"]), ptr(c["arr"]["invq"]), ptr(c["prop"]), ptr(c["aq_off"]), dbl, ptr(q))
            self.maps_qoff[b] = q

    def _window_end(self, a, gop_start):
        end = a + self.depth
        if self.iper > 0:
            end = min(end, gop_start + self.iper - 1)
        nk = [k for k in self.keys if k > a]
        if nk:
            end = min(end, min(nk) - 1)
        end = min(end, self.n - 1)
        return max(end, a)

    def run(self):
        """the scheduler's walk over the whole clip (everything has arrived, then the flush): the block offsets of every picture"""
        coded, gop_start = -1, 0
        while coded + 1 < self.n:
            nxt = coded + 1
            if coded < 0 or (self.iper > 0 and nxt - gop_start >= self.iper) or nxt in self.keys:
                gop_start = nxt
                self._run(nxt, nxt, nxt, self._window_end(nxt, gop_start))
                coded = nxt
                continue
            a = coded + self.gop_b + 1
            if self.iper > 0 and a - gop_start >= self.iper:
                a = gop_start + self.iper - 1
            nk = [k for k in self.keys if k > nxt]
            if nk:
                a = min(a, min(nk) - 1)
            a = min(a, self.n - 1)
            self._run(None, coded, a, self._window_end(a, gop_start))
            coded = a
        return self.maps_qoff

    def ctu_map(self, disp, qp, lo=0, hi=51):
        cols, rows = (self.W + 63) // 64, (self.H + 63) // 64
        m = np.zeros(cols * rows, np.int8)
        self.o.kso_qoff_ctu_map(ptr(self.maps_qoff[disp]), self.nx, self.ny, self.lg, cols, rows, int(qp), lo, hi, ptr(m))
        return m


def read_qpmap_dump(path):
    """KS265_DUMP_QPMAP records: {display index: (kind, qp, map)}"""
    data = open(path, "rb").read()
    out, p = {}, 0
    while p < len(data):
        disp, kind, qp, n = np.frombuffer(data, np.int32, 4, p); p += 16
        out[int(disp)] = (chr(int(kind)), int(qp), np.frombuffer(data, np.int8, int(n), p).copy()); p += int(n)
    return out
