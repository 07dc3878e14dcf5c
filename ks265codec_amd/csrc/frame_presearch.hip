// frame_presearch.hip — Stage A0: motion pre-search on a three-level pyramid (cfg.pre_search).
//
// The reference's lookahead runs a low-resolution motion search on 2:1 pictures (downsample_c enc@0x4a6a60 feeds its frame-cost
// estimate) and meInitPoint enc@0x48af50 starts every integer search from the best of several candidate
// vectors.  A frame-parallel search has no already coded neighbours to take candidates from; what makes the local pattern searches
// of stage A (DIA / HEX / UMH walk downhill from their start point) robust against displaced texture without a gradient comes from
// here instead: an EXHAUSTIVE search where it is cheap.
//   L1 = downsample_c(luma), L2 = downsample_c(L1)                                  (pyr_down_kernel: both levels in one pass)
//   per 8x8 block of L2 (32x32 samples): every vector of +-(range/4 - 1)             (presearch_l2l1_kernel: LDS window, one lane per column,
//   per 8x8 block of L1 (16x16 samples): +-2 around twice the L2 vector               v_sad_u8 running sums; then the L1 children)
//   per 16x16 block of the picture:      +-1 around twice the L1 vector, full-res    (presearch_l0_kernel)
// Cost = SAD + |mx| + |my|, first minimum in raster order of (my, mx) (= minimum of cost << 16 | index).  Low-resolution reads clamp
// to the picture, the full-resolution step reads the padded planes like stage A.  Output: one integer vector per 16x16 block, which
// stage A evaluates as one more start candidate of every PU (frame_me_int.hip).  The CPU checker restates the same definition (tests/).
#include "frame_common.h"
using namespace ks265;

// ------------------------------------------------------------------ pyramid: one thread = 8x4 samples -> 4x2 of L1 -> 2x1 of L2
__device__ __forceinline__ unsigned avg_u8x4(unsigned a, unsigned b)        // per byte (a + b + 1) >> 1
{
    return (a | b) - (((a ^ b) >> 1) & 0x7F7F7F7Fu);
}
__global__ __launch_bounds__(256) void pyr_down_kernel(KsGeom g, const uint8_t *plane, uint8_t *l1, uint8_t *l2, uint8_t *l3)
{
    // one thread = 8x8 samples -> 4x4 of L1 -> 2x2 of L2 -> 1 of L3 (every level = downsample_c of the one above)
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = (blockIdx.x * 16 + tx) * 8, y0 = (blockIdx.y * 16 + ty) * 8;
    if (x0 >= g.W || y0 >= g.H) return;                                     // W, H are multiples of 8: a thread's block is inside or outside as a whole
    const uint8_t *p = ks_org_y(g, plane) + (long)y0 * g.sy + x0;
    const int W1 = g.W >> 1, W2 = g.W >> 2, W3 = g.W >> 3;
    unsigned o1[4];                                                         // four L1 rows of four samples
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint2 a = *(const uint2 *)(p + (long)(2 * r) * g.sy), b = *(const uint2 *)(p + (long)(2 * r + 1) * g.sy);
        const unsigned v0 = avg_u8x4(a.x, b.x), v1 = avg_u8x4(a.y, b.y);
        const unsigned h0 = avg_u8x4(v0, v0 >> 8), h1 = avg_u8x4(v1, v1 >> 8);                      // bytes 0 and 2 hold the pair averages
        o1[r] = (h0 & 0xFFu) | ((h0 >> 8) & 0xFF00u) | ((h1 & 0xFFu) << 16) | ((h1 << 8) & 0xFF000000u);
        *(unsigned *)(l1 + (long)((y0 >> 1) + r) * W1 + (x0 >> 1)) = o1[r];
    }
    unsigned o2[2];                                                         // two L2 rows of two samples (in the low 16 bits)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned vv = avg_u8x4(o1[2 * r], o1[2 * r + 1]), hh = avg_u8x4(vv, vv >> 8);
        o2[r] = (hh & 0xFFu) | ((hh >> 8) & 0xFF00u);
        *(unsigned short *)(l2 + (long)((y0 >> 2) + r) * W2 + (x0 >> 2)) = (unsigned short)o2[r];
    }
    const unsigned v3 = avg_u8x4(o2[0], o2[1]), h3 = avg_u8x4(v3, v3 >> 8);
    l3[(long)(y0 >> 3) * W3 + (x0 >> 3)] = (uint8_t)(h3 & 0xFFu);
}

// SAD of a bw x bh block of packed low-resolution planes, reference reads clamped to the picture (generic: any bw, bh <= 8)
__device__ __forceinline__ unsigned sad_clamped_any(const uint8_t *cur, const uint8_t *ref, int W, int H, int x0, int y0, int bw, int bh, int mx, int my)
{
    unsigned s = 0;
    for (int y = 0; y < bh; ++y) {
        const int ry = min(max(y0 + y + my, 0), H - 1);
        for (int x = 0; x < bw; ++x) {
            const int rx = min(max(x0 + x + mx, 0), W - 1);
            s += (unsigned)abs((int)cur[(long)(y0 + y) * W + x0 + x] - (int)ref[(long)ry * W + rx]);
        }
    }
    return s;
}
// ------------------------------------------------------------------ L3 (1/8 resolution): one 8x8 block = one CTU, every vector of +-range/4.  A CTU VOTES for its
// best vector when that vector lies where the zero-centred window does not reach and matches at least a quarter better than the best vector the window
// covers; a vector with the votes of at least half of the CTUs becomes the picture's WINDOW OFFSET for stage A (x 8, clipped per CTU to the planes' margin).
// One wave per CTU, lane = candidates; then one work-group counts the votes.
__global__ __launch_bounds__(256) void presearch_l3_kernel(KsGeom g, int range, int R3, const uint8_t *c3, const uint8_t *r3, short2 *cand)
{
    // per wave: the CTU's 8x8 block (two dwords per row, masked to the picture) and its (8 + 2 R3) x (8 + 2 R3) window, clamped, in LDS
    extern __shared__ unsigned char lds3[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ctu = blockIdx.x * 4 + wave;
    const int side = 2 * R3 + 1, wd = 8 + 2 * R3, ws = (((wd + 8 + 3) >> 2) | 1) << 2, wbytes = wd * ws + 64;
    uint8_t *win = lds3 + wave * wbytes, *cur = win + wd * ws;
    if (ctu >= g.ctu_cols * g.ctu_rows) return;                            // (no work-group barrier below)
    const int cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols, W3 = g.W >> 3, H3 = g.H >> 3;
    const int bw = min(8, W3 - 8 * cx), bh = min(8, H3 - 8 * cy);
    for (int i = lane; i < wd * (ws >> 2); i += 64) {
        const int y = i / (ws >> 2), xd = (i % (ws >> 2)) * 4;
        const int ry = min(max(8 * cy - R3 + y, 0), H3 - 1);
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (unsigned)r3[(long)ry * W3 + min(max(8 * cx - R3 + xd + b, 0), W3 - 1)] << (8 * b);
        *(unsigned *)(win + y * ws + xd) = v;
    }
    { const int y = lane >> 3, x = lane & 7; cur[lane] = (y < bh && x < bw) ? c3[(long)(8 * cy + y) * W3 + 8 * cx + x] : 0; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long mask = bw >= 8 ? ~0ull : ((1ull << (8 * bw)) - 1ull);
    const unsigned m0 = (unsigned)mask, m1 = (unsigned)(mask >> 32);
    unsigned f0[8], f1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { f0[r] = *(const unsigned *)(cur + 8 * r); f1[r] = *(const unsigned *)(cur + 8 * r + 4); }
    unsigned loc = 0xFFFFFFFFu, near = 0xFFFFFFFFu;
    for (int c = lane; c < side * side; c += 64) {
        const int my = c / side - R3, mx = c % side - R3, wx = mx + R3;
        const uint8_t *p = win + (my + R3) * ws + (wx & ~3);
        const unsigned sh = wx & 3;
        unsigned sad = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < bh) {
                const unsigned w0 = *(const unsigned *)(p + r * ws), w1 = *(const unsigned *)(p + r * ws + 4), w2 = *(const unsigned *)(p + r * ws + 8);
                sad = sad_u8x4(f0[r], align_bytes(w1, w0, sh) & m0, sad);
                sad = sad_u8x4(f1[r], align_bytes(w2, w1, sh) & m1, sad);
            }
        const unsigned cost = sad + (unsigned)(abs(mx) + abs(my));
        loc = min(loc, (cost << 16) | (unsigned)c);
        if (abs(8 * mx) <= range / 2 && abs(8 * my) <= range / 2) near = min(near, cost);       // what the zero-centred window covers with margin
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { loc = min(loc, (unsigned)__shfl_xor((int)loc, m, 64)); near = min(near, (unsigned)__shfl_xor((int)near, m, 64)); }
    if (lane == 0) {
        const int c = (int)(loc & 0xFFFFu);
        int mx = c % side - R3, my = c / side - R3;
        if ((abs(8 * mx) <= range / 2 && abs(8 * my) <= range / 2) || 4ull * (loc >> 16) >= 3ull * near) { mx = 0; my = 0; }
        cand[ctu] = make_short2((short)mx, (short)my);
    }
}
__global__ __launch_bounds__(256) void presearch_vote_kernel(KsGeom g, int R3, const short2 *cand, short2 *ctu_off)
{
    extern __shared__ unsigned votes[];                                     // side x side counters, then the winner
    const int side = 2 * R3 + 1, n = side * side, nctu = g.ctu_cols * g.ctu_rows, t = threadIdx.x;
    __shared__ unsigned long long top;
    for (int i = t; i < n; i += 256) votes[i] = 0;
    if (t == 0) top = 0;
    __syncthreads();
    for (int i = t; i < nctu; i += 256) { const short2 c = cand[i]; if (c.x | c.y) atomicAdd(&votes[(c.y + R3) * side + c.x + R3], 1u); }
    __syncthreads();
    for (int i = t; i < n; i += 256) if (votes[i]) atomicMax(&top, ((unsigned long long)votes[i] << 32) | (unsigned)(n - 1 - i));   // most votes; ties: the first in raster order
    __syncthreads();
    int gmx = 0, gmy = 0;
    if (2 * (int)(top >> 32) >= nctu && (top >> 32)) { const int i = n - 1 - (int)(top & 0xFFFFFFFFull); gmx = i % side - R3; gmy = i / side - R3; }
    for (int i = t; i < nctu; i += 256) {
        const int cx = i % g.ctu_cols, cy = i / g.ctu_cols, xe = min(cx * 64 + 64, g.W), ye = min(cy * 64 + 64, g.H);
        ctu_off[i] = make_short2((short)(clip3(-64 - cx * 64, g.W + 64 - xe, 8 * gmx) & ~15), (short)(clip3(-64 - cy * 64, g.H + 64 - ye, 8 * gmy) & ~15));
    }
}

// one column (fixed mx) of the exhaustive L2 search: walk down the window rows, eight running sums in rotating registers.
// FULL: an 8x8 block completely inside the picture (no masks, the completed slot is known at compile time).
template <bool FULL>
__device__ __forceinline__ unsigned l2_column(const uint8_t *p, int ws, unsigned sh, const unsigned *f0, const unsigned *f1, int nrow, int R, int side, int mx, bool active,
                                              int bh, unsigned m0, unsigned m1, int cx2, int cy2 /* centre of the search: the vector is (cx2 + mx, cy2 + my) */)
{
    unsigned loc = 0xFFFFFFFFu;
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned amx = (unsigned)abs(cx2 + mx);
    for (int j0 = 0; j0 < nrow; j0 += 8) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = j0 + jj;
            const unsigned w0 = *(const unsigned *)(p + j * ws), w1 = *(const unsigned *)(p + j * ws + 4), w2 = *(const unsigned *)(p + j * ws + 8);
            unsigned a = align_bytes(w1, w0, sh), b = align_bytes(w2, w1, sh);
            if (!FULL) { a &= m0; b &= m1; }
#pragma unroll
            for (int r = 0; r < 8; ++r) {                                   // window row j = block row r of the displacement my = j - r - R: running sum (j - r) & 7
                if (FULL || r < bh) { acc[(jj - r) & 7] = sad_u8x4(f0[r], a, acc[(jj - r) & 7]); acc[(jj - r) & 7] = sad_u8x4(f1[r], b, acc[(jj - r) & 7]); }
            }
            // the displacement whose LAST block row this was is complete: my = j - (bh - 1) - R, its sum sits in slot (jj - (bh - 1)) & 7
            const int last = FULL ? 7 : bh - 1, my = j - last - R;
            unsigned sad;
            if (FULL) sad = acc[(jj + 1) & 7];
            else {
                const int slot = (jj - last) & 7;
                sad = acc[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) sad = slot == q ? acc[q] : sad;
            }
            if (my >= -R && my <= R && active) loc = min(loc, ((sad + amx + (unsigned)abs(cy2 + my)) << 16) | (unsigned)((my + R) * side + mx + R));
            if (FULL) acc[(jj + 1) & 7] = 0;                               // the slot starts over with window row j + 1
            else {
                const int slot = (jj - last) & 7;
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = slot == q ? 0u : acc[q];
            }
        }
    }
    return loc;
}

// Work-group = the 2 x 2 blocks of 8x8 L2 samples of ONE CTU, searched +-R around the CTU's window offset / 4.  LB lanes per block (LB = 32 when
// 2R + 1 <= 32: two blocks per wave; else 64), lane = one horizontal displacement; 4 LB threads.  Then every block's four L1 children (+-2 around twice
// its vector) by the same lanes.  dynamic LDS: window (8 + nrow) rows x ws bytes, then the 16 x 32 source tile.
template <int LB>
__global__ __launch_bounds__(4 * LB) void presearch_l2l1_kernel(int W2, int H2, int R, const uint8_t *c2, const uint8_t *r2, int nbx, int nby,
                                                                int W1, int H1, const uint8_t *c1, const uint8_t *r1, short2 *mv1, int nb1x, int nb1y, const short2 *ctu_off)
{
    extern __shared__ unsigned char lds[];
    constexpr int BX = 2, NT = 4 * LB;
    const int side = 2 * R + 1, nrow = (8 + 2 * R + 7) & ~7, wd = BX * 8 + 2 * R, wrows = 8 + nrow;
    const int ws = (((wd + 8 + 3) >> 2) | 1) << 2;
    constexpr int L1_BYTES = 256 + 20 * 24;
    uint8_t *win = lds, *cur = lds + wrows * ws;                            // cur: 16 rows x 32 bytes
    __shared__ unsigned best1[4][4];
    __shared__ __attribute__((aligned(16))) uint8_t l1buf[4 * L1_BYTES];
    const int bx0 = blockIdx.x * BX, by0 = blockIdx.y * 2, x0 = bx0 * 8, y0 = by0 * 8, t = threadIdx.x;
    const short2 off = ctu_off[blockIdx.y * gridDim.x + blockIdx.x];
    const int cx2 = off.x / 4, cy2 = off.y / 4;                             // the offsets are multiples of 16
    for (int i = t; i < wrows * (ws >> 2); i += NT) {
        const int y = i / (ws >> 2), xd = (i % (ws >> 2)) * 4;
        const int ry = min(max(y0 + cy2 - R + y, 0), H2 - 1);
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (unsigned)r2[(long)ry * W2 + min(max(x0 + cx2 - R + xd + b, 0), W2 - 1)] << (8 * b);
        *(unsigned *)(win + y * ws + xd) = v;
    }
    for (int i = t; i < 16 * 32; i += NT) {
        const int y = i >> 5, x = i & 31;
        cur[i] = (y0 + y < H2 && x0 + x < W2 && x < BX * 8) ? c2[(long)(y0 + y) * W2 + x0 + x] : 0;
    }
    if (t < 16) best1[t >> 2][t & 3] = 0xFFFFFFFFu;
    __syncthreads();
    const int blk = t / LB, col = t & (LB - 1);                             // block inside the work-group: x = blk % BX, y = blk / BX
    const int bxl = blk % BX, byl = blk / BX, bx = bx0 + bxl, by = by0 + byl;
    const bool bvalid = bx < nbx && by < nby;
    const int bw = bvalid ? min(8, W2 - 8 * bx) : 8, bh = bvalid ? min(8, H2 - 8 * by) : 8;        // even
    const unsigned m0 = bw >= 4 ? 0xFFFFFFFFu : 0x0000FFFFu, m1 = bw == 8 ? 0xFFFFFFFFu : bw == 6 ? 0x0000FFFFu : 0u;
    unsigned f0[8], f1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint8_t *f = cur + (byl * 8 + r) * 32 + bxl * 8;
        f0[r] = *(const unsigned *)f & m0; f1[r] = *(const unsigned *)(f + 4) & m1;
    }
    const int mx = min(col, side - 1) - R, wx = bxl * 8 + mx + R;          // idle lanes repeat the last column (their result is dropped)
    const uint8_t *p = win + (byl * 8) * ws + (wx & ~3);
    const bool active = bvalid && col < side;
    unsigned loc;
    if (bw == 8 && bh == 8) loc = l2_column<true>(p, ws, wx & 3, f0, f1, nrow, R, side, mx, active, 8, m0, m1, cx2, cy2);
    else loc = l2_column<false>(p, ws, wx & 3, f0, f1, nrow, R, side, mx, active, bh, m0, m1, cx2, cy2);
#pragma unroll
    for (int m = 1; m < LB; m <<= 1) loc = min(loc, (unsigned)__shfl_xor((int)loc, m, 64));     // minimum over the block's lanes
    const int c = (int)(loc & 0xFFFFu), vx = cx2 + c % side - R, vy = cy2 + c / side - R;
    // ---- L1: the four 8x8 children, +-2 around (2 vx, 2 vy).  The block's lanes stage its 16x16 L1 samples and the 20x20 reference window at the
    //      displacement (clamped to the picture like every low-resolution read) in LDS, then item = child (2 bits) x candidate (25)
    uint8_t *c1l = l1buf + blk * L1_BYTES, *r1l = c1l + 256;               // 16 x 16, then 20 rows x 24 bytes
    if (bvalid) {
        const int ox = 16 * bx, oy = 16 * by, rx0 = ox + 2 * vx - 2, ry0 = oy + 2 * vy - 2;
        for (int i = col; i < 64; i += LB) {                                // source: 16 rows x 4 dwords (aligned: W1 is a multiple of 4)
            const int y = i >> 2, x = (i & 3) * 4;
            *(unsigned *)(c1l + y * 16 + x) = (oy + y < H1 && ox + x < W1) ? *(const unsigned *)(c1 + (long)(oy + y) * W1 + ox + x) : 0u;
        }
        const bool inside = rx0 >= 0 && ry0 >= 0 && rx0 + 20 <= W1 && ry0 + 20 <= H1;
        for (int i = col; i < 100; i += LB) {                               // window: 20 rows x 5 dwords
            const int y = i / 5, x = (i - y * 5) * 4;
            unsigned v;
            if (inside) __builtin_memcpy(&v, r1 + (long)(ry0 + y) * W1 + rx0 + x, 4);
            else {
                const int ry = min(max(ry0 + y, 0), H1 - 1);
                v = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) v |= (unsigned)r1[(long)ry * W1 + min(max(rx0 + x + q, 0), W1 - 1)] << (8 * q);
            }
            *(unsigned *)(r1l + y * 24 + x) = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (bvalid)
        for (int it = col; it < 100; it += LB) {
            const int ch = it / 25, k = it - ch * 25;
            const int b1x = 2 * bx + (ch & 1), b1y = 2 * by + (ch >> 1);
            if (b1x >= nb1x || b1y >= nb1y) continue;
            const int dx = k % 5, dy = k / 5, cmx = 2 * vx + dx - 2, cmy = 2 * vy + dy - 2;
            const int cbw = min(8, W1 - 8 * b1x), cbh = min(8, H1 - 8 * b1y);                  // 4 or 8
            const unsigned cm1 = cbw == 8 ? 0xFFFFFFFFu : 0u;
            const uint8_t *pc = c1l + (ch >> 1) * 8 * 16 + (ch & 1) * 8, *pr = r1l + ((ch >> 1) * 8 + dy) * 24 + (((ch & 1) * 8 + dx) & ~3);
            const unsigned rsh = ((ch & 1) * 8 + dx) & 3;
            unsigned sad = 0;
            for (int r = 0; r < cbh; ++r) {
                const unsigned w0 = *(const unsigned *)(pr + r * 24), w1 = *(const unsigned *)(pr + r * 24 + 4), w2 = *(const unsigned *)(pr + r * 24 + 8);
                sad = sad_u8x4(*(const unsigned *)(pc + r * 16), align_bytes(w1, w0, rsh), sad);
                sad = sad_u8x4(*(const unsigned *)(pc + r * 16 + 4) & cm1, align_bytes(w2, w1, rsh) & cm1, sad);
            }
            atomicMin(&best1[blk][ch], ((sad + (unsigned)(abs(cmx) + abs(cmy))) << 8) | (unsigned)k);
        }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (bvalid && col < 4) {
        const int b1x = 2 * bx + (col & 1), b1y = 2 * by + (col >> 1);
        if (b1x < nb1x && b1y < nb1y) {
            const int k = (int)(best1[blk][col] & 255u);
            mv1[b1y * nb1x + b1x] = make_short2((short)(2 * vx + (k % 5 - 2)), (short)(2 * vy + (k / 5 - 2)));
        }
    }
}

// ------------------------------------------------------------------ L0: +-1 around twice the L1 vector, full resolution, padded planes; 16 lanes per block (9 candidates)
__global__ __launch_bounds__(256) void presearch_l0_kernel(KsGeom g, int range, const uint8_t *src, const uint8_t *ref, const short2 *mv1, short2 *field, int nbx, int nby, const short2 *ctu_off)
{
    const int blk = blockIdx.x * 16 + (threadIdx.x >> 4), k = threadIdx.x & 15;
    if (blk >= nbx * nby) return;
    const int bx = blk % nbx, by = blk / nbx;
    const short2 p = mv1[blk];
    unsigned key = 0xFFFFFFFFu;
    int mx = 0, my = 0;
    if (k < 9) {
        const short2 off = ctu_off[(by >> 2) * g.ctu_cols + (bx >> 2)];
        int lox, hix, loy, hiy;
        ctu_mv_limits(g, range, bx >> 2, by >> 2, off.x, off.y, lox, hix, loy, hiy);
        mx = clip3(lox, hix, 2 * p.x + (k % 3 - 1)); my = clip3(loy, hiy, 2 * p.y + (k / 3 - 1));
        const int bw = min(16, g.W - 16 * bx), bh = min(16, g.H - 16 * by);                   // 8 or 16
        const uint8_t *c = ks_org_y(g, src) + (long)(16 * by) * g.sy + 16 * bx, *r = ks_org_y(g, ref) + (long)(16 * by + my) * g.sy + 16 * bx + mx;
        unsigned sad = 0;
        for (int y = 0; y < bh; ++y) {
            for (int x = 0; x < bw; x += 4) {
                unsigned a, b;
                __builtin_memcpy(&a, c + (long)y * g.sy + x, 4); __builtin_memcpy(&b, r + (long)y * g.sy + x, 4);
                sad = sad_u8x4(a, b, sad);
            }
        }
        key = ((sad + (unsigned)(abs(mx) + abs(my))) << 8) | (unsigned)k;
    }
    unsigned b = key;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) b = min(b, (unsigned)__shfl_xor((int)b, m, 64));
    if (k == (int)(b & 15u) && key == b) field[blk] = make_short2((short)mx, (short)my);
}

static int presearch_alloc(ks265_frame *f)
{
    if (f->pyr[0]) return KS265_OK;
    const int W = f->g.W, H = f->g.H;
    const size_t n1 = (size_t)(W / 2) * (H / 2), n2 = (size_t)(W / 4) * (H / 4), n3 = (size_t)(W / 8) * (H / 8);
    const size_t nb2 = (size_t)((W / 4 + 7) / 8) * ((H / 4 + 7) / 8), nb1 = (size_t)((W / 2 + 7) / 8) * ((H / 2 + 7) / 8), nb0 = (size_t)((W + 15) / 16) * ((H + 15) / 16);
    const size_t nctu4 = (size_t)f->g.ctu_cols * f->g.ctu_rows * 4;
    const size_t sz[10] = {n1 + 16, n2 + 16, n1 + 16, n2 + 16, (nb2 * 4 > nctu4 ? nb2 * 4 : nctu4) + 16 /* the CTUs' votes */, nb1 * 4, nb0 * 4, n3 + 16, n3 + 16, nctu4 + 16};
    for (int i = 0; i < 10; ++i) {
        void *p = nullptr;
        const int r = ks265_hip(f->ctx, hipMalloc(&p, sz[i]));
        if (r) return r;
        f->pyr[i] = (uint8_t *)p;
    }
    if (f->pu1)                                                      /* B pictures: list 1's own reference pyramid, vector fields and window offsets (the source pyramid is shared) */
        for (int i = 0; i < 10; ++i) {
            if (i == 0 || i == 1 || i == 7) { f->pyr2[i] = f->pyr[i]; continue; }
            void *p = nullptr;
            const int r = ks265_hip(f->ctx, hipMalloc(&p, sz[i]));
            if (r) return r;
            f->pyr2[i] = (uint8_t *)p;
        }
    return KS265_OK;
}

/* the source picture's pyramid alone (a B picture builds it once for its two searches) */
int ks265_presearch_source(ks265_frame *f, ks265_pic src)
{
    KS_FRAME_CHECK(f);
    int r = presearch_alloc(f);
    if (r) return r;
    const KsGeom &g = f->g;
    const dim3 gd((unsigned)((g.W + 127) / 128), (unsigned)((g.H + 127) / 128));
    hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, f->ctx->stream, g, src.y, f->pyr[0], f->pyr[1], f->pyr[7]);
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_presearch(ks265_frame *f, ks265_pic src, ks265_pic ref, int16_t *dev_field, int16_t *dev_ctu_off)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y) return KS265_POINTER;
    (void)hipSetDevice(f->ctx->device);
    int r = presearch_alloc(f);
    if (r) return r;
    const KsGeom &g = f->g;
    const int W1 = g.W / 2, H1 = g.H / 2, W2 = g.W / 4, H2 = g.H / 4;
    const int nb2x = (W2 + 7) / 8, nb2y = (H2 + 7) / 8, nb1x = (W1 + 7) / 8, nb1y = (H1 + 7) / 8, nb0x = (g.W + 15) / 16, nb0y = (g.H + 15) / 16;
    uint8_t *c1 = f->pyr[0], *c2 = f->pyr[1], *r1 = f->pyr[2], *r2 = f->pyr[3], *c3 = f->pyr[7], *r3 = f->pyr[8];
    short2 *mv1 = (short2 *)f->pyr[5], *field = (short2 *)dev_field;      // dev_field = NULL: no stand-alone full-resolution step (ks265_me_integer does it itself)
    short2 *ctu_off = dev_ctu_off ? (short2 *)dev_ctu_off : (short2 *)f->pyr[9];
    hipStream_t st = f->ctx->stream;
    const dim3 gd((unsigned)((g.W + 127) / 128), (unsigned)((g.H + 127) / 128));
    if (!f->src_pyr_ready) hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, st, g, src.y, c1, c2, c3);
    hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, st, g, ref.y, r1, r2, r3);
    const int R = max((f->cfg.me_range >> 2) - 1, 1), R3 = max(f->cfg.me_range >> 2, 1), wrows = 8 + ((8 + 2 * R + 7) & ~7);
    if (2 * R + 1 > 64 || (2 * R3 + 1) * (2 * R3 + 1) > 65535) return KS265_NOTSUPPORTED;   // one lane per horizontal displacement (me_range <= 128)
    const int nctu = g.ctu_cols * g.ctu_rows;
    const int wd3 = 8 + 2 * R3, ws3 = (((wd3 + 8 + 3) >> 2) | 1) << 2;
    hipLaunchKernelGGL(presearch_l3_kernel, dim3((unsigned)((nctu + 3) / 4)), dim3(256), (size_t)4 * (wd3 * ws3 + 64), st, g, f->cfg.me_range, R3, c3, r3, (short2 *)f->pyr[4]);
    hipLaunchKernelGGL(presearch_vote_kernel, dim3(1), dim3(256), (size_t)(2 * R3 + 1) * (2 * R3 + 1) * 4, st, g, R3, (const short2 *)f->pyr[4], ctu_off);
    const int wd = 16 + 2 * R, ws = (((wd + 8 + 3) >> 2) | 1) << 2;
    const dim3 gc((unsigned)g.ctu_cols, (unsigned)g.ctu_rows);
    if (2 * R + 1 <= 32)
        hipLaunchKernelGGL(presearch_l2l1_kernel<32>, gc, dim3(128), (size_t)(wrows * ws + 512), st, W2, H2, R, c2, r2, nb2x, nb2y, W1, H1, c1, r1, mv1, nb1x, nb1y, ctu_off);
    else
        hipLaunchKernelGGL(presearch_l2l1_kernel<64>, gc, dim3(256), (size_t)(wrows * ws + 512), st, W2, H2, R, c2, r2, nb2x, nb2y, W1, H1, c1, r1, mv1, nb1x, nb1y, ctu_off);
    // the full-resolution step: stage A does it itself from its LDS window (frame_me_int.hip); the stand-alone kernel serves the stage API
    if (field) hipLaunchKernelGGL(presearch_l0_kernel, dim3((unsigned)((nb0x * nb0y + 15) / 16)), dim3(256), 0, st, g, f->cfg.me_range, src.y, ref.y, mv1, field, nb0x, nb0y, ctu_off);
    return ks265_check_launch(f->ctx);
}
