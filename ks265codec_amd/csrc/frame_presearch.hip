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
__global__ __launch_bounds__(256) void pyr_down_kernel(KsGeom g, const uint8_t *plane, uint8_t *l1, uint8_t *l2)
{
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = (blockIdx.x * 16 + tx) * 8, y0 = (blockIdx.y * 16 + ty) * 4;
    if (x0 >= g.W || y0 >= g.H) return;                                     // W, H are multiples of 8: a thread's block is inside or outside as a whole
    const uint8_t *p = ks_org_y(g, plane) + (long)y0 * g.sy + x0;
    unsigned v[2][2];                                                       // vertical averages of row pairs, 8 samples each
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint2 a = *(const uint2 *)(p + (long)(2 * r) * g.sy), b = *(const uint2 *)(p + (long)(2 * r + 1) * g.sy);
        v[r][0] = avg_u8x4(a.x, b.x); v[r][1] = avg_u8x4(a.y, b.y);
    }
    // horizontal: (even + odd + 1) >> 1 of neighbouring bytes
    unsigned o1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned h0 = avg_u8x4(v[r][0], v[r][0] >> 8), h1 = avg_u8x4(v[r][1], v[r][1] >> 8);       // bytes 0 and 2 hold the pair averages
        o1[r] = (h0 & 0xFFu) | ((h0 >> 8) & 0xFF00u) | ((h1 & 0xFFu) << 16) | ((h1 << 8) & 0xFF000000u);
    }
    const int W1 = g.W >> 1, W2 = g.W >> 2;
    *(unsigned *)(l1 + (long)(y0 >> 1) * W1 + (x0 >> 1)) = o1[0];
    *(unsigned *)(l1 + (long)((y0 >> 1) + 1) * W1 + (x0 >> 1)) = o1[1];
    const unsigned vv = avg_u8x4(o1[0], o1[1]), hh = avg_u8x4(vv, vv >> 8);
    *(unsigned short *)(l2 + (long)(y0 >> 2) * W2 + (x0 >> 2)) = (unsigned short)((hh & 0xFFu) | ((hh >> 8) & 0xFF00u));
}

// ------------------------------------------------------------------ L2 exhaustive search + L1 refinement in one kernel
// Work-group = 2x2 blocks of 8x8 L2 samples, ONE WAVE PER BLOCK.  Lane = one horizontal displacement mx (2R + 1 <= 64 lanes busy); the lane walks down
// the window rows once: row j of the window is row r of the block for the vertical displacement my = j - r - R, so each row read (three LDS dwords)
// feeds eight running sums (my = j - R - 7 .. j - R), held in eight registers that rotate with the row index (the row loop is unrolled by eight so the
// rotation is static).  The block's own samples are the same for every lane: sixteen scalar registers.  A vertical displacement is complete - and
// compared - when its eighth row has been added.  8x less LDS traffic than a SAD per candidate, and no barrier inside the search.
// Then the wave refines its four L1 children (+-2 around twice its vector): 100 (block, candidate) pairs over the 64 lanes, samples straight from the
// (cached) L1 planes.  dynamic LDS: window (16 + 2R rounded up to a multiple of 8, + 8) rows x ws bytes, then the 16x16 source tile.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, (unsigned)__shfl_xor((int)v, m, 64));
    return v;
}
// SAD of a bw x bh block (bw, bh <= 8, even) of packed low-resolution planes, reference reads clamped to the picture
__device__ __forceinline__ unsigned sad_clamped(const uint8_t *cur, const uint8_t *ref, int W, int H, int x0, int y0, int bw, int bh, int mx, int my)
{
    unsigned s = 0;
    if (x0 + mx >= 0 && x0 + mx + 8 <= W && y0 + my >= 0 && y0 + my + bh <= H && bw == 8) {       // whole block inside: dword reads
        const uint8_t *c = cur + (long)y0 * W + x0, *r = ref + (long)(y0 + my) * W + x0 + mx;
        for (int y = 0; y < bh; ++y) {
            unsigned a0, a1, b0, b1;
            __builtin_memcpy(&a0, c + (long)y * W, 4); __builtin_memcpy(&a1, c + (long)y * W + 4, 4);
            __builtin_memcpy(&b0, r + (long)y * W, 4); __builtin_memcpy(&b1, r + (long)y * W + 4, 4);
            s = sad_u8x4(a0, b0, s); s = sad_u8x4(a1, b1, s);
        }
        return s;
    }
    for (int y = 0; y < bh; ++y) {
        const int ry = min(max(y0 + y + my, 0), H - 1);
        for (int x = 0; x < bw; ++x) {
            const int rx = min(max(x0 + x + mx, 0), W - 1);
            s += (unsigned)abs((int)cur[(long)(y0 + y) * W + x0 + x] - (int)ref[(long)ry * W + rx]);
        }
    }
    return s;
}
// one column (fixed mx) of the exhaustive L2 search: walk down the window rows, eight running sums in rotating registers.
// FULL: an 8x8 block completely inside the picture (no masks, the completed slot is known at compile time).
template <bool FULL>
__device__ __forceinline__ unsigned l2_column(const uint8_t *p, int ws, unsigned sh, const unsigned *f0, const unsigned *f1, int nrow, int R, int side, int mx, bool active,
                                              int bh, unsigned m0, unsigned m1)
{
    unsigned loc = 0xFFFFFFFFu;
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned amx = (unsigned)abs(mx);
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
            if (my >= -R && my <= R && active) loc = min(loc, ((sad + amx + (unsigned)abs(my)) << 16) | (unsigned)((my + R) * side + mx + R));
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

// Work-group = 4 x 2 blocks of 8x8 L2 samples.  LB lanes per block (LB = 32 when 2R + 1 <= 32: two blocks per wave; else 64), lane = one horizontal
// displacement.  Then every block's four L1 children (+-2 around twice its vector) by the same lanes, samples straight from the (cached) L1 planes.
// dynamic LDS: window (8 + nrow) rows x ws bytes, then the 16 x 32 source tile.
template <int LB>
__global__ __launch_bounds__(256) void presearch_l2l1_kernel(int W2, int H2, int R, const uint8_t *c2, const uint8_t *r2, int nbx, int nby,
                                                             int W1, int H1, const uint8_t *c1, const uint8_t *r1, short2 *mv1, int nb1x, int nb1y)
{
    extern __shared__ unsigned char lds[];
    constexpr int BPW = 64 / LB, BX = LB == 32 ? 4 : 2;                    // blocks per wave; blocks per work-group row (x 2 rows = 8 or 4 blocks)
    const int side = 2 * R + 1, nrow = (8 + 2 * R + 7) & ~7, wd = BX * 8 + 2 * R, wrows = 8 + nrow;
    const int ws = (((wd + 8 + 3) >> 2) | 1) << 2;
    constexpr int L1_BYTES = 256 + 20 * 24;
    uint8_t *win = lds, *cur = lds + wrows * ws;                            // cur: 16 rows x 32 bytes
    __shared__ unsigned best1[8][4];
    __shared__ __attribute__((aligned(16))) uint8_t l1buf[8 * L1_BYTES];
    const int bx0 = blockIdx.x * BX, by0 = blockIdx.y * 2, x0 = bx0 * 8, y0 = by0 * 8, t = threadIdx.x;
    for (int i = t; i < wrows * (ws >> 2); i += 256) {
        const int y = i / (ws >> 2), xd = (i % (ws >> 2)) * 4;
        const int ry = min(max(y0 - R + y, 0), H2 - 1);
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (unsigned)r2[(long)ry * W2 + min(max(x0 - R + xd + b, 0), W2 - 1)] << (8 * b);
        *(unsigned *)(win + y * ws + xd) = v;
    }
    for (int i = t; i < 16 * 32; i += 256) {
        const int y = i >> 5, x = i & 31;
        cur[i] = (y0 + y < H2 && x0 + x < W2 && x < BX * 8) ? c2[(long)(y0 + y) * W2 + x0 + x] : 0;
    }
    if (t < 32) best1[t >> 2][t & 3] = 0xFFFFFFFFu;
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
    if (bw == 8 && bh == 8) loc = l2_column<true>(p, ws, wx & 3, f0, f1, nrow, R, side, mx, active, 8, m0, m1);
    else loc = l2_column<false>(p, ws, wx & 3, f0, f1, nrow, R, side, mx, active, bh, m0, m1);
#pragma unroll
    for (int m = 1; m < LB; m <<= 1) loc = min(loc, (unsigned)__shfl_xor((int)loc, m, 64));     // minimum over the block's lanes
    const int c = (int)(loc & 0xFFFFu), vx = c % side - R, vy = c / side - R;
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
__global__ __launch_bounds__(256) void presearch_l0_kernel(KsGeom g, int range, const uint8_t *src, const uint8_t *ref, const short2 *mv1, short2 *field, int nbx, int nby)
{
    const int blk = blockIdx.x * 16 + (threadIdx.x >> 4), k = threadIdx.x & 15;
    if (blk >= nbx * nby) return;
    const int bx = blk % nbx, by = blk / nbx;
    const short2 p = mv1[blk];
    unsigned key = 0xFFFFFFFFu;
    int mx = 0, my = 0;
    if (k < 9) {
        mx = clip3(-range, range, 2 * p.x + (k % 3 - 1)); my = clip3(-range, range, 2 * p.y + (k / 3 - 1));
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
    const size_t n1 = (size_t)(W / 2) * (H / 2), n2 = (size_t)(W / 4) * (H / 4);
    const size_t nb2 = (size_t)((W / 4 + 7) / 8) * ((H / 4 + 7) / 8), nb1 = (size_t)((W / 2 + 7) / 8) * ((H / 2 + 7) / 8), nb0 = (size_t)((W + 15) / 16) * ((H + 15) / 16);
    const size_t sz[7] = {n1 + 16, n2 + 16, n1 + 16, n2 + 16, nb2 * 4, nb1 * 4, nb0 * 4};
    for (int i = 0; i < 7; ++i) {
        void *p = nullptr;
        const int r = ks265_hip(f->ctx, hipMalloc(&p, sz[i]));
        if (r) return r;
        f->pyr[i] = (uint8_t *)p;
    }
    return KS265_OK;
}

extern "C" int ks265_presearch(ks265_frame *f, ks265_pic src, ks265_pic ref, int16_t *dev_field)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y) return KS265_POINTER;
    (void)hipSetDevice(f->ctx->device);
    int r = presearch_alloc(f);
    if (r) return r;
    const KsGeom &g = f->g;
    const int W1 = g.W / 2, H1 = g.H / 2, W2 = g.W / 4, H2 = g.H / 4;
    const int nb2x = (W2 + 7) / 8, nb2y = (H2 + 7) / 8, nb1x = (W1 + 7) / 8, nb1y = (H1 + 7) / 8, nb0x = (g.W + 15) / 16, nb0y = (g.H + 15) / 16;
    uint8_t *c1 = f->pyr[0], *c2 = f->pyr[1], *r1 = f->pyr[2], *r2 = f->pyr[3];
    short2 *mv1 = (short2 *)f->pyr[5], *field = (short2 *)dev_field;      // dev_field = NULL: L2 / L1 only (ks265_me_integer)
    hipStream_t st = f->ctx->stream;
    const dim3 gd((unsigned)((g.W + 127) / 128), (unsigned)((g.H + 63) / 64));
    hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, st, g, src.y, c1, c2);
    hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, st, g, ref.y, r1, r2);
    const int R = max((f->cfg.me_range >> 2) - 1, 1), wrows = 8 + ((8 + 2 * R + 7) & ~7);
    if (2 * R + 1 > 64) return KS265_NOTSUPPORTED;                            // one lane per horizontal displacement (me_range <= 128)
    if (2 * R + 1 <= 32) {
        const int wd = 32 + 2 * R, ws = (((wd + 8 + 3) >> 2) | 1) << 2;
        hipLaunchKernelGGL(presearch_l2l1_kernel<32>, dim3((unsigned)((nb2x + 3) / 4), (unsigned)((nb2y + 1) / 2)), dim3(256), (size_t)(wrows * ws + 512), st, W2, H2, R, c2, r2, nb2x, nb2y,
                           W1, H1, c1, r1, mv1, nb1x, nb1y);
    } else {
        const int wd = 16 + 2 * R, ws = (((wd + 8 + 3) >> 2) | 1) << 2;
        hipLaunchKernelGGL(presearch_l2l1_kernel<64>, dim3((unsigned)((nb2x + 1) / 2), (unsigned)((nb2y + 1) / 2)), dim3(256), (size_t)(wrows * ws + 512), st, W2, H2, R, c2, r2, nb2x, nb2y,
                           W1, H1, c1, r1, mv1, nb1x, nb1y);
    }
    // the full-resolution step: stage A does it itself from its LDS window (frame_me_int.hip); the stand-alone kernel serves the stage API
    if (field) hipLaunchKernelGGL(presearch_l0_kernel, dim3((unsigned)((nb0x * nb0y + 15) / 16)), dim3(256), 0, st, g, f->cfg.me_range, src.y, ref.y, mv1, field, nb0x, nb0y);
    return ks265_check_launch(f->ctx);
}
