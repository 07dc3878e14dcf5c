// frame_presearch.hip — Stage A0: motion pre-search on a three-level pyramid (cfg.pre_search).
//
// The reference's lookahead runs a low-resolution motion search on 2:1 pictures (downsample_c enc@0x4a6a60 feeds its frame-cost
// estimate) and meInitPoint enc@0x48af50 starts every integer search from the best of several candidate
// vectors.  A frame-parallel search has no already coded neighbours to take candidates from; what makes the local pattern searches
// of stage A (DIA / HEX / UMH walk downhill from their start point) robust against displaced texture without a gradient comes from
// here instead: an EXHAUSTIVE search where it is cheap.
//   L1 = downsample_c(luma), L2 = downsample_c(L1)                                  (pyr_down_kernel: both levels in one pass)
//   per 8x8 block of L2 (32x32 samples): every vector of +-range/4                   (presearch_l2_kernel, LDS window, v_sad_u8)
//   per 8x8 block of L1 (16x16 samples): +-2 around twice the L2 vector              (presearch_l1_kernel)
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

// ------------------------------------------------------------------ L2: exhaustive search, work-group = 2x2 blocks of 8x8
// dynamic LDS: window (16 + 2R) rows x ws bytes (ws = 16 + 2R + 8 rounded so that the dword stride is odd), then the 16x16 source tile
__global__ __launch_bounds__(256) void presearch_l2_kernel(int W2, int H2, int R, const uint8_t *c2, const uint8_t *r2, short2 *mv2, int nbx, int nby)
{
    extern __shared__ unsigned char lds[];
    const int wd = 16 + 2 * R, ws = (((wd + 8 + 3) >> 2) | 1) << 2;
    uint8_t *win = lds, *cur = lds + wd * ws;                               // cur: 16 rows x 16 bytes
    __shared__ unsigned best[4];
    const int bx0 = blockIdx.x * 2, by0 = blockIdx.y * 2, x0 = bx0 * 8, y0 = by0 * 8, t = threadIdx.x;
    for (int i = t; i < wd * (ws >> 2); i += 256) {
        const int y = i / (ws >> 2), xd = (i % (ws >> 2)) * 4;
        const int ry = min(max(y0 - R + y, 0), H2 - 1);
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (unsigned)r2[(long)ry * W2 + min(max(x0 - R + xd + b, 0), W2 - 1)] << (8 * b);
        *(unsigned *)(win + y * ws + xd) = v;
    }
    {
        const int y = t >> 4, x = t & 15;
        cur[t] = (y0 + y < H2 && x0 + x < W2) ? c2[(long)(y0 + y) * W2 + x0 + x] : 0;
    }
    if (t < 4) best[t] = 0xFFFFFFFFu;
    __syncthreads();
    const int side = 2 * R + 1, ncand = side * side;
    for (int blk = 0; blk < 4; ++blk) {
        const int bx = bx0 + (blk & 1), by = by0 + (blk >> 1);
        if (bx >= nbx || by >= nby) continue;
        const int bw = min(8, W2 - 8 * bx), bh = min(8, H2 - 8 * by);       // W2, H2 are even
        const unsigned m0 = bw >= 4 ? 0xFFFFFFFFu : 0x0000FFFFu, m1 = bw == 8 ? 0xFFFFFFFFu : bw == 6 ? 0x0000FFFFu : 0u;
        const uint8_t *f = cur + (blk >> 1) * 8 * 16 + (blk & 1) * 8;
        unsigned loc = 0xFFFFFFFFu;
        for (int c = t; c < ncand; c += 256) {
            const int my = c / side - R, mx = c % side - R;
            const int wx = (blk & 1) * 8 + mx + R, wy = (blk >> 1) * 8 + my + R;
            const uint8_t *p = win + wy * ws + (wx & ~3);
            const unsigned sh = wx & 3;
            unsigned sad = 0;
            for (int r = 0; r < bh; ++r) {
                const unsigned w0 = *(const unsigned *)(p + r * ws), w1 = *(const unsigned *)(p + r * ws + 4), w2 = *(const unsigned *)(p + r * ws + 8);
                const unsigned f0 = *(const unsigned *)(f + r * 16), f1 = *(const unsigned *)(f + r * 16 + 4);
                sad = sad_u8x4(f0 & m0, align_bytes(w1, w0, sh) & m0, sad);
                sad = sad_u8x4(f1 & m1, align_bytes(w2, w1, sh) & m1, sad);
            }
            const unsigned key = ((sad + (unsigned)(abs(mx) + abs(my))) << 16) | (unsigned)c;
            loc = min(loc, key);
        }
        atomicMin(&best[blk], loc);
    }
    __syncthreads();
    if (t < 4) {
        const int bx = bx0 + (t & 1), by = by0 + (t >> 1);
        if (bx < nbx && by < nby) {
            const int c = (int)(best[t] & 0xFFFFu);
            mv2[by * nbx + bx] = make_short2((short)(c % side - R), (short)(c / side - R));
        }
    }
}

// SAD of a bw x bh block (bw, bh <= 16, bw a multiple of 2) at packed low-resolution planes, reference reads clamped to the picture
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
__device__ __forceinline__ unsigned min_over_32(unsigned v)               // minimum over the 32 lanes of a half wave
{
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v = min(v, (unsigned)__shfl_xor((int)v, m, 64));
    return v;
}

// ------------------------------------------------------------------ L1: +-2 around twice the parent's vector; 32 lanes per block (25 candidates)
__global__ __launch_bounds__(256) void presearch_l1_kernel(int W1, int H1, const uint8_t *c1, const uint8_t *r1, const short2 *mv2, int nb2x, short2 *mv1, int nbx, int nby)
{
    const int blk = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    if (blk >= nbx * nby) return;
    const int bx = blk % nbx, by = blk / nbx;
    const short2 p = mv2[(by >> 1) * nb2x + (bx >> 1)];
    unsigned key = 0xFFFFFFFFu;
    int mx = 0, my = 0;
    if (k < 25) {
        mx = 2 * p.x + (k % 5 - 2); my = 2 * p.y + (k / 5 - 2);
        const unsigned sad = sad_clamped(c1, r1, W1, H1, 8 * bx, 8 * by, min(8, W1 - 8 * bx), min(8, H1 - 8 * by), mx, my);
        key = ((sad + (unsigned)(abs(mx) + abs(my))) << 8) | (unsigned)k;
    }
    const unsigned b = min_over_32(key);
    if (k == (int)(b & 31u) && key == b) mv1[blk] = make_short2((short)mx, (short)my);
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
    short2 *mv2 = (short2 *)f->pyr[4], *mv1 = (short2 *)f->pyr[5], *field = dev_field ? (short2 *)dev_field : (short2 *)f->pyr[6];
    hipStream_t st = f->ctx->stream;
    const dim3 gd((unsigned)((g.W + 127) / 128), (unsigned)((g.H + 63) / 64));
    hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, st, g, src.y, c1, c2);
    hipLaunchKernelGGL(pyr_down_kernel, gd, dim3(256), 0, st, g, ref.y, r1, r2);
    const int R = max(f->cfg.me_range >> 2, 1), wd = 16 + 2 * R, ws = (((wd + 8 + 3) >> 2) | 1) << 2;
    hipLaunchKernelGGL(presearch_l2_kernel, dim3((unsigned)((nb2x + 1) / 2), (unsigned)((nb2y + 1) / 2)), dim3(256), (size_t)(wd * ws + 256), st, W2, H2, R, c2, r2, mv2, nb2x, nb2y);
    hipLaunchKernelGGL(presearch_l1_kernel, dim3((unsigned)((nb1x * nb1y + 7) / 8)), dim3(256), 0, st, W1, H1, c1, r1, mv2, nb2x, mv1, nb1x, nb1y);
    hipLaunchKernelGGL(presearch_l0_kernel, dim3((unsigned)((nb0x * nb0y + 15) / 16)), dim3(256), 0, st, g, f->cfg.me_range, src.y, ref.y, mv1, field, nb0x, nb0y);
    return ks265_check_launch(f->ctx);
}
