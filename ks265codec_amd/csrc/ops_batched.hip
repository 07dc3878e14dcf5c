// ops_batched.hip — batched forms of the reference's per-block operator tables (include/ks265_hip.h §2).
// One wavefront (or one workgroup for transforms) per block descriptor; the arithmetic comes from
// ks265_dev.h and is bit-exact with the reference `_c` kernels (tests/test_gpu_golden.py).
#include "ks265_internal.h"
#include "intra_dev.h"
#include "recon_dev.h"

using namespace ks265;

// ------------------------------------------------------------------ distortion family
// mode: 0 SAD, 1 SSE.  One wave per (block, candidate); candidates: K offsets relative to b_off.
template <int MODE>
__device__ __forceinline__ unsigned block_dist(const uint8_t *a, int sa, const uint8_t *b, int sb, int w, int h, int lane)
{
    unsigned acc = 0;
    if ((w & 3) == 0) {
        // lane walks 4-pixel groups: coalesced along rows
        int gw = w >> 2, ng = gw * h;
        for (int g = lane; g < ng; g += 64) {
            int y = g / gw, x = (g - y * gw) << 2;
            const uint8_t *pa = a + (long)y * sa + x, *pb = b + (long)y * sb + x;
            unsigned va = pa[0] | (pa[1] << 8) | (pa[2] << 16) | ((unsigned)pa[3] << 24);
            unsigned vb = pb[0] | (pb[1] << 8) | (pb[2] << 16) | ((unsigned)pb[3] << 24);
            if (MODE == 0) acc = sad_u8x4(va, vb, acc);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { int d = (int)((va >> (8 * i)) & 255) - (int)((vb >> (8 * i)) & 255); acc += (unsigned)(d * d); }
            }
        }
    } else {
        for (int i = lane; i < w * h; i += 64) {
            int y = i / w, x = i - y * w;
            int d = (int)a[(long)y * sa + x] - (int)b[(long)y * sb + x];
            acc += MODE == 0 ? (unsigned)abs(d) : (unsigned)(d * d);
        }
    }
    return wave_sum(acc);
}

// K candidates per block: K = 1 (sad/sse), 3 (sad3), 4 (sad4: up, down, left, right, << 4)
template <int MODE, int K>
__global__ __launch_bounds__(256) void dist_batch_kernel(const uint8_t *a, int sa, const uint8_t *b, int sb, const void *blks, int n,
                                                         uint32_t *out)
{
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n * K) return;
    int i = wave / K, k = wave - i * K;
    int a_off, b_off, w, h;
    if constexpr (K == 3) {
        const ks265_blk3 d = ((const ks265_blk3 *)blks)[i];
        a_off = d.a_off; b_off = d.b_off[k]; w = d.w; h = d.h;
    } else {
        const ks265_blk d = ((const ks265_blk *)blks)[i];
        a_off = d.a_off; b_off = d.b_off; w = d.w; h = d.h;
        if constexpr (K == 4) b_off += k == 0 ? -sb : k == 1 ? sb : k == 2 ? -1 : 1;
    }
    unsigned s = block_dist<MODE>(a + a_off, sa, b + b_off, sb, w, h, lane);
    if (lane == 0) out[wave] = K == 4 ? s << 4 : s;
}

// sad4blk_8x8_c enc@0x4cee30: quadrant k of a 16x16
__global__ __launch_bounds__(256) void sad4blk_kernel(const uint8_t *a, int sa, const uint8_t *b, int sb, const ks265_blk *blks, int n,
                                                      uint32_t *out)
{
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n * 4) return;
    int i = wave >> 2, k = wave & 3;
    const ks265_blk d = blks[i];
    int ox = (k & 1) * 8, oy = (k >> 1) * 8;
    unsigned s = block_dist<0>(a + d.a_off + (long)oy * sa + ox, sa, b + d.b_off + (long)oy * sb + ox, sb, 8, 8, lane);
    if (lane == 0) out[wave] = s;
}

// had_c enc@0x47b680 / xCalcHADs8x8 enc@0x47b3b0: one wave per block, one 8x8 (or 4x4 / 2x2) tile set per pass
__global__ __launch_bounds__(256) void had_batch_kernel(const uint8_t *a, int sa, const uint8_t *b, int sb, const ks265_blk *blks, int n,
                                                        uint32_t *out)
{
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n) return;
    const ks265_blk d = blks[wave];
    const uint8_t *pa = a + d.a_off, *pb = b + d.b_off;
    int w = d.w, h = d.h;
    unsigned total = 0;
    if (((w | h) & 7) == 0) {
        int tx = w >> 3, nt = tx * (h >> 3);
        for (int t = 0; t < nt; ++t) {
            int ty = t / tx, x0 = (t - ty * tx) << 3, y0 = ty << 3;
            int x = lane & 7, y = lane >> 3;
            int v = (int)pa[(long)(y0 + y) * sa + x0 + x] - (int)pb[(long)(y0 + y) * sb + x0 + x];
            total += (had8x8_abs_sum(v, lane) + 2) >> 2;
        }
    } else if (((w | h) & 3) == 0) {
        // four 4x4 tiles per pass: lane = tile*16 + y*4 + x
        int tx = w >> 2, nt = tx * (h >> 2);
        for (int t0 = 0; t0 < nt; t0 += 4) {
            int t = t0 + (lane >> 4);
            int v = 0;
            if (t < nt) {
                int ty = t / tx, x0 = (t - ty * tx) << 2, y0 = ty << 2, x = lane & 3, y = (lane >> 2) & 3;
                v = (int)pa[(long)(y0 + y) * sa + x0 + x] - (int)pb[(long)(y0 + y) * sb + x0 + x];
            }
            unsigned s = had4x4_abs_sum16(v, lane);   // per 16-lane group
            s = (s + 1) >> 1;
            // add the four group results (each uniform within its 16 lanes)
            unsigned g = (lane & 15) == 0 && t < nt ? s : 0;
            total += wave_sum(g);
        }
    } else {
        // sixteen 2x2 tiles per pass: lane = tile*4 + y*2 + x
        int tx = w >> 1, nt = tx * (h >> 1);
        for (int t0 = 0; t0 < nt; t0 += 16) {
            int t = t0 + (lane >> 2);
            int v = 0;
            if (t < nt) {
                int ty = t / tx, x0 = (t - ty * tx) << 1, y0 = ty << 1, x = lane & 1, y = (lane >> 1) & 1;
                v = (int)pa[(long)(y0 + y) * sa + x0 + x] - (int)pb[(long)(y0 + y) * sb + x0 + x];
            }
            int p = lane_xor<1>(v); v = (lane & 1) ? p - v : v + p;
            p = lane_xor<2>(v);     v = (lane & 2) ? p - v : v + p;
            total += wave_sum((unsigned)abs(v));
        }
    }
    if (lane == 0) out[wave] = total;
}

// ------------------------------------------------------------------ residual
__global__ __launch_bounds__(256) void residual_batch_kernel(const uint8_t *org, int so, const uint8_t *pred, int sp, const ks265_blk *blks,
                                                             int n, int16_t *res)
{
    int i = blockIdx.x;
    if (i >= n) return;
    const ks265_blk d = blks[i];
    int nn = d.w;
    long base = 0;
    for (int j = 0; j < i; ++j) base += (long)blks[j].w * blks[j].w;   // packed output; batches are small in the parity path
    for (int t = threadIdx.x; t < nn * nn; t += 256) {
        int y = t / nn, x = t - y * nn;
        res[base + t] = (int16_t)((int)org[d.a_off + (long)y * so + x] - (int)pred[d.b_off + (long)y * sp + x]);
    }
}

// ------------------------------------------------------------------ transforms (one workgroup per TU)
// forward: H265_2dDct*_c enc@0x4c2210.. / H265_2dDst4x4_c enc@0x4c2250 — SURVEY.md B.3
template <int N>
__global__ __launch_bounds__(256) void fwd_transform_kernel(int idx, const int16_t *src, int16_t *dst, int nblk)
{
    __shared__ short M[N * N];
    __shared__ short X[N * N];
    __shared__ short T[N * N];
    int b = blockIdx.x;
    if (b >= nblk) return;
    load_matrix(M, idx, N, threadIdx.x, 256);
    for (int t = threadIdx.x; t < N * N; t += 256) X[t] = src[(long)b * N * N + t];
    __syncthreads();
    fwd_transform_lds<N>(M, X, T, threadIdx.x, 256);
    __syncthreads();
    for (int t = threadIdx.x; t < N * N; t += 256) dst[(long)b * N * N + t] = X[t];
}

// inverse + pred add + clip: H265_2dIDct*_c enc@0x448f60.. / H265_2dIDst4x4_c enc@0x448c40 — SURVEY.md B.4
template <int N>
__global__ __launch_bounds__(256) void inv_transform_kernel(int idx, const int16_t *coef, const uint8_t *pred, uint8_t *dst, int nblk)
{
    __shared__ short M[N * N];
    __shared__ short X[N * N];
    __shared__ short T[N * N];
    int b = blockIdx.x;
    if (b >= nblk) return;
    load_matrix(M, idx, N, threadIdx.x, 256);
    for (int t = threadIdx.x; t < N * N; t += 256) X[t] = coef[(long)b * N * N + t];
    __syncthreads();
    inv_transform_lds<N>(M, X, T, threadIdx.x, 256);
    __syncthreads();
    for (int t = threadIdx.x; t < N * N; t += 256) dst[(long)b * N * N + t] = (uint8_t)clip8((int)pred[(long)b * N * N + t] + (int)X[t]);
}

__global__ __launch_bounds__(256) void quant_batch_kernel(int nn, const int16_t *coef, int16_t *lvl, int16_t *deltaU, int32_t *nz,
                                                          int scale, int off, int qbits, int nblk)
{
    int b = blockIdx.x;
    if (b >= nblk) return;
    unsigned cnt = 0;
    for (int t = threadIdx.x; t < nn; t += 256) {
        int du, l = quant_one(coef[(long)b * nn + t], scale, off, qbits, du);
        lvl[(long)b * nn + t] = (int16_t)l;
        deltaU[(long)b * nn + t] = (int16_t)du;
        cnt += l != 0;
    }
    __shared__ unsigned part[4];
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) nz[b] = (int)(part[0] + part[1] + part[2] + part[3]);
}

// H265DeQuantBlock_c enc@0x439210 with its lastX / lastY arguments: only rows 0..lastY and columns 0..round_up(lastX + 1, 4) - 1 are written
__global__ __launch_bounds__(256) void dequant_rect_batch_kernel(const int16_t *lvl, int16_t *coef, int n, int stride, int scale, int add, int shift, int lastX, int lastY, int nblk)
{
    const int cols = (lastX + 4) & ~3, per = cols * (lastY + 1);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)per * nblk) return;
    const int b = (int)(t / per), r = (int)(t % per), y = r / cols, x = r % cols;
    const long o = (long)b * n * stride + (long)y * stride + x;
    coef[o] = (int16_t)dequant_one(lvl[o], scale, add, shift);
}
__global__ __launch_bounds__(256) void dequant_batch_kernel(const int16_t *lvl, int16_t *coef, int scale, int add, int shift, long total)
{
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < total) coef[t] = (int16_t)dequant_one(lvl[t], scale, add, shift);
}

// ------------------------------------------------------------------ deblocking edges: one thread per 4-line segment
__global__ __launch_bounds__(64) void edge_luma_batch_kernel(uint8_t *plane, int stride, const ks265_edge *edges, int n)
{
    int e = blockIdx.x;
    if (e >= n) return;
    const ks265_edge d = edges[e];
    int seg = threadIdx.x;
    if (seg >= d.length / 4) return;
    long xs = d.dir == 0 ? 1 : stride, ys = d.dir == 0 ? stride : 1;
    uint8_t *p = plane + d.pix_off + (long)seg * 4 * ys;
    int px[4][8];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int i = 0; i < 8; ++i) px[l][i] = p[l * ys + (i - 4) * xs];
    deblock_luma_segment(px, d.beta, d.tc, d.flags & 1, d.flags & 2);
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int i = 1; i < 7; ++i) p[l * ys + (i - 4) * xs] = (uint8_t)px[l][i];
}

__global__ __launch_bounds__(64) void edge_chroma_batch_kernel(uint8_t *plane, int stride, const ks265_edge *edges, int n)
{
    int e = blockIdx.x;
    if (e >= n) return;
    const ks265_edge d = edges[e];
    int l = threadIdx.x;
    if (l >= d.length) return;
    long xs = d.dir == 0 ? 1 : stride, ys = d.dir == 0 ? stride : 1;
    uint8_t *p = plane + d.pix_off + (long)l * ys;
    int p1 = p[-2 * xs], p0 = p[-xs], q0 = p[0], q1 = p[xs];
    deblock_chroma_line(p1, p0, q0, q1, d.tc, d.flags & 1, d.flags & 2);
    p[-xs] = (uint8_t)p0;
    p[0] = (uint8_t)q0;
}

// ------------------------------------------------------------------ interpolation on rectangles
template <typename SRC, typename DST, int NT, int OUT /*0: (s+32)>>6 clip8, 1: raw s16, 2: (s+2048)>>12 clip8, 3: s>>6*/>
__global__ __launch_bounds__(256) void interp_rect_kernel(DST *dst, int ds, const SRC *src, int ss, int w, int h, int frac, int vertical)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    int step = vertical ? ss : 1, sum = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int c = NT == 8 ? (int)kLumaTaps[frac][i] : (int)kChromaTaps[frac][i];
        sum += c * (int)src[(long)y * ss + x + (i - (NT / 2 - 1)) * step];
    }
    int r = OUT == 0 ? clip8((sum + 32) >> 6) : OUT == 1 ? sum : OUT == 2 ? clip8((sum + 2048) >> 12) : (sum >> 6);
    dst[(long)y * ds + x] = (DST)r;
}

// ------------------------------------------------------------------ SAO
// SaoApplyOffsetBo_c enc@0x43e4e0 (in place, width rounded up to 4, no band wrap)
__global__ __launch_bounds__(256) void sao_bo_rect_kernel(uint8_t *rec, int stride, int height, int cols, int band, int o0, int o1, int o2, int o3)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= height) return;
    int v = rec[(long)y * stride + x], k = (v >> 3) - band;
    int o = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : k == 3 ? o3 : 0;
    rec[(long)y * stride + x] = (uint8_t)clip8(v + o);
}

// SaoApplyOffsetEo{0..3}_c enc@0x43e650.. plain mode, out of place
__global__ __launch_bounds__(256) void sao_eo_rect_kernel(const uint8_t *src, uint8_t *dst, int stride, int height, int width, int dx, int dy,
                                                          int o0, int o1, int o2, int o3, int o4)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    int c = src[(long)y * stride + x], a = src[(long)(y - dy) * stride + x - dx], b = src[(long)(y + dy) * stride + x + dx];
    int e = 2 + sgn(c - a) + sgn(c - b);
    int o = e == 0 ? o0 : e == 1 ? o1 : e == 2 ? o2 : e == 3 ? o3 : o4;
    dst[(long)y * stride + x] = (uint8_t)clip8(c + o);
}

// statSaoBoEo01_c enc@0x4ae9c0: packed (sum << 12 | count) accumulators, one workgroup per rectangle
__global__ __launch_bounds__(256) void sao_stats_batch_kernel(const uint8_t *org, int os, const uint8_t *rec, int rs, const ks265_sao_rect *rects,
                                                              int nrect, int rowStep, int32_t *out)
{
    __shared__ int acc[96];
    int r = blockIdx.x;
    if (r >= nrect) return;
    if (threadIdx.x < 96) acc[threadIdx.x] = 0;
    __syncthreads();
    const ks265_sao_rect d = rects[r];
    int rows = (d.h + rowStep - 1) / rowStep;
    for (int t = threadIdx.x; t < rows * d.w; t += 256) {
        int yy = t / d.w, x = t - yy * d.w, y = yy * rowStep;
        const uint8_t *p = rec + d.rec_off + (long)y * rs + x;
        int c = p[0];
        int dlt = (int)(int8_t)(uint8_t)(org[d.org_off + (long)y * os + x] - c);
        int v = (int)(((unsigned)dlt << 12) | 1u);
        int c0 = 2 + sgn(c - (int)p[-1]) + sgn(c - (int)p[1]);
        int c1 = 2 + sgn(c - (int)p[-rs]) + sgn(c - (int)p[rs]);
        atomicAdd(&acc[64 + (c >> 3)], v);
        atomicAdd(&acc[(c1 << 3) | c0], v);
    }
    __syncthreads();
    if (threadIdx.x < 96) out[(long)r * 96 + threadIdx.x] = acc[threadIdx.x];
}

// ------------------------------------------------------------------ host entry points
#define CHECK_CTX(ctx) do { if (!(ctx)) return KS265_POINTER; } while (0)
#define LAUNCH_END(ctx) return ks265_check_launch(ctx)

// ------------------------------------------------------------------ intra prediction (g_IntraPredFunction enc@0x7070a0)
// one wave per block: the 4N + 1 reference samples go to LDS, every lane predicts samples lane, lane + 64, ...
__global__ __launch_bounds__(256) void intra_pred_batch_kernel(const uint8_t *ref, uint8_t *dst, const ks265_intra_blk *blks, int n)
{
    __shared__ uint8_t sref[4][132];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + w;
    if (b >= n) return;                                             // wave-uniform; no block-wide barrier below
    const ks265_intra_blk d = blks[b];
    const int N = 1 << d.log2;
    uint8_t *r = &sref[w][2 * 32 + 1];                              // corner
    for (int i = lane; i < 4 * N + 1; i += 64) r[i - 2 * N] = ref[d.ref_off + i - 2 * N];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    int dc = N;
    if (d.mode == 1) {
        for (int i = 0; i < N; ++i) dc += r[1 + i] + r[-1 - i];
        dc >>= d.log2 + 1;
    }
    for (int p = lane; p < N * N; p += 64) {
        const int x = p & (N - 1), y = p >> d.log2;
        dst[d.dst_off + y * d.dst_stride + x] = (uint8_t)intra_sample(r, d.mode, d.log2, x, y, dc, d.edge_filter != 0);
    }
}

// IntraPredFilterRef_c enc@0x424110: one wave per reference array
__global__ __launch_bounds__(256) void intra_filter_ref_batch_kernel(const uint8_t *src, uint8_t *dst, const ks265_intra_ref *refs, int n)
{
    __shared__ uint8_t sref[4][132];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + w;
    if (b >= n) return;
    const ks265_intra_ref d = refs[b];
    uint8_t *r = &sref[w][2 * 32 + 1];
    for (int i = lane; i < 4 * d.size + 1; i += 64) r[i - 2 * d.size] = src[d.src_off + i - 2 * d.size];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    const bool bil = d.size == 32 && d.strong_enabled && intra_strong_flat(r);
    for (int i = lane; i < 4 * d.size + 1; i += 64) dst[d.dst_off + i - 2 * d.size] = (uint8_t)intra_filtered(r, d.size, i - 2 * d.size, bil);
}

// ------------------------------------------------------------------ lookahead leaf kernels
// downsample_c enc@0x4a6a60: one thread = 4 adjacent output samples (8 source bytes of two rows)
__global__ __launch_bounds__(256) void downsample_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t *p = src + (long)(2 * y) * ss + 2 * x;
    if (x + 4 <= w) {
        uint2 a, b;
        __builtin_memcpy(&a, p, 8);
        __builtin_memcpy(&b, p + ss, 8);
        const unsigned long long ra = a.x | ((unsigned long long)a.y << 32), rb = b.x | ((unsigned long long)b.y << 32);
        for (int i = 0; i < 4; ++i) {
            const int u = (int)((((ra >> (16 * i)) & 255) + ((rb >> (16 * i)) & 255) + 1) >> 1), v = (int)((((ra >> (16 * i + 8)) & 255) + ((rb >> (16 * i + 8)) & 255) + 1) >> 1);
            dst[(long)y * ds + x + i] = (uint8_t)((u + v + 1) >> 1);
        }
    } else {
        for (int i = 0; x + i < w; ++i) {
            const int u = (p[2 * i] + p[2 * i + ss] + 1) >> 1, v = (p[2 * i + 1] + p[2 * i + 1 + ss] + 1) >> 1;
            dst[(long)y * ds + x + i] = (uint8_t)((u + v + 1) >> 1);
        }
    }
}
// the same from PINNED HOST memory (the encoder host's input picture), for the lookahead at the input: 64 work-groups walk the picture - a wave per output row, 8-byte loads, four in
// flight per lane - so that the kernel is bound by PCIe reads without holding the machine's wave slots while it waits for them (the one-thread-per-4-samples grid above, 8 100
// waves stalled on PCIe latency, filled every wave slot of the GPU for the duration: - 28 % encoder throughput, measured in round 4)
__global__ __launch_bounds__(256) void downsample_host_kernel(const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h)
{
    const int lane = threadIdx.x & 63;
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < h; y += gridDim.x * 4) {
        const uint8_t *p = src + (long)(2 * y) * ss;
        for (int x0 = 0; x0 < w; x0 += 512) {                       // 512 output samples per wave iteration: lane = 4 + 4 of them
            uint2 a[2], b[2];
            bool on[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int x = x0 + k * 256 + lane * 4;
                on[k] = x + 4 <= w;
                if (on[k]) { a[k] = *(const uint2 *)(p + 2 * x); b[k] = *(const uint2 *)(p + ss + 2 * x); }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int x = x0 + k * 256 + lane * 4;
                if (on[k]) {
                    const unsigned long long ra = a[k].x | ((unsigned long long)a[k].y << 32), rb = b[k].x | ((unsigned long long)b[k].y << 32);
                    unsigned o = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = (int)((((ra >> (16 * i)) & 255) + ((rb >> (16 * i)) & 255) + 1) >> 1), v = (int)((((ra >> (16 * i + 8)) & 255) + ((rb >> (16 * i + 8)) & 255) + 1) >> 1);
                        o |= (unsigned)((u + v + 1) >> 1) << (8 * i);
                    }
                    *(unsigned *)(dst + (long)y * ds + x) = o;
                } else
                    for (int i = 0; x + i < w; ++i) {
                        const int u = (p[2 * (x + i)] + p[2 * (x + i) + ss] + 1) >> 1, v = (p[2 * (x + i) + 1] + p[2 * (x + i) + 1 + ss] + 1) >> 1;
                        dst[(long)y * ds + x + i] = (uint8_t)((u + v + 1) >> 1);
                    }
            }
        }
    }
}
// weightBi_sad_c enc@0x4a7170: one wave per block
__global__ __launch_bounds__(256) void weight_bi_sad_batch_kernel(const uint8_t *org, int so, const uint8_t *r0, int s0, const uint8_t *r1, int s1,
                                                                  const ks265_blk3 *blks, int n, uint32_t *out)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n) return;
    const ks265_blk3 d = blks[b];
    unsigned acc = 0;
    for (int i = lane; i < d.w * d.h; i += 64) {
        const int x = i % d.w, y = i / d.w;
        const int p = (r0[d.b_off[0] + y * s0 + x] + r1[d.b_off[1] + y * s1 + x] + 1) >> 1;
        acc += (unsigned)abs(p - (int)org[d.a_off + y * so + x]);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[b] = acc;
}
// interMeBiFull_c enc@0x4896d0 / interMeBiHadFull_c enc@0x4897e0: one wave per block, lane = window position (y * 8 + x); out = {cost, (y << 16) | x}
template <bool HAD>
__global__ __launch_bounds__(256) void bi_full_batch_kernel(const uint8_t *org, int so, const uint8_t *ref, int sr, const ks265_blk *blks, const uint16_t *mvcost,
                                                            int n, uint32_t *out)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n) return;
    const ks265_blk d = blks[b];
    const uint16_t *mc = mvcost + (long)b * 16;
    const uint8_t *po = org + d.a_off;
    unsigned cost;
    if (!HAD) {
        const uint8_t *pr = ref + d.b_off + (long)(lane >> 3) * sr + (lane & 7);
        unsigned acc = 0;
        for (int y = 0; y < d.h; ++y)
            for (int x = 0; x < d.w; ++x) acc += (unsigned)abs((int)po[(long)y * so + x] - (int)pr[(long)y * sr + x]);
        cost = acc;
    } else {
        cost = 0;                                                            // had_c tiling of the block (8x8 tiles, else 4x4), positions one after the other
        for (int pos = 0; pos < 64; ++pos) {
            const uint8_t *pr = ref + d.b_off + (long)(pos >> 3) * sr + (pos & 7);
            unsigned total = 0;
            if (((d.w | d.h) & 7) == 0) {
                for (int t = 0; t < (d.w >> 3) * (d.h >> 3); ++t) {
                    const int ty = t / (d.w >> 3), x0 = (t - ty * (d.w >> 3)) << 3, y0 = ty << 3, x = lane & 7, y = lane >> 3;
                    const int v = (int)po[(long)(y0 + y) * so + x0 + x] - (int)pr[(long)(y0 + y) * sr + x0 + x];
                    total += (had8x8_abs_sum(v, lane) + 2) >> 2;
                }
            } else {
                const int tx = d.w >> 2, nt = tx * (d.h >> 2);
                for (int t0 = 0; t0 < nt; t0 += 4) {
                    const int t = t0 + (lane >> 4);
                    int v = 0;
                    if (t < nt) {
                        const int ty = t / tx, x0 = (t - ty * tx) << 2, y0 = ty << 2, x = lane & 3, y = (lane >> 2) & 3;
                        v = (int)po[(long)(y0 + y) * so + x0 + x] - (int)pr[(long)(y0 + y) * sr + x0 + x];
                    }
                    const unsigned sg = (had4x4_abs_sum16(v, lane) + 1) >> 1;
                    total += wave_sum((lane & 15) == 0 && t < nt ? sg : 0);
                }
            }
            if (lane == pos) cost = total;
        }
    }
    cost += (unsigned)mc[lane & 7] + (unsigned)mc[8 + (lane >> 3)];
    unsigned long long key = ((unsigned long long)cost << 6) | (unsigned)lane;      // first minimum in scan order (rows outside, columns inside)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(key, m, 64); key = o < key ? o : key; }
    if (lane == 0) { const int best = (int)(key & 63); out[2 * b] = (uint32_t)(key >> 6); out[2 * b + 1] = (uint32_t)(((best >> 3) << 16) | (best & 7)); }
}
// acEnergyPlane_c enc@0x4650e0: one wave per N x N block; offs == nullptr: the aligned blocks of a w x h plane in raster order
__global__ __launch_bounds__(256) void ac_energy_batch_kernel(const uint8_t *src, int stride, int log2, const int32_t *offs, int n, int bw, uint32_t *out)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n) return;
    const int N = 1 << log2;
    const long off = offs ? (long)offs[b] : ((long)(b / bw) * N) * stride + (long)(b % bw) * N;
    unsigned sum = 0, ssd = 0;
    for (int i = lane; i < N * N; i += 64) {
        const unsigned p = src[off + (long)(i >> log2) * stride + (i & (N - 1))];
        sum += p; ssd += p * p;
    }
    sum = wave_sum(sum); ssd = wave_sum(ssd);
    if (lane == 0) out[b] = ssd - ((sum * sum) >> (2 * log2));     // 32-bit wrap of sum^2 included, as in the reference
}

// postQuant enc@0x4ace80 -> signBitHidingHDQ enc@0x4aa150 on packed N x N blocks: one wave per block, the block staged in LDS, one lane per
// 4x4 coefficient group (the helpers of recon_dev.h that the fused reconstruction kernels use)
__global__ __launch_bounds__(64) void sign_hiding_batch_kernel(int log2n, int scan_idx, int16_t *lvl, const int16_t *coef, const int16_t *deltaU, int nblk)
{
    __shared__ short LV[32 * RP], DU[32 * RP], CF[32 * RP];
    __shared__ int lastcg, nz;
    const int n = 1 << log2n, lane = threadIdx.x, blk = blockIdx.x;
    if (blk >= nblk) return;
    const long base = (long)blk * n * n;
    if (lane == 0) { lastcg = 0; nz = 0; }
    __syncthreads();
    int cnt = 0;
    for (int i = lane; i < n * n; i += 64) {
        const int y = i / n, x = i % n;
        LV[y * RP + x] = lvl[base + i]; DU[y * RP + x] = deltaU[base + i]; CF[y * RP + x] = coef[base + i];
        cnt += lvl[base + i] != 0;
    }
    if (cnt) atomicAdd(&nz, cnt);
    __syncthreads();
    const int nsb = n >> 2;
    unsigned survey = 0;
    int cbase = 0, order = 0;
    const bool owner = lane < nsb * nsb && nz > 1;                    // postQuant calls it only for blocks with more than one level
    if (owner) {
        const int gx = lane % nsb, gy = lane / nsb;
        cbase = gy * 4 * RP + gx * 4;
        survey = sbh_survey(LV, cbase, scan_idx);
        order = sbh_group_order(scan_idx, nsb, gx, gy) + 1;
        if (survey >> 17) atomicMax(&lastcg, order);
    }
    __syncthreads();
    if (owner && survey) sbh_apply(LV, DU, CF, cbase, scan_idx, survey, lastcg == order);
    __syncthreads();
    for (int i = lane; i < n * n; i += 64) lvl[base + i] = LV[(i / n) * RP + (i % n)];
}

extern "C" {

int ks265_sad_batch(ks265_ctx *ctx, const uint8_t *a, int sa, const uint8_t *b, int sb, const ks265_blk *blks, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL((dist_batch_kernel<0, 1>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, sa, b, sb, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_sse_batch(ks265_ctx *ctx, const uint8_t *a, int sa, const uint8_t *b, int sb, const ks265_blk *blks, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL((dist_batch_kernel<1, 1>), dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, sa, b, sb, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_sad4_batch(ks265_ctx *ctx, const uint8_t *f, int sf, const uint8_t *r, int sr, const ks265_blk *blks, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL((dist_batch_kernel<0, 4>), dim3(n), dim3(256), 0, ctx->stream, f, sf, r, sr, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_sad3_batch(ks265_ctx *ctx, const uint8_t *f, int sf, const uint8_t *r, int sr, const ks265_blk3 *blks, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL((dist_batch_kernel<0, 3>), dim3((n * 3 + 3) / 4), dim3(256), 0, ctx->stream, f, sf, r, sr, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_sad4blk_8x8_batch(ks265_ctx *ctx, const uint8_t *a, int sa, const uint8_t *b, int sb, const ks265_blk *blks, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(sad4blk_kernel, dim3(n), dim3(256), 0, ctx->stream, a, sa, b, sb, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_had_batch(ks265_ctx *ctx, const uint8_t *a, int sa, const uint8_t *b, int sb, const ks265_blk *blks, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(had_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, a, sa, b, sb, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_residual_batch(ks265_ctx *ctx, const uint8_t *org, int so, const uint8_t *pred, int sp, const ks265_blk *blks, int n, int16_t *res)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(residual_batch_kernel, dim3(n), dim3(256), 0, ctx->stream, org, so, pred, sp, blks, n, res);
    LAUNCH_END(ctx);
}

static const int kIdxSize[5] = {4, 4, 8, 16, 32};

int ks265_fwd_transform_batch(ks265_ctx *ctx, int idx, const int16_t *src, int16_t *dst, int nblk)
{
    CHECK_CTX(ctx); if (idx < 0 || idx > 4) return KS265_NOTSUPPORTED; if (nblk <= 0) return KS265_OK;
    switch (kIdxSize[idx]) {
    case 4: hipLaunchKernelGGL(fwd_transform_kernel<4>, dim3(nblk), dim3(256), 0, ctx->stream, idx, src, dst, nblk); break;
    case 8: hipLaunchKernelGGL(fwd_transform_kernel<8>, dim3(nblk), dim3(256), 0, ctx->stream, idx, src, dst, nblk); break;
    case 16: hipLaunchKernelGGL(fwd_transform_kernel<16>, dim3(nblk), dim3(256), 0, ctx->stream, idx, src, dst, nblk); break;
    default: hipLaunchKernelGGL(fwd_transform_kernel<32>, dim3(nblk), dim3(256), 0, ctx->stream, idx, src, dst, nblk); break;
    }
    LAUNCH_END(ctx);
}
int ks265_inv_transform_batch(ks265_ctx *ctx, int idx, const int16_t *coef, const uint8_t *pred, uint8_t *dst, int nblk)
{
    CHECK_CTX(ctx); if (idx < 0 || idx > 4) return KS265_NOTSUPPORTED; if (nblk <= 0) return KS265_OK;
    switch (kIdxSize[idx]) {
    case 4: hipLaunchKernelGGL(inv_transform_kernel<4>, dim3(nblk), dim3(256), 0, ctx->stream, idx, coef, pred, dst, nblk); break;
    case 8: hipLaunchKernelGGL(inv_transform_kernel<8>, dim3(nblk), dim3(256), 0, ctx->stream, idx, coef, pred, dst, nblk); break;
    case 16: hipLaunchKernelGGL(inv_transform_kernel<16>, dim3(nblk), dim3(256), 0, ctx->stream, idx, coef, pred, dst, nblk); break;
    default: hipLaunchKernelGGL(inv_transform_kernel<32>, dim3(nblk), dim3(256), 0, ctx->stream, idx, coef, pred, dst, nblk); break;
    }
    LAUNCH_END(ctx);
}
int ks265_quant_batch(ks265_ctx *ctx, int n, const int16_t *coef, int16_t *lvl, int16_t *deltaU, int32_t *nz, int scale, int off, int qbits, int nblk)
{
    CHECK_CTX(ctx); if (n != 4 && n != 8 && n != 16 && n != 32) return KS265_NOTSUPPORTED; if (nblk <= 0) return KS265_OK;
    hipLaunchKernelGGL(quant_batch_kernel, dim3(nblk), dim3(256), 0, ctx->stream, n * n, coef, lvl, deltaU, nz, scale, off, qbits, nblk);
    LAUNCH_END(ctx);
}
int ks265_dequant_rect_batch(ks265_ctx *ctx, int n, int stride, const int16_t *lvl, int16_t *coef, int scale, int add, int shift, int lastX, int lastY, int nblk)
{
    CHECK_CTX(ctx); if (!lvl || !coef) return KS265_POINTER;
    if ((n != 4 && n != 8 && n != 16 && n != 32) || stride < n || lastX < 0 || lastY < 0 || lastX >= n || lastY >= n) return KS265_NOTSUPPORTED;
    if (nblk <= 0) return KS265_OK;
    const long total = (long)((lastX + 4) & ~3) * (lastY + 1) * nblk;
    hipLaunchKernelGGL(dequant_rect_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, lvl, coef, n, stride, scale, add, shift, lastX, lastY, nblk);
    LAUNCH_END(ctx);
}
int ks265_dequant_batch(ks265_ctx *ctx, int n, const int16_t *lvl, int16_t *coef, int scale, int add, int shift, int nblk)
{
    CHECK_CTX(ctx); if (n != 4 && n != 8 && n != 16 && n != 32) return KS265_NOTSUPPORTED; if (nblk <= 0) return KS265_OK;
    long total = (long)n * n * nblk;
    hipLaunchKernelGGL(dequant_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, lvl, coef, scale, add, shift, total);
    LAUNCH_END(ctx);
}
int ks265_edge_filter_luma_batch(ks265_ctx *ctx, uint8_t *plane, int stride, const ks265_edge *edges, int n)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(edge_luma_batch_kernel, dim3(n), dim3(64), 0, ctx->stream, plane, stride, edges, n);
    LAUNCH_END(ctx);
}
int ks265_edge_filter_chroma_batch(ks265_ctx *ctx, uint8_t *plane, int stride, const ks265_edge *edges, int n)
{
    CHECK_CTX(ctx); if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(edge_chroma_batch_kernel, dim3(n), dim3(64), 0, ctx->stream, plane, stride, edges, n);
    LAUNCH_END(ctx);
}

int ks265_interp_rect(ks265_ctx *ctx, int kind, void *dst, int ds, const void *src, int ss, int w, int h, int frac)
{
    CHECK_CTX(ctx);
    int chroma = kind & 1, vertical = (kind >> 1) & 1, io = (kind >> 2) & 3;
    if (frac < 1 || frac > (chroma ? 7 : 3)) return KS265_NOTSUPPORTED;
    if (io >= 2 && !vertical) return KS265_NOTSUPPORTED;   // the reference has no 16-bit-input horizontal filters
    dim3 g((w + 63) / 64, (h + 3) / 4), b(256);
#define IL(SRC, DST, NT, OUT) hipLaunchKernelGGL((interp_rect_kernel<SRC, DST, NT, OUT>), g, b, 0, ctx->stream, (DST *)dst, ds, (const SRC *)src, ss, w, h, frac, vertical)
    if (!chroma) {
        if (io == 0) IL(uint8_t, uint8_t, 8, 0); else if (io == 1) IL(uint8_t, int16_t, 8, 1);
        else if (io == 2) IL(int16_t, uint8_t, 8, 2); else IL(int16_t, int16_t, 8, 3);
    } else {
        if (io == 0) IL(uint8_t, uint8_t, 4, 0); else if (io == 1) IL(uint8_t, int16_t, 4, 1);
        else if (io == 2) IL(int16_t, uint8_t, 4, 2); else IL(int16_t, int16_t, 4, 3);
    }
#undef IL
    LAUNCH_END(ctx);
}

int ks265_sao_apply_bo_rect(ks265_ctx *ctx, const int8_t offsets[4], uint8_t *rec, int stride, int height, int width, int band)
{
    CHECK_CTX(ctx); if (!offsets) return KS265_POINTER;
    int cols = (width + 3) & ~3;
    int o[4];
    for (int k = 0; k < 4; ++k) o[k] = band + k < 32 ? offsets[k] : 0;
    hipLaunchKernelGGL(sao_bo_rect_kernel, dim3((cols + 63) / 64, (height + 3) / 4), dim3(256), 0, ctx->stream, rec, stride, height, cols, band,
                       o[0], o[1], o[2], o[3]);
    LAUNCH_END(ctx);
}
int ks265_sao_apply_eo_rect(ks265_ctx *ctx, int cls, const int8_t offsets[5], const uint8_t *src, uint8_t *dst, int stride, int height, int width)
{
    CHECK_CTX(ctx); if (!offsets) return KS265_POINTER; if (cls < 0 || cls > 3) return KS265_NOTSUPPORTED;
    static const int dx[4] = {1, 0, 1, -1}, dy[4] = {0, 1, 1, 1};
    hipLaunchKernelGGL(sao_eo_rect_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, ctx->stream, src, dst, stride, height, width,
                       dx[cls], dy[cls], (int)offsets[0], (int)offsets[1], (int)offsets[2], (int)offsets[3], (int)offsets[4]);
    LAUNCH_END(ctx);
}
int ks265_sao_stats_batch(ks265_ctx *ctx, const uint8_t *org, int os, const uint8_t *rec, int rs, const ks265_sao_rect *rects, int nrect,
                          int rowStep, int32_t *out)
{
    CHECK_CTX(ctx); if (nrect <= 0) return KS265_OK; if (rowStep < 1) return KS265_NOTSUPPORTED;
    hipLaunchKernelGGL(sao_stats_batch_kernel, dim3(nrect), dim3(256), 0, ctx->stream, org, os, rec, rs, rects, nrect, rowStep, out);
    LAUNCH_END(ctx);
}

int ks265_intra_pred_batch(ks265_ctx *ctx, const uint8_t *ref, uint8_t *dst, const ks265_intra_blk *blks, int n)
{
    CHECK_CTX(ctx); if (!ref || !dst || !blks) return KS265_POINTER; if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(intra_pred_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, ref, dst, blks, n);
    LAUNCH_END(ctx);
}
int ks265_intra_filter_ref_batch(ks265_ctx *ctx, const uint8_t *src, uint8_t *dst, const ks265_intra_ref *refs, int n)
{
    CHECK_CTX(ctx); if (!src || !dst || !refs) return KS265_POINTER; if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(intra_filter_ref_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, src, dst, refs, n);
    LAUNCH_END(ctx);
}

int ks265_downsample_rect(ks265_ctx *ctx, const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h)
{
    CHECK_CTX(ctx); if (!src || !dst) return KS265_POINTER; if (w <= 0 || h <= 0) return KS265_OK;
    hipLaunchKernelGGL(downsample_kernel, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, ctx->stream, src, srcStride, dst, dstStride, w, h);
    LAUNCH_END(ctx);
}
// src = pinned host memory (ks265_host_malloc), rows 8-byte aligned (srcStride and the pointer multiples of 8); dst = device memory, rows 4-byte aligned
int ks265_downsample_from_host(ks265_ctx *ctx, const uint8_t *src, int srcStride, uint8_t *dst, int dstStride, int w, int h)
{
    CHECK_CTX(ctx); if (!src || !dst) return KS265_POINTER; if (w <= 0 || h <= 0) return KS265_OK;
    if (((uintptr_t)src | (uintptr_t)srcStride) & 7 || ((uintptr_t)dst | (uintptr_t)dstStride) & 3) return KS265_NOTSUPPORTED;
    hipLaunchKernelGGL(downsample_host_kernel, dim3(64), dim3(256), 0, ctx->stream, src, srcStride, dst, dstStride, w, h);
    LAUNCH_END(ctx);
}
int ks265_weight_bi_sad_batch(ks265_ctx *ctx, const uint8_t *org, int so, const uint8_t *ref0, int s0, const uint8_t *ref1, int s1, const ks265_blk3 *blks, int n,
                              uint32_t *out)
{
    CHECK_CTX(ctx); if (!org || !ref0 || !ref1 || !blks || !out) return KS265_POINTER; if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(weight_bi_sad_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, org, so, ref0, s0, ref1, s1, blks, n, out);
    LAUNCH_END(ctx);
}
int ks265_bi_full_batch(ks265_ctx *ctx, int use_had, const uint8_t *org, int so, const uint8_t *ref, int sr, const ks265_blk *blks, const uint16_t *mvcost, int n,
                        uint32_t *out)
{
    CHECK_CTX(ctx); if (!org || !ref || !blks || !mvcost || !out) return KS265_POINTER; if (n <= 0) return KS265_OK;
    if (use_had) hipLaunchKernelGGL(bi_full_batch_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, org, so, ref, sr, blks, mvcost, n, out);
    else hipLaunchKernelGGL(bi_full_batch_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, org, so, ref, sr, blks, mvcost, n, out);
    LAUNCH_END(ctx);
}
int ks265_ac_energy_batch(ks265_ctx *ctx, const uint8_t *src, int stride, int log2, const int32_t *offs, int n, uint32_t *out)
{
    CHECK_CTX(ctx); if (!src || !offs || !out) return KS265_POINTER; if (log2 < 2 || log2 > 5) return KS265_NOTSUPPORTED; if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(ac_energy_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, src, stride, log2, offs, n, 1, out);
    LAUNCH_END(ctx);
}
int ks265_ac_energy_map(ks265_ctx *ctx, const uint8_t *plane, int stride, int w, int h, int log2, uint32_t *out)
{
    CHECK_CTX(ctx); if (!plane || !out) return KS265_POINTER; if (log2 < 2 || log2 > 5) return KS265_NOTSUPPORTED;
    const int bw = w >> log2, bh = h >> log2, n = bw * bh;
    if (n <= 0) return KS265_OK;
    hipLaunchKernelGGL(ac_energy_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, plane, stride, log2, (const int32_t *)nullptr, n, bw, out);
    LAUNCH_END(ctx);
}

int ks265_sign_hiding_batch(ks265_ctx *ctx, int n, int scan_idx, int16_t *lvl, const int16_t *coef, const int16_t *deltaU, int nblk)
{
    CHECK_CTX(ctx); if (!lvl || !coef || !deltaU) return KS265_POINTER;
    if ((n != 4 && n != 8 && n != 16 && n != 32) || scan_idx < 0 || scan_idx > 2 || (scan_idx && n > 8)) return KS265_NOTSUPPORTED;
    if (nblk <= 0) return KS265_OK;
    hipLaunchKernelGGL(sign_hiding_batch_kernel, dim3(nblk), dim3(64), 0, ctx->stream, n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5, scan_idx, lvl, coef, deltaU, nblk);
    LAUNCH_END(ctx);
}

}  // extern "C"
