// frame_util.hip — picture plumbing: geometry, border padding (expandPicture_c enc@0x4a6ae0), I420 load/store,
// the fractional-sample planes (stage A0) and picture SSE.
#include "frame_common.h"

using namespace ks265;

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

extern "C" int ks265_frame_geometry(const ks265_frame_cfg *cfg, ks265_frame_geom *g)
{
    if (!cfg || !g) return KS265_POINTER;
    if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7) || cfg->me_range > 64 || cfg->me_range < 1 ||
        cfg->qp < 0 || cfg->qp > 51) return KS265_NOTSUPPORTED;
    g->pad_y = KS_PAD_Y; g->pad_c = KS_PAD_C;
    g->stride_y = align_up(cfg->width + 2 * KS_PAD_Y, 128);
    g->stride_c = align_up(cfg->width / 2 + 2 * KS_PAD_C, 64);
    g->rows_y = cfg->height + 2 * KS_PAD_Y;
    g->rows_c = cfg->height / 2 + 2 * KS_PAD_C;
    g->bytes_y = (int64_t)g->stride_y * (g->rows_y + 1);   /* one slack row: window / tile loads of the last padded row may run up to 64 bytes past it */
    g->bytes_c = (int64_t)g->stride_c * (g->rows_c + 1);
    g->ctu_cols = (cfg->width + 63) / 64;
    g->ctu_rows = (cfg->height + 63) / 64;
    g->pu_per_ctu = 85;
    g->bytes_pu = (int64_t)g->ctu_cols * g->ctu_rows * 85 * (int64_t)sizeof(ks265_pu);
    g->bytes_cu8 = (int64_t)(cfg->width / 8) * (cfg->height / 8) * (int64_t)sizeof(ks265_cu8);
    g->bytes_sao = (int64_t)g->ctu_cols * g->ctu_rows * 3 * (int64_t)sizeof(ks265_sao_param);
    return KS265_OK;
}

// ------------------------------------------------------------------ border padding (expandPicture_c enc@0x4a6ae0): one thread per 4 border
// bytes, ONLY border dwords are enumerated (top / bottom bands, then the left / right strips of the picture rows), the three
// planes of a picture in one launch (blockIdx.y = plane)
struct PadPlane { uint8_t *p; int stride, w, h, pad; };
struct PadArgs { PadPlane pl[3]; };

__global__ __launch_bounds__(256) void pad_picture_kernel(PadArgs a)
{
    const PadPlane q = a.pl[blockIdx.y];
    const int fw4 = (q.w + 2 * q.pad) / 4, side4 = 2 * q.pad / 4;
    const int nband = 2 * q.pad * fw4, nside = q.h * side4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nband + nside) return;
    int y, x4;
    if (i < nband) {
        const int r = i / fw4;
        y = r < q.pad ? r : q.h + r;
        x4 = (i - r * fw4) * 4;
    } else {
        const int j = i - nband, r = j / side4, k = (j - r * side4) * 4;
        y = q.pad + r;
        x4 = k < q.pad ? k : q.w + k;
    }
    const int sy = min(max(y - q.pad, 0), q.h - 1), xx = x4 - q.pad;
    const uint8_t *srow = q.p + (long)(sy + q.pad) * q.stride + q.pad;
    unsigned v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) v |= (unsigned)srow[min(max(xx + b, 0), q.w - 1)] << (8 * b);
    *(unsigned *)(q.p + (long)y * q.stride + x4) = v;
}

extern "C" int ks265_pad_picture(ks265_frame *f, ks265_pic pic)
{
    KS_FRAME_CHECK(f);
    PadArgs a;
    a.pl[0] = PadPlane{pic.y, f->g.sy, f->g.W, f->g.H, KS_PAD_Y};
    a.pl[1] = PadPlane{pic.u, f->g.sc, f->g.W / 2, f->g.H / 2, KS_PAD_C};
    a.pl[2] = PadPlane{pic.v, f->g.sc, f->g.W / 2, f->g.H / 2, KS_PAD_C};
    const int items = 2 * KS_PAD_Y * ((f->g.W + 2 * KS_PAD_Y) / 4) + f->g.H * (2 * KS_PAD_Y / 4);      // luma has the most
    hipLaunchKernelGGL(pad_picture_kernel, dim3((items + 255) / 256, 3), dim3(256), 0, f->ctx->stream, a);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ I420 <-> padded picture (dword per thread)
// the three planes in one launch: blockIdx.z = plane (chroma planes use the upper-left quarter of the grid)
struct CopyPlanes { uint8_t *dst[3]; const uint8_t *src[3]; long ds[3], ss[3]; int w, h; };
__global__ __launch_bounds__(256) void copy_planes_kernel(CopyPlanes a)
{
    const int pl = blockIdx.z, w = pl ? a.w / 2 : a.w, h = pl ? a.h / 2 : a.h;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= w || y >= h) return;
    *(unsigned *)(a.dst[pl] + y * a.ds[pl] + x4) = *(const unsigned *)(a.src[pl] + y * a.ss[pl] + x4);
}
static void launch_copy3(ks265_frame *f, const CopyPlanes &a)
{
    hipLaunchKernelGGL(copy_planes_kernel, dim3((a.w / 4 + 63) / 64, (a.h + 3) / 4, 3), dim3(256), 0, f->ctx->stream, a);
}

extern "C" int ks265_load_i420(ks265_frame *f, const uint8_t *i420, ks265_pic dst)
{
    KS_FRAME_CHECK(f);
    if (!i420) return KS265_POINTER;
    int W = f->g.W, H = f->g.H;
    CopyPlanes a;
    a.dst[0] = dst.y + f->g.org_y; a.dst[1] = dst.u + f->g.org_c; a.dst[2] = dst.v + f->g.org_c; a.ds[0] = f->g.sy; a.ds[1] = a.ds[2] = f->g.sc;
    a.src[0] = i420; a.src[1] = i420 + (long)W * H; a.src[2] = i420 + (long)W * H * 5 / 4; a.ss[0] = W; a.ss[1] = a.ss[2] = W / 2; a.w = W; a.h = H;
    launch_copy3(f, a);
    return ks265_pad_picture(f, dst);
}

extern "C" int ks265_store_i420(ks265_frame *f, ks265_pic src, uint8_t *i420)
{
    KS_FRAME_CHECK(f);
    if (!i420) return KS265_POINTER;
    int W = f->g.W, H = f->g.H;
    CopyPlanes a;
    a.src[0] = src.y + f->g.org_y; a.src[1] = src.u + f->g.org_c; a.src[2] = src.v + f->g.org_c; a.ss[0] = f->g.sy; a.ss[1] = a.ss[2] = f->g.sc;
    a.dst[0] = i420; a.dst[1] = i420 + (long)W * H; a.dst[2] = i420 + (long)W * H * 5 / 4; a.ds[0] = W; a.ds[1] = a.ds[2] = W / 2; a.w = W; a.h = H;
    launch_copy3(f, a);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ Stage A0: the 15 fractional planes
// One workgroup = one 64x16 output tile; the (64+8) x (16+7) source tile is staged in LDS, the three horizontal
// 8-tap intermediates (raw 16-bit sums, interpLumaHor8to16_c enc@0x40eb80) are built once in LDS and every thread
// then produces 4 adjacent samples of all 15 planes (one dword store per plane): fy = 0 planes round the
// intermediate ((s+32)>>6 == interpLumaHor8to8_c enc@0x40e4f0), fx = 0 planes filter the source vertically
// (interpLumaVer8to8_c enc@0x40f0c0), the nine 2-D planes filter the intermediates (interpLumaVer16to8_c enc@0x4100b0).
// hipcc (ROCm 7.2) folds clip8(x >> s) pairs into gfx950's v_ashr_pk_u8_i32 and then ORs the packed pair with
// `v_lshl_or_b32` assuming bits 31:16 of its result are zero; on MI355X they are not (measured: the upper two
// samples of every packed dword came back OR-contaminated).  An empty asm on the shifted value keeps the
// shift and the clamp apart so the instruction is never selected.
__device__ __forceinline__ int no_pk(int v) { asm volatile("" : "+v"(v)); return v; }

#define PT_W 64
#define PT_H 16
#define PT_SW 76      // source tile width  (x-4 .. x+64+7; the horizontal pass reads 16 bytes from column x)
#define PT_SH 23      // source tile height (y-3 .. y+16+3)
__global__ __launch_bounds__(256) void ref_planes_kernel(KsGeom g, const uint8_t *ref, uint8_t *planes)
{
    __shared__ __attribute__((aligned(16))) uint8_t S[PT_SH][PT_SW];   // dword accesses: the array must be 4-byte aligned in LDS
    __shared__ __attribute__((aligned(16))) short Hm[3][PT_SH][PT_W];
    const int tid = threadIdx.x;
    const int ntx = (g.W + 2 * KS_PLANE_MARGIN + PT_W - 1) / PT_W, nty = (g.H + 2 * KS_PLANE_MARGIN + PT_H - 1) / PT_H;
    const int tile = ks_xcd_swizzle(blockIdx.x, ntx * nty);       // XCD-aware: neighbouring tiles (shared source rows / lines) in one L2
    const int x0 = -KS_PLANE_MARGIN + (tile % ntx) * PT_W, y0 = -KS_PLANE_MARGIN + (tile / ntx) * PT_H;
    const uint8_t *R = ks_org_y(g, ref);
    // source tile: dword loads (x0 - 4 is dword aligned: origin, margin and tile width are multiples of 4)
    for (int i = tid; i < PT_SH * (PT_SW / 4); i += 256) {
        int r = i / (PT_SW / 4), c = i - r * (PT_SW / 4);
        int yy = min(y0 - 3 + r, g.H + KS_PAD_Y - 1);                 // rows past the border only feed discarded outputs
        *(unsigned *)&S[r][c * 4] = *(const unsigned *)(R + (long)yy * g.sy + x0 - 4 + c * 4);
    }
    __syncthreads();
    // horizontal intermediates, 4 samples per work item: sample x needs S bytes x+1 .. x+8 (S column of sample x is x + 4).
    // v_dot4_i32_i8 on (pixel - 128): sum c*(p-128) = sum c*p - 128*64, so 8192 is added back.
    for (int i = tid; i < PT_SH * (PT_W / 4); i += 256) {
        const int r = i / (PT_W / 4), x4 = (i - r * (PT_W / 4)) * 4;
        const unsigned *sp = (const unsigned *)&S[r][x4];
        const unsigned d0 = sp[0] ^ 0x80808080u, d1 = sp[1] ^ 0x80808080u, d2 = sp[2] ^ 0x80808080u, d3 = sp[3] ^ 0x80808080u;
        short o[3][4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            // bytes x4+px+1 .. +4 and +5 .. +8
            const unsigned lo = px == 3 ? d1 : align_bytes(d1, d0, px + 1), hi = px == 3 ? d2 : align_bytes(d2, d1, px + 1);
#pragma unroll
            for (int fx = 1; fx < 4; ++fx) {
                const signed char *c = kLumaTaps[fx];
                const int tl = (int)((unsigned)(unsigned char)c[0] | ((unsigned)(unsigned char)c[1] << 8) | ((unsigned)(unsigned char)c[2] << 16) | ((unsigned)(unsigned char)c[3] << 24));
                const int th = (int)((unsigned)(unsigned char)c[4] | ((unsigned)(unsigned char)c[5] << 8) | ((unsigned)(unsigned char)c[6] << 16) | ((unsigned)(unsigned char)c[7] << 24));
                int sum = __builtin_amdgcn_sdot4((int)lo, tl, 8192, false);
                sum = __builtin_amdgcn_sdot4((int)hi, th, sum, false);
                o[fx - 1][px] = (short)sum;
            }
        }
        (void)d3;
#pragma unroll
        for (int fx = 0; fx < 3; ++fx) *(uint2 *)&Hm[fx][r][x4] = *(const uint2 *)o[fx];
    }
    __syncthreads();
    const int tx = (tid & 15) * 4, ty = tid >> 4;      // 4 samples at (x0 + tx .. +3, y0 + ty)
    const int X = x0 + tx, Y = y0 + ty;
    if (X >= g.W + KS_PLANE_MARGIN || Y >= g.H + KS_PLANE_MARGIN) return;
    // the eight tile rows ty .. ty+7 feed every vertical filter of this thread: one dword (4 source bytes) and three
    // 8-byte reads (4 intermediates each) per row, shared by the three vertical fractions
    int acc[3][4][4];                                   // [fy-1][plane column 0..3 (0 = source, 1..3 = Hm)][pixel]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = 0;
    unsigned out[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const unsigned sv = *(const unsigned *)&S[ty + t][tx + 4];
        short hv[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k) *(uint2 *)hv[k] = *(const uint2 *)&Hm[k][ty + t][tx];
        if (t == 3) {                                   // the row of the output sample itself: fy = 0 planes
            out[0] = sv;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                unsigned v = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) v |= (unsigned)clip8(no_pk(((int)hv[k][i] + 32) >> 6)) << (8 * i);
                out[1 + k] = v;
            }
        }
#pragma unroll
        for (int fy = 1; fy < 4; ++fy) {
            const int c = kLumaTaps[fy][t];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[fy - 1][0][i] += c * (int)((sv >> (8 * i)) & 255);
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[fy - 1][1 + k][i] += c * (int)hv[k][i];
            }
        }
    }
#pragma unroll
    for (int fy = 1; fy < 4; ++fy) {
        unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v0 |= (unsigned)clip8(no_pk((acc[fy - 1][0][i] + 32) >> 6)) << (8 * i);
            v1 |= (unsigned)clip8(no_pk((acc[fy - 1][1][i] + 2048) >> 12)) << (8 * i);
            v2 |= (unsigned)clip8(no_pk((acc[fy - 1][2][i] + 2048) >> 12)) << (8 * i);
            v3 |= (unsigned)clip8(no_pk((acc[fy - 1][3][i] + 2048) >> 12)) << (8 * i);
        }
        out[fy * 4 + 0] = v0; out[fy * 4 + 1] = v1; out[fy * 4 + 2] = v2; out[fy * 4 + 3] = v3;
    }
    long off = g.org_y + (long)Y * g.sy + X;
#pragma unroll
    for (int p = 0; p < 16; ++p) *(unsigned *)(planes + p * g.bytes_y + off) = out[p];
}

extern "C" int ks265_ref_planes(ks265_frame *f, ks265_pic ref, uint8_t *planes)
{
    KS_FRAME_CHECK(f);
    if (!ref.y || !planes) return KS265_POINTER;
    dim3 grid(((f->g.W + 2 * KS_PLANE_MARGIN + PT_W - 1) / PT_W) * ((f->g.H + 2 * KS_PLANE_MARGIN + PT_H - 1) / PT_H));
    hipLaunchKernelGGL(ref_planes_kernel, grid, dim3(256), 0, f->ctx->stream, f->g, ref.y, planes);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ picture SSE (PSNR): sse3[plane] += sum of squared differences
// one launch for the three planes: blockIdx.y = plane, a work-group = 8 rows, a thread = dwords of one row (stride 32 dwords)
__global__ __launch_bounds__(256) void sse_picture_kernel(KsGeom g, const uint8_t *ay, const uint8_t *au, const uint8_t *av, const uint8_t *by, const uint8_t *bu, const uint8_t *bv,
                                                          unsigned long long *acc /* [0..2] running sums, [3] work-groups done; all zero between calls */, unsigned long long *out)
{
    const int pl = blockIdx.y;
    const int w = pl ? g.W / 2 : g.W, h = pl ? g.H / 2 : g.H;
    const long stride = pl ? g.sc : g.sy, org = pl ? g.org_c : g.org_y;
    const uint8_t *a = (pl == 0 ? ay : pl == 1 ? au : av) + org, *b = (pl == 0 ? by : pl == 1 ? bu : bv) + org;
    const int y = blockIdx.x * 8 + (threadIdx.x >> 5);
    unsigned s = 0;
    if (y < h)
        for (int x4 = (threadIdx.x & 31) * 4; x4 < w; x4 += 128) {
            const unsigned va = *(const unsigned *)(a + y * stride + x4), vb = *(const unsigned *)(b + y * stride + x4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int d = (int)((va >> (8 * i)) & 255) - (int)((vb >> (8 * i)) & 255); s += (unsigned)(d * d); }
        }
    s = wave_sum(s);                                                 // <= 8 rows x 4096 samples x 255^2 per work-group: fits 32 bits up to 8K pictures
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (unsigned long long)part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(acc + pl, t);
        __threadfence();
        // the last work-group to finish hands the sums out and leaves the accumulators zeroed for the next call (no memset launch per picture)
        if (atomicAdd(acc + 3, 1ull) == (unsigned long long)gridDim.x * gridDim.y - 1ull) {
            __threadfence();
            for (int i = 0; i < 3; ++i) out[i] = atomicExch(acc + i, 0ull);
            atomicExch(acc + 3, 0ull);
        }
    }
}

extern "C" int ks265_sse_picture(ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *sse3)
{
    KS_FRAME_CHECK(f);
    if (!sse3) return KS265_POINTER;
    hipLaunchKernelGGL(sse_picture_kernel, dim3((unsigned)((f->g.H + 7) / 8), 3), dim3(256), 0, f->ctx->stream, f->g, a.y, a.u, a.v, b.y, b.u, b.v, f->sse_acc, (unsigned long long *)sse3);
    return ks265_check_launch(f->ctx);
}
