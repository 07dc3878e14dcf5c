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

static int pad_picture_on(ks265_ctx *cx, ks265_frame *f, ks265_pic pic)
{
    PadArgs a;
    a.pl[0] = PadPlane{pic.y, f->g.sy, f->g.W, f->g.H, KS_PAD_Y};
    a.pl[1] = PadPlane{pic.u, f->g.sc, f->g.W / 2, f->g.H / 2, KS_PAD_C};
    a.pl[2] = PadPlane{pic.v, f->g.sc, f->g.W / 2, f->g.H / 2, KS_PAD_C};
    const int items = 2 * KS_PAD_Y * ((f->g.W + 2 * KS_PAD_Y) / 4) + f->g.H * (2 * KS_PAD_Y / 4);      // luma has the most
    hipLaunchKernelGGL(pad_picture_kernel, dim3((items + 255) / 256, 3), dim3(256), 0, cx->stream, a);
    return ks265_check_launch(cx);
}
extern "C" int ks265_pad_picture(ks265_frame *f, ks265_pic pic)
{
    KS_FRAME_CHECK(f);
    return pad_picture_on(f->ctx, f, pic);
}

// ------------------------------------------------------------------ I420 <-> padded picture
// the three planes in one launch: blockIdx.z = plane (chroma planes use the upper-left quarter of the grid).  A thread moves 8 bytes of four consecutive rows when every
// row start is 8-byte aligned (picture widths that are multiples of 16: all the sizes the encoder is run at), else a dword of one row
struct CopyPlanes { uint8_t *dst[3]; const uint8_t *src[3]; long ds[3], ss[3]; int w, h; };
__global__ __launch_bounds__(256) void copy_planes_kernel(CopyPlanes a)
{
    const int pl = blockIdx.z, w = pl ? a.w / 2 : a.w, h = pl ? a.h / 2 : a.h;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= w || y >= h) return;
    *(unsigned *)(a.dst[pl] + y * a.ds[pl] + x4) = *(const unsigned *)(a.src[pl] + y * a.ss[pl] + x4);
}
__global__ __launch_bounds__(256) void copy_planes8_kernel(CopyPlanes a)
{
    const int pl = blockIdx.z, w = pl ? a.w / 2 : a.w, h = pl ? a.h / 2 : a.h;
    const int x8 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8, y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
    if (x8 >= w || y0 >= h) return;
    uint2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = y0 + r < h ? *(const uint2 *)(a.src[pl] + (long)(y0 + r) * a.ss[pl] + x8) : make_uint2(0u, 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) if (y0 + r < h) *(uint2 *)(a.dst[pl] + (long)(y0 + r) * a.ds[pl] + x8) = v[r];
}
static void launch_copy3(ks265_ctx *cx, const CopyPlanes &a)
{
    bool al = (a.w & 15) == 0;
    for (int p = 0; p < 3 && al; ++p) al = ((uintptr_t)a.dst[p] & 7) == 0 && ((uintptr_t)a.src[p] & 7) == 0 && (a.ds[p] & 7) == 0 && (a.ss[p] & 7) == 0;
    if (al) hipLaunchKernelGGL(copy_planes8_kernel, dim3((a.w / 8 + 63) / 64, (a.h + 15) / 16, 3), dim3(256), 0, cx->stream, a);
    else hipLaunchKernelGGL(copy_planes_kernel, dim3((a.w / 4 + 63) / 64, (a.h + 3) / 4, 3), dim3(256), 0, cx->stream, a);
}

/* the _on forms enqueue on ANOTHER context's stream of the same device (a host that prepares the next source picture, or drains the last one's records, beside the
 * frame's own stream; the host orders the streams with events) */
extern "C" int ks265_load_i420_on(ks265_ctx *cx, ks265_frame *f, const uint8_t *i420, ks265_pic dst)
{
    KS_FRAME_CHECK(f);
    if (!i420 || !cx) return KS265_POINTER;
    ks_use_device(cx);
    int W = f->g.W, H = f->g.H;
    CopyPlanes a;
    a.dst[0] = dst.y + f->g.org_y; a.dst[1] = dst.u + f->g.org_c; a.dst[2] = dst.v + f->g.org_c; a.ds[0] = f->g.sy; a.ds[1] = a.ds[2] = f->g.sc;
    a.src[0] = i420; a.src[1] = i420 + (long)W * H; a.src[2] = i420 + (long)W * H * 5 / 4; a.ss[0] = W; a.ss[1] = a.ss[2] = W / 2; a.w = W; a.h = H;
    launch_copy3(cx, a);
    return pad_picture_on(cx, f, dst);
}
extern "C" int ks265_load_i420(ks265_frame *f, const uint8_t *i420, ks265_pic dst) { return f ? ks265_load_i420_on(f->ctx, f, i420, dst) : KS265_POINTER; }

extern "C" int ks265_store_i420(ks265_frame *f, ks265_pic src, uint8_t *i420)
{
    KS_FRAME_CHECK(f);
    if (!i420) return KS265_POINTER;
    int W = f->g.W, H = f->g.H;
    CopyPlanes a;
    a.src[0] = src.y + f->g.org_y; a.src[1] = src.u + f->g.org_c; a.src[2] = src.v + f->g.org_c; a.ss[0] = f->g.sy; a.ss[1] = a.ss[2] = f->g.sc;
    a.dst[0] = i420; a.dst[1] = i420 + (long)W * H; a.dst[2] = i420 + (long)W * H * 5 / 4; a.ds[0] = W; a.ds[1] = a.ds[2] = W / 2; a.w = W; a.h = H;
    launch_copy3(f->ctx, a);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ picture SSE (PSNR): sse3[plane] += sum of squared differences
// one launch for the three planes: blockIdx.y = plane, a wave = one row at a time (16 bytes per lane and load where the rows allow), rows dealt round-robin to the work-groups
__global__ __launch_bounds__(256) void sse_picture_kernel(KsGeom g, const uint8_t *ay, const uint8_t *au, const uint8_t *av, const uint8_t *by, const uint8_t *bu, const uint8_t *bv,
                                                          unsigned long long *acc /* [0..2] running sums, [3] work-groups done; all zero between calls */, unsigned long long *out)
{
    const int pl = blockIdx.y;
    const int w = pl ? g.W / 2 : g.W, h = pl ? g.H / 2 : g.H;
    const long stride = pl ? g.sc : g.sy, org = pl ? g.org_c : g.org_y;
    const uint8_t *a = (pl == 0 ? ay : pl == 1 ? au : av) + org, *b = (pl == 0 ? by : pl == 1 ? bu : bv) + org;
    const int lane = threadIdx.x & 63;
    unsigned s = 0;
    auto sq4 = [&](unsigned va, unsigned vb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int d = (int)((va >> (8 * i)) & 255) - (int)((vb >> (8 * i)) & 255); s += (unsigned)(d * d); }
    };
    // few work-groups, each over many rows: every work-group ends in two atomics on the same four words, and 1620 of those WERE the kernel's duration (33 us; 86 x 3: 9 us)
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < h; y += gridDim.x * 4) {
        if ((w & 15) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)stride) & 15) == 0) {
            // all loads of a pass are issued before the first sum (a pass = up to 4 x 16 bytes per lane and picture: 4 KB of the row per wave): the kernel is a latency chain otherwise
            for (int x0 = 0; x0 < w; x0 += 4096) {
                uint4 va[4], vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = x0 + k * 1024 + lane * 16;
                    va[k] = make_uint4(0, 0, 0, 0); vb[k] = make_uint4(0, 0, 0, 0);
                    if (x < w) { va[k] = *(const uint4 *)(a + y * stride + x); vb[k] = *(const uint4 *)(b + y * stride + x); }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { sq4(va[k].x, vb[k].x); sq4(va[k].y, vb[k].y); sq4(va[k].z, vb[k].z); sq4(va[k].w, vb[k].w); }
            }
        } else if ((w & 7) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)stride) & 7) == 0) {
            for (int x0 = 0; x0 < w; x0 += 2048) {
                uint2 va[4], vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = x0 + k * 512 + lane * 8;
                    va[k] = make_uint2(0, 0); vb[k] = make_uint2(0, 0);
                    if (x < w) { va[k] = *(const uint2 *)(a + y * stride + x); vb[k] = *(const uint2 *)(b + y * stride + x); }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { sq4(va[k].x, vb[k].x); sq4(va[k].y, vb[k].y); }
            }
        } else
            for (int x = lane * 4; x < w; x += 256) sq4(*(const unsigned *)(a + y * stride + x), *(const unsigned *)(b + y * stride + x));
    }
    // a lane: <= 8 rows x 64 samples x 255^2 < 2^25 at 2160p (rows / (4 x 86) per wave); 8K pictures: < 2^27 - 64 lanes of that pass 2^32, so the wave's sum
    // is taken in two 16-bit halves (each half's sum < 2^22) and put together in 64 bits
    const unsigned long long s64 = (unsigned long long)wave_sum(s & 0xffffu) + ((unsigned long long)wave_sum(s >> 16) << 16);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s64;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
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

extern "C" int ks265_sse_picture_on(ks265_ctx *cx, ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *sse3)
{
    KS_FRAME_CHECK(f);
    if (!sse3 || !cx) return KS265_POINTER;
    ks_use_device(cx);
    hipLaunchKernelGGL(sse_picture_kernel, dim3(86, 3), dim3(256), 0, cx->stream, f->g, a.y, a.u, a.v, b.y, b.u, b.v, f->sse_acc, (unsigned long long *)sse3);
    return ks265_check_launch(cx);
}
extern "C" int ks265_sse_picture(ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *sse3) { return f ? ks265_sse_picture_on(f->ctx, f, a, b, sse3) : KS265_POINTER; }
