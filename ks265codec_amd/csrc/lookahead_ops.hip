// lookahead_ops.hip - two of the reference's lookahead decisions as device operators (round 4):
//   ks265_frame_adapt_quant  = calcFrameAdaptQuant enc@0x4653c0 (mode 1): per 16x16 block the QP offset (double) and the inverse qscale factor (fix 8)
//   ks265_cutree_propagate   = cuTreePropagate enc@0x47d460: one propagation step from a picture into its two references
// Both are restated from the disassembly in oracle/ks265_lookahead_ref.c and pinned there on calls recorded inside the reference binary
// (tests/golden/lookahead_ref.npz); the tests compare these kernels with that oracle bit for bit - the doubles included (IEEE add / mul / div in the
// reference's order: the mean is a sequential sum).
#include "ks265_dev.h"
#include "ks265_internal.h"
#include <cmath>
#include <mutex>

using namespace ks265;

// _log2 enc@0x4c3c20 / qy265_exp2fix8 enc@0x4c3c50: their tables are closed forms (checked against the file when the oracle was written): filled on the host once
struct LaTables { double log2_lut[128]; unsigned char exp2_lut[64]; };
__constant__ LaTables kLa;
static int la_tables_upload(int device)
{
    // per device (a __constant__ symbol lives on each of them; a handle's GOP lanes may sit on several GPUs and call from several threads)
    static std::mutex mu;
    static bool done[64];
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) return 1;
    if (done[device]) return 0;
    LaTables t;
    for (int i = 0; i < 128; ++i) t.log2_lut[i] = std::round(std::log2((128.0 + i) / 128.0) * 1e5) / 1e5;
    for (int i = 0; i < 64; ++i) t.exp2_lut[i] = (unsigned char)std::lround((std::pow(2.0, i / 64.0) - 1.0) * 256.0);
    if (hipMemcpyToSymbol(HIP_SYMBOL(kLa), &t, sizeof t) != hipSuccess) return 1;      // (the caller has made `device` current)
    done[device] = true;
    return 0;
}

// one wave per 16x16 block: lane = 4 luma samples + one sample of each chroma plane; AC energy = ssd - (sum^2 >> 2 log2 N) in 32-bit unsigned (acEnergyPlane_c enc@0x4650e0)
__global__ __launch_bounds__(256) void aq_energy_kernel(const uint8_t *Y, int sy, const uint8_t *U, const uint8_t *V, int sc, int nx, int ny, double *val)
{
    const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= nx * ny) return;
    const int bx = blk % nx, by = blk / nx;
    const unsigned w = *(const unsigned *)(Y + (long)(by * 16 + (lane >> 2)) * sy + bx * 16 + (lane & 3) * 4);
    unsigned s0 = (w & 255u) + ((w >> 8) & 255u) + ((w >> 16) & 255u) + (w >> 24);
    unsigned q0 = (w & 255u) * (w & 255u) + ((w >> 8) & 255u) * ((w >> 8) & 255u) + ((w >> 16) & 255u) * ((w >> 16) & 255u) + (w >> 24) * (w >> 24);
    const unsigned u = U[(long)(by * 8 + (lane >> 3)) * sc + bx * 8 + (lane & 7)], v = V[(long)(by * 8 + (lane >> 3)) * sc + bx * 8 + (lane & 7)];
    unsigned s1 = u, q1 = u * u, s2 = v, q2 = v * v;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s0 += (unsigned)__shfl_xor((int)s0, m, 64); q0 += (unsigned)__shfl_xor((int)q0, m, 64);
        s1 += (unsigned)__shfl_xor((int)s1, m, 64); q1 += (unsigned)__shfl_xor((int)q1, m, 64);
        s2 += (unsigned)__shfl_xor((int)s2, m, 64); q2 += (unsigned)__shfl_xor((int)q2, m, 64);
    }
    if (lane == 0) {
        const unsigned e = (q0 - ((s0 * s0) >> 8)) + (q1 - ((s1 * s1) >> 6)) + (q2 - ((s2 * s2) >> 6)) + 2u;
        const int lz = __clz(e);
        const double l = kLa.log2_lut[((e << lz) >> 24) & 127u] + (double)(31 - lz);
        val[blk] = l * l;
    }
}
// the mean in the reference's order - a SEQUENTIAL sum of doubles (every addition rounds: no other order gives the same bits) - then the two scalars every block needs.
// One work-group: 256 values at a time come in coalesced through LDS, thread 0 adds them in order (the loads run ahead of the one dependent add chain: 32 400 blocks of a
// 2160p picture in ~0.15 ms; the first version read global memory from the one thread: 1.7 ms)
__global__ __launch_bounds__(256) void aq_mean_kernel(const double *val, int n, int count, double strength, double *scal)
{
    __shared__ double buf[2][256];
    double sum = 0.0;
    const int t = threadIdx.x;
    if (t < n) buf[0][t] = val[t];
    __syncthreads();
    for (int base = 0, ph = 0; base < n; base += 256, ph ^= 1) {
        if (base + 256 + t < n) buf[ph ^ 1][t] = val[base + 256 + t];       // the next chunk lands while thread 0 adds this one
        if (t == 0) {
            const int m = min(256, n - base);
#pragma unroll 8
            for (int i = 0; i < m; ++i) sum += buf[ph][i];
        }
        __syncthreads();
    }
    if (t == 0) { const double avg = sum / (double)count; scal[0] = avg; scal[1] = strength * avg / 6000.0; }
}
__global__ __launch_bounds__(256) void aq_offset_kernel(double *val, int n, const double *scal, uint16_t *inv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double off = (val[i] - scal[0]) * scal[1];
    val[i] = off;
    const int k = (int)(off * (double)(-64.0f / 6.0f) + 512.5);
    inv[i] = (uint16_t)(k < 0 ? 0 : k > 1023 ? 0xffff : (((int)kLa.exp2_lut[k & 63] + 256) << (k >> 6)) >> 8);
}

// cuTreePropagate: thread = block.  What a block hands on is split over the (up to) four blocks its vector points at, per list; the reference adds with saturation at
// 0xffff in raster order - saturating additions of non-negative shares commute, so the shares are summed in 64-bit accumulators (atomics) and clipped once.
__global__ __launch_bounds__(256) void cutree_scatter_kernel(int lg, int nx, int ny, const uint16_t *intra, const uint16_t *invq, const uint16_t *own, const uint16_t *inter,
                                                             const uint8_t *bits, const int32_t *mv0, const int32_t *mv1, unsigned long long *acc0, unsigned long long *acc1)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * ny) return;
    const int bx = idx % nx, by = idx / nx;
    const int sh = lg + 2, unit = 1 << sh, sh2 = 2 * sh, rnd = 1 << (sh2 - 1);
    if (intra[idx] == 0) return;     // never stored by the reference (calcFrameCost keeps min(cost + 9, 0xffff)); its idiv would trap - skipped here and in the oracle alike
    const long long have = (long long)(((int)((unsigned)invq[idx] * (unsigned)intra[idx] + 128u) >> 8) + (int)own[idx]);
    const int amt = (int)(have * (long long)((int)intra[idx] - (int)inter[idx]) / (long long)intra[idx]);
    if (amt <= 0) return;
    const int lists = (bits[idx >> 2] >> ((idx & 3) * 2)) & 3;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        if (!((lists >> l) & 1)) continue;
        const int a = lists == 3 ? (amt * 32 + 32) >> 6 : amt;
        const int mv = l ? mv1[idx] : mv0[idx];
        unsigned long long *A = l ? acc1 : acc0;
        if (mv == 0) { atomicAdd(&A[idx], (unsigned long long)a); continue; }
        const int mvx = (int)(short)(mv & 0xffff), mvy = mv >> 16;
        const int x = (mvx >> sh) + bx, y = (mvy >> sh) + by;
        const int xf = mvx & (unit - 1), yf = mvy & (unit - 1);
        const int w[4] = {(unit - yf) * (unit - xf), (unit - yf) * xf, yf * (unit - xf), yf * xf};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tx = x + (k & 1), ty = y + (k >> 1);
            if (tx < 0 || ty < 0 || tx >= nx || ty >= ny) continue;
            const int share = (w[k] * a + rnd) >> sh2;
            if (share) atomicAdd(&A[ty * nx + tx], (unsigned long long)share);     // a > 0 and the weights are >= 0: never negative
        }
    }
}
__global__ __launch_bounds__(256) void cutree_clip_kernel(int n, unsigned long long *acc0, unsigned long long *acc1, uint16_t *ref0, uint16_t *ref1, int same)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long a0 = acc0[i], a1 = acc1[i];
    acc0[i] = 0; acc1[i] = 0;                                   // the accumulators are left zero for the next call
    if (same) { const unsigned long long s = (unsigned long long)ref0[i] + a0 + a1; ref0[i] = (uint16_t)(s > 0xffffull ? 0xffffull : s); }
    else {
        const unsigned long long s0 = (unsigned long long)ref0[i] + a0, s1 = (unsigned long long)ref1[i] + a1;
        ref0[i] = (uint16_t)(s0 > 0xffffull ? 0xffffull : s0); ref1[i] = (uint16_t)(s1 > 0xffffull ? 0xffffull : s1);
    }
}

// the QP of every CTU from the 16x16 blocks' offsets: base + clip(round(mean over the CTU's blocks, summed in raster order), +-12), clipped to [lo, hi] (this build's own rule: the quantisation group
// is the CTU; the reference applies its offsets per CU inside closed code)
__global__ __launch_bounds__(64) void aq_ctu_map_kernel(const double *off, int nx, int ny, int base_qp, int lo, int hi, int8_t *map)
{
    const int cols = (nx + 3) / 4, rows = (ny + 3) / 4, ctu = blockIdx.x * 64 + threadIdx.x;
    if (ctu >= cols * rows) return;
    const int cx = ctu % cols, cy = ctu / cols;
    double sum = 0.0;
    int cnt = 0;
    for (int by = cy * 4; by < min(cy * 4 + 4, ny); ++by)
        for (int bx = cx * 4; bx < min(cx * 4 + 4, nx); ++bx) { sum += off[by * nx + bx]; ++cnt; }
    int d = (int)floor(sum / (double)cnt + 0.5);
    d = d < -12 ? -12 : d > 12 ? 12 : d;                       // two CTUs of a row are then at most 24 apart: CuQpDeltaVal stays inside [-26, 25] (7.4.9.14)
    const int q = base_qp + d;
    map[ctu] = (int8_t)(q < lo ? lo : q > hi ? hi : q);
}

extern "C" {

int ks265_aq_ctu_map(ks265_ctx *ctx, const double *dev_qp_off, int nx, int ny, int base_qp, int qp_lo, int qp_hi, int8_t *dev_map)
{
    if (!ctx || !dev_qp_off || !dev_map) return KS265_POINTER;
    if (nx <= 0 || ny <= 0 || qp_lo > qp_hi) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    const int n = ((nx + 3) / 4) * ((ny + 3) / 4);
    hipLaunchKernelGGL(aq_ctu_map_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, dev_qp_off, nx, ny, base_qp, qp_lo, qp_hi, dev_map);
    return ks265_check_launch(ctx);
}

int ks265_frame_adapt_quant(ks265_ctx *ctx, const uint8_t *dev_y, int stride_y, const uint8_t *dev_u, const uint8_t *dev_v, int stride_c, int nx, int ny, int count,
                            double strength, double *dev_qp_off, uint16_t *dev_inv_qscale, double *dev_scratch2)
{
    if (!ctx || !dev_y || !dev_u || !dev_v || !dev_qp_off || !dev_inv_qscale || !dev_scratch2) return KS265_POINTER;
    if (nx <= 0 || ny <= 0 || count <= 0 || (stride_y & 3)) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    if (la_tables_upload(ctx->device)) return KS265_FAIL;
    const int n = nx * ny;
    hipLaunchKernelGGL(aq_energy_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, dev_y, stride_y, dev_u, dev_v, stride_c, nx, ny, dev_qp_off);
    hipLaunchKernelGGL(aq_mean_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double *)dev_qp_off, n, count, strength, dev_scratch2);
    hipLaunchKernelGGL(aq_offset_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dev_qp_off, n, (const double *)dev_scratch2, dev_inv_qscale);
    return ks265_check_launch(ctx);
}

int ks265_cutree_propagate(ks265_ctx *ctx, int lg, int nx, int ny, const uint16_t *dev_intra, const uint16_t *dev_inv_qscale, const uint16_t *dev_own, const uint16_t *dev_inter,
                           const uint8_t *dev_list_bits, const int32_t *dev_mv0, const int32_t *dev_mv1, uint16_t *dev_ref0, uint16_t *dev_ref1, uint64_t *dev_acc)
{
    if (!ctx || !dev_intra || !dev_inv_qscale || !dev_own || !dev_inter || !dev_list_bits || !dev_mv0 || !dev_mv1 || !dev_ref0 || !dev_ref1 || !dev_acc) return KS265_POINTER;
    if (nx <= 0 || ny <= 0 || lg < 0 || lg > 5) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    const int n = nx * ny;
    unsigned long long *a0 = (unsigned long long *)dev_acc, *a1 = a0 + n;
    hipLaunchKernelGGL(cutree_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, lg, nx, ny, dev_intra, dev_inv_qscale, dev_own, dev_inter, dev_list_bits, dev_mv0, dev_mv1, a0, a1);
    hipLaunchKernelGGL(cutree_clip_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, a0, a1, dev_ref0, dev_ref1, dev_ref0 == dev_ref1 ? 1 : 0);
    return ks265_check_launch(ctx);
}

}  // extern "C"
