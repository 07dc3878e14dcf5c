// frame_recon.hip — Stage D: prediction -> residual -> forward transform -> quant -> dequant -> inverse transform -> recon
// (the reconstruct() chain enc@0x481da0: calc_residual -> g_H265_2dDct_Func -> g_QuantFuncs -> g_DeQuantFuncs ->
// g_H265_2dIDct_Func).  One workgroup per 32x32 luma region (+ its two 16x16 chroma regions); the region holds TUs of
// 8..32 (luma) / 4..16 (chroma) samples, every thread works on elements of its own TU; the four 1-D passes run out of
// LDS with int32 accumulators and the exact stage shifts of the reference.
#include "frame_common.h"

using namespace ks265;

struct QP {
    int scale, offF, dq, qp6;
};

__device__ __forceinline__ QP make_qp(int qp, bool intra)
{
    QP q;
    q.scale = kQuantScales[qp % 6];
    q.qp6 = qp / 6;
    q.offF = intra ? 171 : 85;
    q.dq = kInvQuantScales[qp % 6] << (qp / 6);
    return q;
}

// chroma sample prediction (interpChroma* enc@0x4111c0..): 4-tap, 1/8 sample, normative 2-D order
__device__ __forceinline__ int chroma_pred(const uint8_t *ref, long stride, int fx, int fy)
{
    if (!fx && !fy) return ref[0];
    if (!fy) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += kChromaTaps[fx][t] * (int)ref[t - 1];
        return clip8((s + 32) >> 6);
    }
    if (!fx) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += kChromaTaps[fy][t] * (int)ref[(t - 1) * stride];
        return clip8((s + 32) >> 6);
    }
    int v = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += kChromaTaps[fx][t] * (int)ref[(r - 1) * stride + t - 1];
        v += kChromaTaps[fy][r] * (int)(short)s;
    }
    return clip8((v + 2048) >> 12);
}

template <int RS /*region size in samples: 32 luma, 16 chroma*/>
__device__ __forceinline__ void code_region(const KsGeom &g, int comp, int qp, int rx8, int ry8, const ks265_cu8 *blk /*LDS [16]*/,
                                            const unsigned char *tu_log2 /*LDS [16]: log2 of TU size in 8x8 blocks*/, const short *M32, short *X,
                                            short *T, unsigned char *P, int *nzcnt /*LDS [16]*/, const uint8_t *src, const uint8_t *ref,
                                            const uint8_t *planes, int16_t *lvl, uint8_t *rec, int tid)
{
    constexpr int UNIT = RS / 4;                      // samples per 8x8-luma block along one axis
    constexpr int NE = RS * RS;
    const long stride = comp == 0 ? g.sy : g.sc;
    const int lstride = comp == 0 ? g.W : g.W / 2;
    const int X0 = rx8 * UNIT, Y0 = ry8 * UNIT;       // region origin in this component's samples
    const uint8_t *S = comp == 0 ? ks_org_y(g, src) : ks_org_c(g, src);
    uint8_t *Rc = comp == 0 ? ks_org_y(g, rec) : ks_org_c(g, rec);

    if (tid < 16) nzcnt[tid] = 0;
    // ---- prediction + residual
    for (int e = tid; e < NE; e += 256) {
        const int ex = e % RS, ey = e / RS, b = (ey / UNIT) * 4 + ex / UNIT;
        const ks265_cu8 c = blk[b];
        int pred = 128, res = 0;
        if (c.log2_cu) {                              // block inside the picture
            if (c.pred_mode == 0) {
                if (comp == 0) {
                    const uint8_t *pl = planes + (long)((c.mvy & 3) * 4 + (c.mvx & 3)) * g.bytes_y + g.org_y;
                    pred = pl[(long)(Y0 + ey + (c.mvy >> 2)) * g.sy + X0 + ex + (c.mvx >> 2)];
                } else {
                    const uint8_t *rp = ks_org_c(g, ref) + (long)(Y0 + ey + (c.mvy >> 3)) * g.sc + X0 + ex + (c.mvx >> 3);
                    pred = chroma_pred(rp, g.sc, c.mvx & 7, c.mvy & 7);
                }
            }
            res = (int)S[(long)(Y0 + ey) * stride + X0 + ex] - pred;
        }
        P[e] = (unsigned char)pred;
        X[e] = (short)res;
    }
    __syncthreads();
    // per-element TU geometry
    auto tu_of = [&](int ex, int ey, int &ox, int &oy, int &n, int &log2n, int &tb) {
        const int bx = ex / UNIT, by = ey / UNIT, b = by * 4 + bx, t8 = 1 << tu_log2[b];
        const int tbx = bx & ~(t8 - 1), tby = by & ~(t8 - 1);
        ox = tbx * UNIT; oy = tby * UNIT; n = t8 * UNIT; tb = tby * 4 + tbx;
        log2n = tu_log2[b] + (RS == 32 ? 3 : 2);
    };
    // ---- forward pass 1: T[k][j] = rnd(sum_x M[k][x] X[j][x], 2 log2N - 2)
    for (int e = tid; e < NE; e += 256) {
        const int ex = e % RS, ey = e / RS;
        int ox, oy, n, log2n, tb; tu_of(ex, ey, ox, oy, n, log2n, tb);
        const int k = ey - oy, j = ex - ox, s1 = 2 * log2n - 2;
        const short *m = M32 + (k << (5 - log2n)) * 32, *xr = X + (oy + j) * RS + ox;
        int acc = 0;
        for (int x = 0; x < n; ++x) acc += (int)m[x] * (int)xr[x];
        T[e] = (short)((acc + (1 << (s1 - 1))) >> s1);
    }
    __syncthreads();
    // ---- forward pass 2 + quant + dequant (coefficient (k, j) lives at element (oy + k, ox + j))
    const QP q = make_qp(qp, false);
    for (int e = tid; e < NE; e += 256) {
        const int ex = e % RS, ey = e / RS, b = (ey / UNIT) * 4 + ex / UNIT;
        int ox, oy, n, log2n, tb; tu_of(ex, ey, ox, oy, n, log2n, tb);
        const int k = ey - oy, j = ex - ox;
        const short *m = M32 + (k << (5 - log2n)) * 32, *tr = T + (oy + j) * RS + ox;
        int acc = 0;
        for (int x = 0; x < n; ++x) acc += (int)m[x] * (int)tr[x];
        const int coef = (short)((acc + 64) >> 7);
        const ks265_cu8 c = blk[b];
        int l = 0, dqv = 0;
        if (c.log2_cu) {
            const int qbits = 21 + q.qp6 - log2n, offF = c.pred_mode == 1 ? 171 : 85, du_unused = 0;
            (void)du_unused;
            int du;
            l = quant_one(coef, q.scale, offF << (qbits - 9), qbits, du);
            const int shift = log2n - 1;
            dqv = dequant_one(l, q.dq, 1 << (shift - 1), shift);
            lvl[(long)(Y0 + ey) * lstride + X0 + ex] = (int16_t)l;
            if (l) atomicAdd(&nzcnt[tb], 1);
        }
        X[e] = (short)dqv;
    }
    __syncthreads();
    // ---- inverse pass 1: T[y][x] = clip16((sum_k M[k][y] C[k][x] + 64) >> 7)
    for (int e = tid; e < NE; e += 256) {
        const int ex = e % RS, ey = e / RS;
        int ox, oy, n, log2n, tb; tu_of(ex, ey, ox, oy, n, log2n, tb);
        const int y = ey - oy, sh = 5 - log2n;
        int acc = 0;
        if (nzcnt[tb])
            for (int k = 0; k < n; ++k) acc += (int)M32[(k << sh) * 32 + y] * (int)X[(oy + k) * RS + ex];
        T[e] = (short)clip16((acc + 64) >> 7);
    }
    __syncthreads();
    // ---- inverse pass 2 + pred add: R = (sum_k T[y][k] M[k][x] + 2048) >> 12
    for (int e = tid; e < NE; e += 256) {
        const int ex = e % RS, ey = e / RS, b = (ey / UNIT) * 4 + ex / UNIT;
        if (!blk[b].log2_cu) continue;
        int ox, oy, n, log2n, tb; tu_of(ex, ey, ox, oy, n, log2n, tb);
        const int x = ex - ox, sh = 5 - log2n;
        int r = 0;
        if (nzcnt[tb]) {
            int acc = 0;
            for (int k = 0; k < n; ++k) acc += (int)T[ey * RS + ox + k] * (int)M32[(k << sh) * 32 + x];
            r = (acc + 2048) >> 12;
        }
        Rc[(long)(Y0 + ey) * stride + X0 + ex] = (uint8_t)clip8((int)P[e] + r);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void reconstruct_kernel(KsGeom g, int qp, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v,
                                                          const uint8_t *ref_u, const uint8_t *ref_v, const uint8_t *planes, ks265_cu8 *cu8,
                                                          int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v)
{
    __shared__ short M32[32 * 32];
    __shared__ short X[32 * 32];
    __shared__ short T[32 * 32];
    __shared__ unsigned char P[32 * 32];
    __shared__ ks265_cu8 blk[16];
    __shared__ unsigned char tu_log2[16];
    __shared__ int nzcnt[16];
    __shared__ int cbf[16];
    const int tid = threadIdx.x;
    const int rx = blockIdx.x, ry = blockIdx.y;            // 32x32 region index
    load_matrix(M32, 4, 32, tid, 256);
    if (tid < 16) {
        const int bx = rx * 4 + (tid & 3), by = ry * 4 + (tid >> 2);
        ks265_cu8 c;
        c.mvx = 0; c.mvy = 0; c.log2_cu = 0; c.cbf = 0; c.pred_mode = 0; c.rsv = 0;
        if (bx < g.w8 && by < g.h8) c = cu8[(long)by * g.w8 + bx];
        blk[tid] = c;
        tu_log2[tid] = (unsigned char)(c.log2_cu ? min((int)c.log2_cu - 3, 2) : 0);
        cbf[tid] = 0;
    }
    __syncthreads();
    const int qpc = chroma_qp(qp);
    code_region<32>(g, 0, qp, rx * 4, ry * 4, blk, tu_log2, M32, X, T, P, nzcnt, src_y, nullptr, planes, lvl_y, rec_y, tid);
    if (tid < 16 && blk[tid].log2_cu) {
        const int t8 = 1 << tu_log2[tid], tb = ((tid >> 2) & ~(t8 - 1)) * 4 + ((tid & 3) & ~(t8 - 1));
        if (nzcnt[tb]) cbf[tid] |= 1;
    }
    __syncthreads();
    code_region<16>(g, 1, qpc, rx * 4, ry * 4, blk, tu_log2, M32, X, T, P, nzcnt, src_u, ref_u, planes, lvl_u, rec_u, tid);
    if (tid < 16 && blk[tid].log2_cu) {
        const int t8 = 1 << tu_log2[tid], tb = ((tid >> 2) & ~(t8 - 1)) * 4 + ((tid & 3) & ~(t8 - 1));
        if (nzcnt[tb]) cbf[tid] |= 2;
    }
    __syncthreads();
    code_region<16>(g, 2, qpc, rx * 4, ry * 4, blk, tu_log2, M32, X, T, P, nzcnt, src_v, ref_v, planes, lvl_v, rec_v, tid);
    if (tid < 16 && blk[tid].log2_cu) {
        const int t8 = 1 << tu_log2[tid], tb = ((tid >> 2) & ~(t8 - 1)) * 4 + ((tid & 3) & ~(t8 - 1));
        if (nzcnt[tb]) cbf[tid] |= 4;
        const int bx = rx * 4 + (tid & 3), by = ry * 4 + (tid >> 2);
        cu8[(long)by * g.w8 + bx].cbf = (uint8_t)cbf[tid];
    }
}

extern "C" int ks265_reconstruct(ks265_frame *f, ks265_pic src, ks265_pic ref, const uint8_t *planes, ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u,
                                 int16_t *lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !cu8 || !lvl_y || !lvl_u || !lvl_v || !recon.y) return KS265_POINTER;
    dim3 grid((f->g.W + 31) / 32, (f->g.H + 31) / 32);
    hipLaunchKernelGGL(reconstruct_kernel, grid, dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, src.y, src.u, src.v, ref.u, ref.v, planes, cu8, lvl_y,
                       lvl_u, lvl_v, recon.y, recon.u, recon.v);
    return ks265_check_launch(f->ctx);
}
