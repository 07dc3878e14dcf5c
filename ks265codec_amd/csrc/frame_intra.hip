// frame_intra.hip — intra pictures (intra prediction kernels + mode pre-selection).
//   ks265_intra_decide       all 35 luma modes of every 8x8 / 16x16 / 32x32 block predicted from SOURCE neighbours
//                            (decideBestLumaModeBySadFast enc@0x499170 lineage: pre-selection on source pixels is embarrassingly
//                            parallel), cost = SATD + lambda * mode bits, CU quadtree bottom-up.  One workgroup per CTU.
//   ks265_intra_reconstruct  the sequential part: CTUs as a wavefront = the reference's WPP order (CCtuEncWpp::waitForTopRightCtu
//                            enc@0x46f4a0) (one workgroup per CTU row, two CTUs behind the row above,
//                            progress counters in HBM), CUs in z-order, neighbours from the RECONSTRUCTED picture with the
//                            normative availability / substitution / smoothing rules, then the reconstruct() chain per TU.
// Prediction arithmetic = g_IntraPredFunction enc@0x7070a0 / IntraPredFilterRef_c enc@0x424110 (intra_dev.h, pinned).
#include "frame_common.h"
#include <cstddef>
#include "intra_dev.h"
#include "recon_dev.h"

using namespace ks265;

__device__ __forceinline__ int spread4(int v) { return (v & 1) | ((v & 2) << 1) | ((v & 4) << 2) | ((v & 8) << 3); }
// H.265 6.4.1 (z-scan order availability): is luma sample (nx, ny) available to the block whose first sample is (x, y)?
__device__ __forceinline__ bool intra_avail(const KsGeom &g, int x, int y, int nx, int ny)
{
    if (nx < 0 || ny < 0 || nx >= g.W || ny >= g.H) return false;
    const int ca = (y >> 6) * g.ctu_cols + (x >> 6), na = (ny >> 6) * g.ctu_cols + (nx >> 6);
    if (na != ca) return na < ca;
    return (spread4((nx & 63) >> 2) | (spread4((ny & 63) >> 2) << 1)) < (spread4((x & 63) >> 2) | (spread4((y & 63) >> 2) << 1));
}

// Availability mask of the 2 nu + 1 neighbour UNITS of the n x n luma block at (x, y): a unit = 8 luma samples (all CUs are
// >= 8x8 and aligned, so availability is constant inside a unit); bit j: j < nu left side bottom-up, j == nu the corner,
// j > nu the top side left to right.  The chroma blocks of the CU have the same mask (their units are 4 chroma samples).
// Every wave computes it for itself (lanes 0..2nu), so no barrier is needed.
__device__ __forceinline__ unsigned intra_unit_mask(const KsGeom &g, int x, int y, int n, int lane)
{
    const int nu = n >> 2;                                           // 2n / 8
    bool av = false;
    if (lane <= 2 * nu) {
        const int sx = lane <= nu ? x - 1 : x + (lane - nu - 1) * 8;
        const int sy = lane < nu ? y + 2 * n - 1 - lane * 8 : y - 1;
        av = intra_avail(g, x, y, sx, sy);
    }
    return (unsigned)__ballot(av);
}

// Reference sample at scan position q (0 = bottom-left end, 2n = corner, 4n = top-right end) of the n x n block at (x, y) of a
// plane (component samples, unit = u samples), with the substitution process of H.265 8.4.4.2.2 expressed on the unit mask:
// an unavailable sample takes the last sample of the nearest available unit below it in scan order, or - if there is none -
// the first sample of the first available unit.  COHERENT: the plane is being written by other workgroups (L2-coherent loads).
template <bool COHERENT>
__device__ __forceinline__ int intra_ref_sample(const uint8_t *plane, int stride, unsigned mask, int x, int y, int n, int u, int q)
{
    if (mask == 0) return 128;
    const int nu = 2 * n / u;
    const int j = q < 2 * n ? q / u : (q == 2 * n ? nu : nu + 1 + (q - 2 * n - 1) / u);
    if (!((mask >> j) & 1u)) {
        const unsigned below = mask & ((1u << j) - 1u);
        if (below) {
            const int jj = 31 - __clz((int)below);
            q = jj < nu ? jj * u + u - 1 : (jj == nu ? 2 * n : 2 * n + (jj - nu) * u);
        } else {
            const int jj = __ffs((int)mask) - 1;
            q = jj < nu ? jj * u : (jj == nu ? 2 * n : 2 * n + 1 + (jj - nu - 1) * u);
        }
    }
    const int sx = q > 2 * n ? x + (q - 2 * n - 1) : x - 1, sy = q < 2 * n ? y + (2 * n - 1 - q) : y - 1;
    const uint8_t *p = plane + (long)sy * stride + sx;
    if (COHERENT) return (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)*p;
}

// ------------------------------------------------------------------ reconstruction (wavefront)
#ifdef KS_INTRA_CLOCK
// experiment build (-DKS_INTRA_CLOCK; tools/intra_clock.py reads the sums): where the time of the intra chain goes - cycles of wave 0 per phase over all work-groups.
// Measured in round 6 at 2160p (profiles/r06_intra_clock.txt): a CU is 25 - 30 dependent LDS round trips, not arithmetic - 8x8: 11.9 K cycles (TU pipeline 8.5 K: sign-data
// hiding 3.9 K, prediction + residual 2.0 K, the four transform passes 3.3 K), 16x16: 14.4 K, 32x32: 25 K (its luma block alone; coding chroma beside it on two more waves
// changed nothing).  Cheaper exchanges (DPP for the coefficient groups' rows) and a horizontal-mode fast path changed nothing either.
__device__ unsigned long long g_iclk[2][32];
#define ICLK_DECL unsigned long long ick_t = __builtin_amdgcn_s_memtime(); unsigned long long ick_a[24] = {}, g_tclk_acc[9] = {}
#define ICLK(i) do { const unsigned long long ick_n = __builtin_amdgcn_s_memtime(); ick_a[i] += ick_n - ick_t; ick_t = ick_n; } while (0)
#define ICLK_CNT(i) do { ick_a[i] += 1; } while (0)
#define TCLK(i) do { if (ck) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ck[i] += tn - *ckt; *ckt = tn; } } while (0)
#define TCLK_ARGS , unsigned long long *ck = nullptr, unsigned long long *ckt = nullptr
extern "C" int ks265_debug_clock_read(unsigned long long *out64, int reset)
{
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_iclk), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[64]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_iclk), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#else
#define ICLK_DECL
#define ICLK(i)
#define ICLK_CNT(i)
#define TCLK(i)
#define TCLK_ARGS
#endif
struct IntraLds {
    short Mf[MAT_SHORTS], Mt[MAT_SHORTS];
    short X[3][32 * RP], T[3][32 * RP];                  // [component]; chroma uses the first 16 rows
    unsigned char P[3][32 * 32];
    unsigned char raw[3][132], fil[132];                 // reference arrays, corner at index 66 (luma: 64 + 1 + 64; chroma 32 + 1 + 32)
    int nz[3];
    int lastcg[3];
    int cbf[64];
    ks265_cu8 cu[64];
    // reconstructed samples around the CTU being coded: row 0 = the row above the CTU (x = -1 .. 127: top-left, top, top-right),
    // rows 1..64 = the CTU with column 3 = the last column of the previous CTU of this row; sample (x, y) of the CTU sits at
    // [(1 + y) * pitch + 4 + x].  Every neighbour gather of a CU reads this window, never HBM.
    unsigned char WY[65 * 136];
    unsigned char WC[2][33 * 72];
    unsigned char SY[64 * 64], SC[2][32 * 32];           // the CTU's source samples (one HBM round trip per CTU instead of one per CU)
    int dcv[3];
};

// one thread's role in a TU pipeline: component + quad, or idle
struct TuRole { int comp, n, log2n, qx, qy; bool on; };

// everything a TU pipeline needs besides the role (uniform per kernel / per CU)
struct TuCtx {
    const KsGeom *g;
    int16_t *lvl_y, *lvl_u, *lvl_v;
    uint8_t *R0, *R1, *R2;
    int qsc[2], qdq[2], qp6[2];
    int x0, y0, lx, ly, mode;
    bool filt;
    bool sdh;
    int qoff;                                   // the quantiser's rounding offset: 171 in I slices, 85 in P / B slices (H265_GetBaseQuantParam enc@0x4a9c90)
    long long rdo_lam2k;                        // cfg.rdo x lambda_q4^2 for the intra CUs of P / B pictures, 0 = off
};

// BLOCK: the quads of a TU are spread over several waves -> work-group barriers (LDS only); otherwise the whole TU lives in one
// wave: LDS operations of one wave complete in order, so waiting for this wave's own LDS traffic is enough.
template <bool BLOCK>
__device__ __forceinline__ void tu_sync()
{
    if (BLOCK) lds_barrier();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// prediction -> residual -> forward transform -> quant -> dequant -> inverse transform -> reconstructed samples of one quad
// (the reconstruct() chain enc@0x481da0, same arithmetic as code_region of frame_recon.hip with the intra rounding offset 171)
template <bool BLOCK>
__device__ __forceinline__ void tu_pipeline(IntraLds &L, const TuRole &r, const TuCtx &c TCLK_ARGS)
{
    const int cp = r.comp, nn = r.n, l2 = r.log2n, mp = nn + 4;
    short *X = L.X[cp], *T = L.T[cp];
    unsigned char *P = L.P[cp];
    const short *mf = L.Mf + mat_off(l2), *mt = L.Mt + mat_off(l2);
    const int px = cp ? c.x0 >> 1 : c.x0, py = cp ? c.y0 >> 1 : c.y0, stride = cp ? c.g->sc : c.g->sy, lstride = cp ? c.g->W / 2 : c.g->W;
    if (r.on) {
        const unsigned char *ref = cp == 0 ? (c.filt ? &L.fil[66] : &L.raw[0][66]) : &L.raw[cp][66];
        const int dc = c.mode == 1 ? L.dcv[cp] : 0;
        const unsigned sv = cp == 0 ? *(const unsigned *)&L.SY[(c.ly * 8 + r.qy) * 64 + c.lx * 8 + r.qx]
                                    : *(const unsigned *)&L.SC[cp - 1][(c.ly * 4 + r.qy) * 32 + c.lx * 4 + r.qx];
        int pr[4];
        unsigned short res[4];
        intra_quad(ref, c.mode, l2, r.qx, r.qy, dc, cp == 0, pr);
#pragma unroll
        for (int i = 0; i < 4; ++i) res[i] = (unsigned short)(short)((int)((sv >> (8 * i)) & 255) - pr[i]);
        *(unsigned *)(P + r.qy * 32 + r.qx) = (unsigned)pr[0] | ((unsigned)pr[1] << 8) | ((unsigned)pr[2] << 16) | ((unsigned)pr[3] << 24);
        *(uint2 *)(X + r.qy * RP + r.qx) = make_uint2(res[0] | ((unsigned)res[1] << 16), res[2] | ((unsigned)res[3] << 16));
    }
    tu_sync<BLOCK>();
    TCLK(0);
    if (r.on) {                                                     // forward pass 1: T[k][j] = rnd(M[k] . X[j], 2 log2N - 2)
        const int s1 = 2 * l2 - 2;
        int acc[4];
        quad_dot(mf + r.qy * mp, X + r.qx * RP, RP, nn, acc);
        unsigned short o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned short)(short)((acc[i] + (1 << (s1 - 1))) >> s1);
        *(uint2 *)(T + r.qy * RP + r.qx) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
    tu_sync<BLOCK>();
    TCLK(1);
    // ---- forward pass 2 + quantisation, then the postQuant seam IN REGISTERS: a lane holds one row of four coefficients of its 4x4 coefficient group, the group's four
    // rows sit Q = N / 4 lanes apart in the same wave, so what a group needs to know about itself travels by two xor-shuffles - no LDS round trip, no lane walking
    // sixteen coefficients on its own (that serial walk, not the transforms, was the time of a TU: 5.4 of 11.2 us for a 16x16 TU, 2160p, round 3).
    int lvq[4] = {0, 0, 0, 0}, duq[4] = {0, 0, 0, 0}, cfq[4] = {0, 0, 0, 0};
    const int ci = cp ? 1 : 0, dqs = c.qdq[ci], dshift = l2 - 1, Q = nn >> 2;
    if (r.on) {
        int acc[4];
        quad_dot(mf + r.qy * mp, T + r.qx * RP, RP, nn, acc);
        const int scale = c.qsc[ci], qbits = 21 + c.qp6[ci] - l2, off = c.qoff << (qbits - 9);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cfq[i] = (short)((acc[i] + 64) >> 7);
            int du;
            lvq[i] = (short)quant_one(cfq[i], scale, off, qbits, du);
            duq[i] = (short)du;                                     // (16 bits, as the reference keeps its deltaU)
        }
    }
    TCLK(2);
    if (c.rdo_lam2k && __builtin_amdgcn_readfirstlane(cp) == 0) {
        // cfg.rdo (intra CUs of P / B pictures, luma only: a wave's lanes are all luma or all chroma): coefficient-group pruning before sign-data hiding
        // (recon_dev.h rdo_group_prune, the oracle's code_tu)
        long long gain = 0;
        int bc = 0;                                                   // bits << 8 | levels
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lvq[i]) {
                const int d = dequant_one(lvq[i], dqs, 1 << (dshift - 1), dshift);
                gain += (long long)d * (2 * cfq[i] - d);
                bc += (rdo_level_q2(lvq[i] < 0 ? -lvq[i] : lvq[i]) << 8) | 1;
            }
#pragma unroll
        for (int k = 1; k <= 2; ++k) {
            const int lo = __shfl_xor((int)(unsigned)gain, Q * k, 64), hi = __shfl_xor((int)(gain >> 32), Q * k, 64);
            gain += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
            bc += __shfl_xor(bc, Q * k, 64);
        }
        const int cnt = bc & 255, bits = (bc >> 8) + 10 + (16 - cnt);
        if (cnt && ((gain >> (2 * (7 - l2))) << 12) <= c.rdo_lam2k * bits) { lvq[0] = 0; lvq[1] = 0; lvq[2] = 0; lvq[3] = 0; }
    }
    TCLK(3);
    {   // levels per component of this wave (a wave holds one component, the chroma wave of a small CU both chroma components)
        const int cp0 = __builtin_amdgcn_readfirstlane(cp);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) cnt += __popcll(__ballot(r.on && cp == cp0 + k && lvq[i] != 0));
            if (cnt && (threadIdx.x & 63) == 0) atomicAdd(&L.nz[cp0 + k], cnt);
        }
    }
    TCLK(4);
    int qpos[4] = {0, 0, 0, 0};
    unsigned nzmask = 0, negmask = 0;
    int csum = 0, gorder = 0;
    if (c.sdh) {
        // the postQuant seam (postQuant enc@0x4ace80): sign-data hiding with the TU's scan (H.265 7.4.9.11: intra 4x4 / 8x8 luma and 4x4 chroma follow the
        // prediction mode).  Survey: the group's levels as masks in SCAN order (bit q = position q holds a level / a negative one) and their sum
        const int scan = (nn == 4 || (nn == 8 && cp == 0)) ? ((c.mode >= 6 && c.mode <= 14) ? 2 : (c.mode >= 22 && c.mode <= 30) ? 1 : 0) : 0;
        const int y = r.qy & 3;
        unsigned v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qpos[i] = scan == 0 ? (int)((0xfda6eb73c8419520ull >> (4 * (y * 4 + i))) & 15ull) : scan == 1 ? y * 4 + i : i * 4 + y;
            if (lvq[i]) v |= (lvq[i] < 0 ? 0x10001u : 1u) << qpos[i];
            csum += lvq[i];
        }
#pragma unroll
        for (int k = 1; k <= 2; ++k) { v ^= (unsigned)__shfl_xor((int)v, Q * k, 64); csum += __shfl_xor(csum, Q * k, 64); }   // (the rows' masks are disjoint: xor = or)
        nzmask = v & 0xFFFFu; negmask = v >> 16;
        if (r.on && nzmask) {
            gorder = sbh_group_order(scan, nn >> 2, r.qx >> 2, r.qy >> 2) + 1;
            if (y == 0) atomicMax(&L.lastcg[cp], gorder);            // the last group in scan order that holds a level
        }
        tu_sync<BLOCK>();
        // fix the parity of the group if its first sign is hidden and the parity is wrong (signBitHidingHDQ enc@0x4aa150): the cheapest +-1 among the positions up to
        // `start`, ties to the highest position (the reference walks downwards with a strict '<')
        const int first = nzmask ? __ffs((int)nzmask) - 1 : 0, last = nzmask ? 31 - __clz((int)nzmask) : -1;
        const int signbit = (int)((negmask >> first) & 1u);
        unsigned key = 0xFFFFFFFFu;
        int ksel = 0;
        if (r.on && nzmask && last - first >= 4 && signbit != (csum & 1)) {
            const int start = L.lastcg[cp] == gorder ? last : 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = qpos[i], l = lvq[i], du = duq[i];
                int cost = 0;
                bool ok = q <= start;
                if (l != 0) {
                    if (du > 0) cost = -du;
                    else if (q == first && (l == 1 || l == -1)) ok = false;
                    else cost = du;
                } else if (q < first) {
                    if ((cfq[i] < 0 ? 1 : 0) != signbit) ok = false;
                    else cost = -du;
                } else cost = -du;
                const unsigned kk = ok ? ((unsigned)(cost + 32768) << 4) | (unsigned)(15 - q) : 0xFFFFFFFFu;
                if (kk < key) { key = kk; ksel = i; }
            }
        }
        unsigned kmin = key;
#pragma unroll
        for (int k = 1; k <= 2; ++k) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, Q * k, 64));
        if (key != 0xFFFFFFFFu && key == kmin) {                       // this lane holds the position that moves
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i == ksel) {
                    const int l = lvq[i], du = duq[i];
                    int change = l != 0 ? (du > 0 ? 1 : -1) : 1;
                    if (l == 32767 || l == -32768) change = -1;
                    lvq[i] = (int)(short)(cfq[i] >= 0 ? l + change : l - change);
                }
        }
    }
    TCLK(5);
    if (r.on) {                                                     // the final levels: to the level plane, dequantised (transposed) for the inverse transform
        unsigned short lv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lv[i] = (unsigned short)(short)lvq[i];
            X[(r.qx + i) * RP + r.qy] = (short)dequant_one(lvq[i], dqs, 1 << (dshift - 1), dshift);
        }
        *(uint2 *)((cp == 0 ? c.lvl_y : (cp == 1 ? c.lvl_u : c.lvl_v)) + (long)(py + r.qy) * lstride + px + r.qx) =
            make_uint2(lv[0] | ((unsigned)lv[1] << 16), lv[2] | ((unsigned)lv[3] << 16));
    }
    tu_sync<BLOCK>();
    TCLK(6);
    const bool live = r.on && L.nz[cp] != 0;
    if (r.on) {                                                     // inverse pass 1: T[y][x] = clip16((Mt[y] . Ct[x] + 64) >> 7)
        int acc[4] = {0, 0, 0, 0};
        if (live) quad_dot(mt + r.qy * mp, X + r.qx * RP, RP, nn, acc);
        unsigned short o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned short)(short)clip16((acc[i] + 64) >> 7);
        *(uint2 *)(T + r.qy * RP + r.qx) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
    tu_sync<BLOCK>();
    TCLK(7);
    if (r.on) {                                                     // inverse pass 2 + prediction -> reconstructed samples
        int acc[4] = {0, 0, 0, 0};
        if (live) quad_dot(T + r.qy * RP, mt + r.qx * mp, mp, nn, acc);
        const unsigned pv = *(const unsigned *)(P + r.qy * 32 + r.qx);
        unsigned o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) o |= (unsigned)clip8((int)((pv >> (8 * i)) & 255) + (live ? (acc[i] + 2048) >> 12 : 0)) << (8 * i);
        *(unsigned *)((cp == 0 ? c.R0 : (cp == 1 ? c.R1 : c.R2)) + (long)(py + r.qy) * stride + px + r.qx) = o;
        if (cp == 0) *(unsigned *)&L.WY[(1 + c.ly * 8 + r.qy) * 136 + 4 + c.lx * 8 + r.qx] = o;
        else *(unsigned *)&L.WC[cp - 1][(1 + c.ly * 4 + r.qy) * 72 + 4 + c.lx * 4 + r.qx] = o;
    }
    tu_sync<BLOCK>();
    TCLK(8);
}

// PMODE = false: an intra picture (every CU).  PMODE = true (cfg.intra_inter): the intra CUs (pred_mode 2) of a P / B picture AFTER reconstruct_kernel has written every
// inter CU - a CTU without intra CUs is passed over at once, the others load their own reconstructed samples into the window first (inter neighbours count like intra
// ones: constrained_intra_pred_flag = 0), quantise with the slice's rounding offset and prune coefficient groups like the inter TUs (cfg.rdo).
template <bool PMODE>
__global__ __launch_bounds__(256) void intra_recon_kernel(KsGeom g, int qp, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, ks265_cu8 *cu8,
                                                          int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v,
                                                          int *progress, unsigned *err_word, int spin_limit, int sdh_on, long long rdo_lam2k, const int8_t *qp_map)
{
    __shared__ __attribute__((aligned(16))) IntraLds L;
    // PMODE: one work-group per CTU in raster order (a CTU's neighbours have smaller indices, so the work-groups it may wait for were dispatched before it);
    // an intra picture: one work-group per CTU ROW walking its CTUs (every CTU has work, the row is the natural unit)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cy = PMODE ? (int)blockIdx.x / g.ctu_cols : (int)blockIdx.x;
    const int cx_first = PMODE ? (int)blockIdx.x % g.ctu_cols : 0, cx_end = PMODE ? cx_first + 1 : g.ctu_cols;
    __shared__ int s_need;                                           // PMODE: which of the four neighbour CTUs (bit 0 left, 1 top-left, 2 top, 3 top-right) this CTU's intra CUs read
    if (PMODE) {                                                     // most CTUs of a P / B picture hold no intra CU: leave before any set-up
        __shared__ int s_any;
        if (tid < 64) {
            const int bx = cx_first * 8 + (tid & 7), by = cy * 8 + (tid >> 3);
            bool intra = false;
            if (bx < g.w8 && by < g.h8) { const ks265_cu8 q = cu8[(long)by * g.w8 + bx]; intra = q.pred_mode == 2; }
            const unsigned long long any = __ballot(intra);              // bit = 8x8 block, raster
            if (tid == 0) {
                s_any = any != 0ull;
                // an intra CU reads a neighbour CTU's samples only across the border it touches: left column -> the left CTU (incl. its below-left samples), block (0, 0) -> the
                // top-left one, top row -> the top one, top row from x = 32 on -> the top-right one (x0 + 2 n > 64 needs x0 >= 32).  The others need not be waited for: round 4,
                // the wavefront of a 2160p P picture is cut where intra CUs lie inside their CTUs
                s_need = ((any & 0x0101010101010101ull) ? 1 : 0) | ((any & 1ull) ? 2 : 0) | ((any & 0xFFull) ? 4 : 0) | ((any & 0xF0ull) ? 8 : 0);
            }
        }
        __syncthreads();
        if (!s_any) {
            if (tid == 0) __hip_atomic_store(progress + blockIdx.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    ICLK_DECL;
    build_matrices(L.Mf, L.Mt, tid, 256);                            // (cheaper than fetching the frame's copy: no memory latency)
    TuCtx c;
    c.sdh = sdh_on != 0;
    c.rdo_lam2k = PMODE ? rdo_lam2k : 0; c.qoff = PMODE ? 85 : 171;
    c.g = &g; c.lvl_y = lvl_y; c.lvl_u = lvl_u; c.lvl_v = lvl_v;
    // quantiser constants of the two QPs, fetched once per CTU (a table load inside the CU loop would sit behind every outstanding store); with a QP per CTU
    // (qp_map: cu_qp_delta, quantisation group = CTU) a key picture's work-group sets them again at every CTU of its row
    auto set_qp = [&](int q) {
        const int qc = chroma_qp(q);
        c.qsc[0] = kQuantScales[q % 6]; c.qsc[1] = kQuantScales[qc % 6];
        c.qdq[0] = kInvQuantScales[q % 6] << (q / 6); c.qdq[1] = kInvQuantScales[qc % 6] << (qc / 6);
        c.qp6[0] = q / 6; c.qp6[1] = qc / 6;
    };
    set_qp(qp_map ? qp_map[cy * g.ctu_cols + cx_first] : qp);
    // (component pointers are picked with selects, not from an array: a dynamically indexed pointer array loses the global address
    //  space, its stores become FLAT stores, and FLAT stores count against lgkmcnt - every LDS barrier would wait for HBM)
    const uint8_t *const S0 = ks_org_y(g, src_y), *const S1 = ks_org_c(g, src_u), *const S2 = ks_org_c(g, src_v);
    c.R0 = ks_org_y(g, rec_y); c.R1 = ks_org_c(g, rec_u); c.R2 = ks_org_c(g, rec_v);
    for (int cx = cx_first; cx < cx_end; ++cx) {
        __syncthreads();                                             // nobody still walks the previous CTU's map
        if (!PMODE && qp_map && cx > cx_first) set_qp(qp_map[cy * g.ctu_cols + cx]);
        if (tid < 64) {
            const int bx = cx * 8 + (tid & 7), by = cy * 8 + (tid >> 3);
            ks265_cu8 cu;
            cu.mvx = 0; cu.mvy = 0; cu.mv1x = 0; cu.mv1y = 0; cu.log2_cu = 0; cu.cbf = 0; cu.pred_mode = 0; cu.inter_dir = 0;
            if (bx < g.w8 && by < g.h8) cu = cu8[(long)by * g.w8 + bx];
            L.cu[tid] = cu;
            L.cbf[tid] = PMODE && cu.pred_mode != 2 ? cu.cbf : 0;        // the inter CUs keep what reconstruct_kernel found
            if (!PMODE) {
                // the previous CTU's last column becomes this CTU's left neighbour column
                L.WY[(1 + tid) * 136 + 3] = L.WY[(1 + tid) * 136 + 4 + 63];
                L.WC[tid >> 5][(1 + (tid & 31)) * 72 + 3] = L.WC[tid >> 5][(1 + (tid & 31)) * 72 + 4 + 31];
            }
        }
        auto load_own = [&]() {
            // source samples of the CTU: 64 rows x 64 bytes luma (16-byte pieces), 2 x 32 x 32 chroma; rows below the picture are padding, never used
            const int r = tid >> 2, cc16 = (tid & 3) * 16;
            *(uint4 *)&L.SY[r * 64 + cc16] = *(const uint4 *)(S0 + (long)(cy * 64 + r) * g.sy + cx * 64 + cc16);
            const int cc = tid >> 7, t = tid & 127, rc = t >> 2, c8 = (t & 3) * 8;
            *(uint2 *)&L.SC[cc][rc * 32 + c8] = *(const uint2 *)((cc ? S2 : S1) + (long)(cy * 32 + rc) * g.sc + cx * 32 + c8);
            if (PMODE) {
                // the CTU's own reconstructed samples (the inter CUs, written by reconstruct_kernel before this launch; window rows start 4 bytes into an
                // 8-byte grid: dword stores)
                const uint4 wy = *(const uint4 *)(c.R0 + (long)(cy * 64 + r) * g.sy + cx * 64 + cc16);
                unsigned *dy = (unsigned *)&L.WY[(1 + r) * 136 + 4 + cc16];
                dy[0] = wy.x; dy[1] = wy.y; dy[2] = wy.z; dy[3] = wy.w;
                const uint2 wc = *(const uint2 *)((cc ? c.R2 : c.R1) + (long)(cy * 32 + rc) * g.sc + cx * 32 + c8);
                unsigned *dc = (unsigned *)&L.WC[cc][(1 + rc) * 72 + 4 + c8];
                dc[0] = wc.x; dc[1] = wc.y;
            }
        };
        ICLK(0);                                                     // [0] set-up: matrices, map load
        if (PMODE) load_own();                                       // nothing of this depends on the neighbours: under way before the wait
        if (PMODE) {
            // the neighbour CTUs whose samples this CTU's intra CUs read (of left, top-left, top, top-right: s_need) must be through - with or without intra CUs of
            // their own (those without flagged themselves at once)
            if (tid < 4 && ((s_need >> tid) & 1)) {
                const int nx = tid == 0 ? cx - 1 : cx - 2 + tid, ny = tid == 0 ? cy : cy - 1;
                // ... and only if that CTU holds an intra CU along the shared border (its right column / bottom-right block / bottom row / left half of its bottom row): what
                // reconstruct_kernel wrote there is final otherwise
                bool dep = false;
                if (nx >= 0 && nx < g.ctu_cols && ny >= 0)
                    for (int k = 0; k < (tid == 1 ? 1 : tid == 3 ? 4 : 8); ++k) {
                        const int bx = nx * 8 + (tid == 0 || tid == 1 ? 7 : k), by = ny * 8 + (tid == 0 ? k : 7);
                        if (bx < g.w8 && by < g.h8 && cu8[(long)by * g.w8 + bx].pred_mode == 2) dep = true;
                    }
                if (dep) {
                    int spins = 0;
                    while (__hip_atomic_load(progress + ny * g.ctu_cols + nx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        if (spins++ >= spin_limit) { __hip_atomic_fetch_or(err_word, KS_DEVERR_WAVEFRONT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
            }
        } else if (cy > 0 && tid == 0) {
            // wavefront: the row above must be two CTUs ahead (top-right neighbours)
            const int need = min(cx + 2, g.ctu_cols);
            // bounded: rows are normally dispatched in order, so the row above is resident and this never spins long; if it ever does
            // (a lost launch, a row that is not resident under heavy multi-stream load), give up after ~1 s instead of hanging the GPU
            // and SAY SO: the error word makes ks265_synchronize return KS265_FAIL (the picture is invalid and must be re-encoded)
            int spins = 0;
            while (__hip_atomic_load(progress + cy - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                if (spins++ >= spin_limit) { __hip_atomic_fetch_or(err_word, KS_DEVERR_WAVEFRONT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        ICLK(1);                                                     // [1] waiting for the neighbour CTUs
        if (!PMODE) load_own();
        if (PMODE && cx > 0) {                                       // the left neighbour column: another work-group's stores (L2-coherent loads)
            if (tid < 64) L.WY[(1 + tid) * 136 + 3] = __hip_atomic_load(c.R0 + (long)(cy * 64 + tid) * g.sy + cx * 64 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (tid < 128) {
                const int k = tid - 64, pc = k >> 5, rr = k & 31;
                L.WC[pc][(1 + rr) * 72 + 3] = __hip_atomic_load((pc ? c.R2 : c.R1) + (long)(cy * 32 + rr) * g.sc + cx * 32 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (cy > 0) {                                                // the row above: finished by another workgroup -> L2-coherent loads
            if (tid < 129) {
                const int x = cx * 64 - 1 + tid;
                if (x >= 0 && x < g.W) L.WY[3 + tid] = __hip_atomic_load(c.R0 + (long)(cy * 64 - 1) * g.sy + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < 130) {
                const int cc = tid / 65, k = tid % 65, x = cx * 32 - 1 + k;
                if (x >= 0 && x < g.W / 2) L.WC[cc][3 + k] = __hip_atomic_load((cc ? c.R2 : c.R1) + (long)(cy * 32 - 1) * g.sc + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        ICLK(2);                                                     // [2] window loads
#pragma unroll 1
        for (int z = 0; z < 64; ++z) {                              // 8x8 blocks of the CTU in z-order; a CU is coded at its first block
            const int lx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ly = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
            const ks265_cu8 cu = L.cu[ly * 8 + lx];
            if (cu.log2_cu == 0) continue;                          // outside the picture
            if (PMODE && cu.pred_mode != 2) continue;               // an inter CU: reconstructed already
            const int n8 = 1 << ((cu.log2_cu & 15) - 3);
            if ((lx & (n8 - 1)) || (ly & (n8 - 1))) continue;
            const int n = 8 * n8, log2 = cu.log2_cu & 15;
            c.mode = cu.mvx; c.lx = lx; c.ly = ly; c.x0 = cx * 64 + lx * 8; c.y0 = cy * 64 + ly * 8;
            c.filt = intra_filter_flag(c.mode, n);
            ICLK(3);                                                 // [3] walking the z loop (skipped blocks included)
            ICLK_CNT(16 + (n == 32 ? 2 : n == 16 ? 1 : 0));          // [16..18] CUs of 8 / 16 / 32
            if (n == 32) {
                // ---- 32x32: the luma TU needs all four waves (256 quads) -> work-group barriers; chroma as a second phase
                lds_barrier();                                       // waves 0 / 1 may still be inside a small CU
                const unsigned mask = intra_unit_mask(g, c.x0, c.y0, n, lane);
                if (tid < 3) { L.nz[tid] = 0; L.lastcg[tid] = 0; }
                if (tid < 129) L.raw[0][66 - 64 + tid] = (unsigned char)intra_ref_sample<false>(&L.WY[136 + 4], 136, mask, lx * 8, ly * 8, 32, 8, tid);
                if (tid < 130) {
                    const int cc = 1 + tid / 65, q = tid % 65;
                    L.raw[cc][66 - 32 + q] = (unsigned char)intra_ref_sample<false>(&L.WC[cc - 1][72 + 4], 72, mask, lx * 4, ly * 4, 16, 4, q);
                }
                lds_barrier();
                if (c.filt) {
                    if (tid <= 128) L.fil[66 - 64 + tid] = (unsigned char)intra_filtered(&L.raw[0][66], 32, tid - 64, intra_strong_flat(&L.raw[0][66]));
                } else if (c.mode == 1 && wave < 3) {
                    const int nn = wave ? 16 : 32;
                    const unsigned char *ref = &L.raw[wave][66];
                    int v = lane < nn ? ref[1 + lane] : (lane < 2 * nn ? ref[-1 - (lane - nn)] : 0);
                    v = (int)wave_sum((unsigned)v);
                    if (lane == 0) L.dcv[wave] = (v + nn) >> (wave ? 5 : 6);
                }
                lds_barrier();
                TuRole r;
                r.on = true; r.comp = 0; r.n = 32; r.log2n = 5; r.qx = (tid & 7) * 4; r.qy = tid >> 3;
                tu_pipeline<true>(L, r, c);
                r.on = tid < 128; r.comp = 1 + ((tid >> 6) & 1); r.n = 16; r.log2n = 4; r.qx = (tid & 3) * 4; r.qy = (tid & 63) >> 2;
                tu_pipeline<true>(L, r, c);
                ICLK(4);                                             // [4] a 32x32 CU, everything
                if (tid < 16 && (L.nz[0] | L.nz[1] | L.nz[2])) L.cbf[(ly + (tid >> 2)) * 8 + lx + (tid & 3)] = (L.nz[0] ? 1 : 0) | (L.nz[1] ? 2 : 0) | (L.nz[2] ? 4 : 0);
                lds_barrier();
            } else if (wave < 2) {
                // ---- 8x8 / 16x16: wave 0 codes the luma TU, wave 1 both chroma TUs, each on its own (no work-group barrier: a wave's
                //      gathers only read what the same wave reconstructed, or what was there before the CTU started)
                const unsigned mask = intra_unit_mask(g, c.x0, c.y0, n, lane);
                TuRole r;
                if (wave == 0) {
                    if (lane == 0) { L.nz[0] = 0; L.lastcg[0] = 0; }
                    ICLK(5);                                         // [5] small CU: availability mask
                    for (int q = lane; q <= 4 * n; q += 64) L.raw[0][66 - 2 * n + q] = (unsigned char)intra_ref_sample<false>(&L.WY[136 + 4], 136, mask, lx * 8, ly * 8, n, 8, q);
                    tu_sync<false>();
                    ICLK(6);                                         // [6] reference gather
                    if (c.filt) {
                        for (int q = lane; q <= 4 * n; q += 64) L.fil[66 - 2 * n + q] = (unsigned char)intra_filtered(&L.raw[0][66], n, q - 2 * n, false);
                    } else if (c.mode == 1) {
                        const unsigned char *ref = &L.raw[0][66];
                        int v = lane < n ? ref[1 + lane] : (lane < 2 * n ? ref[-1 - (lane - n)] : 0);
                        v = (int)wave_sum((unsigned)v);
                        if (lane == 0) L.dcv[0] = (v + n) >> (log2 + 1);
                    }
                    tu_sync<false>();
                    const int sh = log2 - 2;                         // quads per row = n / 4
                    r.on = lane < (n * n >> 2); r.comp = 0; r.n = n; r.log2n = log2; r.qx = (lane & ((1 << sh) - 1)) * 4; r.qy = lane >> sh;
                    ICLK(7);                                         // [7] smoothing / DC
#ifdef KS_INTRA_CLOCK
                    unsigned long long tk[9] = {}, tkt = ick_t;
                    tu_pipeline<false>(L, r, c, tk, &tkt);
                    for (int i = 0; i < 9; ++i) g_tclk_acc[i] += tk[i];
#else
                    tu_pipeline<false>(L, r, c);
#endif
                    ICLK(n == 8 ? 8 : 9);                            // [8] / [9] the TU pipeline of an 8x8 / 16x16 luma block
                    if (lane < n8 * n8 && L.nz[0]) atomicOr(&L.cbf[(ly + lane / n8) * 8 + lx + lane % n8], 1);
                } else {
                    const int nc = n >> 1, lenC = 2 * n + 1;
                    if (lane < 2) { L.nz[1 + lane] = 0; L.lastcg[1 + lane] = 0; }
                    for (int q = lane; q < 2 * lenC; q += 64) {
                        const int cc = q >= lenC ? 1 : 0, qq = q - cc * lenC;
                        L.raw[1 + cc][66 - n + qq] = (unsigned char)intra_ref_sample<false>(&L.WC[cc][72 + 4], 72, mask, lx * 4, ly * 4, nc, 4, qq);
                    }
                    tu_sync<false>();
                    if (c.mode == 1 && lane < 2) {
                        const unsigned char *ref = &L.raw[1 + lane][66];
                        int v = nc;
                        for (int i = 0; i < nc; ++i) v += ref[1 + i] + ref[-1 - i];
                        L.dcv[1 + lane] = v >> log2;                 // log2(nc) + 1
                    }
                    tu_sync<false>();
                    const int sh = log2 - 3, nq = nc * nc >> 2;      // quads per row = nc / 4; quads per component
                    const int t = lane < nq ? lane : lane - nq;
                    r.on = lane < 2 * nq; r.comp = lane < nq ? 1 : 2; r.n = nc; r.log2n = log2 - 1; r.qx = (t & ((1 << sh) - 1)) * 4; r.qy = t >> sh;
                    tu_pipeline<false>(L, r, c);
                    if (lane < n8 * n8 && (L.nz[1] | L.nz[2])) atomicOr(&L.cbf[(ly + lane / n8) * 8 + lx + lane % n8], (L.nz[1] ? 2 : 0) | (L.nz[2] ? 4 : 0));
                }
            }
        }
        lds_barrier();
        if (tid < 64) {
            const int bx = cx * 8 + (tid & 7), by = cy * 8 + (tid >> 3);
            if (bx < g.w8 && by < g.h8) cu8[(long)by * g.w8 + bx].cbf = (uint8_t)L.cbf[tid];
        }
        __threadfence();                                             // this CTU's samples are in L2 before the row below is released
        __syncthreads();
        ICLK(11);                                                    // [11] cbf write-back, fence
        ICLK_CNT(19);                                                // [19] CTUs coded
#ifdef KS_INTRA_CLOCK
        if (tid == 0) for (int i = 0; i < 24; ++i) if (ick_a[i]) { atomicAdd(&g_iclk[PMODE ? 1 : 0][i], ick_a[i]); ick_a[i] = 0; }
        if (tid == 0) for (int i = 0; i < 9; ++i) if (g_tclk_acc[i]) { atomicAdd(&g_iclk[PMODE ? 1 : 0][20 + i], g_tclk_acc[i]); g_tclk_acc[i] = 0; }
#endif
        if (tid == 0) __hip_atomic_store(progress + (PMODE ? (int)blockIdx.x : cy), PMODE ? 1 : cx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" int ks265_intra_reconstruct(ks265_frame *f, ks265_pic src, ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !cu8 || !lvl_y || !lvl_u || !lvl_v || !recon.y) return KS265_POINTER;
    if (hipMemsetAsync(f->progress, 0, sizeof(int) * (size_t)f->g.ctu_rows, f->ctx->stream) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
    hipLaunchKernelGGL(intra_recon_kernel<false>, dim3(f->g.ctu_rows), dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, src.y, src.u, src.v, cu8, lvl_y, lvl_u, lvl_v,
                       recon.y, recon.u, recon.v, f->progress, f->ctx->err_dev, f->ctx->wavefront_spin_limit, f->cfg.sdh, 0ll, f->qp_map);
    return ks265_check_launch(f->ctx);
}
// cfg.intra_inter: the intra CUs of a P / B picture, after ks265_reconstruct[_b / _mref] has written the inter CUs into `recon`
extern "C" int ks265_intra_inter_reconstruct(ks265_frame *f, ks265_pic src, ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !cu8 || !lvl_y || !lvl_u || !lvl_v || !recon.y) return KS265_POINTER;
    const int nctu = f->g.ctu_cols * f->g.ctu_rows;
    if (hipMemsetAsync(f->progress, 0, sizeof(int) * (size_t)nctu, f->ctx->stream) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
    const long long lam2k = (long long)f->cfg.lambda_q4 * f->cfg.lambda_q4 * (f->cfg.rdo > 0 ? f->cfg.rdo : 0);
    hipLaunchKernelGGL(intra_recon_kernel<true>, dim3(nctu), dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, src.y, src.u, src.v, cu8, lvl_y, lvl_u, lvl_v,
                       recon.y, recon.u, recon.v, f->progress, f->ctx->err_dev, f->ctx->wavefront_spin_limit, f->cfg.sdh, lam2k, f->qp_map);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ mode pre-selection + CU quadtree (one workgroup per CTU)
typedef int ks_v4i __attribute__((ext_vector_type(4)));

struct DecideLds {
    unsigned char ref[84][2][132];          // [block][raw / smoothed], corner at index 66; blocks: 4 of 32x32, 16 of 16x16, 64 of 8x8 (raster per level)
    unsigned short dc[84];
    unsigned best[4][84];                   // per wave: min over its modes of (cost << 6) | mode
    unsigned cost[85], node[85];            // node: the CU tree's running values (a private array would live in scratch memory)
    unsigned char mode[85], split[85];
    // the source samples every reference array is gathered from: row 0 = the row above the CTU (x = -4 .. 131), rows 1 .. 64 = the CTU with the four samples left
    // of it; sample (x, y) of the CTU at [(1 + y) * 136 + 4 + x] (a byte load per reference sample from HBM / L2 was this kernel's critical path)
    __attribute__((aligned(16))) unsigned char W[65 * 136];
    // round 6: the two MFMA operand sets that do not change over the modes - the Hadamard rows and the source tiles (the same for all four waves) - live here, 16 bytes per lane and
    // set, instead of in 32 registers per lane: at three waves per SIMD those were the 34 dwords the compiler spilled (136 bytes of scratch per lane = 24.7 MiB of HBM writes per launch)
    __attribute__((aligned(16))) int Hm[4][64][4], Sop[4][64][4];
};

__device__ __forceinline__ int intra_mode_bits(int mode) { return (mode == 0 || mode == 1 || mode == 26) ? 3 : 6; }   // default MPM set vs. escape code

// Two uses.  Key pictures (gate_pu null): one work-group per CTU does everything and builds the CU tree.  Candidates of a P / B picture (gate_pu = the CTU's inter
// PU records, ks265_pu / ks265_pu_b: 16 bytes, cost at byte 8; cfg.intra_inter): 19 of the 35 modes, THREE work-groups per CTU - one per level (32 / 16 / 8) - each writes (cost << 6 | mode) of its blocks to best_out (the
// caller sets the array to 0xFFFFFFFF = "no candidate").  Gate: a CTU is evaluated only if one of its 8x8 PUs costs at
// least what an intra CU costs before its first residual bit, lambda x KS_INTRA_GATE_BITS >> 4 (= the bias of the CU decision): where every 8x8 block is predicted
// better than that, no block goes intra - most CTUs of a P / B picture leave here.
// (three waves per SIMD: 168 VGPRs with 33 spilled to scratch beat 207 VGPRs at two waves per SIMD - 205 -> 161 us for the candidates of a 2160p P picture; four waves
//  per SIMD, 75 spills: 170 us)
#define KS_INTRA_GATE_BITS 96
// the gate of the candidates (see intra_decide_kernel) as a kernel of its own, one wave per CTU: the CTUs that pass go into a list (work[0] = how many, work[4 ..] = which; any
// order - every CTU's results are its own).  Round 5: the candidates kernel then finds its heavy work-groups at the FRONT of the grid instead of scattered over it (a work-group
// that evaluates a level lives 50 - 70 us, one that leaves at the gate 2; with a tenth of the CTUs passing, the last heavy one started when the kernel could have ended)
// It also does the memset the stage needed: every CTU's candidates start as "none".  (The list's counter is zeroed by a 16-byte memset in front of it: a ticket per
// work-group of the candidates kernel - the last one to arrive resets the counter - cost 140 us of same-address atomics.)
__global__ __launch_bounds__(256) void intra_gate_kernel(int nctu, int lam, const uint4 *gate_pu, int *work, unsigned *best_out)
{
    const int lane = threadIdx.x & 63, ctu = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ctu >= nctu) return;
    best_out[(long)ctu * 85 + lane] = 0xFFFFFFFFu;
    if (lane < 21) best_out[(long)ctu * 85 + 64 + lane] = 0xFFFFFFFFu;
    const unsigned c = gate_pu[(long)ctu * 85 + 21 + lane].z;                          // cost of 8x8 PU `lane`
    const unsigned long long any = __ballot(c != KS_COST_INVALID && c >= (unsigned)((lam * KS_INTRA_GATE_BITS) >> 4));
    if (lane == 0 && any) work[4 + atomicAdd(&work[0], 1)] = ctu;
}

__global__ __launch_bounds__(256, 3) void intra_decide_kernel(KsGeom g, int lam, const uint8_t *src_y, ks265_cu8 *cu8, unsigned *cost_out, unsigned *best_out, const uint4 *gate_pu, unsigned nlev,
                                                                const int *work)
{
    if (work && (int)(blockIdx.x / nlev) >= work[0]) return;         // candidates: past the list of the CTUs that passed the gate
    __shared__ __attribute__((aligned(16))) DecideLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lsel = gate_pu ? 1 + (int)(blockIdx.x % nlev) : 0;      // lsel: the one level this work-group handles (0 = all); nlev = 3, or 2: no 8x8 candidates
    // the modes by index: key pictures all 35; candidates planar, DC and every second angular mode (2, 4 .. 34: the oracle's INTRA_INTER_MODE_STEP - no measurable
    // loss against all 35)
    const int mi0 = 0, mi1 = gate_pu ? 19 : 35;
    const int ctu = work ? work[4 + blockIdx.x / nlev] : ks_xcd_swizzle(gate_pu ? (int)(blockIdx.x / nlev) : (int)blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const uint8_t *S = ks_org_y(g, src_y);
    if (gate_pu && !work) {
        __shared__ int s_go;
        if (tid < 64) {
            const unsigned c = gate_pu[(long)ctu * 85 + 21 + tid].z;                    // cost of 8x8 PU `tid`
            const unsigned long long any = __ballot(c != KS_COST_INVALID && c >= (unsigned)((lam * KS_INTRA_GATE_BITS) >> 4));
            if (tid == 0) s_go = any != 0ull;
        }
        __syncthreads();
        if (!s_go) return;
    }
    // ---- (1) reference arrays of all 84 blocks from the source picture, raw and smoothed; one block per wave at a time
    {   // the window: row 0 = 34 dwords (x = -4 .. 131), rows 1 .. 64 = 17 dwords (x = -4 .. 63): 1122 dwords, all loads of a thread in flight together
        unsigned v[5];
        int at[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = tid + 256 * k, r = i < 34 ? 0 : 1 + (i - 34) / 17, d = i < 34 ? i : (i - 34) % 17;
            at[k] = i < 34 + 64 * 17 && cx * 64 + 4 * d < g.W + KS_PAD_Y ? r * 136 + 4 * d : -1;
            v[k] = at[k] >= 0 ? *(const unsigned *)(S + (long)(cy * 64 + r - 1) * g.sy + cx * 64 - 4 + 4 * d) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) if (at[k] >= 0) *(unsigned *)&L.W[at[k]] = v[k];
    }
    __syncthreads();
    const unsigned char *const Wo = &L.W[136 + 4] - (long)(cy * 64) * 136 - cx * 64;     // picture coordinates into the window
#pragma unroll 1
    for (int b = wave; b < 84; b += 4) {
        const int l = b < 4 ? 1 : (b < 20 ? 2 : 3), i = b - (l == 1 ? 0 : (l == 2 ? 4 : 20)), n = 64 >> l;
        const int x0 = cx * 64 + (i & ((1 << l) - 1)) * n, y0 = cy * 64 + (i >> l) * n;
        if (lsel && l != lsel) continue;
        if (x0 + n > g.W || y0 + n > g.H) continue;                 // not (completely) inside the picture: never a CU
        const unsigned mask = intra_unit_mask(g, x0, y0, n, lane);
        unsigned char *raw = &L.ref[b][0][66], *fil = &L.ref[b][1][66];
        for (int q = lane; q <= 4 * n; q += 64) raw[q - 2 * n] = (unsigned char)intra_ref_sample<false>(Wo, 136, mask, x0, y0, n, 8, q);
        __builtin_amdgcn_wave_barrier();
        const bool bil = n == 32 && intra_strong_flat(raw);
        for (int q = lane; q <= 4 * n; q += 64) fil[q - 2 * n] = (unsigned char)intra_filtered(raw, n, q - 2 * n, bil);
        int dc = lane < n ? raw[1 + lane] + raw[-1 - lane] : 0;     // DC: the n samples above and the n samples left
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) dc += __shfl_xor(dc, o, 64);
        if (lane == 0) L.dc[b] = (unsigned short)((dc + n) >> (7 - l));              // log2(n) + 1
    }
    __syncthreads();
    // ---- (2) all 35 modes of every block: SATD of (source - prediction) on the matrix cores (see me_subpel_kernel for the operand
    //      layout: lane = (tile column n16, row pair gk); tiles numbered in Z-order so that a block is an aligned lane group)
    const int n16 = lane & 15, gk = lane >> 4;
    int ttx[4], tty[4];                                             // tile coordinates (8-sample units) of this lane's four operand columns
    {
        const unsigned pat = (n16 & 2) ? ((n16 & 1) ? 0x01FFFF01u : 0xFFFF0101u) : ((n16 & 1) ? 0xFF01FF01u : 0x01010101u);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int t = nb * 16 + n16;
            ttx[nb] = (t & 1) | ((t >> 1) & 2) | ((t >> 2) & 4); tty[nb] = ((t >> 1) & 1) | ((t >> 2) & 2) | ((t >> 3) & 4);
        }
        if (wave == 0) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int w = 0; w < 4; ++w) L.Hm[mb][lane][w] = (int)((__popc((mb * 16 + n16) & (gk * 16 + w * 4)) & 1) ? pat ^ 0xFEFEFEFEu : pat);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const uint8_t *p = S + (long)(cy * 64 + tty[nb] * 8 + 2 * gk) * g.sy + cx * 64 + ttx[nb] * 8;
                const uint2 a0 = *(const uint2 *)p, a1 = *(const uint2 *)(p + g.sy);
                L.Sop[nb][lane][0] = (int)(a0.x ^ 0x7F7F7F7Fu); L.Sop[nb][lane][1] = (int)(a0.y ^ 0x7F7F7F7Fu); L.Sop[nb][lane][2] = (int)(a1.x ^ 0x7F7F7F7Fu); L.Sop[nb][lane][3] = (int)(a1.y ^ 0x7F7F7F7Fu);
            }
        }
    }
    __syncthreads();
    const ks_v4i CinN = {0x8000, 0x8000, 0x8000, 0x8000};
    const ks_v4i Cin0 = {gk == 0 ? 0x8000 + 64 : 0x8000, 0x8000, 0x8000, 0x8000};
    const int ltx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), lty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        if (lsel && l != lsel) continue;
        const int n = 64 >> l, log2 = 6 - l, t8 = n >> 3, base = l == 1 ? 0 : (l == 2 ? 4 : 20);
        unsigned best = 0xFFFFFFFFu;
#pragma unroll 1
        for (int mi = mi0 + wave; mi < mi1; mi += 4) {
            const int mode = gate_pu && mi >= 2 ? 2 + (mi - 2) * 2 : mi;
            const int which = intra_filter_flag(mode, n) ? 1 : 0;
            unsigned acc[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int tt = nb * 16 + n16, tx_ = (tt & 1) | ((tt >> 1) & 2) | ((tt >> 2) & 4), ty_ = ((tt >> 1) & 1) | ((tt >> 2) & 2) | ((tt >> 3) & 4);   // (recomputed: eight registers less than keeping ttx / tty)
                const int b = base + (ty_ >> (3 - l)) * (1 << l) + (tx_ >> (3 - l));
                const unsigned char *ref = &L.ref[b][which][66];
                const int ox = (tx_ & (t8 - 1)) * 8, oy = (ty_ & (t8 - 1)) * 8 + 2 * gk, dc = L.dc[b];
                unsigned w[4];
                intra_rows2x8(ref, mode, log2, ox, oy, dc, true, w);
                const ks_v4i B = {(int)(w[0] ^ 0x80808080u), (int)(w[1] ^ 0x80808080u), (int)(w[2] ^ 0x80808080u), (int)(w[3] ^ 0x80808080u)};
                unsigned a = 0;
                asm volatile("" ::: "memory");                              // (the operands are loop-invariant: without this the compiler hoists the loads and keeps all 32 dwords in registers again)
                const ks_v4i sop = *(const ks_v4i *)&L.Sop[nb][lane][0];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const ks_v4i hm = *(const ks_v4i *)&L.Hm[mb][lane][0];
                    ks_v4i C = __builtin_amdgcn_mfma_i32_16x16x64_i8(hm, B, mb == 0 ? Cin0 : CinN, 0, 0, 0);
                    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(hm, sop, C, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_sad_u16((unsigned)C[r], 0x8000u, a);
                }
                acc[nb] = a;
            }
            const bool o1 = gk & 1, o2 = gk & 2;
            const unsigned t0 = (o1 ? acc[1] : acc[0]) + (unsigned)__builtin_amdgcn_ds_swizzle((int)(o1 ? acc[0] : acc[1]), 0x1F | (16 << 10));
            const unsigned t1 = (o1 ? acc[3] : acc[2]) + (unsigned)__builtin_amdgcn_ds_swizzle((int)(o1 ? acc[2] : acc[3]), 0x1F | (16 << 10));
            const unsigned tot = (o2 ? t1 : t0) + (unsigned)__shfl_xor((int)(o2 ? t0 : t1), 32, 64);
            unsigned sd = (tot + 2) >> 2;                               // SATD of tile `lane` (Z-order), had_c normalisation
            if (l <= 2) { sd += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)sd); sd += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)sd); }
            if (l <= 1) { sd += (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)sd); sd += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)sd); }
            const unsigned c = sd + (unsigned)((lam * intra_mode_bits(mode)) >> 4);
            best = min(best, (c << 6) | (unsigned)mode);
        }
        // the first lane of every block publishes this wave's winner
        if ((lane & (t8 * t8 - 1)) == 0) L.best[wave][base + (lty / t8) * (1 << l) + ltx / t8] = best;
    }
    __syncthreads();
    if (tid < 84) {
        const int b = tid, l = b < 4 ? 1 : (b < 20 ? 2 : 3), i = b - (l == 1 ? 0 : (l == 2 ? 4 : 20)), n = 64 >> l;
        const int x0 = cx * 64 + (i & ((1 << l) - 1)) * n, y0 = cy * 64 + (i >> l) * n;
        const unsigned m = min(min(L.best[0][b], L.best[1][b]), min(L.best[2][b], L.best[3][b]));
        const bool inside = x0 + n <= g.W && y0 + n <= g.H;
        L.cost[1 + b] = inside ? m >> 6 : KS_COST_INVALID;           // PU index = 1 + b (level bases 1, 5, 21)
        L.mode[1 + b] = (unsigned char)(m & 63u);
    }
    __syncthreads();
    if (best_out) {                                                  // candidates of a P / B picture: this work-group's level and modes
        if (tid >= 1 && tid < 85) {
            const int b = tid - 1, l = b < 4 ? 1 : (b < 20 ? 2 : 3);
            if (l == lsel && L.cost[tid] != KS_COST_INVALID) best_out[(long)ctu * 85 + tid] = (L.cost[tid] << 6) | (unsigned)L.mode[tid];
        }
        return;
    }
    if (cost_out && tid < 85) cost_out[(long)ctu * 85 + tid] = tid == 0 ? KS_COST_INVALID : L.cost[tid];     // pre-selection costs (lookahead)
    // ---- (3) CU quadtree bottom-up (a node keeps its own cost if it is <= the children's sum + split overhead)
    if (tid == 0) {
        // 85 nodes, leaves first; node value v[idx]
        unsigned *const v = L.node;
        for (int l = 3; l >= 0; --l)
            for (int i = 0; i < (1 << (2 * l)); ++i) {
                const int px = i & ((1 << l) - 1), py = i >> l, s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, idx = ks_pu_index(l, px, py);
                if (x0 >= g.W || y0 >= g.H) { v[idx] = 0; L.split[idx] = 0; continue; }
                const unsigned own = l == 0 ? KS_COST_INVALID : L.cost[idx];
                if (l == 3) { v[idx] = own; L.split[idx] = 0; continue; }
                unsigned long long sum = (unsigned long long)((lam * 12) >> 4);
                for (int k = 0; k < 4; ++k) sum += v[ks_pu_index(l + 1, px * 2 + (k & 1), py * 2 + (k >> 1))];
                if (own != KS_COST_INVALID && (unsigned long long)own <= sum) { v[idx] = own; L.split[idx] = 0; }
                else { v[idx] = sum > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)sum; L.split[idx] = 1; }
            }
    }
    __syncthreads();
    if (tid < 64) {
        const int bx = tid & 7, by = tid >> 3, X = cx * 8 + bx, Y = cy * 8 + by;
        if (X < g.w8 && Y < g.h8) {
            int l = 1, idx = ks_pu_index(1, bx >> 2, by >> 2);
            while (l < 3 && L.split[idx]) { ++l; idx = ks_pu_index(l, bx >> (3 - l), by >> (3 - l)); }
            ks265_cu8 c;
            c.mvx = L.mode[idx]; c.mvy = 0; c.mv1x = 0; c.mv1y = 0; c.log2_cu = (uint8_t)(6 - l); c.cbf = 0; c.pred_mode = 2; c.inter_dir = 0;
            cu8[(long)Y * g.w8 + X] = c;
        }
    }
}

extern "C" int ks265_intra_decide_ex(ks265_frame *f, ks265_pic src, ks265_cu8 *cu8, uint32_t *cost_out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !cu8) return KS265_POINTER;
    hipLaunchKernelGGL(intra_decide_kernel, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(256), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, cu8, cost_out, (unsigned *)nullptr, (const uint4 *)nullptr, 3u, (const int *)nullptr);
    return ks265_check_launch(f->ctx);
}
// cfg.intra_inter: the intra candidates of a P / B picture - per block (85 per CTU, PU indexing) cost << 6 | best mode, 0xFFFFFFFF where there is none (a CTU the
// gate left out, a block outside the picture, the 64x64 level)
extern "C" int ks265_intra_candidates(ks265_frame *f, ks265_pic src, const void *dev_pu_records, uint32_t *dev_best)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !dev_pu_records || !dev_best) return KS265_POINTER;
    static_assert(sizeof(ks265_pu) == 16 && sizeof(ks265_pu_b) == 16 && offsetof(ks265_pu, cost) == 8 && offsetof(ks265_pu_b, cost) == 8, "the gate reads the cost of either record type at byte 8");
    const int nctu = f->g.ctu_cols * f->g.ctu_rows;
    const unsigned nlev = f->cfg.intra_inter >= 2 ? 2u : 3u;        // intra_inter 2: 32x32 and 16x16 candidates only
    if (!f->ic_work && !getenv("KS265_IC_SCATTERED")) {
        if (hipMalloc((void **)&f->ic_work, sizeof(int) * ((size_t)nctu + 4)) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
        if (hipMemsetAsync(f->ic_work, 0, 16, f->ctx->stream) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
    }
    if (f->ic_work && hipMemsetAsync(f->ic_work, 0, 16, f->ctx->stream) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
    if (f->ic_work) hipLaunchKernelGGL(intra_gate_kernel, dim3((nctu + 3) / 4), dim3(256), 0, f->ctx->stream, nctu, f->cfg.lambda_q4, (const uint4 *)dev_pu_records, f->ic_work, dev_best);
    else if (hipMemsetAsync(dev_best, 0xFF, sizeof(uint32_t) * 85 * (size_t)nctu, f->ctx->stream) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
    hipLaunchKernelGGL(intra_decide_kernel, dim3(nctu * nlev), dim3(256), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, (ks265_cu8 *)nullptr, (unsigned *)nullptr, dev_best,
                       (const uint4 *)dev_pu_records, nlev, f->ic_work);
    return ks265_check_launch(f->ctx);
}
extern "C" int ks265_intra_decide(ks265_frame *f, ks265_pic src, ks265_cu8 *cu8) { return ks265_intra_decide_ex(f, src, cu8, nullptr); }

// ------------------------------------------------------------------ lookahead frame cost
// calcFrameCost enc@0x4a7410 / scenecut enc@0x47e9d0 lineage on half-resolution pictures (downsample_c): per 8x8 block the intra
// pre-selection cost against the integer-search cost of the 8x8 PU; out = { sum intra, sum inter, sum min, blocks | intra-cheaper << 32 }
__global__ __launch_bounds__(256) void lookahead_reduce_kernel(long nctu, const unsigned *intra_cost, const ks265_pu *pu, unsigned long long *out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;             // one thread per (CTU, 8x8 PU)
    unsigned long long v[4] = {0, 0, 0, 0};
    if (i < nctu * 64) {
        const long k = (i >> 6) * 85 + 21 + (i & 63);
        const unsigned b = pu[k].cost, a = intra_cost ? intra_cost[k] : b;      // (no intra costs: the sums of the search alone - a block's two costs are valid together, inside the picture)
        if (a != KS_COST_INVALID && b != KS_COST_INVALID) { v[0] = a; v[1] = b; v[2] = min(a, b); v[3] = 1ull | ((unsigned long long)(a < b) << 32); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned lo = wave_sum((unsigned)(v[q] & 0xFFFFFFFFull)), hi = wave_sum((unsigned)(v[q] >> 32));
        // sums of 64 values below 2^26 fit 32 bits; v[3] carries two counters in its halves
        if ((threadIdx.x & 63) == 0) atomicAdd(&out[q], (unsigned long long)lo | ((unsigned long long)hi << 32));
    }
}

extern "C" int ks265_lookahead_reduce(ks265_frame *f, const uint32_t *intra_cost, const ks265_pu *pu, uint64_t *out)
{
    KS_FRAME_CHECK(f);
    if (!pu || !out) return KS265_POINTER;
    const long nctu = (long)f->g.ctu_cols * f->g.ctu_rows;
    if (hipMemsetAsync(out, 0, 32, f->ctx->stream) != hipSuccess) return ks265_hip(f->ctx, hipGetLastError());
    hipLaunchKernelGGL(lookahead_reduce_kernel, dim3((unsigned)((nctu * 64 + 255) / 256)), dim3(256), 0, f->ctx->stream, nctu, intra_cost, pu, (unsigned long long *)out);
    return ks265_check_launch(f->ctx);
}

// the whole low-resolution analysis of one picture: cur / ref are pictures of THIS (half-size) frame object, made with
// ks265_downsample_rect + ks265_pad_picture; cost_ws = nctu x 85 uint32 scratch; uses the frame's cu8 / PU workspace
extern "C" int ks265_lookahead_picture(ks265_frame *f, ks265_pic cur, ks265_pic ref, uint32_t *cost_ws, uint64_t *out)
{
    KS_FRAME_CHECK(f);
    if (!cur.y || !ref.y || !cost_ws || !out) return KS265_POINTER;
    int r;
    if ((r = ks265_intra_decide_ex(f, cur, f->cu8, cost_ws))) return r;
    if ((r = ks265_me_integer(f, cur, ref, nullptr, f->pu[f->cur_pu]))) return r;
    return ks265_lookahead_reduce(f, cost_ws, f->pu[f->cur_pu], out);
}

// the same picture against another reference: its intra costs are in cost_ws from the ks265_lookahead_picture call before (one intra pass per picture; the
// slice-type decision compares a picture with the pictures 1, 4 and 8 back).  cost_ws = null (round 5): no intra costs at all - out[1] = the search's sum as always,
// out[0] = out[2] = the same, no block counted as intra-cheaper: what the slice-type decision alone needs (it reads out[1]), without the 0.3 ms intra pass
extern "C" int ks265_lookahead_inter(ks265_frame *f, ks265_pic cur, ks265_pic ref, const uint32_t *cost_ws, uint64_t *out)
{
    KS_FRAME_CHECK(f);
    if (!cur.y || !ref.y || !out) return KS265_POINTER;
    int r;
    if ((r = ks265_me_integer(f, cur, ref, nullptr, f->pu[f->cur_pu]))) return r;
    return ks265_lookahead_reduce(f, cost_ws, f->pu[f->cur_pu], out);
}
