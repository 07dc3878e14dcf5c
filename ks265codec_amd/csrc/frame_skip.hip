// frame_skip.hip — Stage D2: the skip pass (cfg.skip_rd; round 6)
// The reference decides skip / merge on the distortion the coded block really has (skipFastDecision enc@0x486090, skipFullMergeDecision enc@0x482da0, tuDecisionSkipMerge
// enc@0x482990 under processCuMdInter enc@0x485800: closed code).  The CU tree and the merge pass judge on Hadamard cost + rate, which cannot know what a residual costs and
// what it buys.  This pass runs AFTER the reconstruction of the inter CUs, where both sides of the comparison exist.  Every inter CU that carries residual:
//   J_cur  = SSE(source, reconstruction) of Y + 4 (Cb + Cr)  +  lambda x (bits of the CU's levels + 6)
//   J_skip = min over the first two DISTINCT merge candidates k (of A1 B1 B0 A0 B2 of H.265 8.5.3.2.3 + the zero vector; from a snapshot of the CU map, as in the merge pass:
//            no order between CUs or CTUs)  of  SSE(source, prediction with k's motion)  +  lambda x (1 + position of k) bits
// J_skip < J_cur: the CU becomes a 2Nx2N CU without residual carrying k's motion - levels cleared, reconstruction = the prediction (exactly what the decoder predicts).
// Bits of levels as the coefficient-group pruning prices them (rdo_level_q2 + 10 + 16 - n per 4x4 group, quarter bits); lambda = 1.5 (lambda_q4 / 16)^2, so
// J x 1024 = SSE << 10 + (lambda_q4^2 x 24 >> 4) x quarter bits.  (The form first built - the same test on every node of the quadtree, uniting CUs - cost four times the
// evaluations for 0.1 % of the bytes: the oracle's header has the measurements.)
#include "frame_common.h"
#include "pred_dev.h"
#include "interp_dev.h"

using namespace ks265;

#ifndef KS_SKIP_OCC
#define KS_SKIP_OCC 2          // waves per SIMD the register allocation aims at
#endif
#define SP_SYN_CODED 24         // syntax of a CU with residual, quarter bits
#define SP_SKIP_BASE 4
#define SP_SKIP_POS 4
#define SP_LAMBDA_Q4 24         // lambda scale in sixteenths
#define SP_CHROMA_W 16          // weight of the chroma distortion in quarters
#define SP_CANDS 2              // distinct candidates tried

struct KsSkipRefs { const uint8_t *y0[4], *u0[4], *v0[4], *y1[4], *u1[4], *v1[4]; };
struct SpMotion { int dir8, mvx, mvy, mv1x, mv1y; bool ok; };

__device__ __forceinline__ const uint8_t *sp_pick(const uint8_t *const (&p)[4], int i) { return i == 0 ? p[0] : (i == 1 ? p[1] : (i == 2 ? p[2] : p[3])); }
__device__ __forceinline__ int sp_z_of_8(int x, int y)
{
    const int bx = (x >> 3) & 7, by = (y >> 3) & 7;
    return (bx & 1) | ((by & 1) << 1) | ((bx & 2) << 1) | ((by & 2) << 2) | ((bx & 4) << 2) | ((by & 4) << 3);
}
// candidate k of the node (x, y, n) from the snapshot: A1 B1 B0 A0 B2, 5 = zero.  Vectors of lists the motion does not use are 0; MR: the neighbour's pictures come with its motion
template <bool MR>
__device__ __forceinline__ SpMotion sp_cand(const KsGeom &g, const ks265_cu8 *snap, int x, int y, int n, int k, bool bi_zero)
{
    SpMotion m; m.dir8 = bi_zero ? 3 : 1; m.mvx = m.mvy = m.mv1x = m.mv1y = 0; m.ok = true;
    if (k == 5) return m;
    const int nx = k == 1 ? x + n - 1 : k == 2 ? x + n : x - 1, ny = k == 0 ? y + n - 1 : k == 3 ? y + n : y - 1;
    m.ok = false;
    if (nx < 0 || ny < 0 || nx >= g.W || ny >= g.H) return m;
    const int ctb = (y >> 6) * g.ctu_cols + (x >> 6), nctb = (ny >> 6) * g.ctu_cols + (nx >> 6);
    if (nctb > ctb || (nctb == ctb && sp_z_of_8(nx, ny) >= sp_z_of_8(x, y))) return m;
    const ks265_cu8 c = snap[(long)(ny >> 3) * g.w8 + (nx >> 3)];
    if (c.pred_mode != 0 || (c.log2_cu & 15) < 3) return m;
    m.dir8 = MR ? (int)c.inter_dir : (c.inter_dir & 3);
    const int dir = m.dir8 & 3;
    m.mvx = (dir & 1) ? c.mvx : 0; m.mvy = (dir & 1) ? c.mvy : 0; m.mv1x = (dir & 2) ? c.mv1x : 0; m.mv1y = (dir & 2) ? c.mv1y : 0;
    m.ok = true;
    if ((dir & 1) && (x + (c.mvx >> 2) < -70 || x + (c.mvx >> 2) + n > g.W + 70 || y + (c.mvy >> 2) < -70 || y + (c.mvy >> 2) + n > g.H + 70)) m.ok = false;
    if ((dir & 2) && (x + (c.mv1x >> 2) < -70 || x + (c.mv1x >> 2) + n > g.W + 70 || y + (c.mv1y >> 2) < -70 || y + (c.mv1y >> 2) + n > g.H + 70)) m.ok = false;
    return m;
}
__device__ __forceinline__ bool sp_same(const SpMotion &a, const SpMotion &b) { return a.dir8 == b.dir8 && a.mvx == b.mvx && a.mvy == b.mvy && a.mv1x == b.mv1x && a.mv1y == b.mv1y; }

// ---- the prediction of a lane's samples.  Every input row is filtered horizontally once and feeds all output rows that tap it (interp_dev.h's scheme), in RAW form so that
// uni- and bi-prediction share it:   kind 0: v = sample   1: one filter pass, v = tap sum (scale 64)   2: both passes, v = vertical taps over the 16-bit horizontal sums.
//   uni (interp*8to8 / 16to8): 0: v   1: clip8((v + 32) >> 6)   2: clip8((v + 2048) >> 12)        14-bit for bi (interp*8to16 / 16to16): 0: v << 6   1: v   2: v >> 6
// The kind is chosen per WAVE (all lanes integer / all without vertical fraction / all without horizontal fraction / else both passes): a pass with the fraction-0 taps
// {.. 64 ..} multiplies by 64, which the rounding of the higher kind takes out again exactly ((64 a + 2048) >> 12 = (a + 32) >> 6, (4096 p + 2048) >> 12 = p), so a lane whose
// own vector has fewer fractions gets the same samples - and the lanes of a wave (nodes with different vectors) do not serialise three code paths.  Called by ALL lanes.
__device__ __forceinline__ int sp_luma_raw(const uint8_t *p, long st, int fx, int fy, int (&v)[4][8])
{
    if (__all(!(fx | fy))) {
#pragma unroll
        for (int r = 0; r < 4; ++r) luma_row8(p + r * st, v[r]);
        return 0;
    }
    int tl, th;
    luma_taps_packed(fx, tl, th);
    if (__all(!fy)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) luma_hrow8(p + r * st, tl, th, v[r]);
        return 1;
    }
    int cy[8];
    luma_taps(fy, cy);
    const bool hor = !__all(!fx);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[r][i] = 0;
#pragma unroll
    for (int ir = 0; ir < 11; ++ir) {                                // input row ir - 3
        int h[8];
        if (hor) {
            luma_hrow8(p + (ir - 3) * st, tl, th, h);
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = (int)(short)h[i];     // the reference's 16-bit intermediate
        } else luma_row8(p + (ir - 3) * st, h);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = ir - r;
            if (t < 0 || t > 7) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[r][i] += cy[t] * h[i];
        }
    }
    return hor ? 2 : 1;
}
// chroma: 4 rows x 4 samples, 4-tap filters at 1/8 sample (interpChroma* enc@0x4111c0..)
__device__ __forceinline__ void sp_chroma_hrow(const uint8_t *row, int taps, int (&h)[4])
{
    const uint8_t *q = row - 1;
    const unsigned sh = (unsigned)((uintptr_t)q & 3);
    const unsigned *a = (const unsigned *)(q - sh);
    const unsigned a0 = a[0] ^ 0x80808080u, a1 = a[1] ^ 0x80808080u, a2 = a[2] ^ 0x80808080u;
    const unsigned w0 = align_bytes(a1, a0, sh), w1 = align_bytes(a2, a1, sh);   // bytes -1..2, 3..6
    h[0] = __builtin_amdgcn_sdot4((int)w0, taps, 8192, false);
    h[1] = __builtin_amdgcn_sdot4((int)align_bytes(w1, w0, 1), taps, 8192, false);
    h[2] = __builtin_amdgcn_sdot4((int)align_bytes(w1, w0, 2), taps, 8192, false);
    h[3] = __builtin_amdgcn_sdot4((int)align_bytes(w1, w0, 3), taps, 8192, false);
}
__device__ __forceinline__ int sp_chroma_raw(const uint8_t *p, long st, int fx, int fy, int (&v)[4][4])
{
    if (__all(!(fx | fy))) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { unsigned w; __builtin_memcpy(&w, p + r * st, 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[r][i] = (int)((w >> (8 * i)) & 255u); }
        return 0;
    }
    const int taps = pack_taps4(kChromaTaps[fx]);
    if (__all(!fy)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sp_chroma_hrow(p + r * st, taps, v[r]);
        return 1;
    }
    const bool hor = !__all(!fx);
    int cy[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) cy[t] = kChromaTaps[fy][t];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[r][i] = 0;
#pragma unroll
    for (int ir = 0; ir < 7; ++ir) {                                 // input row ir - 1
        int h[4];
        if (hor) {
            sp_chroma_hrow(p + (ir - 1) * st, taps, h);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = (int)(short)h[i];
        } else { unsigned w; __builtin_memcpy(&w, p + (ir - 1) * st, 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = (int)((w >> (8 * i)) & 255u); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = ir - r;
            if (t < 0 || t > 3) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[r][i] += cy[t] * h[i];
        }
    }
    return hor ? 2 : 1;
}
// a lane's luma samples (4 rows x 8, packed) for the motion m: pA = the picture of the list a uni-directional motion uses (list 0's for bi), pB = list 1's picture (bi)
__device__ __forceinline__ void sp_pred_luma(const uint8_t *pA, const uint8_t *pB, long st, int X, int Y, const SpMotion &m, uint2 (&out)[4])
{
    const int dir = m.dir8 & 3;
    const int ax = (dir & 1) ? m.mvx : m.mv1x, ay = (dir & 1) ? m.mvy : m.mv1y;
    int v[4][8];
    const int k = sp_luma_raw(pA + (long)(Y + (ay >> 2)) * st + X + (ax >> 2), st, ax & 3, ay & 3, v);
    if (__any(dir == 3)) {
        unsigned a14[4][4];                                        // list 0 as 14-bit pairs
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) a14[r][i] = ((unsigned)to14(k, v[r][2 * i]) & 0xFFFFu) | ((unsigned)to14(k, v[r][2 * i + 1]) << 16);
        int u[4][8];
        const bool bi = dir == 3;
        const int bx = bi ? m.mv1x : 0, by = bi ? m.mv1y : 0;
        const int k1 = sp_luma_raw(pB + (long)(Y + (by >> 2)) * st + X + (bx >> 2), st, bx & 3, by & 3, u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int px[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int a = (int)(short)((i & 1) ? (a14[r][i >> 1] >> 16) : (a14[r][i >> 1] & 0xFFFFu));
                px[i] = bi ? clip8(ks_no_pk((a + (int)(short)to14(k1, u[r][i]) + 64) >> 7)) : uni_round(k, v[r][i]);
            }
            out[r] = ks_pack_row8(px);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int px[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) px[i] = uni_round(k, v[r][i]);
        out[r] = ks_pack_row8(px);
    }
}
__device__ __forceinline__ void sp_pred_chroma(const uint8_t *pA, const uint8_t *pB, long st, int X, int Y, const SpMotion &m, unsigned (&out)[4])
{
    const int dir = m.dir8 & 3;
    const int ax = (dir & 1) ? m.mvx : m.mv1x, ay = (dir & 1) ? m.mvy : m.mv1y;
    int v[4][4];
    const int k = sp_chroma_raw(pA + (long)(Y + (ay >> 3)) * st + X + (ax >> 3), st, ax & 7, ay & 7, v);
    int u[4][4];
    int k1 = 0;
    const bool any_bi = __any(dir == 3), bi = dir == 3;
    if (any_bi) {
        const int bx = bi ? m.mv1x : 0, by = bi ? m.mv1y : 0;
        k1 = sp_chroma_raw(pB + (long)(Y + (by >> 3)) * st + X + (bx >> 3), st, bx & 7, by & 7, u);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        unsigned w = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = (any_bi && bi) ? clip8(ks_no_pk(((int)(short)to14(k, v[r][i]) + (int)(short)to14(k1, u[r][i]) + 64) >> 7)) : uni_round(k, v[r][i]);
            w |= (unsigned)px << (8 * i);
        }
        out[r] = w;
    }
}
__device__ __forceinline__ unsigned sp_sse4(unsigned a, unsigned b)
{
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int d = (int)((a >> (8 * i)) & 255) - (int)((b >> (8 * i)) & 255); s += (unsigned)(d * d); }
    return s;
}
// quarter bits and count of the non-zero levels among four packed s16
__device__ __forceinline__ void sp_lvl4(uint2 w, int &bits, int &cnt)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned ww = i < 2 ? w.x : w.y;
        const int l = (int)(short)((i & 1) ? (ww >> 16) : (ww & 0xFFFFu));
        if (l) { bits += rdo_level_q2(l < 0 ? -l : l); ++cnt; }
    }
}

// One work-group per CTU, 128 lanes: lane = (8x8 tile in z-order) x 2 + half; a half holds four luma rows of the tile (two 4x4 coefficient groups) and one of its two 4x4
// chroma blocks (half 0: Cb, half 1: Cr).  A CU of 8 / 16 / 32 / 64 samples is 2 / 8 / 32 / 128 consecutive lanes: CU sums by cross-lane adds (every lane takes the sum of
// its own CU's size), a CU of 64 through LDS.
template <bool MR>
__global__ __launch_bounds__(128, KS_SKIP_OCC) void skip_pass_kernel(KsGeom g, long long lam2, int bi_zero_, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, const KsSkipRefs R,
                                                                     const ks265_cu8 *snap, ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v)
{
    __shared__ unsigned red[16][2];
    __shared__ unsigned keep[12][128];                             // the winning candidate's samples of every lane (lane-private columns: no hand-over between lanes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, z = tid >> 1, sub = tid & 1;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int tx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ty = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;                // the lane's tile
    const bool inside = x0 < g.W && y0 < g.H, bi_zero = bi_zero_ != 0;
    ks265_cu8 c;
    c.mvx = c.mvy = c.mv1x = c.mv1y = 0; c.log2_cu = 0; c.cbf = 0; c.pred_mode = 1; c.inter_dir = 0;
    if (inside) c = snap[(long)(y0 >> 3) * g.w8 + (x0 >> 3)];
    const int log2c = c.log2_cu & 15;
    const bool inter = inside && c.pred_mode == 0 && log2c >= 3;
    int slot = 0;
    // sum of v over the lanes of the lane's CU (called by all lanes of the work-group: the 64x64 case goes through LDS)
    auto cusum = [&](unsigned v) -> unsigned {
        const unsigned s3 = v + (unsigned)__shfl_xor((int)v, 1, 64);
        unsigned s2 = s3 + (unsigned)__shfl_xor((int)s3, 2, 64); s2 += (unsigned)__shfl_xor((int)s2, 4, 64);
        unsigned s1 = s2 + (unsigned)__shfl_xor((int)s2, 8, 64); s1 += (unsigned)__shfl_xor((int)s1, 16, 64);
        const unsigned sw = s1 + (unsigned)__shfl_xor((int)s1, 32, 64);
        const int sl = slot; slot = (slot + 1) & 15;
        if (lane == 0) red[sl][wave] = sw;
        __syncthreads();
        const unsigned s0 = red[sl][0] + red[sl][1];
        return log2c >= 6 ? s0 : log2c == 5 ? s1 : log2c == 4 ? s2 : s3;
    };
    const int n = inter ? 1 << log2c : 8, cux = x0 & ~(n - 1), cuy = y0 & ~(n - 1);     // the lane's CU
    const bool eval = inter && cusum((inter && sub == 0 && c.cbf) ? 1u : 0u) != 0;      // an inter CU with residual
    // the lane's samples: luma rows 4 sub .. 4 sub + 3 of the tile, the 4x4 block of Cb (sub 0) or Cr (sub 1)
    const uint8_t *Sy = ks_org_y(g, src_y), *Sc = ks_org_c(g, sub ? src_v : src_u);
    uint8_t *Ry = ks_org_y(g, rec_y), *Rc = ks_org_c(g, sub ? rec_v : rec_u);
    int16_t *lvl_c = sub ? lvl_v : lvl_u;
    const int ly = inside ? y0 + 4 * sub : 0, lx = inside ? x0 : 0, cxx = lx >> 1, cyy = inside ? (y0 >> 1) : 0;     // (a lane outside the picture predicts - unused - samples at the origin)
    uint2 sY[4];
    unsigned sC[4];
    unsigned dY = 0, dC = 0, bits = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { sY[r] = make_uint2(0u, 0u); sC[r] = 0u; }
    if (eval) {
        int bl = 0, cl = 0, br = 0, cr = 0, bc = 0, cc = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint2 q = *(const uint2 *)(Ry + (long)(ly + r) * g.sy + lx);
            sY[r] = *(const uint2 *)(Sy + (long)(ly + r) * g.sy + lx);
            dY += sp_sse4(sY[r].x, q.x) + sp_sse4(sY[r].y, q.y);
            sC[r] = *(const unsigned *)(Sc + (long)(cyy + r) * g.sc + cxx);
            dC += sp_sse4(sC[r], *(const unsigned *)(Rc + (long)(cyy + r) * g.sc + cxx));
            // bits of the levels per 4x4 group: the lane's luma rows are two groups (left, right), its chroma block one
            const uint4 w = *(const uint4 *)(lvl_y + (long)(ly + r) * g.W + lx);
            sp_lvl4(make_uint2(w.x, w.y), bl, cl); sp_lvl4(make_uint2(w.z, w.w), br, cr);
            sp_lvl4(*(const uint2 *)(lvl_c + (long)(cyy + r) * (g.W / 2) + cxx), bc, cc);
        }
        if (cl) bits += (unsigned)(bl + 10 + 16 - cl);
        if (cr) bits += (unsigned)(br + 10 + 16 - cr);
        if (cc) bits += (unsigned)(bc + 10 + 16 - cc);
    }
    const unsigned ndY = cusum(dY), ndC = cusum(dC), nbits = cusum(bits);
    const unsigned long long jcur = ((unsigned long long)(ndY + ((unsigned)(((unsigned long long)ndC * SP_CHROMA_W) >> 2))) << 10) + (unsigned long long)(lam2 * (long long)(nbits + SP_SYN_CODED));
    // ---- the candidates: which exist, their positions, which repeat an earlier one; the first SP_CANDS distinct ones are tried
    unsigned valid = 0, distinct = 0;
    {
        SpMotion mm[6];
        int nd = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            mm[k] = sp_cand<MR>(g, snap, cux, cuy, n, k, bi_zero);
            const bool ok = eval && mm[k].ok;
            bool rep = false;
#pragma unroll
            for (int j = 0; j < k; ++j) rep |= ((valid >> j) & 1u) && sp_same(mm[j], mm[k]);
            valid |= (ok ? 1u : 0u) << k;
            if (ok && !rep && nd < SP_CANDS) { distinct |= 1u << k; ++nd; }
        }
    }
    unsigned long long best = jcur;
    int bestk = -1;
    // the lane's prediction for the motion of candidate k (a lane that is not `on` gets the zero vector of list 0: valid addresses, result unused)
    auto motion_of = [&](bool on, int k) -> SpMotion {
        SpMotion m = sp_cand<MR>(g, snap, cux, cuy, n, k, bi_zero);
        if (!on) { m.dir8 = 1; m.mvx = m.mvy = m.mv1x = m.mv1y = 0; }
        return m;
    };
#pragma unroll 1
    for (int it = 0; it < SP_CANDS; ++it) {
        const bool on = (int)__popc(distinct) > it;
        if (!__syncthreads_or(on ? 1 : 0)) break;                  // nobody in the CTU has that many candidates
        int k = 0;
        { unsigned d = distinct; for (int i = 0; i < it; ++i) d &= d - 1u; k = d ? __ffs((int)d) - 1 : 0; }
        const SpMotion m = motion_of(on, k);
        const int dir = m.dir8 & 3, i0 = MR ? (m.dir8 >> 4) & 3 : 0, i1 = MR ? (m.dir8 >> 6) & 3 : 0;
        unsigned eY = 0;
        uint2 pY[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pY[r] = make_uint2(0u, 0u);
        if (__any(on)) {
            sp_pred_luma(ks_org_y(g, (dir & 1) ? sp_pick(R.y0, i0) : sp_pick(R.y1, i1)), ks_org_y(g, sp_pick(R.y1, i1)), g.sy, lx, ly, m, pY);
#pragma unroll
            for (int r = 0; r < 4; ++r) eY += sp_sse4(sY[r].x, pY[r].x) + sp_sse4(sY[r].y, pY[r].y);
        }
        const unsigned nY = cusum(on ? eY : 0u);
        const int pos = __popc(valid & ((1u << k) - 1u));
        const unsigned long long rate = (unsigned long long)(lam2 * (long long)(SP_SKIP_BASE + SP_SKIP_POS * pos));
        // the chroma samples only where the luma distortion + rate alone does not already lose
        const bool need_c = on && (((unsigned long long)nY << 10) + rate < best);
        if (!__syncthreads_or(need_c ? 1 : 0)) continue;
        unsigned eC = 0;
        unsigned pC[4] = {0u, 0u, 0u, 0u};
        if (__any(need_c)) {
            sp_pred_chroma(ks_org_c(g, (dir & 1) ? (sub ? sp_pick(R.v0, i0) : sp_pick(R.u0, i0)) : (sub ? sp_pick(R.v1, i1) : sp_pick(R.u1, i1))),
                           ks_org_c(g, sub ? sp_pick(R.v1, i1) : sp_pick(R.u1, i1)), g.sc, cxx, cyy, m, pC);
#pragma unroll
            for (int r = 0; r < 4; ++r) eC += sp_sse4(sC[r], pC[r]);
        }
        const unsigned nC = cusum(need_c ? eC : 0u);
        if (need_c) {
            const unsigned long long j = ((unsigned long long)(nY + ((unsigned)(((unsigned long long)nC * SP_CHROMA_W) >> 2))) << 10) + rate;
            if (j < best) {
                best = j; bestk = k;
#pragma unroll
                for (int r = 0; r < 4; ++r) { keep[2 * r][tid] = pY[r].x; keep[2 * r + 1][tid] = pY[r].y; keep[8 + r][tid] = pC[r]; }
            }
        }
    }
    // ---- a CU that drops its residual: its lanes write the samples they kept and clear their levels, the tile's first lane the record
    if (bestk >= 0) {
        const SpMotion m = sp_cand<MR>(g, snap, cux, cuy, n, bestk, bi_zero);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            *(uint2 *)(Ry + (long)(ly + r) * g.sy + lx) = make_uint2(keep[2 * r][tid], keep[2 * r + 1][tid]);
            *(uint4 *)(lvl_y + (long)(ly + r) * g.W + lx) = make_uint4(0u, 0u, 0u, 0u);
            *(unsigned *)(Rc + (long)(cyy + r) * g.sc + cxx) = keep[8 + r][tid];
            *(uint2 *)(lvl_c + (long)(cyy + r) * (g.W / 2) + cxx) = make_uint2(0u, 0u);
        }
        if (sub == 0) {
            ks265_cu8 o;
            o.mvx = (int16_t)m.mvx; o.mvy = (int16_t)m.mvy; o.mv1x = (int16_t)m.mv1x; o.mv1y = (int16_t)m.mv1y;
            o.log2_cu = (uint8_t)log2c; o.cbf = 0; o.pred_mode = 0; o.inter_dir = (uint8_t)m.dir8;
            cu8[(long)(y0 >> 3) * g.w8 + (x0 >> 3)] = o;
        }
    }
}

// dev_cu8: the picture's CU map after ks265_reconstruct* (cbf set); the frame's spare map takes the snapshot the candidates are read from
extern "C" int ks265_skip_pass(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, ks265_cu8 *dev_cu8, int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !src.u || !src.v || !ref0.y || !ref0.u || !ref0.v || !dev_cu8 || !dev_lvl_y || !dev_lvl_u || !dev_lvl_v || !recon.y || !recon.u || !recon.v) return KS265_POINTER;
    if (!f->cu8_tmp) return KS265_NOTSUPPORTED;                          // the frame object was created without cfg.merge / cfg.skip_rd
    if (dev_cu8 == f->cu8_tmp) return KS265_POINTER;
    int r = ks265_hip(f->ctx, hipMemcpyAsync(f->cu8_tmp, dev_cu8, (size_t)f->geom.bytes_cu8, hipMemcpyDeviceToDevice, f->ctx->stream));
    if (r) return r;
    const bool is_b = ref1.y != nullptr || (f->mrefb && !f->mr_pslice);
    KsSkipRefs R;
    for (int i = 0; i < 4; ++i) {
        const ks265_pic a = f->mrefb ? f->mr_pic[0][i] : ref0, b = f->mrefb ? f->mr_pic[1][i] : (ref1.y ? ref1 : ref0);
        R.y0[i] = a.y; R.u0[i] = a.u; R.v0[i] = a.v; R.y1[i] = b.y; R.u1[i] = b.u; R.v1[i] = b.v;
    }
    const long long lam2 = ((long long)f->cfg.lambda_q4 * f->cfg.lambda_q4 * SP_LAMBDA_Q4) >> 4;
    const int nctu = f->g.ctu_cols * f->g.ctu_rows;
    if (f->mrefb) hipLaunchKernelGGL(skip_pass_kernel<true>, dim3(nctu), dim3(128), 0, f->ctx->stream, f->g, lam2, is_b ? 1 : 0, src.y, src.u, src.v, R, f->cu8_tmp, dev_cu8, dev_lvl_y, dev_lvl_u, dev_lvl_v, recon.y, recon.u, recon.v);
    else hipLaunchKernelGGL(skip_pass_kernel<false>, dim3(nctu), dim3(128), 0, f->ctx->stream, f->g, lam2, is_b ? 1 : 0, src.y, src.u, src.v, R, f->cu8_tmp, dev_cu8, dev_lvl_y, dev_lvl_u, dev_lvl_v, recon.y, recon.u, recon.v);
    return ks265_check_launch(f->ctx);
}
