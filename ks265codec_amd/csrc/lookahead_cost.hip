// lookahead_cost.hip - the reference's lookahead cost function on the device (the core of row f2; round 6):
//   ks265_calc_frame_cost = calcFrameCost enc@0x4a7410: per block of the half-size picture the list-0 / list-1 diamond search from the neighbours' vectors
//                           (meInitPoint enc@0x48af50 + interMeDia enc@0x48fbe0), the bi-predictive average (weightBi_sad_c enc@0x4a7170), seven intra modes
//                           + refinement, the picture sums and the motion statistics
//   ks265_cutree_finish   = the loop inlined in CInputPicManage::updateQueue enc@0x480964..0x480a54: AQ offset - 1.8 log2((propagate + intra') / intra')
// Restated from the disassembly in oracle/ks265_lookahead_ref.c and pinned there on 461 calls recorded inside the reference binary (tests/golden/calc_frame_cost.npz);
// tests/test_gpu_lookahead_ops.py holds these kernels against the recorded outputs and against that oracle bit for bit.
//
// The reference visits the blocks from the last to the first and predicts a block's vector from its right and lower neighbours of the SAME pass: the two searches are
// dependency chains.  Everything else is not, so the function is cut into
//   cfc_intra_kernel    every block at once: the SAD of the seven modes + the +-2 / +-1 refinement (all variants the inter result can select between)
//   cfc_search_kernel   one wave per block row and list, rows bottom-up as a wavefront (a row may start a block when the row below has finished the block under it;
//                       the last column also needs the lower-left one): start point from global memory (three SADs in flight at once), then the diamond walk inside a
//                       window of the reference staged in LDS - a walk step costs LDS latency, not an L2 round trip
//   cfc_combine_kernel  every block at once: bi-predictive SAD, list / bi / intra decision, list bits, inter cost plane, integer sums by atomics (order-free)
//   cfc_final_kernel    one thread: the sums' closing arithmetic (x 100 / 130 for B, the intra-done flag, the return value)
// Roofline: the half-size picture is read ~3 times (2.8 MB at 2160p); the chain kernel is latency-bound by construction (nx + ny dependent steps).
#include "intra_dev.h"
#include "ks265_dev.h"
#include "ks265_internal.h"
#include <cmath>

using namespace ks265;

struct CfcP {
    int w, h, nx, ny, cnt, stride;
    int d0, d1, flag, slice_type;
    int merange, lg, zero_thr, fast_intra, scenecut, preset, p8, aq, b_intra, f3a8, f36c, f538, f3b4;
    int do_list[2], intra_done;
    int mer, m_row, movthr, bigthr;
    unsigned short lam[52];
};
enum { ACC_WINS = 0, ACC_SI, ACC_SIAQ, ACC_S88, ACC_S7C, ACC_C40, ACC_C4C, ACC_C30, ACC_C78, ACC_N = 16 };

__device__ __forceinline__ int mvd_bits(int d)                       // createMvdCostTable enc@0x48b850: signed exp-Golomb length of a quarter-pel difference
{
    const unsigned v = d > 0 ? 2u * (unsigned)d : 1u + 2u * (unsigned)(-d);
    return 1 + 2 * (31 - __clz((int)v));
}
// entry i (relative to the centre of row 12) of the u16 table [52][m_row] the reference indexes without a range test: past the row's ends it reads the neighbouring rows
__device__ __forceinline__ unsigned tab_cost(const CfcP &p, int i)
{
    const int half = p.m_row >> 1;
    if (abs(i) <= half) return (unsigned)(unsigned short)(p.lam[12] * mvd_bits(i));
    const int f = 12 * p.m_row + half + i;
    if (f < 0) return 0xffffu;
    const int row = f / p.m_row, col = f - row * p.m_row;
    return row < 52 ? (unsigned)(unsigned short)(p.lam[row] * mvd_bits(col - half)) : 0xffffu;
}
__device__ __forceinline__ unsigned far_cost(const CfcP &p, int d)   // meInitPoint enc@0x48b220..0x48b24e: the table for |d| <= 256, else lambda x (3 + 2 floor(log2 |d|))
{
    const int a = abs(d);
    if (a <= 0x100) return tab_cost(p, d);
    return (unsigned)(unsigned short)(3 + 2 * (31 - __clz(a))) * (unsigned)p.lam[12];
}
__device__ __forceinline__ int clamp16(int v, int lo, int hi) { return v < lo ? lo : (v <= hi ? v : hi); }   // enc@0x48afce..0x48b02f (lo > hi happens in a partial last row)

// ------------------------------------------------------------------------------------------------------------------------------------ intra (all blocks at once)
// ws[4 blk + v] = SAD << 8 | mode of: v = 0 best of {planar, DC}; 1 best of the first four; 2 best of the seven; 3 the seven refined by +-2, +-1 (full intra only)
// SAD of one intra mode over the block (a function with the reference arrays as arguments, not a lambda that captures them: the captured LDS pointers went through scratch memory
// as generic pointers - 304 bytes per lane at 16 x 16 - and their loads became FLAT loads)
template <int LG, int P>
__device__ __forceinline__ unsigned cfc_sad_mode(const uint8_t *unf, const uint8_t *fil, bool fast_intra, int mode, const int (&f)[P], const int (&xs)[P], const int (&ys)[P], int dc)
{
    const uint8_t *r = (!fast_intra && intra_filter_flag(mode, 1 << LG)) ? fil : unf;   // g_intraNeedFilter enc@0x4df3a0 = the standard's filter rule (rows 8 and 16 checked)
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) s += (unsigned)abs(f[k] - intra_sample(r, mode, LG, xs[k], ys[k], dc, true));
    return wave_sum(s);
}
template <int LG>
__global__ __launch_bounds__(256) void cfc_intra_kernel(CfcP p, const uint8_t *cur, unsigned *ws)
{
    constexpr int BS = 1 << LG, P = BS * BS / 64, C = 2 * BS + 4;
    __shared__ uint8_t unf_[4][4 * BS + 8], fil_[4][4 * BS + 8];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, blk_ = blockIdx.x * 4 + wv;
    const bool live = blk_ < p.nx * p.ny;
    const int blk = live ? blk_ : p.nx * p.ny - 1;
    const int bx = blk % p.nx, by = blk / p.nx, px = bx << LG, py = by << LG;
    const uint8_t *fenc = cur + (long)py * p.stride + px;
    uint8_t *unf = unf_[wv] + C, *fil = fil_[wv] + C;
    // IntraPredLoadRefLeftTopAvaible enc@0x423b20: corner, top row, left column; top-right / lower-left repeat the last sample
    for (int i = lane; i < BS; i += 64) {
        unf[1 + i] = fenc[-p.stride + i]; unf[1 + BS + i] = fenc[-p.stride + BS - 1];
        unf[-1 - i] = fenc[(long)i * p.stride - 1]; unf[-1 - BS - i] = fenc[(long)(BS - 1) * p.stride - 1];
    }
    if (lane == 0) unf[0] = fenc[-p.stride - 1];
    __syncthreads();
    for (int k = -2 * BS + lane; k <= 2 * BS; k += 64)              // IntraPredFilterRef_c enc@0x424110 (no strong filter: 8 / 16 only)
        fil[k] = (uint8_t)((k == -2 * BS || k == 2 * BS) ? unf[k] : (unf[k - 1] + 2 * unf[k] + unf[k + 1] + 2) >> 2);
    __syncthreads();
    int f[P], xs[P], ys[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { const int idx = lane + 64 * k; ys[k] = idx >> LG; xs[k] = idx & (BS - 1); f[k] = fenc[(long)ys[k] * p.stride + xs[k]]; }
    int dc = BS;
    for (int i = 0; i < BS; ++i) dc += unf[1 + i] + unf[-1 - i];
    dc >>= (LG + 1);
    const bool fast = p.fast_intra != 0;
#define sad_mode(m) cfc_sad_mode<LG, P>(unf, fil, fast, (m), f, xs, ys, dc)
    unsigned best = 0xfffffffu; int bm = 0;
    unsigned out[4];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int mode = i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 26 : i == 3 ? 10 : i == 4 ? 18 : i == 5 ? 2 : 34;
        const unsigned s = sad_mode(mode);
        if (s < best) { best = s; bm = mode; }
        if (i == 1) out[0] = best << 8 | (unsigned)bm;
        if (i == 3) out[1] = best << 8 | (unsigned)bm;
    }
    out[2] = best << 8 | (unsigned)bm;
    if (!p.fast_intra) {                                             // enc@0x4a863e..0x4a8818
        int centre = bm, curm = bm;
#pragma unroll
        for (int step = 2; step >= 1; --step) {
            int m2 = centre + step; curm = centre;
            if ((unsigned)(m2 - 3) <= 31u) { const unsigned s = sad_mode(m2); if (s < best) { best = s; curm = m2; } }
            m2 = centre - step;
            if ((unsigned)(m2 - 3) <= 31u) { const unsigned s = sad_mode(m2); if (s < best) { best = s; curm = m2; } }
            centre = curm;
        }
        bm = curm;
    }
    out[3] = best << 8 | (unsigned)bm;
    if (live && lane < 4) ws[4 * blk + lane] = lane == 0 ? out[0] : (lane == 1 ? out[1] : (lane == 2 ? out[2] : out[3]));
#undef sad_mode
}

// ------------------------------------------------------------------------------------------------------------------------------------ the search chains
// One wave per block row and list; the row walks right to left.  What a block needs from its neighbours is ONE word each - the vector of the block to its right (this wave's own,
// in a register) and of the block below (the wave of the row below; the last column also the lower-left one) - so the vector plane itself is the progress mark: it is filled
// with the reference's own "not there yet" word 0x7fff before the launch, a block's vector is stored with release order after its cost, and the row above polls that word.
// Everything else of a block is in LDS before its turn comes: while block bx is being searched the window of the reference around block bx - 1 (the block's place +- CFC_RB
// samples) and its source samples are on their way (registers, written to the other LDS buffer at the end of the turn).  Start points, the zero vector and the diamond walk read
// that window; a candidate or a walk that leaves it (a vector beyond +- CFC_RB) falls back to global loads / a window re-centred on the walk.  A turn is then one L2 round trip
// (the poll) + LDS arithmetic instead of four dependent round trips: 2.18 -> see profiles/r06_crf.txt for a 1920 x 1080 half-size picture (120 + 68 turns).
#define CFC_SENTINEL 0x7fff
#define CFC_RB 20
template <int LG>
__global__ __launch_bounds__(64) void cfc_search_kernel(CfcP p, const uint8_t *cur, const uint8_t *ref0, const uint8_t *ref1, int32_t *mv0, int32_t *c0, int32_t *mv1, int32_t *c1,
                                                        int l_first, unsigned *err_word, int spin_limit)
{
    constexpr int BS = 1 << LG, P = BS * BS / 64, RB = CFC_RB, WS = BS + 2 * RB, WP = WS + 4, WD = WP / 4, ND = (WS * WD + 63) / 64;
    __shared__ __attribute__((aligned(16))) uint8_t win[2][WS * WP];
    const int lane = threadIdx.x, l = l_first + (int)blockIdx.y, by = p.ny - 1 - (int)blockIdx.x, nx = p.nx, py = by << LG;
    const uint8_t *ref = l ? ref1 : ref0;
    int32_t *mv = l ? mv1 : mv0, *cs = l ? c1 : c0;
    const int thr = (int)((unsigned)p.zero_thr << (2 * LG)) >> 1;
    const int iters = p.mer >> (p.preset > 1 ? 0 : (p.p8 != 4));
    int xs[P], ys[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { const int idx = lane + 64 * k; ys[k] = idx >> LG; xs[k] = idx & (BS - 1); }
    // window of the block at picture column px: origin (ox, oy), ox dword-aligned in memory
    auto win_ox = [&](int px_) { const int x = px_ - RB; return x - (int)((uintptr_t)(ref + x) & 3u); };
    unsigned pre[ND]; int fpre[P];
    auto fetch = [&](int px_) {                                    // the loads of a block's window and source samples (all in flight together)
        const uint8_t *src = ref + (long)(py - RB) * p.stride + win_ox(px_);
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int q = lane + 64 * k, yy = q / WD, xx = q - yy * WD;
            pre[k] = q < WS * WD ? *(const unsigned *)(src + (long)yy * p.stride + 4 * xx) : 0u;
        }
        const uint8_t *fe = cur + (long)py * p.stride + px_;
#pragma unroll
        for (int k = 0; k < P; ++k) fpre[k] = fe[(long)ys[k] * p.stride + xs[k]];
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < ND; ++k) { const int q = lane + 64 * k; if (q < WS * WD) *(unsigned *)&win[buf][4 * q] = pre[k]; }
    };
    int cb = 0, right = 0;
    fetch((nx - 1) << LG); commit(0);
    int f[P];
#pragma unroll
    for (int k = 0; k < P; ++k) f[k] = fpre[k];
    __syncthreads();
    for (int bx = nx - 1; bx >= 0; --bx) {
        const int px = bx << LG, blk = by * nx + bx;
        if (bx > 0) fetch((bx - 1) << LG);                          // the next turn's data: under way while this block is searched
        int ox = win_ox(px), oy = py - RB;                          // the window in win[cb]
        const uint8_t *W = win[cb];
        auto inside = [&](int X, int Y) { return X >= ox && X + BS <= ox + WP && Y >= oy && Y + BS <= oy + WS; };
        auto sad_at = [&](int X, int Y) -> unsigned {              // SAD of the source block against the reference block at picture position (X, Y); wave-uniform
            unsigned s = 0;
            if (inside(X, Y)) {
                const uint8_t *w = W + (Y - oy) * WP + (X - ox);
#pragma unroll
                for (int k = 0; k < P; ++k) s += (unsigned)abs(f[k] - (int)w[ys[k] * WP + xs[k]]);
            } else {
                const uint8_t *g = ref + (long)Y * p.stride + X;
#pragma unroll
                for (int k = 0; k < P; ++k) s += (unsigned)abs(f[k] - (int)g[(long)ys[k] * p.stride + xs[k]]);
            }
            return wave_sum(s);
        };
        // ---- the neighbours' vectors (enc@0x4a7dc0..0x4a7e73: the first two of {right, below, lower-left, lower-right} that exist become the AMVP pair)
        int cand[2] = {0, 0};
        if (by < p.ny - 1) {
            auto poll = [&](const int32_t *q) -> int {
                int v, spins = 0;
                while ((v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == CFC_SENTINEL) {     // (relaxed: the word itself is all that is read of the other row)
                    if (spins++ >= spin_limit) { __hip_atomic_fetch_or(err_word, KS_DEVERR_WAVEFRONT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); v = 0; break; }
                }
                return v;
            };
            const int below = poll(mv + blk + nx);
            if (bx < nx - 1) { cand[0] = right; cand[1] = below; }
            else { cand[0] = below; if (bx > 0) cand[1] = poll(mv + blk + nx - 1); }
        } else if (bx < nx - 1) cand[0] = right;
        // ---- meInitPoint enc@0x48af50 (no CTU object: no look-ahead vector, no stored candidates)
        const int lim0 = -px, lim1 = p.w - px - BS, lim2 = -py, lim3 = p.h - py - BS;
        int mvpx[2], mvpy[2], cx[2], cy[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            mvpx[k] = (short)(cand[k] & 0xffff); mvpy[k] = cand[k] >> 16;
            cx[k] = clamp16((mvpx[k] + 2) >> 2, lim0, lim1); cy[k] = clamp16((mvpy[k] + 2) >> 2, lim2, lim3);
        }
        const bool same = cx[0] == cx[1] && cy[0] == cy[1];
        auto sad_raw = [&](int X, int Y) -> unsigned {              // this lane's part of sad_at
            unsigned t = 0;
            if (inside(X, Y)) {
                const uint8_t *w2 = W + (Y - oy) * WP + (X - ox);
#pragma unroll
                for (int k = 0; k < P; ++k) t += (unsigned)abs(f[k] - (int)w2[ys[k] * WP + xs[k]]);
            } else {
                const uint8_t *g = ref + (long)Y * p.stride + X;
#pragma unroll
                for (int k = 0; k < P; ++k) t += (unsigned)abs(f[k] - (int)g[(long)ys[k] * p.stride + xs[k]]);
            }
            return t;
        };
        unsigned s0, s1;
        if (same) s0 = s1 = sad_at(px + cx[0], py + cy[0]);
        else { const unsigned t = wave_sum(sad_raw(px + cx[0], py + cy[0]) | (sad_raw(px + cx[1], py + cy[1]) << 16)); s0 = t & 0xffffu; s1 = t >> 16; }
        const int i = same ? 0 : (s0 + 0u > s1 + 1u);              // index costs tME+0x2e0 / +0x2e4 = 0 / 1 (enc@0x4a7e7d..0x4a7e8a)
        const unsigned sad = i ? s1 : s0;
        const bool zero_tried = (cx[0] == 0 && cy[0] == 0) || (cx[1] == 0 && cy[1] == 0);
        int mx = cx[i], my = cy[i];
        const int pxq = mvpx[i], pyq = mvpy[i], cmx = -pxq, cmy = -pyq;
        int w;
        w = (pxq >> 2) - p.mer; const int win0 = (short)(w >= lim0 ? w : lim0);
        w = (pxq >> 2) + p.mer; const int win1 = (short)(w > lim1 ? lim1 : w);
        w = (pyq >> 2) - p.mer; const int win2 = (short)(w >= lim2 ? w : lim2);
        w = (pyq >> 2) + p.mer; const int win3 = (short)(w > lim3 ? lim3 : w);
        bool outside = true;
        if (mx >= win0 && mx <= win1 && my >= win2) outside = my > win3;
        unsigned cost = outside ? sad + (far_cost(p, 4 * mx - pxq) + far_cost(p, 4 * my - pyq)) : sad + (tab_cost(p, cmx + 4 * mx) + tab_cost(p, cmy + 4 * my));
        if (!((unsigned long long)sad < (unsigned long long)(long long)thr)) {      // enc@0x4a7f30..0x4a7f50
            if (!zero_tried) {                                                       // enc@0x4a8198..0x4a82db
                const unsigned cz = sad_at(px, py) + far_cost(p, -pyq) + far_cost(p, -pxq);
                if (cz < cost) { cost = cz; mx = my = 0; }
            }
            // interMeDia enc@0x48fbe0: the four neighbours of the centre per step
            unsigned b = cost << 4;
            for (int it = 0; it < iters; ++it) {
                const int X = px + mx, Y = py + my;
                if (!(X - 1 >= ox && X + 1 + BS <= ox + WP && Y - 1 >= oy && Y + 1 + BS <= oy + WS)) {      // the walk has reached the window's rim: a window around where it is now
                    __syncthreads();
                    ox = X - RB - (int)((uintptr_t)(ref + (X - RB)) & 3u); oy = Y - RB;
                    const uint8_t *src = ref + (long)oy * p.stride + ox;
                    for (int q = lane; q < WS * WD; q += 64) { const int yy = q / WD, xx = q - yy * WD; *(unsigned *)&win[cb][4 * q] = *(const unsigned *)(src + (long)yy * p.stride + 4 * xx); }
                    __syncthreads();
                }
                unsigned su = 0, sd = 0, sl = 0, sr = 0;
                const uint8_t *wc = W + (Y - oy) * WP + (X - ox);
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const uint8_t *q = wc + ys[k] * WP + xs[k];
                    su += (unsigned)abs(f[k] - (int)q[-WP]); sd += (unsigned)abs(f[k] - (int)q[WP]);
                    sl += (unsigned)abs(f[k] - (int)q[-1]); sr += (unsigned)abs(f[k] - (int)q[1]);
                }
                {   // two sums per reduction: a SAD of at most 256 samples fits 16 bits
                    const unsigned a = wave_sum(su | (sd << 16)), c2 = wave_sum(sl | (sr << 16));
                    su = a & 0xffffu; sd = a >> 16; sl = c2 & 0xffffu; sr = c2 >> 16;
                }
                const unsigned tx = tab_cost(p, cmx + 4 * mx), ty = tab_cost(p, cmy + 4 * my);
                b = min(b, ((su + tx + tab_cost(p, cmy + 4 * (my - 1))) << 4) + 1u);
                b = min(b, ((sd + tx + tab_cost(p, cmy + 4 * (my + 1))) << 4) + 3u);
                b = min(b, ((sl + tab_cost(p, cmx + 4 * (mx - 1)) + ty) << 4) + 4u);
                b = min(b, ((sr + tab_cost(p, cmx + 4 * (mx + 1)) + ty) << 4) + 12u);
                if (!(b & 15u)) break;
                mx -= (int)(b << 28) >> 30; my -= (int)(b << 30) >> 30;
                b &= ~15u;
            }
            cost = b >> 4;
        }
        const int packed = (int)((unsigned)(unsigned short)(short)(mx << 2) | ((unsigned)(unsigned short)(short)(my << 2) << 16));
        right = packed;
        if (lane == 0) {
            cs[blk] = (int)cost;
            __hip_atomic_store(mv + blk, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the vector is the progress mark of the row above (its cost is read by the next kernel only)
        }
        if (bx > 0) {                                               // the next turn's window and source samples have arrived by now
            __syncthreads();
            commit(cb ^ 1); cb ^= 1;
#pragma unroll
            for (int k = 0; k < P; ++k) f[k] = fpre[k];
            __syncthreads();
        }
    }
}

// the vector planes of the lists about to be searched start as "not there yet" (the reference's own mark, 0x7fff in the first word: here in every word - the search polls them)
__global__ __launch_bounds__(256) void cfc_mark_kernel(int32_t *mv0, int32_t *mv1, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (mv0) mv0[i] = CFC_SENTINEL;
    if (mv1) mv1[i] = CFC_SENTINEL;
}

// ------------------------------------------------------------------------------------------------------------------------------------ decision + sums (all blocks at once)
template <int LG>
__global__ __launch_bounds__(256) void cfc_combine_kernel(CfcP p, const uint8_t *cur, const uint8_t *ref0, const uint8_t *ref1, const int32_t *mv0, const int32_t *c0, const int32_t *mv1,
                                                          const int32_t *c1, const unsigned *ws_intra, uint16_t *intra, uint8_t *imode, const uint16_t *invq, uint16_t *inter,
                                                          uint8_t *bits, int *acc)
{
    constexpr int BS = 1 << LG, P = BS * BS / 16;
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = blockIdx.x * 4 + (threadIdx.x >> 6), n = p.nx * p.ny;
    if (grp * 4 >= n) return;
    const int blk = grp * 4 + (lane >> 4);
    const bool valid = blk < n;
    const int bq = valid ? blk : n - 1, bx = bq % p.nx, by = bq / p.nx, px = bx << LG, py = by << LG;
    const bool any_inter = (p.d0 + p.d1) != 0, has1 = p.d1 > 0;
    const int m0 = p.d0 ? mv0[bq] : 0, m1 = has1 ? mv1[bq] : 0;
    const int m0x = (short)(m0 & 0xffff), m0y = m0 >> 16, m1x = (short)(m1 & 0xffff), m1y = m1 >> 16;
    unsigned bcost; int bitsv;
    if (!any_inter) { bcost = 0x10000003u; bitsv = 0; }
    else {
        bcost = 0xfffffffu; int bestbits = 0;
        { const unsigned lc = (unsigned)c0[bq]; if (bcost > lc) { bcost = lc; bestbits = 1; } }
        if (has1) { const unsigned lc = (unsigned)c1[bq]; if (bcost > lc) { bcost = lc; bestbits = 2; } }
        if (has1) {                                                  // enc@0x4a8910..0x4a8a01: SAD against the rounded average of the two predictions
            const uint8_t *fe = cur + (long)py * p.stride + px, *a = ref0 + (long)(py + (m0y >> 2)) * p.stride + px + (m0x >> 2), *b = ref1 + (long)(py + (m1y >> 2)) * p.stride + px + (m1x >> 2);
            unsigned s = 0;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int idx = sub + 16 * k, y = idx >> LG, x = idx & (BS - 1);
                const long o = (long)y * p.stride + x;
                s += (unsigned)abs((((int)a[o] + (int)b[o] + 1) >> 1) - (int)fe[o]);
            }
            s = group_sum<16>(s);
            if (s + 5u >= bcost) { bcost += 4u; bitsv = bestbits; } else { bcost = s + 9u; bitsv = 3; }
        } else { bcost += 4u; bitsv = bestbits; }
    }
    int wins = 0, si = 0, siaq = 0;
    const bool counted = (bx > 0 && bx < p.nx - 1 && by > 0 && by < p.ny - 1) ? true : (p.nx <= 2 ? true : p.ny <= 2);
    if (!(has1 && !p.b_intra)) {
        unsigned icost;
        if (!p.intra_done) {
            int v = p.fast_intra ? (bcost < ((((unsigned)((int)((unsigned)p.zero_thr << (2 * LG)) >> 1)) >> 1) << p.fast_intra) ? 1 : 2) : 3;
            if (p.scenecut == 0 && p.preset <= 1 && bcost < (1u << (2 * LG))) v = 0;
            const unsigned e = ws_intra[4 * bq + v];
            icost = (e >> 8) + 9u;
            if (valid && sub == 0) { intra[bq] = (uint16_t)min(icost, 0xffffu); imode[bq] = (uint8_t)(e & 255u); }
            if (counted) { si = (int)icost; if (p.aq) siaq = (int)(((unsigned)invq[bq] * icost + 128u) >> 8); }
        } else icost = intra[bq];
        if (icost < bcost) { bcost = icost; wins = 1; }
    }
    if (any_inter && valid && sub == 0) inter[bq] = (uint16_t)min(bcost, 0xffffu);
    const int v88 = counted ? (int)bcost : 0;
    int v7c = v88;
    if (p.aq) v7c = ((int)((unsigned)invq[bq] * (unsigned)v88) + 128) >> 8;
    int c40 = 0, c4c = 0, c30 = 0, c78 = 0;
    if (p.slice_type != 2) {
        const int ax = abs(m0x), ay = abs(m0y), s = ax + ay;
        if (p.f3a8 != 0 || p.f36c == 2 || p.f538 != 0) { if (s > 2) c40 = p.f3a8 != 0; c4c = ((ax >> 6) + (ay >> 6)) > 0; }
        c78 = p.movthr <= s;
        if (p.f3b4) {
            if (p.d0 != 0 && p.bigthr < abs(m0x >> 2) + abs(m0y >> 2)) c30 = 1;
            else if (p.d1 != 0 && p.bigthr < abs(m1x >> 2) + abs(m1y >> 2)) c30 = 1;
        }
    }
    // one lane per block carries its contribution; the wave's four blocks are summed and lane 0 issues the atomics
    const bool lead = valid && sub == 0;
    int vals[9] = {wins, si, siaq, v88, v7c, c40, c4c, c30, c78};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int t = lead ? vals[k] : 0;
        t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
        if (lane == 0 && t) atomicAdd(&acc[k], t);
    }
    const int b1 = __shfl(bitsv, 16, 64), b2 = __shfl(bitsv, 32, 64), b3 = __shfl(bitsv, 48, 64);
    if (lane == 0) {
        const int nb = min(4, n - grp * 4);
        unsigned byte = bits[grp];
        const int bv[4] = {bitsv, b1, b2, b3};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nb) byte = (byte & ~(3u << (2 * k))) | ((unsigned)bv[k] << (2 * k));
        bits[grp] = (uint8_t)byte;
    }
}

__global__ void cfc_final_kernel(CfcP p, int *acc, ks265_cfc_sums *s)
{
    const int idx = p.d0 * 9 + p.d1;
    int wins = s->intra_wins, si = s->sum_intra, siaq = s->sum_intra_aq;
    if (p.d1 == 0 && !p.intra_done) { wins = 0; si = -1; siaq = -1; }               // enc@0x4a8084..0x4a80b3
    wins += acc[ACC_WINS]; si += acc[ACC_SI]; siaq += acc[ACC_SIAQ];
    int s88 = acc[ACC_S88];
    const int s7c = acc[ACC_S7C];
    if (p.slice_type != 2) { s->stats[0] = acc[ACC_C40]; s->stats[1] = acc[ACC_C4C]; s->stats[2] = acc[ACC_C30]; s->stats[3] = acc[ACC_C78]; }
    if (p.d1 != 0) s88 = (int)((unsigned)s88 * 100u) / 130; else s->intra_done = 1;    // enc@0x4a8a83..0x4a8aad
    if (idx) { s->sum = s88; s->sum_aq = s7c; } else { si = s88; siaq = s7c; s->sum = s88; s->sum_aq = s7c; }
    s->intra_wins = wins; s->sum_intra = si; s->sum_intra_aq = siaq;
    int cost = s88;
    if (p.flag) cost += (int)(((unsigned)wins / (unsigned)(p.cnt * 8)) * (unsigned)cost);
    s->ret = cost;
    for (int k = 0; k < ACC_N; ++k) acc[k] = 0;
}

// the cuTree finish: thread = block
__constant__ double kCfcLog2[128];
__global__ __launch_bounds__(256) void cutree_finish_kernel(int cnt, const uint16_t *intra, const uint16_t *invq, const uint16_t *prop, const double *aq_off, int dbl, double *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    const int iw = ((int)((unsigned)intra[i] * (unsigned)invq[i]) + 128) >> 8;
    if (iw == 0) return;
    unsigned pr = prop[i];
    if (dbl) pr *= 2;
    auto lg2 = [](unsigned x) -> double { const int lz = __clz((int)x); return kCfcLog2[((x << lz) >> 24) & 127u] + (double)(31 - lz); };   // _log2 enc@0x4c3c20
    double prod = 1.8 * (lg2(pr + (unsigned)iw) - lg2((unsigned)iw));                 // mulsd, then subsd (enc@0x480a17..0x480a20): two roundings, never a fused multiply-add
    asm volatile("" : "+v"(prod));                                                      // (hipcc contracts a * b - c across __dmul_rn / __dsub_rn as well; this keeps the product a value of its own)
    const double q = aq_off[i] - prod;
    out[i] = -15.0 > q ? -15.0 : (q < 20.0 ? q : 20.0);
}


// ---- helpers of the host's cuTree pass -----------------------------------------------------------------------------------------------------------------------------------------
// edge replication around a w x h plane (the reference pads its half-size pictures by 32; the search may leave the picture by merange / 2 + 1 and the block grid may overhang it)
__global__ __launch_bounds__(256) void pad_plane_kernel(uint8_t *p00, int stride, int w, int h, int pad)
{
    const int W2 = w + 2 * pad, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W2 * (h + 2 * pad)) return;
    const int y = i / W2 - pad, x = i % W2 - pad;
    if (x >= 0 && x < w && y >= 0 && y < h) return;
    p00[(long)y * stride + x] = p00[(long)min(max(y, 0), h - 1) * stride + min(max(x, 0), w - 1)];
}
__global__ __launch_bounds__(256) void fill_u16_kernel(uint16_t *d, int n, int v) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = (uint16_t)v; }
// one QP per CTU from the per-block offsets (blocks of 2^(lg + 1) luma samples: 4 x 4 or 2 x 2 per CTU): base + clip(round(mean, summed in raster order), +-12), clipped to [lo, hi] -
// this build's rule (quantisation group = CTU), the same as ks265_aq_ctu_map's
__global__ __launch_bounds__(64) void qoff_ctu_map_kernel(const double *off, int nx, int ny, int bpc, int cols, int rows, int base_qp, int lo, int hi, int8_t *map)
{
    const int ctu = blockIdx.x * 64 + threadIdx.x;
    if (ctu >= cols * rows) return;
    const int cx = ctu % cols, cy = ctu / cols;
    double sum = 0.0; int cnt = 0;
    for (int by = cy * bpc; by < min(cy * bpc + bpc, ny); ++by)
        for (int bx = cx * bpc; bx < min(cx * bpc + bpc, nx); ++bx) { sum += off[by * nx + bx]; ++cnt; }
    int d = cnt ? (int)floor(sum / (double)cnt + 0.5) : 0;
    d = d < -12 ? -12 : d > 12 ? 12 : d;
    const int q = base_qp + d;
    map[ctu] = (int8_t)(q < lo ? lo : q > hi ? hi : q);
}

extern "C" {

size_t ks265_calc_frame_cost_workspace(int nx, int ny)
{
    return (size_t)ACC_N * sizeof(int) + (size_t)2 * (size_t)ny * sizeof(int) + (size_t)4 * (size_t)nx * (size_t)ny * sizeof(unsigned) + 256;
}

int ks265_calc_frame_cost(ks265_ctx *ctx, const ks265_cfc_params *q, const uint8_t *dev_cur, const uint8_t *dev_ref0, const uint8_t *dev_ref1, uint16_t *dev_intra, uint8_t *dev_imode,
                          const uint16_t *dev_inv_qscale, uint16_t *dev_inter, uint8_t *dev_list_bits, int32_t *dev_mv0, int32_t *dev_cost0, int32_t *dev_mv1, int32_t *dev_cost1,
                          ks265_cfc_sums *dev_sums, void *dev_ws)
{
    if (!ctx || !q || !dev_cur || !dev_intra || !dev_imode || !dev_list_bits || !dev_sums || !dev_ws) return KS265_POINTER;
    if (q->lg != 3 && q->lg != 4) return KS265_NOTSUPPORTED;
    if (!q->fast_intra && q->scenecut == 0 && q->preset <= 1) return KS265_NOTSUPPORTED;          /* two intra modes AND the refinement: no preset combines them */
    if (q->stride & 3) return KS265_NOTSUPPORTED;                                                      /* the search stages windows of the planes by dwords */
    if (q->nx <= 0 || q->ny <= 0 || q->d0 < 0 || q->d1 < 0 || (q->d0 == 0 && q->d1 != 0)) return KS265_NOTSUPPORTED;
    if ((q->d0 && (!dev_ref0 || !dev_mv0 || !dev_cost0)) || (q->d1 && (!dev_ref1 || !dev_mv1 || !dev_cost1)) || ((q->d0 + q->d1) && !dev_inter) || (q->aq && !dev_inv_qscale)) return KS265_POINTER;
    ks_use_device(ctx);
    CfcP p{};
    p.w = q->w; p.h = q->h; p.nx = q->nx; p.ny = q->ny; p.cnt = q->cnt; p.stride = q->stride; p.d0 = q->d0; p.d1 = q->d1; p.flag = q->flag; p.slice_type = q->slice_type;
    p.merange = q->merange; p.lg = q->lg; p.zero_thr = q->zero_thr; p.fast_intra = q->fast_intra; p.scenecut = q->scenecut; p.preset = q->preset; p.p8 = q->p8; p.aq = q->aq;
    p.b_intra = q->b_intra; p.f3a8 = q->f3a8; p.f36c = q->f36c; p.f538 = q->f538; p.f3b4 = q->f3b4;
    p.do_list[0] = q->d0 ? q->do_list[0] : 0; p.do_list[1] = q->d1 ? q->do_list[1] : 0; p.intra_done = q->intra_done;
    p.mer = q->merange >> 1; if (p.mer > 32) p.mer = 32;                                      // enc@0x4a75fb..0x4a7609
    p.m_row = 8 * q->merange + 33;
    static const int kTbl[9] = {2, 2, 3, 3, 4, 4, 6, 6, 8};
    p.movthr = q->d0 <= 8 ? (kTbl[q->d0] * 12) >> 1 : 48;                                      // enc@0x4a8065..0x4a807b
    double t = (double)((q->w + q->h) * 2) / 656.0;                                           // enc@0x4a77ba..0x4a7842
    if (t >= 2.0) t = t * t * 0.5; else if (q->preset > 4) t = t * t * 0.75;
    p.bigthr = (int)(t * 4.0);
    for (int i = 0; i < 52; ++i) p.lam[i] = q->lambda_tab[i];
    const int n = p.nx * p.ny;
    int *acc = (int *)dev_ws;
    unsigned *ws_intra = (unsigned *)(acc + ACC_N + 2 * p.ny);
    hipStream_t st = ctx->stream;
    const bool need_intra = !p.intra_done && !(p.d1 > 0 && !p.b_intra);
    const int nl = (p.do_list[0] ? 1 : 0) + (p.do_list[1] ? 1 : 0), l_first = p.do_list[0] ? 0 : 1;
    if (hipMemsetAsync(dev_ws, 0, (size_t)ACC_N * sizeof(int), st) != hipSuccess) return ks265_hip(ctx, hipGetLastError());
    if (nl) hipLaunchKernelGGL(cfc_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p.do_list[0] ? dev_mv0 : (int32_t *)nullptr, p.do_list[1] ? dev_mv1 : (int32_t *)nullptr, n);
#define CFC_LAUNCH(LG) do { \
        if (need_intra) hipLaunchKernelGGL(cfc_intra_kernel<LG>, dim3((n + 3) / 4), dim3(256), 0, st, p, dev_cur, ws_intra); \
        if (nl) hipLaunchKernelGGL(cfc_search_kernel<LG>, dim3(p.ny, nl), dim3(64), 0, st, p, dev_cur, dev_ref0, dev_ref1, dev_mv0, dev_cost0, dev_mv1, dev_cost1, l_first, \
                                   ctx->err_dev, ctx->wavefront_spin_limit); \
        hipLaunchKernelGGL(cfc_combine_kernel<LG>, dim3((n + 15) / 16), dim3(256), 0, st, p, dev_cur, dev_ref0, dev_ref1, (const int32_t *)dev_mv0, (const int32_t *)dev_cost0, \
                           (const int32_t *)dev_mv1, (const int32_t *)dev_cost1, (const unsigned *)ws_intra, dev_intra, dev_imode, dev_inv_qscale, dev_inter, dev_list_bits, acc); \
    } while (0)
    if (p.lg == 3) CFC_LAUNCH(3); else CFC_LAUNCH(4);
#undef CFC_LAUNCH
    hipLaunchKernelGGL(cfc_final_kernel, dim3(1), dim3(1), 0, st, p, acc, dev_sums);
    return ks265_check_launch(ctx);
}

int ks265_cutree_finish(ks265_ctx *ctx, int cnt, const uint16_t *dev_intra, const uint16_t *dev_inv_qscale, const uint16_t *dev_propagate, const double *dev_aq_off, int dbl, double *dev_out)
{
    if (!ctx || !dev_intra || !dev_inv_qscale || !dev_propagate || !dev_aq_off || !dev_out) return KS265_POINTER;
    if (cnt <= 0) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    {
        static bool done[64];
        if (ctx->device < 0 || ctx->device >= 64) return KS265_FAIL;
        if (!done[ctx->device]) {
            double t[128];
            for (int i = 0; i < 128; ++i) t[i] = std::round(std::log2((128.0 + i) / 128.0) * 1e5) / 1e5;
            if (hipMemcpyToSymbol(HIP_SYMBOL(kCfcLog2), t, sizeof t) != hipSuccess) return KS265_FAIL;
            done[ctx->device] = true;
        }
    }
    hipLaunchKernelGGL(cutree_finish_kernel, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, cnt, dev_intra, dev_inv_qscale, dev_propagate, dev_aq_off, dbl, dev_out);
    return ks265_check_launch(ctx);
}

int ks265_pad_plane(ks265_ctx *ctx, uint8_t *dev_p00, int stride, int w, int h, int pad)
{
    if (!ctx || !dev_p00) return KS265_POINTER;
    if (w <= 0 || h <= 0 || pad < 0 || stride < w + 2 * pad) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    const int n = (w + 2 * pad) * (h + 2 * pad);
    hipLaunchKernelGGL(pad_plane_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dev_p00, stride, w, h, pad);
    return ks265_check_launch(ctx);
}
int ks265_fill_u16(ks265_ctx *ctx, uint16_t *dev, int n, int value)
{
    if (!ctx || !dev) return KS265_POINTER;
    if (n <= 0) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    hipLaunchKernelGGL(fill_u16_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dev, n, value);
    return ks265_check_launch(ctx);
}
int ks265_qoff_ctu_map(ks265_ctx *ctx, const double *dev_off, int nx, int ny, int lg, int ctu_cols, int ctu_rows, int base_qp, int qp_lo, int qp_hi, int8_t *dev_map)
{
    if (!ctx || !dev_off || !dev_map) return KS265_POINTER;
    if (nx <= 0 || ny <= 0 || (lg != 3 && lg != 4) || ctu_cols <= 0 || ctu_rows <= 0 || qp_lo > qp_hi) return KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    const int n = ctu_cols * ctu_rows;
    hipLaunchKernelGGL(qoff_ctu_map_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, dev_off, nx, ny, 64 >> (lg + 1), ctu_cols, ctu_rows, base_qp, qp_lo, qp_hi, dev_map);
    return ks265_check_launch(ctx);
}

}  // extern "C"
