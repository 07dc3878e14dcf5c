// frame_skip.hip — Stage D2: the skip pass (cfg.skip_rd; round 6)
// The reference decides skip / merge / CU size on the distortion the coded block really has (skipFastDecision enc@0x486090, skipFullMergeDecision enc@0x482da0,
// tuDecisionSkipMerge enc@0x482990 under processCuMdInter enc@0x485800: closed code).  The CU tree and the merge pass judge on Hadamard cost + rate, which cannot know
// that a residual quantises to nothing.  This pass runs AFTER the reconstruction of the inter CUs, where both sides of the comparison exist.  Per CTU, top-down over the
// nodes of 64 / 32 / 16 / 8 samples that lie inside the picture and hold inter CUs only:
//   J_cur  = SSE(source, reconstruction) of Y + 4 (Cb + Cr)  +  lambda x (bits of the node's levels + the syntax of its CUs)
//   J_skip = min over the node's merge candidates k (A1 B1 B0 A0 B2 of H.265 8.5.3.2.3 at the node's geometry + the zero vector; from a snapshot of the CU map, as in
//            the merge pass: no order between nodes or CTUs)  of  SSE(source, prediction with k's motion)  +  lambda x (1 + position of k) bits
// J_skip < J_cur: the node becomes ONE CU without residual carrying k's motion - levels cleared, reconstruction = the prediction (formed by pred_dev.h: exactly what the
// decoder predicts), descendants not looked at.  Bits are estimates: levels as the coefficient-group pruning prices them (rdo_level_q2 + 10 + 16 - n per 4x4 group, quarter
// bits), a CU without residual 2 bits, with residual 6, 1.25 per CU below the node; lambda_mode = (lambda_q4 / 16)^2, so J x 1024 = SSE << 10 + lambda_q4^2 x quarter bits.
//
// One work-group per CTU, 256 lanes: lane = (8x8 tile in z-order) x 4 + quarter; a quarter holds two luma rows of the tile and one row of its 4x4 Cb and Cr blocks.  A node
// of level 3 / 2 / 1 / 0 is 4 / 16 / 64 / 256 consecutive lanes: node sums by cross-lane adds, level 0 through LDS.
#include "frame_common.h"
#include "pred_dev.h"

using namespace ks265;

#define SP_SYN_SKIP 8
#define SP_SYN_CODED 24
#define SP_SYN_BELOW 5
#define SP_SKIP_BASE 4
#define SP_SKIP_POS 4
#define SP_CHROMA_W 16          // weight of the chroma distortion in quarters

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

// prediction of 4 adjacent samples of one component at (X, Y) in that component's samples: packed bytes
template <bool LUMA>
__device__ __forceinline__ unsigned sp_pred4(const uint8_t *o0, const uint8_t *o1, long st, int X, int Y, int dir, int mvx, int mvy, int mv1x, int mv1y)
{
    constexpr int SH = LUMA ? 2 : 3, MK = LUMA ? 3 : 7;
    int p[4];
    if (dir == 3) {
        int v0[4], v1[4], k0, k1;
        if (LUMA) { k0 = luma_raw4(o0 + (long)(Y + (mvy >> SH)) * st + X + (mvx >> SH), st, mvx & MK, mvy & MK, v0); k1 = luma_raw4(o1 + (long)(Y + (mv1y >> SH)) * st + X + (mv1x >> SH), st, mv1x & MK, mv1y & MK, v1); }
        else { k0 = chroma_raw4(o0 + (long)(Y + (mvy >> SH)) * st + X + (mvx >> SH), st, mvx & MK, mvy & MK, v0); k1 = chroma_raw4(o1 + (long)(Y + (mv1y >> SH)) * st + X + (mv1x >> SH), st, mv1x & MK, mv1y & MK, v1); }
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = clip8(((int)(short)to14(k0, v0[i]) + (int)(short)to14(k1, v1[i]) + 64) >> 7);
    } else {
        const int ux = dir == 2 ? mv1x : mvx, uy = dir == 2 ? mv1y : mvy;
        const uint8_t *rp = (dir == 2 ? o1 : o0) + (long)(Y + (uy >> SH)) * st + X + (ux >> SH);
        int v[4];
        const int k = LUMA ? luma_raw4(rp, st, ux & MK, uy & MK, v) : chroma_raw4(rp, st, ux & MK, uy & MK, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = uni_round(k, v[i]);
    }
    return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
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

template <bool MR>
__global__ __launch_bounds__(256) void skip_pass_kernel(KsGeom g, long long lam2, int deepest, int bi_zero_, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, const KsSkipRefs R,
                                                        const ks265_cu8 *snap, ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v)
{
    __shared__ unsigned red[32][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, z = tid >> 2, sub = tid & 3;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int tx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ty = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;                // the lane's tile
    const bool inside = x0 < g.W && y0 < g.H, bi_zero = bi_zero_ != 0;
    int slot = 0;
    // node sum of v at level l (uniform l): every lane of the node gets it.  Level 0 goes through LDS - called by all lanes of the work-group
    auto nsum = [&](unsigned v, int l) -> unsigned {
        v += (unsigned)__shfl_xor((int)v, 1, 64); v += (unsigned)__shfl_xor((int)v, 2, 64);
        if (l <= 2) { v += (unsigned)__shfl_xor((int)v, 4, 64); v += (unsigned)__shfl_xor((int)v, 8, 64); }
        if (l <= 1) { v += (unsigned)__shfl_xor((int)v, 16, 64); v += (unsigned)__shfl_xor((int)v, 32, 64); }
        if (l == 0) {
            const int s = slot; slot = (slot + 1) & 31;
            if (lane == 0) red[s][wave] = v;
            __syncthreads();
            v = red[s][0] + red[s][1] + red[s][2] + red[s][3];
        }
        return v;
    };
    // ---- the tile as it is coded now
    ks265_cu8 c;
    c.mvx = c.mvy = c.mv1x = c.mv1y = 0; c.log2_cu = 0; c.cbf = 0; c.pred_mode = 1; c.inter_dir = 0;
    if (inside) c = snap[(long)(y0 >> 3) * g.w8 + (x0 >> 3)];
    const int log2c = c.log2_cu & 15;
    const bool inter = inside && c.pred_mode == 0 && log2c >= 3;
    const int c8 = inter ? 1 << (log2c - 3) : 1;
    const bool first = inter && !(tx & (c8 - 1)) && !(ty & (c8 - 1)) && sub == 0;           // the lane that counts the tile's CU
    // does the tile's CU carry residual: the OR of cbf over the CU's tiles = over a node of the CU's own level
    unsigned cbf_any;
    {
        const unsigned own = (inter && sub == 0 && c.cbf) ? 1u : 0u;
        const unsigned o3 = nsum(own, 3), o2 = nsum(own, 2), o1 = nsum(own, 1), o0 = nsum(own, 0);
        cbf_any = log2c >= 6 ? o0 : log2c == 5 ? o1 : log2c == 4 ? o2 : o3;
    }
    const unsigned syn_own = first ? (cbf_any ? SP_SYN_CODED : SP_SYN_SKIP) : 0u;
    // the lane's samples: luma rows 2 sub, 2 sub + 1 of the tile (two quads each), row sub of the 4x4 chroma blocks
    const uint8_t *Sy = ks_org_y(g, src_y), *Su = ks_org_c(g, src_u), *Sv = ks_org_c(g, src_v);
    uint8_t *Ry = ks_org_y(g, rec_y), *Ru = ks_org_c(g, rec_u), *Rv = ks_org_c(g, rec_v);
    const int ly = y0 + 2 * sub, cxx = x0 >> 1, cyy = (y0 >> 1) + sub;
    unsigned sY[4] = {0, 0, 0, 0}, sC[2] = {0, 0};
    unsigned dY = 0, dC = 0, bits = 0;
    if (inter) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint2 s = *(const uint2 *)(Sy + (long)(ly + r) * g.sy + x0), q = *(const uint2 *)(Ry + (long)(ly + r) * g.sy + x0);
            sY[2 * r] = s.x; sY[2 * r + 1] = s.y;
            dY += sp_sse4(s.x, q.x) + sp_sse4(s.y, q.y);
        }
        sC[0] = *(const unsigned *)(Su + (long)cyy * g.sc + cxx); sC[1] = *(const unsigned *)(Sv + (long)cyy * g.sc + cxx);
        dC = sp_sse4(sC[0], *(const unsigned *)(Ru + (long)cyy * g.sc + cxx)) + sp_sse4(sC[1], *(const unsigned *)(Rv + (long)cyy * g.sc + cxx));
    }
    {
        // bits of the levels: 4x4 groups - a luma group is the left or right half of the rows of a lane pair, a chroma group the four rows of the tile's lanes
        int bl = 0, cl = 0, br = 0, cr = 0, bu = 0, cu = 0, bv = 0, cv = 0;
        if (inter) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4 w = *(const uint4 *)(lvl_y + (long)(ly + r) * g.W + x0);
                sp_lvl4(make_uint2(w.x, w.y), bl, cl); sp_lvl4(make_uint2(w.z, w.w), br, cr);
            }
            sp_lvl4(*(const uint2 *)(lvl_u + (long)cyy * (g.W / 2) + cxx), bu, cu);
            sp_lvl4(*(const uint2 *)(lvl_v + (long)cyy * (g.W / 2) + cxx), bv, cv);
        }
        bl += __shfl_xor(bl, 1, 64); cl += __shfl_xor(cl, 1, 64); br += __shfl_xor(br, 1, 64); cr += __shfl_xor(cr, 1, 64);
        bu += __shfl_xor(bu, 1, 64); cu += __shfl_xor(cu, 1, 64); bv += __shfl_xor(bv, 1, 64); cv += __shfl_xor(cv, 1, 64);
        bu += __shfl_xor(bu, 2, 64); cu += __shfl_xor(cu, 2, 64); bv += __shfl_xor(bv, 2, 64); cv += __shfl_xor(cv, 2, 64);
        const int gb = (sub & 1) ? br : bl, gc = (sub & 1) ? cr : cl;
        if (gc) bits += (unsigned)(gb + 10 + 16 - gc);
        if (sub == 0 && cu) bits += (unsigned)(bu + 10 + 16 - cu);
        if (sub == 1 && cv) bits += (unsigned)(bv + 10 + 16 - cv);
    }
    bool done = false;
#pragma unroll 1
    for (int l = 0; l <= deepest; ++l) {
        const int s = 64 >> l, n8 = s >> 3;
        const int nx0 = cx * 64 + ((tx & ~(n8 - 1)) << 3), ny0 = cy * 64 + ((ty & ~(n8 - 1)) << 3);      // the lane's node
        const bool node_in = nx0 + s <= g.W && ny0 + s <= g.H;
        const unsigned bad = nsum((!inter || log2c > 6 - l) && sub == 0 ? 1u : 0u, l);
        const unsigned ncu = nsum(first ? 1u : 0u, l);
        const unsigned syn = nsum(syn_own + ((first && log2c < 6 - l) ? SP_SYN_BELOW : 0u), l);
        const unsigned any_cbf = nsum((inter && sub == 0 && c.cbf) ? 1u : 0u, l);
        const unsigned ndY = nsum(dY, l), ndC = nsum(dC, l), nbits = nsum(bits, l);
        // (the first tile of the node tells whether the node is one CU in two partitions)
        const int part0 = __shfl((int)(c.log2_cu >> 4), (lane & ~((4 << (2 * (3 - l))) - 1)) & 63, 64);
        unsigned part_first = (unsigned)part0;
        if (l == 0) { const int sl = slot; slot = (slot + 1) & 31; if (tid == 0) red[sl][0] = (unsigned)(c.log2_cu >> 4); __syncthreads(); part_first = red[sl][0]; }
        bool q = node_in && !done && bad == 0;
        if (q && ncu == 1 && !part_first && !any_cbf) q = false;                 // one CU without residual already
        const unsigned long long jcur = ((unsigned long long)(ndY + ((unsigned)(((unsigned long long)ndC * SP_CHROMA_W) >> 2))) << 10) + (unsigned long long)(lam2 * (long long)(nbits + syn));
        // ---- the candidates: which exist, their positions, which repeat an earlier one
        unsigned valid = 0, distinct = 0;
        {
            SpMotion mm[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                mm[k] = sp_cand<MR>(g, snap, nx0, ny0, s, k, bi_zero);
                const bool ok = q && mm[k].ok;
                bool rep = false;
#pragma unroll
                for (int j = 0; j < k; ++j) rep |= ((valid >> j) & 1u) && sp_same(mm[j], mm[k]);
                valid |= (ok ? 1u : 0u) << k;
                distinct |= (ok && !rep ? 1u : 0u) << k;
            }
        }
        unsigned long long best = jcur;
        int bestk = -1;
        const int nd_max = l == 0 ? __popc(distinct) : 6;             // level 0: one node, the loop is uniform (it holds barriers)
#pragma unroll 1
        for (int it = 0; it < nd_max; ++it) {
            if (l != 0 && !__any((int)__popc(distinct) > it)) break;
            const bool on = (int)__popc(distinct) > it;
            int k = 0;
            { unsigned d = distinct; for (int i = 0; i < it; ++i) d &= d - 1u; k = d ? __ffs((int)d) - 1 : 0; }
            const SpMotion m = sp_cand<MR>(g, snap, nx0, ny0, s, k, bi_zero);
            unsigned pY = 0, pC = 0;
            if (on) {
                const int dir = m.dir8 & 3, i0 = MR ? (m.dir8 >> 4) & 3 : 0, i1 = MR ? (m.dir8 >> 6) & 3 : 0;
                const uint8_t *ry0 = ks_org_y(g, sp_pick(R.y0, i0)), *ry1 = ks_org_y(g, sp_pick(R.y1, i1));
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int h = 0; h < 2; ++h) pY += sp_sse4(sY[2 * r + h], sp_pred4<true>(ry0, ry1, g.sy, x0 + 4 * h, ly + r, dir, m.mvx, m.mvy, m.mv1x, m.mv1y));
                pC = sp_sse4(sC[0], sp_pred4<false>(ks_org_c(g, sp_pick(R.u0, i0)), ks_org_c(g, sp_pick(R.u1, i1)), g.sc, cxx, cyy, dir, m.mvx, m.mvy, m.mv1x, m.mv1y))
                   + sp_sse4(sC[1], sp_pred4<false>(ks_org_c(g, sp_pick(R.v0, i0)), ks_org_c(g, sp_pick(R.v1, i1)), g.sc, cxx, cyy, dir, m.mvx, m.mvy, m.mv1x, m.mv1y));
            }
            const unsigned nY = nsum(pY, l), nC = nsum(pC, l);
            if (on) {
                const int pos = __popc(valid & ((1u << k) - 1u));
                const unsigned long long j = ((unsigned long long)(nY + ((unsigned)(((unsigned long long)nC * SP_CHROMA_W) >> 2))) << 10) + (unsigned long long)(lam2 * (long long)(SP_SKIP_BASE + SP_SKIP_POS * pos));
                if (j < best) { best = j; bestk = k; }
            }
        }
        // ---- an accepted node: one CU without residual; its lanes write their samples' prediction, clear their levels, the tile's first lane the record
        if (bestk >= 0) {
            const SpMotion m = sp_cand<MR>(g, snap, nx0, ny0, s, bestk, bi_zero);
            const int dir = m.dir8 & 3, i0 = MR ? (m.dir8 >> 4) & 3 : 0, i1 = MR ? (m.dir8 >> 6) & 3 : 0;
            const uint8_t *ry0 = ks_org_y(g, sp_pick(R.y0, i0)), *ry1 = ks_org_y(g, sp_pick(R.y1, i1));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                uint2 o;
                o.x = sp_pred4<true>(ry0, ry1, g.sy, x0, ly + r, dir, m.mvx, m.mvy, m.mv1x, m.mv1y);
                o.y = sp_pred4<true>(ry0, ry1, g.sy, x0 + 4, ly + r, dir, m.mvx, m.mvy, m.mv1x, m.mv1y);
                *(uint2 *)(Ry + (long)(ly + r) * g.sy + x0) = o;
                *(uint4 *)(lvl_y + (long)(ly + r) * g.W + x0) = make_uint4(0u, 0u, 0u, 0u);
            }
            *(unsigned *)(Ru + (long)cyy * g.sc + cxx) = sp_pred4<false>(ks_org_c(g, sp_pick(R.u0, i0)), ks_org_c(g, sp_pick(R.u1, i1)), g.sc, cxx, cyy, dir, m.mvx, m.mvy, m.mv1x, m.mv1y);
            *(unsigned *)(Rv + (long)cyy * g.sc + cxx) = sp_pred4<false>(ks_org_c(g, sp_pick(R.v0, i0)), ks_org_c(g, sp_pick(R.v1, i1)), g.sc, cxx, cyy, dir, m.mvx, m.mvy, m.mv1x, m.mv1y);
            *(uint2 *)(lvl_u + (long)cyy * (g.W / 2) + cxx) = make_uint2(0u, 0u);
            *(uint2 *)(lvl_v + (long)cyy * (g.W / 2) + cxx) = make_uint2(0u, 0u);
            if (sub == 0) {
                ks265_cu8 o;
                o.mvx = (int16_t)m.mvx; o.mvy = (int16_t)m.mvy; o.mv1x = (int16_t)m.mv1x; o.mv1y = (int16_t)m.mv1y;
                o.log2_cu = (uint8_t)(6 - l); o.cbf = 0; o.pred_mode = 0; o.inter_dir = (uint8_t)m.dir8;
                cu8[(long)(y0 >> 3) * g.w8 + (x0 >> 3)] = o;
            }
            done = true;
        }
        if (l == 0 && __syncthreads_or(done ? 1 : 0)) break;              // the whole CTU became one CU (uniform: level 0 has one node)
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
    const long long lam2 = (long long)f->cfg.lambda_q4 * f->cfg.lambda_q4;
    const int nctu = f->g.ctu_cols * f->g.ctu_rows, deepest = 3;
    if (f->mrefb) hipLaunchKernelGGL(skip_pass_kernel<true>, dim3(nctu), dim3(256), 0, f->ctx->stream, f->g, lam2, deepest, is_b ? 1 : 0, src.y, src.u, src.v, R, f->cu8_tmp, dev_cu8, dev_lvl_y, dev_lvl_u, dev_lvl_v, recon.y, recon.u, recon.v);
    else hipLaunchKernelGGL(skip_pass_kernel<false>, dim3(nctu), dim3(256), 0, f->ctx->stream, f->g, lam2, deepest, is_b ? 1 : 0, src.y, src.u, src.v, R, f->cu8_tmp, dev_cu8, dev_lvl_y, dev_lvl_u, dev_lvl_v, recon.y, recon.u, recon.v);
    return ks265_check_launch(f->ctx);
}
