// frame_me.hip — motion estimation stages.
//   Stage A  ks265_me_integer : one workgroup per CTU; the 64x64 source block and the +-64 reference window are staged
//            in LDS once (every window byte leaves HBM once per CTU), then all 85 PUs (64x64 .. 8x8) run the reference's
//            diamond loop (interMeDia enc@0x48fbe0) coarse to fine.  One lane = one row segment of one PU;
//            v_sad_u8 on packed dwords, v_alignbyte for the unaligned window reads, DPP/ds_swizzle group reductions.
//   Stage B  ks265_me_subpel  : 8 half-pel + 8 quarter-pel SATD candidates per PU on the precomputed fractional planes
//            (subMeSquare enc@0x4b5660; had_c enc@0x47b680).  Distinct (tile, centre) items per CTU group, their 8x8
//            Hadamard transforms as i8 MFMA GEMMs (H8 (x) H8), PU sums by DPP.
//   Stage C  ks265_cu_decide  : bottom-up quadtree compare.
#include "frame_common.h"
#include "interp_dev.h"
#include <type_traits>

using namespace ks265;

#define KS_COST_INF_ME 0x07FFFFFFu
// ------------------------------------------------------------------ Stage B: sub-pel SATD refinement
typedef int ks_v4i __attribute__((ext_vector_type(4)));
typedef short ks_s16x2 __attribute__((ext_vector_type(2)));

// sum over the aligned group of 1 / 4 / 16 / 64 lanes that forms one PU at `level` (wave-uniform)
__device__ __forceinline__ unsigned pu_group_sum(unsigned v, int level)
{
    if (level <= 2) {
        v += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)v);
        v += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)v);
    }
    if (level <= 1) {
        v += (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)v);
        v += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)v);
    }
    if (level == 0) {
        v += (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (16 << 10));
        v += (unsigned)__shfl_xor((int)v, 32, 64);
    }
    return v;
}

struct KsGroupUniform { int level; __device__ __forceinline__ unsigned operator()(unsigned v) const { return pu_group_sum(v, level); } };
// the same with a level per lane (lanes of one PU agree): all three sums, each lane takes its own
struct KsGroupPerLane {
    int level;
    __device__ __forceinline__ unsigned operator()(unsigned v) const
    {
        unsigned s2 = v + (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)v);
        s2 += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)s2);
        unsigned s1 = s2 + (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)s2);
        s1 += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)s1);
        unsigned s0 = s1 + (unsigned)__builtin_amdgcn_ds_swizzle((int)s1, 0x1F | (16 << 10));
        s0 += (unsigned)__shfl_xor((int)s0, 32, 64);
        return level == 3 ? v : level == 2 ? s2 : level == 1 ? s1 : s0;
    }
};

// Stage B: sub-pel refinement of all 85 PUs of a CTU = the reference's, per PU (round 4; restated in oracle/ks265_subme_ref.c and pinned on recorded calls of the reference, tests/test_subme.py):
//   phase R  getMvResolution enc@0x483ca0: the four integer neighbours' SADs of the integer winner decide whether the PU is refined at all (cfg.sub_thr / sub_cap);
//   phase H  subMeHpel_RealInterp enc@0x4b4e90: eight half-sample candidates, evaluation order 3 4 1 6 0 2 5 7 (raster indices), cost = SAD (or Hadamard, cfg.sub_satd)
//            of the normative samples + rate, strict '<' against the integer cost; its "flat cost surface" verdict skips the quarter step (cfg.sub_thr != 0);
//   phase Q  subMeQpel_8Sad_v{0,2}h{0,2}_RealInterp enc@0x4b2bc0-0x4b43a0: eight quarter-sample candidates around the half-step winner, order 1 6 3 0 5 4 2 7;
//            cfg.subme 1 ("fast") keeps to +-2 quarter samples around the integer position and tries a diagonal only next to the running winner;
//   phase F  (this pipeline's own) Hadamard of the chosen prediction + rate becomes the record's cost - the CU tree, merge pass and bi decision compare SATD.
//
// The four quadtree levels of a CTU look at the same 64 8x8 tiles, and where a PU and its ancestors search around the same
// centre (most of them: only ~30 % of the 256 (level, tile) pairs of a CTU are distinct on the bench clip) the tile distortions of
// the whole ring are identical.  Each phase therefore (1) builds the list of DISTINCT (tile, centre) items of the PUs that take part in it,
// (2) evaluates every candidate of every item once, items spread over the threads, tile distortions into LDS, and (3) lets each
// (level, tile) pair pick its item's values, sum them over the PU and walk the candidates in the reference's order.  Pure memoisation: every value used is
// the one the pair would have computed itself, so the result is that of the straightforward evaluation, bit for bit.
#ifndef KS_SUBPEL_NC
#define KS_SUBPEL_NC 8                                             // CTUs pooled per work-group (one wave each)
#endif
#ifndef KS_SUBPEL_OCC
#define KS_SUBPEL_OCC 2
#endif
struct KsSubme { int subme, satd, thr, flat, cap, cap_step, diag_fast; };
template <int NC, bool SATD>
__global__ __launch_bounds__(NC * 64, KS_SUBPEL_OCC) void me_subpel_kernel(KsGeom g, int lam, KsSubme K, const uint8_t *src, const uint8_t *ref, ks265_pu *pus)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nctu = g.ctu_cols * g.ctu_rows;
    const int grp = ks_xcd_swizzle(blockIdx.x, (nctu + NC - 1) / NC);
    // this wave keeps the books of one CTU, all four levels.  The group's CTUs are SPREAD over the picture (wave w takes CTU w x groups + g), not neighbours: at 2160p the kernel is
    // one round of 255 work-groups on 256 compute units, so it lasts as long as its busiest work-group, and a work-group's work is the number of distinct (tile, centre) items of
    // its eight CTUs - busy CTUs are neighbours (a moving edge), spread they even out: 0.182 -> 0.157 ms (round 5; KS_SUBPEL_ADJACENT restores the old grouping)
#ifdef KS_SUBPEL_ADJACENT
    const int ctu = grp * NC + wave;
#else
    const int ctu = wave * ((nctu + NC - 1) / NC) + grp;
#endif
    const bool have = ctu < nctu;
    ks265_pu *cp = pus + (long)(have ? ctu : 0) * 85;
    const uint8_t *Sp = ks_org_y(g, src);
    __shared__ int s_org[NC];                                      // pixel origin of each wave's CTU: x | y << 16
    __shared__ int s_key[NC][4][64];
    __shared__ unsigned short s_idx[NC][4][64];
    __shared__ unsigned short s_item[NC * 256];
    __shared__ int s_cnt[NC * 4];
    __shared__ unsigned short s_sat[9][NC * 256];                 // slot = gy * 3 + gx of the ring (4 = centre); an 8x8 SATD is at most 8 * 8 * 8 * 255 / 4 = 32640, a SAD 16320
    __shared__ __attribute__((aligned(16))) unsigned short s_hx[NC][16][200];   // per wave and item column: horizontally filtered rows, [pixel][row 0..15]; + 4 dwords: the 16 columns of a
                                                                                      // block start in different banks (100 dwords = 4 mod 32; unpadded, all of them hit the same); nine pixel columns in phase H's shared form
    if (lane == 0) s_org[wave] = have ? ((ctu % g.ctu_cols) * 64) | (((ctu / g.ctu_cols) * 64) << 16) : 0;
    // Z-order: lane bits (y2 x2 y1 x1 y0 x0)
    const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    bool valid[4], dosub[4], qrun[4];
    int pidx[4], bx[4], by[4], mvpx[4], mvpy[4], hmx[4], hmy[4];
    unsigned bc[4], bd[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int px = tx >> (3 - l), py = ty >> (3 - l);
        pidx[l] = ks_level_base(l) + py * (1 << l) + px;
        const ks265_pu p = cp[pidx[l]];
        valid[l] = have && p.cost != KS_COST_INVALID;              // the whole PU lies inside the picture
        bx[l] = p.mvx; by[l] = p.mvy; mvpx[l] = p.mvpx; mvpy[l] = p.mvpy; bc[l] = p.cost; bd[l] = 0;
        dosub[l] = true; qrun[l] = true; hmx[l] = 0; hmy[l] = 0;
    }
    // MFMA operands of the Hadamard SATD (see (2) below).  A operand = rows mb*16 + n16 of H = H8 (x) H8 in natural
    // (Sylvester) order: H[m][k] = (-1)^popcount(m & k); this lane's 16 K slots are k = gk*16 .. gk*16 + 15.
    const int n16 = lane & 15, gk = lane >> 4;
    ks_v4i Hm[4];
    {
        const unsigned pat = (n16 & 2) ? ((n16 & 1) ? 0x01FFFF01u : 0xFFFF0101u) : ((n16 & 1) ? 0xFF01FF01u : 0x01010101u);   // signs over k & 3
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                Hm[mb][w] = (int)((__popc((mb * 16 + n16) & (gk * 16 + w * 4)) & 1) ? pat ^ 0xFEFEFEFEu : pat);
    }
    const ks_v4i CinN = {0x8000, 0x8000, 0x8000, 0x8000};
    const ks_v4i Cin0 = {gk == 0 ? 0x8000 + 64 : 0x8000, 0x8000, 0x8000, 0x8000};   // output row = 4 * gk + register
    // `before` = the (Z-ordered) lanes whose tile precedes this lane's tile in RASTER order.  Items are listed level by level
    // in raster order, so that the 16 columns of an MFMA operand block are mostly x-adjacent tiles: with equal centres their
    // rows are adjacent 8-byte pieces of the same cache lines (the L1 sees a few lines per load instead of 64).
    unsigned long long before = 0;
    {
        const unsigned long long row0 = 0x0000000000330033ull, col0 = 0x0000050500000505ull;   // lanes with ty == 0 / tx == 0
        unsigned long long cols = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int zy = ((t & 1) << 1) | ((t & 2) << 2) | ((t & 4) << 3), zx = (t & 1) | ((t & 2) << 1) | ((t & 4) << 2);
            if (t < ty) before |= row0 << zy;
            if (t < tx) cols |= col0 << zx;
        }
        const int zyme = ((ty & 1) << 1) | ((ty & 2) << 2) | ((ty & 4) << 3);
        before |= (row0 << zyme) & cols;
    }
    // phases: -1 = R (only when the configuration can say no), 0 = H, 1 = Q, 2 = F (not with sub_satd: the winner's Hadamard cost is the running best)
#pragma unroll 1
    for (int phase = (K.thr | K.cap) ? -1 : 0; phase < (SATD ? 2 : 3); ++phase) {
        const int step = phase == 0 ? 2 : phase == 1 ? 1 : 0;
        const bool single = phase == 2;                            // F: one candidate, the centre itself
        const bool had = SATD || single;                           // the measure of this phase's candidates
        const bool skip_centre = phase == 1 || (phase == 0 && !SATD);   // H by SAD starts from the integer search's cost; Q from H's best
        int owner[4];
        bool act[4];
        __syncthreads();                                           // the previous phase's readers are done (and s_org is written)
        // (1) the distinct (tile, centre) items of the group
        int key[4];
        unsigned long long bal[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            act[l] = valid[l] && (phase < 0 || single || (dosub[l] && (phase == 0 || qrun[l])));
            key[l] = act[l] ? ((bx[l] & 0xFFFF) | (by[l] << 16)) : (int)(0x80000000u | (unsigned)l);
            s_key[wave][l][lane] = key[l];
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            owner[l] = l;
#pragma unroll
            for (int m = 3; m >= 0; --m) if (key[m] == key[l]) owner[l] = m;   // the coarsest level with this centre
            bal[l] = __ballot(act[l] && owner[l] == l);
            if (lane == 0) s_cnt[wave * 4 + l] = __popcll(bal[l]);
        }
        __syncthreads();
        int nitems = 0;
        {
            int off = 0;
#pragma unroll
            for (int q = 0; q < NC * 4; ++q) { const int c = s_cnt[q]; off += q < wave * 4 ? c : 0; nitems += c; }
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (act[l] && owner[l] == l) {
                    const int idx = off + __popcll(bal[l] & before);   // raster order within the level
                    s_idx[wave][l][lane] = (unsigned short)idx;
                    s_item[idx] = (unsigned short)(lane | (l << 6) | (wave << 8));
                }
                off += __popcll(bal[l]);
            }
        }
        __syncthreads();
        // (2) every candidate of every distinct item, once.  A wave takes 16 items at a time (lane = (item column n16 = lane & 15, K group gk = lane >> 4): the lane's two rows
        // 2 gk, 2 gk + 1 of the item's 8x8 tile).
        // Hadamard (sub_satd, and phase F): the 8x8 transforms run on the matrix cores: with x the 64 pixels of a tile, the 2-D Hadamard transform is the 64x64 +-1 matrix
        // H = H8 (x) H8 applied to x, i.e. an i8 GEMM  C[64 coefficients][16 tiles] = H . X  with exact i32 accumulation (v_mfma_i32_16x16x64_i8).
        // Two MFMAs chained on one accumulator give the transform of the
        // DIFFERENCE directly: C = bias + H.(ref - 128) + H.(~(src - 128)), and ~(s - 128) = -(s - 128) - 1 whose "-1" lands
        // on coefficient 0 only (+64 folded into that accumulator's start value).  bias = 2^15 keeps every coefficient
        // positive in 16 bits, so |c| + accumulate is one v_sad_u16 against the bias.  sum |c| does not depend on the order
        // of the Hadamard rows, nor on how the K slots of the A and B operands are numbered (both operands use the same
        // numbering), so the sum equals had_c's (enc@0x47b680) butterfly network bit for bit.
        // SAD (the reference's measure up to -preset slower): four v_sad_u8 per candidate and lane.
        // The candidates' samples are interpolated from the reference picture (no fractional planes): ONE instruction stream for every fraction - horizontal taps of the
        // lane's own fx (the integer position is the tap set {0 0 0 64 0 0 0 0}) into 16-bit intermediates, vertical taps of its own fy, (sum + 2048) >> 12 - which
        // equals the one-dimensional filters' (sum + 32) >> 6 and the plain sample exactly (a factor 64 moves through the shift), so lanes whose centres have
        // different fractions do not diverge.  The three x positions of a ring are filtered ONCE per item: the item's four lanes (K groups) filter four rows each of
        // its 16-row neighbourhood, exchange them through LDS (row pairs packed for v_dot2_i32_i16; a wave's own LDS traffic completes in order, no barrier), and
        // every lane then reads the ten rows its two output rows of the three y positions need.
        // Items are dealt to the waves in blocks of 16, block b to wave b mod NC: the pooled items of the group's CTUs spread evenly over the waves.
        // The loads are software-pipelined: the 20 dwords a lane filters for x position gx + 1 (or for the first x position of the wave's NEXT block) are requested
        // before the lane turns to the three y positions of gx - at two waves per SIMD nothing else would cover an L2 round trip per x position.
        struct Item { int ii, cx, cy; unsigned ro; ks_v4i S; };
        auto setup = [&](int blk, Item &t) {
            t.ii = blk * 16 + n16;
            const int it = s_item[t.ii < nitems ? t.ii : blk * 16];    // the last block is padded with copies of its first item
            const int il = it & 63, iw = it >> 8, ikey = s_key[iw][(it >> 6) & 3][il], org = s_org[iw];
            const int itx = (il & 1) | ((il >> 1) & 2) | ((il >> 2) & 4), ity = ((il >> 1) & 1) | ((il >> 2) & 2) | ((il >> 3) & 4);
            t.ro = (unsigned)(((org >> 16) + ity * 8 + 2 * gk) * g.sy + (org & 0xFFFF) + itx * 8);   // this lane's two rows of the tile
            t.cx = (int)(short)(ikey & 0xFFFF); t.cy = ikey >> 16;
            const uint2 a0 = *(const uint2 *)(Sp + t.ro), a1 = *(const uint2 *)(Sp + t.ro + (unsigned)g.sy);
            t.S = ks_v4i{(int)(a0.x ^ 0x7F7F7F7Fu), (int)(a0.y ^ 0x7F7F7F7Fu), (int)(a1.x ^ 0x7F7F7F7Fu), (int)(a1.y ^ 0x7F7F7F7Fu)};
        };
        // phase H with every centre on an integer position (always, after the integer search and the propagation): the shared form below
        bool lane_int = true;
#pragma unroll
        for (int l = 0; l < 4; ++l) lane_int = lane_int && (!act[l] || ((bx[l] | by[l]) & 3) == 0);
        const bool h_shared = phase == 0 && __syncthreads_and(lane_int ? 1 : 0) != 0;
        if (phase < 0) {
            // R: the SADs of the four integer neighbours (above, below, left, right: sad4_c enc@0x47ae90's order) of the integer winner - plain rows, no filter
#pragma unroll 1
            for (int blk = wave; blk * 16 < nitems; blk += NC) {
                Item t;
                setup(blk, t);
                const unsigned s0 = (unsigned)t.S[0] ^ 0x7F7F7F7Fu, s1 = (unsigned)t.S[1] ^ 0x7F7F7F7Fu, s2 = (unsigned)t.S[2] ^ 0x7F7F7F7Fu, s3 = (unsigned)t.S[3] ^ 0x7F7F7F7Fu;
                const uint8_t *rp = ref + (unsigned)((int)t.ro + (int)g.org_y + (t.cy >> 2) * g.sy + (t.cx >> 2));
                uint2 rm, r0, r1, r2, l0, l1, q0, q1;
                __builtin_memcpy(&rm, rp - g.sy, 8); __builtin_memcpy(&r0, rp, 8); __builtin_memcpy(&r1, rp + g.sy, 8); __builtin_memcpy(&r2, rp + 2 * g.sy, 8);
                __builtin_memcpy(&l0, rp - 1, 8); __builtin_memcpy(&l1, rp + g.sy - 1, 8); __builtin_memcpy(&q0, rp + 1, 8); __builtin_memcpy(&q1, rp + g.sy + 1, 8);
                unsigned a[4];
                a[0] = __builtin_amdgcn_sad_u8(s0, rm.x, __builtin_amdgcn_sad_u8(s1, rm.y, __builtin_amdgcn_sad_u8(s2, r0.x, __builtin_amdgcn_sad_u8(s3, r0.y, 0u))));
                a[1] = __builtin_amdgcn_sad_u8(s0, r1.x, __builtin_amdgcn_sad_u8(s1, r1.y, __builtin_amdgcn_sad_u8(s2, r2.x, __builtin_amdgcn_sad_u8(s3, r2.y, 0u))));
                a[2] = __builtin_amdgcn_sad_u8(s0, l0.x, __builtin_amdgcn_sad_u8(s1, l0.y, __builtin_amdgcn_sad_u8(s2, l1.x, __builtin_amdgcn_sad_u8(s3, l1.y, 0u))));
                a[3] = __builtin_amdgcn_sad_u8(s0, q0.x, __builtin_amdgcn_sad_u8(s1, q0.y, __builtin_amdgcn_sad_u8(s2, q1.x, __builtin_amdgcn_sad_u8(s3, q1.y, 0u))));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[k] += (unsigned)__builtin_amdgcn_ds_swizzle((int)a[k], 0x1F | (16 << 10));
                    a[k] += (unsigned)__shfl_xor((int)a[k], 32, 64);
                    if (gk == 0 && t.ii < nitems) s_sat[k][t.ii] = (unsigned short)a[k];
                }
            }
        } else {
        unsigned raw[4][5], rsh = 0;
        // rows 4 gk .. 4 gk + 3 of the item's 16-row neighbourhood at x position gx: picture row = tile row (2 gk) + rbase + 2 gk + j
        auto request = [&](const Item &t, int gx) {
            const int ax = t.cx + (gx - 1) * step, rbase = ((t.cy - step) >> 2) - 3;
            const uint8_t *hp = ref + (unsigned)((int)t.ro + (int)g.org_y + (rbase + 2 * gk) * g.sy + (ax >> 2));
            rsh = luma_hrow8_shift(hp);                              // (the row pitch is a multiple of 4: one shift for the four rows)
#pragma unroll
            for (int j = 0; j < 4; ++j) luma_hrow8_load(hp + j * g.sy, raw[j]);
        };
        const int gx_first = single ? 1 : 0;
        Item cur;
        int blk = wave;
        bool more = blk * 16 < nitems;
        // measure of one candidate (this lane's two rows of it), summed over the item's four lanes, into the item's slot
        auto finish = [&](const unsigned (&rw)[4], const ks_v4i &S, int slot, int ii) {
            unsigned a = 0;
            if (had) {
                const ks_v4i B = {(int)(rw[0] ^ 0x80808080u), (int)(rw[1] ^ 0x80808080u), (int)(rw[2] ^ 0x80808080u), (int)(rw[3] ^ 0x80808080u)};
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    ks_v4i C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Hm[mb], B, mb == 0 ? Cin0 : CinN, 0, 0, 0);
                    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Hm[mb], S, C, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_sad_u16((unsigned)C[r], 0x8000u, a);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_sad_u8(rw[r], (unsigned)S[r] ^ 0x7F7F7F7Fu, a);
            }
            a += (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x1F | (16 << 10));
            a += (unsigned)__shfl_xor((int)a, 32, 64);
            if (gk == 0 && ii < nitems) s_sat[slot][ii] = (unsigned short)(had ? (a + 2) >> 2 : a);
        };
        if (h_shared) {
            // ---- phase H around integer centres (round 5): the candidates at x - 1/2 and x + 1/2 are the SAME nine half-sample columns read one apart, those at y - 1/2 and
            //      y + 1/2 the same rows read one apart, the integer column needs no horizontal filter and the integer row no vertical one.  Per item: 9 instead of 24 filtered
            //      samples per input row, and per lane and column three vertically filtered rows (taps starting at its rows 2 gk, 2 gk + 1, 2 gk + 2) serve both y positions of both x
            //      positions.  Every value is the one the general form computes (same taps, same rounding): the candidates' costs are unchanged.
            auto request9 = [&](const Item &t) {
                const uint8_t *hp = ref + (unsigned)((int)t.ro + (int)g.org_y + ((t.cy >> 2) - 4 + 2 * gk) * g.sy + (t.cx >> 2) - 1);
                rsh = luma_hrow8_shift(hp);
#pragma unroll
                for (int j = 0; j < 4; ++j) luma_hrow8_load(hp + j * g.sy, raw[j]);
            };
            // the half filter's taps as row-pair weights: an even start takes (c0 c1)(c2 c3)(c4 c5)(c6 c7), an odd one (0 c0)(c1 c2)(c3 c4)(c5 c6)(c7 0)
            const unsigned E0[4] = {0x0004FFFFu, 0x0028FFF5u, 0xFFF50028u, 0xFFFF0004u};
            const unsigned E1[5] = {0xFFFF0000u, 0xFFF50004u, 0x00280028u, 0x0004FFF5u, 0x0000FFFFu};
            // three vertically filtered samples of one pixel column (row pairs p[0..4] = this lane's ten rows): taps starting at row 0, 1, 2
            auto vert3 = [&](const unsigned (&p)[5], int (&o)[3]) {
                int v0 = 2048, v1 = 2048, v2 = 2048;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ks_s16x2, E0[k]), __builtin_bit_cast(ks_s16x2, p[k]), v0, false);
                    v2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ks_s16x2, E0[k]), __builtin_bit_cast(ks_s16x2, p[k + 1]), v2, false);
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) v1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ks_s16x2, E1[k]), __builtin_bit_cast(ks_s16x2, p[k]), v1, false);
                o[0] = clip8(ks_no_pk(v0 >> 12)); o[1] = clip8(ks_no_pk(v1 >> 12)); o[2] = clip8(ks_no_pk(v2 >> 12));
            };
            if (more) { setup(blk, cur); request9(cur); }
#pragma unroll 1
            while (more) {
                const int nblk = blk + NC;
                const bool nmore = nblk * 16 < nitems;
                Item nxt = cur;
                const int ii = cur.ii;
                const ks_v4i S = cur.S;
                unsigned short *hx = &s_hx[wave][n16][0];                    // [pixel 0..8][row 0..15] of this item
                unsigned Gp[2][8];                                            // the plain samples x 64 of this lane's four rows, row pairs (the integer column's "filtered" rows)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    int h0[9], h1[9], g0[8], g1[8];
                    luma_hrow9_half(raw[2 * jp], rsh, h0, g0);
                    luma_hrow9_half(raw[2 * jp + 1], rsh, h1, g1);
#pragma unroll
                    for (int i = 0; i < 9; ++i) *(unsigned *)&hx[i * 16 + 4 * gk + 2 * jp] = ((unsigned)h0[i] & 0xFFFFu) | ((unsigned)h1[i] << 16);
#pragma unroll
                    for (int i = 0; i < 8; ++i) Gp[jp][i] = (unsigned)g0[i] | ((unsigned)g1[i] << 16);
                }
                if (nmore) { setup(nblk, nxt); request9(nxt); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                {
                    // the half-sample columns: x - 1/2 = columns 0 .. 7, x + 1/2 = columns 1 .. 8; per column the three vertical samples and the two rows of the integer y position
                    unsigned P[9][5];
#pragma unroll
                    for (int i = 0; i < 9; ++i)
#pragma unroll
                        for (int k = 0; k < 5; ++k) P[i][k] = ((const unsigned *)hx)[i * 8 + gk + k];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the integer column overwrites the exchange area
                    int va[9], vb[9], vc[9], ia[9], ib[9];                    // taps from row 0 / 1 / 2; integer y: rows 4, 5 of the ten (the pair P[.][2])
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        int o[3];
                        vert3(P[i], o);
                        va[i] = o[0]; vb[i] = o[1]; vc[i] = o[2];
                        ia[i] = clip8(ks_no_pk(((int)(short)(P[i][2] & 0xFFFFu) + 32) >> 6)); ib[i] = clip8(ks_no_pk(((int)P[i][2] >> 16) + 32 >> 6));
                    }
#pragma unroll
                    for (int gx = 0; gx < 3; gx += 2) {
                        const int o = gx >> 1;                                 // column offset 0 (x - 1/2) or 1 (x + 1/2)
                        int r0[8], r1[8];
                        unsigned rw[4];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { r0[i] = va[i + o]; r1[i] = vb[i + o]; }
                        { const uint2 q0 = ks_pack_row8(r0), q1 = ks_pack_row8(r1); rw[0] = q0.x; rw[1] = q0.y; rw[2] = q1.x; rw[3] = q1.y; }
                        finish(rw, S, 0 * 3 + gx, ii);                         // y - 1/2
#pragma unroll
                        for (int i = 0; i < 8; ++i) { r0[i] = vb[i + o]; r1[i] = vc[i + o]; }
                        { const uint2 q0 = ks_pack_row8(r0), q1 = ks_pack_row8(r1); rw[0] = q0.x; rw[1] = q0.y; rw[2] = q1.x; rw[3] = q1.y; }
                        finish(rw, S, 2 * 3 + gx, ii);                         // y + 1/2
#pragma unroll
                        for (int i = 0; i < 8; ++i) { r0[i] = ia[i + o]; r1[i] = ib[i + o]; }
                        { const uint2 q0 = ks_pack_row8(r0), q1 = ks_pack_row8(r1); rw[0] = q0.x; rw[1] = q0.y; rw[2] = q1.x; rw[3] = q1.y; }
                        finish(rw, S, 1 * 3 + gx, ii);                         // y
                    }
                }
                {
                    // the integer column: its rows through the same exchange, vertical half filter only (the centre itself is not a candidate of this phase unless the
                    // Hadamard measure recomputes the start cost)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                        for (int i = 0; i < 8; ++i) *(unsigned *)&hx[i * 16 + 4 * gk + 2 * jp] = Gp[jp][i];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    unsigned P[8][5];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int k = 0; k < 5; ++k) P[i][k] = ((const unsigned *)hx)[i * 8 + gk + k];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next block overwrites the exchange area
                    int va[8], vb[8], vc[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { int o[3]; vert3(P[i], o); va[i] = o[0]; vb[i] = o[1]; vc[i] = o[2]; }
                    unsigned rw[4];
                    { const uint2 q0 = ks_pack_row8(va), q1 = ks_pack_row8(vb); rw[0] = q0.x; rw[1] = q0.y; rw[2] = q1.x; rw[3] = q1.y; }
                    finish(rw, S, 0 * 3 + 1, ii);
                    { const uint2 q0 = ks_pack_row8(vb), q1 = ks_pack_row8(vc); rw[0] = q0.x; rw[1] = q0.y; rw[2] = q1.x; rw[3] = q1.y; }
                    finish(rw, S, 2 * 3 + 1, ii);
                    if (!skip_centre) {
                        int c0[8], c1[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { c0[i] = (int)(P[i][2] & 0xFFFFu) >> 6; c1[i] = (int)(P[i][2] >> 16) >> 6; }
                        { const uint2 q0 = ks_pack_row8(c0), q1 = ks_pack_row8(c1); rw[0] = q0.x; rw[1] = q0.y; rw[2] = q1.x; rw[3] = q1.y; }
                        finish(rw, S, 4, ii);
                    }
                }
                cur = nxt; blk = nblk; more = nmore;
            }
        }
        if (more) { setup(blk, cur); request(cur, gx_first); }
#pragma unroll 1
        while (more) {
            const int nblk = blk + NC;
            const bool nmore = nblk * 16 < nitems;
            Item nxt = cur;
            {
                const int ii = cur.ii, cx = cur.cx, cy = cur.cy;
                const ks_v4i S = cur.S;
                const int aymin = cy - step;                                 // first input row of the neighbourhood: (aymin >> 2) - 3 relative to the tile row
                unsigned short *hx = &s_hx[wave][n16][0];                    // [pixel 0..7][row 0..15] of this item
                // vertical taps of the three y positions as row-pair weights: output row r starts at this lane's row s = roff + r (0, 1 or 2)
                unsigned W0[3][5], W1[3][5];
#pragma unroll
                for (int gy = 0; gy < 3; ++gy) {
                    const int ay = cy + (gy - 1) * step, roff = (ay >> 2) - (aymin >> 2);     // 0 or 1
                    int c[8];
                    luma_taps(ay & 3, c);
                    unsigned E0[6], E1[5];                             // E0[k + 1]: taps (2k, 2k + 1) = an even start; E1: an odd start
                    E0[0] = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) E0[k + 1] = ((unsigned)c[2 * k] & 0xFFFFu) | ((unsigned)c[2 * k + 1] << 16);
                    E0[5] = 0;
                    E1[0] = (unsigned)c[0] << 16;
#pragma unroll
                    for (int k = 1; k < 4; ++k) E1[k] = ((unsigned)c[2 * k - 1] & 0xFFFFu) | ((unsigned)c[2 * k] << 16);
                    E1[4] = (unsigned)c[7] & 0xFFFFu;
#pragma unroll
                    for (int k = 0; k < 5; ++k) { W0[gy][k] = roff ? E1[k] : E0[k + 1]; W1[gy][k] = roff ? E0[k] : E1[k]; }     // s = 0: E0[k + 1]; s = 1: E1[k]; s = 2: E0[k]
                }
#pragma unroll
                for (int gx = 0; gx < 3; ++gx) {
                    if (single && gx != 1) continue;
                    const int ax = cx + (gx - 1) * step;
                    int tl, th;
                    luma_taps_packed(ax & 3, tl, th);
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        int h0[8], h1[8];
                        luma_hrow8_calc(raw[2 * jp], rsh, tl, th, h0);
                        luma_hrow8_calc(raw[2 * jp + 1], rsh, tl, th, h1);
#pragma unroll
                        for (int i = 0; i < 8; ++i) *(unsigned *)&hx[i * 16 + 4 * gk + 2 * jp] = ((unsigned)h0[i] & 0xFFFFu) | ((unsigned)h1[i] << 16);
                    }
                    if (gx < 2 && !single) request(cur, gx + 1);
                    else if (nmore) { setup(nblk, nxt); request(nxt, gx_first); }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    unsigned P[8][5];                                    // per pixel: row pairs (0,1) (2,3) .. (8,9) of this lane's ten rows 2 gk .. 2 gk + 9
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int k = 0; k < 5; ++k) P[i][k] = ((const unsigned *)hx)[i * 8 + gk + k];      // dword reads (the start is only 4-byte aligned for odd K groups)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next x position overwrites the exchange area
#pragma unroll
                    for (int gy = 0; gy < 3; ++gy) {
                        if (single && gy != 1) continue;
                        if (skip_centre && gx == 1 && gy == 1) continue;
                        unsigned rw[4];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            int px[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                int v = 2048;
#pragma unroll
                                for (int k = 0; k < 5; ++k) v = __builtin_amdgcn_sdot2(__builtin_bit_cast(ks_s16x2, r ? W1[gy][k] : W0[gy][k]), __builtin_bit_cast(ks_s16x2, P[i][k]), v, false);
                                px[i] = clip8(ks_no_pk(v >> 12));
                            }
                            const uint2 pr = ks_pack_row8(px);
                            rw[2 * r] = pr.x; rw[2 * r + 1] = pr.y;
                        }
                        unsigned a = 0;
                        if (had) {
                            const ks_v4i B = {(int)(rw[0] ^ 0x80808080u), (int)(rw[1] ^ 0x80808080u), (int)(rw[2] ^ 0x80808080u), (int)(rw[3] ^ 0x80808080u)};
#pragma unroll
                            for (int mb = 0; mb < 4; ++mb) {
                                ks_v4i C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Hm[mb], B, mb == 0 ? Cin0 : CinN, 0, 0, 0);
                                C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Hm[mb], S, C, 0, 0, 0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_sad_u16((unsigned)C[r], 0x8000u, a);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_sad_u8(rw[r], (unsigned)S[r] ^ 0x7F7F7F7Fu, a);
                        }
                        // a = this lane's share of item n16 of the block: sum over the four K groups
                        a += (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x1F | (16 << 10));
                        a += (unsigned)__shfl_xor((int)a, 32, 64);
                        if (gk == 0 && ii < nitems) s_sat[gy * 3 + gx][ii] = (unsigned short)(had ? (a + 2) >> 2 : a);
                    }
                }
            }
            cur = nxt; blk = nblk; more = nmore;
        }
        }
        __syncthreads();
        // (3) per (level, tile): PU sums, then the reference's walk over the candidates.  The mv rate is separable: lambda * (bits(x) + bits(y)) >> 4 with three
        // possible x and three possible y per PU.  The group sums are cross-lane operations: every lane computes all of them, the walk itself is per lane (all lanes
        // of a PU hold the same values and take the same steps).
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int myitem = act[l] ? s_idx[wave][owner[l]][lane] : 0;
            const int cx0 = bx[l], cy0 = by[l], W2 = (64 >> l) * (64 >> l), l2 = 6 - l;
            if (phase < 0) {
                unsigned c4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c4[k] = pu_group_sum(act[l] ? s_sat[k][myitem] : 0, l);
                const unsigned rate0 = (unsigned)((lam * (se_bits(cx0 - mvpx[l]) + se_bits(cy0 - mvpy[l]))) >> 4);
                bool ds = true;
                if (K.cap && (unsigned)(((6 - l2) * K.cap_step + K.cap) << (2 * l2)) < bc[l]) ds = false;
                else if (K.thr) {
                    const int thr = (int)((W2 * K.thr) >> 3);
                    const unsigned m = max(max(c4[0], c4[1]), max(c4[2], c4[3])) << 2;      // the reference keeps them << 4 and compares >> 2
                    ds = (int)(m - ((bc[l] - rate0) << 2)) >= thr;
                }
                dosub[l] = ds;
                continue;
            }
            if (single) {
                const unsigned d = pu_group_sum(act[l] ? s_sat[4][myitem] : 0, l);
                bd[l] = d; bc[l] = d + (unsigned)((lam * (se_bits(cx0 - mvpx[l]) + se_bits(cy0 - mvpy[l]))) >> 4);
                continue;
            }
            unsigned dd[9];
#pragma unroll
            for (int s = 0; s < 9; ++s) dd[s] = (s == 4 && skip_centre) ? 0u : pu_group_sum(act[l] ? s_sat[s][myitem] : 0, l);
            if (!act[l]) continue;
            int bitx[3], bity[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { bitx[j] = se_bits(cx0 + (j - 1) * step - mvpx[l]); bity[j] = se_bits(cy0 + (j - 1) * step - mvpy[l]); }
            // ring index k (raster, 0..7) -> slot k + (k > 3); rate of slot s = lambda (bitx[s % 3] + bity[s / 3]) >> 4
            auto rate = [&](int s) { return (unsigned)((lam * (bitx[s % 3] + bity[s / 3])) >> 4); };
            int idx = -1, mvdx = 0, mvdy = 0;                     // the winner: ring index, its offset in steps
            unsigned best = bc[l], brate = 0;
            if (phase == 0) {
                const unsigned rate0 = rate(4);
                brate = rate0;
                if (SATD) best = dd[4] + rate0;                      // tME+0x64: the start cost recomputed with the sub-pel measure
                unsigned maxd = best - rate0;                        // the integer position's distortion
                auto TRY = [&](int k) { const int s = k + (k > 3); const unsigned r = rate(s), c = dd[s] + r; if (c < best) { best = c; idx = k; brate = r; mvdx = s % 3 - 1; mvdy = s / 3 - 1; } maxd = max(maxd, dd[s]); };
                TRY(3); TRY(4); TRY(1); TRY(6);
                if (!(K.diag_fast && idx == -1)) {
                    if (!K.diag_fast) { TRY(0); TRY(2); TRY(5); TRY(7); }
                    else {
                        if ((idx & ~2) == 1) TRY(0);
                        if (idx == 1 || idx == 4) TRY(2);
                        if (idx == 3 || idx == 6) TRY(5);
                        if ((idx & ~2) == 4) TRY(7);
                    }
                }
                const int thrf = (int)(((unsigned)W2 * (unsigned)K.flat) >> 3), spread = (int)(maxd - best + brate);
                qrun[l] = !(K.thr != 0 && thrf >= spread);           // a flat cost surface skips the quarter step when the configuration allows skipping
                hmx[l] = 2 * mvdx; hmy[l] = 2 * mvdy;
            } else {
                const bool fast = K.subme == 1;
                const bool up = !fast || hmy[l] >= 0, down = !fast || hmy[l] <= 0, left = !fast || hmx[l] >= 0, right = !fast || hmx[l] <= 0;
                auto TRY = [&](int k) { const int s = k + (k > 3); const unsigned c = dd[s] + rate(s); if (c < best) { best = c; idx = k; mvdx = s % 3 - 1; mvdy = s / 3 - 1; } };
                if (up) TRY(1);
                if (down) TRY(6);
                if (left) {
                    TRY(3);
                    if (up && (!fast || (idx & ~2) == 1)) TRY(0);
                    if (down && (!fast || idx == 3 || idx == 6)) TRY(5);
                }
                if (right) {
                    TRY(4);
                    if (up && (!fast || idx == 4 || idx == 1)) TRY(2);
                    if (down && (!fast || (idx & ~2) == 4)) TRY(7);
                }
            }
            bc[l] = best; bx[l] = cx0 + mvdx * step; by[l] = cy0 + mvdy * step;
        }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int G = 1 << (2 * (3 - l));                          // lanes (tiles) per PU: 64, 16, 4, 1
        if (valid[l] && (lane & (G - 1)) == 0) {
            ks265_pu o;
            if (SATD) bd[l] = bc[l] - (unsigned)((lam * (se_bits(bx[l] - mvpx[l]) + se_bits(by[l] - mvpy[l]))) >> 4);
            o.mvx = (int16_t)bx[l]; o.mvy = (int16_t)by[l]; o.mvpx = (int16_t)mvpx[l]; o.mvpy = (int16_t)mvpy[l]; o.cost = bc[l]; o.dist = bd[l];
            cp[pidx[l]] = o;
        }
    }
}

extern "C" int ks265_me_subpel(ks265_frame *f, ks265_pic src, ks265_pic ref, ks265_pu *pu)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y || !pu) return KS265_POINTER;
    const ks265_frame_cfg &c = f->cfg;
    const KsSubme K = {c.subme, c.sub_satd, c.sub_thr, c.sub_flat, c.sub_cap, c.sub_cap_step, c.sub_diag_fast};
    const dim3 grid((f->g.ctu_cols * f->g.ctu_rows + KS_SUBPEL_NC - 1) / KS_SUBPEL_NC), block(KS_SUBPEL_NC * 64);
    if (c.sub_satd) hipLaunchKernelGGL((me_subpel_kernel<KS_SUBPEL_NC, true>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, K, src.y, ref.y, pu);
    else hipLaunchKernelGGL((me_subpel_kernel<KS_SUBPEL_NC, false>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, K, src.y, ref.y, pu);
    return ks265_check_launch(f->ctx);
}


// Price of splitting an inter CU into four, in bits at the motion lambda, on top of the children's SATD + vector rate (flags, vectors, the smaller
// transforms of one TU per CU): 40 for the one-list records of P pictures, 80 for the two-list records (B pictures, multi-reference P).  At 12 (the flags
// alone, round 1) the P pictures of the test clips were 3-5 % and the P / B pictures of a hierarchical GOP 15 % (qp 27) to 32 % (qp 35) larger for 0.03-0.07 dB
// (the oracle's comment has the table).
#define KS_BI_BIAS_SHIFT 5
#define KS_SPLIT_BITS_P 40
#define KS_SPLIT_BITS_B 80
// ------------------------------------------------------------------ Stage C: CU quadtree (64 threads per CTU)
// REC = ks265_pu (P pictures: list 0 only) or ks265_pu_b (B pictures: the per-PU winner with its direction)
// cfg.intra_inter: ibest (85 per CTU from ks265_intra_candidates: cost << 6 | mode, all ones = none; else null) - a block's intra pre-selection cost + lambda x KS_INTRA_BIAS_BITS competes with
// its inter cost (oracle: node_own_cost)
#define KS_INTRA_BIAS_BITS 96
#define KS_PART_BITS 6                                              // what a CU in two partitions costs beyond its halves (the oracle's PART_BITS)
template <typename REC>
__global__ __launch_bounds__(64) void cu_decide_kernel(KsGeom g, int lam, const REC *pus, ks265_cu8 *cu8, const unsigned *ibest, const KsRect *rect)
{
    __shared__ unsigned bestc[85];
    __shared__ unsigned char split[85], use_intra[85], part[85];
    const int t = threadIdx.x, ctu = blockIdx.x, cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const REC *cp = pus + (long)ctu * 85;
    const unsigned pen = (unsigned)((lam * (std::is_same<REC, ks265_pu_b>::value ? KS_SPLIT_BITS_B : KS_SPLIT_BITS_P)) >> 4);
    for (int l = 3; l >= 0; --l) {
        const int n = 1 << l, s = 64 >> l;
        for (int i = t; i < n * n; i += 64) {
            const int px = i & (n - 1), py = i >> l, idx = ks_level_base(l) + i;
            const int x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
            unsigned own = cp[idx].cost, res; unsigned char sp = 0, ui = 0, pm = 0;
            if (rect && l < 3) {                                     // cfg.part: 2NxN, then Nx2N, each only if strictly cheaper
                const KsRect *rr = &rect[(long)ctu * 21 + idx];         // (read in place: a copy of the record indexed by pm / hf below would live in scratch memory)
                if (rr->cost[0] < own) { own = rr->cost[0]; pm = 1; }
                if (rr->cost[1] < own) { own = rr->cost[1]; pm = 2; }
            }
            part[idx] = pm;
            if (ibest && l > 0) {
                const unsigned v = ibest[(long)ctu * 85 + idx];
                if (v != 0xFFFFFFFFu) {
                    const unsigned long long ic = (unsigned long long)(v >> 6) + (unsigned long long)((lam * KS_INTRA_BIAS_BITS) >> 4);
                    if (own == KS_COST_INVALID || ic < own) { ui = 1; own = ic > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)ic; }
                }
            }
            use_intra[idx] = ui;
            if (x0 >= g.W || y0 >= g.H) res = 0;
            else if (l == 3) res = own;
            else {
                unsigned long long sum = pen;
                for (int k = 0; k < 4; ++k) sum += bestc[ks_pu_index(l + 1, px * 2 + (k & 1), py * 2 + (k >> 1))];
                if (own != KS_COST_INVALID && (unsigned long long)own <= sum) res = own;
                else { sp = 1; res = sum > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)sum; }
            }
            bestc[idx] = res; split[idx] = sp;
        }
        __syncthreads();
    }
    // emit: thread t = 8x8 block (bx, by) of the CTU; walk down from the root
    const int bx = t & 7, by = t >> 3, X = cx * 64 + bx * 8, Y = cy * 64 + by * 8;
    if (X >= g.W || Y >= g.H) return;
    int l = 0;
    while (l < 3 && split[ks_pu_index(l, bx >> (3 - l), by >> (3 - l))]) ++l;
    const int pidx = ks_pu_index(l, bx >> (3 - l), by >> (3 - l));
    const REC p = cp[pidx];
    ks265_cu8 c;
    c.mvx = p.mvx; c.mvy = p.mvy; c.log2_cu = (uint8_t)(6 - l); c.cbf = 0; c.pred_mode = 0;
    if constexpr (std::is_same<REC, ks265_pu_b>::value) {
        c.mv1x = p.mv1x; c.mv1y = p.mv1y; c.inter_dir = (uint8_t)p.inter_dir;
        const int pm = rect ? part[pidx] : 0;
        if (pm && !use_intra[pidx]) {                                // this block's half of the CU: its own direction and vector(s)
            const KsRect *rr = &rect[(long)ctu * 21 + pidx];
            const int hf = pm == 1 ? (by >> (2 - l)) & 1 : (bx >> (2 - l)) & 1;
            c.mvx = rr->mv[pm - 1][hf][0]; c.mvy = rr->mv[pm - 1][hf][1]; c.mv1x = rr->mv1[pm - 1][hf][0]; c.mv1y = rr->mv1[pm - 1][hf][1];
            c.inter_dir = rr->dir[pm - 1][hf]; c.log2_cu = (uint8_t)((6 - l) | (pm << 4));
        }
    } else {
        c.mv1x = 0; c.mv1y = 0; c.inter_dir = 1;
        const int pm = rect ? part[pidx] : 0;
        if (pm && !use_intra[pidx]) {                                // this block's half of the CU: 2NxN by its row, Nx2N by its column
            const KsRect *rr = &rect[(long)ctu * 21 + pidx];
            const int hf = pm == 1 ? (by >> (2 - l)) & 1 : (bx >> (2 - l)) & 1;
            c.mvx = rr->mv[pm - 1][hf][0]; c.mvy = rr->mv[pm - 1][hf][1]; c.log2_cu = (uint8_t)((6 - l) | (pm << 4));
        }
    }
    if (use_intra[pidx]) { c.mvx = (int16_t)(ibest[(long)ctu * 85 + pidx] & 63u); c.mvy = 0; c.mv1x = 0; c.mv1y = 0; c.pred_mode = 2; c.inter_dir = 0; }    // an intra CU: mvx = its luma mode
    cu8[(long)(Y >> 3) * g.w8 + (X >> 3)] = c;
}

__global__ __launch_bounds__(256) void cu_flat_intra_kernel(KsGeom g, ks265_cu8 *cu8)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.w8 * g.h8) return;
    int bx = i % g.w8, by = i / g.w8, lg = 3;
    for (int t = 5; t > 3; --t) {
        int n = 1 << (t - 3), ax = bx / n * n, ay = by / n * n;
        if (ax + n <= g.w8 && ay + n <= g.h8) { lg = t; break; }
    }
    ks265_cu8 c;
    c.mvx = 0; c.mvy = 0; c.mv1x = 0; c.mv1y = 0; c.log2_cu = (uint8_t)lg; c.cbf = 0; c.pred_mode = 1; c.inter_dir = 0;
    cu8[i] = c;
}

extern "C" int ks265_cu_decide_ii(ks265_frame *f, const ks265_pu *pu, const uint32_t *ibest, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!pu || !cu8) return KS265_POINTER;
    hipLaunchKernelGGL(cu_decide_kernel<ks265_pu>, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(64), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, pu, cu8, ibest, (const KsRect *)nullptr);
    return ks265_check_launch(f->ctx);
}
extern "C" int ks265_cu_decide(ks265_frame *f, const ks265_pu *pu, ks265_cu8 *cu8) { return ks265_cu_decide_ii(f, pu, nullptr, cu8); }

extern "C" int ks265_cu_decide_b_ii(ks265_frame *f, const ks265_pu_b *pub, const uint32_t *ibest, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!pub || !cu8) return KS265_POINTER;
    hipLaunchKernelGGL(cu_decide_kernel<ks265_pu_b>, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(64), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, pub, cu8, ibest, (const KsRect *)nullptr);
    return ks265_check_launch(f->ctx);
}
extern "C" int ks265_cu_decide_b(ks265_frame *f, const ks265_pu_b *pub, ks265_cu8 *cu8) { return ks265_cu_decide_b_ii(f, pub, nullptr, cu8); }

// ------------------------------------------------------------------ Stage B': bi-predictive candidate of a B picture
// One workgroup per CTU, wave = PU level, lane = 8x8 tile in Z-order: SATD of the source tile against the rounded average of the
// two list winners' plane tiles, PU cost by DPP group sum, then L0 / L1 / bi by cost (ties: L0, L1, bi).  One SATD per
// (level, tile), so the Hadamard stays on the VALU here: all 64 differences of a tile live in registers, six butterfly stages of
// plain v_add_u32 / v_sub_u32, and |.| + accumulate is ONE v_sad_u32 per coefficient (a bias of 2^15 added to difference
// (0,0) reaches every Hadamard output with weight +1, so all outputs are positive).
__device__ __forceinline__ unsigned satd8x8_regs(int (&d)[64])
{
    d[0] += 0x8000;
#pragma unroll
    for (int len = 1; len < 64; len <<= 1)
#pragma unroll
        for (int i = 0; i < 64; i += 2 * len)
#pragma unroll
            for (int j = i; j < i + len; ++j) { const int u = d[j], v = d[j + len]; d[j] = u + v; d[j + len] = u - v; }
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const unsigned bias = 0x8000u;
#pragma unroll
    for (int i = 0; i < 64; i += 4) {                                   // no clang builtin for v_sad_u32
        asm("v_sad_u32 %0, %1, %2, %0" : "+v"(a0) : "v"(d[i]), "s"(bias));
        asm("v_sad_u32 %0, %1, %2, %0" : "+v"(a1) : "v"(d[i + 1]), "s"(bias));
        asm("v_sad_u32 %0, %1, %2, %0" : "+v"(a2) : "v"(d[i + 2]), "s"(bias));
        asm("v_sad_u32 %0, %1, %2, %0" : "+v"(a3) : "v"(d[i + 3]), "s"(bias));
    }
    const unsigned acc = (a0 + a1) + (a2 + a3);
    return (acc + 2) >> 2;
}

__device__ __forceinline__ int bitx_of(unsigned long long packed, int i) { return (int)((packed >> (8 * i)) & 255); }

// 8x8 tile at an arbitrary byte address -> 16 packed dwords
__device__ __forceinline__ void load_tile8(const uint8_t *p, long stride, unsigned (&t)[16])
{
    const unsigned sh = (unsigned)((uintptr_t)p & 3);
    const uint8_t *q = p - sh;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const unsigned *rw = (const unsigned *)(q + r * stride);
        const unsigned a0 = rw[0], a1 = rw[1], a2 = rw[2];
        t[2 * r] = align_bytes(a1, a0, sh); t[2 * r + 1] = align_bytes(a2, a1, sh);
    }
}

// SATD of the source tile f against the rounded average of two prediction tiles (16 packed dwords each: luma_pred_tile8)
__device__ __forceinline__ unsigned satd8x8_avg(const unsigned (&f)[16], const unsigned (&A)[16], const unsigned (&B)[16])
{
    int d[64];
#pragma unroll
    for (int w = 0; w < 16; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int avg = (int)(((A[w] >> (8 * i)) & 255) + ((B[w] >> (8 * i)) & 255) + 1) >> 1;
            d[w * 4 + i] = (int)((f[w] >> (8 * i)) & 255) - avg;
        }
    return satd8x8_regs(d);
}

// cfg.bi_refine - joint refinement of the pair (motionSearchBI enc@0x484910 / interMeBiFull_opt enc@0x4898e0 / interMeBiFull_c enc@0x4896d0), per
// (level, tile) lane like the rest of the kernel.  The cheaper list keeps its vector; T = clip8(2 org - pred_kept) (calcBiMeOrg enc@0x47b1a0) is the
// target of the other one.  Integer step: the 15 x 15 window of the tile sits in 60 registers (rows realigned once), the 64 positions are SADs of
// register rows (v_sad_u8, static v_alignbyte per column), summed over the PU's lanes by DPP, + vector rate, first minimum in row-major order
// (key = cost << 6 | position).  Sub-pel step: the two rings of stage B on T, SAD + rate like the integer step.  The refined pair replaces the decision when
// its SATD against the rounded average + both vector rates is lower.
// The joint refinement of a bi-predictive pair for the lane's 8 x 8 tile of its PU: a / b = the lists' records, PA / PB their prediction tiles, f the source
// tile, gs = the sum over the PU's lanes, o = the decision so far (takes the refined pair if it is cheaper).  Called by every lane of the wave.
template <class GS>
__device__ __forceinline__ void bi_refine_lane(const KsGeom &g, int lam, int cx, int cy, long base, bool valid, const GS &gs, const ks265_pu &a, const ks265_pu &b,
                                               const uint8_t *ref0, const uint8_t *ref1, const unsigned (&f)[16], const unsigned (&PA)[16], const unsigned (&PB)[16],
                                               unsigned rbits, unsigned dir3, ks265_pu_b &o)
{
    const int ax = valid ? a.mvx : 0, ay = valid ? a.mvy : 0, bx = valid ? b.mvx : 0, by = valid ? b.mvy : 0;
    const bool keep1 = valid && b.cost < a.cost;                 // list whose vector stays (uniform over the PU's lanes)
    const uint8_t *refO = keep1 ? ref0 : ref1;
    const int omx = keep1 ? ax : bx, omy = keep1 ? ay : by;
    const int opx = valid ? (keep1 ? a.mvpx : b.mvpx) : 0, opy = valid ? (keep1 ? a.mvpy : b.mvpy) : 0;
    unsigned T[16];
    {
        unsigned k[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) k[i] = keep1 ? PB[i] : PA[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned w = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int v = 2 * (int)((f[i] >> (8 * q)) & 255) - (int)((k[i] >> (8 * q)) & 255);
                w |= (unsigned)clip3(0, 255, v) << (8 * q);
            }
            T[i] = w;
        }
    }
    const int xe = min(cx * 64 + 64, g.W), ye = min(cy * 64 + 64, g.H);
    const int lox = -64 - cx * 64, hix = g.W + 64 - xe, loy = -64 - cy * 64, hiy = g.H + 64 - ye;
    int imx = clip3(4 * lox, 4 * hix, omx) >> 2, imy = clip3(4 * loy, 4 * hiy, omy) >> 2;
    imx = imx <= lox + 3 ? lox + 4 : (imx >= hix - 3 ? hix - 4 : imx);
    imy = imy <= loy + 3 ? loy + 4 : (imy >= hiy - 3 ? hiy - 4 : imy);
    const int sx = valid ? imx - 3 - (opx < 0) : 0, sy = valid ? imy - 3 - (opy < 0) : 0;   // idle lanes read around the picture origin
    unsigned long long bitx = 0;                                   // eight column bit counts (< 64 each), one byte apiece: indexed by the runtime column
    int bity[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bitx |= (unsigned long long)se_bits(4 * (sx + i) - opx) << (8 * i); bity[i] = se_bits(4 * (sy + i) - opy); }
    unsigned bkey = 0xFFFFFFFFu;
    {
        unsigned w[15][4];
        const uint8_t *r0 = refO + base + (long)sy * g.sy + sx;
        const unsigned sh = (unsigned)((uintptr_t)r0 & 3);
        const uint8_t *q0 = r0 - sh;
#pragma unroll
        for (int r = 0; r < 15; ++r) {
            const unsigned *rw = (const unsigned *)(q0 + (long)r * g.sy);
            const unsigned a0 = rw[0], a1 = rw[1], a2 = rw[2], a3 = rw[3], a4 = rw[4];
            w[r][0] = align_bytes(a1, a0, sh); w[r][1] = align_bytes(a2, a1, sh); w[r][2] = align_bytes(a3, a2, sh); w[r][3] = align_bytes(a4, a3, sh);
        }
#pragma unroll 1
        for (int dx = 0; dx < 8; ++dx) {                          // columns one after the other: the window rows move one byte per step
#pragma unroll
            for (int dy = 0; dy < 8; ++dy) {
                unsigned sd2 = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) { sd2 = sad_u8x4(T[2 * r], w[dy + r][0], sd2); sd2 = sad_u8x4(T[2 * r + 1], w[dy + r][1], sd2); }
                const unsigned tot = gs(valid ? sd2 : 0) + (unsigned)((lam * (bitx_of(bitx, dx) + bity[dy])) >> 4);
                bkey = min(bkey, (tot << 6) | (unsigned)(dy * 8 + dx));
            }
#pragma unroll
            for (int r = 0; r < 15; ++r) {
                w[r][0] = align_bytes(w[r][1], w[r][0], 1); w[r][1] = align_bytes(w[r][2], w[r][1], 1);
                w[r][2] = align_bytes(w[r][3], w[r][2], 1); w[r][3] >>= 8;
            }
        }
    }
    int rbx = 4 * (sx + (int)(bkey & 7)), rby = 4 * (sy + (int)((bkey >> 3) & 7));
    unsigned bc = bkey >> 6;                                                   // SAD + rate of the integer winner
    // The two sub-pel rings (half, then quarter steps around the running best).  Round 4: the eight candidates of a ring share their horizontal filtering - the three x
    // positions are filtered once each over the 16 input rows the three y positions tap (raw 16-bit tap sums, row pairs packed for v_dot2_i32_i16), the vertical taps
    // of each y position run on those; ONE instruction stream for every fraction (the integer position is the tap set {0 0 0 64 0 0 0 0}; (sum + 2048) >> 12 equals the
    // one-dimensional filters' (sum + 32) >> 6 and the plain sample exactly), so lanes with different fractions do not diverge.  Before: a full separable
    // interpolation per candidate and lane, its three fraction cases executed under divergence (750 us per 2160p B picture).
#pragma unroll 1
    for (int step = 2; step >= 1; --step) {
        const int c0x = rbx, c0y = rby;
        const int ybase = (c0y - step) >> 2;                                   // integer row of output row 0 at the ring's top y position
        unsigned cc[9];
#pragma unroll 1
        for (int gx = 0; gx < 3; ++gx) {
            const int axq = c0x + (gx - 1) * step;
            int tl, th;
            luma_taps_packed(axq & 3, tl, th);
            const uint8_t *hp = refO + base + (long)(ybase - 3) * g.sy + (axq >> 2);
            unsigned HP[8][8];                                                  // [pixel][row pair]: rows 2 j, 2 j + 1 of the 16 filtered rows ybase - 3 .. ybase + 12
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int h0[8], h1[8];
                luma_hrow8(hp + (long)(2 * j) * g.sy, tl, th, h0);
                luma_hrow8(hp + (long)(2 * j + 1) * g.sy, tl, th, h1);
#pragma unroll
                for (int i = 0; i < 8; ++i) HP[i][j] = ((unsigned)h0[i] & 0xFFFFu) | ((unsigned)h1[i] << 16);
            }
#pragma unroll 1
            for (int gy = 0; gy < 3; ++gy) {
                if (gx == 1 && gy == 1) continue;
                const int ayq = c0y + (gy - 1) * step, roff = (ayq >> 2) - ybase;     // 0 or 1
                int c[8];
                luma_taps(ayq & 3, c);
                // output row r reads input rows r + roff .. r + roff + 7; in pairs: an even start takes (c0 c1)(c2 c3)(c4 c5)(c6 c7), an odd one (0 c0)(c1 c2)(c3 c4)(c5 c6)(c7 0)
                unsigned E0[5], E1[5];
#pragma unroll
                for (int k = 0; k < 4; ++k) E0[k] = ((unsigned)c[2 * k] & 0xFFFFu) | ((unsigned)c[2 * k + 1] << 16);
                E0[4] = 0;
                E1[0] = (unsigned)c[0] << 16;
#pragma unroll
                for (int k = 1; k < 4; ++k) E1[k] = ((unsigned)c[2 * k - 1] & 0xFFFFu) | ((unsigned)c[2 * k] << 16);
                E1[4] = (unsigned)c[7] & 0xFFFFu;
                unsigned We[5], Wo[5];                                            // even r: window of pairs r / 2 ..; odd r: (r - 1) / 2 ..
#pragma unroll
                for (int k = 0; k < 5; ++k) { We[k] = roff ? E1[k] : E0[k]; Wo[k] = roff ? (k ? E0[k - 1] : 0u) : E1[k]; }
                unsigned sd3 = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    int px[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        int v = 2048;
#pragma unroll
                        for (int k = 0; k < 5; ++k)
                            v = __builtin_amdgcn_sdot2(__builtin_bit_cast(ks_s16x2, (r & 1) ? Wo[k] : We[k]), __builtin_bit_cast(ks_s16x2, HP[i][(r >> 1) + k < 8 ? (r >> 1) + k : 7]), v, false);
                        px[i] = clip8(ks_no_pk(v >> 12));
                    }
                    const uint2 pr = ks_pack_row8(px);
                    sd3 = sad_u8x4(T[2 * r], pr.x, sd3); sd3 = sad_u8x4(T[2 * r + 1], pr.y, sd3);
                }
                cc[gy * 3 + gx] = gs(valid ? sd3 : 0) + (unsigned)mv_cost(axq, ayq, opx, opy, lam);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kk = k < 4 ? k : k + 1;                                  // ring order of stage B: (-1,-1) (0,-1) (1,-1) (-1,0) (1,0) (-1,1) (0,1) (1,1), first strict minimum
            if (cc[kk] < bc) { bc = cc[kk]; rbx = c0x + (kk % 3 - 1) * step; rby = c0y + (kk / 3 - 1) * step; }
        }
    }
    unsigned PO[16], PK[16];
    luma_pred_tile8(refO + base, g.sy, rbx, rby, PO);
#pragma unroll
    for (int i = 0; i < 16; ++i) PK[i] = keep1 ? PB[i] : PA[i];
    const unsigned d2 = gs(valid ? satd8x8_avg(f, PK, PO) : 0);
    if (valid) {
        unsigned c2 = d2 + (unsigned)(keep1 ? mv_cost(b.mvx, b.mvy, b.mvpx, b.mvpy, lam) : mv_cost(a.mvx, a.mvy, a.mvpx, a.mvpy, lam))
                      + (unsigned)mv_cost(rbx, rby, opx, opy, lam) + rbits;
        c2 -= c2 >> KS_BI_BIAS_SHIFT;
        if (c2 < o.cost) {
            o.cost = c2; o.inter_dir = dir3;
            if (keep1) { o.mvx = (int16_t)rbx; o.mvy = (int16_t)rby; } else { o.mv1x = (int16_t)rbx; o.mv1y = (int16_t)rby; }
        }
    }
}

template <bool REFINE, bool MR>
__global__ __launch_bounds__(256, REFINE ? 2 : 1) void bi_decide_kernel(KsGeom g, int lam, const uint8_t *src, const uint8_t *ref0_, const uint8_t *ref1_,
                                                        const ks265_pu *pu0, const ks265_pu *pu1, ks265_pu_b *pub, const KsMrefB mr)
{
    const int tid = threadIdx.x, lane = tid & 63, level = tid >> 6;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int G = 1 << (2 * (3 - level));
    const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;
    const int px = tx >> (3 - level), py = ty >> (3 - level), pidx = ks_level_base(level) + py * (1 << level) + px;
    const ks265_pu a = pu0[(long)ctu * 85 + pidx], b = pu1[(long)ctu * 85 + pidx];
    const bool valid = a.cost != KS_COST_INVALID;
    // several pictures per list (MR): the two records are the lists' winners (ks265_ref_pick: their costs hold the index bits), i0 / i1 their pictures
    const int i0 = MR ? mr.idx0[(long)ctu * 85 + pidx] : 0, i1 = MR ? mr.idx1[(long)ctu * 85 + pidx] : 0;
    const uint8_t *const ref0 = MR ? ks_pick(mr.y0, i0) : ref0_, *const ref1 = MR ? ks_pick(mr.y1, i1) : ref1_;
    const unsigned rbits = MR ? (unsigned)((i0 == 0 ? mr.bits0[0] : i0 == 1 ? mr.bits0[1] : i0 == 2 ? mr.bits0[2] : mr.bits0[3]) + (i1 == 0 ? mr.bits1[0] : i1 == 1 ? mr.bits1[1] : i1 == 2 ? mr.bits1[2] : mr.bits1[3])) : 0u;
    auto bidir = [&](unsigned d) { return d | ((d & 1u) ? (unsigned)i0 << 4 : 0u) | ((d & 2u) ? (unsigned)i1 << 6 : 0u); };
    ks265_pu_b o;
    o.mvx = a.mvx; o.mvy = a.mvy; o.mv1x = b.mvx; o.mv1y = b.mvy; o.cost = a.cost; o.inter_dir = bidir(1);
    if (__any(valid)) {
        const uint8_t *Sp = ks_org_y(g, src);
        unsigned f[16];
        const uint8_t *frow = Sp + (long)(valid ? y0 : cy * 64) * g.sy + (valid ? x0 : cx * 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const uint2 v = *(const uint2 *)(frow + (long)r * g.sy); f[2 * r] = v.x; f[2 * r + 1] = v.y; }
        const long base = (long)(valid ? y0 : 0) * g.sy + (valid ? x0 : 0) + g.org_y;
        const int ax = valid ? a.mvx : 0, ay = valid ? a.mvy : 0, bx = valid ? b.mvx : 0, by = valid ? b.mvy : 0;
        unsigned PA[16], PB[16];                                       // the two lists' prediction tiles, interpolated from the reference pictures (interp_dev.h)
        luma_pred_tile8(ref0 + base, g.sy, ax, ay, PA);
        luma_pred_tile8(ref1 + base, g.sy, bx, by, PB);
        const unsigned sd = satd8x8_avg(f, PA, PB);
        const unsigned dd = pu_group_sum(valid ? sd : 0, level);
        if (valid) {
            if (b.cost < o.cost) { o.cost = b.cost; o.inter_dir = bidir(2); }
            unsigned c = dd + (unsigned)mv_cost(a.mvx, a.mvy, a.mvpx, a.mvpy, lam) + (unsigned)mv_cost(b.mvx, b.mvy, b.mvpx, b.mvpy, lam) + rbits;
            c -= c >> KS_BI_BIAS_SHIFT;                            // a bi-predictive pair counts 31 / 32 of its cost (the oracle's BI_BIAS_SHIFT: - 3.3 % bytes on hierarchical B)
            if (c < o.cost) { o.cost = c; o.inter_dir = bidir(3); }
        }
        if (REFINE) bi_refine_lane(g, lam, cx, cy, base, valid, KsGroupUniform{level}, a, b, ref0, ref1, f, PA, PB, rbits, bidir(3), o);
    }
    if ((lane & (G - 1)) == 0) pub[(long)ctu * 85 + pidx] = o;
}

// cfg.bi_refine == 2 (round 5): the joint refinement AFTER the CU decision, for the CUs it chose - every picture area refined once, not once per quadtree level
// (bi_decide_kernel<false> has paired the lists' winners).  One wave per CTU, lane = 8 x 8 tile = one ks265_cu8 record; the lane's PU is the
// 2N x 2N inter CU its block belongs to (lanes of intra CUs, of CUs in halves and outside the picture idle), the level differs from lane to lane (KsGroupPerLane).
// A refined pair that is cheaper goes into the PU's record (the merge pass compares against its cost) and into the CU's blocks.
template <bool MR>
__global__ __launch_bounds__(64, 2) void bi_refine_chosen_kernel(KsGeom g, int lam, const uint8_t *src, const uint8_t *ref0_, const uint8_t *ref1_,
                                                                 const ks265_pu *pu0, const ks265_pu *pu1, ks265_pu_b *pub, ks265_cu8 *cu8, const KsMrefB mr)
{
    const int lane = threadIdx.x;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8, w8 = g.W >> 3;
    const bool inside = x0 < g.W && y0 < g.H;
    ks265_cu8 *const blk = cu8 + (long)(inside ? y0 >> 3 : 0) * w8 + (inside ? x0 >> 3 : 0);
    const ks265_cu8 c = *blk;
    const bool chosen = inside && c.pred_mode == 0 && c.log2_cu >= 3 && c.log2_cu <= 6;      // (bits 4..5 set: a CU in halves)
    const int level = chosen ? 6 - c.log2_cu : 3;
    const int px = tx >> (3 - level), py = ty >> (3 - level), pidx = ks_level_base(level) + py * (1 << level) + px;
    const ks265_pu a = pu0[(long)ctu * 85 + pidx], b = pu1[(long)ctu * 85 + pidx];
    const bool valid = chosen && a.cost != KS_COST_INVALID;
    if (!__any(valid)) return;
    const int i0 = MR ? mr.idx0[(long)ctu * 85 + pidx] : 0, i1 = MR ? mr.idx1[(long)ctu * 85 + pidx] : 0;
    const uint8_t *const ref0 = MR ? ks_pick(mr.y0, i0) : ref0_, *const ref1 = MR ? ks_pick(mr.y1, i1) : ref1_;
    const unsigned rbits = MR ? (unsigned)((i0 == 0 ? mr.bits0[0] : i0 == 1 ? mr.bits0[1] : i0 == 2 ? mr.bits0[2] : mr.bits0[3]) + (i1 == 0 ? mr.bits1[0] : i1 == 1 ? mr.bits1[1] : i1 == 2 ? mr.bits1[2] : mr.bits1[3])) : 0u;
    ks265_pu_b o = pub[(long)ctu * 85 + pidx];
    const unsigned before = o.cost;
    const uint8_t *Sp = ks_org_y(g, src);
    unsigned f[16];
    const uint8_t *frow = Sp + (long)(valid ? y0 : cy * 64) * g.sy + (valid ? x0 : cx * 64);
#pragma unroll
    for (int r = 0; r < 8; ++r) { const uint2 v = *(const uint2 *)(frow + (long)r * g.sy); f[2 * r] = v.x; f[2 * r + 1] = v.y; }
    const long base = (long)(valid ? y0 : 0) * g.sy + (valid ? x0 : 0) + g.org_y;
    unsigned PA[16], PB[16];
    luma_pred_tile8(ref0 + base, g.sy, valid ? a.mvx : 0, valid ? a.mvy : 0, PA);
    luma_pred_tile8(ref1 + base, g.sy, valid ? b.mvx : 0, valid ? b.mvy : 0, PB);
    bi_refine_lane(g, lam, cx, cy, base, valid, KsGroupPerLane{level}, a, b, ref0, ref1, f, PA, PB, rbits, 3u | (unsigned)i0 << 4 | (unsigned)i1 << 6, o);
    if (valid && o.cost != before) {
        if ((lane & ((1 << (2 * (3 - level))) - 1)) == 0) pub[(long)ctu * 85 + pidx] = o;
        blk->mvx = o.mvx; blk->mvy = o.mvy; blk->mv1x = o.mv1x; blk->mv1y = o.mv1y; blk->inter_dir = (uint8_t)o.inter_dir;
    }
}

// the lists of the multi-reference B picture being coded as kernel arguments (entries past a list's size repeat its last picture); all zero when none is
static KsMrefB ks_mrefb(const ks265_frame *f)
{
    KsMrefB m{};
    if (!f->mrefb) return m;
    for (int i = 0; i < 4; ++i) {
        m.y0.p[i] = f->mr_pic[0][i < f->mr_n[0] ? i : f->mr_n[0] - 1].y; m.y1.p[i] = f->mr_pic[1][i < f->mr_n[1] ? i : f->mr_n[1] - 1].y;
        const int b0 = f->mr_n[0] <= 1 ? 0 : (i < f->mr_n[0] - 1 ? i + 1 : f->mr_n[0] - 1), b1 = f->mr_n[1] <= 1 ? 0 : (i < f->mr_n[1] - 1 ? i + 1 : f->mr_n[1] - 1);
        m.bits0[i] = (f->cfg.lambda_q4 * b0) >> 4; m.bits1[i] = (f->cfg.lambda_q4 * b1) >> 4;
    }
    m.idx0 = f->ridx[0]; m.idx1 = f->ridx[1];
    return m;
}

extern "C" int ks265_bi_decide(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *pu0, const ks265_pu *pu1,
                               ks265_pu_b *pub)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref0.y || !ref1.y || !pu0 || !pu1 || !pub) return KS265_POINTER;
    const dim3 grid(f->g.ctu_cols * f->g.ctu_rows), block(256);
    const KsMrefB mr = ks_mrefb(f);
    if (f->mrefb) {
        if (f->cfg.bi_refine == 1) hipLaunchKernelGGL((bi_decide_kernel<true, true>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, mr);
        else hipLaunchKernelGGL((bi_decide_kernel<false, true>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, mr);
    } else {
        if (f->cfg.bi_refine == 1) hipLaunchKernelGGL((bi_decide_kernel<true, false>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, mr);
        else hipLaunchKernelGGL((bi_decide_kernel<false, false>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, mr);
    }
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_bi_refine_chosen(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *pu0, const ks265_pu *pu1,
                                      ks265_pu_b *pub, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref0.y || !ref1.y || !pu0 || !pu1 || !pub || !cu8) return KS265_POINTER;
    const dim3 grid(f->g.ctu_cols * f->g.ctu_rows), block(64);
    const KsMrefB mr = ks_mrefb(f);
    if (f->mrefb) hipLaunchKernelGGL((bi_refine_chosen_kernel<true>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, cu8, mr);
    else hipLaunchKernelGGL((bi_refine_chosen_kernel<false>), grid, block, 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, cu8, mr);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ cfg.part: the halves of every 64 / 32 / 16 CU of a P picture (-part 1: 2NxN / Nx2N)
// The reference searches every rectangular PU on its own inside its RD loop (closed code); the frame-parallel form prices each half with the vectors the square search
// already refined for this area: the CU's own and those of the half's two quarter-size PUs, by Hadamard cost of the half's prediction + rate against the CU's predictor,
// first strict minimum in that order (the oracle's rect_eval).  One work-group per CTU, wave = CU level (0 .. 2), lane = 8x8 tile in z-order: a tile needs its SATD under
// four vectors - the CU's (P), its own quarter's (O) and those of the quarter's horizontal (H) and vertical (V) neighbour; quarter sums by DPP, the neighbour quarters'
// sums by two lane exchanges; equal vectors share their SATD.
__device__ __forceinline__ unsigned satd8x8_one(const unsigned (&f)[16], const unsigned (&A)[16])
{
    int d[64];
#pragma unroll
    for (int w = 0; w < 16; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) d[w * 4 + i] = (int)((f[w] >> (8 * i)) & 255) - (int)((A[w] >> (8 * i)) & 255);
    return satd8x8_regs(d);
}
__global__ __launch_bounds__(192) void rect_eval_kernel(KsGeom g, int lam, const uint8_t *src, const uint8_t *ref, const ks265_pu *pus, KsRect *rect)
{
    const int tid = threadIdx.x, lane = tid & 63, l = tid >> 6;                  // l = level of the CU (0: 64, 1: 32, 2: 16)
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;
    const int cux = tx >> (3 - l), cuy = ty >> (3 - l), qx = (tx >> (2 - l)) & 1, qy = (ty >> (2 - l)) & 1;
    const int pidx = ks_level_base(l) + cuy * (1 << l) + cux;
    const ks265_pu *cp = pus + (long)ctu * 85;
    const ks265_pu P = cp[pidx];
    const bool valid = P.cost != KS_COST_INVALID;                                // the CU lies inside the picture, and so do its quarters
    const int cb = ks_level_base(l + 1), cw = 2 << l;
    const ks265_pu cO = cp[cb + (2 * cuy + qy) * cw + 2 * cux + qx], cH = cp[cb + (2 * cuy + qy) * cw + 2 * cux + (qx ^ 1)], cV = cp[cb + (2 * cuy + (qy ^ 1)) * cw + 2 * cux + qx];
    const int vP = valid ? ((int)(unsigned short)P.mvx | ((int)P.mvy << 16)) : 0;
    const int vO = valid && cO.cost != KS_COST_INVALID ? ((int)(unsigned short)cO.mvx | ((int)cO.mvy << 16)) : vP;
    const int vH = valid && cH.cost != KS_COST_INVALID ? ((int)(unsigned short)cH.mvx | ((int)cH.mvy << 16)) : vP;
    const int vV = valid && cV.cost != KS_COST_INVALID ? ((int)(unsigned short)cV.mvx | ((int)cV.mvy << 16)) : vP;
    unsigned f[16];
    {
        const uint8_t *frow = ks_org_y(g, src) + (long)(valid ? y0 : cy * 64) * g.sy + (valid ? x0 : cx * 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const uint2 v = *(const uint2 *)(frow + (long)r * g.sy); f[2 * r] = v.x; f[2 * r + 1] = v.y; }
    }
    const long base = (long)(valid ? y0 : 0) * g.sy + (valid ? x0 : 0) + g.org_y;
    auto tile = [&](int v) { unsigned A[16]; luma_pred_tile8(ref + base, g.sy, (int)(short)(v & 0xFFFF), v >> 16, A); return satd8x8_one(f, A); };
    unsigned sP = 0, sO = 0, sH = 0, sV = 0;
    if (__any(valid)) {
        sP = tile(vP);
        sO = sP; sH = sP; sV = sP;
        if (__any(vO != vP)) { const unsigned t = tile(vO); if (vO != vP) sO = t; }
        if (__any(vH != vP && vH != vO)) { const unsigned t = tile(vH); if (vH != vP && vH != vO) sH = t; }
        if (vH == vO) sH = sO;
        if (__any(vV != vP && vV != vO && vV != vH)) { const unsigned t = tile(vV); if (vV != vP && vV != vO && vV != vH) sV = t; }
        if (vV == vO) sV = sO; else if (vV == vH && vV != vP) sV = sH;
    }
    if (!valid) { sP = sO = sH = sV = 0; }
    // sums over this tile's quarter (= a PU of level l + 1), then the two neighbour quarters' sums
    const unsigned qP = pu_group_sum(sP, l + 1), qO = pu_group_sum(sO, l + 1), qH = pu_group_sum(sH, l + 1), qV = pu_group_sum(sV, l + 1);
    const int gq = 1 << (2 * (2 - l));                                              // lanes per quarter: 16, 4, 1
    const unsigned hP = (unsigned)__shfl_xor((int)qP, gq, 64), hO = (unsigned)__shfl_xor((int)qO, gq, 64), hH = (unsigned)__shfl_xor((int)qH, gq, 64);
    const unsigned wP = (unsigned)__shfl_xor((int)qP, 2 * gq, 64), wO = (unsigned)__shfl_xor((int)qO, 2 * gq, 64), wV = (unsigned)__shfl_xor((int)qV, 2 * gq, 64);
    const int px = P.mvpx, py = P.mvpy;
    auto rate = [&](int v) { return (unsigned)mv_cost((int)(short)(v & 0xFFFF), v >> 16, px, py, lam); };
    // this lane's half of each orientation: candidates in the order CU vector, first quarter's, second quarter's (2NxN: quarters (0, qy), (1, qy); Nx2N: (qx, 0), (qx, 1))
    unsigned best[2]; int bv[2];
    {
        const unsigned cP = qP + hP + rate(vP), cMe = qO + hH + rate(vO), cNb = qH + hO + rate(vH);      // 2NxN: the half = my quarter + its horizontal neighbour
        const unsigned c0 = qx ? cNb : cMe, c1 = qx ? cMe : cNb; const int v0 = qx ? vH : vO, v1 = qx ? vO : vH;
        best[0] = cP; bv[0] = vP;
        if (c0 < best[0]) { best[0] = c0; bv[0] = v0; }
        if (c1 < best[0]) { best[0] = c1; bv[0] = v1; }
    }
    {
        const unsigned cP = qP + wP + rate(vP), cMe = qO + wV + rate(vO), cNb = qV + wO + rate(vV);      // Nx2N: my quarter + its vertical neighbour
        const unsigned c0 = qy ? cNb : cMe, c1 = qy ? cMe : cNb; const int v0 = qy ? vV : vO, v1 = qy ? vO : vV;
        best[1] = cP; bv[1] = vP;
        if (c0 < best[1]) { best[1] = c0; bv[1] = v0; }
        if (c1 < best[1]) { best[1] = c1; bv[1] = v1; }
    }
    // the other half: 2NxN = the quarters below / above (vertical neighbour), Nx2N = the horizontal neighbour
    const unsigned ob0 = (unsigned)__shfl_xor((int)best[0], 2 * gq, 64), ob1 = (unsigned)__shfl_xor((int)best[1], gq, 64);
    const int ov0 = __shfl_xor(bv[0], 2 * gq, 64), ov1 = __shfl_xor(bv[1], gq, 64);
    const int G = 1 << (2 * (3 - l));
    if (valid && (lane & (G - 1)) == 0) {                                         // the CU's first tile: quarter (0, 0) = first half of both orientations
        KsRect o;
        const unsigned long long pen = (unsigned long long)((lam * KS_PART_BITS) >> 4);
        const unsigned long long t0 = (unsigned long long)best[0] + ob0 + pen, t1 = (unsigned long long)best[1] + ob1 + pen;
        o.cost[0] = (bv[0] != vP || ov0 != vP) ? (t0 > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)t0) : KS_COST_INVALID;      // considered only if a half moves off the CU's vector
        o.cost[1] = (bv[1] != vP || ov1 != vP) ? (t1 > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)t1) : KS_COST_INVALID;
        o.mv[0][0][0] = (short)(bv[0] & 0xFFFF); o.mv[0][0][1] = (short)(bv[0] >> 16); o.mv[0][1][0] = (short)(ov0 & 0xFFFF); o.mv[0][1][1] = (short)(ov0 >> 16);
        o.mv[1][0][0] = (short)(bv[1] & 0xFFFF); o.mv[1][0][1] = (short)(bv[1] >> 16); o.mv[1][1][0] = (short)(ov1 & 0xFFFF); o.mv[1][1][1] = (short)(ov1 >> 16);
        rect[(long)ctu * 21 + pidx] = o;
    } else if (!valid && (lane & (G - 1)) == 0) {
        KsRect o; o.cost[0] = o.cost[1] = KS_COST_INVALID;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { o.mv[a][b][0] = 0; o.mv[a][b][1] = 0; }
        rect[(long)ctu * 21 + pidx] = o;
    }
}

// cfg.part in B pictures (round 5): the halves take MOTIONS - direction and vector(s) of the CU's own record or of one of the half's two quarter-size records, as the
// bi-predictive decision left them (oracle: rect_eval_b).  Same lane layout and exchanges as rect_eval_kernel; a tile's SATD under a motion is against the interpolated
// samples of its list or against the rounded average of both lists' 8-bit predictions (bi_decide_kernel's measure), a bi-predictive half counts 31 / 32.
struct KsMot { int v0, v1, dir; };
__device__ __forceinline__ bool mot_same(const KsMot &a, const KsMot &b) { return a.dir == b.dir && (!(a.dir & 1) || a.v0 == b.v0) && (!(a.dir & 2) || a.v1 == b.v1); }
template <bool MR>
__global__ __launch_bounds__(192) void rect_eval_b_kernel(KsGeom g, int lam, const uint8_t *src, const uint8_t *ref0, const uint8_t *ref1, const ks265_pu *pu0, const ks265_pu *pu1,
                                                          const ks265_pu_b *pubs, KsRect *rect, const KsMrefB mr)
{
    const int tid = threadIdx.x, lane = tid & 63, l = tid >> 6;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;
    const int cux = tx >> (3 - l), cuy = ty >> (3 - l), qx = (tx >> (2 - l)) & 1, qy = (ty >> (2 - l)) & 1;
    const int pidx = ks_level_base(l) + cuy * (1 << l) + cux;
    const ks265_pu_b *cp = pubs + (long)ctu * 85;
    const ks265_pu_b P = cp[pidx];
    const ks265_pu a = pu0[(long)ctu * 85 + pidx], b = pu1[(long)ctu * 85 + pidx];
    const bool valid = P.cost != KS_COST_INVALID;
    const int cb = ks_level_base(l + 1), cw = 2 << l;
    const ks265_pu_b cO = cp[cb + (2 * cuy + qy) * cw + 2 * cux + qx], cH = cp[cb + (2 * cuy + qy) * cw + 2 * cux + (qx ^ 1)], cV = cp[cb + (2 * cuy + (qy ^ 1)) * cw + 2 * cux + qx];
    auto mot = [&](const ks265_pu_b &r) { KsMot m; m.v0 = (int)(unsigned short)r.mvx | ((int)r.mvy << 16); m.v1 = (int)(unsigned short)r.mv1x | ((int)r.mv1y << 16); m.dir = (int)(MR ? (r.inter_dir & 255u) : (r.inter_dir & 3u)); return m; };      // (several pictures per list: the whole byte - the motion's pictures belong to it)
    KsMot mP = mot(P);
    if (!valid) { mP.v0 = 0; mP.v1 = 0; mP.dir = 1; }
    const KsMot mO = valid && cO.cost != KS_COST_INVALID ? mot(cO) : mP, mH = valid && cH.cost != KS_COST_INVALID ? mot(cH) : mP, mV = valid && cV.cost != KS_COST_INVALID ? mot(cV) : mP;
    unsigned f[16];
    {
        const uint8_t *frow = ks_org_y(g, src) + (long)(valid ? y0 : cy * 64) * g.sy + (valid ? x0 : cx * 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const uint2 v = *(const uint2 *)(frow + (long)r * g.sy); f[2 * r] = v.x; f[2 * r + 1] = v.y; }
    }
    const long base = (long)(valid ? y0 : 0) * g.sy + (valid ? x0 : 0) + g.org_y;
    auto tile = [&](const KsMot &m) {
        unsigned A[16], B[16];
        const uint8_t *r0 = MR ? ks_pick(mr.y0, (m.dir >> 4) & 3) : ref0, *r1 = MR ? ks_pick(mr.y1, (m.dir >> 6) & 3) : ref1;
        if (__any(m.dir & 1)) luma_pred_tile8(r0 + base, g.sy, (int)(short)(m.v0 & 0xFFFF), m.v0 >> 16, A);
        if (__any(m.dir & 2)) luma_pred_tile8(r1 + base, g.sy, (int)(short)(m.v1 & 0xFFFF), m.v1 >> 16, B);
        if ((m.dir & 3) == 1) { for (int i = 0; i < 16; ++i) B[i] = A[i]; }
        else if ((m.dir & 3) == 2) { for (int i = 0; i < 16; ++i) A[i] = B[i]; }
        return satd8x8_avg(f, A, B);                                  // one list: (p + p + 1) >> 1 = p
    };
    unsigned sP = 0, sO = 0, sH = 0, sV = 0;
    if (__any(valid)) {
        sP = tile(mP);
        sO = sP; sH = sP; sV = sP;
        const bool nO = !mot_same(mO, mP);
        if (__any(nO)) { const unsigned t = tile(mO); if (nO) sO = t; }
        const bool nH = !mot_same(mH, mP) && !mot_same(mH, mO);
        if (__any(nH)) { const unsigned t = tile(mH); if (nH) sH = t; }
        if (mot_same(mH, mO)) sH = sO;
        const bool nV = !mot_same(mV, mP) && !mot_same(mV, mO) && !mot_same(mV, mH);
        if (__any(nV)) { const unsigned t = tile(mV); if (nV) sV = t; }
        if (mot_same(mV, mO)) sV = sO; else if (mot_same(mV, mH) && !mot_same(mV, mP)) sV = sH;
    }
    if (!valid) { sP = sO = sH = sV = 0; }
    const unsigned qP = pu_group_sum(sP, l + 1), qO = pu_group_sum(sO, l + 1), qH = pu_group_sum(sH, l + 1), qV = pu_group_sum(sV, l + 1);
    const int gq = 1 << (2 * (2 - l));
    const unsigned hP = (unsigned)__shfl_xor((int)qP, gq, 64), hO = (unsigned)__shfl_xor((int)qO, gq, 64), hH = (unsigned)__shfl_xor((int)qH, gq, 64);
    const unsigned wP = (unsigned)__shfl_xor((int)qP, 2 * gq, 64), wO = (unsigned)__shfl_xor((int)qO, 2 * gq, 64), wV = (unsigned)__shfl_xor((int)qV, 2 * gq, 64);
    auto price = [&](unsigned satd, const KsMot &m) {                 // Hadamard cost of the half + the rate of every vector used against the CU's predictor of that list; bi at 31 / 32
        unsigned c = satd;
        if (m.dir & 1) c += (unsigned)mv_cost((int)(short)(m.v0 & 0xFFFF), m.v0 >> 16, a.mvpx, a.mvpy, lam);
        if (m.dir & 2) c += (unsigned)mv_cost((int)(short)(m.v1 & 0xFFFF), m.v1 >> 16, b.mvpx, b.mvpy, lam);
        if ((m.dir & 3) == 3) c -= c >> KS_BI_BIAS_SHIFT;
        return c;
    };
    unsigned best[2]; KsMot bm[2];
    {
        const unsigned cP = price(qP + hP, mP), cMe = price(qO + hH, mO), cNb = price(qH + hO, mH);      // 2NxN: the half = my quarter + its horizontal neighbour
        const unsigned c0 = qx ? cNb : cMe, c1 = qx ? cMe : cNb; const KsMot m0 = qx ? mH : mO, m1 = qx ? mO : mH;
        best[0] = cP; bm[0] = mP;
        if (c0 < best[0]) { best[0] = c0; bm[0] = m0; }
        if (c1 < best[0]) { best[0] = c1; bm[0] = m1; }
    }
    {
        const unsigned cP = price(qP + wP, mP), cMe = price(qO + wV, mO), cNb = price(qV + wO, mV);      // Nx2N: my quarter + its vertical neighbour
        const unsigned c0 = qy ? cNb : cMe, c1 = qy ? cMe : cNb; const KsMot m0 = qy ? mV : mO, m1 = qy ? mO : mV;
        best[1] = cP; bm[1] = mP;
        if (c0 < best[1]) { best[1] = c0; bm[1] = m0; }
        if (c1 < best[1]) { best[1] = c1; bm[1] = m1; }
    }
    const unsigned ob0 = (unsigned)__shfl_xor((int)best[0], 2 * gq, 64), ob1 = (unsigned)__shfl_xor((int)best[1], gq, 64);
    KsMot om[2];
    om[0].v0 = __shfl_xor(bm[0].v0, 2 * gq, 64); om[0].v1 = __shfl_xor(bm[0].v1, 2 * gq, 64); om[0].dir = __shfl_xor(bm[0].dir, 2 * gq, 64);
    om[1].v0 = __shfl_xor(bm[1].v0, gq, 64); om[1].v1 = __shfl_xor(bm[1].v1, gq, 64); om[1].dir = __shfl_xor(bm[1].dir, gq, 64);
    const int G = 1 << (2 * (3 - l));
    if ((lane & (G - 1)) == 0) {
        KsRect o;
        if (valid) {
            const unsigned long long pen = (unsigned long long)((lam * KS_PART_BITS) >> 4);
            for (int k = 0; k < 2; ++k) {
                const unsigned long long t = (unsigned long long)best[k] + (k ? ob1 : ob0) + pen;
                o.cost[k] = (!mot_same(bm[k], mP) || !mot_same(om[k], mP)) ? (t > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)t) : KS_COST_INVALID;      // considered only if a half's motion is not the CU's
                const KsMot h2[2] = {bm[k], om[k]};
                for (int hf = 0; hf < 2; ++hf) {
                    o.mv[k][hf][0] = (short)(h2[hf].v0 & 0xFFFF); o.mv[k][hf][1] = (short)(h2[hf].v0 >> 16);
                    o.mv1[k][hf][0] = (short)(h2[hf].v1 & 0xFFFF); o.mv1[k][hf][1] = (short)(h2[hf].v1 >> 16); o.dir[k][hf] = (unsigned char)h2[hf].dir;
                }
            }
        } else {
            o.cost[0] = o.cost[1] = KS_COST_INVALID;
            for (int k = 0; k < 2; ++k) for (int hf = 0; hf < 2; ++hf) { o.mv[k][hf][0] = o.mv[k][hf][1] = o.mv1[k][hf][0] = o.mv1[k][hf][1] = 0; o.dir[k][hf] = 1; }
        }
        rect[(long)ctu * 21 + pidx] = o;
    }
}

extern "C" int ks265_cu_decide_part_b(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *pu0, const ks265_pu *pu1, const ks265_pu_b *pub, const uint32_t *ibest, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref0.y || !ref1.y || !pu0 || !pu1 || !pub || !cu8) return KS265_POINTER;
    if (!f->rect) return KS265_NOTSUPPORTED;                                      // the frame object was created without cfg.part
    const int nctu = f->g.ctu_cols * f->g.ctu_rows;
    if (f->mrefb) hipLaunchKernelGGL(rect_eval_b_kernel<true>, dim3(nctu), dim3(192), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, (KsRect *)f->rect, ks_mrefb(f));
    else hipLaunchKernelGGL(rect_eval_b_kernel<false>, dim3(nctu), dim3(192), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu0, pu1, pub, (KsRect *)f->rect, KsMrefB{});
    hipLaunchKernelGGL(cu_decide_kernel<ks265_pu_b>, dim3(nctu), dim3(64), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, pub, cu8, ibest, (const KsRect *)f->rect);
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_cu_decide_part(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *pu, const uint32_t *ibest, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y || !pu || !cu8) return KS265_POINTER;
    if (!f->rect) return KS265_NOTSUPPORTED;                                      // the frame object was created without cfg.part
    const int nctu = f->g.ctu_cols * f->g.ctu_rows;
    hipLaunchKernelGGL(rect_eval_kernel, dim3(nctu), dim3(192), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref.y, pu, (KsRect *)f->rect);
    hipLaunchKernelGGL(cu_decide_kernel<ks265_pu>, dim3(nctu), dim3(64), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, pu, cu8, ibest, (const KsRect *)f->rect);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ Stage C2: merge pass (cfg.merge)
// The reference decides merge / skip per CU against the candidates of already coded neighbours (GetMergeCandsFor*, skipFastDecision; closed code).
// A frame-parallel decision has no coded neighbours: this pass works on the motion field the CU decision left behind.  Every CU looks at its five
// spatial merge neighbours (A1 B1 B0 A0 B2 of H.265 8.5.3.2.3: inside the picture, earlier in z-scan order, inter) and at the zero vector, takes each
// one's motion as its own and keeps the cheapest if it beats what the search found: SATD of the prediction + lambda x (position in the list + 1)
// against the CU's search cost + 2 lambda.  All CUs decide on the same input field (cu_in -> cu_out): the result does not depend on any order.
// One work-group per CTU; lane = 8x8 tile in z-order (as in bi_decide), wave w evaluates candidates w and w + 4; a tile's SATD against the (averaged)
// plane tiles, CU sums by DPP at the CU's own level, the winner per CU through one LDS minimum.
struct MergeMotion { int dir, mvx, mvy, mv1x, mv1y; bool ok; };
__device__ __forceinline__ int z_of_8(int x, int y)
{
    const int bx = (x >> 3) & 7, by = (y >> 3) & 7;
    return (bx & 1) | ((by & 1) << 1) | ((bx & 2) << 1) | ((by & 2) << 2) | ((bx & 4) << 2) | ((by & 4) << 3);
}
template <bool MR>
__device__ __forceinline__ MergeMotion merge_cand(const KsGeom &g, const ks265_cu8 *cu_in, int x, int y, int n, int k, bool bi_zero)
{
    MergeMotion m; m.dir = bi_zero ? 3 : 1; m.mvx = m.mvy = m.mv1x = m.mv1y = 0; m.ok = true;
    if (k == 5) return m;
    const int nx = k == 1 ? x + n - 1 : k == 2 ? x + n : x - 1, ny = k == 0 ? y + n - 1 : k == 3 ? y + n : y - 1;       // A1 B1 B0 A0 B2
    m.ok = false;
    if (nx < 0 || ny < 0 || nx >= g.W || ny >= g.H) return m;
    const int ctb = (y >> 6) * g.ctu_cols + (x >> 6), nctb = (ny >> 6) * g.ctu_cols + (nx >> 6);
    if (nctb > ctb || (nctb == ctb && z_of_8(nx, ny) >= z_of_8(x, y))) return m;
    const ks265_cu8 c = cu_in[(long)(ny >> 3) * g.w8 + (nx >> 3)];
    if (c.pred_mode != 0 || (c.log2_cu & 15) < 3) return m;
    m.dir = MR ? (int)c.inter_dir : (c.inter_dir & 3); m.mvx = c.mvx; m.mvy = c.mvy; m.mv1x = c.mv1x; m.mv1y = c.mv1y; m.ok = true;      // (MR: the neighbour's pictures come with its motion)
    // a neighbour's vector may come from a CTU with another window offset: taken over here it must keep this CU's block inside the planes' margin
    if ((m.dir & 1) && (x + (m.mvx >> 2) < -70 || x + (m.mvx >> 2) + n > g.W + 70 || y + (m.mvy >> 2) < -70 || y + (m.mvy >> 2) + n > g.H + 70)) m.ok = false;
    if ((m.dir & 2) && (x + (m.mv1x >> 2) < -70 || x + (m.mv1x >> 2) + n > g.W + 70 || y + (m.mv1y >> 2) < -70 || y + (m.mv1y >> 2) + n > g.H + 70)) m.ok = false;
    return m;
}
template <bool MR>
__global__ __launch_bounds__(256) void merge_pass_kernel(KsGeom g, int lam, const uint8_t *src, const uint8_t *ref0, const uint8_t *ref1, const ks265_pu *pu,
                                                         const ks265_pu_b *pub, const ks265_cu8 *cu_in, ks265_cu8 *cu_out, const KsMrefB mr, int p_slice)
{
    __shared__ unsigned long long jbest[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const int tx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ty = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;
    const bool inside = x0 < g.W && y0 < g.H, is_b = pub != nullptr, bi_zero = is_b && !p_slice;   // (p_slice: the two-list records of a multi-reference P picture - the zero candidate has one list)
    ks265_cu8 c;
    c.mvx = c.mvy = c.mv1x = c.mv1y = 0; c.log2_cu = 0; c.cbf = 0; c.pred_mode = 1; c.inter_dir = 0;
    if (inside) c = cu_in[(long)(y0 >> 3) * g.w8 + (x0 >> 3)];
    const int log2 = (c.log2_cu & 15) >= 3 ? (c.log2_cu & 15) : 3, n = 1 << log2, cux = x0 & ~(n - 1), cuy = y0 & ~(n - 1), level = 6 - log2;
    const int leader = lane & ~((1 << (2 * (log2 - 3))) - 1);
    const long rb = (long)ctu * 85 + ks_level_base(level) + ((cuy & 63) >> log2) * (1 << level) + ((cux & 63) >> log2);
    const unsigned cur = is_b ? pub[rb].cost : pu[rb].cost;
    const bool valid = inside && c.pred_mode == 0 && (c.log2_cu & 15) >= 3 && !(c.log2_cu >> 4) && cur != KS_COST_INVALID;      // (a CU in two partitions keeps its partitions' vectors)
    if (tid < 64) jbest[tid] = ~0ull;
    // which candidates exist for this tile's CU (every tile of a CU computes the same mask), and which of them repeat the motion of an earlier one: a repeat costs what the
    // first one costs plus a longer index - it never wins (strict '<' in candidate order), so it is not evaluated.  The distinct ones are dealt to the waves in order:
    // with four or fewer of them (the usual case: neighbours share vectors) no wave does a second evaluation
    unsigned mask = 0, distinct = 0;
    {
        MergeMotion mm[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            mm[k] = merge_cand<MR>(g, cu_in, cux, cuy, n, k, bi_zero);
            const bool ok = valid && mm[k].ok;
            mask |= (ok ? 1u : 0u) << k;
            bool rep = false;
#pragma unroll
            for (int j = 0; j < k; ++j) {
                const bool same = mm[j].dir == mm[k].dir && (!(mm[k].dir & 1) || (mm[j].mvx == mm[k].mvx && mm[j].mvy == mm[k].mvy)) &&
                                  (!(mm[k].dir & 2) || (mm[j].mv1x == mm[k].mv1x && mm[j].mv1y == mm[k].mv1y));
                rep |= ((mask >> j) & 1u) && same;
            }
            distinct |= (ok && !rep ? 1u : 0u) << k;
        }
    }
    unsigned f[16];
    {
        const uint8_t *frow = ks_org_y(g, src) + (long)(valid ? y0 : cy * 64) * g.sy + (valid ? x0 : cx * 64);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const uint2 v = *(const uint2 *)(frow + (long)r * g.sy); f[2 * r] = v.x; f[2 * r + 1] = v.y; }
    }
    __syncthreads();
    const long base = (long)(valid ? y0 : 0) * g.sy + (valid ? x0 : 0) + g.org_y;
#pragma unroll 1
    for (int it = wave; it < 6; it += 4) {
        if (!__any(__popc(distinct) > it)) break;                      // nobody in this wave has that many distinct candidates
        int k = 0;                                                     // the it-th distinct candidate of this lane's CU
        { unsigned d = distinct; for (int q = 0; q < it; ++q) d &= d - 1u; k = d ? __ffs((int)d) - 1 : 0; }
        const bool on = __popc(distinct) > it;
        const MergeMotion m = merge_cand<MR>(g, cu_in, cux, cuy, n, k, bi_zero);
        const int ax = on ? m.mvx : 0, ay = on ? m.mvy : 0, bx = on ? m.mv1x : 0, by = on ? m.mv1y : 0, dir = on ? m.dir : 1;
        unsigned sd = 0;
        if (__any(on)) {
            unsigned PA[16], PB[16];                                   // the candidate's prediction tiles, interpolated from the reference pictures (interp_dev.h)
            const uint8_t *r0 = MR ? ks_pick(mr.y0, (dir >> 4) & 3) : ref0, *r1 = MR ? ks_pick(mr.y1, (dir >> 6) & 3) : ref1;
            if (dir & 1) luma_pred_tile8(r0 + base, g.sy, ax, ay, PA);
            if (dir & 2) luma_pred_tile8(r1 + base, g.sy, bx, by, PB);
#pragma unroll
            for (int i = 0; i < 16; ++i) { if (!(dir & 1)) PA[i] = PB[i]; if (!(dir & 2)) PB[i] = PA[i]; }
            sd = satd8x8_avg(f, PA, PB);
        }
        if (!on) sd = 0;
        // CU sums at all four levels, each lane picks its CU's
        const unsigned s3 = sd;
        unsigned s2 = s3 + (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)s3); s2 += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)s2);
        unsigned s1 = s2 + (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)s2); s1 += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)s1);
        unsigned s0 = s1 + (unsigned)__builtin_amdgcn_ds_swizzle((int)s1, 0x1F | (16 << 10)); s0 += (unsigned)__shfl_xor((int)s0, 32, 64);
        const unsigned sum = level == 3 ? s3 : level == 2 ? s2 : level == 1 ? s1 : s0;
        if (on && lane == leader) {
            const int pos = __popc(mask & ((1u << k) - 1u));
            const unsigned long long j = (unsigned long long)sum + (unsigned long long)((lam * 16 * (pos + 1)) >> 4);
            atomicMin(&jbest[leader], (j << 8) | (unsigned long long)k);
        }
    }
    __syncthreads();
    if (tid < 64 && inside) {
        ks265_cu8 o = c;
        if (valid) {
            const unsigned long long jb = jbest[leader], jc = (unsigned long long)cur + (unsigned long long)((lam * 32) >> 4);
            if ((jb >> 8) < jc) {
                const MergeMotion m = merge_cand<MR>(g, cu_in, cux, cuy, n, (int)(jb & 255ull), bi_zero);
                o.mvx = (int16_t)m.mvx; o.mvy = (int16_t)m.mvy; o.mv1x = (int16_t)m.mv1x; o.mv1y = (int16_t)m.mv1y; o.inter_dir = (uint8_t)(MR ? m.dir : (m.dir & 3));
            }
        }
        cu_out[(long)(y0 >> 3) * g.w8 + (x0 >> 3)] = o;
    }
}

extern "C" int ks265_merge_pass(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *pu, const ks265_pu_b *pub,
                                const ks265_cu8 *cu_in, ks265_cu8 *cu_out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref0.y || !cu_in || !cu_out || cu_in == cu_out || (!pu && !pub) || (pub && !ref1.y)) return KS265_POINTER;
    if (f->mrefb && pub) hipLaunchKernelGGL(merge_pass_kernel<true>, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(256), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu, pub, cu_in, cu_out, ks_mrefb(f), f->mr_pslice ? 1 : 0);
    else hipLaunchKernelGGL(merge_pass_kernel<false>, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(256), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, ref0.y, ref1.y, pu, pub, cu_in, cu_out, KsMrefB{}, 0);
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_cu_flat_intra(ks265_frame *f, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!cu8) return KS265_POINTER;
    hipLaunchKernelGGL(cu_flat_intra_kernel, dim3((f->g.w8 * f->g.h8 + 255) / 256), dim3(256), 0, f->ctx->stream, f->g, cu8);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ multi-reference P pictures (-ref / -ref0)
// motionSearchOneRef enc@0x483f40 runs once per reference picture; per PU the picture with the smallest cost + lambda * ref_idx bits
// wins (truncated unary: idx < nref - 1 ? idx + 1 : nref - 1 bits; ties to the nearest picture).  Winner: inter_dir = 1 | idx << 4.
__global__ __launch_bounds__(256) void ref_decide_kernel(long n, int nref, int lam, const ks265_pu *p0, const ks265_pu *p1, const ks265_pu *p2, const ks265_pu *p3,
                                                         ks265_pu_b *pub)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ks265_pu a = p0[i];
    ks265_pu_b o;
    o.mvx = a.mvx; o.mvy = a.mvy; o.mv1x = 0; o.mv1y = 0; o.cost = a.cost; o.inter_dir = 1;
    if (a.cost != KS_COST_INVALID) {
        unsigned best = KS_COST_INVALID;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                  // (unrolled: a run-time choice among the four pointers put the record into scratch memory)
            if (r >= nref) break;
            const ks265_pu q = r == 0 ? a : (r == 1 ? p1[i] : (r == 2 ? p2[i] : p3[i]));
            const int bits = nref == 1 ? 0 : (r < nref - 1 ? r + 1 : nref - 1);
            const unsigned c = q.cost + (unsigned)((lam * bits) >> 4);
            if (c < best) { best = c; o.mvx = q.mvx; o.mvy = q.mvy; o.cost = c; o.inter_dir = 1u | ((unsigned)r << 4); }
        }
    }
    pub[i] = o;
}

// one list of a multi-reference B picture: per PU the picture with the smallest cost + lambda x ref_idx bits (the same rule the oracle pipeline states); out may be p0 (in place)
__global__ __launch_bounds__(256) void ref_pick_kernel(long n, int nref, int lam, const ks265_pu *p0, const ks265_pu *p1, const ks265_pu *p2, const ks265_pu *p3, ks265_pu *out, uint8_t *idx)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const ks265_pu a = p0[i];
    ks265_pu o = a; int bi = 0;
    if (a.cost != KS_COST_INVALID) {
        unsigned best = KS_COST_INVALID;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nref) break;
            const ks265_pu q = r == 0 ? a : (r == 1 ? p1[i] : (r == 2 ? p2[i] : p3[i]));
            const int bits = nref <= 1 ? 0 : (r < nref - 1 ? r + 1 : nref - 1);
            const unsigned long long c = (unsigned long long)q.cost + (unsigned long long)((lam * bits) >> 4);
            if (c < best) { best = (unsigned)c; o = q; o.cost = best; bi = r; }
        }
    }
    out[i] = o; idx[i] = (uint8_t)bi;
}
extern "C" int ks265_ref_pick(ks265_frame *f, int nref, const ks265_pu *const *pu, ks265_pu *out, uint8_t *dev_idx)
{
    KS_FRAME_CHECK(f);
    if (!pu || !out || !dev_idx) return KS265_POINTER;
    if (nref < 1 || nref > 4) return KS265_NOTSUPPORTED;
    for (int r = 0; r < nref; ++r) if (!pu[r]) return KS265_POINTER;
    const long n = (long)f->g.ctu_cols * f->g.ctu_rows * 85;
    hipLaunchKernelGGL(ref_pick_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, f->ctx->stream, n, nref, f->cfg.lambda_q4, pu[0], nref > 1 ? pu[1] : pu[0],
                       nref > 2 ? pu[2] : pu[0], nref > 3 ? pu[3] : pu[0], out, dev_idx);
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_ref_decide(ks265_frame *f, int nref, const ks265_pu *const *pu, ks265_pu_b *pub)
{
    KS_FRAME_CHECK(f);
    if (!pu || !pub) return KS265_POINTER;
    if (nref < 1 || nref > 4) return KS265_NOTSUPPORTED;
    for (int r = 0; r < nref; ++r) if (!pu[r]) return KS265_POINTER;
    const long n = (long)f->g.ctu_cols * f->g.ctu_rows * 85;
    hipLaunchKernelGGL(ref_decide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, f->ctx->stream, n, nref, f->cfg.lambda_q4, pu[0], nref > 1 ? pu[1] : pu[0],
                       nref > 2 ? pu[2] : pu[0], nref > 3 ? pu[3] : pu[0], pub);
    return ks265_check_launch(f->ctx);
}
