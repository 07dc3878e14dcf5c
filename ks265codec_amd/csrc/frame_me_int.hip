// frame_me_int.hip — Stage A, integer-pel motion search (motionSearchOneRef enc@0x483f40) for every PU of every CTU.
//
// The search PATTERNS are the reference's own control flow — interMeDia enc@0x48fbe0, interMeHex enc@0x48fde0, interMeUMH
// enc@0x4907b0 — as restated from the disassembly in oracle/ks265_me_ref.c and pinned there against traces of the reference binary
// (tests/golden/me_search.npz).  This file runs the same state machines for many PUs at once:
//
//   * one work-group (4 waves) per CTU; the 64x64 source block and the +-66 reference window are staged in LDS once (every window
//     byte leaves HBM once per CTU); row pitches 51 / 17 dwords (odd);
//   * the 64x64 PU is searched by the whole work-group; then every wave takes one 32x32 quadrant and runs its 32x32, 16x16 and 8x8
//     levels ON ITS OWN, without work-group barriers (a PU's predictor is its nearest valid ancestor's vector, which lies in the
//     same quadrant; a wave's LDS operations complete in order): waves drift apart and hide each other's latencies;
//   * OWNERS: the first lanes of a group hold the search state of one PU each (phase, best vector, cost, direction, iteration
//     count) in registers.  Once per round an owner publishes a descriptor of its phase and takes one JOB slot per candidate —
//     4 for a diamond step, 3 / 6 for hexagon steps, 8 for the square refinement, up to 64 for the sparse cross and 128 for the
//     big-hexagon rings (fixed-centre stages: all candidates are independent);
//   * all lanes of the group then evaluate the jobs: ONE LANE = ONE 8x8 TILE OF ONE CANDIDATE (candidate offset from a small LDS
//     table, v_sad_u8 on packed dwords, v_alignbyte for the unaligned window rows); the 1 / 4 / 16 / 64 tiles of a candidate are
//     adjacent lanes (DPP / swizzle sums); a diamond step of any level fills the group exactly;
//   * the job's first lane adds the mv rate and folds (cost, key, x, y) into a 64-bit LDS slot of its PU with ds_min_u64.  Keys
//     reproduce the reference's tie rules: the direction codes of the packed (cost << 4) + code / (cost << 3) + code forms, or the
//     scan position for the "first strictly better" stages; key 0 = the incumbent;
//   * owners read the winner and advance their state machine.  Rounds repeat until every owner of the group is done;
//   * the FIRST round is speculative: the start point, its four diamond neighbours (first step of interMeDia / interMeUMH), the six
//     hexagon points (first step of interMeHex) and the four diagonals (interMeHex's closing square) are evaluated together.  A PU
//     whose search converges at its start point - most of them - is finished after that single round.
#include "frame_common.h"

using namespace ks265;

#define WIN_LOAD_XL 80            // global loads start at picture x = ctu_x*64 - 80 (16-byte aligned), 14 x 16 bytes per row
#define WIN_XL 68                 // LDS window column 0 is picture x = ctu_x*64 - 68
#define WIN_YT 66                 // window row 0 is picture y = ctu_y*64 - 66
#define WIN_ROWS 196              // y in [-66, 130)
#ifndef KS_WIN_STRIDE
#define KS_WIN_STRIDE 204
#endif
#define WIN_STRIDE KS_WIN_STRIDE  // 204 = 51 dwords (odd): x in [-68, 136); KS_WIN_STRIDE: experiments with the row pitch (bank conflicts, tools/r6_me_stride.sh)
#define FENC_STRIDE 68            // 17 dwords (odd)
#define ME_WLIM 66                // candidates further than this from the PU position are not staged: skipped (oracle chk = 2)

enum { PH_INIT, PH_START, PH_DIA, PH_H6, PH_HSTEP, PH_SQUARE, PH_U1, PH_UCROSS, PH_UHEX6, PH_UBIG, PH_UFINAL, PH_UHW0, PH_UHW, PH_UDW, PH_DONE };

// Candidate offsets (dx, dy), key and result slot: dx & 0xFF | (dy & 0xFF) << 8 | key << 16 | slot << 24.  key 0 = "key is k + keyadd".
// The pattern tables are rodata of the reference binary (values read from the file).
#define CT(dx, dy, key, slot) ((unsigned)((dx) & 0xFF) | ((unsigned)((dy) & 0xFF) << 8) | ((unsigned)(key) << 16) | ((unsigned)(slot) << 24))
#define TB_SQUARE 0               // the 4-neighbour step codes 1 up, 3 down, 4 left, 12 right, then interMeHex's diagonals 5, 7, 13, 15
#define TB_HEX2 8                 // hex2 enc@0x4e52e0 (8 entries: the hexagon repeated so that dir + 2 never wraps)
#define TB_HEXAGON 16             // Hexagon enc@0x4e5340
#define TB_BIG 24                 // Big_Hexagon_X / _Y enc@0x4e5320 / 0x4e5300
#define TB_CROSS 40
#define TB_START 44               // speculative first round: start, diamond (slot 1), diagonals (slot 2), hex2[1..6] with interMeHex's codes (slot 3)
#define TB_SIZE 59
__device__ const unsigned kCandTab[TB_SIZE] = {
    CT(0, -1, 1, 0), CT(0, 1, 3, 0), CT(-1, 0, 4, 0), CT(1, 0, 12, 0), CT(-1, -1, 5, 0), CT(-1, 1, 7, 0), CT(1, -1, 13, 0), CT(1, 1, 15, 0),
    CT(-1, -2, 0, 0), CT(-2, 0, 0, 0), CT(-1, 2, 0, 0), CT(1, 2, 0, 0), CT(2, 0, 0, 0), CT(1, -2, 0, 0), CT(-1, -2, 0, 0), CT(-2, 0, 0, 0),
    CT(-2, 0, 0, 0), CT(2, 0, 0, 0), CT(-1, -2, 0, 0), CT(1, 2, 0, 0), CT(-1, 2, 0, 0), CT(1, -2, 0, 0), 0, 0,
    CT(-4, 0, 0, 0), CT(4, 0, 0, 0), CT(0, -4, 0, 0), CT(0, 4, 0, 0), CT(-4, -1, 0, 0), CT(4, 1, 0, 0), CT(-4, 1, 0, 0), CT(4, -1, 0, 0),
    CT(-4, -2, 0, 0), CT(4, 2, 0, 0), CT(-4, 2, 0, 0), CT(4, -2, 0, 0), CT(-2, -3, 0, 0), CT(2, 3, 0, 0), CT(-2, 3, 0, 0), CT(2, -3, 0, 0),
    CT(1, 0, 0, 0), CT(-1, 0, 0, 0), CT(0, 1, 0, 0), CT(0, -1, 0, 0),
    CT(0, 0, 1, 0), CT(0, -1, 1, 1), CT(0, 1, 3, 1), CT(-1, 0, 4, 1), CT(1, 0, 12, 1), CT(-1, -1, 5, 2), CT(-1, 1, 7, 2), CT(1, -1, 13, 2), CT(1, 1, 15, 2),
    CT(-2, 0, 2, 3), CT(-1, 2, 3, 3), CT(1, 2, 4, 3), CT(2, 0, 5, 3), CT(1, -2, 6, 3), CT(-1, -2, 7, 3),
};
__device__ __forceinline__ int mod6m1(int i) { return (int)((0x05432105u >> (4 * i)) & 15u); }      // mod6m1 enc@0x4e52c0

__device__ __forceinline__ int se_bits_dev(int v)
{
    const unsigned u = (unsigned)(v <= 0 ? -2 * v : 2 * v - 1) + 1u;
    return 2 * (31 - __clz(u)) + 1;
}
// lambda x se-Golomb bits of the quarter-pel difference to the predictor, per component (createMvdCostTable enc@0x48b850)
__device__ __forceinline__ unsigned mv_rate(int lam, int x, int y, int pmx, int pmy)
{
    return (unsigned)((lam * se_bits_dev((x - pmx) << 2)) >> 4) + (unsigned)((lam * se_bits_dev((y - pmy) << 2)) >> 4);
}

// per-wave engine state (the 64x64 level uses wave 0's copy for the whole work-group)
struct GroupLds {
    int4 desc[16];                         // per PU of the group, published by its owner every round: centre x | y << 16, candidate-table parameters (DP_*), number of candidates (0: done)
    unsigned long long best[16][4];        // per PU and slot: (cost << 8 | key) << 32 | (x + 128) << 8 | (y + 128)
    int pred[16];                          // per PU: predictor, x | y << 16
    int fld[16];                           // per PU: pre-search candidate of PH_INIT, x | y << 16
    int active;
};
struct MeLds {
    uint8_t win[WIN_ROWS * WIN_STRIDE];
    uint8_t fenc[64 * FENC_STRIDE];
    unsigned ctab[64];
    GroupLds grp[4];
    int pmv[85];                           // integer vectors of the finished PUs (predictors of the finer levels)
    int fld16[16];                         // pre-search vector of the CTU's sixteen 16x16 blocks, x | y << 16 (0x80000000: none)
    unsigned best16[16];
};

struct MeLim { int ox, oy, lox, hix, loy, hiy; };        // the CTU's window offset and the vectors its PUs may take

struct Owner {
    int ph, mx, my, pmx, pmy, merange, it, dir, iflags;
    unsigned cost, cost0;
};

// descriptor parameters: candidate k of a phase is ctab[tb + (k & km)] scaled by m0 + ms * (k >> ksh); its key is the table's, or k + keyadd
#define DP(tb, km, ksh, m0, ms, keyadd, ranged) ((tb) | ((km) << 6) | ((ksh) << 13) | ((m0) << 16) | ((ms) << 20) | ((keyadd) << 24) | ((ranged) << 28))
#define DP_INIT (1 << 29)
#define DP_INIT_ZERO (1 << 30)       // PH_INIT: candidate 1 (the zero vector) takes part (root PU with a non-zero temporal predictor)
#define DP_INIT_FIELD (1 << 31)      // PH_INIT: candidate 2 (the pre-search vector, GroupLds::fld) takes part

// number of candidates of the owner's phase and its descriptor parameters
__device__ __forceinline__ int phase_desc(const Owner &o, int nstart, int &dp)
{
    switch (o.ph) {
    case PH_INIT: dp = DP_INIT | o.iflags; return 3;                                      // predictor (key 1), zero vector (key 2), pre-search vector (key 3)
    case PH_START: dp = DP(TB_START, 15, 0, 1, 0, 0, 0); return nstart;
    case PH_DIA: case PH_U1: case PH_UFINAL: case PH_UDW: dp = DP(TB_SQUARE, 7, 0, 1, 0, 0, 0); return 4;
    case PH_SQUARE: dp = DP(TB_SQUARE, 7, 0, 1, 0, 0, 0); return 8;
    case PH_H6: dp = DP(TB_HEX2 + 1, 7, 0, 1, 0, 2, 0); return 6;                        // hex2[1..6], codes 2..7
    case PH_HSTEP: dp = DP(TB_HEX2, 7, 0, 1, 0, 1, 0) + o.dir; return 3;                 // hex2[dir .. dir + 2], codes 1..3
    case PH_UCROSS: dp = DP(TB_CROSS, 3, 2, 4, 8, 1, 1); return 4 * (o.merange >> 2);    // i = 4, 12, .. <= 2 * merange - 4; +i, -i along x, then y
    case PH_UHEX6: dp = DP(TB_HEXAGON, 7, 0, 1, 0, 1, 1); return 6;
    case PH_UBIG: dp = DP(TB_BIG, 15, 4, 1, 1, 1, 1); return 16 * (o.merange >> 3);      // ring r = 1 + (k >> 4)
    case PH_UHW0: dp = DP(TB_HEX2, 7, 0, 1, 0, 1, 1); return 6;
    case PH_UHW: dp = DP(TB_HEX2, 7, 0, 1, 0, 1, 1) + o.dir; return 3;                   // dir is kept reduced mod 6
    default: dp = 0; return 0;
    }
}

#ifdef KS_EXP_ME_TRACE
// experiment build (-DKS_EXP_ME_TRACE, scratch/me_trace.py): per work-group the 100 MHz wall clock at its start, after the prologue, after the 64x64 level and when each of
// its waves ends, + the compute unit it ran on - where the kernel's time goes between "mean wave lifetime x rounds" and what rocprof reports
__device__ unsigned long long ks_me_trace[8 * 4096];
extern "C" void ks265_me_trace(unsigned long long *out, int n) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ks_me_trace), sizeof(unsigned long long) * (size_t)n); }
#endif
#ifdef KS_EXP_ME_CLOCK
__device__ unsigned long long ks_me_dbg[32];      // per level: [0] rounds, [1] emit, [2] eval, [3] advance cycles, [4] jobs, [5] level cycles (wave 0 lane 0)
extern "C" void ks265_me_dbg(unsigned long long *out, int reset) { if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ks_me_dbg), z, sizeof z); } else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ks_me_dbg), sizeof(unsigned long long) * 32); }
#define ME_NOW() ((long long)__builtin_readcyclecounter())
#endif

// One LEVEL of one GROUP of PUs.  WG = true: the group is the CTU's single 64x64 PU and all 256 lanes of the work-group evaluate its
// candidates (work-group barriers between the steps of a round).  WG = false: the group is the part of a level that lies in one 32x32
// quadrant (1 / 4 / 16 PUs) and belongs to ONE WAVE: owners = its first lanes, evaluation = its 64 lanes, no barrier at all.
template <bool WG, int LEVEL>
__device__ __forceinline__ int me_group(const KsGeom &g, int cx, int cy, int range, const MeLim &lm, int lam, int method, int hex_thr, MeLds &L, GroupLds &Q,
                                         int l2n_ /* log2 PUs per side of the group */, int qx0, int qy0 /* PU-grid origin of the group */,
                                         const ks265_pu *prev_ctu, ks265_pu *out_ctu, const short2 *field, int nb0x, int nb0y, int t /* lane index inside the group's lanes */,
                                         const unsigned (&fe)[16] /* the lane's source tile: 8 rows x 2 dwords (WG: tile t & 63 of the CTU in raster order; else tile t & 15 of the quadrant in z-order) */)
{
    constexpr int NT = WG ? 256 : 64;
    constexpr int level = LEVEL, l2n = WG ? 0 : LEVEL - 1; (void)l2n_;        // the engine is compiled per level (round 4: - 2.5 % on the kernel; the tile geometry and the DPP sums are constants)
    const int S = 64 >> level, npu = 1 << (2 * l2n), l2t = 6 - 2 * level;              // tiles per PU = 1 << l2t (64, 16, 4, 1)
    const int tpr = 8 >> level;                                                          // tiles per PU row
    auto sync = [&]() { if (WG) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    // speculative first round: start + diamond for interMeDia / an always-UMH search, + diagonals + hexagon when interMeHex can run
    const int nstart = (method == 1 || (method == 2 && hex_thr > 0)) ? 15 : 5;
    Owner o;
    o.ph = PH_DONE; o.mx = o.my = o.pmx = o.pmy = 0; o.merange = 0; o.it = 0; o.dir = 0; o.cost = 0; o.cost0 = 0; o.iflags = 0;
    bool inside = false;
    const int px = qx0 + (t & ((1 << l2n) - 1)), py = qy0 + ((t >> l2n) & ((1 << l2n) - 1));
    const bool is_owner = t < npu;
    // evaluation geometry of this lane, fixed for the level: its tile, the PU the tile belongs to, which quarter of the PU's candidates it takes
    int tbx, tby, my_pu, grp; bool first_tile;
    if (WG) { tbx = (t & 7) * 8; tby = ((t >> 3) & 7) * 8; my_pu = 0; grp = t >> 6; first_tile = (t & 63) == 0; }
    else {
        const int z = t & 15, tx = (z & 1) | ((z >> 1) & 2), ty = ((z >> 1) & 1) | ((z >> 2) & 2), shp = 3 - level;
        tbx = qx0 * S + tx * 8; tby = qy0 * S + ty * 8; grp = t >> 4;
        my_pu = ((ty >> shp) << l2n) | (tx >> shp);
        first_tile = level == 3 ? true : level == 2 ? (z & 3) == 0 : z == 0;
    }
    if (is_owner) {
        inside = ks_pu_inside(g, cx, cy, level, px, py);
        if (inside) {
            bool root = true;
            for (int a = level - 1; a >= 0; --a) {
                const int ax = px >> (level - a), ay = py >> (level - a);
                if (ks_pu_inside(g, cx, cy, a, ax, ay)) {
                    const int v = L.pmv[ks_pu_index(a, ax, ay)];
                    o.pmx = (int)(short)(v & 0xFFFF); o.pmy = v >> 16; root = false;
                    break;
                }
            }
            if (root) {                                                                // no predictor: the legal vector nearest to zero
                o.pmx = clip3(lm.lox, lm.hix, 0); o.pmy = clip3(lm.loy, lm.hiy, 0);
                if (prev_ctu && prev_ctu[0].cost != KS_COST_INVALID) {
                    o.pmx = clip3(lm.lox, lm.hix, ((int)prev_ctu[0].mvx + 2) >> 2);
                    o.pmy = clip3(lm.loy, lm.hiy, ((int)prev_ctu[0].mvy + 2) >> 2);
                }
            }
            o.merange = root ? range : max(range >> 2, 4);
            o.mx = o.pmx; o.my = o.pmy;
            if (root && (o.pmx | o.pmy) && lm.lox <= 0 && lm.hix >= 0 && lm.loy <= 0 && lm.hiy >= 0) o.iflags |= DP_INIT_ZERO;   // a root PU with a predictor also tries the zero vector
            if (field) {                                                               // and every PU the pre-search vector of the 16x16 block under its centre
                const int fw = L.fld16[(((py * S + S / 2) >> 4) << 2) + ((px * S + S / 2) >> 4)];
                const int fx = (int)(short)(fw & 0xFFFF), fy = fw >> 16;
                if (fw != (int)0x80000000 && (fx != o.pmx || fy != o.pmy)) { o.iflags |= (int)DP_INIT_FIELD; Q.fld[t] = fw; }
            }
            o.ph = o.iflags ? PH_INIT : PH_START;
            Q.pred[t] = (o.pmx & 0xFFFF) | (o.pmy << 16);
        } else {                                                                        // PU not (completely) inside the picture: marked, never searched
            ks265_pu e; e.mvx = e.mvy = e.mvpx = e.mvpy = 0; e.cost = KS_COST_INVALID; e.dist = KS_COST_INVALID;
            out_ctu[ks_pu_index(level, px, py)] = e;
        }
    }
    const unsigned t1 = 62u << (2 * (6 - level) - 4), t2 = 50u << (2 * (6 - level) - 4);
#ifdef KS_EXP_ME_CLOCK
    long long acc[5] = {0, 0, 0, 0, 0}; const long long tl0 = ME_NOW();
#endif

    int rounds = 0;
#pragma unroll 1
    for (;; ++rounds) {
#ifdef KS_EXP_ME_CLOCK
        const long long tr0 = ME_NOW();
#endif
        // ---- owners: publish the phase descriptor (centre, candidate-table parameters, number of candidates) and reset the result slots; the candidates
        //      themselves are expanded by the evaluating lanes
        const bool pending = o.ph != PH_DONE;
        bool emitted = false;
        if (WG ? t < 64 : true) {
            if (is_owner) {
                int dp = 0;
                const int cnt = pending ? phase_desc(o, nstart, dp) : 0;
                emitted = cnt > 0;
                Q.desc[t] = make_int4((o.mx & 0xFFFF) | (o.my << 16), dp, cnt, 0);
                if (emitted) {
                    const unsigned long long inc = (o.ph == PH_INIT || o.ph == PH_START) ? ~0ull : ((unsigned long long)(o.cost << 8) << 32);
                    Q.best[t][0] = inc;
                    if (o.ph == PH_START) { Q.best[t][1] = ~0ull; Q.best[t][2] = ~0ull; Q.best[t][3] = ~0ull; }
                }
            }
            if (WG) { const unsigned long long anyp = __ballot(pending); if (t == 0) Q.active = anyp != 0ull; }
        }
        sync();
        bool active;
        if (WG) active = Q.active != 0; else active = __ballot(pending) != 0ull;
        if (!active) break;
#ifdef KS_EXP_ME_CLOCK
        const long long tr1 = ME_NOW();
#endif
        // ---- all lanes: ONE LANE = ONE FIXED 8x8 TILE (its source rows stay in registers, `fe`) x every fourth candidate of the PU the tile belongs to.  Lanes of a PU and a
        //      candidate group are adjacent (z-order tiles): PU sums by DPP.  No job slots, no source reads from LDS (round 5: the LDS pipe was 55 % busy, 40 % of it source rows)
        {
            const int4 dsc = Q.desc[my_pu];
            const int pr = Q.pred[my_pu];
            const int dp = dsc.y, cnt = dsc.z;
            const int cxm = (int)(short)(dsc.x & 0xFFFF), cym = dsc.x >> 16;
            const int keyadd = (dp >> 24) & 15;
#ifdef KS_EXP_ME_CLOCK
            if (t == 0) acc[4] += cnt;
#endif
#pragma unroll 1
            for (int k = grp; k < cnt; k += 4) {
                const unsigned e = L.ctab[(dp & 63) + (k & ((dp >> 6) & 127))];
                const int mult = ((dp >> 16) & 15) + ((dp >> 20) & 15) * (k >> ((dp >> 13) & 7));
                int x = cxm + (int)(signed char)(e & 0xFF) * mult;
                int y = cym + (int)(signed char)((e >> 8) & 0xFF) * mult;
                int key = keyadd ? k + keyadd : (int)((e >> 16) & 0xFF);
                int slot = (int)(e >> 24);
                bool live = true;
                if (dp & DP_INIT) {
                    if (k == 1) { x = 0; y = 0; live = (dp & DP_INIT_ZERO) != 0; }
                    else if (k == 2) { const int fv = Q.fld[my_pu]; x = (int)(short)(fv & 0xFFFF); y = fv >> 16; live = (dp & (int)DP_INIT_FIELD) != 0; }
                    key = k + 1; slot = 0;
                }
                // "ranged" phases test the mv range (interMeUMH's cross / hexagon / rings); every candidate must lie in the staged window
                const bool inr = (dp >> 28) & 1 ? (x >= lm.lox && x <= lm.hix && y >= lm.loy && y <= lm.hiy)
                                                : (abs(x - lm.ox) <= ME_WLIM && abs(y - lm.oy) <= ME_WLIM);
                live = live && inr;
                const int wx = tbx + (live ? x - lm.ox : 0) + WIN_XL, wy = tby + (live ? y - lm.oy : 0) + WIN_YT;   // window coordinates are relative to the offset; a dead item reads (and discards) the centre
                const uint8_t *pw = L.win + wy * WIN_STRIDE + (wx & ~3);
                const unsigned sh = wx & 3;
                unsigned sd = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned w0 = *(const unsigned *)(pw + r * WIN_STRIDE), w1 = *(const unsigned *)(pw + r * WIN_STRIDE + 4), w2 = *(const unsigned *)(pw + r * WIN_STRIDE + 8);
                    sd = sad_u8x4(fe[2 * r], align_bytes(w1, w0, sh), sd);
                    sd = sad_u8x4(fe[2 * r + 1], align_bytes(w2, w1, sh), sd);
                }
                // the tiles of a PU are adjacent lanes: 1 / 4 / 16 (quadrant levels, z-order) / 64 (the work-group's 64x64 PU: one wave per candidate)
                if (level == 2 || level <= 1) { sd += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)sd); sd += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)sd); }
                if (level <= 1) { sd += (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)sd); sd += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)sd); }
                if (level == 0) { sd += (unsigned)__builtin_amdgcn_ds_swizzle((int)sd, 0x1F | (16 << 10)); sd += (unsigned)__shfl_xor((int)sd, 32, 64); }
                if (live && first_tile) {
                    const unsigned cost = sd + mv_rate(lam, x, y, (int)(short)(pr & 0xFFFF), pr >> 16);
                    const unsigned long long v = ((unsigned long long)((cost << 8) | (unsigned)key) << 32) | (unsigned long long)(((unsigned)(x - lm.ox + 128) << 8) | (unsigned)(y - lm.oy + 128));
                    atomicMin(&Q.best[my_pu][slot], v);
                }
            }
        }
        sync();
#ifdef KS_EXP_ME_CLOCK
        const long long tr2 = ME_NOW();
#endif
        // ---- owners advance (the transitions of oracle/ks265_me_ref.c)
        if (emitted) {
            bool first = true;
            do {
                bool improved = false;
                int key = 0, wx = o.mx, wy = o.my;
                unsigned wcost = o.cost;
                auto take = [&](unsigned long long b) { key = (int)((b >> 32) & 255u); wcost = (unsigned)(b >> 40); wx = (int)((b >> 8) & 255u) - 128 + lm.ox; wy = (int)(b & 255u) - 128 + lm.oy; };
                if (first && o.ph != PH_START) {
                    const unsigned long long b = Q.best[t][0];
                    improved = ((b >> 32) & 255u) != 0;
                    if (improved) take(b);
                }
                if (o.ph != PH_HSTEP && o.ph != PH_START) { o.cost = wcost; o.mx = wx; o.my = wy; }   // interMeHex's walk may refuse the move (below)
                switch (o.ph) {
                case PH_INIT: o.ph = PH_START; break;                                   // the better of the two start candidates is the start point
                case PH_START: {                                                        // start cost, then the first step of the PU's pattern
                    const unsigned long long b0 = Q.best[t][0], b1 = Q.best[t][1];
                    o.cost = (unsigned)(b0 >> 40);
                    const unsigned sad0 = o.cost - mv_rate(lam, o.mx, o.my, o.pmx, o.pmy);
                    if (method == 0) {                                                  // interMeDia, first step
                        improved = (unsigned)(b1 >> 40) < o.cost;
                        if (improved) { take(b1); o.cost = wcost; o.mx = wx; o.my = wy; }
                        o.it = 0; o.ph = PH_DIA;
                        if (!improved || ++o.it >= o.merange) o.ph = PH_DONE;
                    } else if (method == 1 || (hex_thr > 0 && sad0 < ((unsigned)hex_thr << (2 * (6 - level))))) {   // interMeHex
                        const unsigned long long b3 = Q.best[t][3];
                        if ((unsigned)(b3 >> 40) < o.cost) {                            // the hexagon moves: continue as after PH_H6
                            take(b3); o.cost = wcost; o.mx = wx; o.my = wy;
                            o.dir = key - 2; o.it = (o.merange >> 1) - 1; o.ph = o.it > 0 ? PH_HSTEP : PH_SQUARE;
                        } else {                                                        // it does not: the closing square was evaluated as well
                            const unsigned long long b2 = Q.best[t][2], bs = b1 < b2 ? b1 : b2;
                            if ((unsigned)(bs >> 40) < o.cost) { take(bs); o.cost = wcost; o.mx = wx; o.my = wy; }
                            o.ph = PH_DONE;
                        }
                    } else {                                                            // interMeUMH, step 1
                        o.cost0 = o.cost;
                        if ((unsigned)(b1 >> 40) < o.cost) { take(b1); o.cost = wcost; o.mx = wx; o.my = wy; }
                        if (t1 > o.cost0) o.ph = PH_DONE;
                        else o.ph = o.cost > t2 ? PH_UCROSS : PH_UHEX6;
                    }
                    break;
                }
                case PH_DIA:                                                            // interMeDia: no range test, merange steps
                    if (!improved || ++o.it >= o.merange) o.ph = PH_DONE;
                    break;
                case PH_H6:
                    if (!improved) o.ph = PH_SQUARE;
                    else { o.dir = key - 2; o.it = (o.merange >> 1) - 1; o.ph = o.it > 0 ? PH_HSTEP : PH_SQUARE; }
                    break;
                case PH_HSTEP:                                                          // enc@0x490350: a move that leaves the mv range is undone and ends the walk
                    if (!improved || wx < lm.lox || wx > lm.hix || wy < lm.loy || wy > lm.hiy) o.ph = PH_SQUARE;
                    else { o.cost = wcost; o.mx = wx; o.my = wy; o.dir = mod6m1(o.dir + key - 1); --o.it; o.ph = o.it > 0 ? PH_HSTEP : PH_SQUARE; }
                    break;
                case PH_SQUARE: case PH_UFINAL: o.ph = PH_DONE; break;
                case PH_U1:
                    if (t1 > o.cost0) o.ph = PH_DONE;
                    else o.ph = o.cost > t2 ? PH_UCROSS : PH_UHEX6;
                    break;
                case PH_UCROSS: o.ph = PH_UHEX6; break;
                case PH_UHEX6: case PH_UBIG:
                    if (o.ph == PH_UHEX6 && o.merange > 7) o.ph = PH_UBIG;
                    else o.ph = t1 >= o.cost ? PH_UFINAL : PH_UHW0;
                    break;
                case PH_UHW0:                                                           // dir is kept reduced mod 6: (k + 5) % 6 = mod6m1[k]
                    if (!improved) { o.it = 0; o.ph = (o.merange >> 1) > 0 ? PH_UDW : PH_DONE; }
                    else {
                        o.dir = mod6m1(key - 1); o.it = 1;
                        if (o.it < (o.merange >> 1)) o.ph = PH_UHW;
                        else { o.it = 0; o.ph = (o.merange >> 1) > 0 ? PH_UDW : PH_DONE; }
                    }
                    break;
                case PH_UHW:
                    if (!improved) { o.it = 0; o.ph = PH_UDW; }
                    else { o.dir = mod6m1(o.dir + key - 1); ++o.it; if (o.it >= (o.merange >> 1)) { o.it = 0; o.ph = PH_UDW; } }
                    break;
                case PH_UDW:                                                            // a step that leaves the mv range is taken and ends the walk
                    if (!improved) o.ph = PH_DONE;
                    else if (o.mx < lm.lox || o.mx > lm.hix || o.my < lm.loy || o.my > lm.hiy || ++o.it >= (o.merange >> 1)) o.ph = PH_DONE;
                    break;
                default: break;
                }
                first = false;
                int dpx;
                if (o.ph == PH_DONE || phase_desc(o, nstart, dpx) != 0) break;          // a phase without candidates passes as "no improvement"
            } while (true);
        }
#ifdef KS_EXP_ME_CLOCK
        { const long long tr3 = ME_NOW(); acc[0] += 1; acc[1] += tr1 - tr0; acc[2] += tr2 - tr1; acc[3] += tr3 - tr2; }
#endif
    }
#ifdef KS_EXP_ME_CLOCK
    if (t == 0 && (WG || qx0 + qy0 == 0)) { for (int i = 0; i < 5; ++i) atomicAdd(&ks_me_dbg[level * 8 + i], (unsigned long long)acc[i]); atomicAdd(&ks_me_dbg[level * 8 + 5], (unsigned long long)(ME_NOW() - tl0)); }
#endif
    // ---- results of the group
    if (is_owner && inside) {
        const int idx = ks_pu_index(level, px, py);
        L.pmv[idx] = (o.mx & 0xFFFF) | (o.my << 16);
        ks265_pu e;
        e.mvx = (int16_t)(o.mx << 2); e.mvy = (int16_t)(o.my << 2); e.mvpx = (int16_t)(o.pmx << 2); e.mvpy = (int16_t)(o.pmy << 2);
        e.cost = o.cost; e.dist = o.cost - mv_rate(lam, o.mx, o.my, o.pmx, o.pmy);
        out_ctu[idx] = e;
    }
    return rounds;
}

// Heavy CTUs first (round 5).  A work-group's life is its number of search rounds (1.2 - 1.7 us each under load): 29 us on average, but a few dozen CTUs of a picture - a PU
// that walks a hexagon and a diamond for merange / 2 steps each - live 60 - 110 us, and when one of them is dispatched late the kernel ends with 35 % of its duration on
// fewer than 30 of 768 work-group slots (scratch/me_trace.py: 140 us for 78 us of slot time).  The kernel therefore leaves every CTU's rounds behind (work: one word per
// wave); the next search of this frame object - next picture or other list, the same content a few samples on - dispatches the CTUs that were in the heaviest tenth first,
// the rest in the usual XCD-aware order.  Scheduling only: every CTU computes what it always computed.
// The order: block index b runs on XCD b % 8, and ks_xcd_swizzle gives every XCD a contiguous raster range of CTUs (neighbours share window halos in one L2).  Each XCD's range
// keeps its XCD: within it the heavy CTUs (the heaviest scores that together hold at most a tenth of the picture's CTUs) come first, the others follow in their usual order;
// the k-th CTU of XCD x's list is dispatched as block 8 k + x.  One wave per XCD, ranks by ballots.
#define ME_ORDER_MAX 16384
__global__ __launch_bounds__(512) void me_order_kernel(int n, int cols, const unsigned *work, int *order)
{
    __shared__ unsigned char sc[ME_ORDER_MAX];
    __shared__ int part[2][8];
    const int tid = threadIdx.x, lane = tid & 63, x = tid >> 6;
    // a CTU's score: the most rounds any wave of it or of its eight neighbours ran last time (what made a CTU heavy - a moving edge - is in it or next to it now).  Computed here
    // (until the end of round 6 a kernel of its own in front of this one: two dependent launches in front of every search, ~ 17 us of a chain that is all latency)
    for (int ctu = tid; ctu < n; ctu += 512) {
        const int cx = ctu % cols, cy = ctu / cols, rows = n / cols;
        unsigned m = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = min(max(cx + dx, 0), cols - 1), yy = min(max(cy + dy, 0), rows - 1);
                const uint4 w = *(const uint4 *)(work + 4 * ((long)yy * cols + xx));
                m = max(m, max(max(w.x, w.y), max(w.z, w.w)));
            }
        sc[ctu] = (unsigned char)min(255u, m);
    }
    __syncthreads();
    // the threshold: the smallest score v >= 1 with at most n / 10 CTUs at or above it (256: none).  The count falls with v: eight halvings, each a count by ballots (no
    // histogram - most scores are the same few values, 2 000 atomics on three LDS words were most of this kernel's 15 us)
    int lo = 1, hi = 256;
    for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        int c = 0;
        for (int i = tid; i < n; i += 512) c += sc[i] >= mid;
        c = (int)wave_sum((unsigned)c);
        if (lane == 0) part[it & 1][x] = c;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < 8; ++w) tot += part[it & 1][w];
        if (tot <= n / 10) hi = mid; else lo = mid + 1;
    }
    const int thr = hi, per = n >> 3, rem = n & 7, cnt = per + (x < rem ? 1 : 0), first = x * per + min(x, rem);     // XCD x's CTUs: first .. first + cnt - 1 (ks_xcd_swizzle)
    int nheavy = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) { const int j = j0 + lane; nheavy += __popcll(__ballot(j < cnt && sc[first + j] >= thr)); }
    int hs = 0, ls = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < cnt, h = in && sc[first + j] >= thr, l = in && !h;
        const unsigned long long bh = __ballot(h), bl = __ballot(l);
        if (in) { const int k = h ? hs + __popcll(bh & lt) : nheavy + ls + __popcll(bl & lt); order[8 * k + x] = first + j; }
        hs += __popcll(bh); ls += __popcll(bl);
    }
}

__global__ __launch_bounds__(256, 3) void me_int_kernel(KsGeom g, int range, int lam, int method, int hex_thr, const uint8_t *src, const uint8_t *ref,
                                                        const ks265_pu *prev, ks265_pu *out, const short2 *field, int nb0x, int nb0y, const short2 *ctu_off,
                                                        const int *order, unsigned *work)
{
    __shared__ __attribute__((aligned(16))) MeLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef KS_EXP_ME_CLOCK
    const long long tk0 = ME_NOW();
#endif
    const int ctu = order ? order[blockIdx.x] : ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
#ifdef KS_EXP_ME_TRACE
    if (tid == 0 && ctu < 4096) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); ks_me_trace[ctu * 8] = wall_clock64(); ks_me_trace[ctu * 8 + 7] = ((unsigned long long)blockIdx.x << 32) | hw; }
#endif
    const uint8_t *R = ks_org_y(g, ref), *Sp = ks_org_y(g, src);
    MeLim lm;
    {
        const short2 ov = ctu_off ? ctu_off[ctu] : make_short2(0, 0);       // multiples of 16
        lm.ox = ov.x; lm.oy = ov.y;
        ctu_mv_limits(g, range, cx, cy, lm.ox, lm.oy, lm.lox, lm.hix, lm.loy, lm.hiy);
    }
    // reference window: 16-byte global loads from x0 - 80 (16-byte aligned), the dwords of x in [-68, 136) go to LDS.  All loads of a
    // thread are issued before the first store (11 pieces in flight per lane: the prologue is one memory latency, not eleven)
    {
        constexpr int NPIECE = (WIN_ROWS * 14 + 255) / 256;
        uint4 v[NPIECE];
        const uint4 fsrc = *(const uint4 *)(Sp + (long)(cy * 64 + (tid >> 2)) * g.sy + cx * 64 + (tid & 3) * 16);
#pragma unroll
        for (int n = 0; n < NPIECE; ++n) {
            const int i = min(tid + 256 * n, WIN_ROWS * 14 - 1), r = i / 14, c = i - r * 14;
            // rows / 16-byte pieces outside the padded plane are never used by a legal candidate (ctu_mv_limits): clamp the addresses
            const int yy = min(max(cy * 64 - WIN_YT + lm.oy + r, -KS_PAD_Y), g.H + KS_PAD_Y - 1);
            const int xx = min(max(cx * 64 - WIN_LOAD_XL + lm.ox + c * 16, -KS_PAD_Y), g.sy - KS_PAD_Y - 16);
            v[n] = *(const uint4 *)(R + (long)yy * g.sy + xx);
        }
#pragma unroll
        for (int n = 0; n < NPIECE; ++n) {
            const int i = tid + 256 * n, r = i / 14, c = i - r * 14;
            const int j0 = c * 4 - (WIN_LOAD_XL - WIN_XL) / 4;            // first LDS dword of this piece (may hang over either end of the row)
            unsigned *d = (unsigned *)(L.win + r * WIN_STRIDE) + j0;
            if (i < WIN_ROWS * 14) {
                if (j0 >= 0) d[0] = v[n].x;
                if (j0 + 1 >= 0 && j0 + 1 < WIN_STRIDE / 4) d[1] = v[n].y;
                if (j0 + 2 >= 0 && j0 + 2 < WIN_STRIDE / 4) d[2] = v[n].z;
                if (j0 + 3 < WIN_STRIDE / 4) d[3] = v[n].w;
            }
        }
        unsigned *d = (unsigned *)(L.fenc + (tid >> 2) * FENC_STRIDE + (tid & 3) * 16);
        d[0] = fsrc.x; d[1] = fsrc.y; d[2] = fsrc.z; d[3] = fsrc.w;
    }
    if (tid < TB_SIZE) L.ctab[tid] = kCandTab[tid];
    if (tid < 16) { L.best16[tid] = 0xFFFFFFFFu; L.fld16[tid] = (int)0x80000000; }
    const ks265_pu *prev_ctu = prev ? prev + (long)ctu * 85 : nullptr;
    ks265_pu *out_ctu = out + (long)ctu * 85;
    __syncthreads();
    // stage A0, last step: the full-resolution +-1 refinement of the pre-search vectors (`field` = the L1 vectors) of the CTU's sixteen 16x16 blocks,
    // from the window that is in LDS anyway.  item = block (16) x candidate (9) x 8x8 tile (4); cost = SAD + |mx| + |my|, first minimum in raster order
    if (field) {
        for (int it = tid; it < 16 * 36; it += 256) {
            const int blk = it / 36, rem = it - blk * 36, k = rem >> 2, tile = rem & 3;
            const int bx = blk & 3, by = blk >> 2, gx = cx * 4 + bx, gy = cy * 4 + by;
            const bool bvalid = gx < nb0x && gy < nb0y;
            const short2 p = field[min(gy, nb0y - 1) * nb0x + min(gx, nb0x - 1)];
            const int mx = clip3(lm.lox, lm.hix, 2 * p.x + (k % 3 - 1)), my = clip3(lm.loy, lm.hiy, 2 * p.y + (k / 3 - 1));
            const int tx = bx * 16 + (tile & 1) * 8, ty = by * 16 + (tile >> 1) * 8;
            const bool tvalid = bvalid && cx * 64 + tx < g.W && cy * 64 + ty < g.H;
            const int wx = tx + (tvalid ? mx - lm.ox : 0) + WIN_XL, wy = ty + (tvalid ? my - lm.oy : 0) + WIN_YT;
            const uint8_t *pw = L.win + wy * WIN_STRIDE + (wx & ~3), *pf = L.fenc + ty * FENC_STRIDE + tx;
            const unsigned sh = wx & 3;
            unsigned sd = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned w0 = *(const unsigned *)(pw + r * WIN_STRIDE), w1 = *(const unsigned *)(pw + r * WIN_STRIDE + 4), w2 = *(const unsigned *)(pw + r * WIN_STRIDE + 8);
                const unsigned f0 = *(const unsigned *)(pf + r * FENC_STRIDE), f1 = *(const unsigned *)(pf + r * FENC_STRIDE + 4);
                sd = sad_u8x4(f0, align_bytes(w1, w0, sh), sd);
                sd = sad_u8x4(f1, align_bytes(w2, w1, sh), sd);
            }
            if (!tvalid) sd = 0;
            sd += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)sd); sd += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)sd);
            if (tile == 0 && bvalid) atomicMin(&L.best16[blk], ((sd + (unsigned)(abs(mx) + abs(my))) << 4) | (unsigned)k);
        }
        __syncthreads();
        if (tid < 16) {
            const int gx = cx * 4 + (tid & 3), gy = cy * 4 + (tid >> 2);
            if (gx < nb0x && gy < nb0y) {
                const short2 p = field[gy * nb0x + gx];
                const int k = (int)(L.best16[tid] & 15u);
                const int mx = clip3(lm.lox, lm.hix, 2 * p.x + (k % 3 - 1)), my = clip3(lm.loy, lm.hiy, 2 * p.y + (k / 3 - 1));
                L.fld16[tid] = (mx & 0xFFFF) | (my << 16);
            }
        }
        __syncthreads();
    }
#ifdef KS_EXP_ME_CLOCK
    const long long tk1 = ME_NOW();
#endif
#ifdef KS_EXP_ME_PROLOGUE_ONLY
    if (tid == 0) out_ctu[0].cost = L.win[tid * 7] + L.fenc[9];
    return;
#endif
#ifdef KS_EXP_ME_TRACE
    if (tid == 0 && ctu < 4096) ks_me_trace[ctu * 8 + 1] = wall_clock64();
#endif
    // the 64x64 PU: the whole work-group
    unsigned fe[16];
    {
        const uint8_t *pf = L.fenc + ((tid >> 3) & 7) * 8 * FENC_STRIDE + (tid & 7) * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) { fe[2 * r] = *(const unsigned *)(pf + r * FENC_STRIDE); fe[2 * r + 1] = *(const unsigned *)(pf + r * FENC_STRIDE + 4); }
    }
    int rounds = me_group<true, 0>(g, cx, cy, range, lm, lam, method, hex_thr, L, L.grp[0], 0, 0, 0, prev_ctu, out_ctu, field, nb0x, nb0y, tid, fe);
    __syncthreads();                                                                    // its vector is the predictor of everything below
#ifdef KS_EXP_ME_CLOCK
    const long long tk2 = ME_NOW();
#endif
#ifdef KS_EXP_ME_TRACE
    if (tid == 0 && ctu < 4096) ks_me_trace[ctu * 8 + 2] = wall_clock64();
#endif
    // 32x32, 16x16, 8x8: one quadrant per wave, each wave on its own
    const int qx = wave & 1, qy = wave >> 1;
    {
        const int z = lane & 15, tx = (z & 1) | ((z >> 1) & 2), ty = ((z >> 1) & 1) | ((z >> 2) & 2);
        const uint8_t *pf = L.fenc + (qy * 32 + ty * 8) * FENC_STRIDE + qx * 32 + tx * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) { fe[2 * r] = *(const unsigned *)(pf + r * FENC_STRIDE); fe[2 * r + 1] = *(const unsigned *)(pf + r * FENC_STRIDE + 4); }
    }
    rounds += me_group<false, 1>(g, cx, cy, range, lm, lam, method, hex_thr, L, L.grp[wave], 0, qx, qy, prev_ctu, out_ctu, field, nb0x, nb0y, lane, fe);
    rounds += me_group<false, 2>(g, cx, cy, range, lm, lam, method, hex_thr, L, L.grp[wave], 1, qx << 1, qy << 1, prev_ctu, out_ctu, field, nb0x, nb0y, lane, fe);
    rounds += me_group<false, 3>(g, cx, cy, range, lm, lam, method, hex_thr, L, L.grp[wave], 2, qx << 2, qy << 2, prev_ctu, out_ctu, field, nb0x, nb0y, lane, fe);
    if (work && lane == 0) work[4 * (long)ctu + wave] = (unsigned)rounds;
#ifdef KS_EXP_ME_TRACE
    if (lane == 0 && ctu < 4096) ks_me_trace[ctu * 8 + 3 + wave] = wall_clock64();
#endif
#ifdef KS_EXP_ME_CLOCK
    const long long tk3 = ME_NOW();
    if (lane == 0) { atomicAdd(&ks_me_dbg[6], (unsigned long long)(tk1 - tk0)); atomicAdd(&ks_me_dbg[7], (unsigned long long)(tk2 - tk1)); atomicAdd(&ks_me_dbg[14], (unsigned long long)(tk3 - tk2)); atomicAdd(&ks_me_dbg[15], 1ull); }
#endif
}

// ------------------------------------------------------------------ stage A2: vector propagation between neighbouring PUs (include/ks265_hip.h: ks265_me_propagate)
// One work-group per CTU.  Step 1: thread p < 85 collects PU p's candidates: the integer vectors of its left / above / right / below neighbours of the same size
// in `in` (across CTU borders), inside the CTU's limits, not its own, no repeats.  Step 2, per level: thread = (direction k, 8x8 block of the CTU) computes the SAD
// of its block under candidate k of the PU the block belongs to (source: 8 aligned bytes per row, kept in registers for the four levels; reference: three aligned
// dwords + v_alignbyte - the candidates are neighbours' vectors, their samples sit in L2); one thread per PU sums its blocks and takes the candidates in order,
// strict '<' against the running best.  All 85 records go to `out`.
#define PROP_NONE ((int)0x80000000)
__global__ __launch_bounds__(256) void me_propagate_kernel(KsGeom g, int range, int lam, const uint8_t *src, const uint8_t *ref, const ks265_pu *in, ks265_pu *out,
                                                           const short2 *ctu_off)
{
    __shared__ int cand[85][4];                // x & 0xFFFF | y << 16, PROP_NONE = no candidate
    __shared__ unsigned sad8[4][64];           // per direction and 8x8 block: the level in flight
    const int tid = threadIdx.x;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const ks265_pu *in_ctu = in + (long)ctu * 85;
    ks265_pu *out_ctu = out + (long)ctu * 85;
    if (tid < 85) {
        const int l = tid < 1 ? 0 : tid < 5 ? 1 : tid < 21 ? 2 : 3, n = 1 << l, base = ((1 << (2 * l)) - 1) / 3;
        const int px = (tid - base) & (n - 1), py = (tid - base) >> l;
        const ks265_pu o = in_ctu[tid];
        const short2 ov = ctu_off ? ctu_off[ctu] : make_short2(0, 0);
        int lox, hix, loy, hiy;
        ctu_mv_limits(g, range, cx, cy, ov.x, ov.y, lox, hix, loy, hiy);
        int c0 = PROP_NONE, c1 = PROP_NONE, c2 = PROP_NONE, c3 = PROP_NONE;
        if (o.cost != KS_COST_INVALID) {
            const int own_x = o.mvx >> 2, own_y = o.mvy >> 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gx = cx * n + px + (k == 0 ? -1 : k == 2 ? 1 : 0), gy = cy * n + py + (k == 1 ? -1 : k == 3 ? 1 : 0);
                if (gx < 0 || gy < 0 || gx >= g.ctu_cols * n || gy >= g.ctu_rows * n) continue;
                const ks265_pu q = in[(long)((gy >> l) * g.ctu_cols + (gx >> l)) * 85 + base + (gy & (n - 1)) * n + (gx & (n - 1))];
                if (q.cost == KS_COST_INVALID) continue;
                const int mx = q.mvx >> 2, my = q.mvy >> 2;
                if (mx < lox || mx > hix || my < loy || my > hiy) continue;
                if (mx == own_x && my == own_y) continue;
                const int v = (mx & 0xFFFF) | (int)((unsigned)my << 16);
                if (v == c0 || v == c1 || v == c2) continue;             // c3 is the last one set
                if (k == 0) c0 = v; else if (k == 1) c1 = v; else if (k == 2) c2 = v; else c3 = v;
            }
        }
        cand[tid][0] = c0; cand[tid][1] = c1; cand[tid][2] = c2; cand[tid][3] = c3;
    }
    __syncthreads();
    const uint8_t *Sp = ks_org_y(g, src), *R = ks_org_y(g, ref);
    const int blk = tid & 63, k = tid >> 6, bx = blk & 7, by = blk >> 3;
    const int x0 = cx * 64 + bx * 8, y0 = cy * 64 + by * 8;
    const bool inside = x0 + 8 <= g.W && y0 + 8 <= g.H;
    uint2 s8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) s8[r] = inside ? *(const uint2 *)(Sp + (long)(y0 + r) * g.sy + x0) : make_uint2(0u, 0u);
    for (int l = 0; l < 4; ++l) {
        const int n = 1 << l, base = ((1 << (2 * l)) - 1) / 3, sh = 3 - l;
        {
            const int v = cand[base + (by >> sh) * n + (bx >> sh)][k];
            unsigned sd = 0;
            if (v != PROP_NONE && inside) {                              // a PU with candidates lies inside the picture, and so do its blocks
                const int mx = (int)(short)(v & 0xFFFF), my = v >> 16;
                const long off0 = (long)(y0 + my) * g.sy + x0 + mx;
                const unsigned shb = (unsigned)(off0 & 3);               // the strides are multiples of 4: the same for every row
                const uint8_t *p = R + (off0 - (long)shb);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned *q = (const unsigned *)(p + (long)r * g.sy);
                    const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
                    sd = sad_u8x4(s8[r].x, align_bytes(w1, w0, shb), sd);
                    sd = sad_u8x4(s8[r].y, align_bytes(w2, w1, shb), sd);
                }
            }
            sad8[k][blk] = sd;
        }
        __syncthreads();
        if (tid < n * n) {
            const int idx = base + tid, px = tid & (n - 1), py = tid >> l, m = 8 >> l;
            ks265_pu o = in_ctu[idx];
            if (o.cost != KS_COST_INVALID) {
                for (int kk = 0; kk < 4; ++kk) {
                    const int v = cand[idx][kk];
                    if (v == PROP_NONE) continue;
                    unsigned d = 0;
                    for (int yy = 0; yy < m; ++yy)
                        for (int xx = 0; xx < m; ++xx) d += sad8[kk][(py * m + yy) * 8 + px * m + xx];
                    const int mx = (int)(short)(v & 0xFFFF), my = v >> 16;
                    const unsigned c = d + (unsigned)((lam * se_bits_dev((mx << 2) - o.mvpx)) >> 4) + (unsigned)((lam * se_bits_dev((my << 2) - o.mvpy)) >> 4);
                    if (c < o.cost) { o.cost = c; o.dist = d; o.mvx = (short)(mx << 2); o.mvy = (short)(my << 2); }
                }
            }
            out_ctu[idx] = o;
        }
        __syncthreads();
    }
}

extern "C" int ks265_me_propagate(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *in, ks265_pu *out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y || !in || !out) return KS265_POINTER;
    if (in == out) return KS265_NOTSUPPORTED;
    const dim3 grid(f->g.ctu_cols * f->g.ctu_rows), block(256);
    hipLaunchKernelGGL(me_propagate_kernel, grid, block, 0, f->ctx->stream, f->g, f->cfg.me_range, f->cfg.lambda_q4, src.y, ref.y, in, out,
                       f->cfg.pre_search ? (const short2 *)f->pyr[9] : nullptr);
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_me_integer(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *prev_pu, ks265_pu *pu)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y || !pu) return KS265_POINTER;
    if (f->cfg.me_method < 0 || f->cfg.me_method > 2) return KS265_NOTSUPPORTED;
    const short2 *field = nullptr;
    if (f->cfg.pre_search) {                                                 // stage A0 first: the pre-search field of this (source, reference) pair
        const int r = ks265_presearch(f, src, ref, nullptr, nullptr);
        if (r) return r;
        field = (const short2 *)f->pyr[5];                                   // the L1 vectors: the kernel does the full-resolution step itself
    }
    const int nctu = f->g.ctu_cols * f->g.ctu_rows;
    const dim3 grid(nctu), block(256);
    if (!f->me_work && !f->me_order_off && nctu <= ME_ORDER_MAX) {                                   // (first search of this frame object: the counters start at zero = the usual order)
        for (int i = 0; i < 2; ++i) {
            if (hipMalloc((void **)&f->me_work_all[i], (size_t)nctu * 16) != hipSuccess || hipMalloc((void **)&f->me_order_all[i], (size_t)nctu * 5 + 16) != hipSuccess) return KS265_OUTOFMEMORY;
            (void)hipMemsetAsync(f->me_work_all[i], 0, (size_t)nctu * 16, f->ctx->stream);
        }
        if (hipStreamSynchronize(f->ctx->stream) != hipSuccess) return KS265_FAIL;      // (once per frame object: the side stream's first search reads its counters too)
        f->me_work = f->me_work_all[0]; f->me_order = f->me_order_all[0];
    }
    if (f->me_work) {
        hipLaunchKernelGGL(me_order_kernel, dim3(1), dim3(512), 0, f->ctx->stream, nctu, f->g.ctu_cols, (const unsigned *)f->me_work, f->me_order);
    }
    const bool timed = f->profiling && f->ev_k[0] && !(f->side && f->ctx->stream == f->side);      // the main chain's launch alone (a B picture's list-1 search runs on the side stream: its marks would pair with list 0's)
    if (timed) (void)hipEventRecord(f->ev_k[0], f->ctx->stream);
    hipLaunchKernelGGL(me_int_kernel, grid, block, 0, f->ctx->stream, f->g, f->cfg.me_range, f->cfg.lambda_q4, f->cfg.me_method, f->cfg.me_hex_thr, src.y, ref.y, prev_pu, pu,
                       field, (f->g.W + 15) / 16, (f->g.H + 15) / 16, field ? (const short2 *)f->pyr[9] : nullptr, (const int *)f->me_order, f->me_work);
    if (timed && f->ev_k[1]) { (void)hipEventRecord(f->ev_k[1], f->ctx->stream); f->ev_k_valid = true; }
    return ks265_check_launch(f->ctx);
}
