// frame_me_int.hip — Stage A, integer-pel motion search (motionSearchOneRef enc@0x483f40) for every PU of every CTU.
//
// The search PATTERNS are the reference's own control flow — interMeDia enc@0x48fbe0, interMeHex enc@0x48fde0, interMeUMH
// enc@0x4907b0 — as restated from the disassembly in oracle/ks265_me_ref.c and pinned there against traces of the reference binary
// (tests/golden/me_search.npz).  This file runs the same state machines for all PUs of a CTU level at once:
//
//   * one work-group (4 waves) per CTU; the 64x64 source block and the +-66 / +-80 reference window are staged in LDS once
//     (every window byte leaves HBM once per CTU); row pitches 57 / 17 dwords (odd);
//   * levels 64x64 .. 8x8 coarse to fine (a PU's predictor is its nearest valid ancestor's vector);
//   * per level, lanes 0 .. NPU-1 of wave 0 are the OWNERS of the PUs: each holds its PU's search state (phase, best vector, cost,
//     direction, iteration count) in registers and, once per round, emits the candidates of its current phase as JOBS into LDS —
//     4 for a diamond step, 3 / 6 for hexagon steps, 8 for the square refinement, up to 64 for the sparse cross and 128 for the
//     big-hexagon rings (fixed-centre stages: all candidates are independent);
//   * all 256 lanes then evaluate the jobs: ONE LANE = ONE 8x8 TILE OF ONE CANDIDATE (v_sad_u8 on packed dwords, v_alignbyte for
//     the unaligned window rows) — no per-candidate reduction chain for 8x8 PUs, 2 / 4 / 6 xor-shuffle steps for 16 / 32 / 64;
//     a diamond step of any level fills the work-group exactly (NPU x 4 candidates x tiles = 256 lanes);
//   * the job's first lane adds the mv rate and folds (cost, key, x, y) into its PU's 64-bit LDS slot with ds_min_u64.  Keys
//     reproduce the reference's tie rules: the direction codes of the packed (cost << 4) + code / (cost << 3) + code forms, or the
//     scan position for the "first strictly better" stages; key 0 = the incumbent;
//   * owners read the winner and advance their state machine.  Rounds repeat until every owner is done.
#include "frame_common.h"

using namespace ks265;

#define WIN_XL 80                 // window column 0 is picture x = ctu_x*64 - 80 (16-byte aligned loads)
#define WIN_YT 66                 // window row 0 is picture y = ctu_y*64 - 66
#define WIN_W 224                 // loaded bytes per row (x in [-80, 144))
#define WIN_ROWS 196              // y in [-66, 130)
#define WIN_STRIDE 228            // 57 dwords (odd)
#define FENC_STRIDE 68            // 17 dwords (odd)
#define ME_WLIM 66                // candidates further than this from the PU position are not staged: skipped (oracle chk = 2)
#define JOB_CAP 256                // job slots per group and round (an owner that does not fit waits a round)

enum { PH_INIT, PH_DIA, PH_H6, PH_HSTEP, PH_SQUARE, PH_U1, PH_UCROSS, PH_UHEX6, PH_UBIG, PH_UFINAL, PH_UHW0, PH_UHW, PH_UDW, PH_DONE };

__device__ __forceinline__ int nib32(unsigned w, int k) { return (int)((w >> (4 * k)) & 15u); }
__device__ __forceinline__ int nib64(unsigned long long w, int k) { return (int)((w >> (4 * k)) & 15ull); }
// search-pattern tables of the reference (rodata of the binary, values read from the file), stored with a bias
__device__ __forceinline__ int hex2x(int i) { return nib32(0x01343101u, i) - 2; }                   // hex2 enc@0x4e52e0
__device__ __forceinline__ int hex2y(int i) { return nib32(0x20024420u, i) - 2; }
__device__ __forceinline__ int hexagon_x(int i) { return nib32(0x00313140u, i) - 2; }               // Hexagon enc@0x4e5340
__device__ __forceinline__ int hexagon_y(int i) { return nib32(0x00044022u, i) - 2; }
__device__ __forceinline__ int bigx(int i) { return nib64(0x6262808080804480ull, i) - 4; }          // Big_Hexagon_X enc@0x4e5320
__device__ __forceinline__ int bigy(int i) { return nib64(0x1771266235538044ull, i) - 4; }          // Big_Hexagon_Y enc@0x4e5300
__device__ __forceinline__ int mod6m1(int i) { return nib32(0x05432105u, i); }                      // mod6m1 enc@0x4e52c0

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

// per-wave engine state (level 0 uses wave 0's copy for the whole work-group)
struct GroupLds {
    unsigned short jobs[JOB_CAP];          // stub: owner (6) | candidate index k << 6; 0xFFFF = unused slot
    int2 desc[16];                         // per PU of the group, published by its owner: phase | dir << 8 | merange << 16, best x | y << 16
    unsigned long long best[16];           // per PU: (cost << 8 | key) << 32 | (x + 128) << 8 | (y + 128)
    int pred[16];                          // per PU: predictor, x | y << 16
    int njobs, active;
};
struct MeLds {
    uint8_t win[WIN_ROWS * WIN_STRIDE];
    uint8_t fenc[64 * FENC_STRIDE];
    GroupLds grp[4];
    int pmv[85];                           // integer vectors of the finished PUs (predictors of the finer levels)
};

struct Owner {
    int ph, mx, my, pmx, pmy, merange, it, dir;
    unsigned cost, cost0;
};

// number of candidates phase `ph` can emit (before eligibility)
__device__ __forceinline__ int phase_count(const Owner &o, bool root_zero)
{
    switch (o.ph) {
    case PH_INIT: return root_zero ? 2 : 1;
    case PH_DIA: case PH_U1: case PH_UFINAL: case PH_UDW: return 4;
    case PH_H6: case PH_UHEX6: case PH_UHW0: return 6;
    case PH_HSTEP: case PH_UHW: return 3;
    case PH_SQUARE: return 8;
    case PH_UCROSS: return 4 * (o.merange >> 2);                 // i = 4, 12, .. <= 2 * merange - 4: merange >> 2 steps
    case PH_UBIG: return 16 * (o.merange >> 3);
    default: return 0;
    }
}
// k-th candidate of the owner's phase: position, key, and whether the reference would evaluate it (mv-range test where its code has
// one) and the window holds it
__device__ __forceinline__ bool phase_cand(const Owner &o, int k, int range, int &x, int &y, int &key)
{
    bool ranged = false;
    int dx = 0, dy = 0;
    switch (o.ph) {
    case PH_INIT: x = k ? 0 : o.pmx; y = k ? 0 : o.pmy; key = k + 1; return true;
    case PH_DIA: case PH_U1: case PH_UFINAL: case PH_UDW:                               // up 1, down 3, left 4, right 12
        dx = k == 2 ? -1 : (k == 3 ? 1 : 0); dy = k == 0 ? -1 : (k == 1 ? 1 : 0); key = k == 0 ? 1 : (k == 1 ? 3 : (k == 2 ? 4 : 12));
        break;
    case PH_SQUARE:                                                                      // + (-1,-1) 5, (-1,1) 7, (1,-1) 13, (1,1) 15
        if (k < 4) { dx = k == 2 ? -1 : (k == 3 ? 1 : 0); dy = k == 0 ? -1 : (k == 1 ? 1 : 0); key = k == 0 ? 1 : (k == 1 ? 3 : (k == 2 ? 4 : 12)); }
        else { dx = k < 6 ? -1 : 1; dy = (k & 1) ? 1 : -1; key = k == 4 ? 5 : (k == 5 ? 7 : (k == 6 ? 13 : 15)); }
        break;
    case PH_H6: dx = hex2x(k + 1); dy = hex2y(k + 1); key = k + 2; break;
    case PH_HSTEP: dx = hex2x(o.dir + k); dy = hex2y(o.dir + k); key = k + 1; break;
    case PH_UCROSS: { const int i = 4 + 8 * (k >> 2), d = k & 3; dx = d == 0 ? i : (d == 1 ? -i : 0); dy = d == 2 ? i : (d == 3 ? -i : 0); key = k + 1; ranged = true; break; }
    case PH_UHEX6: dx = hexagon_x(k); dy = hexagon_y(k); key = k + 1; ranged = true; break;
    case PH_UBIG: { const int r = (k >> 4) + 1, j = k & 15; dx = r * bigx(j); dy = r * bigy(j); key = k + 1; ranged = true; break; }
    case PH_UHW0: dx = hex2x(k); dy = hex2y(k); key = k + 1; ranged = true; break;
    case PH_UHW: { const int j = o.dir + k; dx = hex2x(j); dy = hex2y(j); key = k + 1; ranged = true; break; }      // dir already reduced mod 6
    default: return false;
    }
    x = o.mx + dx; y = o.my + dy;
    const int lim = ranged ? range : ME_WLIM;
    return abs(x) <= lim && abs(y) <= lim;
}

// One LEVEL of one GROUP of PUs.  WG = true: the group is the CTU's single 64x64 PU and all 256 lanes of the work-group evaluate its
// candidates (work-group barriers between the steps of a round).  WG = false: the group is the part of a level that lies in one 32x32
// quadrant (1 / 4 / 16 PUs) and belongs to ONE WAVE, which runs its rounds on its own: owners = its first lanes, evaluation = its 64
// lanes, no barrier at all (a wave's LDS operations complete in order) - the four waves of a CTU and the waves of the other CTUs on
// the CU drift apart and hide each other's latencies, and a slow PU only holds up its own quadrant.
template <bool WG>
__device__ __forceinline__ void me_group(const KsGeom &g, int cx, int cy, int range, int lam, int method, int hex_thr, MeLds &L, GroupLds &Q, int level,
                                         int l2n /* log2 PUs per side of the group */, int qx0, int qy0 /* PU-grid origin of the group */,
                                         const ks265_pu *prev_ctu, ks265_pu *out_ctu, int t /* lane index inside the group's lanes */)
{
    constexpr int NT = WG ? 256 : 64;
    const int S = 64 >> level, npu = 1 << (2 * l2n), l2t = 6 - 2 * level;              // tiles per PU = 1 << l2t (64, 16, 4, 1)
    const int tpr = 8 >> level;                                                          // tiles per PU row
    auto sync = [&]() { if (WG) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    Owner o;
    o.ph = PH_DONE; o.mx = o.my = o.pmx = o.pmy = 0; o.merange = 0; o.it = 0; o.dir = 0; o.cost = 0; o.cost0 = 0;
    bool root = false, inside = false;
    const int px = qx0 + (t & ((1 << l2n) - 1)), py = qy0 + ((t >> l2n) & ((1 << l2n) - 1));
    const bool is_owner = t < npu;
    if (is_owner) {
        inside = ks_pu_inside(g, cx, cy, level, px, py);
        if (inside) {
            root = true;
            for (int a = level - 1; a >= 0; --a) {
                const int ax = px >> (level - a), ay = py >> (level - a);
                if (ks_pu_inside(g, cx, cy, a, ax, ay)) {
                    const int v = L.pmv[ks_pu_index(a, ax, ay)];
                    o.pmx = (int)(short)(v & 0xFFFF); o.pmy = v >> 16; root = false;
                    break;
                }
            }
            if (root && prev_ctu && prev_ctu[0].cost != KS_COST_INVALID) {
                o.pmx = clip3(-range, range, ((int)prev_ctu[0].mvx + 2) >> 2);
                o.pmy = clip3(-range, range, ((int)prev_ctu[0].mvy + 2) >> 2);
            }
            o.merange = root ? range : max(range >> 2, 4);
            o.ph = PH_INIT;
            Q.pred[t] = (o.pmx & 0xFFFF) | (o.pmy << 16);
        } else {                                                                        // PU not (completely) inside the picture: marked, never searched
            ks265_pu e; e.mvx = e.mvy = e.mvpx = e.mvpy = 0; e.cost = KS_COST_INVALID; e.dist = KS_COST_INVALID;
            out_ctu[ks_pu_index(level, px, py)] = e;
        }
    }
    const bool root_zero = root && (o.pmx | o.pmy);
    const unsigned t1 = 62u << (2 * (6 - level) - 4), t2 = 50u << (2 * (6 - level) - 4);

#pragma unroll 1
    for (;;) {
        // ---- owners: publish the state, take phase_count() job slots (a slot = one candidate, eligible or not), fill them with stubs
        //      (owner, k); the candidates themselves are expanded by the evaluating lanes
        bool emitted = false;
        const bool pending = o.ph != PH_DONE;
        if (WG ? t < 64 : true) {
            const int cnt = pending ? phase_count(o, root_zero) : 0;
            if (t == 0) Q.njobs = 0;
            __builtin_amdgcn_wave_barrier();
            if (cnt > 0) {
                const int base = atomicAdd(&Q.njobs, cnt);                              // slot order is arbitrary: the winner does not depend on it
                emitted = base + cnt <= JOB_CAP;                                        // an owner that does not fit waits for the next round
                const int end = min(base + cnt, JOB_CAP);
                for (int s = base; s < end; ++s) Q.jobs[s] = emitted ? (unsigned short)(t | ((s - base) << 6)) : (unsigned short)0xFFFF;
                if (emitted) {
                    Q.desc[t] = make_int2(o.ph | (o.dir << 8) | (o.merange << 16), (o.mx & 0xFFFF) | (o.my << 16));
                    Q.best[t] = o.ph == PH_INIT ? ~0ull : ((unsigned long long)(o.cost << 8) << 32);
                }
            }
            if (WG) { const unsigned long long anyp = __ballot(pending); if (t == 0) Q.active = anyp != 0ull; }
        }
        sync();
        bool active;
        if (WG) active = Q.active != 0; else active = __ballot(pending) != 0ull;
        if (!active) break;
        const int njobs = min(Q.njobs, JOB_CAP);
        // ---- all lanes of the group: one lane = one 8x8 tile of one candidate
        const int items = njobs << l2t;
        for (int base = 0; base < items; base += NT) {
            const int it = base + t;
            unsigned sad = 0;
            int pu = 0, x = 0, y = 0, key = 0;
            bool live = it < items;
            if (live) {
                const unsigned stub = Q.jobs[it >> l2t];
                live = stub != 0xFFFFu;
                pu = stub & 63;
                const int2 d = Q.desc[pu & 15];
                Owner c;
                c.ph = d.x & 255; c.dir = (d.x >> 8) & 255; c.merange = d.x >> 16; c.mx = (int)(short)(d.y & 0xFFFF); c.my = d.y >> 16;
                c.pmx = c.pmy = 0;
                if (c.ph == PH_INIT) { const int pr = Q.pred[pu & 15]; c.pmx = (int)(short)(pr & 0xFFFF); c.pmy = pr >> 16; }
                live = live && phase_cand(c, (int)(stub >> 6), range, x, y, key);
            }
            if (live) {
                const int tile = it & ((1 << l2t) - 1);
                const int ppx = qx0 + (pu & ((1 << l2n) - 1)), ppy = qy0 + (pu >> l2n);
                const int bx = ppx * S + (tile & (tpr - 1)) * 8, by = ppy * S + (tile >> (3 - level)) * 8;
                const int wx = bx + x + WIN_XL, wy = by + y + WIN_YT;
                const uint8_t *p = L.win + wy * WIN_STRIDE + (wx & ~3);
                const uint8_t *f = L.fenc + by * FENC_STRIDE + bx;
                const unsigned sh = wx & 3;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned w0 = *(const unsigned *)(p + r * WIN_STRIDE), w1 = *(const unsigned *)(p + r * WIN_STRIDE + 4), w2 = *(const unsigned *)(p + r * WIN_STRIDE + 8);
                    const unsigned f0 = *(const unsigned *)(f + r * FENC_STRIDE), f1 = *(const unsigned *)(f + r * FENC_STRIDE + 4);
                    sad = sad_u8x4(f0, align_bytes(w1, w0, sh), sad);
                    sad = sad_u8x4(f1, align_bytes(w2, w1, sh), sad);
                }
            }
            // the tiles of a job are adjacent lanes: 1 / 4 / 16 / 64 of them
            if (level <= 2) { sad += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)sad); sad += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)sad); }
            if (level <= 1) { sad += (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)sad); sad += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)sad); }
            if (level == 0) { sad += (unsigned)__builtin_amdgcn_ds_swizzle((int)sad, 0x1F | (16 << 10)); sad += (unsigned)__shfl_xor((int)sad, 32, 64); }
            if (live && (it & ((1 << l2t) - 1)) == 0) {
                const int pr = Q.pred[pu & 15];
                const unsigned cost = sad + mv_rate(lam, x, y, (int)(short)(pr & 0xFFFF), pr >> 16);
                const unsigned long long v = ((unsigned long long)((cost << 8) | (unsigned)key) << 32) | (unsigned long long)(((unsigned)(x + 128) << 8) | (unsigned)(y + 128));
                atomicMin(&Q.best[pu & 15], v);
            }
        }
        sync();
        // ---- owners advance (the transitions of oracle/ks265_me_ref.c)
        if (emitted) {
            bool first = true;
            do {
                bool improved = false;
                int key = 0, wx = o.mx, wy = o.my;
                unsigned wcost = o.cost;
                if (first) {
                    const unsigned long long b = Q.best[t];
                    key = (int)((b >> 32) & 255u);
                    improved = key != 0;
                    if (improved) { wcost = (unsigned)(b >> 40); wx = (int)((b >> 8) & 255u) - 128; wy = (int)(b & 255u) - 128; }
                }
                first = false;
                if (o.ph != PH_HSTEP) { o.cost = wcost; o.mx = wx; o.my = wy; }         // interMeHex's walk may refuse the move (below)
                switch (o.ph) {
                case PH_INIT: {
                    const unsigned sad0 = o.cost - mv_rate(lam, o.mx, o.my, o.pmx, o.pmy);
                    if (method == 0) { o.ph = o.merange > 0 ? PH_DIA : PH_DONE; o.it = 0; }
                    else if (method == 1 || (hex_thr > 0 && sad0 < ((unsigned)hex_thr << (2 * (6 - level))))) o.ph = PH_H6;
                    else { o.ph = PH_U1; o.cost0 = o.cost; }
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
                    if (!improved || abs(wx) > range || abs(wy) > range) o.ph = PH_SQUARE;
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
                    else if (abs(o.mx) > range || abs(o.my) > range || ++o.it >= (o.merange >> 1)) o.ph = PH_DONE;
                    break;
                default: break;
                }
            } while (o.ph != PH_DONE && phase_count(o, root_zero) == 0);                // a phase without candidates passes as "no improvement"
        }
    }
    // ---- results of the group
    if (is_owner && inside) {
        const int idx = ks_pu_index(level, px, py);
        L.pmv[idx] = (o.mx & 0xFFFF) | (o.my << 16);
        ks265_pu e;
        e.mvx = (int16_t)(o.mx << 2); e.mvy = (int16_t)(o.my << 2); e.mvpx = (int16_t)(o.pmx << 2); e.mvpy = (int16_t)(o.pmy << 2);
        e.cost = o.cost; e.dist = o.cost - mv_rate(lam, o.mx, o.my, o.pmx, o.pmy);
        out_ctu[idx] = e;
    }
}

__global__ __launch_bounds__(256, 3) void me_int_kernel(KsGeom g, int range, int lam, int method, int hex_thr, const uint8_t *src, const uint8_t *ref,
                                                        const ks265_pu *prev, ks265_pu *out)
{
    __shared__ __attribute__((aligned(16))) MeLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctu = ks_xcd_swizzle(blockIdx.x, g.ctu_cols * g.ctu_rows), cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const uint8_t *R = ks_org_y(g, ref), *Sp = ks_org_y(g, src);
    // reference window: 16-byte global loads (x0 - 80 is 16-byte aligned), dword LDS stores
    for (int i = tid; i < WIN_ROWS * (WIN_W / 16); i += 256) {
        const int r = i / (WIN_W / 16), c = i - r * (WIN_W / 16);
        const int yy = min(cy * 64 - WIN_YT + r, g.H + KS_PAD_Y - 1);     // rows past the border are never used by a valid PU
        const uint4 v = *(const uint4 *)(R + (long)yy * g.sy + cx * 64 - WIN_XL + c * 16);
        unsigned *d = (unsigned *)(L.win + r * WIN_STRIDE + c * 16);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    {
        const int r = tid >> 2, c = tid & 3;
        const uint4 v = *(const uint4 *)(Sp + (long)(cy * 64 + r) * g.sy + cx * 64 + c * 16);
        unsigned *d = (unsigned *)(L.fenc + r * FENC_STRIDE + c * 16);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    const ks265_pu *prev_ctu = prev ? prev + (long)ctu * 85 : nullptr;
    ks265_pu *out_ctu = out + (long)ctu * 85;
    __syncthreads();
    // the 64x64 PU: the whole work-group
    me_group<true>(g, cx, cy, range, lam, method, hex_thr, L, L.grp[0], 0, 0, 0, 0, prev_ctu, out_ctu, tid);
    __syncthreads();                                                                    // its vector is the predictor of everything below
    // 32x32, 16x16, 8x8: one quadrant per wave, each wave on its own
    const int qx = wave & 1, qy = wave >> 1;
#pragma unroll 1
    for (int level = 1; level < 4; ++level)
        me_group<false>(g, cx, cy, range, lam, method, hex_thr, L, L.grp[wave], level, level - 1, qx << (level - 1), qy << (level - 1), prev_ctu, out_ctu, lane);
}

extern "C" int ks265_me_integer(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *prev_pu, ks265_pu *pu)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y || !pu) return KS265_POINTER;
    if (f->cfg.me_method < 0 || f->cfg.me_method > 2) return KS265_NOTSUPPORTED;
    const dim3 grid(f->g.ctu_cols * f->g.ctu_rows), block(256);
    hipLaunchKernelGGL(me_int_kernel, grid, block, 0, f->ctx->stream, f->g, f->cfg.me_range, f->cfg.lambda_q4, f->cfg.me_method, f->cfg.me_hex_thr, src.y, ref.y, prev_pu, pu);
    return ks265_check_launch(f->ctx);
}
