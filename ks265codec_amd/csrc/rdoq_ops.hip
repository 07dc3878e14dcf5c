// rdoq_ops.hip — SURVEY.md 8(f) rank 3: the reference's rate-distortion optimised quantisation as a device operator.
//
// h265_codec::rdoQuant enc@0x4aac50 (EncQuant.cpp; 28 % of the reference's CPU time at -preset slow) decides, per transform block, which of the quantiser's levels
// to keep, lower by one or drop, where the block's last significant level sits and which level carries a hidden sign - a walk over the coefficients in REVERSE
// scan order whose rates depend on the entropy coder's context state at each step (greater-1 / greater-2 context sets, Rice parameter).  The bit tables it prices
// with are estBitRdoq enc@0x46a8a0's (180 words per block size and component, built from the CABAC context states): the host snapshots them, the device evaluates
// every transform block of a batch in parallel.
//
// Mapping: ONE WAVE PER TRANSFORM BLOCK.  Levels, coefficients, scan tables and the per-position cost records live in LDS; a 4x4 coefficient group is loaded with
// lane = scan position (its distortions, significance contexts and table words are computed 16 at a time), the state walk itself runs wave-uniformly on values
// taken from those lanes with v_readlane (the reverse scan "in registers"); groups the quantiser left empty, the sign restoration and the zeroing behind the new
// last position are plain lane-parallel loops; sign-data hiding prices its 16 candidates of a group in parallel and picks with a lexicographic wave minimum.
// All costs are int64 exactly as in the reference (distortion << (2 log2 N + 1), rate x lambda >> 8).
//
// Pinned: tests/test_gpu_rdoq.py replays the calls recorded inside real appencoder runs (tests/golden/rdoq.npz) and transform blocks taken from this pipeline.
#include "ks265_internal.h"

namespace {

// normative tables (H.265 9.3.4.2.5, 6.5.3, 9.3.3.12)
__device__ const unsigned char kSigCtxD[5][16] = {{2, 1, 1, 0, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0}, {2, 2, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0}, {2, 1, 0, 0, 2, 1, 0, 0, 2, 1, 0, 0, 2, 1, 0, 0},
                                                  {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2}, {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}};
__device__ const unsigned char kScan4D[3][16] = {{0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15}};
__device__ const unsigned char kLastGrpD[32] = {0, 1, 2, 3, 0x14, 0x14, 0x15, 0x15, 0x26, 0x26, 0x26, 0x26, 0x27, 0x27, 0x27, 0x27, 0x38, 0x38, 0x38, 0x38, 0x38, 0x38, 0x38, 0x38, 0x39, 0x39, 0x39, 0x39, 0x39, 0x39, 0x39, 0x39};
__device__ const unsigned char kRiceThrD[5] = {7, 14, 26, 46, 78};

struct RdoqLds {
    long long cost_coeff[1024], cost_sig[1024], cost0[1024], cost_cgsig[64];       // by scan position / by scan position / by raster position / by group
    int inc_up[1024], inc_down[1024], sig_delta[1024];                              // by raster position
    short S[1024], G[64];                                                           // scan position -> raster position; group scan index -> group raster index
    short lvl[1024], coef[1024];
    unsigned short sigmask[64];
    int T[180];
};

__device__ __forceinline__ long long rl64(long long v, int p)
{
    const int lo = __builtin_amdgcn_readlane((int)v, p), hi = __builtin_amdgcn_readlane((int)(v >> 32), p);
    return ((long long)hi << 32) | (long long)(unsigned)lo;
}
__device__ __forceinline__ long long shx64(long long v, int m)
{
    const int lo = __shfl_xor((int)v, m, 64), hi = __shfl_xor((int)(v >> 32), m, 64);
    return ((long long)hi << 32) | (long long)(unsigned)lo;
}
__device__ __forceinline__ long long wave_sum64(long long v) { for (int m = 32; m; m >>= 1) v += shx64(v, m); return v; }
__device__ __forceinline__ int wave_sum32(int v) { for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m, 64); return v; }

// bits (x 32768) of coeff_abs_level_remaining as the function counts them (enc@0x4ab741.., 0x4abc4b, 0x4abec8..)
__device__ __forceinline__ int remain_bits(unsigned sym, int r)
{
    const unsigned p = sym >> r;
    if (p <= 2) return (int)((unsigned)r + p + 1) << 15;
    int lg = 0;
    if (p != 3) lg = 31 - __clz(p - 2);
    return (r + 2 * lg + 4) << 15;
}
__device__ __forceinline__ int remain_bits_cap(unsigned sym, int r) { unsigned v = (sym >> r) + (unsigned)r + 1; if (v > 8) v = 8; return (int)(v << 15); }
__device__ __forceinline__ int remain_bits_thr(unsigned sym, int r, unsigned thr)
{
    int extra = 0;
    if (thr < sym) { const unsigned d = sym - thr; const int lg = 31 - __clz(d); extra = (2 * lg + 1) << 15; sym = thr + 1; }
    return extra + remain_bits_cap(sym, r);
}

__global__ __launch_bounds__(64) void rdoq_kernel(const ks265_rdoq_tu *tus, int n, short *lvl_g, const short *coef_g, const int *tabs, unsigned short *sigmask_g, int *out, unsigned long long *hidden_g,
                                                  const int *n_dev /* the pixel path (cfg.rdoq): the number of blocks is on the device, the grid is its upper bound */)
{
    __shared__ RdoqLds L;
    const int lane = threadIdx.x, ti = blockIdx.x;
    if (n_dev) n = *n_dev;
    if (ti >= n) return;
    const ks265_rdoq_tu d = tus[ti];
    const int log2 = d.log2, N = 1 << log2, NN = N * N, w = N >> 2, ncg = w * w, scan_idx = d.scan_idx, luma = d.comp == 0;
    const int dq = d.dq, per = d.per, last_pos = d.last_pos, sdh = d.sdh;
    const long long lam = d.lam, lam_sdh = d.lam_sdh;
    const int shift = log2 - 1, add = shift > per ? 1 << (shift - per - 1) : 0, esh = 2 * log2 + 1;
#define SQ(c) ((long long)(c) * (c) << esh)
#define RATE(bits) (((long long)(bits) * lam) >> 8)
    for (int k = lane; k < NN; k += 64) {
        L.lvl[k] = lvl_g[d.off + k]; L.coef[k] = coef_g[d.off + k];
        L.cost_coeff[k] = 0; L.cost_sig[k] = 0; L.cost0[k] = 0; L.inc_up[k] = 0; L.inc_down[k] = 0; L.sig_delta[k] = 0;
    }
    for (int k = lane; k < 180; k += 64) L.T[k] = tabs[(long)d.tab * 180 + k];
    L.sigmask[lane] = lane < ncg ? sigmask_g[(long)ti * 64 + lane] : (unsigned short)0;
    L.cost_cgsig[lane] = 0;
    if (lane == 0) {                                             // the scan of the 4x4 groups (the table behind enc@0x6ff360)
        int m = 0;
        if (scan_idx == 0) { for (int dd = 0; m < ncg; ++dd) for (int y = dd; y >= 0; --y) { const int x = dd - y; if (x < w && y < w) L.G[m++] = (short)(y * w + x); } }
        else if (scan_idx == 1) { for (int q = 0; q < ncg; ++q) L.G[q] = (short)q; }
        else { for (int x = 0; x < w; ++x) for (int y = 0; y < w; ++y) L.G[m++] = (short)(y * w + x); }
    }
    __syncthreads();
    for (int sp = lane; sp < NN; sp += 64) {
        const int g = L.G[sp >> 4], cx = g % w, cy = g / w, p = kScan4D[scan_idx][sp & 15];
        L.S[sp] = (short)((cy * 4 + (p >> 2)) * N + cx * 4 + (p & 3));
    }
    __syncthreads();
    const int last_cg = last_pos >> 4;
    long long uncoded, base;
    {   // groups behind the last significant one: uncoded
        long long acc = 0;
        for (int sp = (last_cg + 1) * 16 + lane; sp < NN; sp += 64) { const int blk = L.S[sp]; const long long e = SQ((int)L.coef[blk]); L.cost0[blk] = e; acc += e; }
        uncoded = base = wave_sum64(acc);
    }
    unsigned long long cgmask = 0;
    int c1 = 1;
    const int sig_off = log2 == 2 ? 0 : log2 == 3 ? ((luma && scan_idx) ? 15 : 9) : (luma ? 21 : 12);
    const int pl = lane & 15;                                    // scan position inside a group (lanes 16..63 mirror lanes 0..15: their values are never taken)
#pragma unroll 1
    for (int cg = last_cg; cg >= 0; --cg) {
        const int cgpos = L.G[cg], cgx = cgpos % w, cgy = cgpos / w;
        const int right = cgx + 1 < w ? (int)((cgmask >> (cgpos + 1)) & 1) : 0, lower = cgy + 1 < w ? (int)((cgmask >> (cgpos + w)) & 1) : 0;
        const int pattern = log2 == 2 ? 4 : right + 2 * lower;
        int ctx_set = (luma && cg != 0) ? 2 : 0;
        if (c1 == 0) ++ctx_set;
        const int sig_base = sig_off + ((luma && cg != 0) ? 3 : 0);
        unsigned mask = L.sigmask[cg];
        const int blk0 = L.S[cg * 16];                           // top-left sample of the group (first in every scan)
        if (cg != 0 && mask == 0) {                              // enc@0x4ab0c8: nothing quantised to non-zero here.  (k walks the group in RASTER order while the context is that of scan position k - as the function does)
            const int blk = blk0 + (pl >> 2) * N + (pl & 3);
            const long long e = SQ((int)L.coef[blk]);
            const int ctx = kSigCtxD[pattern][kScan4D[scan_idx][pl]] + sig_base;
            if (lane < 16) {
                L.cost0[blk] = e; L.cost_coeff[cg * 16 + pl] = e;
                L.sig_delta[blk] = L.T[46 + ctx] - L.T[4 + ctx];
                L.cost_sig[cg * 16 + pl] = RATE(L.T[4 + ctx]);
            }
            const long long es = wave_sum64(lane < 16 ? e : 0);
            uncoded += es; base += es;
            c1 = 1;
            const long long cs = RATE(L.T[2 * ((right | lower) & 1)]);
            if (lane == 0) L.cost_cgsig[cg] = cs;
            base += cs;
            continue;
        }
        // enc@0x4ab310: the group's coefficients, lane = scan position
        const int sp_l = cg * 16 + pl, blk_l = L.S[sp_l];
        const int c_l = L.coef[blk_l];
        const long long e0_l = SQ(c_l);
        const int ctx_l = blk_l == 0 ? 0 : kSigCtxD[pattern][kScan4D[scan_idx][pl]] + sig_base;
        const int t4_l = L.T[4 + ctx_l], t46_l = L.T[46 + ctx_l];
        const int q_l = L.lvl[blk_l] < 0 ? -L.lvl[blk_l] : L.lvl[blk_l], ac_l = c_l < 0 ? -c_l : c_l;
        if (lane < 16) L.cost0[blk_l] = e0_l;
        uncoded += wave_sum64(lane < 16 ? e0_l : 0);
        // per-position results, kept by lane == position
        long long o_cc = 0, o_cs = 0;
        int o_sd = 0, o_up = 0, o_dn = 0, o_lvl = L.lvl[blk_l], o_w = 0;           // o_w bits: 1 cost_coeff / cost_sig, 2 sig_delta, 4 inc_up, 8 inc_down, 16 lvl
        int c1idx = 0, c2idx = 0, c2 = 0, rice = 0, nnz_before0 = 0;
        long long sig_sum = 0, sig_cost0 = 0, st_coded = 0, st_uncoded = 0;
        c1 = 1;
#pragma unroll
        for (int p = 15; p >= 0; --p) {
            const int sp = cg * 16 + p;
            const long long e0 = rl64(e0_l, p);
            if (sp > last_pos) { base += e0; if (lane == p) { o_cc = 0; o_cs = 0; o_w |= 1; } continue; }
            const int bit = 15 - p;
            const int t4 = __builtin_amdgcn_readlane(t4_l, p), t46 = __builtin_amdgcn_readlane(t46_l, p);
            const int *g1 = &L.T[108 + 2 * (c1 + 4 * ctx_set)];
            const int g1_0 = g1[0], g1_1 = g1[1];
            if (!((mask >> bit) & 1)) {                          // quantised to zero: stays zero
                const long long cs = RATE(t4), cc = e0 + cs;
                base += cc;
                if (lane == p) { o_cs = cs; o_cc = cc; o_sd = t46 - t4; o_up = g1_0; o_w |= 1 | 2 | 4; }
                sig_sum += cs; if (p == 0) sig_cost0 = cs;
                continue;
            }
            const int q = __builtin_amdgcn_readlane(q_l, p), ac = __builtin_amdgcn_readlane(ac_l, p);
            const int fl = (c1idx < 8 ? 1 : 0) + (c2idx == 0 ? 2 : 0);     // bit 0: a greater1 flag is coded for this one, bit 1: a greater2 flag could be
            const int base_level = (0xd9 >> (2 * fl)) & 3;
            const int *g2 = &L.T[156 + 2 * (c2 + ctx_set)];
            const int g2_0 = g2[0], g2_1 = g2[1];
            long long best = 0x7fffffffffffffffll, cc = 0x7fffffffffffffffll, cs = 0;
            int sig1 = 0, sdelta = 0;
            if (sp != last_pos) {
                if (q <= 2) { cs = RATE(t4); cc = cs + e0; best = cc; }
                sig1 = t46;
                sdelta = sig1 - t4;
            }
            int chosen = 0;
            {
                int rq[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {                     // rate of the magnitude beyond the significance flag (enc@0x4ab730..0x4ab7e8), for q and q - 1
                    const int v = q - k;
                    int r = 0;
                    if (v != 0) {
                        if (v < base_level) r = v == 2 ? g2_0 + g1_1 : g1_0;
                        else {
                            r = remain_bits((unsigned)(v - base_level), rice);
                            if (fl & 1) r += g1_1;
                            if (fl == 3) r += g2_1;
                        }
                    }
                    rq[k] = r;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int v = q - k;
                    if (v == 0) break;                            // (q == 1: the alternative is zero, handled above)
                    const int rec = (int)((unsigned)(dq * v + add) >> shift);
                    const long long dd = ac - rec;
                    const long long cost = (dd * dd << esh) + RATE((long long)sig1 + 0x8000 + rq[k]);
                    if (cost < best) { best = cost; cc = cost; cs = RATE(sig1); chosen = v; }
                }
            }
            if (chosen == 0) mask &= ~(1u << bit);
            base += best;
            int up_v = g1_0, dn_v = 0;                            // what one more / one less would cost in rate (the sign-hiding step reads these)
            if (sdh && chosen) {
                const unsigned thr = kRiceThrD[rice];
#define MAGRATE(v) ((v) == 0 ? 0 : (v) < base_level ? ((v) == 2 ? g2_0 + g1_1 : g1_0) : remain_bits_thr((unsigned)((v) - base_level), rice, thr) + ((fl & 1) ? g1_1 : 0) + (fl == 3 ? g2_1 : 0))
                const int cur = MAGRATE(chosen), up = MAGRATE(chosen + 1), dn = chosen == 1 ? 0 : MAGRATE(chosen - 1);
#undef MAGRATE
                up_v = up - cur; dn_v = dn - cur;
            }
            if (lane == p) { o_cc = cc; o_cs = cs; o_sd = sdelta; o_up = up_v; o_dn = dn_v; o_lvl = chosen; o_w |= 1 | 2 | 4 | 8 | 16; }
            if (chosen >= base_level && rice <= 3 && chosen > (3 << rice)) ++rice;
            if (chosen) ++c1idx;
            if (chosen > 1) { ++c2idx; c1 = 0; if (c2 < 2) ++c2; }
            else if ((c1 == 1 || c1 == 2) && chosen) ++c1;
            sig_sum += cs; if (p == 0) sig_cost0 = cs;
            if (chosen) { st_coded += cc - cs; st_uncoded += e0; cgmask |= 1ull << cgpos; nnz_before0 += p; }
        }
        if (lane < 16) {
            if (o_w & 1) { L.cost_coeff[sp_l] = o_cc; L.cost_sig[sp_l] = o_cs; }
            if (o_w & 2) L.sig_delta[blk_l] = o_sd;
            if (o_w & 4) L.inc_up[blk_l] = o_up;
            if (o_w & 8) L.inc_down[blk_l] = o_dn;
            if (o_w & 16) L.lvl[blk_l] = (short)o_lvl;
        }
        long long cgs = 0;
        if (cg != last_cg && cg != 0) {
            const int ctx = (right | lower) & 1;
            if (!((cgmask >> cgpos) & 1)) { cgs = RATE(L.T[2 * ctx]); base += cgs; base -= sig_sum; }
            else {
                if (nnz_before0 == 0) { sig_sum -= sig_cost0; base -= sig_cost0; }
                const long long zero_cost = RATE(L.T[2 * ctx]) + base + st_uncoded - st_coded - sig_sum;
                const long long one = RATE(L.T[2 * ctx + 1]);
                base += one;
                if (zero_cost < base) {
                    cgs = RATE(L.T[2 * ctx]); base = zero_cost; cgmask &= ~(1ull << cgpos); mask = 0;
                    if (lane < 16) L.lvl[blk_l] = 0;
                } else cgs = one;
            }
        }
        if (lane == 0) { L.cost_cgsig[cg] = cgs; L.sigmask[cg] = (unsigned short)mask; }
    }
    __syncthreads();
    // the coded-block flag of the block against "all zero", then the last position: walking down from the last level while levels are <= 1
    long long best_cost, bc = base;
    {
        int i0, i1;
        if (!d.flag_a4c0 && luma && d.tu5 == 0) { i0 = 178; i1 = 179; }
        else { const int idx = luma ? (d.tu5 == 0) : (int)d.tu5; i0 = 168 + 2 * idx; i1 = 169 + 2 * idx; }
        best_cost = uncoded + RATE(L.T[i0]); bc += RATE(L.T[i1]);
    }
    int best_last = 0;
    if (last_cg >= 0) {
        bool done = false;
#pragma unroll 1
        for (int cg = last_cg; cg >= 0 && !done; --cg) {
            if (cg != last_cg && cg != 0) { bc -= L.cost_cgsig[cg]; if (!((cgmask >> L.G[cg]) & 1)) continue; }
            unsigned sm = L.sigmask[cg];
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const int sp = cg * 16 + 15 - k;
                if (sp > last_pos) continue;
                const int blk = L.S[sp], l = L.lvl[blk];
                const long long csig = L.cost_sig[sp];
                if (!l) { bc -= csig; continue; }
                int px = blk & (N - 1), py = blk >> log2;
                if (scan_idx == 2) { const int t = px; px = py; py = t; }
                const int gx = kLastGrpD[px], gy = kLastGrpD[py];
                const int bits = ((gx >> 4) << 15) + L.T[88 + (gx & 15)] + ((gy >> 4) << 15) + L.T[98 + (gy & 15)];
                const long long tot = bc - csig + RATE(bits);
                if (tot < best_cost) { best_cost = tot; best_last = sp + 1; sm &= (0xffffu << k) & 0xffffu; }
                if (l > 1) { done = true; break; }
                bc -= L.cost_coeff[sp]; bc += L.cost0[blk];
            }
            if (lane == 0) L.sigmask[cg] = (unsigned short)sm;
        }
    }
    __syncthreads();
    // signs from the coefficients, zeros behind the chosen last position (enc@0x4ac500..0x4ac5e8: the rest of the group that holds it up to the old last position, then
    // every later group that starts at or before the old last position)
    int nz = 0;
    for (int i = lane; i < best_last; i += 64) { const int blk = L.S[i]; const int s = (int)L.coef[blk] >> 15; const int l = L.lvl[blk]; if (l) ++nz; L.lvl[blk] = (short)((l ^ s) - s); }
    nz = wave_sum32(nz);
    {
        const int e = (best_last | 15) < last_pos ? (best_last | 15) : last_pos;
        for (int i = best_last + lane; i <= e; i += 64) L.lvl[L.S[i]] = 0;
        const int b0 = (best_last & ~15) + 16;
        if (b0 <= last_pos) { const int b1 = (last_pos & ~15) + 16; for (int i = b0 + lane; i < b1; i += 64) L.lvl[L.S[i]] = 0; }
    }
    __syncthreads();
    // enc@0x4ac5ea..0x4acd3b: sign-data hiding with a rate-distortion choice of the level to move (the second lambda).  Levels carry their signs here.
    unsigned long long hidden = 0;
    if (sdh && nz > 1 && best_last > 0) {
        int first_cg = 1;                                        // the group that holds the last position: its candidates start at that position
        const int rec1 = (add + dq) >> shift;
#define RATE2(bits) (((long long)(bits) * lam_sdh) >> 8)
#pragma unroll 1
        for (int cg = (best_last - 1) >> 4; cg >= 0; --cg) {
            unsigned mask = L.sigmask[cg];
            if (!mask) continue;
            const int blk = L.S[cg * 16 + pl], lv = L.lvl[blk], cf = L.coef[blk];
            const unsigned nzb = (unsigned)(__ballot(lv != 0) & 0xffffull);
            const int last = nzb ? 31 - __clz(nzb) : -1, first = nzb ? __ffs(nzb) - 1 : 16;
            if ((last & 0xffff) - first <= 3) { first_cg = 0; continue; }
            const int lvf = __builtin_amdgcn_readlane(lv, first & 15);          // (first < 16 here: last - first > 3 with last <= 15, or the function's own last = -1 case, which a non-zero mask excludes)
            const int neg = lvf <= 0;
            const int sum = wave_sum32(lane < 16 ? lv : 0);
            unsigned long long keep = 1;
            if ((sum & 1) != neg) {
                const int n = pl, ac = cf < 0 ? -cf : cf, al = lv < 0 ? -lv : lv;
                const int dlt = ac - ((add + al * dq) >> shift);
                const long long d0 = SQ(dlt);
                bool ok = lane < 16 && n <= (first_cg ? last : 15);
                long long cost = 0; int delta = 0;
                const int up_i = L.inc_up[blk], dn_i = L.inc_down[blk], sd_i = L.sig_delta[blk];
                if (lv == 0) {
                    if (first > n && (cf < 0) != neg) ok = false;                  // would become the first level of the group: its sign must be the one hidden now
                    const int d1 = ac - rec1;
                    cost = SQ(d1) - d0 + RATE2(up_i + sd_i + 0x8000); delta = 1;
                } else {
                    const int du = ac - ((add + (al + 1) * dq) >> shift), dd = ac - ((add + (al - 1) * dq) >> shift);
                    const long long up = SQ(du) - d0 + RATE2(up_i);
                    long long dn = SQ(dd) - d0;
                    if (al == 1) {
                        dn += RATE2(dn_i - (sd_i + 0x8000));
                        if (last == n && first_cg) dn -= 0x20000;
                    } else dn += RATE2(dn_i);
                    if (dn > up) { cost = up; delta = 1; }
                    else { if (al == 1 && n == first) ok = false; cost = dn; delta = -1; }
                }
                // the walk goes from the highest position down and takes a strictly smaller cost: minimum cost, ties to the highest position
                long long kc = ok ? cost : 0x7fffffffffffffffll;
                int kn = ok ? n : -1;
                for (int m = 8; m; m >>= 1) {
                    const long long oc = shx64(kc, m); const int on = __shfl_xor(kn, m, 64);
                    if (oc < kc || (oc == kc && on > kn)) { kc = oc; kn = on; }
                }
                kn = __builtin_amdgcn_readfirstlane(kn);
                if (kn >= 0) {
                    const int best_n = kn;
                    const int best_blk = L.S[cg * 16 + best_n];
                    int best_delta = __shfl(delta, best_n, 64);
                    const int old = L.lvl[best_blk];
                    if (old == 32767 || old == -32768) best_delta = -1;
                    if (old == 0) ++nz; else if (best_delta == -1 && (old == 1 || old == -1)) --nz;
                    const int nv = L.coef[best_blk] < 0 ? old - best_delta : old + best_delta;
                    __syncthreads();
                    if (lane == 0) L.lvl[best_blk] = (short)nv;
                    if (old == 0) mask |= 1u << (15 - best_n);
                    if (nv == 0) {
                        mask &= ~(1u << (15 - best_n));
                        if (best_n == last) {
                            int nl = 15; while (nl >= 0 && !((mask >> (15 - nl)) & 1)) --nl;       // the group's new last level
                            keep = nl - first > 3;
                            if (best_blk == L.S[best_last - 1]) best_last -= best_n - nl;
                        }
                    }
                    if (lane == 0) L.sigmask[cg] = (unsigned short)mask;
                    __syncthreads();
                }
            }
            hidden |= keep << cg;
            first_cg = 0;
        }
#undef RATE2
    }
    __syncthreads();
    for (int k = lane; k < NN; k += 64) lvl_g[d.off + k] = L.lvl[k];
    if (lane < ncg) sigmask_g[(long)ti * 64 + lane] = L.sigmask[lane];
    if (lane == 0) { out[2 * ti] = nz; out[2 * ti + 1] = best_last - 1; hidden_g[ti] = hidden; }
#undef SQ
#undef RATE
}

}  // namespace

extern "C" int ks265_rdoq_batch(ks265_ctx *ctx, const ks265_rdoq_tu *dev_tus, int n, int16_t *dev_lvl, const int16_t *dev_coef, const int32_t *dev_tables, uint16_t *dev_sigmask,
                                int32_t *dev_out, uint64_t *dev_hidden)
{
    if (!ctx || !dev_tus || !dev_lvl || !dev_coef || !dev_tables || !dev_sigmask || !dev_out || !dev_hidden) return KS265_POINTER;
    if (n <= 0) return n == 0 ? KS265_OK : KS265_NOTSUPPORTED;
    ks_use_device(ctx);
    hipLaunchKernelGGL(rdoq_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, dev_tus, n, (short *)dev_lvl, (const short *)dev_coef, (const int *)dev_tables, (unsigned short *)dev_sigmask,
                       (int *)dev_out, (unsigned long long *)dev_hidden, (const int *)nullptr);
    return ks265_check_launch(ctx);
}

// the pixel path's form (frame_recon.hip, cfg.rdoq): the transform blocks were listed on the device - one work-group per list slot up to max_n, those past *dev_n leave at once
int ks265_rdoq_listed(ks265_ctx *ctx, const ks265_rdoq_tu *dev_tus, const int *dev_n, int max_n, int16_t *dev_lvl, const int16_t *dev_coef, const int32_t *dev_tables, uint16_t *dev_sigmask,
                      int32_t *dev_out, uint64_t *dev_hidden)
{
    if (!ctx || !dev_tus || !dev_n || !dev_lvl || !dev_coef || !dev_tables || !dev_sigmask || !dev_out || !dev_hidden) return KS265_POINTER;
    if (max_n <= 0) return KS265_OK;
    hipLaunchKernelGGL(rdoq_kernel, dim3((unsigned)max_n), dim3(64), 0, ctx->stream, dev_tus, max_n, (short *)dev_lvl, (const short *)dev_coef, (const int *)dev_tables, (unsigned short *)dev_sigmask,
                       (int *)dev_out, (unsigned long long *)dev_hidden, dev_n);
    return ks265_check_launch(ctx);
}
