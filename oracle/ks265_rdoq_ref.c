/* ks265_rdoq_ref.c — TEST INFRASTRUCTURE (oracle), NOT PRODUCT CODE.
 *
 * h265_codec::rdoQuant enc@0x4aac50 of /root/reference/ubuntu_x64/appencoder (v2.6.1.3, binary only) restated from its disassembly: the reference's rate-distortion
 * optimised quantisation (HM-lineage xRateDistOptQuant on ALREADY QUANTISED levels, 64-bit fixed-point costs).  Pinned by tests/test_rdoq.py on calls recorded inside
 * real `appencoder` runs (tests/golden/rdoq.npz, oracle/ref_probe/gen_rdoq_traces.py).  SURVEY.md 8(f) rank 3; VERDICT r3 next-7.
 *
 * What the function does with one transform block (levels lvl = H265QuantBlock's output, coefficients coef, both N x N, stride N):
 *   - costs are int64: distortion = (|coef| - ((level * dq + add) >> shift))^2 << (2 log2 N + 1) with dq = QuantParam+0xc, shift = log2 N - 1,
 *     add = shift > per ? 1 << (shift - per - 1) : 0 (per = QuantParam+0x14; NOT the decoder's rounding); rate = (bits of the tables estBitRdoq enc@0x46a8a0 built, in
 *     units of 1 / 32768 bit) x lambda >> 8, lambda = (int64)(cfg multiplier x lambda table[qp] + 0.5);
 *   - coefficient groups from the last significant one down; per coefficient (reverse scan) the candidates are the quantised level q, q - 1 and - for q <= 2 - zero
 *     (the last position: never zero by this step); contexts c1 / c2 / Rice parameter as the entropy coder will see them;
 *   - per group: the all-zero group is taken if cheaper (coded_sub_block_flag rate, the significance flags it saves);
 *   - the last position: walking down from the last non-zero level while levels are <= 1, the cheapest truncation (x / y prefix + suffix bits);
 *   - signs from the coefficients, everything behind the chosen last position zeroed;
 *   - sign-data hiding (cfg+0x3e0) with its own rate-distortion choice of the coefficient to change (lambda from the second multiplier). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ks265_rdoq_ref.h"

/* normative tables (H.265 9.3.4.2.5 / 6.5.3 / 9.3.3.12): significance context of a position inside its 4x4 sub-block for the four neighbour patterns and for 4x4 blocks,
 * the three scans of a 4x4 sub-block as y * 4 + x, last-position group index (low nibble) and suffix bits (high nibble) */
static const uint8_t kSigCtx[5][16] = {{2, 1, 1, 0, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0}, {2, 2, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0}, {2, 1, 0, 0, 2, 1, 0, 0, 2, 1, 0, 0, 2, 1, 0, 0},
                                       {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2}, {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8}};
static const uint8_t kScan4[3][16] = {{0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15}};
static const uint8_t kLastGrp[32] = {0, 1, 2, 3, 0x14, 0x14, 0x15, 0x15, 0x26, 0x26, 0x26, 0x26, 0x27, 0x27, 0x27, 0x27, 0x38, 0x38, 0x38, 0x38, 0x38, 0x38, 0x38, 0x38, 0x39, 0x39, 0x39, 0x39, 0x39, 0x39, 0x39, 0x39};

/* scan position -> raster position of the block, and sub-block scan index -> sub-block raster index (the tables behind enc@0x6ff3c0 / 0x6ff360) */
static void build_scan(int log2, int scan_idx, int16_t *S /* N*N */, int16_t *G /* N*N/16 */)
{
    const int N = 1 << log2, w = N >> 2, ncg = w * w;
    int n = 0;
    if (scan_idx == 0) {                                  /* up-right diagonal over the sub-blocks */
        for (int d = 0; n < ncg; ++d)
            for (int y = d; y >= 0; --y) { const int x = d - y; if (x < w && y < w) G[n++] = (int16_t)(y * w + x); }
    } else if (scan_idx == 1) { for (int i = 0; i < ncg; ++i) G[i] = (int16_t)i; }
    else { for (int x = 0; x < w; ++x) for (int y = 0; y < w; ++y) G[n++] = (int16_t)(y * w + x); }
    for (int c = 0; c < ncg; ++c) {
        const int cx = G[c] % w, cy = G[c] / w;
        for (int k = 0; k < 16; ++k) { const int p = kScan4[scan_idx][k]; S[c * 16 + k] = (int16_t)((cy * 4 + (p >> 2)) * N + cx * 4 + (p & 3)); }
    }
}

/* bits (x 32768) of coeff_abs_level_remaining for `sym` at Rice parameter r as the function counts them (enc@0x4ab741.. / 0x4abea0) */
static int remain_bits(unsigned sym, int r)
{
    const unsigned p = sym >> r;
    if (p <= 2) return (int)((unsigned)r + p + 1) << 15;
    { int lg = 0; if (p != 3) { unsigned v = p - 2; while (v >>= 1) ++lg; } return (r + 2 * lg + 4) << 15; }
}
/* the variant of the sign-hiding bookkeeping (enc@0x4abc4b / 0x4abec8..): escape length capped at 8 prefix bits, values beyond a threshold in two parts */
static int remain_bits_cap(unsigned sym, int r) { unsigned v = (sym >> r) + (unsigned)r + 1; if (v > 8) v = 8; return (int)(v << 15); }
static int remain_bits_thr(unsigned sym, int r, unsigned thr)
{
    int extra = 0;
    if (thr < sym) { unsigned d = sym - thr; int lg = 0; while (d >>= 1) ++lg; extra = (2 * lg + 1) << 15; sym = thr + 1; }
    return extra + remain_bits_cap(sym, r);
}

typedef struct { int64_t coded, uncoded; int nnz_before0; } cg_stats;

int kso_ref_rdo_quant(int16_t *lvl, const int16_t *coef, int log2, int scan_idx, int comp, int dq, int per, int64_t lam, int64_t lam_sdh, const int32_t *T,
                      int tu5, int last_pos, uint16_t *sigmask, int flag_a4c0, int sdh, int32_t *out_last, uint64_t *out_cgmask)
{
    const int N = 1 << log2, luma = comp == 0, w = N >> 2, ncg = w * w;
    const int shift = log2 - 1, add = shift > per ? 1 << (shift - per - 1) : 0, esh = 2 * log2 + 1;
    static const uint8_t kRiceThr[5] = {7, 14, 26, 46, 78};
    int16_t S[1024], G[64];
    int64_t *cost_coeff = calloc(1024 * 3 + 64, sizeof(int64_t)), *cost_sig = cost_coeff + 1024, *cost0 = cost_sig + 1024, *cost_cgsig = cost0 + 1024;
    int32_t *inc_up = calloc(1024 * 3, sizeof(int32_t)), *inc_down = inc_up + 1024, *sig_delta = inc_down + 1024;
    build_scan(log2, scan_idx, S, G);
    const int last_cg = last_pos >> 4;
    int64_t uncoded = 0, base = 0;
    uint64_t cgmask = 0;
    int c1 = 1;
    const int sig_off = log2 == 2 ? 0 : log2 == 3 ? ((luma && scan_idx) ? 15 : 9) : (luma ? 21 : 12);
    #define SQ(c) ((int64_t)(c) * (c) << esh)
    #define RATE(bits) (((int64_t)(bits) * lam) >> 8)
    /* sub-blocks behind the last significant one: uncoded */
    for (int cg = last_cg + 1; cg < ncg; ++cg)
        for (int k = 0; k < 16; ++k) { const int blk = S[cg * 16] + (k >> 2) * N + (k & 3); const int64_t e = SQ(coef[blk]); cost0[blk] = e; uncoded += e; base += e; }
    for (int cg = last_cg; cg >= 0; --cg) {
        const int cgpos = G[cg], cgx = cgpos % w, cgy = cgpos / w;
        const int right = cgx + 1 < w ? (int)((cgmask >> (cgpos + 1)) & 1) : 0, lower = cgy + 1 < w ? (int)((cgmask >> (cgpos + w)) & 1) : 0;
        const int pattern = log2 == 2 ? 4 : right + 2 * lower;
        int ctx_set = (luma && cg != 0) ? 2 : 0;
        if (c1 == 0) ++ctx_set;
        const int sig_base = sig_off + ((luma && cg != 0) ? 3 : 0);
        uint16_t mask = sigmask[cg];
        if (cg != 0 && mask == 0) {                              /* enc@0x4ab0c8: nothing quantised to non-zero here */
            for (int k = 0; k < 16; ++k) {
                const int blk = S[cg * 16] + (k >> 2) * N + (k & 3);
                const int64_t e = SQ(coef[blk]);
                uncoded += e; base += e; cost0[blk] = e; cost_coeff[cg * 16 + k] = e;
                const int ctx = kSigCtx[pattern][kScan4[scan_idx][k]] + sig_base;
                sig_delta[blk] = T[46 + ctx] - T[4 + ctx];
                cost_sig[cg * 16 + k] = RATE(T[4 + ctx]);
            }
            c1 = 1;
            cost_cgsig[cg] = RATE(T[2 * ((right | lower) & 1)]);
            base += cost_cgsig[cg];
            continue;
        }
        /* enc@0x4ab310: the sub-block's coefficients in reverse scan order */
        int c1idx = 0, c2idx = 0, c2 = 0, rice = 0;
        int64_t sig_sum = 0, sig_cost0 = 0;
        cg_stats st = {0, 0, 0};
        c1 = 1;
        for (int p = 15; p >= 0; --p) {
            const int sp = cg * 16 + p, blk = S[sp];
            const int c = coef[blk];
            const int64_t e0 = SQ(c);
            uncoded += e0; cost0[blk] = e0;
            const int ctx_sig = blk == 0 ? 0 : kSigCtx[pattern][kScan4[scan_idx][p]] + sig_base;
            if (sp > last_pos) { base += e0; cost_coeff[sp] = 0; cost_sig[sp] = 0; continue; }
            const int bit = 15 - p;
            const int32_t *g1 = &T[108 + 2 * (c1 + 4 * ctx_set)];
            int chosen;
            if (!((mask >> bit) & 1)) {                              /* quantised to zero: stays zero */
                cost_sig[sp] = RATE(T[4 + ctx_sig]);
                cost_coeff[sp] = e0 + cost_sig[sp];
                base += cost_coeff[sp];
                sig_delta[blk] = T[46 + ctx_sig] - T[4 + ctx_sig];
                inc_up[blk] = g1[0];
                chosen = 0;
                sig_sum += cost_sig[sp]; if (p == 0) sig_cost0 = cost_sig[sp];
                continue;
            }
            const int q = lvl[blk] < 0 ? -lvl[blk] : lvl[blk];
            const int fl = (c1idx < 8 ? 1 : 0) + (c2idx == 0 ? 2 : 0);      /* bit 0: a greater1 flag is coded for this one, bit 1: a greater2 flag could be */
            const int base_level = (0xd9 >> (2 * fl)) & 3;
            const int32_t *g2 = &T[156 + 2 * (c2 + ctx_set)];
            const int ac = c < 0 ? -c : c;
            int64_t best = INT64_MAX;
            int sig1 = 0;
            cost_coeff[sp] = INT64_MAX;
            if (sp == last_pos) { sig_delta[blk] = 0; }
            else {
                if (q <= 2) { cost_sig[sp] = RATE(T[4 + ctx_sig]); cost_coeff[sp] = cost_sig[sp] + e0; best = cost_coeff[sp]; }
                sig1 = T[46 + ctx_sig];
                sig_delta[blk] = sig1 - T[4 + ctx_sig];
            }
            chosen = 0;
            {
                /* rate of the level's magnitude beyond its significance flag (enc@0x4ab730..0x4ab7e8), for q and q - 1 */
                int rq[2];
                for (int k = 0; k < 2; ++k) {
                    const int v = q - k;
                    if (v == 0) { rq[k] = 0; continue; }
                    int r;
                    if (v < base_level) r = v == 2 ? g2[0] + g1[1] : g1[0];
                    else {
                        r = remain_bits((unsigned)(v - base_level), rice);
                        if (fl & 1) r += g1[1];
                        if (fl == 3) r += g2[1];
                    }
                    rq[k] = r;
                }
                for (int k = 0; k < 2; ++k) {
                    const int v = q - k;
                    if (v == 0) break;                               /* (q == 1: the alternative is zero, handled above) */
                    const int rec = (int)((unsigned)(dq * v + add) >> shift);
                    const int64_t d = ac - rec;
                    const int64_t cost = (d * d << esh) + RATE((int64_t)sig1 + 0x8000 + rq[k]);
                    if (cost < best) { best = cost; cost_coeff[sp] = cost; cost_sig[sp] = RATE(sig1); chosen = v; }
                }
            }
            if (chosen == 0) mask &= (uint16_t)~(1u << bit);
            lvl[blk] = (int16_t)chosen;
            base += best;
            /* bookkeeping for the sign-hiding step: what one more / one less would cost in rate */
            inc_down[blk] = 0; inc_up[blk] = g1[0];
            if (sdh && chosen) {
                const int d = chosen - base_level;
                const unsigned thr = kRiceThr[rice];
                int up, dn, cur;
                /* enc@0x4ab99c..0x4ac0e5: current rate, rate at + 1, rate at - 1 of the magnitude part */
                #define MAGRATE(v) ((v) == 0 ? 0 : (v) < base_level ? ((v) == 2 ? g2[0] + g1[1] : g1[0]) : remain_bits_thr((unsigned)((v) - base_level), rice, thr) + ((fl & 1) ? g1[1] : 0) + (fl == 3 ? g2[1] : 0))
                (void)d;
                cur = MAGRATE(chosen); up = MAGRATE(chosen + 1); dn = chosen == 1 ? 0 : MAGRATE(chosen - 1);
                inc_up[blk] = up - cur; inc_down[blk] = dn - cur;
                #undef MAGRATE
            }
            if (chosen >= base_level && rice <= 3 && chosen > (3 << rice)) ++rice;
            if (chosen) ++c1idx;
            if (chosen > 1) { ++c2idx; c1 = 0; if (c2 < 2) ++c2; }
            else if ((c1 == 1 || c1 == 2) && chosen) ++c1;
            sig_sum += cost_sig[sp]; if (p == 0) sig_cost0 = cost_sig[sp];
            if (chosen) { st.coded += cost_coeff[sp] - cost_sig[sp]; st.uncoded += cost0[blk]; cgmask |= 1ull << cgpos; st.nnz_before0 += p; }
        }
        cost_cgsig[cg] = 0;
        if (cg != last_cg && cg != 0) {
            const int ctx = (right | lower) & 1;
            if (!((cgmask >> cgpos) & 1)) { cost_cgsig[cg] = RATE(T[2 * ctx]); base += cost_cgsig[cg]; base -= sig_sum; }
            else {
                if (st.nnz_before0 == 0) { sig_sum -= sig_cost0; base -= sig_cost0; }
                const int64_t zero_cost = RATE(T[2 * ctx]) + base + st.uncoded - st.coded - sig_sum;
                const int64_t one = RATE(T[2 * ctx + 1]);
                base += one;
                if (zero_cost < base) {
                    cost_cgsig[cg] = RATE(T[2 * ctx]); base = zero_cost; cgmask &= ~(1ull << cgpos); mask = 0;
                    for (int k = 0; k < 16; ++k) lvl[S[cg * 16] + (k >> 2) * N + (k & 3)] = 0;
                } else cost_cgsig[cg] = one;
            }
        }
        sigmask[cg] = mask;
    }
    /* the coded-block flag of the block against "all zero" */
    int64_t best_cost, bc = base;
    {
        int i0, i1;
        if (!flag_a4c0 && luma && tu5 == 0) { i0 = 178; i1 = 179; }
        else { const int idx = luma ? (tu5 == 0) : (int)(int8_t)tu5; i0 = 168 + 2 * idx; i1 = 169 + 2 * idx; }
        best_cost = uncoded + RATE(T[i0]); bc += RATE(T[i1]);
    }
    int best_last = 0;
    if (last_cg >= 0) {
        int done = 0;
        for (int cg = last_cg; cg >= 0 && !done; --cg) {
            if (cg != last_cg && cg != 0) { bc -= cost_cgsig[cg]; if (!((cgmask >> G[cg]) & 1)) continue; }
            for (int k = 0; k < 16; ++k) {
                const int sp = cg * 16 + 15 - k;
                if (sp > last_pos) continue;
                const int blk = S[sp], l = lvl[blk];
                if (!l) { bc -= cost_sig[sp]; continue; }
                int px = blk & (N - 1), py = blk >> log2;
                if (scan_idx == 2) { const int t = px; px = py; py = t; }
                const int gx = kLastGrp[px], gy = kLastGrp[py];
                const int bits = ((gx >> 4) << 15) + T[88 + (gx & 15)] + ((gy >> 4) << 15) + T[98 + (gy & 15)];
                const int64_t tot = bc - cost_sig[sp] + RATE(bits);
                if (tot < best_cost) { best_cost = tot; best_last = sp + 1; sigmask[cg] &= (uint16_t)(0xffffu << k); }
                if (l > 1) { done = 1; break; }
                bc -= cost_coeff[sp]; bc += cost0[blk];
            }
        }
    }
    /* signs, and zeros behind the chosen last position */
    int nz = 0;
    for (int i = 0; i < best_last; ++i) { const int blk = S[i]; const int s = coef[blk] >> 15; if (lvl[blk]) ++nz; lvl[blk] = (int16_t)((lvl[blk] ^ s) - s); }
    {   /* enc@0x4ac500..0x4ac5e8: the rest of the sub-block that holds the new last position (up to the old last position), then every later sub-block that starts at or before it */
        const int e = (best_last | 15) < last_pos ? (best_last | 15) : last_pos;
        for (int i = best_last; i <= e; ++i) lvl[S[i]] = 0;
        for (int b = (best_last & ~15) + 16; b <= last_pos; b += 16) for (int k = 0; k < 16; ++k) lvl[S[b] + (k >> 2) * N + (k & 3)] = 0;
    }
    /* enc@0x4ac5ea..0x4acd3b: sign-data hiding with a rate-distortion choice of the level to move (the second lambda).  Levels carry their signs here. */
    uint64_t hidden = 0;
    if (sdh && nz > 1 && best_last > 0) {
        int first_cg = 1;                                        /* the sub-block that holds the last position: its candidates start at that position */
        const int rec1 = (add + dq) >> shift;
        for (int cg = (best_last - 1) >> 4; cg >= 0; --cg) {
            uint16_t mask = sigmask[cg];
            if (!mask) continue;
            const int16_t *Sc = &S[cg * 16];
            int last = -1, first = 16;
            for (int n = 15; n >= 0; --n) if (lvl[Sc[n]]) { last = n; break; }
            for (int n = 0; n < 16; ++n) if (lvl[Sc[n]]) { first = n; break; }
            if ((last & 0xffff) - first <= 3) { first_cg = 0; continue; }
            const int neg = lvl[Sc[first]] <= 0;
            int sum = 0;
            for (int n = first; n <= last; ++n) sum += lvl[Sc[n]];
            uint64_t keep = 1;
            if ((sum & 1) != neg) {
                int64_t bestc = INT64_MAX; int best_n = -1, best_blk = -1, best_delta = 0;
                for (int n = first_cg ? last : 15; n >= 0; --n) {
                    const int blk = Sc[n], lv = lvl[blk], cf = coef[blk];
                    const int ac = cf < 0 ? -cf : cf, al = lv < 0 ? -lv : lv;
                    const int64_t d0 = SQ(ac - ((add + al * dq) >> shift));
                    int64_t cost; int delta;
                    #define RATE2(bits) (((int64_t)(bits) * lam_sdh) >> 8)
                    if (lv == 0) {
                        if (first > n && (cf < 0) != neg) continue;                  /* would become the first level of the sub-block: its sign must be the one hidden now */
                        cost = SQ(ac - rec1) - d0 + RATE2(inc_up[blk] + sig_delta[blk] + 0x8000); delta = 1;
                    } else {
                        const int64_t up = SQ(ac - ((add + (al + 1) * dq) >> shift)) - d0 + RATE2(inc_up[blk]);
                        int64_t dn = SQ(ac - ((add + (al - 1) * dq) >> shift)) - d0;
                        if (al == 1) {
                            dn += RATE2(inc_down[blk] - (sig_delta[blk] + 0x8000));
                            if (last == n && first_cg) dn -= 0x20000;
                        } else dn += RATE2(inc_down[blk]);
                        if (dn > up) { cost = up; delta = 1; }
                        else { if (al == 1 && n == first) continue; cost = dn; delta = -1; }
                    }
                    #undef RATE2
                    if (cost < bestc) { bestc = cost; best_n = n; best_blk = blk; best_delta = delta; }
                }
                if (best_blk >= 0) {
                    const int old = lvl[best_blk];
                    if (old == 32767 || old == -32768) best_delta = -1;
                    if (old == 0) ++nz; else if (best_delta == -1 && (old == 1 || old == -1)) --nz;
                    const int nv = coef[best_blk] < 0 ? old - best_delta : old + best_delta;
                    lvl[best_blk] = (int16_t)nv;
                    if (old == 0) { mask |= (uint16_t)(1u << (15 - best_n)); sigmask[cg] = mask; }
                    if (nv == 0) {
                        mask &= (uint16_t)~(1u << (15 - best_n)); sigmask[cg] = mask;
                        if (best_n == last) {
                            int nl = 15; while (nl >= 0 && !((mask >> (15 - nl)) & 1)) --nl;       /* the sub-block's new last level */
                            keep = nl - first > 3;
                            if (best_blk == S[best_last - 1]) best_last -= best_n - nl;
                        }
                    }
                }
            }
            hidden |= keep << cg;
            first_cg = 0;
        }
    }
    if (out_last) *out_last = best_last - 1;
    if (out_cgmask) *out_cgmask = hidden;
    free(cost_coeff); free(inc_up);
    return nz;
}

/* what the quantiser's bookkeeping (scanSigFlags enc@0x4a9b00 lineage) hands to rdoQuant: per sub-block (scan order) the mask of non-zero levels (bit 15 - position in
 * the sub-block's scan) and the scan position of the last non-zero level (-1: none) */
int kso_rdoq_scan_flags(const int16_t *lvl, int log2, int scan_idx, uint16_t *sigmask /*N*N/16*/)
{
    const int N = 1 << log2;
    int16_t S[1024], G[64];
    int last = -1;
    build_scan(log2, scan_idx, S, G);
    for (int cg = 0; cg < N * N / 16; ++cg) {
        uint16_t m = 0;
        for (int k = 0; k < 16; ++k) if (lvl[S[cg * 16 + k]]) { m |= (uint16_t)(1u << (15 - k)); last = cg * 16 + k; }
        sigmask[cg] = m;
    }
    return last;
}

