/* ks265_subme_ref.c — TEST INFRASTRUCTURE (oracle), NOT PRODUCT CODE.
 *
 * The reference's SUB-PEL REFINEMENT CONTROL restated from the disassembly of /root/reference/ubuntu_x64/appencoder (v2.6.1.3, binary only):
 *   h265_codec::getMvResolution       enc@0x483ca0   does this PU get a sub-pel refinement at all (tME+0x3bc)
 *   h265_codec::subMeSquare           enc@0x4b5660   start cost, half step, quarter step, choice between the two AMVP predictors
 *   h265_codec::subMeHpel_RealInterp  enc@0x4b4e90   eight half-sample candidates; its return value gates the quarter step
 *   h265_codec::subMeQpel_8Sad_v{0,2}h{0,2}_RealInterp enc@0x4b2bc0 / 0x4b3360 / 0x4b3b80 / 0x4b43a0 (table g_SubMeQpel_8Sad_RealInterpFunc enc@0x6ff6a0,
 *                                                     index = (half step moved horizontally) + 2 (moved vertically))
 * Pinned by tests/test_subme.py on calls recorded inside real `appencoder` runs (tests/golden/subme.npz, oracle/ref_probe/gen_subme_traces.py).
 *
 * What the binary does (all four presets' worth of flags are fields of kso_subme):
 *  - measure = TPredUnit+0x40: sad (satdInter 0) or had (satdInter 1: veryslow / placebo); with tME+0x64 the start cost is first recomputed with it;
 *  - candidates are indexed 0..7 in raster order (hpel_x / hpel_y enc@0x4e61f1 / 0x4e61e9 = +-2, qpel_x / qpel_y enc@0x4e61e1 / 0x4e61d9 = +-1);
 *    every candidate's samples are the NORMATIVE ones (one-dimensional: 8-tap to 8 bit; two-dimensional: horizontal to 14 bit, vertical (sum + 2048) >> 12),
 *    cost = measure + rate, compared with a strict '<' against the running best, which starts at the integer search's cost;
 *  - half step: evaluation order 3, 4 (horizontal), 1, 6 (vertical), 0, 2, 5, 7 (diagonal); with cfg+0x580 (no preset sets it) the diagonals are
 *    skipped when none of the four won and otherwise restricted to the two next to the winner;
 *  - the function returns "cost surface not flat": max over the evaluated SADs (integer position included) minus the winner's SAD must exceed
 *    ((W * H * tME+0x3c4) >> 3) << tME+0x60, and 20 x the winner's SAD must not exceed (cfg+0x568 + 20) x TPredUnit+0x140 (32-bit arithmetic) unless
 *    cfg+0x568 is 0; when cfg+0x464 is non-zero (every preset below veryslow) a "flat" verdict skips the quarter step;
 *  - quarter step around the half-step winner (or the integer position): order 1, 6, 3, 0, 5, 4, 2, 7.  subme 2 ("square full") evaluates all eight;
 *    subme 1 ("fast") (a) evaluates only candidates that stay within +-2 quarter samples of the integer position (around a half-sample centre: the side
 *    towards the integer position) and (b) evaluates a diagonal only while the running winner of this step is one of its two neighbours
 *    (0: {1, 3}, 5: {3, 6}, 2: {4, 1}, 7: {4, 6});
 *  - finally the rate is re-expressed for the other AMVP predictor (table tME+0x10, index cost tME+0x2e0[i]) and the cheaper predictor index kept.
 *
 * One thing is NOT restated because it is not behaviour but a buffer overlap: when the half step moved RIGHT (hpel_x = +2) the quarter functions
 * v0h2 / v2h2 copy the 14-bit half-sample rows (H + 8 rows of 160 bytes) on top of the prediction buffers at TCtuCache+0x44b60 (two buffers of 0x2080
 * bytes) and then filter from there INTO one of those buffers; the result for candidates 1 and 6 then depends on the vector implementation's loop order.
 * `aliased` is set for those calls; the replay test counts them separately (tests/test_subme.py). */
#include <stdlib.h>
#include <string.h>
#include "ks265_oracle.h"
#include "ks265_subme_ref.h"

static const int8_t kSqX[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, kSqY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};

/* normative prediction of the W x H block at quarter offset (qx, qy) from ref0 */
static void pred_at(const kso_subme *s, int qx, int qy, uint8_t *dst /* stride 64 */)
{
    const int W = 1 << s->log2w, H = 1 << s->log2h, fx = qx & 3, fy = qy & 3;
    const uint8_t *src = s->ref0 + (long)(qy >> 2) * s->stride + (qx >> 2);
    if (!fx && !fy) { for (int y = 0; y < H; ++y) memcpy(dst + 64 * y, src + (long)y * s->stride, (size_t)W); return; }
    if (!fy) { ks265o_interp_luma_hor_8to8(dst, 64, src, s->stride, W, H, fx); return; }
    if (!fx) { ks265o_interp_luma_ver_8to8(dst, 64, src, s->stride, W, H, fy); return; }
    { int16_t tmp[(64 + 7) * 64];
      ks265o_interp_luma_hor_8to16(tmp, 64, src - 3L * s->stride, s->stride, W, H + 7, fx);
      ks265o_interp_luma_ver_16to8(dst, 64, tmp + 3 * 64, 64, W, H, fy); }
}
static uint32_t dist_at(const kso_subme *s, int qx, int qy)
{
    uint8_t p[64 * 64];
    pred_at(s, qx, qy, p);
    return s->dist(s->fenc, p, s->fstride, 64, 1 << s->log2h, 1 << s->log2w);
}
static uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

/* subMeHpel_RealInterp enc@0x4b4e90.  (cx, cy) = integer start in quarter samples relative to ref0 (always 0, 0 here); mvc = the eight rates */
static int hpel_step(kso_subme *s, const uint32_t mvc[8], uint32_t *best, int *idx, uint32_t rate0)
{
    const int W = 1 << s->log2w;
    uint32_t sad[8], maxsad = s->cost - rate0;                      /* [rsp+0x58]: the integer position's distortion */
    #define TRY(k) do { sad[k] = dist_at(s, 2 * kSqX[k], 2 * kSqY[k]); if (sad[k] + mvc[k] < *best) { *best = sad[k] + mvc[k]; *idx = (k); } maxsad = umax(maxsad, sad[k]); } while (0)
    TRY(3); TRY(4); TRY(1); TRY(6);
    if (!(s->hpel_diag_fast && *idx == -1)) {
        if (!s->hpel_diag_fast) { TRY(0); TRY(2); TRY(5); TRY(7); }
        else {                                                       /* enc@0x4b52c2..0x4b55e8: the two diagonals next to the running winner */
            if ((*idx & ~2) == 1) TRY(0);
            if (*idx == 1 || *idx == 4) TRY(2);
            if (*idx == 3 || *idx == 6) TRY(5);
            if ((*idx & ~2) == 4) TRY(7);
        }
    }
    #undef TRY
    {
        const int32_t thr = (int32_t)((((uint32_t)(W << s->log2h) * (uint32_t)s->flat_factor) >> 3) << s->flat_shift);   /* imul edx, tME+0x3c4; sar 3; shl cl */
        const uint32_t rate = *idx < 0 ? rate0 : mvc[*idx];
        const int32_t spread = (int32_t)(maxsad - *best + rate);
        if (thr >= spread) return 0;
        { const uint32_t sadbest = *best - rate;
          if ((uint32_t)(sadbest * 20u) <= (uint32_t)((uint32_t)(s->c568 + 20) * s->pu140)) return 1;
          return s->c568 == 0; }
    }
}

/* the four quarter-sample functions: (hx, hy) = the half step's move */
static void qpel_step(kso_subme *s, int cx, int cy, int hx, int hy, const uint32_t mvc[8], uint32_t *best, int *idx)
{
    const int fast = s->subme == 1;
    /* (a) sides: fast mode keeps the total offset from the integer position within +-2 */
    const int up = !fast || hy >= 0, down = !fast || hy <= 0, left = !fast || hx >= 0, right = !fast || hx <= 0;
    #define TRY(k) do { const uint32_t c = dist_at(s, cx + kSqX[k], cy + kSqY[k]) + mvc[k]; if (c < *best) { *best = c; *idx = (k); } } while (0)
    if (up) TRY(1);
    if (down) TRY(6);
    if (left) {
        TRY(3);
        if (up && (!fast || (*idx & ~2) == 1)) TRY(0);
        if (down && (!fast || *idx == 3 || *idx == 6)) TRY(5);
    }
    if (right) {
        TRY(4);
        if (up && (!fast || *idx == 4 || *idx == 1)) TRY(2);
        if (down && (!fast || (*idx & ~2) == 4)) TRY(7);
    }
    #undef TRY
}

void kso_ref_subme_square(kso_subme *s)
{
    int mx = s->mx, my = s->my;
    uint32_t best, r15 = 0;
    s->hpel_idx = s->qpel_idx = -1; s->qpel_ran = 0; s->aliased = 0;
    if (s->recost)                                                   /* enc@0x4b5c00: tME+0x64 */
        s->cost = s->dist(s->fenc, s->ref0, s->fstride, s->stride, 1 << s->log2h, 1 << s->log2w) + s->rate(s->rate_ctx, mx, my);
    best = s->cost;
    if (s->do_subpel) {
        uint32_t mvc[8]; int idx = -1, hx = 0, hy = 0, go;
        const uint32_t rate0 = s->rate(s->rate_ctx, mx, my);
        for (int k = 0; k < 8; ++k) mvc[k] = s->rate(s->rate_ctx, mx + 2 * kSqX[k], my + 2 * kSqY[k]);
        go = hpel_step(s, mvc, &best, &idx, rate0);
        s->hpel_idx = idx;
        if (idx >= 0) { hx = 2 * kSqX[idx]; hy = 2 * kSqY[idx]; r15 = mvc[idx]; mx += hx; my += hy; }
        if (!(s->mvres_thr != 0 && !go)) {                           /* enc@0x4b58e6: cfg+0x464 != 0 and a flat verdict -> no quarter step */
            idx = -1;
            for (int k = 0; k < 8; ++k) mvc[k] = s->rate(s->rate_ctx, mx + kSqX[k], my + kSqY[k]);
            s->qpel_ran = 1;
            s->aliased = hx == 2;
            qpel_step(s, hx, hy, hx, hy, mvc, &best, &idx);
            s->qpel_idx = idx;
            if (idx >= 0) { r15 = mvc[idx]; mx += kSqX[idx]; my += kSqY[idx]; }
        }
    }
    if (r15 == 0) r15 = s->rate(s->rate_ctx, mx, my);               /* enc@0x4b5a3a: test r15d, r15d; je 0x4b56e7 */
    s->mx = mx; s->my = my; s->cost = best; s->rate_out = r15; s->dist_out = best - r15;
}

/* getMvResolution enc@0x483ca0 */
int kso_ref_mv_resolution(int c498, int c49c, int log2w, int log2h, uint32_t cost, int me3c9, int me65, uint32_t rate0, int mvres_thr, int searched,
                          uint32_t c[4], int c8_21, int me3c0, int me60, const uint8_t *fenc, int fstride, const uint8_t *ref0, int stride)
{
    if (c498 && (uint32_t)(((6 - log2h) * c49c + c498) << (2 * log2w)) < cost) return 0;
    if (!me3c9 || me65) return 0;
    if (!mvres_thr) return 1;
    if (!searched) ks265o_sad4(fenc, ref0, fstride, stride, 1 << log2h, c, 1 << log2w);   /* enc@0x483e38: g_sad4_Function[log2w - 2] */
    {
        const int32_t thr = (int32_t)(((int32_t)(((1 << log2h) << log2w) * (c8_21 + me3c0)) >> 3) << me60);
        const uint32_t m01 = umax(c[0], c[1]), m23 = umax(c[2], c[3]);
        const uint32_t m = (m01 > m23 ? m01 : m23) >> 2;
        return (int32_t)(m - ((cost - rate0) << 2)) >= thr;
    }
}

/* ---------------------------------------------------------------- trace replay */
typedef struct { const uint16_t *cm; int mx0, my0; } rate_tab;
static uint32_t tab_rate(void *ctx, int qx, int qy)
{
    const rate_tab *t = ctx;
    const int dx = qx - t->mx0, dy = qy - t->my0;
    if (dx < -8 || dx > 8 || dy < -8 || dy > 8) abort();
    return (uint32_t)t->cm[8 + dx] + t->cm[25 + dy];
}
void kso_subme_replay(const int32_t *h, const uint8_t *fenc, const uint8_t *region, const uint16_t *cm, int32_t out[9])
{
    kso_subme s; rate_tab t = {cm, h[5], h[6]};
    const int W = h[3], H = h[4];
    int l2w = 0, l2h = 0;
    while ((1 << l2w) < W) ++l2w;
    while ((1 << l2h) < H) ++l2h;
    memset(&s, 0, sizeof s);
    s.fenc = fenc; s.fstride = W; s.stride = W + 16; s.ref0 = region + 8 * (W + 16) + 8; s.log2w = l2w; s.log2h = l2h;
    s.dist = h[26] ? ks265o_sad : ks265o_had; s.recost = h[8]; s.subme = h[11]; s.mvres_thr = h[12]; s.hpel_diag_fast = h[13]; s.c568 = h[14];
    s.flat_factor = h[15]; s.flat_shift = h[16]; s.pu140 = (uint32_t)h[17]; s.rate = tab_rate; s.rate_ctx = &t; s.do_subpel = h[10];
    s.mx = h[5]; s.my = h[6]; s.cost = (uint32_t)h[7];
    kso_ref_subme_square(&s);
    {   /* enc@0x4b5a4f..0x4b5b04: the cheaper of the two predictor indices */
        const int p = h[22], o = p ^ 1;
        uint32_t best = s.cost, r = s.rate_out + (uint32_t)h[23 + p];
        const uint32_t r2 = (uint32_t)h[23 + o] + (uint32_t)(o ? h[40] + h[41] : h[38] + h[39]);
        int pidx = p;
        best += (uint32_t)h[23 + p];
        if (r > r2) { pidx = o; best = best - r + r2; r = r2; }
        out[0] = s.mx; out[1] = s.my; out[2] = (int32_t)best; out[3] = (int32_t)r; out[4] = (int32_t)s.dist_out; out[5] = pidx;
    }
    out[6] = s.aliased; out[7] = s.hpel_idx; out[8] = s.qpel_idx;
}
int kso_mvres_replay(const int32_t *h, const uint8_t *fenc, const uint8_t *region, int32_t out[2])
{
    const int W = h[3], H = h[4];
    int l2w = 0, l2h = 0;
    uint32_t c[4] = {(uint32_t)h[14], (uint32_t)h[15], (uint32_t)h[16], (uint32_t)h[17]};
    while ((1 << l2w) < W) ++l2w;
    while ((1 << l2h) < H) ++l2h;
    out[0] = kso_ref_mv_resolution(h[8], h[9], l2w, l2h, (uint32_t)h[7], h[10], h[11], (uint32_t)h[21], h[12], h[13] || !fenc, c, h[18], h[19], h[20],
                                   fenc, W, region ? region + (W + 2) + 1 : NULL, W + 2);
    out[1] = (int32_t)c[0];
    return 0;
}
