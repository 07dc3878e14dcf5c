/* ks265_lookahead_ref.c - TEST INFRASTRUCTURE (oracle), NOT PRODUCT CODE.
 *
 * Three lookahead decisions of /root/reference/ubuntu_x64/appencoder (v2.6.1.3, binary only) restated from the disassembly:
 *   calcFrameAdaptQuant enc@0x4653c0, cuTreePropagate enc@0x47d460, scenecut enc@0x47e9d0 (its rule; the costs it compares come from calcFrameCost enc@0x4a7410).
 * Pinned by tests/test_lookahead_ref.py on calls recorded inside real `appencoder` runs (tests/golden/lookahead_ref.npz, oracle/ref_probe/gen_la_traces.py).
 * The arithmetic is x264-lineage (adaptive quantisation by block variance, macroblock-tree propagation, scene-cut bias) in the reference's own fixed-point form. */
#include <math.h>
#include <stdint.h>
#include "ks265_lookahead_ref.h"
#include "ks265_oracle.h"

/* _log2 enc@0x4c3c20: table[(x << lz >> 24) & 127] + (31 - lz); the table enc@0x4e6a40 holds log2(1 + i / 128) rounded to five decimals (checked against the file) */
double kso_ref_log2(uint32_t x)
{
    const int lz = __builtin_clz(x);
    const unsigned i = ((x << lz) >> 24) & 127u;
    return round(log2((128.0 + i) / 128.0) * 1e5) / 1e5 + (31 - lz);
}
/* qy265_exp2fix8 enc@0x4c3c50: i = (int)(x * (float)(-64 / 6) + 512.5); 0 below, 0xffff above 1023; else (lut[i & 63] + 256) << (i >> 6) >> 8 with
 * lut[k] = round((2^(k / 64) - 1) * 256) (enc@0x4e6e40, checked against the file) */
int kso_ref_exp2fix8(double x)
{
    const int i = (int)(x * (double)(-64.0f / 6.0f) + 512.5);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    const int lut = (int)lround((pow(2.0, (i & 63) / 64.0) - 1.0) * 256.0);
    return ((lut + 256) << (i >> 6)) >> 8;
}

void kso_ref_frame_adapt_quant(const uint8_t *Y, const uint8_t *U, const uint8_t *V, int nx, int ny, int count, double strength, double *qp_off, uint16_t *inv_qscale)
{
    double sum = 0.0;
    for (int by = 0; by < ny; ++by)
        for (int bx = 0; bx < nx; ++bx) {
            const uint32_t e = ks265o_ac_energy_plane(Y + (long)by * 16 * nx * 16 + bx * 16, nx * 16, 4) + ks265o_ac_energy_plane(U + (long)by * 8 * nx * 8 + bx * 8, nx * 8, 3) +
                               ks265o_ac_energy_plane(V + (long)by * 8 * nx * 8 + bx * 8, nx * 8, 3);
            const double l = kso_ref_log2(e + 2u), v = l * l;
            qp_off[by * nx + bx] = v; sum += v;
        }
    const double avg = sum / count, scale = strength * avg / 6000.0;
    for (int i = 0; i < nx * ny; ++i) {
        qp_off[i] = (qp_off[i] - avg) * scale;
        inv_qscale[i] = (uint16_t)kso_ref_exp2fix8(qp_off[i]);
    }
}

static void clip_add(uint16_t *p, int v) { const int s = *p + v; *p = (uint16_t)(s > 0xffff ? 0xffff : s); }

void kso_ref_cutree_propagate(int lg, int nx, int ny, const uint16_t *intra, const uint16_t *inv_qscale, const uint16_t *own, const uint16_t *inter, const uint8_t *list_bits,
                              const int32_t *mv0, const int32_t *mv1, uint16_t *ref0, uint16_t *ref1)
{
    const int sh = lg + 2, unit = 1 << sh, sh2 = 2 * sh, rnd = 1 << (sh2 - 1);
    const int32_t *mvs[2] = {mv0, mv1};
    uint16_t *refs[2] = {ref0, ref1};
    for (int by = 0; by < ny; ++by)
        for (int bx = 0; bx < nx; ++bx) {
            const int idx = by * nx + bx;
            /* what this block hands on: (its own cost, AQ-weighted, + what it has inherited) x the share the prediction explains (enc@0x47d5c8..0x47d5fb, 64-bit signed division) */
            if (intra[idx] == 0) continue;                                    /* the reference's idiv would trap; it never stores 0 (calcFrameCost keeps min(cost + 9, 0xffff)) - same guard as the device operator */
            const int64_t have = ((int)((unsigned)inv_qscale[idx] * (unsigned)intra[idx] + 128u) >> 8) + own[idx];   /* 32-bit product as the reference forms it */
            const int amt = (int)(have * ((int)intra[idx] - (int)inter[idx]) / (int64_t)intra[idx]);
            if (amt <= 0) continue;
            const int lists = (list_bits[idx >> 2] >> ((idx & 3) * 2)) & 3;
            for (int l = 0; l < 2; ++l) {
                if (!((lists >> l) & 1)) continue;
                int a = amt;
                if (lists == 3) a = (a * 32 + 32) >> 6;                                    /* two lists: half each (the weights at [rsp+8] / [rsp+0xc] are 32) */
                const int32_t mv = mvs[l][idx];
                uint16_t *R = refs[l];
                if (mv == 0) { clip_add(&R[idx], a); continue; }
                const int mvx = (int16_t)(mv & 0xffff), mvy = mv >> 16;
                const int x = (mvx >> sh) + bx, y = (mvy >> sh) + by;
                const int xf = (int16_t)(mvx & (unit - 1)), yf = (int16_t)(mvy & (unit - 1));
                const int w0 = (unit - yf) * (unit - xf), w1 = (unit - yf) * xf, w2 = yf * (unit - xf), w3 = yf * xf;
                const int i0 = y * nx + x;
                #define SHARE(w) (((w) * a + rnd) >> sh2)
                if (x < nx - 1 && y < ny - 1 && x >= 0 && y >= 0) {
                    clip_add(&R[i0], SHARE(w0)); clip_add(&R[i0 + 1], SHARE(w1)); clip_add(&R[i0 + nx], SHARE(w2)); clip_add(&R[i0 + nx + 1], SHARE(w3));
                } else {
                    if (x < nx && y < ny && x >= 0 && y >= 0) clip_add(&R[i0], SHARE(w0));
                    if (x + 1 < nx && y < ny && x >= -1 && y >= 0) clip_add(&R[i0 + 1], SHARE(w1));
                    if (x < nx && y + 1 < ny && x >= 0 && y >= -1) clip_add(&R[i0 + nx], SHARE(w2));
                    if (x + 1 < nx && y + 1 < ny && x >= -1 && y >= -1) clip_add(&R[i0 + nx + 1], SHARE(w3));
                }
                #undef SHARE
            }
        }
}

int kso_ref_scenecut(int pcost, int icost, int prev_icost, int blocks, int lg, int thr, int keyint, int poc, int last_key)
{
    if (prev_icost != -1) {                                           /* flat pictures (intra cost below 2^(2 lg - 4) per block): a change of flatness is the verdict */
        const int T = (int)((unsigned)blocks << (2 * lg - 4));
        if (prev_icost < T) { if (icost > T) return 1; if (icost < T) return 0; }
        else if (prev_icost > T) { if (icost < T) return 1; }
    }
    const double bias = (double)(poc - last_key) * ((double)thr / 100.0) / (double)(keyint < 320 ? keyint : 320);
    return (double)pcost >= (1.0 - bias) * (double)icost;
}

/* this build's own rule (not the reference's): the QP of every CTU from the 16x16 blocks' offsets - base + round(mean over the CTU's blocks, summed in raster order), clipped */
void kso_aq_ctu_map(const double *off, int nx, int ny, int base_qp, int lo, int hi, int8_t *map)
{
    const int cols = (nx + 3) / 4, rows = (ny + 3) / 4;
    for (int cy = 0; cy < rows; ++cy)
        for (int cx = 0; cx < cols; ++cx) {
            double sum = 0.0; int cnt = 0;
            for (int by = cy * 4; by < (cy * 4 + 4 < ny ? cy * 4 + 4 : ny); ++by)
                for (int bx = cx * 4; bx < (cx * 4 + 4 < nx ? cx * 4 + 4 : nx); ++bx) { sum += off[by * nx + bx]; ++cnt; }
            int d = (int)floor(sum / (double)cnt + 0.5);
            d = d < -12 ? -12 : d > 12 ? 12 : d;                 /* CuQpDeltaVal between two CTUs stays inside [-26, 25] */
            const int q = base_qp + d;
            map[cy * cols + cx] = (int8_t)(q < lo ? lo : q > hi ? hi : q);
        }
}


/* ---- calcFrameCost enc@0x4a7410 -------------------------------------------------------------------------------------------------------------------------------------------------
 * Restated from the disassembly; pinned by tests/test_calc_frame_cost.py on calls recorded inside real `appencoder` runs (oracle/ref_probe/cfc_shim.c, gen_cfc_traces.py,
 * tests/golden/calc_frame_cost.npz).  Blocks are visited from the last to the first (rows bottom-up, columns right to left: enc@0x4a7862..0x4a8a16), so a block's vector
 * predictors are its right, lower, lower-left and lower-right neighbours' vectors of THIS call (enc@0x4a7dc0..0x4a7e4a; the first two that exist become the AMVP pair of
 * meInitPoint enc@0x48af50, which runs without CTU / CU objects: no look-ahead vector, no stored candidates).  Costs: list cost + 4; bi-predictive average
 * (g_weightBiSadFunc enc@0x707af0) + 9 if its SAD + 5 beats the better list; intra = best of {planar, DC, 26, 10, 18, 2, 34} refined by +-2, +-1, + 9. */
#include <string.h>
#include "ks265_me_ref.h"

typedef struct { const uint8_t *fenc, *plane; int stride; int bs; kso_cfc *c; int px, py; } cfc_dist;
static int cfc_in(const kso_cfc *c, int x, int y, int bs)            /* block at picture position (x, y) readable? */
{
    return x >= -c->margin_x && y >= -c->margin_y && x + bs <= c->w + c->margin_x && y + bs <= c->h + c->margin_y;
}
static uint32_t cfc_dist_fn(void *user, long off)
{
    cfc_dist *g = (cfc_dist *)user;
    int yy = (int)(off >= 0 ? off / g->stride : -((-off + g->stride - 1) / g->stride)), xx = (int)(off - (long)yy * g->stride);
    if (xx >= g->c->w + g->c->margin_x) { xx -= g->stride; yy += 1; }      /* a position left of the picture comes out as a column of the row before */
    if (!cfc_in(g->c, xx, yy, g->bs)) { g->c->oob = 1; return 0; }
    return ks265o_sad(g->fenc, g->plane + off, g->stride, g->stride, g->bs, g->bs);
}
static const uint8_t kCfcNeedFilter[4][35] = {                           /* g_intraNeedFilter enc@0x4df3a0 (read from the file): rows log2 2..5 */
    {0},
    {1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1}};
/* IntraPredLoadRefLeftTopAvaible enc@0x423b20 (g_IntraPredLoadRefFunction[6]): corner, top row, left column from the picture; top-right / lower-left repeat the last sample */
static void cfc_load_ref(const uint8_t *fenc, int stride, int bs, uint8_t *ref)
{
    ref[0] = fenc[-stride - 1];
    for (int i = 0; i < bs; ++i) { ref[1 + i] = fenc[-stride + i]; ref[1 + bs + i] = fenc[-stride + bs - 1]; }
    for (int i = 0; i < bs; ++i) { ref[-1 - i] = fenc[(long)i * stride - 1]; ref[-1 - bs - i] = fenc[(long)(bs - 1) * stride - 1]; }
}
typedef struct { const uint8_t *fenc; int stride, lg, fast; const uint8_t *unf, *fil; } cfc_intra;
static uint32_t cfc_intra_sad(const cfc_intra *t, int mode)
{
    uint8_t pred[32 * 32];
    const int bs = 1 << t->lg;
    const uint8_t *ref = (!t->fast && kCfcNeedFilter[t->lg - 2][mode]) ? t->fil : t->unf;       /* enc@0x4a8540..0x4a8561 */
    ks265o_intra_pred(pred, bs, ref, mode, t->lg, 1);
    return ks265o_sad(t->fenc, pred, t->stride, bs, bs, bs);
}
static inline int cfc_abs(int v) { return v < 0 ? -v : v; }

void kso_ref_calc_frame_cost(kso_cfc *c)
{
    const int idx = c->d0 * 9 + c->d1, lg = c->lg, bs = 1 << lg, nx = c->nx, ny = c->ny, stride = c->stride;
    int32_t *s684 = idx ? &c->sum : &c->sum_intra, *s7c8 = idx ? &c->sum_aq : &c->sum_intra_aq;       /* slot idx; slot 0 is where the intra pass accumulates */
    c->oob = c->table_oob = 0;
    if (*s684 < 0) {
        if (c->do_list[0]) c->mv[0][0] = 0;                                                              /* enc@0x4a7576..0x4a75b9 */
        if (c->do_list[1]) c->mv[1][0] = 0;
        if (c->d1 == 0 && !c->intra_done) { c->intra_wins = 0; c->sum_intra = -1; c->sum_intra_aq = -1; }  /* enc@0x4a8084..0x4a80b3 */
        int mer = c->merange >> 1; if (mer > 32) mer = 32;                                               /* enc@0x4a75fb..0x4a7609 */
        static const int kTbl[9] = {2, 2, 3, 3, 4, 4, 6, 6, 8};
        const int movthr = c->d0 <= 8 ? (kTbl[c->d0] * 12) >> 1 : 48;                                     /* enc@0x4a8065..0x4a807b */
        double t = (double)((c->w + c->h) * 2) / 656.0;                                                   /* enc@0x4a77ba..0x4a7842 */
        if (t >= 2.0) t = t * t * 0.5; else if (c->preset > 4) t = t * t * 0.75;
        const int bigthr = (int)(t * 4.0);
        /* the mvd cost table tME+0x10 (enc@0x4a7622..0x4a7663): the centre of row 12 of a u16 table [52][m], m = 8 merange + 33, row q = lambda(q) x bits(d), |d| <= m / 2
         * (createMvdCostTable enc@0x48b850).  interMeDia indexes it without a range test: far from the predictor it reads the neighbouring rows - restated as the reference lays them out */
        enum { TH = 1024 };
        static _Thread_local uint16_t tab[2 * TH + 1];
        const int m_row = 8 * c->merange + 33, tab_half = m_row >> 1;
        for (int d = -TH; d <= TH; ++d) {
            const long f = 12L * m_row + tab_half + d;
            const int row = (int)(f / m_row), col = (int)(f % m_row);
            tab[d + TH] = (f >= 0 && row < 52) ? (uint16_t)(c->lambda_tab[row] * kso_mvd_bits(col - tab_half)) : 0xffff;
        }
        const uint16_t *base = tab + TH;
        const int lambda = c->lambda_tab[12];
        int cnt30 = 0, cnt78 = 0, cnt4c = 0, cnt40 = 0, sum7c = 0, sum88 = 0;
        const int any_inter = (c->d0 + c->d1) != 0, has1 = c->d1 > 0;
        const uint8_t *planes[2] = {c->ref0, c->ref1};
        for (int by = ny - 1; by >= 0; --by)
            for (int bx = nx - 1; bx >= 0; --bx) {
                const int px = bx << lg, py = by << lg, blk = by * nx + bx, sh = (blk & 3) * 2;
                const uint8_t *fenc = c->cur + (long)py * stride + px;
                int counted = 1;                                                                          /* enc@0x4a7c47..0x4a7ca2 */
                if (!(bx > 0 && bx < nx - 1 && by > 0 && by < ny - 1)) counted = nx <= 2 ? 1 : ny <= 2;
                const int thr = (int)((uint32_t)c->zero_thr << (2 * lg)) >> 1;                            /* enc@0x4a7caa..0x4a7cd0 */
                int mvq[2][2] = {{0, 0}, {0, 0}};                                                         /* tME+0x54 / +0x56 of the two lists, quarter pel at the end */
                uint32_t bcost; int bitsv;
                if (!any_inter) { bcost = 0x10000003u; bitsv = 0; }
                else {
                    bcost = 0xfffffffu; int bestbits = 0;
                    for (int l = 0; l <= has1; ++l) {
                        uint32_t lcost;
                        if (!c->do_list[l]) {                                                             /* enc@0x4a8051 */
                            const int32_t v = c->mv[l][blk];
                            mvq[l][0] = (int16_t)(v & 0xffff); mvq[l][1] = v >> 16; lcost = (uint32_t)c->cost[l][blk];
                        } else {
                            const int32_t *mvp = c->mv[l] + blk;
                            int32_t cand[4] = {0, 0, 0, 0}; int n = 0;
                            if (bx < nx - 1) cand[n++] = mvp[1];
                            if (by < ny - 1) {
                                cand[n++] = mvp[nx];
                                if (bx > 0) cand[n++] = mvp[nx - 1];
                                if (bx < nx - 1) cand[n++] = mvp[nx + 1];
                            }
                            cfc_dist g = {fenc, planes[l], stride, bs, c, px, py};
                            kso_me_init m; memset(&m, 0, sizeof m);
                            m.log2w = m.log2h = lg; m.pux = px; m.puy = py; m.stride = stride;
                            for (int k = 0; k < 2 && k < n; ++k) { m.mvp[k][0] = (int16_t)(cand[k] & 0xffff); m.mvp[k][1] = cand[k] >> 16; }
                            m.lim[0] = (int16_t)-px; m.lim[1] = (int16_t)(c->w - px - bs); m.lim[2] = (int16_t)-py; m.lim[3] = (int16_t)(c->h - py - bs);
                            m.merange = mer; m.lambda = lambda; m.idx_cost[0] = 0; m.idx_cost[1] = 1;
                            m.base = base; m.dist = cfc_dist_fn; m.user = &g;
                            kso_ref_me_init_point(&m);
                            if (cfc_abs(m.cmx_off) + 4 * cfc_abs(m.mx) > tab_half || cfc_abs(m.cmy_off) + 4 * cfc_abs(m.my) > TH) c->table_oob |= 1;
                            int mx = m.mx, my = m.my; uint32_t cost = m.cost;
                            if (!((uint64_t)m.sad < (uint64_t)(int64_t)thr)) {                            /* enc@0x4a7f30..0x4a7f50 */
                                if (!m.zero_tried) {                                                      /* enc@0x4a8198..0x4a82db: the zero vector against the start point */
                                    const int pxq = m.mvp[m.mvp_idx][0], pyq = m.mvp[m.mvp_idx][1];
                                    const uint32_t s0 = cfc_dist_fn(&g, (long)(py * stride) + px);
                                    const uint32_t c0 = s0 + kso_ref_mvd_cost_far(base, lambda, -pyq) + kso_ref_mvd_cost_far(base, lambda, -pxq);
                                    if (c0 < cost) { cost = c0; mx = my = 0; }
                                }
                                kso_me d; memset(&d, 0, sizeof d);
                                d.fenc = fenc; d.fstride = stride; d.ref0 = planes[l] + (long)py * stride + px; d.stride = stride; d.log2w = d.log2h = lg;
                                d.cmx = base + m.cmx_off; d.cmy = base + m.cmy_off; d.merange = mer;
                                d.range_shift = c->preset > 1 ? 0 : (c->p8 != 4);                        /* enc@0x4a7f5c..0x4a7f8b, 0x4a82f0 */
                                d.dist = ks265o_sad; d.mx = mx; d.my = my; d.cost = cost;
                                d.chk = 1; d.gx0 = -c->margin_x - px; d.gx1 = c->w + c->margin_x - bs - px; d.gy0 = -c->margin_y - py; d.gy1 = c->h + c->margin_y - bs - py;
                                kso_ref_me_dia(&d);
                                if (d.oob) c->oob = 1;
                                mx = d.mx; my = d.my; cost = d.cost;
                                if (cfc_abs(m.cmx_off) + 4 * (cfc_abs(mx) + 1) > tab_half || cfc_abs(m.cmy_off) + 4 * (cfc_abs(my) + 1) > TH) c->table_oob |= 2;
                            }
                            mvq[l][0] = (int16_t)(mx << 2); mvq[l][1] = (int16_t)(my << 2);              /* enc@0x4a7faf..0x4a7fc5 */
                            c->mv[l][blk] = (int32_t)((uint32_t)(uint16_t)mvq[l][0] | ((uint32_t)(uint16_t)mvq[l][1] << 16));
                            c->cost[l][blk] = (int32_t)cost; lcost = cost;
                        }
                        if (bcost > lcost) { bcost = lcost; bestbits = l + 1; }                           /* enc@0x4a7fc8..0x4a7fef */
                    }
                    if (has1) {                                                                           /* enc@0x4a8910..0x4a8a01 */
                        const int x0 = px + (mvq[0][0] >> 2), y0 = py + (mvq[0][1] >> 2), x1 = px + (mvq[1][0] >> 2), y1 = py + (mvq[1][1] >> 2);
                        uint32_t sb = 0;
                        if (!cfc_in(c, x0, y0, bs) || !cfc_in(c, x1, y1, bs)) c->oob = 1;
                        else sb = ks265o_weight_bi_sad(fenc, (unsigned)stride, c->ref0 + (long)y0 * stride + x0, c->ref1 + (long)y1 * stride + x1, (unsigned)stride, (unsigned)stride, bs, bs);
                        if (sb + 5 >= bcost) { bcost += 4; bitsv = bestbits; } else { bcost = sb + 9; bitsv = 3; }
                    } else { bcost += 4; bitsv = bestbits; }                                              /* enc@0x4a8161..0x4a818f */
                }
                c->bits[blk >> 2] = (uint8_t)((c->bits[blk >> 2] & ~(3 << sh)) | (bitsv << sh));          /* enc@0x4a7930..0x4a7963 */
                if (!(has1 && !c->b_intra)) {                                                             /* enc@0x4a7966..0x4a79b3 */
                    uint32_t icost;
                    if (!c->intra_done) {                                                                 /* enc@0x4a8368..0x4a889f */
                        uint8_t unf_[4 * 32 + 1 + 32], fil_[4 * 32 + 1 + 32];
                        uint8_t *unf = unf_ + 2 * bs + 8, *fil = fil_ + 2 * bs + 8;
                        cfc_load_ref(fenc, stride, bs, unf);
                        if (!c->fast_intra) ks265o_intra_filter_ref(unf, fil, bs, 0);
                        static const int kModes[7] = {0, 1, 26, 10, 18, 2, 34};
                        int nmodes = 7;
                        if (c->fast_intra) nmodes = bcost < (((uint32_t)thr >> 1) << c->fast_intra) ? 4 : 7;
                        if (c->scenecut == 0 && c->preset <= 1 && bcost < (1u << (2 * lg))) nmodes = 2;     /* enc@0x4a84a4..0x4a84bc, 0x4a8bb4 */
                        const cfc_intra t = {fenc, stride, lg, c->fast_intra, unf, fil};
                        uint32_t best = 0xfffffffu; int bm = 0;
                        for (int i = 0; i < nmodes; ++i) { const uint32_t s = cfc_intra_sad(&t, kModes[i]); if (s < best) { best = s; bm = kModes[i]; } }
                        if (!c->fast_intra) {                                                             /* enc@0x4a863e..0x4a8818 */
                            int centre = bm, cur = bm;
                            for (int step = 2; step >= 1; --step) {
                                int m2 = centre + step; cur = centre;
                                if ((unsigned)(m2 - 3) <= 31u) { const uint32_t s = cfc_intra_sad(&t, m2); if (s < best) { best = s; cur = m2; } }
                                m2 = centre - step;
                                if ((unsigned)(m2 - 3) <= 31u) { const uint32_t s = cfc_intra_sad(&t, m2); if (s < best) { best = s; cur = m2; } }
                                centre = cur;
                            }
                            bm = cur;
                        }
                        icost = best + 9;                                                                 /* enc@0x4a881f..0x4a8898 */
                        c->intra[blk] = (uint16_t)(icost > 0xffff ? 0xffff : icost); c->imode[blk] = (uint8_t)bm;
                        if (counted) {
                            c->sum_intra += (int32_t)icost;
                            if (c->aq) c->sum_intra_aq += (int32_t)(((uint32_t)c->invq[blk] * icost + 128u) >> 8);
                        }
                    } else icost = c->intra[blk];
                    if (icost < bcost) { bcost = icost; c->intra_wins += 1; }
                }
                if (any_inter) c->inter[blk] = (uint16_t)(bcost > 0xffff ? 0xffff : bcost);               /* enc@0x4a79bc..0x4a79e8 */
                const int v = counted ? (int)bcost : 0;
                int vq = v;
                if (c->aq) vq = ((int)((uint32_t)c->invq[blk] * (uint32_t)v) + 128) >> 8;                 /* enc@0x4a7a12..0x4a7a3d */
                sum88 += v; sum7c += vq;
                if (c->slice_type != 2) {                                                                 /* enc@0x4a7a47..0x4a7b22, 0x4a80e8..0x4a8148, 0x4a8310..0x4a835c */
                    const int ax = cfc_abs(mvq[0][0]), ay = cfc_abs(mvq[0][1]), s = ax + ay;
                    if (c->f3a8 != 0 || c->f36c == 2 || c->f538 != 0) {
                        if (s > 2) cnt40 += c->f3a8 != 0;
                        cnt4c += ((ax >> 6) + (ay >> 6)) > 0;
                    }
                    cnt78 += movthr <= s;
                    if (c->f3b4) {
                        if (c->d0 != 0 && bigthr < cfc_abs(mvq[0][0] >> 2) + cfc_abs(mvq[0][1] >> 2)) ++cnt30;
                        else if (c->d1 != 0 && bigthr < cfc_abs(mvq[1][0] >> 2) + cfc_abs(mvq[1][1] >> 2)) ++cnt30;
                    }
                }
            }
        if (c->slice_type != 2) { c->stats[0] = cnt40; c->stats[1] = cnt4c; c->stats[2] = cnt30; c->stats[3] = cnt78; }   /* enc@0x4a8a2a..0x4a8a6f */
        if (c->d1 != 0) sum88 = (int)((int32_t)((uint32_t)sum88 * 100u) / 130);                                          /* enc@0x4a8a83..0x4a8aad */
        else c->intra_done = 1;
        *s684 = sum88; *s7c8 = sum7c;
    }
    int cost = *s684;                                                                                     /* enc@0x4a749b..0x4a74d3 */
    if (c->flag) cost += (int)(((uint32_t)c->intra_wins / (uint32_t)(c->cnt * 8)) * (uint32_t)cost);
    c->ret = cost;
}

void kso_ref_cutree_finish(int cnt, const uint16_t *intra, const uint16_t *inv_qscale, const uint16_t *propagate, const double *aq_off, int dbl, double *out)
{
    for (int i = 0; i < cnt; ++i) {
        const int iw = ((int)((uint32_t)intra[i] * (uint32_t)inv_qscale[i]) + 128) >> 8;                  /* enc@0x4809a8..0x4809be */
        if (iw == 0) continue;
        uint32_t p = propagate[i];
        if (dbl) p *= 2;
        const double q = aq_off[i] - 1.8 * (kso_ref_log2(p + (uint32_t)iw) - kso_ref_log2((uint32_t)iw));   /* enc@0x4809df..0x480a20 */
        out[i] = -15.0 > q ? -15.0 : (q < 20.0 ? q : 20.0);
    }
}

/* this build's own rule (not the reference's), the mirror of ks265_qoff_ctu_map: one QP per CTU from the offsets of the lookahead's blocks (2^(lg + 1) luma samples: 4 x 4 or 2 x 2 per CTU) */
void kso_qoff_ctu_map(const double *off, int nx, int ny, int lg, int cols, int rows, int base_qp, int lo, int hi, int8_t *map)
{
    const int bpc = 64 >> (lg + 1);
    for (int cy = 0; cy < rows; ++cy)
        for (int cx = 0; cx < cols; ++cx) {
            double sum = 0.0; int cnt = 0;
            for (int by = cy * bpc; by < (cy * bpc + bpc < ny ? cy * bpc + bpc : ny); ++by)
                for (int bx = cx * bpc; bx < (cx * bpc + bpc < nx ? cx * bpc + bpc : nx); ++bx) { sum += off[by * nx + bx]; ++cnt; }
            int d = cnt ? (int)floor(sum / (double)cnt + 0.5) : 0;
            d = d < -12 ? -12 : d > 12 ? 12 : d;
            const int q = base_qp + d;
            map[cy * cols + cx] = (int8_t)(q < lo ? lo : q > hi ? hi : q);
        }
}
