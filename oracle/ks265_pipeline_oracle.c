/* ks265_pipeline_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ks265_pipeline_oracle.h).
 * Whole-frame stages restated on the CPU from the pinned kernels of ks265_oracle.c. */
#include "ks265_pipeline_oracle.h"
#include "ks265_subme_ref.h"
#include "ks265_me_ref.h"
#include "ks265_oracle.h"
#include <stdlib.h>
#include <string.h>

#define PAD_Y 80
#define PAD_C 40
#define PLANE_MARGIN 72              /* fractional planes are defined on [-72, W+72) x [-72, H+72) */
#define COST_INVALID 0xFFFFFFFFu
/* kso_cu8.log2_cu: bits 0..3 = log2 of the CU, bits 4..5 = inter partition (cfg->part): 0 = 2Nx2N, 1 = 2NxN, 2 = Nx2N (then the CU holds four TUs: interSplitFlag, H.265 7.4.9.8) */
#define CU_LOG2(c) ((c)->log2_cu & 15)
#define CU_PART(c) ((c)->log2_cu >> 4)

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclip(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs_(int v) { return v < 0 ? -v : v; }
static inline int isgn(int v) { return (v > 0) - (v < 0); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

int kso_frame_geometry(const kso_frame_cfg *cfg, kso_frame_geom *g)
{
    if (!cfg || !g || cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7)) return -4;
    g->pad_y = PAD_Y; g->pad_c = PAD_C;
    g->stride_y = align_up(cfg->width + 2 * PAD_Y, 128);
    g->stride_c = align_up(cfg->width / 2 + 2 * PAD_C, 64);
    g->rows_y = cfg->height + 2 * PAD_Y;
    g->rows_c = cfg->height / 2 + 2 * PAD_C;
    g->bytes_y = (int64_t)g->stride_y * (g->rows_y + 1);   /* one slack row: window / tile loads of the last padded row may run up to 64 bytes past it */
    g->bytes_c = (int64_t)g->stride_c * (g->rows_c + 1);
    g->ctu_cols = (cfg->width + 63) / 64;
    g->ctu_rows = (cfg->height + 63) / 64;
    g->pu_per_ctu = 85;
    g->bytes_pu = (int64_t)g->ctu_cols * g->ctu_rows * 85 * (int64_t)sizeof(kso_pu);
    g->bytes_cu8 = (int64_t)(cfg->width / 8) * (cfg->height / 8) * (int64_t)sizeof(kso_cu8);
    g->bytes_sao = (int64_t)g->ctu_cols * g->ctu_rows * 3 * (int64_t)sizeof(kso_sao_param);
    return 0;
}

/* pointer to sample (0,0) of a padded plane */
static inline uint8_t *org_y(const kso_frame_geom *g, uint8_t *p) { return p + (long)PAD_Y * g->stride_y + PAD_Y; }
static inline uint8_t *org_c(const kso_frame_geom *g, uint8_t *p) { return p + (long)PAD_C * g->stride_c + PAD_C; }

/* expandPicture_c enc@0x4a6ae0: replicate the edge samples into the border */
static void pad_plane(uint8_t *o, int stride, int w, int h, int pad)
{
    for (int y = 0; y < h; ++y) {
        uint8_t *r = o + (long)y * stride;
        memset(r - pad, r[0], (size_t)pad);
        memset(r + w, r[w - 1], (size_t)pad);
    }
    for (int y = 1; y <= pad; ++y) {
        memcpy(o - (long)y * stride - pad, o - pad, (size_t)(w + 2 * pad));
        memcpy(o + (long)(h - 1 + y) * stride - pad, o + (long)(h - 1) * stride - pad, (size_t)(w + 2 * pad));
    }
}

void kso_pad_picture(const kso_frame_cfg *cfg, kso_pic pic)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    pad_plane(org_y(&g, pic.y), g.stride_y, cfg->width, cfg->height, PAD_Y);
    pad_plane(org_c(&g, pic.u), g.stride_c, cfg->width / 2, cfg->height / 2, PAD_C);
    pad_plane(org_c(&g, pic.v), g.stride_c, cfg->width / 2, cfg->height / 2, PAD_C);
}

void kso_load_i420(const kso_frame_cfg *cfg, const uint8_t *i420, kso_pic dst)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int w = cfg->width, h = cfg->height;
    for (int y = 0; y < h; ++y) memcpy(org_y(&g, dst.y) + (long)y * g.stride_y, i420 + (long)y * w, (size_t)w);
    const uint8_t *u = i420 + (long)w * h, *v = u + (long)(w / 2) * (h / 2);
    for (int y = 0; y < h / 2; ++y) {
        memcpy(org_c(&g, dst.u) + (long)y * g.stride_c, u + (long)y * (w / 2), (size_t)(w / 2));
        memcpy(org_c(&g, dst.v) + (long)y * g.stride_c, v + (long)y * (w / 2), (size_t)(w / 2));
    }
    kso_pad_picture(cfg, dst);
}

void kso_store_i420(const kso_frame_cfg *cfg, kso_pic src, uint8_t *i420)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int w = cfg->width, h = cfg->height;
    for (int y = 0; y < h; ++y) memcpy(i420 + (long)y * w, org_y(&g, src.y) + (long)y * g.stride_y, (size_t)w);
    uint8_t *u = i420 + (long)w * h, *v = u + (long)(w / 2) * (h / 2);
    for (int y = 0; y < h / 2; ++y) {
        memcpy(u + (long)y * (w / 2), org_c(&g, src.u) + (long)y * g.stride_c, (size_t)(w / 2));
        memcpy(v + (long)y * (w / 2), org_c(&g, src.v) + (long)y * g.stride_c, (size_t)(w / 2));
    }
}

/* ------------------------------------------------------------------ Stage A0: fractional planes
 * plane[fy*4+fx](x,y) = normative luma sample at (x + fx/4, y + fy/4): interpLumaHor8to8_c enc@0x40e4f0 (fy = 0),
 * interpLumaVer8to8_c enc@0x40f0c0 (fx = 0), interpLumaHor8to16_c enc@0x40eb80 + interpLumaVer16to8_c enc@0x4100b0. */
void kso_ref_planes(const kso_frame_cfg *cfg, kso_pic ref, uint8_t *planes)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int W = cfg->width + 2 * PLANE_MARGIN, H = cfg->height + 2 * PLANE_MARGIN, s = g.stride_y;
    const uint8_t *src = org_y(&g, ref.y) - (long)PLANE_MARGIN * s - PLANE_MARGIN;
    memset(planes, 0, (size_t)(16 * g.bytes_y));
    memcpy(planes, ref.y, (size_t)g.bytes_y);
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 1; k < 16; ++k) {
        const int fy = k >> 2, fx = k & 3;
        uint8_t *dst = org_y(&g, planes + (long)k * g.bytes_y) - (long)PLANE_MARGIN * s - PLANE_MARGIN;
        if (!fy) ks265o_interp_luma_hor_8to8(dst, s, src, s, W, H, fx);
        else if (!fx) ks265o_interp_luma_ver_8to8(dst, s, src, s, W, H, fy);
        else {
            int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)W * (size_t)(H + 7));
            ks265o_interp_luma_hor_8to16(tmp, W, src - 3 * (long)s, s, W, H + 7, fx);
            ks265o_interp_luma_ver_16to8(dst, s, tmp + 3 * W, W, W, H, fy);
            free(tmp);
        }
    }
}

/* ------------------------------------------------------------------ motion-vector rate
 * createMvdCostTable enc@0x48b850: lambda(qp) x signed exp-Golomb length of the quarter-pel mvd (kso_mvd_bits, pinned on the reference's own table
 * slices: tests/test_me_search.py); lambda_q4 = 16 x the reference's integer lambda, so (lambda_q4 x bits) >> 4 is the table entry */
static int se_bits(int v) { return kso_mvd_bits(v); }
static int mv_cost(int mvx, int mvy, int px, int py, int lambda_q4) { return (lambda_q4 * (se_bits(mvx - px) + se_bits(mvy - py))) >> 4; }

/* PU index helpers: level l (0: 64x64 .. 3: 8x8), raster inside the CTU */
static const int kLevelBase[4] = {0, 1, 5, 21};
static inline int pu_index(int l, int px, int py) { return kLevelBase[l] + py * (1 << l) + px; }
/* 1 if the PU lies completely inside the picture */
static int pu_inside(const kso_frame_cfg *cfg, int cx, int cy, int l, int px, int py)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    return x0 + s <= cfg->width && y0 + s <= cfg->height;
}

/* motion predictor of a PU: integer MV of the nearest valid ancestor; for a root PU the temporal predictor
 * (co-located 64x64 MV of the previous picture, rounded to integer pel) or zero.  (meInitPoint enc@0x48af50
 * gathers spatial/merge candidates from already coded CTUs; a frame-parallel search cannot — SURVEY.md §7.3.) */
static void pu_predictor(const kso_frame_cfg *cfg, const kso_pu *ctu_pu, const kso_pu *prev_ctu_pu, int cx, int cy, int l, int px, int py, const int lim[4],
                         int *mx, int *my, int *root)
{
    for (int a = l - 1; a >= 0; --a) {
        int ax = px >> (l - a), ay = py >> (l - a);
        if (pu_inside(cfg, cx, cy, a, ax, ay)) {
            const kso_pu *p = &ctu_pu[pu_index(a, ax, ay)];
            *mx = p->mvx >> 2; *my = p->mvy >> 2; *root = 0;
            return;
        }
    }
    *root = 1; *mx = iclip(lim[0], lim[1], 0); *my = iclip(lim[2], lim[3], 0);      /* no predictor: the legal vector nearest to zero */
    if (prev_ctu_pu && prev_ctu_pu[0].cost != COST_INVALID) {
        *mx = iclip(lim[0], lim[1], (prev_ctu_pu[0].mvx + 2) >> 2);
        *my = iclip(lim[2], lim[3], (prev_ctu_pu[0].mvy + 2) >> 2);
    }
}

/* ------------------------------------------------------------------ Stage A0: motion pre-search on a three-level pyramid
 * The reference's lookahead runs a low-resolution motion search on 2:1 pictures (downsample_c enc@0x4a6a60 feeds estimateFrameCost; SURVEY.md
 * section 8(f) rank 2) and meInitPoint enc@0x48af50 starts the integer search from the best of several candidates.  A frame-parallel search has
 * no spatial neighbours to draw candidates from, so the candidate that makes the local pattern searches (DIA / HEX / UMH walk downhill from
 * their start point and cannot find a displaced match in texture without a gradient) robust comes from here: an EXHAUSTIVE search where it is
 * cheap.  L1 = downsample_c(luma), L2 = downsample_c(L1).  Per 8x8 block of L2 (32x32 samples) every vector of +-(range/4 - 1) (31 columns at range 64: two blocks share a 64-lane wave on the GPU); per 8x8 block of
 * L1 (16x16 samples) +-2 around twice the L2 vector; per 16x16 block of the picture +-1 around twice the L1 vector.  Cost = SAD + |mx| + |my|
 * (ties and flat areas fall to the shorter vector), first minimum in raster order of (my, mx).  Low-resolution reads clamp to the picture
 * (no border), the full-resolution step reads the padded planes like stage A.  Output: one integer vector per 16x16 block. */
static void pyr_down(const uint8_t *src, long st, int w, int h, uint8_t *dst /* w/2 x h/2, packed */)
{
    ks265o_downsample(dst, src, w / 2, (int)st, w / 2, h / 2);
}
static uint32_t sad_clamped(const uint8_t *cur, const uint8_t *ref, int W, int H, int x0, int y0, int bw, int bh, int mx, int my)
{
    uint32_t s = 0;
    for (int y = 0; y < bh; ++y) {
        const int ry = iclip(0, H - 1, y0 + y + my);
        for (int x = 0; x < bw; ++x) {
            const int rx = iclip(0, W - 1, x0 + x + mx);
            const int d = cur[(long)(y0 + y) * W + x0 + x] - ref[(long)ry * W + rx];
            s += (uint32_t)(d < 0 ? -d : d);
        }
    }
    return s;
}
/* legal vectors of the PUs of CTU (cx, cy) around the window offset (ox, oy): +-range around the offset, and never so far that a block of the CTU
 * leaves the 64-sample margin of the padded planes */
void kso_ctu_mv_limits(const kso_frame_cfg *cfg, int cx, int cy, int ox, int oy, int lim[4] /* lox, hix, loy, hiy */)
{
    const int xe = imin(cx * 64 + 64, cfg->width), ye = imin(cy * 64 + 64, cfg->height), r = cfg->me_range;
    lim[0] = imax(ox - r, -64 - cx * 64); lim[1] = imin(ox + r, cfg->width + 64 - xe);
    lim[2] = imax(oy - r, -64 - cy * 64); lim[3] = imin(oy + r, cfg->height + 64 - ye);
}
void kso_presearch(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, int16_t *field, int16_t *ctu_off /* ctu_cols x ctu_rows x {ox, oy} */)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const int W = cfg->width, H = cfg->height, W1 = W / 2, H1 = H / 2, W2 = W / 4, H2 = H / 4, W3 = W / 8, H3 = H / 8;
    const uint8_t *S = org_y(&g, src.y), *R = org_y(&g, ref.y);
    const long st = g.stride_y;
    uint8_t *c1 = malloc((size_t)W1 * H1), *r1 = malloc((size_t)W1 * H1), *c2 = malloc((size_t)W2 * H2), *r2 = malloc((size_t)W2 * H2);
    uint8_t *c3 = malloc((size_t)W3 * H3 + 8), *r3 = malloc((size_t)W3 * H3 + 8);
    pyr_down(S, st, W, H, c1); pyr_down(R, st, W, H, r1); pyr_down(c1, W1, W1, H1, c2); pyr_down(r1, W1, W1, H1, r2);
    /* L3 = 1/8 resolution: downsample_c needs even source sizes; W2, H2 are even, W3 = W2 / 2 */
    pyr_down(c2, W2, W2, H2, c3); pyr_down(r2, W2, W2, H2, r3);
    const int nb2x = (W2 + 7) / 8, nb2y = (H2 + 7) / 8, nb1x = (W1 + 7) / 8, nb1y = (H1 + 7) / 8, nb0x = (W + 15) / 16, nb0y = (H + 15) / 16;
    int16_t *mv2 = malloc(sizeof(int16_t) * 2 * (size_t)nb2x * nb2y), *mv1 = malloc(sizeof(int16_t) * 2 * (size_t)nb1x * nb1y);
    const int R2 = imax((cfg->me_range >> 2) - 1, 1), R3 = imax(cfg->me_range >> 2, 1);
    /* L3: one 8x8 block = one CTU; every vector of +-range/4 (+-2 range in samples).  A CTU VOTES for its best vector when that vector lies where the
     * zero-centred window (+-range, searched with margin) does not reach and matches at least a quarter better than the best vector that window covers.
     * If one vector collects the votes of at least half of the CTUs (a pan, a reference several pictures away) it becomes the picture's WINDOW OFFSET
     * (x 8, per CTU clipped so that the CTU's blocks stay inside the planes' margin): stage A then searches +-range around it instead of around zero.
     * Periodic content and 1/8-resolution aliasing produce scattered votes, never a majority: the windows stay where they are. */
    const int side3 = 2 * R3 + 1;
    int *votes = calloc((size_t)side3 * side3, sizeof(int));
    int16_t *cand = malloc(sizeof(int16_t) * 2 * (size_t)g.ctu_cols * g.ctu_rows);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            const int bw = imin(8, W3 - 8 * cx), bh = imin(8, H3 - 8 * cy);
            uint32_t best = 0xffffffffu, near = 0xffffffffu; int bmx = 0, bmy = 0;
            for (int my = -R3; my <= R3; ++my)
                for (int mx = -R3; mx <= R3; ++mx) {
                    const uint32_t c = sad_clamped(c3, r3, W3, H3, 8 * cx, 8 * cy, bw, bh, mx, my) + (uint32_t)(iabs_(mx) + iabs_(my));
                    if (c < best) { best = c; bmx = mx; bmy = my; }
                    if (iabs_(8 * mx) <= cfg->me_range / 2 && iabs_(8 * my) <= cfg->me_range / 2 && c < near) near = c;
                }
            if ((iabs_(8 * bmx) <= cfg->me_range / 2 && iabs_(8 * bmy) <= cfg->me_range / 2) || 4 * (uint64_t)best >= 3 * (uint64_t)near) { bmx = 0; bmy = 0; }
            cand[2 * (cy * g.ctu_cols + cx)] = (int16_t)bmx; cand[2 * (cy * g.ctu_cols + cx) + 1] = (int16_t)bmy;
        }
    int gmx = 0, gmy = 0, top = 0;
    for (int i = 0; i < g.ctu_cols * g.ctu_rows; ++i) if (cand[2 * i] || cand[2 * i + 1]) ++votes[(cand[2 * i + 1] + R3) * side3 + cand[2 * i] + R3];
    for (int i = 0; i < side3 * side3; ++i) if (votes[i] > top) { top = votes[i]; gmx = i % side3 - R3; gmy = i / side3 - R3; }     /* ties: the first in raster order */
    if (2 * top < g.ctu_cols * g.ctu_rows) { gmx = 0; gmy = 0; }
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            const int xe = imin(cx * 64 + 64, W), ye = imin(cy * 64 + 64, H);
            /* multiples of 16: the GPU stages its window with 16-byte loads */
            ctu_off[2 * (cy * g.ctu_cols + cx)] = (int16_t)(iclip(-64 - cx * 64, W + 64 - xe, 8 * gmx) & ~15);
            ctu_off[2 * (cy * g.ctu_cols + cx) + 1] = (int16_t)(iclip(-64 - cy * 64, H + 64 - ye, 8 * gmy) & ~15);
        }
    free(votes); free(cand);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int by = 0; by < nb2y; ++by)
        for (int bx = 0; bx < nb2x; ++bx) {
            const int bw = imin(8, W2 - 8 * bx), bh = imin(8, H2 - 8 * by);
            const int16_t *o = &ctu_off[2 * ((by >> 1) * g.ctu_cols + (bx >> 1))];
            const int c2x = o[0] / 4, c2y = o[1] / 4;              /* multiples of 16: exact */
            uint32_t best = 0xffffffffu; int bmx = 0, bmy = 0;
            for (int my = c2y - R2; my <= c2y + R2; ++my)
                for (int mx = c2x - R2; mx <= c2x + R2; ++mx) {
                    const uint32_t c = sad_clamped(c2, r2, W2, H2, 8 * bx, 8 * by, bw, bh, mx, my) + (uint32_t)(iabs_(mx) + iabs_(my));
                    if (c < best) { best = c; bmx = mx; bmy = my; }
                }
            mv2[2 * (by * nb2x + bx)] = (int16_t)bmx; mv2[2 * (by * nb2x + bx) + 1] = (int16_t)bmy;
        }
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int by = 0; by < nb1y; ++by)
        for (int bx = 0; bx < nb1x; ++bx) {
            const int bw = imin(8, W1 - 8 * bx), bh = imin(8, H1 - 8 * by);
            const int16_t *p = &mv2[2 * ((by >> 1) * nb2x + (bx >> 1))];
            uint32_t best = 0xffffffffu; int bmx = 0, bmy = 0;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int mx = 2 * p[0] + dx, my = 2 * p[1] + dy;
                    const uint32_t c = sad_clamped(c1, r1, W1, H1, 8 * bx, 8 * by, bw, bh, mx, my) + (uint32_t)(iabs_(mx) + iabs_(my));
                    if (c < best) { best = c; bmx = mx; bmy = my; }
                }
            mv1[2 * (by * nb1x + bx)] = (int16_t)bmx; mv1[2 * (by * nb1x + bx) + 1] = (int16_t)bmy;
        }
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int by = 0; by < nb0y; ++by)
        for (int bx = 0; bx < nb0x; ++bx) {
            const int bw = imin(16, W - 16 * bx), bh = imin(16, H - 16 * by);
            const int16_t *p = &mv1[2 * (by * nb1x + bx)];
            const int16_t *o = &ctu_off[2 * ((by >> 2) * g.ctu_cols + (bx >> 2))];
            int lim[4];
            kso_ctu_mv_limits(cfg, bx >> 2, by >> 2, o[0], o[1], lim);
            const uint8_t *fenc = S + (long)(16 * by) * st + 16 * bx;
            uint32_t best = 0xffffffffu; int bmx = 0, bmy = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int mx = iclip(lim[0], lim[1], 2 * p[0] + dx), my = iclip(lim[2], lim[3], 2 * p[1] + dy);
                    const uint32_t c = ks265o_sad(fenc, R + (long)(16 * by + my) * st + 16 * bx + mx, st, st, bh, bw) + (uint32_t)(iabs_(mx) + iabs_(my));
                    if (c < best) { best = c; bmx = mx; bmy = my; }
                }
            field[2 * (by * nb0x + bx)] = (int16_t)bmx; field[2 * (by * nb0x + bx) + 1] = (int16_t)bmy;
        }
    free(c1); free(r1); free(c2); free(r2); free(c3); free(r3); free(mv2); free(mv1);
}

/* ------------------------------------------------------------------ Stage A: integer search (motionSearchOneRef enc@0x483f40)
 * For every PU of every CTU, coarse to fine.  The search PATTERNS are the reference's own functions, restated in ks265_me_ref.c from the
 * disassembly and pinned against traces of the reference binary (tests/golden/me_search.npz): interMeDia enc@0x48fbe0 (-me 0),
 * interMeHex enc@0x48fde0 (-me 1), interMeUMH enc@0x4907b0 (-me 2).  What surrounds them follows motionSearchOneRef as far as a
 * frame-parallel search can:
 *   - start point: the predictor (meInitPoint enc@0x48af50 picks the best of the spatial / merge / zero candidates of already coded
 *     CTUs; here: nearest valid ancestor's vector, for a root PU the co-located vector of the previous picture and the zero vector);
 *   - mvd cost tables p_cost_mvx / p_cost_mvy = lambda x se-Golomb bits of the quarter-pel difference to the predictor, one u16 table
 *     per component (createMvdCostTable enc@0x48b850);
 *   - -me 2 at -preset slow: tME+0x368 = 16 -> a PU whose start-point SAD is below 16 per sample runs interMeHex instead of interMeUMH
 *     (enc@0x483fe8..0x48400d, 0x484060); cfg->me_hex_thr carries that threshold (0 = always UMH, what -preset veryslow resolves to);
 *   - merange: the full range for a root PU, a quarter of it (>= 4) around an inherited vector (adaptiveMeSearchRange enc@0x483e70
 *     shrinks tME+0x68 from the spread of the neighbouring candidates; closed heuristics);
 *   - mv limits +-me_range around the PU position; candidates further than 66 samples away are skipped (the GPU's staged window). */
#define ME_TAB 208                      /* |window offset| <= 2 range, + range, + slack */
void kso_me_integer(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const kso_pu *prev_pu, kso_pu *pu) { kso_me_integer_ex(cfg, src, ref, prev_pu, pu, NULL); }
/* off_out (may be NULL): the CTUs' window offsets the search used, 2 per CTU (zero without cfg->pre_search) - what kso_me_propagate needs for the same limits */
void kso_me_integer_ex(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const kso_pu *prev_pu, kso_pu *pu, int16_t *off_out)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y), *R = org_y(&g, ref.y);
    long st = g.stride_y;
    int range = cfg->me_range, lam = cfg->lambda_q4;
    const int nb0x = (cfg->width + 15) / 16, nb0y = (cfg->height + 15) / 16;
    int16_t *field = NULL, *ctu_off = NULL;
    if (cfg->pre_search) {
        field = malloc(sizeof(int16_t) * 2 * (size_t)nb0x * nb0y); ctu_off = malloc(sizeof(int16_t) * 2 * (size_t)g.ctu_cols * g.ctu_rows);
        kso_presearch(cfg, src, ref, field, ctu_off);
    }
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            kso_pu *cp = pu + (long)(cy * g.ctu_cols + cx) * 85;
            const kso_pu *pp = prev_pu ? prev_pu + (long)(cy * g.ctu_cols + cx) * 85 : NULL;
            /* the CTU's window offset (pre-search, else zero) and the vectors its PUs may take */
            const int ox = ctu_off ? ctu_off[2 * (cy * g.ctu_cols + cx)] : 0, oy = ctu_off ? ctu_off[2 * (cy * g.ctu_cols + cx) + 1] : 0;
            int lim[4];
            kso_ctu_mv_limits(cfg, cx, cy, ox, oy, lim);
            for (int l = 0; l < 4; ++l)
                for (int py = 0; py < (1 << l); ++py)
                    for (int px = 0; px < (1 << l); ++px) {
                        kso_pu *o = &cp[pu_index(l, px, py)];
                        int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
                        if (!pu_inside(cfg, cx, cy, l, px, py)) { memset(o, 0, sizeof *o); o->cost = COST_INVALID; o->dist = COST_INVALID; continue; }
                        int pmx, pmy, root;
                        pu_predictor(cfg, cp, pp, cx, cy, l, px, py, lim, &pmx, &pmy, &root);
                        uint16_t tx[4 * (2 * ME_TAB + 1)], ty[4 * (2 * ME_TAB + 1)];       /* indexed by the quarter-pel mv like the reference's */
                        for (int v = -ME_TAB; v <= ME_TAB; ++v) {
                            tx[4 * (v + ME_TAB)] = (uint16_t)((lam * se_bits((v - pmx) << 2)) >> 4);
                            ty[4 * (v + ME_TAB)] = (uint16_t)((lam * se_bits((v - pmy) << 2)) >> 4);
                        }
                        kso_me m;
                        memset(&m, 0, sizeof m);
                        m.fenc = S + (long)y0 * st + x0; m.fstride = (int)st;
                        m.ref0 = R + (long)y0 * st + x0; m.stride = (int)st;
                        m.log2w = m.log2h = 6 - l;
                        m.cmx = tx + 4 * ME_TAB; m.cmy = ty + 4 * ME_TAB;
                        m.merange = root ? range : imax(range >> 2, 4);
                        m.mv_min_x = lim[0]; m.mv_max_x = lim[1]; m.mv_min_y = lim[2]; m.mv_max_y = lim[3];
                        m.dist = ks265o_sad;
                        m.chk = 2; m.gx0 = ox - 66; m.gx1 = ox + 66; m.gy0 = oy - 66; m.gy1 = oy + 66;       /* the staged window */
                        m.mx = pmx; m.my = pmy;
                        uint32_t sad0 = ks265o_sad(m.fenc, m.ref0 + (long)pmy * st + pmx, st, st, s, s);
                        m.cost = sad0 + m.cmx[4 * pmx] + m.cmy[4 * pmy];
                        if (root && (pmx || pmy) && lim[0] <= 0 && lim[1] >= 0 && lim[2] <= 0 && lim[3] >= 0) {   /* second start candidate: the zero vector */
                            uint32_t s0 = ks265o_sad(m.fenc, m.ref0, st, st, s, s);
                            uint32_t c0 = s0 + m.cmx[0] + m.cmy[0];
                            if (c0 < m.cost) { m.cost = c0; m.mx = 0; m.my = 0; sad0 = s0; }
                        }
                        if (field) {                            /* third start candidate: the pre-search vector of the 16x16 block under the PU's centre */
                            const int16_t *f = &field[2 * (imin((y0 + s / 2) >> 4, nb0y - 1) * nb0x + imin((x0 + s / 2) >> 4, nb0x - 1))];
                            const int fx = f[0], fy = f[1];
                            if (fx != m.mx || fy != m.my) {
                                uint32_t s1 = ks265o_sad(m.fenc, m.ref0 + (long)fy * st + fx, st, st, s, s);
                                uint32_t c1 = s1 + m.cmx[4 * fx] + m.cmy[4 * fy];
                                if (c1 < m.cost) { m.cost = c1; m.mx = fx; m.my = fy; sad0 = s1; }
                            }
                        }
                        if (cfg->me_method == 0) kso_ref_me_dia(&m);
                        else if (cfg->me_method == 1 || (cfg->me_hex_thr > 0 && sad0 < ((uint32_t)cfg->me_hex_thr << (2 * (6 - l))))) kso_ref_me_hex(&m);
                        else kso_ref_me_umh(&m);
                        o->mvx = (int16_t)(m.mx << 2); o->mvy = (int16_t)(m.my << 2);
                        o->mvpx = (int16_t)(pmx << 2); o->mvpy = (int16_t)(pmy << 2);
                        o->cost = m.cost;
                        o->dist = m.cost - (uint32_t)(m.cmx[4 * m.mx] + m.cmy[4 * m.my]);
                    }
        }
    if (off_out) { if (ctu_off) memcpy(off_out, ctu_off, sizeof(int16_t) * 2 * (size_t)g.ctu_cols * g.ctu_rows); else memset(off_out, 0, sizeof(int16_t) * 2 * (size_t)g.ctu_cols * g.ctu_rows); }
    free(field); free(ctu_off);
}

/* ------------------------------------------------------------------ Stage A2: vector propagation between neighbouring PUs (cfg->propagate rounds)
 * meInitPoint enc@0x48af50 starts every search from the vectors of the already coded neighbours (AMVP / merge candidates); a frame-parallel search has none, and
 * a pattern search that starts from the wrong place cannot find a displaced match in texture without a gradient (a small object in front of a panning background:
 * the pre-search sees it only where it fills a 32x32 block).  After the integer search every PU therefore tries the integer vectors its four neighbours of the
 * same size found - left, above, right, below, in that order, across CTU borders, as they were BEFORE the round (in -> out: no order between PUs) - skipping
 * vectors outside its CTU's limits, its own vector and repeats; cost = SAD + the PU's own vector rate (predictor mvp of its record), strict '<' against the
 * running best.  ctu_off: the CTUs' window offsets of the search (kso_me_integer_ex), NULL = zero.
 * Measured with tools/rd_eval.py (832x480 bench-style clip, qp 27, one round): IPPP P pictures -23 % bytes at +0.09 dB, hierarchical-B 8 all P / B pictures
 * -21 % at +0.1 dB; a second round -1 %, eight neighbours -1 %, parent / child vectors nothing. */
void kso_me_propagate(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const int16_t *ctu_off, const kso_pu *in, kso_pu *out)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y), *R = org_y(&g, ref.y);
    const long st = g.stride_y;
    const int lam = cfg->lambda_q4;
    static const int nx[4] = {-1, 0, 1, 0}, ny[4] = {0, -1, 0, 1};
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            const long ctu = cy * g.ctu_cols + cx;
            int lim[4];
            kso_ctu_mv_limits(cfg, cx, cy, ctu_off ? ctu_off[2 * ctu] : 0, ctu_off ? ctu_off[2 * ctu + 1] : 0, lim);
            for (int l = 0; l < 4; ++l)
                for (int py = 0; py < (1 << l); ++py)
                    for (int px = 0; px < (1 << l); ++px) {
                        kso_pu o = in[ctu * 85 + pu_index(l, px, py)];
                        if (o.cost != COST_INVALID) {
                            const int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, n = 1 << l;
                            const int own_x = o.mvx >> 2, own_y = o.mvy >> 2;
                            int tried[4][2], nt = 0;
                            for (int k = 0; k < 4; ++k) {
                                const int gx = cx * n + px + nx[k], gy = cy * n + py + ny[k];                 /* the neighbour in units of this level's PUs */
                                if (gx < 0 || gy < 0 || gx >= g.ctu_cols * n || gy >= g.ctu_rows * n) continue;
                                const kso_pu *q = &in[(long)((gy >> l) * g.ctu_cols + (gx >> l)) * 85 + pu_index(l, gx & (n - 1), gy & (n - 1))];
                                if (q->cost == COST_INVALID) continue;
                                const int mx = q->mvx >> 2, my = q->mvy >> 2;
                                if (mx < lim[0] || mx > lim[1] || my < lim[2] || my > lim[3]) continue;
                                if (mx == own_x && my == own_y) continue;
                                int dup = 0;
                                for (int j = 0; j < nt; ++j) dup |= tried[j][0] == mx && tried[j][1] == my;
                                if (dup) continue;
                                tried[nt][0] = mx; tried[nt][1] = my; ++nt;
                                const uint32_t d = ks265o_sad(S + (long)y0 * st + x0, R + (long)(y0 + my) * st + x0 + mx, st, st, s, s);
                                const uint32_t c = d + (uint32_t)((lam * se_bits((mx << 2) - o.mvpx)) >> 4) + (uint32_t)((lam * se_bits((my << 2) - o.mvpy)) >> 4);
                                if (c < o.cost) { o.cost = c; o.dist = d; o.mvx = (int16_t)(mx << 2); o.mvy = (int16_t)(my << 2); }
                            }
                        }
                        out[ctu * 85 + pu_index(l, px, py)] = o;
                    }
        }
}

/* ------------------------------------------------------------------ Stage B: sub-pel refinement = the reference's, per PU (round 4)
 * getMvResolution enc@0x483ca0 -> subMeSquare enc@0x4b5660 (subMeHpel_RealInterp enc@0x4b4e90, subMeQpel_8Sad_*_RealInterp enc@0x4b2bc0-0x4b43a0) as restated in
 * ks265_subme_ref.c and pinned on 1 840 recorded calls (tests/test_subme.py).  What the frame-parallel stage feeds them:
 *   start = the integer stage's vector and cost (SAD + rate);   rate = this pipeline's mv_cost around the PU's predictor (the reference: its tables around AMVP);
 *   the four neighbour SADs getMvResolution looks at are computed at the integer winner (the reference re-uses the search's last sad4 when it is that: 99.6 % of calls);
 *   cfg->subme 1 / 2 = tME+0x36c, cfg->sub_satd = tME+0x64 / TPredUnit+0x40 (Hadamard; veryslow, placebo), cfg->sub_thr = cfg+0x464 = tME+0x3c0, cfg->sub_flat = tME+0x3c4,
 *   cfg->sub_cap / sub_cap_step = cfg+0x498 / +0x49c, cfg->sub_diag_fast = cfg+0x580; tME+0x60 = 0 (uni-directional search), cfg+0x568 = 0 (its partner TPredUnit+0x140 comes
 *   out of the reference's closed RD loop; only ultrafast / superfast set it).
 * Then - this pipeline's own, not the reference's - the record's cost becomes Hadamard distortion of the chosen prediction + rate: the CU tree, the merge pass and the
 * bi decision compare SATD-based costs (the reference decides by RD after the search; closed code). */
typedef struct { int px, py, lam; } sub_rate_ctx;
static uint32_t sub_rate(void *ctx, int qx, int qy) { const sub_rate_ctx *c = ctx; return (uint32_t)mv_cost(qx, qy, c->px, c->py, c->lam); }
void kso_me_subpel(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes, kso_pu *pu)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y), *R = org_y(&g, (uint8_t *)planes);          /* plane 0 = the padded reference picture */
    long st = g.stride_y;
    int lam = cfg->lambda_q4;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            kso_pu *cp = pu + (long)(cy * g.ctu_cols + cx) * 85;
            for (int l = 0; l < 4; ++l)
                for (int py = 0; py < (1 << l); ++py)
                    for (int px = 0; px < (1 << l); ++px) {
                        kso_pu *o = &cp[pu_index(l, px, py)];
                        if (o->cost == COST_INVALID) continue;
                        int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
                        sub_rate_ctx rc = {o->mvpx, o->mvpy, lam};
                        kso_subme m;
                        memset(&m, 0, sizeof m);
                        m.fenc = S + (long)y0 * st + x0; m.fstride = (int)st;
                        m.ref0 = R + (long)(y0 + (o->mvy >> 2)) * st + x0 + (o->mvx >> 2); m.stride = (int)st;
                        m.log2w = m.log2h = 6 - l;
                        m.dist = cfg->sub_satd ? ks265o_had : ks265o_sad; m.recost = cfg->sub_satd;
                        m.subme = cfg->subme; m.mvres_thr = cfg->sub_thr; m.hpel_diag_fast = cfg->sub_diag_fast; m.c568 = 0; m.pu140 = 0x0FFFFFFFu;
                        m.flat_factor = cfg->sub_flat; m.flat_shift = 0;
                        m.rate = sub_rate; m.rate_ctx = &rc;
                        m.mx = o->mvx; m.my = o->mvy; m.cost = o->cost;
                        {
                            uint32_t c4[4];
                            m.do_subpel = kso_ref_mv_resolution(cfg->sub_cap, cfg->sub_cap_step, m.log2w, m.log2h, o->cost, 1, 0, sub_rate(&rc, o->mvx, o->mvy), cfg->sub_thr, 0, c4, 0,
                                                                cfg->sub_thr, 0, m.fenc, m.fstride, m.ref0, m.stride);
                        }
                        kso_ref_subme_square(&m);
                        {   /* the decision cost of the record: Hadamard of the chosen prediction + rate */
                            const uint8_t *pl = org_y(&g, (uint8_t *)planes + (long)((m.my & 3) * 4 + (m.mx & 3)) * g.bytes_y);
                            const uint32_t d = ks265o_had(m.fenc, pl + (long)(y0 + (m.my >> 2)) * st + x0 + (m.mx >> 2), st, st, s, s);
                            o->mvx = (int16_t)m.mx; o->mvy = (int16_t)m.my; o->dist = d; o->cost = d + (uint32_t)mv_cost(m.mx, m.my, o->mvpx, o->mvpy, lam);
                        }
                    }
        }
}

/* Price of splitting an inter CU into four, in bits at the motion lambda, on top of the children's SATD + vector rate: their flags and vectors, and the
 * smaller transforms the residual then gets (one TU per CU).  12 in round 1 (the flags alone).  Measured in round 2 with this oracle + the stream writer, bytes
 * of the P / B pictures and their PSNR-Y at the same QP (one QP step: 17 % for 0.71 dB):
 *   P pictures (IPPP, three clip types, 416x240 / 832x480, qp 27..30):  40: -2.9 .. -5.3 % at -0.02 .. -0.04 dB;  80: -0.4 .. -7.2 % at -0.05 .. -0.08 dB
 *   hierarchical-B 8, 832x480 bench-style clip, all P / B pictures:     40: -11.3 % at -0.03 dB;  80: -15.0 % at -0.07 dB (qp 27), -31.8 % at -0.06 dB (qp 35);
 *                                                                      100: -15.4 % at -0.09 dB; 200: -14 % at -0.15 dB
 * hence 40 for the one-list records of P pictures and 80 for the two-list records (B pictures, multi-reference P); intra pictures barely move (own constant). */
#ifndef SPLIT_BITS_P
#define SPLIT_BITS_P 40
#endif
#ifndef SPLIT_BITS_B
#define SPLIT_BITS_B 80
#endif
/* ------------------------------------------------------------------ Stage C: CU quadtree
 * bottom-up compare of processTree enc@0x4722a0 (the reference adds RD cost and early exits; closed code). */
/* cfg->intra_inter: P / B pictures may hold intra CUs (EncIntraMD.cpp lineage; decideLumaMode enc@0x49acc0 is closed RD code).  A frame-parallel decision has the
 * pre-selection cost of every block from SOURCE neighbours (kso_intra_decide_ex: SATD + lambda x mode bits): it competes with the block's inter cost, with a
 * price for what an intra CU costs beyond its mode (pred_mode_flag, no merge / skip, a residual that the reconstructed neighbours make larger than the source
 * neighbours promise): 96 bits at the picture's lambda - measured with tools/rd_eval.py (832x480, rdo 4, HM lambda): 24 / 64 / 96 / 160 bits give 1.447 / 1.440 / 1.430 / 1.429 x the
 * reference's bytes at equal PSNR for IPPP (no intra CUs: 1.45) and 96 gives 1.337 against 1.385 for hierarchical B (anchors 8 pictures apart uncover a lot).
 * icost / imode: 85 per CTU, PU indexing; NULL = no intra candidates. */
#ifndef INTRA_BIAS_BITS
#define INTRA_BIAS_BITS 96
#endif
static uint32_t node_own_cost(const kso_frame_cfg *cfg, uint32_t inter, const uint32_t *icost, int idx, int l, uint8_t *use_intra)
{
    use_intra[idx] = 0;
    if (!icost || l == 0 || icost[idx] == COST_INVALID || (l == 3 && cfg->intra_inter >= 2)) return inter;   /* intra_inter 2: no 8x8 intra CUs in P / B pictures */
    const uint64_t ic = (uint64_t)icost[idx] + (uint64_t)((cfg->lambda_q4 * INTRA_BIAS_BITS) >> 4);
    if (inter == COST_INVALID || ic < inter) { use_intra[idx] = 1; return ic > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)ic; }
    return inter;
}
/* cfg->part (-part 1, qy265enc.h:131: 2NxN / Nx2N prediction units; slower, veryslow, placebo): a CU of 64 / 32 / 16 samples may be coded in two halves.  The
 * reference searches every such PU on its own (motionSearchOneRef enc@0x483f40 per TPredUnit: the traces of tests/golden/subme.npz hold 64x32 .. 8x16 calls) inside its
 * closed RD loop; the frame-parallel form prices each half with the vectors the square search already refined for this area - the CU's own 2Nx2N vector and the vectors of
 * the half's two quarter-size PUs - by Hadamard cost of the half's prediction + rate against the CU's predictor, first strict minimum in that order.  A partition is
 * considered when at least one half moves off the 2Nx2N vector; its cost = both halves + lambda x PART_BITS (part_mode bins, a second merge flag / reference, the CU's
 * four transform units instead of one).  P pictures with one reference picture; 8x8 CUs are left whole (their halves would need vectors per 8x4 block). */
#ifndef PART_BITS
#define PART_BITS 6
#endif
typedef struct { uint32_t cost[2]; int16_t mv[2][2][2]; int16_t mv1[2][2][2]; uint8_t dir[2][2]; } rect_rec;     /* [orientation: 0 = 2NxN (top, bottom), 1 = Nx2N (left, right)][half][x, y]; cost COST_INVALID = not considered; B pictures: mv1 / dir = the half's list-1 vector and direction */
typedef struct { const kso_frame_cfg *cfg; const kso_frame_geom *g; const uint8_t *S, *planes; } rect_ctx;
static uint32_t rect_half_cost(const rect_ctx *rc, int x0, int y0, int w, int h, int mvx, int mvy, int px, int py)
{
    const long st = rc->g->stride_y;
    const uint8_t *pl = org_y(rc->g, (uint8_t *)rc->planes + (long)((mvy & 3) * 4 + (mvx & 3)) * rc->g->bytes_y);
    return ks265o_had(rc->S + (long)y0 * st + x0, pl + (long)(y0 + (mvy >> 2)) * st + x0 + (mvx >> 2), st, st, h, w) + (uint32_t)mv_cost(mvx, mvy, px, py, rc->cfg->lambda_q4);
}
static void rect_eval(const rect_ctx *rc, const kso_pu *cp, int cx, int cy, int l, int px, int py, rect_rec *out)
{
    const int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    const kso_pu *P = &cp[pu_index(l, px, py)];
    out->cost[0] = out->cost[1] = COST_INVALID;
    if (l > 2 || P->cost == COST_INVALID) return;
    for (int o = 0; o < 2; ++o) {
        uint64_t tot = (uint64_t)((rc->cfg->lambda_q4 * PART_BITS) >> 4);
        int moved = 0;
        for (int hf = 0; hf < 2; ++hf) {
            const int hx = o ? x0 + hf * (s / 2) : x0, hy = o ? y0 : y0 + hf * (s / 2), w = o ? s / 2 : s, h = o ? s : s / 2;
            /* the half's two quarter-size PUs: 2NxN half hf = children (0, hf), (1, hf); Nx2N half hf = children (hf, 0), (hf, 1) */
            const kso_pu *c0 = &cp[pu_index(l + 1, px * 2 + (o ? hf : 0), py * 2 + (o ? 0 : hf))], *c1 = &cp[pu_index(l + 1, px * 2 + (o ? hf : 1), py * 2 + (o ? 1 : hf))];
            int bx = P->mvx, by = P->mvy;
            uint32_t best = rect_half_cost(rc, hx, hy, w, h, bx, by, P->mvpx, P->mvpy);
            const kso_pu *cand[2] = {c0, c1};
            for (int k = 0; k < 2; ++k) {
                if (cand[k]->cost == COST_INVALID) continue;
                const uint32_t c = rect_half_cost(rc, hx, hy, w, h, cand[k]->mvx, cand[k]->mvy, P->mvpx, P->mvpy);
                if (c < best) { best = c; bx = cand[k]->mvx; by = cand[k]->mvy; }
            }
            moved |= bx != P->mvx || by != P->mvy;
            out->mv[o][hf][0] = (int16_t)bx; out->mv[o][hf][1] = (int16_t)by;
            tot += best;
        }
        if (moved) out->cost[o] = tot > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)tot;
    }
}
static uint32_t decide_node(const kso_frame_cfg *cfg, const rect_ctx *rc, const kso_pu *cp, const uint32_t *icost, int cx, int cy, int l, int px, int py, uint8_t *split /*[85]*/, uint8_t *use_intra /*[85]*/,
                            uint8_t *part /*[85]*/, rect_rec *rect /*[21]*/)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    if (x0 >= cfg->width || y0 >= cfg->height) return 0;           /* not in the picture: nothing to code */
    int idx = pu_index(l, px, py);
    uint32_t inter = cp[idx].cost;
    part[idx] = 0;
    if (rc && l < 3) {
        rect_eval(rc, cp, cx, cy, l, px, py, &rect[idx]);
        for (int o = 0; o < 2; ++o) if (rect[idx].cost[o] < inter) { inter = rect[idx].cost[o]; part[idx] = (uint8_t)(o + 1); }
    }
    uint32_t own = node_own_cost(cfg, inter, icost, idx, l, use_intra);
    if (l == 3) { split[idx] = 0; return own; }
    uint64_t sum = (uint64_t)((cfg->lambda_q4 * SPLIT_BITS_P) >> 4); /* what three extra CUs cost beyond their own SATD + vector rate */
    for (int k = 0; k < 4; ++k) sum += decide_node(cfg, rc, cp, icost, cx, cy, l + 1, px * 2 + (k & 1), py * 2 + (k >> 1), split, use_intra, part, rect);
    if (own != COST_INVALID && (uint64_t)own <= sum) { split[idx] = 0; return own; }
    split[idx] = 1;
    return sum > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)sum;
}
static void emit_intra(const kso_frame_cfg *cfg, int x0, int y0, int s, int mode, int l, kso_cu8 *cu8)
{
    const int w8 = cfg->width / 8;
    for (int by = 0; by < s / 8; ++by)
        for (int bx = 0; bx < s / 8; ++bx) {
            kso_cu8 *c = &cu8[(long)(y0 / 8 + by) * w8 + x0 / 8 + bx];
            c->mvx = (int16_t)mode; c->mvy = 0; c->mv1x = 0; c->mv1y = 0; c->log2_cu = (uint8_t)(6 - l); c->cbf = 0; c->pred_mode = 2; c->inter_dir = 0;
        }
}
static void emit_node(const kso_frame_cfg *cfg, const kso_pu *cp, const uint8_t *imode, int cx, int cy, int l, int px, int py, const uint8_t *split, const uint8_t *use_intra,
                      const uint8_t *part, const rect_rec *rect, kso_cu8 *cu8)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, w8 = cfg->width / 8;
    if (x0 >= cfg->width || y0 >= cfg->height) return;
    int idx = pu_index(l, px, py);
    if (l < 3 && split[idx]) {
        for (int k = 0; k < 4; ++k) emit_node(cfg, cp, imode, cx, cy, l + 1, px * 2 + (k & 1), py * 2 + (k >> 1), split, use_intra, part, rect, cu8);
        return;
    }
    if (use_intra[idx]) { emit_intra(cfg, x0, y0, s, imode[idx], l, cu8); return; }
    const int pm = part ? part[idx] : 0;
    for (int by = 0; by < s / 8; ++by)
        for (int bx = 0; bx < s / 8; ++bx) {
            kso_cu8 *c = &cu8[(long)(y0 / 8 + by) * w8 + x0 / 8 + bx];
            c->mvx = cp[idx].mvx; c->mvy = cp[idx].mvy;
            if (pm) { const int hf = pm == 1 ? by >= s / 16 : bx >= s / 16; c->mvx = rect[idx].mv[pm - 1][hf][0]; c->mvy = rect[idx].mv[pm - 1][hf][1]; }
            c->mv1x = 0; c->mv1y = 0; c->log2_cu = (uint8_t)((6 - l) | (pm << 4)); c->cbf = 0; c->pred_mode = 0; c->inter_dir = 1;
        }
}
static void cu_decide_impl(const kso_frame_cfg *cfg, const rect_ctx *rc, const kso_pu *pu, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            const long cb = (long)(cy * g.ctu_cols + cx) * 85;
            uint8_t split[85], use_intra[85], part[85];
            rect_rec rect[21];
            memset(split, 0, sizeof split); memset(use_intra, 0, sizeof use_intra); memset(part, 0, sizeof part);
            decide_node(cfg, rc, pu + cb, icost ? icost + cb : NULL, cx, cy, 0, 0, 0, split, use_intra, part, rect);
            emit_node(cfg, pu + cb, imode ? imode + cb : NULL, cx, cy, 0, 0, 0, split, use_intra, part, rect, cu8);
        }
}
void kso_cu_decide_ii(const kso_frame_cfg *cfg, const kso_pu *pu, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8) { cu_decide_impl(cfg, NULL, pu, icost, imode, cu8); }
/* the CU decision of a P picture with cfg->part: needs the pixels (source, the reference's planes) for the halves' Hadamard costs */
void kso_cu_decide_part(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes, const kso_pu *pu, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const rect_ctx rc = {cfg, &g, org_y(&g, src.y), planes};
    cu_decide_impl(cfg, cfg->part ? &rc : NULL, pu, icost, imode, cu8);
}
void kso_cu_decide(const kso_frame_cfg *cfg, const kso_pu *pu, kso_cu8 *cu8) { kso_cu_decide_ii(cfg, pu, NULL, NULL, cu8); }

/* ------------------------------------------------------------------ multi-reference B pictures (round 5; -preset veryslow = config 5: ref 4 / 4)
 * Frame-parallel form of "motionSearchOneRef enc@0x483f40 once per reference picture of each list": every picture of a list is searched on its own (stages A0 / A / A2 / B),
 * kso_ref_pick keeps per PU and list the picture with the smallest cost + lambda x ref_idx bits (truncated unary; ties to the nearest picture), the bi-predictive decision
 * pairs the two lists' winners, and every later stage takes a block's pictures from its record: ks265_cu8.inter_dir = direction | idx0 << 4 | idx1 << 6, an index of a
 * list the block does not use is 0 (so that equal bytes = equal reference pictures: the boundary strength compares them).  List 0 holds past pictures only, list 1 future
 * ones: no picture is in both lists.  The stage functions keep their one-picture-per-list arguments; while a context is set (kso_set_mref) they take planes / pictures from it. */
static const kso_mref *g_mr;                                      /* (kso_mref: ks265_pipeline_oracle.h; idx0 / idx1 = per PU record the list's chosen picture) */
void kso_set_mref(const kso_mref *m) { g_mr = m; }
#define MR_PL0(arg, i) (g_mr ? g_mr->planes0[i] : (arg))
#define MR_PL1(arg, i) (g_mr ? g_mr->planes1[i] : (arg))
static int ref_idx_bits(int r, int nref) { return nref <= 1 ? 0 : (r < nref - 1 ? r + 1 : nref - 1); }
void kso_ref_pick(const kso_frame_cfg *cfg, int nref, const kso_pu *const *pu, kso_pu *out, uint8_t *idx)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const long n = (long)g.ctu_cols * g.ctu_rows * 85;
    for (long i = 0; i < n; ++i) {
        out[i] = pu[0][i]; idx[i] = 0;
        if (pu[0][i].cost == COST_INVALID) continue;
        uint32_t best = COST_INVALID;
        for (int r = 0; r < nref; ++r) {
            const uint64_t c = (uint64_t)pu[r][i].cost + (uint64_t)((cfg->lambda_q4 * ref_idx_bits(r, nref)) >> 4);
            if (c < best) { best = (uint32_t)c; out[i] = pu[r][i]; out[i].cost = best; idx[i] = (uint8_t)r; }
        }
    }
}

/* ------------------------------------------------------------------ Stage C2: merge pass (cfg->merge)
 * The reference decides merge / skip per CU against the candidates of already coded neighbours (GetMergeCandsFor*, skipFastDecision; closed code,
 * SURVEY.md B.9).  A frame-parallel decision has no coded neighbours, so this pass works on the motion field the CU decision left behind: every CU
 * looks at its five spatial merge neighbours (A1, B1, B0, A0, B2 of H.265 8.5.3.2.3: inside the picture, earlier in z-scan order, inter) and at the
 * zero vector, takes each one's motion as its own, and keeps the cheapest if it beats what the search found: SATD of the prediction + lambda x
 * (position in the list + 1) against the CU's search cost + lambda x 2.  All CUs decide at once on the SAME input field (cu_in -> cu_out), so the
 * result does not depend on any order; where a neighbour changes its motion in the same pass the adopted vector may no longer be a merge candidate
 * when the slice is written - it is then coded explicitly, which is only a (rare) loss of bits, never of correctness.  Single reference per list. */
static int z_of(int x, int y)                         /* z-scan address of the 8x8 block holding luma sample (x, y): CTB raster, then Morton inside the CTB */
{
    int bx = (x >> 3) & 7, by = (y >> 3) & 7, m = 0;
    for (int b = 0; b < 3; ++b) m |= ((bx >> b) & 1) << (2 * b) | ((by >> b) & 1) << (2 * b + 1);
    return m;
}
static int g_split_bits_b = SPLIT_BITS_B;
void kso_experiment_split_bits_b(int q4) { g_split_bits_b = q4; }
static int g_merge_bits_b = 32;                       /* experiment (test infrastructure): what explicit motion is taken to cost, in 1/16 bit, when a B picture's CU weighs a merge candidate */
void kso_experiment_merge_bits_b(int q4) { g_merge_bits_b = q4; }
void kso_merge_pass(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0, const uint8_t *planes1, const kso_pu *pu, const kso_pu_b *pub,
                    const kso_cu8 *cu_in, kso_cu8 *cu_out)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y);
    const long st = g.stride_y;
    const int lam = cfg->lambda_q4, w8 = cfg->width / 8, h8 = cfg->height / 8, is_b = pub != NULL;
    memcpy(cu_out, cu_in, sizeof(kso_cu8) * (size_t)w8 * h8);
#pragma omp parallel for schedule(dynamic, 4)
    for (int by = 0; by < h8; ++by)
        for (int bx = 0; bx < w8; ++bx) {
            const kso_cu8 *c = &cu_in[(long)by * w8 + bx];
            if (c->pred_mode != 0 || CU_LOG2(c) < 3 || CU_PART(c)) continue;                    /* (a CU in two partitions keeps the vectors of its partitions) */
            const int n = 1 << CU_LOG2(c), x = bx * 8, y = by * 8;
            if ((x & (n - 1)) || (y & (n - 1))) continue;                         /* a CU is handled at its first 8x8 block */
            const int cx = x >> 6, cy = y >> 6, l = 6 - CU_LOG2(c), idx = pu_index(l, (x & 63) >> CU_LOG2(c), (y & 63) >> CU_LOG2(c));
            const long rb = (long)(cy * g.ctu_cols + cx) * 85 + idx;
            const uint32_t cur = (is_b ? pub[rb].cost : pu[rb].cost);
            if (cur == COST_INVALID) continue;
            uint64_t best = (uint64_t)cur + (uint64_t)((lam * (is_b ? g_merge_bits_b : 32)) >> 4);     /* (g_merge_bits_b: experiment hook, tools/rd_eval.py; 32 = the pipeline's value) */
            int bestk = -1;
            kso_cu8 bm = *c;
            const int nx[5] = {x - 1, x + n - 1, x + n, x - 1, x - 1}, ny[5] = {y + n - 1, y - 1, y - 1, y + n, y - 1};   /* A1 B1 B0 A0 B2 */
            const int ctb = cy * g.ctu_cols + cx, zc = z_of(x, y);
            int pos = 0;
            for (int k = 0; k < 6; ++k) {
                kso_cu8 m;
                if (k < 5) {
                    if (nx[k] < 0 || ny[k] < 0 || nx[k] >= cfg->width || ny[k] >= cfg->height) continue;
                    const int nctb = (ny[k] >> 6) * g.ctu_cols + (nx[k] >> 6);
                    if (nctb > ctb || (nctb == ctb && z_of(nx[k], ny[k]) >= zc)) continue;                              /* not yet coded when this CU is */
                    m = cu_in[(long)(ny[k] >> 3) * w8 + (nx[k] >> 3)];
                    if (m.pred_mode != 0 || (m.log2_cu & 15) < 3) continue;
                } else { memset(&m, 0, sizeof m); m.inter_dir = (is_b && !(g_mr && g_mr->n1 == 0)) ? 3 : 1; }     /* (a context without list-1 pictures = a multi-reference P picture: its records are two-list records, its slice has one list) */
                const int dir = m.inter_dir & 3;
                /* a neighbour's vector may come from a CTU with another window offset: taken over here it must keep this CU's block inside the planes' margin */
                if ((dir & 1) && (x + (m.mvx >> 2) < -70 || x + (m.mvx >> 2) + n > cfg->width + 70 || y + (m.mvy >> 2) < -70 || y + (m.mvy >> 2) + n > cfg->height + 70)) continue;
                if ((dir & 2) && (x + (m.mv1x >> 2) < -70 || x + (m.mv1x >> 2) + n > cfg->width + 70 || y + (m.mv1y >> 2) < -70 || y + (m.mv1y >> 2) + n > cfg->height + 70)) continue;
                const uint8_t *p0 = NULL, *p1 = NULL;
                if (dir & 1) p0 = org_y(&g, (uint8_t *)MR_PL0(planes0, (m.inter_dir >> 4) & 3) + (long)((m.mvy & 3) * 4 + (m.mvx & 3)) * g.bytes_y) + (long)(y + (m.mvy >> 2)) * st + x + (m.mvx >> 2);
                if (dir & 2) p1 = org_y(&g, (uint8_t *)MR_PL1(planes1, (m.inter_dir >> 6) & 3) + (long)((m.mv1y & 3) * 4 + (m.mv1x & 3)) * g.bytes_y) + (long)(y + (m.mv1y >> 2)) * st + x + (m.mv1x >> 2);
                if (!p0) p0 = p1;
                if (!p1) p1 = p0;
                uint8_t avg[64 * 64];
                for (int yy = 0; yy < n; ++yy)
                    for (int xx = 0; xx < n; ++xx) avg[yy * n + xx] = (uint8_t)((p0[(long)yy * st + xx] + p1[(long)yy * st + xx] + 1) >> 1);
                const uint64_t j = (uint64_t)ks265o_had(S + (long)y * st + x, avg, st, n, n, n) + (uint64_t)((lam * 16 * (pos + 1)) >> 4);
                ++pos;
                if (j < best) { best = j; bestk = k; bm = m; }
            }
            if (bestk < 0) continue;
            for (int yy = 0; yy < n / 8; ++yy)
                for (int xx = 0; xx < n / 8; ++xx) {
                    kso_cu8 *o = &cu_out[(long)(by + yy) * w8 + bx + xx];
                    o->mvx = bm.mvx; o->mvy = bm.mvy; o->mv1x = bm.mv1x; o->mv1y = bm.mv1y; o->inter_dir = (uint8_t)(g_mr ? bm.inter_dir : (bm.inter_dir & 3));   /* (several pictures per list: the neighbour's pictures come with its motion) */
                }
        }
}

/* ------------------------------------------------------------------ B pictures
 * Per PU: L0 cost and L1 cost come from the two uni-directional searches; the bi-predictive candidate pairs the two winners and is judged on SATD
 * against the rounded average of the two 8-bit predictions (the final reconstruction uses the exact 14-bit average).  Ties prefer L0, then L1.
 *
 * cfg->bi_refine: joint refinement (motionSearchB enc@0x484e90 -> motionSearchBI enc@0x484910 -> interMeBiFull_opt enc@0x4898e0).  The cheaper list
 * stays as it is; the search target of the other one is T = clip8(2 org - pred_kept) (calcBiMeOrg enc@0x47b1a0), so that SAD(T, p) = 2 |org - (kept + p) / 2|.
 * Integer step: the list's own vector, clamped to the legal range in quarter samples and floored to integer (motionSearchBI), is moved at least four
 * samples inside the range, and the 8 x 8 window starts 3 + (mvp < 0) samples before it (interMeBiFull_opt); each position costs SAD + vector rate,
 * rows outside, columns inside, first minimum (interMeBiFull_c enc@0x4896d0, restated and pinned as ks265o_inter_me_bi_full; the rate here is this
 * pipeline's mv_cost on both components, not the reference's two 16-bit tables).  Sub-pel step: the two rings of kso_me_subpel on T, judged by SAD + rate like
 * the integer step (one measure from the window to the quarter sample; the reference's sub-pel search uses its Hadamard cost).  The refined pair
 * replaces the decision if its cost - SATD of the source against the rounded average + both vector rates, the measure of the unrefined pair - is lower. */
/* A bi-predictive pair is judged at 31 / 32 of its cost: luma SATD + vector rate undervalues what the average of two references is worth (the two pictures' coding
 * errors cancel, in chroma too, and the smaller residual costs fewer bits than its SATD says).  Measured with tools/rd_eval.py (832x480, hierarchical B, 33 pictures):
 * - 3.3 % bytes at the same PSNR-Y, chroma + 0.6 dB; the top-layer B pictures - 17 %; a bias of 1 / 16 and more loses again. */
#define BI_BIAS_SHIFT 5
#define BI_DIR_OF(d, i0, i1) ((uint32_t)(d) | (((d) & 1) ? (uint32_t)(i0) << 4 : 0u) | (((d) & 2) ? (uint32_t)(i1) << 6 : 0u))
/* the joint refinement of one PU's pair (see kso_bi_decide): a / b = the lists' records, o = the decision so far (updated if the refined pair is cheaper) */
static void bi_refine_pu(const kso_frame_cfg *cfg, const kso_frame_geom *gp, const uint8_t *S, const uint8_t *planes0, const uint8_t *planes1, uint32_t rbits,
                         int i0, int i1, int cx, int cy, int l, int px, int py, const kso_pu *a, const kso_pu *b, kso_pu_b *o)
{
    const kso_frame_geom g = *gp;
    const long st = g.stride_y;
    const int lam = cfg->lambda_q4;
    const int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    const uint8_t *p0 = org_y(&g, (uint8_t *)planes0 + (long)((a->mvy & 3) * 4 + (a->mvx & 3)) * g.bytes_y) + (long)(y0 + (a->mvy >> 2)) * st + x0 + (a->mvx >> 2);
    const uint8_t *p1 = org_y(&g, (uint8_t *)planes1 + (long)((b->mvy & 3) * 4 + (b->mvx & 3)) * g.bytes_y) + (long)(y0 + (b->mvy >> 2)) * st + x0 + (b->mvx >> 2);
    uint8_t avg[64 * 64];
    const int keep1 = b->cost < a->cost;                         /* list whose vector stays */
    const kso_pu *K = keep1 ? b : a, *O = keep1 ? a : b;
    const uint8_t *pk = keep1 ? p1 : p0, *planesO = keep1 ? planes0 : planes1;
    uint8_t T[64 * 64], orgp[64 * 64], kp[64 * 64];
    for (int y = 0; y < s; ++y) { memcpy(orgp + y * s, S + (long)(y0 + y) * st + x0, (size_t)s); memcpy(kp + y * s, pk + (long)y * st, (size_t)s); }
    ks265o_calc_bi_me_org(T, kp, orgp, s, s, s);
    const int xe = imin(cx * 64 + 64, cfg->width), ye = imin(cy * 64 + 64, cfg->height);
    const int lox = -64 - cx * 64, hix = cfg->width + 64 - xe, loy = -64 - cy * 64, hiy = cfg->height + 64 - ye;
    int imx = iclip(4 * lox, 4 * hix, O->mvx) >> 2, imy = iclip(4 * loy, 4 * hiy, O->mvy) >> 2;
    if (imx <= lox + 3) imx = lox + 4; else if (imx >= hix - 3) imx = hix - 4;
    if (imy <= loy + 3) imy = loy + 4; else if (imy >= hiy - 3) imy = hiy - 4;
    const int sx = imx - 3 - (O->mvpx < 0), sy = imy - 3 - (O->mvpy < 0);
    const uint8_t *R = org_y(&g, (uint8_t *)planesO);
    uint32_t bc = 0xfffffffu; int bx = 0, by = 0;
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            uint32_t cc = ks265o_sad(T, R + (long)(y0 + sy + y) * st + x0 + sx + x, s, st, s, s)
                          + (uint32_t)mv_cost(4 * (sx + x), 4 * (sy + y), O->mvpx, O->mvpy, lam);
            if (cc < bc) { bc = cc; bx = 4 * (sx + x); by = 4 * (sy + y); }
        }
    static const int rx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, ry[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    for (int step = 2; step >= 1; --step) {                       /* bc carries on: SAD + rate of the integer winner */
        int cx0 = bx, cy0 = by;
        for (int k = 0; k < 8; ++k) {
            int qx = cx0 + rx[k] * step, qy = cy0 + ry[k] * step;
            const uint8_t *pl = org_y(&g, (uint8_t *)planesO + (long)((qy & 3) * 4 + (qx & 3)) * g.bytes_y);
            uint32_t cc = ks265o_sad(T, pl + (long)(y0 + (qy >> 2)) * st + x0 + (qx >> 2), s, st, s, s) + (uint32_t)mv_cost(qx, qy, O->mvpx, O->mvpy, lam);
            if (cc < bc) { bc = cc; bx = qx; by = qy; }
        }
    }
    const uint8_t *po = org_y(&g, (uint8_t *)planesO + (long)((by & 3) * 4 + (bx & 3)) * g.bytes_y) + (long)(y0 + (by >> 2)) * st + x0 + (bx >> 2);
    for (int y = 0; y < s; ++y)
        for (int x = 0; x < s; ++x) avg[y * s + x] = (uint8_t)((kp[y * s + x] + po[(long)y * st + x] + 1) >> 1);
    uint32_t c2 = ks265o_had(S + (long)y0 * st + x0, avg, st, s, s, s) + (uint32_t)mv_cost(K->mvx, K->mvy, K->mvpx, K->mvpy, lam)
                  + (uint32_t)mv_cost(bx, by, O->mvpx, O->mvpy, lam) + rbits;
    c2 -= c2 >> BI_BIAS_SHIFT;
    if (c2 < o->cost) {
        o->cost = c2; o->inter_dir = BI_DIR_OF(3, i0, i1);
        if (keep1) { o->mvx = (int16_t)bx; o->mvy = (int16_t)by; } else { o->mv1x = (int16_t)bx; o->mv1y = (int16_t)by; }
    }
}

void kso_bi_decide(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0_, const uint8_t *planes1_, const kso_pu *pu0, const kso_pu *pu1,
                   kso_pu_b *pub)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y);
    long st = g.stride_y;
    int lam = cfg->lambda_q4;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            long cb = (long)(cy * g.ctu_cols + cx) * 85;
            for (int l = 0; l < 4; ++l)
                for (int py = 0; py < (1 << l); ++py)
                    for (int px = 0; px < (1 << l); ++px) {
                        int i = pu_index(l, px, py);
                        const kso_pu *a = &pu0[cb + i], *b = &pu1[cb + i];
                        kso_pu_b *o = &pub[cb + i];
                        /* several pictures per list (kso_set_mref): the two records are the lists' winners (kso_ref_pick: their costs hold the index bits), i0 / i1 their pictures */
                        const int i0 = g_mr ? g_mr->idx0[cb + i] : 0, i1 = g_mr ? g_mr->idx1[cb + i] : 0;
                        const uint8_t *const planes0 = MR_PL0(planes0_, i0), *const planes1 = MR_PL1(planes1_, i1);
                        const uint32_t rbits = g_mr ? (uint32_t)((lam * ref_idx_bits(i0, g_mr->n0)) >> 4) + (uint32_t)((lam * ref_idx_bits(i1, g_mr->n1)) >> 4) : 0u;   /* (each list's index rate as kso_ref_pick counted it) */
                        #define BI_DIR(d) ((uint32_t)(d) | (((d) & 1) ? (uint32_t)i0 << 4 : 0u) | (((d) & 2) ? (uint32_t)i1 << 6 : 0u))
                        o->mvx = a->mvx; o->mvy = a->mvy; o->mv1x = b->mvx; o->mv1y = b->mvy; o->cost = a->cost; o->inter_dir = BI_DIR(1);
                        if (a->cost == COST_INVALID) continue;
                        if (b->cost < o->cost) { o->cost = b->cost; o->inter_dir = BI_DIR(2); }
                        int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
                        const uint8_t *p0 = org_y(&g, (uint8_t *)planes0 + (long)((a->mvy & 3) * 4 + (a->mvx & 3)) * g.bytes_y) + (long)(y0 + (a->mvy >> 2)) * st + x0 + (a->mvx >> 2);
                        const uint8_t *p1 = org_y(&g, (uint8_t *)planes1 + (long)((b->mvy & 3) * 4 + (b->mvx & 3)) * g.bytes_y) + (long)(y0 + (b->mvy >> 2)) * st + x0 + (b->mvx >> 2);
                        uint8_t avg[64 * 64];
                        for (int y = 0; y < s; ++y)
                            for (int x = 0; x < s; ++x) avg[y * s + x] = (uint8_t)((p0[(long)y * st + x] + p1[(long)y * st + x] + 1) >> 1);
                        uint32_t d = ks265o_had(S + (long)y0 * st + x0, avg, st, s, s, s);
                        uint32_t c = d + (uint32_t)mv_cost(a->mvx, a->mvy, a->mvpx, a->mvpy, lam) + (uint32_t)mv_cost(b->mvx, b->mvy, b->mvpx, b->mvpy, lam) + rbits;
                        c -= c >> BI_BIAS_SHIFT;
                        if (c < o->cost) { o->cost = c; o->inter_dir = BI_DIR(3); }
                        if (cfg->bi_refine == 1) bi_refine_pu(cfg, &g, S, planes0, planes1, rbits, i0, i1, cx, cy, l, px, py, a, b, o);   /* (2: after the CU decision, kso_bi_refine_chosen) */
                    }
        }
}

/* cfg->bi_refine == 2 (round 5): the joint refinement AFTER the CU decision, for the CUs it chose - one refinement per picture area instead of one per quadtree level
 * (kso_bi_decide then only pairs the lists' winners; the decision sees unrefined pairs).  A 2N x 2N inter CU's record and its 8 x 8 blocks take the refined pair if it is
 * cheaper than what the CU had; CUs in halves (cfg->part) and intra CUs stay as they are.  Runs in front of the merge pass. */
void kso_bi_refine_chosen(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0_, const uint8_t *planes1_, const kso_pu *pu0, const kso_pu *pu1,
                          kso_pu_b *pub, kso_cu8 *cu8)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y);
    const int lam = cfg->lambda_q4, w8 = cfg->width / 8;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            const long cb = (long)(cy * g.ctu_cols + cx) * 85;
            for (int l = 0; l < 4; ++l)
                for (int py = 0; py < (1 << l); ++py)
                    for (int px = 0; px < (1 << l); ++px) {
                        const int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, i = pu_index(l, px, py);
                        if (x0 >= cfg->width || y0 >= cfg->height) continue;
                        const kso_cu8 *c0 = &cu8[(long)(y0 / 8) * w8 + x0 / 8];
                        if (c0->pred_mode != 0 || c0->log2_cu != 6 - l) continue;           /* another size, an intra CU, or a CU in halves (bits 4..5) */
                        const kso_pu *a = &pu0[cb + i], *b = &pu1[cb + i];
                        kso_pu_b *o = &pub[cb + i];
                        if (a->cost == COST_INVALID) continue;
                        const int i0 = g_mr ? g_mr->idx0[cb + i] : 0, i1 = g_mr ? g_mr->idx1[cb + i] : 0;
                        const uint32_t rbits = g_mr ? (uint32_t)((lam * ref_idx_bits(i0, g_mr->n0)) >> 4) + (uint32_t)((lam * ref_idx_bits(i1, g_mr->n1)) >> 4) : 0u;
                        const uint32_t before = o->cost;
                        bi_refine_pu(cfg, &g, S, MR_PL0(planes0_, i0), MR_PL1(planes1_, i1), rbits, i0, i1, cx, cy, l, px, py, a, b, o);
                        if (o->cost == before) continue;
                        for (int by = 0; by < s / 8; ++by)
                            for (int bx = 0; bx < s / 8; ++bx) {
                                kso_cu8 *c = &cu8[(long)(y0 / 8 + by) * w8 + x0 / 8 + bx];
                                c->mvx = o->mvx; c->mvy = o->mvy; c->mv1x = o->mv1x; c->mv1y = o->mv1y; c->inter_dir = (uint8_t)o->inter_dir;
                            }
                    }
        }
}

/* cfg->part in B pictures (round 5; -preset veryslow = config 5 codes hierarchical B with part 1): the same frame-parallel form as rect_eval, on MOTIONS instead of vectors.  A half
 * may take the motion the bi-predictive decision left for the CU itself or for one of the half's two quarter-size PUs - direction and vector(s) together - priced like the
 * decision prices a PU: Hadamard cost of the half's prediction (one list: the interpolated samples; both: the rounded average of the two 8-bit predictions) + the rate of
 * every vector used against the CU's predictor of that list, a bi-predictive half at 31 / 32 (BI_BIAS_SHIFT); first strict minimum in the order CU, first quarter, second
 * quarter.  Considered when a half's motion differs from the CU's; cost = both halves + lambda x PART_BITS. */
typedef struct { const kso_frame_cfg *cfg; const kso_frame_geom *g; const uint8_t *S, *planes0, *planes1; const kso_pu *pu0, *pu1; } rect_ctx_b;
static uint32_t rect_half_cost_b(const rect_ctx_b *rc, int x0, int y0, int w, int h, const kso_pu_b *M, const kso_pu *a, const kso_pu *b)
{
    const long st = rc->g->stride_y;
    const int dir = (int)(M->inter_dir & 3), lam = rc->cfg->lambda_q4;
    const uint8_t *p0 = org_y(rc->g, (uint8_t *)MR_PL0(rc->planes0, (M->inter_dir >> 4) & 3) + (long)((M->mvy & 3) * 4 + (M->mvx & 3)) * rc->g->bytes_y) + (long)(y0 + (M->mvy >> 2)) * st + x0 + (M->mvx >> 2);
    const uint8_t *p1 = org_y(rc->g, (uint8_t *)MR_PL1(rc->planes1, (M->inter_dir >> 6) & 3) + (long)((M->mv1y & 3) * 4 + (M->mv1x & 3)) * rc->g->bytes_y) + (long)(y0 + (M->mv1y >> 2)) * st + x0 + (M->mv1x >> 2);
    const uint8_t *S = rc->S + (long)y0 * st + x0;
    if (dir == 1) return ks265o_had(S, p0, st, st, h, w) + (uint32_t)mv_cost(M->mvx, M->mvy, a->mvpx, a->mvpy, lam);
    if (dir == 2) return ks265o_had(S, p1, st, st, h, w) + (uint32_t)mv_cost(M->mv1x, M->mv1y, b->mvpx, b->mvpy, lam);
    uint8_t avg[64 * 64];
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) avg[y * w + x] = (uint8_t)((p0[(long)y * st + x] + p1[(long)y * st + x] + 1) >> 1);
    uint32_t c = ks265o_had(S, avg, st, w, h, w) + (uint32_t)mv_cost(M->mvx, M->mvy, a->mvpx, a->mvpy, lam) + (uint32_t)mv_cost(M->mv1x, M->mv1y, b->mvpx, b->mvpy, lam);
    c -= c >> BI_BIAS_SHIFT;
    return c;
}
static int same_motion_b(const kso_pu_b *m, const kso_pu_b *n)
{
    const int d = (int)(m->inter_dir & 3);
    if (m->inter_dir != n->inter_dir) return 0;                     /* direction and the pictures of the lists used (unused indices are 0) */
    if ((d & 1) && (m->mvx != n->mvx || m->mvy != n->mvy)) return 0;
    if ((d & 2) && (m->mv1x != n->mv1x || m->mv1y != n->mv1y)) return 0;
    return 1;
}
static void rect_eval_b(const rect_ctx_b *rc, const kso_pu_b *cp, long cb, int cx, int cy, int l, int px, int py, rect_rec *out)
{
    const int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, idx = pu_index(l, px, py);
    const kso_pu_b *P = &cp[idx];
    const kso_pu *a = &rc->pu0[cb + idx], *b = &rc->pu1[cb + idx];
    out->cost[0] = out->cost[1] = COST_INVALID;
    if (l > 2 || P->cost == COST_INVALID) return;
    for (int o = 0; o < 2; ++o) {
        uint64_t tot = (uint64_t)((rc->cfg->lambda_q4 * PART_BITS) >> 4);
        int moved = 0;
        for (int hf = 0; hf < 2; ++hf) {
            const int hx = o ? x0 + hf * (s / 2) : x0, hy = o ? y0 : y0 + hf * (s / 2), w = o ? s / 2 : s, h = o ? s : s / 2;
            const int i0 = pu_index(l + 1, px * 2 + (o ? hf : 0), py * 2 + (o ? 0 : hf)), i1 = pu_index(l + 1, px * 2 + (o ? hf : 1), py * 2 + (o ? 1 : hf));
            const kso_pu_b *cand[2] = {&cp[i0], &cp[i1]}, *bm = P;
            uint32_t best = rect_half_cost_b(rc, hx, hy, w, h, P, a, b);
            for (int k = 0; k < 2; ++k) {
                if (cand[k]->cost == COST_INVALID) continue;
                const uint32_t c = rect_half_cost_b(rc, hx, hy, w, h, cand[k], a, b);
                if (c < best) { best = c; bm = cand[k]; }
            }
            moved |= !same_motion_b(bm, P);
            out->mv[o][hf][0] = bm->mvx; out->mv[o][hf][1] = bm->mvy; out->mv1[o][hf][0] = bm->mv1x; out->mv1[o][hf][1] = bm->mv1y; out->dir[o][hf] = (uint8_t)bm->inter_dir;
            tot += best;
        }
        if (moved) out->cost[o] = tot > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)tot;
    }
}
static uint32_t decide_node_b(const kso_frame_cfg *cfg, const rect_ctx_b *rc, long cb, const kso_pu_b *cp, const uint32_t *icost, int cx, int cy, int l, int px, int py, uint8_t *split, uint8_t *use_intra,
                              uint8_t *part, rect_rec *rect)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    if (x0 >= cfg->width || y0 >= cfg->height) return 0;
    int idx = pu_index(l, px, py);
    uint32_t inter = cp[idx].cost;
    part[idx] = 0;
    if (rc && l < 3) {
        rect_eval_b(rc, cp, cb, cx, cy, l, px, py, &rect[idx]);
        for (int o = 0; o < 2; ++o) if (rect[idx].cost[o] < inter) { inter = rect[idx].cost[o]; part[idx] = (uint8_t)(o + 1); }
    }
    uint32_t own = node_own_cost(cfg, inter, icost, idx, l, use_intra);
    if (l == 3) { split[idx] = 0; return own; }
    uint64_t sum = (uint64_t)((cfg->lambda_q4 * g_split_bits_b) >> 4);        /* (g_split_bits_b: experiment hook; SPLIT_BITS_B = the pipeline's value) */
    for (int k = 0; k < 4; ++k) sum += decide_node_b(cfg, rc, cb, cp, icost, cx, cy, l + 1, px * 2 + (k & 1), py * 2 + (k >> 1), split, use_intra, part, rect);
    if (own != COST_INVALID && (uint64_t)own <= sum) { split[idx] = 0; return own; }
    split[idx] = 1;
    return sum > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)sum;
}
static void emit_node_b(const kso_frame_cfg *cfg, const kso_pu_b *cp, const uint8_t *imode, int cx, int cy, int l, int px, int py, const uint8_t *split, const uint8_t *use_intra,
                        const uint8_t *part, const rect_rec *rect, kso_cu8 *cu8)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, w8 = cfg->width / 8;
    if (x0 >= cfg->width || y0 >= cfg->height) return;
    int idx = pu_index(l, px, py);
    if (l < 3 && split[idx]) {
        for (int k = 0; k < 4; ++k) emit_node_b(cfg, cp, imode, cx, cy, l + 1, px * 2 + (k & 1), py * 2 + (k >> 1), split, use_intra, part, rect, cu8);
        return;
    }
    if (use_intra[idx]) { emit_intra(cfg, x0, y0, s, imode[idx], l, cu8); return; }
    const int pm = part[idx];
    for (int by = 0; by < s / 8; ++by)
        for (int bx = 0; bx < s / 8; ++bx) {
            kso_cu8 *c = &cu8[(long)(y0 / 8 + by) * w8 + x0 / 8 + bx];
            c->mvx = cp[idx].mvx; c->mvy = cp[idx].mvy; c->mv1x = cp[idx].mv1x; c->mv1y = cp[idx].mv1y;
            c->log2_cu = (uint8_t)(6 - l); c->cbf = 0; c->pred_mode = 0; c->inter_dir = (uint8_t)cp[idx].inter_dir;
            if (pm) {
                const int hf = pm == 1 ? by >= s / 16 : bx >= s / 16;
                c->mvx = rect[idx].mv[pm - 1][hf][0]; c->mvy = rect[idx].mv[pm - 1][hf][1]; c->mv1x = rect[idx].mv1[pm - 1][hf][0]; c->mv1y = rect[idx].mv1[pm - 1][hf][1];
                c->inter_dir = rect[idx].dir[pm - 1][hf]; c->log2_cu = (uint8_t)((6 - l) | (pm << 4));
            }
        }
}
static void cu_decide_b_impl(const kso_frame_cfg *cfg, const rect_ctx_b *rc, const kso_pu_b *pub, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            const long cb = (long)(cy * g.ctu_cols + cx) * 85;
            uint8_t split[85], use_intra[85], part[85];
            rect_rec rect[21];
            memset(split, 0, sizeof split); memset(use_intra, 0, sizeof use_intra); memset(part, 0, sizeof part);
            decide_node_b(cfg, rc, cb, pub + cb, icost ? icost + cb : NULL, cx, cy, 0, 0, 0, split, use_intra, part, rect);
            emit_node_b(cfg, pub + cb, imode ? imode + cb : NULL, cx, cy, 0, 0, 0, split, use_intra, part, rect, cu8);
        }
}
void kso_cu_decide_b_ii(const kso_frame_cfg *cfg, const kso_pu_b *pub, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8) { cu_decide_b_impl(cfg, NULL, pub, icost, imode, cu8); }
/* the CU decision of a B picture with cfg->part: pu0 / pu1 = the two uni-directional searches' records (the predictors the rates are counted against) */
void kso_cu_decide_part_b(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0, const uint8_t *planes1, const kso_pu *pu0, const kso_pu *pu1, const kso_pu_b *pub,
                          const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const rect_ctx_b rc = {cfg, &g, org_y(&g, src.y), planes0, planes1, pu0, pu1};
    cu_decide_b_impl(cfg, cfg->part ? &rc : NULL, pub, icost, imode, cu8);
}
void kso_cu_decide_b(const kso_frame_cfg *cfg, const kso_pu_b *pub, kso_cu8 *cu8) { kso_cu_decide_b_ii(cfg, pub, NULL, NULL, cu8); }

/* key picture stand-in for the (out-of-scope) intra path: largest CU in {32,16,8} that fits, flat prediction */
void kso_cu_flat_intra(const kso_frame_cfg *cfg, kso_cu8 *cu8)
{
    int w8 = cfg->width / 8, h8 = cfg->height / 8;
    for (int by = 0; by < h8; ++by)
        for (int bx = 0; bx < w8; ++bx) {
            int lg = 3;
            for (int t = 5; t > 3; --t) {
                int n = 1 << (t - 3), ax = bx / n * n, ay = by / n * n;
                if (ax + n <= w8 && ay + n <= h8) { lg = t; break; }
            }
            kso_cu8 *c = &cu8[(long)by * w8 + bx];
            c->mvx = 0; c->mvy = 0; c->mv1x = 0; c->mv1y = 0; c->log2_cu = (uint8_t)lg; c->cbf = 0; c->pred_mode = 1; c->inter_dir = 0;
        }
}

/* ------------------------------------------------------------------ Stage D: reconstruct() enc@0x481da0 */
static int chroma_qp(int qp)
{
    static const int tab[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
    return qp < 30 ? qp : (qp >= 44 ? qp - 6 : tab[qp - 30]);
}

/* residual -> fwd transform -> quant -> dequant -> inverse -> recon for one NxN TU; returns cbf */
/* scan of a TU's coefficients (H.265 7.4.9.11): intra 4x4 / 8x8 luma and 4x4 chroma follow the prediction mode, everything else is diagonal */
static int tu_scan_idx(int intra, int mode, int n, int is_chroma)
{
    if (!intra || !(n == 4 || (n == 8 && !is_chroma))) return 0;
    return (mode >= 6 && mode <= 14) ? 2 : (mode >= 22 && mode <= 30) ? 1 : 0;
}
/* sdh: the postQuant seam (postQuant enc@0x4ace80) - after the quantiser, before dequantisation: sign-data hiding with the TU's scan */
/* cfg->rdo = K > 0: coefficient-group pruning at the postQuant seam - the sub-block decision of HM-lineage RDOQ (rdoQuant enc@0x4aac50 is closed code; its tables
 * estBitRdoq enc@0x46a8a0 are pinned, this is the frame-parallel form with STATIC bit costs).  A 4x4 group of levels is kept only if the distortion its levels remove
 * outweighs lambda_mode x K / 4 x their bits:
 *   gain  = sum over the group's levels of c^2 - (c - d)^2   (c = transform coefficient, d = dequantised level; the forward transform scales by 2^(7 - log2 N), so
 *           the pixel-domain SSE is gain >> 2 (7 - log2 N))
 *   bits  = sum of rdo_level_q2(|level|) + 10 + (16 - levels)   in quarter bits: significance + greater-1/2 flags + sign (3.5 bits for +-1, 5 for +-2, then the
 *           exp-Golomb remainder), coded_sub_block_flag, the significance flags of the zeros around
 *   kept iff (gain >> 2 (7 - log2 N)) << 12 > lambda_q4^2 x bits x K        (lambda_mode = (lambda_q4 / 16)^2; K = 4 is lambda x 1)
 * Inter TUs of every component, and the intra CUs of P / B pictures; key pictures keep every level (their quality is inherited by the whole GOP: measured with
 * tools/rd_eval.py, pruning them costs 3 % at equal PSNR).  Runs before sign-data hiding, which then works on the pruned levels. */
/* EXPERIMENT: the reference's own rdoQuant (oracle/ks265_rdoq_ref.c, pinned) at the postQuant seam with STATIC bit tables - what a
 * frame-parallel port could use, since the adaptive context states of the entropy coder do not exist when the device quantises.  Off unless kso_experiment_rdoq is
 * called (tools/rd_eval.py --rdoq); the product's specification is the code below it. */
#include "ks265_rdoq_ref.h"
#include <math.h>
static const int32_t *g_rq_T;            /* [log2 - 2][chroma][180] */
static int g_rq_mode, g_rq_mult[4] = {256, 256, 90, 90};
void kso_experiment_rdoq(const int32_t *T, int mode, const int *mult) { g_rq_T = T; g_rq_mode = mode; if (mult) memcpy(g_rq_mult, mult, sizeof g_rq_mult); }
static int rdo_level_q2(int a) { return a == 1 ? 14 : a == 2 ? 20 : 26 + 8 * (31 - __builtin_clz((unsigned)(a - 1))); }
static int code_tu(const uint8_t *org, int so, const uint8_t *pred /*packed n*/, int n, int qp, int intra, int16_t *lvl, int lstride,
                   uint8_t *rec, int rstride, int sdh, int scan_idx, int decimate, int rdo, int lambda_q4)
{
    int16_t res[32 * 32], coef[32 * 32], lv[32 * 32], du[32 * 32], dq[32 * 32], tmp[32 * 32];
    int log2n = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5, idx = log2n - 1;   /* DCT table index (DST4 is intra-luma-4x4 only) */
    ks265o_calc_residual(res, org, pred, so, n, n, n);
    ks265o_fwd_transform(idx, res, coef, n, n, tmp);
    ks265o_quant_param p;
    ks265o_get_base_quant_param(qp, intra == 1 ? 2 : 0, &p);        /* intra: 1 = a CU of an I slice (offset 171), 2 = an intra CU of a P / B slice (the slice's offset, 85) */
    int qbits = p.qbits - log2n;
    int nz = ks265o_quant(coef, lv, n, p.scale, p.offF << (qbits - 9), qbits, du, n);
    /* cfg->decimate = K > 0 (inter LUMA TUs): a block whose levels are all +-1 and at most 2K (8x8), 3K (16x16), 4K (32x32) of them is not worth its bits - the
     * levels are dropped, the block becomes prediction only (x264-lineage coefficient decimation; where the reference does this is inside its closed RD code).
     * Chroma is left alone: measured with this oracle + the stream writer over 19 P pictures (416x240, 832x480; qp 27), K = 2 on chroma saved 1 % of the bytes
     * and cost 2.4 - 3.2 dB of chroma PSNR (nearly all chroma levels are +-1); luma only: -10 % / -18 % of the bytes for -0.31 / -0.37 dB PSNR-Y (one QP step
     * is -17 % for -0.71 dB). */
    if (g_rq_T && (rdo > 0 || rdo == -1) && (intra != 1 || (g_rq_mode & 4)) && !(intra == 2 && (g_rq_mode & 8))) {   /* mode bit 8 (round 6, the product's -rdoq 1): inter CUs only - the intra CUs of P / B pictures keep the seam */
        const int chroma = rdo == -1, per = qp / 6;
        uint16_t sm[64];
        for (int i = 0; i < n * n; ++i) { const int c = coef[i], a = c < 0 ? -c : c; int q = (int)(((int64_t)a * p.scale + (1ll << (qbits - 1))) >> qbits); if (q > 32767) q = 32767; lv[i] = (int16_t)(c < 0 ? -q : q); }
        const int last = kso_rdoq_scan_flags(lv, log2n, scan_idx, sm);
        nz = 0;
        if (last >= 0) {
            const double lam = 0.85 * pow(2.0, (qp - 12) / 3.0);
            nz = kso_ref_rdo_quant(lv, coef, log2n, scan_idx, chroma, p.dq, per, (int64_t)(g_rq_mult[2 * chroma + 1] * lam + 0.5), (int64_t)(g_rq_mult[2 * chroma] * lam + 0.5),
                                   g_rq_T + ((log2n - 2) * 2 + chroma) * 180, 1, last, sm, 1, sdh, NULL, NULL);
        }
        rdo = 0; sdh = 0; decimate = 0;
    }
    if (rdo < 0) rdo = 0;
    if (decimate > 0 && !intra && nz > 0 && nz <= decimate * (log2n - 1)) {
        int mx = 0;
        for (int i = 0; i < n * n; ++i) { const int a = lv[i] < 0 ? -lv[i] : lv[i]; if (a > mx) mx = a; }
        if (mx <= 1) { memset(lv, 0, sizeof(int16_t) * (size_t)(n * n)); nz = 0; }
    }
    /* (luma only: the callers pass rdo = 0 for chroma - pruning chroma groups with K = 4 saved 2.1 % of the bytes of the hierarchical-B clip and nothing on IPPP, for
     *  2 dB of chroma PSNR, 45.8 -> 43.8; tools/rd_eval.py, round 3) */
    if (rdo > 0 && intra != 1 && nz > 0) {
        const int sh2 = 2 * (7 - log2n), dshift = log2n - 1;
        const int64_t lam2 = (int64_t)lambda_q4 * lambda_q4;
        for (int gy = 0; gy < n; gy += 4)
            for (int gx = 0; gx < n; gx += 4) {
                int64_t gain = 0; int bits = 0, cnt = 0;
                for (int y = gy; y < gy + 4; ++y)
                    for (int x = gx; x < gx + 4; ++x) {
                        const int l = lv[y * n + x];
                        if (!l) continue;
                        const int c = coef[y * n + x];
                        int d = (l * p.dq + (1 << (dshift - 1))) >> dshift;
                        d = d < -32768 ? -32768 : d > 32767 ? 32767 : d;
                        gain += (int64_t)d * (2 * c - d);                  /* c^2 - (c - d)^2 */
                        bits += rdo_level_q2(l < 0 ? -l : l); ++cnt;
                    }
                if (!cnt) continue;
                bits += 10 + (16 - cnt);
                if (((gain >> sh2) << 12) <= lam2 * bits * rdo) {
                    for (int y = gy; y < gy + 4; ++y) for (int x = gx; x < gx + 4; ++x) lv[y * n + x] = 0;
                    nz -= cnt;
                }
            }
    }
    if (sdh && nz > 1) nz = ks265o_sign_bit_hiding(lv, coef, du, n, log2n, scan_idx);
    for (int y = 0; y < n; ++y) memcpy(lvl + (long)y * lstride, lv + y * n, sizeof(int16_t) * (size_t)n);
    if (!nz) {
        for (int y = 0; y < n; ++y) memcpy(rec + (long)y * rstride, pred + y * n, (size_t)n);
        return 0;
    }
    int shift = log2n - 1;
    ks265o_dequant(lv, dq, n, p.dq, 1 << (shift - 1), shift, n - 1, n - 1);
    ks265o_inv_transform(idx, dq, rec, pred, n, rstride, n, tmp, n - 1, n - 1);
    return 1;
}

/* 14-bit prediction block of one list for bi-prediction (normative separable order, no -8192 offset in this codec):
 * integer position: sample << 6 (InterpolateCopy8to16); one fraction: the raw tap sum (interp*8to16); two: H 8to16 then V 16to16 */
static void pred14_luma(const uint8_t *ref0 /*sample (0,0)*/, long st, int x0, int y0, int n, int mvx, int mvy, int16_t *out /*n x n*/)
{
    const uint8_t *p = ref0 + (long)(y0 + (mvy >> 2)) * st + x0 + (mvx >> 2);
    int fx = mvx & 3, fy = mvy & 3;
    if (!fx && !fy) { for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) out[y * n + x] = (int16_t)(p[(long)y * st + x] << 6); }
    else if (!fy) ks265o_interp_luma_hor_8to16(out, n, p, (int)st, n, n, fx);
    else if (!fx) ks265o_interp_luma_ver_8to16(out, n, p, (int)st, n, n, fy);
    else {
        int16_t tmp[32 * 39];
        ks265o_interp_luma_hor_8to16(tmp, n, p - 3 * st, (int)st, n, n + 7, fx);
        ks265o_interp_luma_ver_16to16(out, n, tmp + 3 * n, n, n, n, fy);
    }
}
static void pred14_chroma(const uint8_t *ref0, long st, int xc, int yc, int n, int mvx, int mvy, int16_t *out)
{
    const uint8_t *p = ref0 + (long)(yc + (mvy >> 3)) * st + xc + (mvx >> 3);
    int fx = mvx & 7, fy = mvy & 7;
    if (!fx && !fy) { for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) out[y * n + x] = (int16_t)(p[(long)y * st + x] << 6); }
    else if (!fy) ks265o_interp_chroma_hor_8to16(out, n, p, (int)st, n, n, fx);
    else if (!fx) ks265o_interp_chroma_ver_8to16(out, n, p, (int)st, n, n, fy);
    else {
        int16_t tmp[16 * 19];
        ks265o_interp_chroma_hor_8to16(tmp, n, p - st, (int)st, n, n + 3, fx);
        ks265o_interp_chroma_ver_16to16(out, n, tmp + n, n, n, n, fy);
    }
}

/* ------------------------------------------------------------------ QP per CTU (round 4: cu_qp_delta with the quantisation group = the CTU; adaptive quantisation)
 * kso_set_qp_map(map): one QP per CTU in raster order for everything coded after the call (NULL: the slice QP everywhere - the state every other test runs in).  The
 * residual of a CTU is quantised with its map entry (chroma through the table).  What the DECODER takes as a CU's QpY is not always that value (H.265 8.6.1 with
 * Log2MinCuQpDeltaSize = CtbLog2SizeY): cu_qp_delta is sent with the first coded residual of the CTU, so the CUs in front of it (z-order, no residual) keep the predicted
 * QP = the QpY of the previous CTU's last CU (the slice QP at the start of every CTU row: entropy_coding_sync) - the deblocking filter reads those (kso_effective_qp). */
static const int8_t *g_qpmap;
void kso_set_qp_map(const int8_t *map) { g_qpmap = map; }
static int ctu_qp(const kso_frame_cfg *cfg, int x0, int y0) { return g_qpmap ? g_qpmap[(y0 >> 6) * ((cfg->width + 63) >> 6) + (x0 >> 6)] : cfg->qp; }
/* QpY of every 8x8 block as the decoder derives it; eff: w8 * h8 bytes */
void kso_effective_qp(const kso_frame_cfg *cfg, const kso_cu8 *cu8, uint8_t *eff)
{
    const int W = cfg->width, H = cfg->height, w8 = W / 8, h8 = H / 8, cols = (W + 63) >> 6, rows = (H + 63) >> 6;
    for (int cy = 0; cy < rows; ++cy) {
        int prev = cfg->qp;
        for (int cx = 0; cx < cols; ++cx) {
            const int want = g_qpmap ? g_qpmap[cy * cols + cx] : cfg->qp;
            int cur = prev;                                              /* until the CTU's first coded residual */
            for (int z = 0; z < 64; ++z) {
                const int lx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ly = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
                const int bx = cx * 8 + lx, by = cy * 8 + ly;
                if (bx >= w8 || by >= h8) continue;
                const kso_cu8 *c = &cu8[(long)by * w8 + bx];
                const int n8 = 1 << (CU_LOG2(c) - 3);
                if (!(lx & (n8 - 1)) && !(ly & (n8 - 1)) && cur != want) {      /* a CU starts here: does it carry residual? */
                    int any = 0;
                    for (int yy = 0; yy < n8 && !any; ++yy)
                        for (int xx = 0; xx < n8; ++xx) if (by + yy < h8 && bx + xx < w8 && cu8[(long)(by + yy) * w8 + bx + xx].cbf) { any = 1; break; }
                    if (any) cur = want;
                }
                eff[(long)by * w8 + bx] = (uint8_t)cur;
            }
            prev = cur;
        }
    }
}

/* list 0 may hold several reference pictures (multi-reference P pictures, -ref / -ref0): the CU's picture is refs0[inter_dir >> 4] */
#ifndef TU_SPLIT_K
#define TU_SPLIT_K 4
#endif
static int g_tu_split_k = TU_SPLIT_K;                    /* (experiment hook of tools/rd_eval.py: kso_experiment_tu_split_k) */
void kso_experiment_tu_split_k(int k) { g_tu_split_k = k; }
static void reconstruct_impl(const kso_frame_cfg *cfg, kso_pic src, const kso_pic *refs0, const uint8_t *const *planes0, kso_pic ref1_, const uint8_t *planes1_,
                             kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int W = cfg->width, H = cfg->height, w8 = W / 8, h8 = H / 8;
    long sy = g.stride_y, sc = g.stride_c;
    for (int by = 0; by < h8; ++by)
        for (int bx = 0; bx < w8; ++bx) {
            kso_cu8 *c = &cu8[(long)by * w8 + bx];
            const int qp = ctu_qp(cfg, bx * 8, by * 8), qpc = chroma_qp(qp);
            int n8 = 1 << (CU_LOG2(c) - 3);
            /* cfg->tu_inter (round 5; -intertu 1 = tuInter, qy265enc.h:133: the residual quadtree of inter CUs one level deep; the reference decides it inside its closed RD code,
             * tuDecision enc@0x4825a0): a 2Nx2N inter CU of 32 or 16 samples is coded with FOUR transform units when its luma residual is concentrated in part of it - the
             * quarters' residual SADs under the CU's final motion: max > TU_SPLIT_K x min + (N / 2)^2 - so that the quiet quarters cost a coded-block flag instead of sharing a
             * large transform's spread.  Decided at the CU's first block; log2_cu bits 4..5 = 3 mark it (2Nx2N, four TUs) for every later reader */
            if (cfg->tu_inter && c->pred_mode == 0 && !CU_PART(c) && (n8 == 2 || n8 == 4) && !(bx % n8) && !(by % n8)) {
                const int N = n8 * 8, cx0 = bx * 8, cy0 = by * 8, d = c->inter_dir & 3;
                uint8_t pc[32 * 32];
                const kso_pic rf = refs0[(c->inter_dir >> 4) & 3];
                const uint8_t *pl0 = planes0[(c->inter_dir >> 4) & 3];
                const kso_pic rf1 = g_mr ? g_mr->pic1[(c->inter_dir >> 6) & 3] : ref1_;
                const uint8_t *pl1 = g_mr ? g_mr->planes1[(c->inter_dir >> 6) & 3] : planes1_;
                if (d == 3) {
                    int16_t a0[32 * 32], a1[32 * 32];
                    pred14_luma(org_y(&g, rf.y), sy, cx0, cy0, N, c->mvx, c->mvy, a0);
                    pred14_luma(org_y(&g, rf1.y), sy, cx0, cy0, N, c->mv1x, c->mv1y, a1);
                    ks265o_default_weighted_bi(pc, a0, a1, N, N, N, N);
                } else {
                    const uint8_t *pb = d == 2 ? pl1 : pl0;
                    const int ux = d == 2 ? c->mv1x : c->mvx, uy = d == 2 ? c->mv1y : c->mvy;
                    const uint8_t *pl = org_y(&g, (uint8_t *)pb + (long)((uy & 3) * 4 + (ux & 3)) * g.bytes_y) + (long)(cy0 + (uy >> 2)) * sy + cx0 + (ux >> 2);
                    for (int y = 0; y < N; ++y) memcpy(pc + y * N, pl + (long)y * sy, (size_t)N);
                }
                int q[4] = {0, 0, 0, 0};
                const uint8_t *So = org_y(&g, src.y) + (long)cy0 * sy + cx0;
                for (int y = 0; y < N; ++y)
                    for (int x = 0; x < N; ++x) q[(y >= N / 2) * 2 + (x >= N / 2)] += iabs_((int)So[(long)y * sy + x] - (int)pc[y * N + x]);
                const int mn = imin(imin(q[0], q[1]), imin(q[2], q[3])), mx = imax(imax(q[0], q[1]), imax(q[2], q[3]));
                if (mx > g_tu_split_k * mn + (N / 2) * (N / 2))
                    for (int yy = 0; yy < n8; ++yy)
                        for (int xx = 0; xx < n8; ++xx) cu8[(long)(by + yy) * w8 + bx + xx].log2_cu |= 3 << 4;
            }
            int tu8 = imin(CU_PART(c) ? n8 >> 1 : n8, 4);       /* TU = min(CU, 32); a CU in two partitions (or 2Nx2N with split transform units): four TUs */
            if ((bx % tu8) || (by % tu8)) continue;             /* visit each TU once, at its top-left 8x8 block */
            if (c->pred_mode == 2) continue;                    /* an intra CU of a P / B picture: coded afterwards from reconstructed neighbours (kso_intra_inter_reconstruct) */
            int n = tu8 * 8, x0 = bx * 8, y0 = by * 8, intra = c->pred_mode == 1;
            int mvx = c->mvx, mvy = c->mvy, cbf = 0;
            uint8_t pred[32 * 32];
            /* luma */
            const int dir = intra ? 0 : (c->inter_dir & 3), mv1x = c->mv1x, mv1y = c->mv1y;
            const kso_pic ref = refs0[intra ? 0 : ((c->inter_dir >> 4) & 3)];
            const uint8_t *planes = planes0[intra ? 0 : ((c->inter_dir >> 4) & 3)];
            const kso_pic ref1 = g_mr ? g_mr->pic1[intra ? 0 : ((c->inter_dir >> 6) & 3)] : ref1_;      /* several list-1 pictures: kso_set_mref */
            const uint8_t *planes1 = g_mr ? g_mr->planes1[intra ? 0 : ((c->inter_dir >> 6) & 3)] : planes1_;
            if (intra) memset(pred, 128, sizeof pred);
            else if (dir == 3) {                                /* bi: DefaultWeightedBi_c enc@0x435160 on the two 14-bit predictions */
                int16_t a0[32 * 32], a1[32 * 32];
                pred14_luma(org_y(&g, ref.y), sy, x0, y0, n, mvx, mvy, a0);
                pred14_luma(org_y(&g, ref1.y), sy, x0, y0, n, mv1x, mv1y, a1);
                ks265o_default_weighted_bi(pred, a0, a1, n, n, n, n);
            } else {
                const uint8_t *pb = dir == 2 ? planes1 : planes;
                const int ux = dir == 2 ? mv1x : mvx, uy = dir == 2 ? mv1y : mvy;
                const uint8_t *pl = org_y(&g, (uint8_t *)pb + (long)((uy & 3) * 4 + (ux & 3)) * g.bytes_y) + (long)(y0 + (uy >> 2)) * sy + x0 + (ux >> 2);
                for (int y = 0; y < n; ++y) memcpy(pred + y * n, pl + (long)y * sy, (size_t)n);
            }
            cbf |= code_tu(org_y(&g, src.y) + (long)y0 * sy + x0, (int)sy, pred, n, qp, intra, lvl_y + (long)y0 * W + x0, W,
                           org_y(&g, recon.y) + (long)y0 * sy + x0, (int)sy, cfg->sdh, 0, cfg->decimate, cfg->rdo, cfg->lambda_q4);
            /* chroma: 4-tap 1/8-sample MC (interpChroma* enc@0x4111c0..), TU n/2 */
            int nc = n / 2, xc = x0 / 2, yc = y0 / 2;
            for (int comp = 0; comp < 2; ++comp) {
                const uint8_t *rp = org_c(&g, dir == 2 ? (comp ? ref1.v : ref1.u) : (comp ? ref.v : ref.u));
                if (intra) memset(pred, 128, sizeof pred);
                else if (dir == 3) {
                    int16_t a0[16 * 16], a1[16 * 16];
                    pred14_chroma(org_c(&g, comp ? ref.v : ref.u), sc, xc, yc, nc, mvx, mvy, a0);
                    pred14_chroma(org_c(&g, comp ? ref1.v : ref1.u), sc, xc, yc, nc, mv1x, mv1y, a1);
                    ks265o_default_weighted_bi(pred, a0, a1, nc, nc, nc, nc);
                } else {
                    const int ux = dir == 2 ? mv1x : mvx, uy = dir == 2 ? mv1y : mvy;
                    const uint8_t *p0 = rp + (long)(yc + (uy >> 3)) * sc + xc + (ux >> 3);
                    int fx = ux & 7, fy = uy & 7;
                    if (!fx && !fy) for (int y = 0; y < nc; ++y) memcpy(pred + y * nc, p0 + (long)y * sc, (size_t)nc);
                    else if (!fy) ks265o_interp_chroma_hor_8to8(pred, nc, p0, (int)sc, nc, nc, fx);
                    else if (!fx) ks265o_interp_chroma_ver_8to8(pred, nc, p0, (int)sc, nc, nc, fy);
                    else {
                        int16_t t16[16 * 19];
                        ks265o_interp_chroma_hor_8to16(t16, nc, p0 - sc, (int)sc, nc, nc + 3, fx);
                        ks265o_interp_chroma_ver_16to8(pred, nc, t16 + nc, nc, nc, nc, fy);
                    }
                }
                int16_t *lv = (comp ? lvl_v : lvl_u) + (long)yc * (W / 2) + xc;
                uint8_t *rc = org_c(&g, comp ? recon.v : recon.u) + (long)yc * sc + xc;
                const uint8_t *oc = org_c(&g, comp ? src.v : src.u) + (long)yc * sc + xc;
                if (code_tu(oc, (int)sc, pred, nc, qpc, intra, lv, W / 2, rc, (int)sc, cfg->sdh, 0, 0, (g_rq_mode & 2) ? -1 : 0, cfg->lambda_q4)) cbf |= 2 << comp;      /* no decimation and no group pruning of chroma: see code_tu */
            }
            for (int yy = 0; yy < tu8; ++yy)
                for (int xx = 0; xx < tu8; ++xx) cu8[(long)(by + yy) * w8 + bx + xx].cbf = (uint8_t)cbf;
        }
}

void kso_reconstruct(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const uint8_t *planes, kso_pic ref1, const uint8_t *planes1,
                     kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon)
{
    reconstruct_impl(cfg, src, &ref, &planes, ref1, planes1, cu8, lvl_y, lvl_u, lvl_v, recon);
}
void kso_reconstruct_mref(const kso_frame_cfg *cfg, kso_pic src, int nref, const kso_pic *refs, const uint8_t *const *planes, kso_cu8 *cu8, int16_t *lvl_y,
                          int16_t *lvl_u, int16_t *lvl_v, kso_pic recon)
{
    (void)nref;
    kso_pic none = {0, 0, 0};
    reconstruct_impl(cfg, src, refs, planes, none, 0, cu8, lvl_y, lvl_u, lvl_v, recon);
}

/* ------------------------------------------------------------------ Stage D2: skip pass (cfg->skip_rd; round 6)
 * The reference decides skip / merge on the distortion the coded block really has (skipFastDecision enc@0x486090, skipFullMergeDecision enc@0x482da0, tuDecisionSkipMerge
 * enc@0x482990 under processCuMdInter enc@0x485800: closed code).  The CU tree and the merge pass of this pipeline judge on Hadamard cost + rate, which cannot know what a
 * residual costs and what it buys.  This pass runs AFTER the reconstruction, where both sides of the comparison exist.  Every inter CU that carries residual:
 *   J_cur  = SSE(source, reconstruction) of Y + 4 (Cb + Cr)  +  lambda x (bits of the CU's levels + 6)
 *   J_skip = min over the first two DISTINCT merge candidates k (of A1 B1 B0 A0 B2 of H.265 8.5.3.2.3 + the zero vector, taken from the field the pass reads - as in the
 *            merge pass: no order between CUs) of  SSE(source, prediction with k's motion) of Y + 4 (Cb + Cr)  +  lambda x (1 + position of k) bits
 * and J_skip < J_cur makes the CU a 2Nx2N CU without residual carrying k's motion: levels cleared, reconstruction = the prediction (the exact one: bi-prediction from the
 * 14-bit intermediates).  Distortions are exact, bits estimates: levels as the coefficient-group pruning prices them (rdo_level_q2 + 10 + 16 - n per 4 x 4 group, quarter
 * bits).  lambda = 1.5 x lambda_mode = 1.5 (lambda_q4 / 16)^2: J x 1024 = SSE << 10 + (lambda_q4^2 x 24 >> 4) x quarter bits.  A CU whose adopted motion is no longer a merge
 * candidate when the slice is written is coded with explicit motion and rqt_root_cbf = 0.
 * What was measured on the way (tools/rd_eval.py --host, 832x480 pyramid, bytes at equal PSNR-Y): this rule - 1.9 % (B layers 1 / 2 / 3: - 3 / - 11 / - 15 %, P pictures: nothing
 * - the host runs it on B pictures only); the same test on every NODE of the quadtree, top-down, uniting CUs into one (the form first built): - 1.5 % with the same constants, of
 * which the unions contribute 0.1 %: the tree the Hadamard decision leaves is not what costs the bits, the residuals that survive in it are; all candidates instead of two:
 * + 0.05 %; the quadrants of a 64x64 CU as CUs of their own: 0.1 %. */
static int g_sp[8] = {24, 4, 4, 24, 16, 2, 0, 0};         /* syntax of a CU with residual, skip base, per position (quarter bits); lambda scale (1/16); weight of the chroma distortion (1/4); distinct candidates tried */
void kso_experiment_skip(const int *v) { if (v) memcpy(g_sp, v, sizeof g_sp); }
/* the exact 8-bit prediction of an n x n block (luma) or its n/2 x n/2 chroma blocks from one or two pictures, as reconstruct_impl forms it; n <= 32 */
static void sp_pred_comp(const kso_frame_geom *g, int comp, kso_pic r0, kso_pic r1, int dir, int x0, int y0, int n, int mvx, int mvy, int mv1x, int mv1y, uint8_t *pred /* packed n (luma) or n/2 */)
{
    const long st = comp ? g->stride_c : g->stride_y;
    const int m = comp ? n / 2 : n, xc = comp ? x0 / 2 : x0, yc = comp ? y0 / 2 : y0;
    const uint8_t *p0 = comp == 0 ? org_y(g, r0.y) : org_c(g, comp == 1 ? r0.u : r0.v);
    const uint8_t *p1 = (dir & 2) ? (comp == 0 ? org_y(g, r1.y) : org_c(g, comp == 1 ? r1.u : r1.v)) : NULL;
    if (dir == 3) {
        int16_t a0[32 * 32], a1[32 * 32];
        if (comp == 0) { pred14_luma(p0, st, xc, yc, m, mvx, mvy, a0); pred14_luma(p1, st, xc, yc, m, mv1x, mv1y, a1); }
        else { pred14_chroma(p0, st, xc, yc, m, mvx, mvy, a0); pred14_chroma(p1, st, xc, yc, m, mv1x, mv1y, a1); }
        ks265o_default_weighted_bi(pred, a0, a1, m, m, m, m);
        return;
    }
    const uint8_t *pb = dir == 2 ? p1 : p0;
    const int ux = dir == 2 ? mv1x : mvx, uy = dir == 2 ? mv1y : mvy;
    if (comp == 0) {
        const uint8_t *p = pb + (long)(yc + (uy >> 2)) * st + xc + (ux >> 2);
        const int fx = ux & 3, fy = uy & 3;
        if (!fx && !fy) for (int y = 0; y < m; ++y) memcpy(pred + y * m, p + (long)y * st, (size_t)m);
        else if (!fy) ks265o_interp_luma_hor_8to8(pred, m, p, (int)st, m, m, fx);
        else if (!fx) ks265o_interp_luma_ver_8to8(pred, m, p, (int)st, m, m, fy);
        else { int16_t t16[32 * 39]; ks265o_interp_luma_hor_8to16(t16, m, p - 3 * st, (int)st, m, m + 7, fx); ks265o_interp_luma_ver_16to8(pred, m, t16 + 3 * m, m, m, m, fy); }
    } else {
        const uint8_t *p = pb + (long)(yc + (uy >> 3)) * st + xc + (ux >> 3);
        const int fx = ux & 7, fy = uy & 7;
        if (!fx && !fy) for (int y = 0; y < m; ++y) memcpy(pred + y * m, p + (long)y * st, (size_t)m);
        else if (!fy) ks265o_interp_chroma_hor_8to8(pred, m, p, (int)st, m, m, fx);
        else if (!fx) ks265o_interp_chroma_ver_8to8(pred, m, p, (int)st, m, m, fy);
        else { int16_t t16[16 * 19]; ks265o_interp_chroma_hor_8to16(t16, m, p - st, (int)st, m, m + 3, fx); ks265o_interp_chroma_ver_16to8(pred, m, t16 + m, m, m, m, fy); }
    }
}
static uint64_t sp_sse(const uint8_t *a, long sa, const uint8_t *b, long sb, int n)
{
    uint64_t s = 0;
    for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) { const int d = (int)a[(long)y * sa + x] - (int)b[(long)y * sb + x]; s += (uint64_t)(d * d); }
    return s;
}
static int sp_level_bits(const int16_t *lv, long st, int n)             /* quarter bits of the levels of an n x n area of a level plane, per 4 x 4 group */
{
    int bits = 0;
    for (int gy = 0; gy < n; gy += 4)
        for (int gx = 0; gx < n; gx += 4) {
            int b = 0, cnt = 0;
            for (int y = gy; y < gy + 4; ++y) for (int x = gx; x < gx + 4; ++x) { const int l = lv[(long)y * st + x]; if (l) { b += rdo_level_q2(l < 0 ? -l : l); ++cnt; } }
            if (cnt) bits += b + 10 + (16 - cnt);
        }
    return bits;
}
void kso_skip_pass(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref0_, kso_pic ref1_, const kso_cu8 *cu_in, kso_cu8 *cu_out,
                   int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const int W = cfg->width, H = cfg->height, w8 = W / 8, h8 = H / 8;
    const long sy = g.stride_y, sc = g.stride_c;
    const int is_b = ref1_.y != NULL || (g_mr && g_mr->n1 > 0);
    const int64_t lam2 = ((int64_t)cfg->lambda_q4 * cfg->lambda_q4 * g_sp[3]) >> 4;
    memcpy(cu_out, cu_in, sizeof(kso_cu8) * (size_t)w8 * h8);
#pragma omp parallel for schedule(dynamic, 4)
    for (int by = 0; by < h8; ++by)
        for (int bx = 0; bx < w8; ++bx) {
            const kso_cu8 *c0 = &cu_in[(long)by * w8 + bx];
            if (c0->pred_mode != 0 || CU_LOG2(c0) < 3) continue;
            const int s = 1 << CU_LOG2(c0), n8 = s / 8, x = bx * 8, y = by * 8;
            if ((x & (s - 1)) || (y & (s - 1))) continue;                         /* a CU is handled at its first 8x8 block */
            int coded = 0;
            for (int yy = 0; yy < n8; ++yy) for (int xx = 0; xx < n8; ++xx) coded |= cu_in[(long)(by + yy) * w8 + bx + xx].cbf;
            if (!coded) continue;
            const uint8_t *So = org_y(&g, src.y) + (long)y * sy + x, *Su = org_c(&g, src.u) + (long)(y / 2) * sc + x / 2, *Sv = org_c(&g, src.v) + (long)(y / 2) * sc + x / 2;
            uint8_t *Ro = org_y(&g, recon.y) + (long)y * sy + x, *Ru = org_c(&g, recon.u) + (long)(y / 2) * sc + x / 2, *Rv = org_c(&g, recon.v) + (long)(y / 2) * sc + x / 2;
            const uint64_t dcur = sp_sse(So, sy, Ro, sy, s) + (((sp_sse(Su, sc, Ru, sc, s / 2) + sp_sse(Sv, sc, Rv, sc, s / 2)) * (uint64_t)g_sp[4]) >> 2);
            const int bits = sp_level_bits(lvl_y + (long)y * W + x, W, s) + sp_level_bits(lvl_u + (long)(y / 2) * (W / 2) + x / 2, W / 2, s / 2)
                             + sp_level_bits(lvl_v + (long)(y / 2) * (W / 2) + x / 2, W / 2, s / 2) + g_sp[0];
            const uint64_t jcur = (dcur << 10) + (uint64_t)(lam2 * bits);
            /* the candidates */
            const int cx = x >> 6, cy = y >> 6, ctb = cy * g.ctu_cols + cx;
            const int nx[5] = {x - 1, x + s - 1, x + s, x - 1, x - 1}, ny[5] = {y + s - 1, y - 1, y - 1, y + s, y - 1};   /* A1 B1 B0 A0 B2 */
            const int zc = z_of(x, y);
            uint64_t best = jcur; int bestk = -1, pos = 0; kso_cu8 bm = *c0, seen[6]; int nseen = 0;
            uint8_t PY[64 * 64], PU_[32 * 32], PV[32 * 32], BY[64 * 64], BU[32 * 32], BV[32 * 32];
            for (int k = 0; k < 6; ++k) {
                kso_cu8 m;
                if (k < 5) {
                    if (nx[k] < 0 || ny[k] < 0 || nx[k] >= W || ny[k] >= H) continue;
                    const int nctb = (ny[k] >> 6) * g.ctu_cols + (nx[k] >> 6);
                    if (nctb > ctb || (nctb == ctb && z_of(nx[k], ny[k]) >= zc)) continue;
                    m = cu_in[(long)(ny[k] >> 3) * w8 + (nx[k] >> 3)];
                    if (m.pred_mode != 0 || (m.log2_cu & 15) < 3) continue;
                } else { memset(&m, 0, sizeof m); m.inter_dir = (is_b && !(g_mr && g_mr->n1 == 0)) ? 3 : 1; }
                const int dir = m.inter_dir & 3;
                if (!g_mr) m.inter_dir = (uint8_t)dir;
                if ((dir & 1) && (x + (m.mvx >> 2) < -70 || x + (m.mvx >> 2) + s > W + 70 || y + (m.mvy >> 2) < -70 || y + (m.mvy >> 2) + s > H + 70)) continue;
                if ((dir & 2) && (x + (m.mv1x >> 2) < -70 || x + (m.mv1x >> 2) + s > W + 70 || y + (m.mv1y >> 2) < -70 || y + (m.mv1y >> 2) + s > H + 70)) continue;
                if (!(dir & 1)) { m.mvx = m.mvy = 0; }
                if (!(dir & 2)) { m.mv1x = m.mv1y = 0; }
                const int mypos = pos++;
                int rep = 0;
                for (int j = 0; j < nseen; ++j) rep |= seen[j].inter_dir == m.inter_dir && seen[j].mvx == m.mvx && seen[j].mvy == m.mvy && seen[j].mv1x == m.mv1x && seen[j].mv1y == m.mv1y;
                if (rep) continue;                                                   /* the same motion at a later position never wins */
                if (nseen >= g_sp[5]) break;                                         /* the first two distinct candidates */
                seen[nseen++] = m;
                const kso_pic r0 = g_mr ? g_mr->pic0[(m.inter_dir >> 4) & 3] : ref0_, r1 = g_mr ? g_mr->pic1[(m.inter_dir >> 6) & 3] : ref1_;
                const int q = s > 32 ? 32 : s;
                for (int oy = 0; oy < s; oy += q)
                    for (int ox = 0; ox < s; ox += q) {
                        uint8_t t[32 * 32];
                        sp_pred_comp(&g, 0, r0, r1, dir, x + ox, y + oy, q, m.mvx, m.mvy, m.mv1x, m.mv1y, t);
                        for (int yy = 0; yy < q; ++yy) memcpy(PY + (oy + yy) * s + ox, t + yy * q, (size_t)q);
                        sp_pred_comp(&g, 1, r0, r1, dir, x + ox, y + oy, q, m.mvx, m.mvy, m.mv1x, m.mv1y, t);
                        for (int yy = 0; yy < q / 2; ++yy) memcpy(PU_ + (oy / 2 + yy) * (s / 2) + ox / 2, t + yy * (q / 2), (size_t)(q / 2));
                        sp_pred_comp(&g, 2, r0, r1, dir, x + ox, y + oy, q, m.mvx, m.mvy, m.mv1x, m.mv1y, t);
                        for (int yy = 0; yy < q / 2; ++yy) memcpy(PV + (oy / 2 + yy) * (s / 2) + ox / 2, t + yy * (q / 2), (size_t)(q / 2));
                    }
                const uint64_t d = sp_sse(So, sy, PY, s, s) + (((sp_sse(Su, sc, PU_, s / 2, s / 2) + sp_sse(Sv, sc, PV, s / 2, s / 2)) * (uint64_t)g_sp[4]) >> 2);
                const uint64_t j = (d << 10) + (uint64_t)(lam2 * (g_sp[1] + g_sp[2] * mypos));
                if (j < best) { best = j; bestk = k; bm = m; memcpy(BY, PY, (size_t)(s * s)); memcpy(BU, PU_, (size_t)(s * s / 4)); memcpy(BV, PV, (size_t)(s * s / 4)); }
            }
            if (bestk < 0) continue;
            for (int yy = 0; yy < n8; ++yy)
                for (int xx = 0; xx < n8; ++xx) {
                    kso_cu8 *o = &cu_out[(long)(by + yy) * w8 + bx + xx];
                    o->mvx = bm.mvx; o->mvy = bm.mvy; o->mv1x = bm.mv1x; o->mv1y = bm.mv1y; o->inter_dir = bm.inter_dir; o->log2_cu = (uint8_t)CU_LOG2(c0); o->cbf = 0; o->pred_mode = 0;
                }
            for (int yy = 0; yy < s; ++yy) { memset(lvl_y + (long)(y + yy) * W + x, 0, sizeof(int16_t) * (size_t)s); memcpy(Ro + (long)yy * sy, BY + yy * s, (size_t)s); }
            for (int yy = 0; yy < s / 2; ++yy) {
                memset(lvl_u + (long)(y / 2 + yy) * (W / 2) + x / 2, 0, sizeof(int16_t) * (size_t)(s / 2)); memset(lvl_v + (long)(y / 2 + yy) * (W / 2) + x / 2, 0, sizeof(int16_t) * (size_t)(s / 2));
                memcpy(Ru + (long)yy * sc, BU + yy * (s / 2), (size_t)(s / 2)); memcpy(Rv + (long)yy * sc, BV + yy * (s / 2), (size_t)(s / 2));
            }
        }
}

/* Multi-reference P pictures (-ref / -ref0: motionSearchOneRef enc@0x483f40 is called once per reference picture): per PU the reference
 * with the smallest cost + lambda * ref_idx bits wins (truncated unary: idx < nref - 1 ? idx + 1 : nref - 1 bits; ties to the
 * nearest picture).  pub: the winner's vector, inter_dir = 1 | idx << 4. */
void kso_ref_decide(const kso_frame_cfg *cfg, int nref, const kso_pu *const *pu, kso_pu_b *pub)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    long n = (long)g.ctu_cols * g.ctu_rows * 85;
    for (long i = 0; i < n; ++i) {
        kso_pu_b *o = &pub[i];
        o->mvx = pu[0][i].mvx; o->mvy = pu[0][i].mvy; o->mv1x = 0; o->mv1y = 0; o->cost = pu[0][i].cost; o->inter_dir = 1;
        if (pu[0][i].cost == COST_INVALID) continue;
        uint32_t best = COST_INVALID;
        for (int r = 0; r < nref; ++r) {
            int bits = nref == 1 ? 0 : (r < nref - 1 ? r + 1 : nref - 1);
            uint32_t c = pu[r][i].cost + (uint32_t)((cfg->lambda_q4 * bits) >> 4);
            if (c < best) { best = c; o->mvx = pu[r][i].mvx; o->mvy = pu[r][i].mvy; o->cost = c; o->inter_dir = 1u | ((uint32_t)r << 4); }
        }
    }
}

/* ------------------------------------------------------------------ Stage E: deblocking
 * bS as CalcBsInterP enc@0x402960 (single reference picture); all vertical edges of the picture, then all
 * horizontal ones (ctuDeblockFilterVer enc@0x403de0 / CtuDeblockFilterHorT enc@0x477200 do the same per CTU). */
static int edge_bs(const kso_cu8 *p, const kso_cu8 *q, int pos8 /*edge position in 8-sample units along its normal*/)
{
    int cu8n = 1 << (CU_LOG2(q) - 3), tu8n = imin(CU_PART(q) ? cu8n >> 1 : cu8n, 4);
    int tu_edge = (pos8 % tu8n) == 0, cu_edge = (pos8 % cu8n) == 0;
    if (!tu_edge && !cu_edge) return 0;
    if (p->pred_mode != 0 || q->pred_mode != 0) return 2;
    if (tu_edge && ((p->cbf | q->cbf) & 1)) return 1;
    if (cu_edge || CU_PART(q)) {                        /* a prediction-block edge: the CU's border, or inside a CU in two partitions (its TU edges; where the vectors are equal nothing fires) */
        /* CalcBsInterP enc@0x402960 / CalcBsInterB enc@0x4029d0 with one picture per list: different reference sets, or any
         * used vector differing by a full sample */
        if (p->inter_dir != q->inter_dir) return 1;
        if ((p->inter_dir & 1) && (iabs_(p->mvx - q->mvx) >= 4 || iabs_(p->mvy - q->mvy) >= 4)) return 1;
        if ((p->inter_dir & 2) && (iabs_(p->mv1x - q->mv1x) >= 4 || iabs_(p->mv1y - q->mv1y) >= 4)) return 1;
    }
    return 0;
}

/* test hook: the boundary strength this pipeline gives an edge (tests/test_oracle_golden.py holds it against ks265o_calc_bs, the restatement pinned on the reference) */
int kso_test_edge_bs(const kso_cu8 *p, const kso_cu8 *q, int pos8) { return edge_bs(p, q, pos8); }

void kso_deblock(const kso_frame_cfg *cfg, const kso_cu8 *cu8, kso_pic recon)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int W = cfg->width, H = cfg->height, w8 = W / 8, h8 = H / 8;
    uint8_t *eff = NULL;
    if (g_qpmap) { eff = malloc((size_t)w8 * h8); kso_effective_qp(cfg, cu8, eff); }
    uint8_t *Y = org_y(&g, recon.y);
    for (int dir = 0; dir < 2; ++dir) {
        /* luma, 8x8 grid; QpL = (QpP + QpQ + 1) >> 1 of the two CUs (8.7.2.5.3) */
        for (int by = 0; by < h8; ++by)
            for (int bx = 0; bx < w8; ++bx) {
                if (dir == 0 ? bx == 0 : by == 0) continue;
                const kso_cu8 *q = &cu8[(long)by * w8 + bx], *p = dir == 0 ? q - 1 : q - w8;
                int bs = edge_bs(p, q, dir == 0 ? bx : by);
                if (!bs) continue;
                const int qp = eff ? (eff[(long)by * w8 + bx] + eff[(long)by * w8 + bx - (dir == 0 ? 1 : w8)] + 1) >> 1 : cfg->qp;
                int beta = ks265o_beta_table[iclip(0, 51, qp + 2 * cfg->beta_offset_div2)];
                int tc = ks265o_tc_table[iclip(0, 53, qp + 2 * (bs - 1) + 2 * cfg->tc_offset_div2)];
                uint8_t *pix = Y + (long)by * 8 * g.stride_y + bx * 8;
                if (dir == 0) ks265o_edge_filter_luma_ver(pix, g.stride_y, beta, tc, 8, 1, 1);
                else ks265o_edge_filter_luma_hor(pix, g.stride_y, beta, tc, 8, 1, 1);
            }
        /* chroma, 8x8 chroma grid = 16 luma samples, only bS == 2; QpC from the table at the average of the two CUs' QpY (8.7.2.5.5) */
        for (int by = 0; by < h8; ++by)
            for (int bx = 0; bx < w8; ++bx) {
                if (dir == 0 ? (bx == 0 || (bx & 1)) : (by == 0 || (by & 1))) continue;
                const kso_cu8 *q = &cu8[(long)by * w8 + bx], *p = dir == 0 ? q - 1 : q - w8;
                if (edge_bs(p, q, dir == 0 ? bx : by) != 2) continue;
                const int qp = eff ? (eff[(long)by * w8 + bx] + eff[(long)by * w8 + bx - (dir == 0 ? 1 : w8)] + 1) >> 1 : cfg->qp;
                int tc = ks265o_tc_table[iclip(0, 53, chroma_qp(qp) + 2 + 2 * cfg->tc_offset_div2)];
                for (int comp = 0; comp < 2; ++comp) {
                    uint8_t *pix = org_c(&g, comp ? recon.v : recon.u) + (long)by * 4 * g.stride_c + bx * 4;
                    if (dir == 0) ks265o_pixel_filter_chroma_ver(pix, g.stride_c, tc, 4, 1, 1);
                    else ks265o_pixel_filter_chroma_hor(pix, g.stride_c, tc, 4, 1, 1);
                }
            }
    }
    free(eff);
}

/* ------------------------------------------------------------------ Stage F: SAO
 * statistics with the reference's s8 difference truncation (statSaoBoEo01_c enc@0x4ae9c0), all four
 * EO classes + BO over the whole CTU; decision = distortion estimate + lambda * rate estimate (CEncSao::modeDecisionCtu
 * enc@0x4af690 is closed RD code); apply out of place from the deblocked picture (SaoApplyOffset*_c enc@0x43e4e0..). */
typedef struct { int cnt[5][32]; int sum[5][32]; } sao_stats;   /* [0] BO 32 bands, [1..4] EO class 0..3 x 4 categories */

static const int kEoDx[4] = {1, 0, 1, -1}, kEoDy[4] = {0, 1, 1, 1};

static void sao_collect(const uint8_t *org, const uint8_t *rec, long stride, int x0, int y0, int w, int h, int picW, int picH, sao_stats *s)
{
    memset(s, 0, sizeof *s);
    for (int y = y0; y < y0 + h; ++y)
        for (int x = x0; x < x0 + w; ++x) {
            int c = rec[(long)y * stride + x];
            int d = (int8_t)(uint8_t)(org[(long)y * stride + x] - c);
            s->cnt[0][c >> 3]++; s->sum[0][c >> 3] += d;
            for (int k = 0; k < 4; ++k) {
                int ax = x - kEoDx[k], ay = y - kEoDy[k], bx = x + kEoDx[k], by = y + kEoDy[k];
                if (ax < 0 || bx < 0 || ax >= picW || bx >= picW || ay < 0 || by >= picH) continue;
                int e = 2 + isgn(c - rec[(long)ay * stride + ax]) + isgn(c - rec[(long)by * stride + bx]);
                if (e == 2) continue;
                int cat = e < 2 ? e : e - 1;
                s->cnt[1 + k][cat]++; s->sum[1 + k][cat] += d;
            }
        }
}

static int sao_offset(int sum, int cnt, int lo, int hi)
{
    if (!cnt) return 0;
    int o = sum >= 0 ? (sum + cnt / 2) / cnt : -((-sum + cnt / 2) / cnt);
    return iclip(lo, hi, o);
}

/* best (cost, params) of one type for one component; cost = 256 * deltaD + lambda_q4^2 * bits */
static int64_t sao_eval(const sao_stats *s, int type, int lam, kso_sao_param *out)
{
    int64_t lam2 = (int64_t)lam * lam;
    memset(out, 0, sizeof *out);
    out->type = (int8_t)type;
    if (type == 0) {
        int off[32]; int64_t dd[32];
        for (int b = 0; b < 32; ++b) {
            off[b] = sao_offset(s->sum[0][b], s->cnt[0][b], -7, 7);
            dd[b] = (int64_t)s->cnt[0][b] * off[b] * off[b] - 2LL * off[b] * s->sum[0][b];
        }
        int best = 0; int64_t bd = 0;
        for (int p = 0; p <= 28; ++p) {
            int64_t d = dd[p] + dd[p + 1] + dd[p + 2] + dd[p + 3];
            if (p == 0 || d < bd) { bd = d; best = p; }
        }
        int bits = 7;
        for (int k = 0; k < 4; ++k) { out->offset[k] = (int8_t)off[best + k]; bits += iabs_(off[best + k]) + 2; }
        out->band = (int8_t)best;
        return bd * 256 + lam2 * bits;
    }
    int64_t d = 0; int bits = 4;
    for (int c = 0; c < 4; ++c) {
        int o = sao_offset(s->sum[type][c], s->cnt[type][c], c < 2 ? 0 : -7, c < 2 ? 7 : 0);
        out->offset[c] = (int8_t)o;
        d += (int64_t)s->cnt[type][c] * o * o - 2LL * o * s->sum[type][c];
        bits += iabs_(o) + 1;
    }
    return d * 256 + lam2 * bits;
}

/* cfg->sao == 2 (round 6; VERDICT r5 missing 5): the decision of CEncSao::modeDecisionCtu enc@0x4af690 on its `-sao 4` path, read from the disassembly:
 *   modeDecisionBoEo01 enc@0x4af300 - per component group (luma; Cb + Cr together) the candidates in the order EO class 0, EO class 1, band offset, each priced by the pinned
 *   type estimations (ks265o_sao_eo_type_estimation / _bo_type_estimation = EoTypeDistEstimation enc@0x4adf60 / BoTypeDistEstimation enc@0x4adc70) plus the type's own rate -
 *   calcRDcostEoY enc@0x4ae290: + ((4 lambda + 128) >> 8); calcRDcostBoY enc@0x4ade40: + ((7 lambda + 128) >> 8); calcRDcostEoUV enc@0x4ae2e0: both components' costs +
 *   ((4 lambda_c + 128) >> 8); calcRDcostBoUV enc@0x4adeb0: + ((12 lambda_c + 128) >> 8), a band position per component - against "off" = one bin, (lambda + 128) >> 8;
 *   checkRDCostY / checkRDCostUV enc@0x4ad810 / 0x4ad860: strictly cheaper wins.  lambda = g_lambdaOptforSAO enc@0x4df240 [QpY] / [QpC] (Q8, the table below).
 * 32-bit arithmetic as in the binary.  NOT taken over: the statistics window (the reference collects 60 / 28 columns of a CTU because its last four are not deblocked yet when its
 * CTU pipeline gets there; here the whole picture is deblocked first and every CTU counts all its samples) and the merge candidates (checkMerge enc@0x4ae7f0 prices
 * the FINAL parameters of the left / upper CTU - a chain through the whole picture in coding order; the slices are written with both merge flags 0). */
static const int32_t kLambdaSaoQ8[52] = {9, 12, 15, 19, 24, 31, 39, 50, 63, 79, 100, 127, 161, 203, 257, 325, 411, 519, 656, 829, 1048, 1324, 1674, 2115, 2673, 3377, 4268, 5393, 6815, 8612, 10883,
                                         13752, 17378, 21960, 27750, 35066, 44311, 55994, 70757, 89411, 112984, 142772, 180413, 227978, 288084, 364036, 460012, 581291, 734546, 928205, 1172921, 1482155};
static int chroma_qp(int qp);
static void sao_decide_ref(const sao_stats *st /*[3]*/, int qp, kso_sao_param *out /*[3]*/)
{
    /* the pinned restatement (ks265o_sao_mode_decision: replayed on 630 recorded calls, tests/golden/sao_decision.npz) on this CTU's statistics in the object's layout, no
     * neighbour records: the candidates, their order, rates and lambdas are the reference's, the merge step is not taken */
    int32_t stats[312];
    int8_t rec[32];
    memset(stats, 0, sizeof stats);
    for (int c = 0; c < 3; ++c) {
        for (int b = 0; b < 32; ++b) { stats[32 * c + b] = st[c].cnt[0][b]; stats[156 + 32 * c + b] = st[c].sum[0][b]; }
        for (int cls = 0; cls < 2; ++cls)
            for (int k = 0; k < 4; ++k) { stats[96 + 20 * c + 5 * cls + k] = st[c].cnt[1 + cls][k]; stats[156 + 96 + 20 * c + 5 * cls + k] = st[c].sum[1 + cls][k]; }
    }
    ks265o_sao_mode_decision(stats, kLambdaSaoQ8[qp], kLambdaSaoQ8[chroma_qp(qp)], 0, 0, NULL, NULL, 0x13, 0x13, rec, NULL);
    for (int c = 0; c < 3; ++c) {
        memset(&out[c], 0, sizeof out[c]);
        const int t = rec[c ? 1 : 0];
        out[c].type = (int8_t)(t == -1 ? -1 : t == 4 ? 0 : 1 + t);           /* this pipeline's codes: -1 off, 0 band offset, 1 + edge class */
        if (t == -1) continue;
        out[c].band = (int8_t)(t == 4 ? rec[c == 0 ? 2 : 2 + c] : 0);
        for (int k = 0; k < 4; ++k) out[c].offset[k] = rec[(c == 0 ? 5 : c == 1 ? 0xa : 0xf) + k];
    }
}

static void sao_apply_ctu(const uint8_t *rec, uint8_t *dst, long stride, int x0, int y0, int w, int h, int picW, int picH, const kso_sao_param *p)
{
    for (int y = y0; y < y0 + h; ++y)
        for (int x = x0; x < x0 + w; ++x) {
            int c = rec[(long)y * stride + x], o = 0;
            if (p->type == 0) {
                int k = (c >> 3) - p->band;
                if (k >= 0 && k < 4) o = p->offset[k];
            } else if (p->type > 0) {
                int k = p->type - 1;
                int ax = x - kEoDx[k], ay = y - kEoDy[k], bx = x + kEoDx[k], by = y + kEoDy[k];
                if (!(ax < 0 || bx < 0 || ax >= picW || bx >= picW || ay < 0 || by >= picH)) {
                    int e = 2 + isgn(c - rec[(long)ay * stride + ax]) + isgn(c - rec[(long)by * stride + bx]);
                    if (e != 2) o = p->offset[e < 2 ? e : e - 1];
                }
            }
            int v = c + o;
            dst[(long)y * stride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

void kso_sao(const kso_frame_cfg *cfg, kso_pic src, kso_pic deb, kso_sao_param *sao, kso_pic dst)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int W = cfg->width, H = cfg->height, lam = cfg->lambda_q4;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)   /* reads src / deb, writes the CTU's own part of dst and sao */
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            kso_sao_param *sp = sao + (long)(cy * g.ctu_cols + cx) * 3;
            int x0 = cx * 64, y0 = cy * 64, w = imin(64, W - x0), h = imin(64, H - y0);
            sao_stats st[3];
            sao_collect(org_y(&g, src.y), org_y(&g, deb.y), g.stride_y, x0, y0, w, h, W, H, &st[0]);
            sao_collect(org_c(&g, src.u), org_c(&g, deb.u), g.stride_c, x0 / 2, y0 / 2, w / 2, h / 2, W / 2, H / 2, &st[1]);
            sao_collect(org_c(&g, src.v), org_c(&g, deb.v), g.stride_c, x0 / 2, y0 / 2, w / 2, h / 2, W / 2, H / 2, &st[2]);
            /* luma decides alone; Cb and Cr share the type (sao_type_idx_chroma / sao_eo_class_chroma) */
            kso_sao_param best[3], cand[3];
            int64_t bj = 0, bjc = 0;
            for (int c = 0; c < 3; ++c) { memset(&best[c], 0, sizeof best[c]); best[c].type = -1; }
            if (cfg->sao == 2) sao_decide_ref(st, ctu_qp(cfg, x0, y0), best);
            else if (cfg->sao) {
                for (int t = 0; t < 5; ++t) {
                    int64_t j = sao_eval(&st[0], t, lam, &cand[0]);
                    if (j < bj) { bj = j; best[0] = cand[0]; }
                    int64_t jc = sao_eval(&st[1], t, lam, &cand[1]) + sao_eval(&st[2], t, lam, &cand[2]);
                    if (jc < bjc) { bjc = jc; best[1] = cand[1]; best[2] = cand[2]; }
                }
            }
            for (int c = 0; c < 3; ++c) sp[c] = best[c];
            sao_apply_ctu(org_y(&g, deb.y), org_y(&g, dst.y), g.stride_y, x0, y0, w, h, W, H, &sp[0]);
            sao_apply_ctu(org_c(&g, deb.u), org_c(&g, dst.u), g.stride_c, x0 / 2, y0 / 2, w / 2, h / 2, W / 2, H / 2, &sp[1]);
            sao_apply_ctu(org_c(&g, deb.v), org_c(&g, dst.v), g.stride_c, x0 / 2, y0 / 2, w / 2, h / 2, W / 2, H / 2, &sp[2]);
        }
    kso_pad_picture(cfg, dst);
}

/* ================================================================== intra pictures
 * Kernels: ks265o_intra_pred / ks265o_intra_filter_ref (pinned against IntraPred*_c / IntraPredFilterRef_c, tests/golden/intra.npz).
 * Sequencing (the build's own, like the CU decision of P pictures: decideBestLumaModeBySadFast enc@0x499170 / decideLumaMode
 * enc@0x49acc0 are closed RDO code):
 *   kso_intra_decide       every 8x8 / 16x16 / 32x32 block: all 35 luma modes predicted from SOURCE neighbours ("pre-selection is
 *                          embarrassingly parallel on source pixels"), cost = SATD + lambda * mode bits, then the CU quadtree bottom-up;
 *   kso_intra_reconstruct  CTUs in raster order, CUs in z-order: neighbours from the RECONSTRUCTED picture with the normative
 *                          availability (H.265 6.4.1) and substitution (8.4.4.2.2) rules, reference smoothing (8.4.4.2.3), prediction,
 *                          residual -> transform -> quant -> dequant -> inverse -> recon per TU (= CU, at most 32x32);
 *                          chroma uses the luma mode (DM), unsmoothed references, no edge filter.
 * cu8 of an intra CU: pred_mode = 2, mvx = luma mode. */
static int morton4(int x, int y)       /* z-scan address of the 4x4 unit (x, y), x, y < 16 */
{
    int z = 0;
    for (int b = 0; b < 4; ++b) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
    return z;
}
/* is luma sample (nx, ny) available to the block whose first sample is (x, y)?  inside the picture and earlier in decoding order */
static int intra_avail(int W, int H, int x, int y, int nx, int ny)
{
    if (nx < 0 || ny < 0 || nx >= W || ny >= H) return 0;
    int cols = (W + 63) / 64, ca = (y >> 6) * cols + (x >> 6), na = (ny >> 6) * cols + (nx >> 6);
    if (na != ca) return na < ca;
    return morton4((nx & 63) >> 2, (ny & 63) >> 2) < morton4((x & 63) >> 2, (y & 63) >> 2);
}
/* reference samples of the n x n block at (x, y) (component samples; sh = 0 luma, 1 chroma) from plane `p` (sample (0,0), stride st);
 * r = corner pointer of a 4n + 1 array */
static void intra_gather(const uint8_t *p, long st, int W, int H, int sh, int x, int y, int n, uint8_t *r)
{
    uint8_t av[4 * 32 + 1];
    int any = 0;
    for (int k = -2 * n; k <= 2 * n; ++k) {
        int sx = k > 0 ? x + k - 1 : x - 1, sy = k < 0 ? y - k - 1 : y - 1;
        int a = intra_avail(W, H, x << sh, y << sh, sx < 0 ? -1 : sx << sh, sy < 0 ? -1 : sy << sh);
        av[k + 2 * n] = (uint8_t)a;
        any |= a;
        r[k] = a ? p[(long)sy * st + sx] : 0;
    }
    if (!any) { for (int k = -2 * n; k <= 2 * n; ++k) r[k] = 128; return; }
    if (!av[0]) {
        int k = -2 * n + 1;
        while (!av[k + 2 * n]) ++k;
        r[-2 * n] = r[k];
    }
    for (int k = -2 * n + 1; k <= 2 * n; ++k)
        if (!av[k + 2 * n]) r[k] = r[k - 1];
}
static int intra_filter_flag(int mode, int n)
{
    if (mode == 1 || n <= 4) return 0;
    int d1 = mode > 26 ? mode - 26 : 26 - mode, d2 = mode > 10 ? mode - 10 : 10 - mode, d = d1 < d2 ? d1 : d2;
    return d > (n == 8 ? 7 : (n == 16 ? 1 : 0));
}
static int intra_mode_bits(int mode) { return (mode == 0 || mode == 1 || mode == 26) ? 3 : 6; }   /* default MPM set vs. escape code */

/* best luma mode of one block from source neighbours; returns the cost */
/* coarse: the candidates of P / B pictures try planar, DC and every second angular mode (2, 4, .. 34: 19 of the 35).  Measured with tools/rd_eval.py on the
 * 832x480 clips: no loss against all 35 (hierarchical B: 126481 vs 126673 bytes at the same PSNR); every fourth mode (11 of 35) cost +0.5 % there and +1.4 % of
 * the bytes of the 2160p IPPP bench clip - more than the third of the candidates kernel's time it saved was worth */
#define INTRA_INTER_MODE_STEP 2
static uint32_t intra_best_mode(const kso_frame_cfg *cfg, const uint8_t *S, long st, int x, int y, int n, int *best_mode, int coarse)
{
    uint8_t raw[4 * 32 + 1], fil[4 * 32 + 1], pred[32 * 32];
    int log2 = n == 8 ? 3 : (n == 16 ? 4 : 5);
    intra_gather(S, st, cfg->width, cfg->height, 0, x, y, n, raw + 2 * n);
    ks265o_intra_filter_ref(raw + 2 * n, fil + 2 * n, n, 1);
    uint32_t best = COST_INVALID;
    for (int mode = 0; mode < 35; ++mode) {
        if (coarse && mode >= 2 && ((mode - 2) % INTRA_INTER_MODE_STEP)) continue;
        ks265o_intra_pred(pred, n, (intra_filter_flag(mode, n) ? fil : raw) + 2 * n, mode, log2, 1);
        uint32_t c = ks265o_had(S + (long)y * st + x, pred, st, n, n, n) + (uint32_t)((cfg->lambda_q4 * intra_mode_bits(mode)) >> 4);
        if (c < best) { best = c; *best_mode = mode; }
    }
    return best;
}
typedef struct { uint32_t cost[85]; uint8_t mode[85], split[85]; } intra_ctu;
static uint32_t intra_node(const kso_frame_cfg *cfg, intra_ctu *t, int cx, int cy, int l, int px, int py)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    if (x0 >= cfg->width || y0 >= cfg->height) return 0;
    int idx = pu_index(l, px, py);
    uint32_t own = l == 0 ? COST_INVALID : t->cost[idx];          /* no 64x64 intra CU: its TUs would be four 32x32 anyway */
    if (l == 3) { t->split[idx] = 0; return own; }
    uint64_t sum = (uint64_t)((cfg->lambda_q4 * 12) >> 4);
    for (int k = 0; k < 4; ++k) sum += intra_node(cfg, t, cx, cy, l + 1, px * 2 + (k & 1), py * 2 + (k >> 1));
    if (own != COST_INVALID && (uint64_t)own <= sum) { t->split[idx] = 0; return own; }
    t->split[idx] = 1;
    return sum > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)sum;
}
static void intra_emit(const kso_frame_cfg *cfg, const intra_ctu *t, int cx, int cy, int l, int px, int py, kso_cu8 *cu8)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, w8 = cfg->width / 8;
    if (x0 >= cfg->width || y0 >= cfg->height) return;
    int idx = pu_index(l, px, py);
    if (l < 3 && t->split[idx]) {
        for (int k = 0; k < 4; ++k) intra_emit(cfg, t, cx, cy, l + 1, px * 2 + (k & 1), py * 2 + (k >> 1), cu8);
        return;
    }
    for (int by = 0; by < s / 8; ++by)
        for (int bx = 0; bx < s / 8; ++bx) {
            kso_cu8 *c = &cu8[(long)(y0 / 8 + by) * w8 + x0 / 8 + bx];
            c->mvx = t->mode[idx]; c->mvy = 0; c->mv1x = 0; c->mv1y = 0; c->log2_cu = (uint8_t)(6 - l); c->cbf = 0; c->pred_mode = 2; c->inter_dir = 0;
        }
}
/* cost_out (optional): the pre-selection cost of every block, PU indexing (85 per CTU; level 0 and blocks not inside the picture: COST_INVALID) */
static void intra_decide_impl(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, uint32_t *cost_out, uint8_t *mode_out);
void kso_intra_decide(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8) { intra_decide_impl(cfg, src, cu8, NULL, NULL); }
void kso_intra_decide_ex(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, uint32_t *cost_out) { intra_decide_impl(cfg, src, cu8, cost_out, NULL); }
/* the candidates of P / B pictures (cfg->intra_inter): cost and best mode of every block, no CU tree.  Gate: a CTU is evaluated only if one of its 8x8 PUs has an
 * inter cost of at least lambda x INTRA_BIAS_BITS >> 4 - what an intra CU costs before its first residual bit; where every 8x8 block is predicted better than that no
 * block goes intra (costs invalid).  pu_records: the picture's ks265_pu / ks265_pu_b records (16 bytes each, cost at byte 8).  Measured on hierarchical-B pictures
 * at 832x480 and 1920x1080: 7 - 26 % of the CTUs pass, none of the CTUs that end up with an intra CU is gated out. */
void kso_intra_candidates(const kso_frame_cfg *cfg, kso_pic src, const void *pu_records, uint32_t *cost_out, uint8_t *mode_out)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    intra_decide_impl(cfg, src, NULL, cost_out, mode_out);
    const uint32_t thr = (uint32_t)((cfg->lambda_q4 * INTRA_BIAS_BITS) >> 4);
    for (long c = 0; c < (long)g.ctu_cols * g.ctu_rows; ++c) {
        int go = 0;
        for (int i = 21; i < 85 && !go; ++i) {
            const uint32_t cost = ((const uint32_t *)pu_records)[(c * 85 + i) * 4 + 2];
            go = cost != COST_INVALID && cost >= thr;
        }
        if (!go) for (int i = 0; i < 85; ++i) { cost_out[c * 85 + i] = COST_INVALID; mode_out[c * 85 + i] = 0; }
    }
}
static void intra_decide_impl(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, uint32_t *cost_out, uint8_t *mode_out)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    const uint8_t *S = org_y(&g, src.y);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx) {
            intra_ctu t;
            memset(&t, 0, sizeof t);
            for (int l = 1; l < 4; ++l)
                for (int py = 0; py < (1 << l); ++py)
                    for (int px = 0; px < (1 << l); ++px) {
                        int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s, idx = pu_index(l, px, py), m = 0;
                        t.cost[idx] = COST_INVALID;
                        if (x0 + s > cfg->width || y0 + s > cfg->height) continue;
                        t.cost[idx] = intra_best_mode(cfg, S, g.stride_y, x0, y0, s, &m, !cu8 && mode_out);
                        t.mode[idx] = (uint8_t)m;
                    }
            t.cost[0] = COST_INVALID;
            if (cost_out) memcpy(cost_out + (long)(cy * g.ctu_cols + cx) * 85, t.cost, sizeof t.cost);
            if (mode_out) memcpy(mode_out + (long)(cy * g.ctu_cols + cx) * 85, t.mode, sizeof t.mode);
            if (!cu8) continue;
            intra_node(cfg, &t, cx, cy, 0, 0, 0);
            intra_emit(cfg, &t, cx, cy, 0, 0, 0, cu8);
        }
}

/* ------------------------------------------------------------------ lookahead frame cost (calcFrameCost enc@0x4a7410 / scenecut
 * enc@0x47e9d0 lineage on the half-resolution pictures of downsample_c; the reference's cost logic is closed, this is the x264-lineage
 * composition its kernels belong to).  Per 8x8 block of the low-resolution picture: intra cost = the pre-selection cost of the block,
 * inter cost = the integer search cost of the 8x8 PU against the previous low-resolution picture; the frame sums feed the slice-type /
 * scene-cut decision of the host (a cut when the inter sum is not clearly below the intra sum). */
void kso_lookahead_reduce(const kso_frame_cfg *cfg, const uint32_t *intra_cost, const kso_pu *pu, uint64_t out[4])
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    uint64_t si = 0, sp = 0, sm = 0, nb = 0, ni = 0;
    for (long c = 0; c < (long)g.ctu_cols * g.ctu_rows; ++c)
        for (int i = 21; i < 85; ++i) {
            uint32_t a = intra_cost[c * 85 + i], b = pu[c * 85 + i].cost;
            if (a == COST_INVALID || b == COST_INVALID) continue;
            si += a; sp += b; sm += a < b ? a : b; ++nb; ni += a < b;
        }
    out[0] = si; out[1] = sp; out[2] = sm; out[3] = nb | (ni << 32);
}

/* islice: the quantiser's rounding offset follows the SLICE type (H265_GetBaseQuantParam enc@0x4a9c90: 171 in I slices, 85 otherwise), also for intra CUs */
static void intra_code_cu(const kso_frame_cfg *cfg, const kso_frame_geom *g, kso_pic src, kso_cu8 *cu8, int bx, int by, int16_t *lvl_y, int16_t *lvl_u,
                          int16_t *lvl_v, kso_pic recon, int islice)
{
    int W = cfg->width, H = cfg->height, w8 = W / 8, qp = ctu_qp(cfg, bx * 8, by * 8), qpc = chroma_qp(qp);
    long sy = g->stride_y, sc = g->stride_c;
    kso_cu8 *c = &cu8[(long)by * w8 + bx];
    int n = 1 << CU_LOG2(c), x0 = bx * 8, y0 = by * 8, mode = c->mvx, log2 = CU_LOG2(c), cbf = 0;
    uint8_t raw[4 * 32 + 1], fil[4 * 32 + 1], pred[32 * 32];
    uint8_t *Ry = org_y(g, recon.y);
    intra_gather(Ry, sy, W, H, 0, x0, y0, n, raw + 2 * n);
    const uint8_t *r = raw + 2 * n;
    if (intra_filter_flag(mode, n)) { ks265o_intra_filter_ref(raw + 2 * n, fil + 2 * n, n, 1); r = fil + 2 * n; }
    ks265o_intra_pred(pred, n, r, mode, log2, 1);
    cbf |= code_tu(org_y(g, src.y) + (long)y0 * sy + x0, (int)sy, pred, n, qp, islice ? 1 : 2, lvl_y + (long)y0 * W + x0, W, Ry + (long)y0 * sy + x0, (int)sy, cfg->sdh,
                   tu_scan_idx(1, mode, n, 0), 0, islice ? ((g_rq_mode & 4) ? 1 : 0) : cfg->rdo, cfg->lambda_q4);
    int nc = n / 2, xc = x0 / 2, yc = y0 / 2;
    for (int comp = 0; comp < 2; ++comp) {
        uint8_t *Rc = org_c(g, comp ? recon.v : recon.u);
        intra_gather(Rc, sc, W, H, 1, xc, yc, nc, raw + 2 * nc);
        ks265o_intra_pred(pred, nc, raw + 2 * nc, mode, log2 - 1, 0);
        int16_t *lv = (comp ? lvl_v : lvl_u) + (long)yc * (W / 2) + xc;
        const uint8_t *oc = org_c(g, comp ? src.v : src.u) + (long)yc * sc + xc;
        if (code_tu(oc, (int)sc, pred, nc, qpc, islice ? 1 : 2, lv, W / 2, Rc + (long)yc * sc + xc, (int)sc, cfg->sdh, tu_scan_idx(1, mode, nc, 1), 0, (g_rq_mode & 2) ? -1 : 0, cfg->lambda_q4)) cbf |= 2 << comp;
    }
    for (int yy = 0; yy < n / 8; ++yy)
        for (int xx = 0; xx < n / 8; ++xx) cu8[(long)(by + yy) * w8 + bx + xx].cbf = (uint8_t)cbf;
}
void kso_intra_reconstruct(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int w8 = cfg->width / 8, h8 = cfg->height / 8;
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx)
            for (int z = 0; z < 64; ++z) {                          /* 8x8 blocks of the CTU in z-order */
                int lx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ly = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
                int bx = cx * 8 + lx, by = cy * 8 + ly;
                if (bx >= w8 || by >= h8) continue;
                int n8 = 1 << ((cu8[(long)by * w8 + bx].log2_cu & 15) - 3);
                if ((lx % n8) || (ly % n8)) continue;                /* visit each CU once, at its first 8x8 block */
                intra_code_cu(cfg, &g, src, cu8, bx, by, lvl_y, lvl_u, lvl_v, recon, 1);
            }
}
/* cfg->intra_inter: the intra CUs (pred_mode 2) of a P / B picture, after kso_reconstruct has written every inter CU: CTUs in raster order, CUs in z-order, neighbours
 * from the reconstructed picture (inter and intra alike: constrained_intra_pred_flag = 0) */
void kso_intra_inter_reconstruct(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon)
{
    kso_frame_geom g; kso_frame_geometry(cfg, &g);
    int w8 = cfg->width / 8, h8 = cfg->height / 8;
    for (int cy = 0; cy < g.ctu_rows; ++cy)
        for (int cx = 0; cx < g.ctu_cols; ++cx)
            for (int z = 0; z < 64; ++z) {
                int lx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), ly = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
                int bx = cx * 8 + lx, by = cy * 8 + ly;
                if (bx >= w8 || by >= h8) continue;
                if (cu8[(long)by * w8 + bx].pred_mode != 2) continue;
                int n8 = 1 << ((cu8[(long)by * w8 + bx].log2_cu & 15) - 3);
                if ((lx % n8) || (ly % n8)) continue;
                intra_code_cu(cfg, &g, src, cu8, bx, by, lvl_y, lvl_u, lvl_v, recon, 0);
            }
}
