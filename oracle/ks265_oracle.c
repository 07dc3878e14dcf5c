/* ks265_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ks265_oracle.h).
 *
 * Plain-C restatement of the KSC265 encoder's hot-path pixel kernels.  The reference is
 * binary-only; each function cites the symbol it restates as `enc@0xADDR name`
 * (/root/reference/ubuntu_x64/appencoder) and the SURVEY.md paragraph that records the
 * verified contract.  Pinned bit-exactly against the reference binary's own outputs:
 * tests/golden/ (npz files) (made by oracle/ref_probe/gen_golden.py), tests/test_oracle_golden.py.
 */
#include "ks265_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline uint8_t clip8(int v) { return (uint8_t)clip3(0, 255, v); }
static inline int16_t clip16(int v) { return (int16_t)clip3(-32768, 32767, v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int sgn(int v) { return (v > 0) - (v < 0); }

/* ------------------------------------------------------------------ SAD family */

/* enc@0x47ae30 sad_c — SURVEY.md B.1 */
uint32_t ks265o_sad(const uint8_t *a, const uint8_t *b, long sa, long sb, long h, long w)
{
    uint32_t s = 0;
    for (long y = 0; y < h; ++y, a += sa, b += sb)
        for (long x = 0; x < w; ++x) s += (uint32_t)iabs((int)a[x] - (int)b[x]);
    return s;
}

/* enc@0x47ae90 sad4_c — SURVEY.md B.1: {up, down, left, right} each << 4 */
void ks265o_sad4(const uint8_t *fenc, const uint8_t *ref, long sFenc, long sRef, long h, uint32_t out[4], long w)
{
    out[0] = ks265o_sad(fenc, ref - sRef, sFenc, sRef, h, w) << 4;
    out[1] = ks265o_sad(fenc, ref + sRef, sFenc, sRef, h, w) << 4;
    out[2] = ks265o_sad(fenc, ref - 1, sFenc, sRef, h, w) << 4;
    out[3] = ks265o_sad(fenc, ref + 1, sFenc, sRef, h, w) << 4;
}

/* enc@0x47b060 sad3_c — SURVEY.md B.1: three arbitrary reference pointers, unshifted */
void ks265o_sad3(const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2, long sFenc, long sRef,
                 long h, uint32_t out[3], long w)
{
    out[0] = ks265o_sad(fenc, r0, sFenc, sRef, h, w);
    out[1] = ks265o_sad(fenc, r1, sFenc, sRef, h, w);
    out[2] = ks265o_sad(fenc, r2, sFenc, sRef, h, w);
}

/* enc@0x4cee30 sad4blk_8x8_c — the four 8x8 quadrants of a 16x16, raster order */
void ks265o_sad4blk_8x8(const uint8_t *a, const uint8_t *b, long sa, long sb, uint32_t out[4])
{
    out[0] = ks265o_sad(a, b, sa, sb, 8, 8);
    out[1] = ks265o_sad(a + 8, b + 8, sa, sb, 8, 8);
    out[2] = ks265o_sad(a + 8 * sa, b + 8 * sb, sa, sb, 8, 8);
    out[3] = ks265o_sad(a + 8 * sa + 8, b + 8 * sb + 8, sa, sb, 8, 8);
}

/* enc@0x47b230.. sse_c<N> */
uint32_t ks265o_sse(const uint8_t *a, const uint8_t *b, int sa, int sb, int n)
{
    uint32_t s = 0;
    for (int y = 0; y < n; ++y, a += sa, b += sb)
        for (int x = 0; x < n; ++x) { int d = (int)a[x] - (int)b[x]; s += (uint32_t)(d * d); }
    return s;
}

/* Hadamard helpers: unnormalised +-1 butterflies; the |.| sum is order independent */
static uint32_t had_tile(const uint8_t *a, const uint8_t *b, long sa, long sb, int n)
{
    int m[64], t[64];
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) m[y * n + x] = (int)a[y * sa + x] - (int)b[y * sb + x];
    for (int y = 0; y < n; ++y) {                       /* rows */
        int *r = m + y * n;
        for (int len = 1; len < n; len <<= 1)
            for (int i = 0; i < n; i += 2 * len)
                for (int j = i; j < i + len; ++j) { int u = r[j], v = r[j + len]; r[j] = u + v; r[j + len] = u - v; }
    }
    for (int x = 0; x < n; ++x) {                       /* columns */
        for (int y = 0; y < n; ++y) t[y] = m[y * n + x];
        for (int len = 1; len < n; len <<= 1)
            for (int i = 0; i < n; i += 2 * len)
                for (int j = i; j < i + len; ++j) { int u = t[j], v = t[j + len]; t[j] = u + v; t[j + len] = u - v; }
        for (int y = 0; y < n; ++y) m[y * n + x] = t[y];
    }
    uint32_t s = 0;
    for (int i = 0; i < n * n; ++i) s += (uint32_t)iabs(m[i]);
    return s;
}

/* enc@0x47b680 had_c (+ enc@0x47b3b0 xCalcHADs8x8) — SURVEY.md B.2 */
uint32_t ks265o_had(const uint8_t *a, const uint8_t *b, long sa, long sb, long h, long w)
{
    uint32_t s = 0;
    if (((h | w) & 7) == 0) {
        for (long y = 0; y < h; y += 8)
            for (long x = 0; x < w; x += 8) s += (had_tile(a + y * sa + x, b + y * sb + x, sa, sb, 8) + 2) >> 2;
    } else if (((h | w) & 3) == 0) {
        for (long y = 0; y < h; y += 4)
            for (long x = 0; x < w; x += 4) s += (had_tile(a + y * sa + x, b + y * sb + x, sa, sb, 4) + 1) >> 1;
    } else {
        for (long y = 0; y < h; y += 2)
            for (long x = 0; x < w; x += 2) s += had_tile(a + y * sa + x, b + y * sb + x, sa, sb, 2);
    }
    return s;
}

/* enc@0x4345f0.. H265_CalResidual<N>(res, org, pred, strideOrg, stridePred): residual rows are packed (stride N) */
void ks265o_calc_residual(int16_t *res, const uint8_t *org, const uint8_t *pred, int strideOrg, int stridePred,
                          int strideRes, int n)
{
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) res[y * strideRes + x] = (int16_t)((int)org[y * strideOrg + x] - (int)pred[y * stridePred + x]);
}

/* ------------------------------------------------------------------ transforms */

/* |cos(j*pi/64)| * 64*sqrt(2) rounded as in HEVC (g_uiTr32 enc@0x4e06a0 holds the expanded 32x32 matrix) */
static const int8_t kMag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
/* DST4x4_COEFF enc@0x4e0aa0 */
static const int8_t kDst4[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

static int dct_coef(int n, int k, int x)
{
    int m = (k * (32 / n) * (2 * x + 1)) & 127;          /* angle index, period 128 */
    if (m > 64) m = 128 - m;
    return m > 32 ? -kMag[64 - m] : kMag[m];
}

static void load_matrix(int idx, int M[32][32], int *n)
{
    static const int sizes[5] = {4, 4, 8, 16, 32};
    *n = sizes[idx];
    for (int k = 0; k < *n; ++k)
        for (int x = 0; x < *n; ++x) M[k][x] = idx == 0 ? kDst4[k][x] : dct_coef(*n, k, x);
}

/* enc@0x4c2210.. H265_2dDct{4,8,16,32}_c / enc@0x4c2250 H265_2dDst4x4_c — SURVEY.md B.3.
 * pass 1 (rows of src, output transposed into tmp) shift 2*log2N-2; pass 2 shift 7; round half up. */
void ks265o_fwd_transform(int idx, const int16_t *src, int16_t *dst, int srcStride, int dstStride, int16_t *tmp)
{
    int M[32][32], n;
    load_matrix(idx, M, &n);
    int log2n = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5;
    int s1 = 2 * log2n - 2, r1 = 1 << (s1 - 1);
    for (int j = 0; j < n; ++j)          /* source row j */
        for (int k = 0; k < n; ++k) {
            int acc = 0;
            for (int x = 0; x < n; ++x) acc += M[k][x] * src[j * srcStride + x];
            tmp[k * n + j] = (int16_t)((acc + r1) >> s1);
        }
    for (int j = 0; j < n; ++j)          /* tmp row j */
        for (int k = 0; k < n; ++k) {
            int acc = 0;
            for (int x = 0; x < n; ++x) acc += M[k][x] * tmp[j * n + x];
            dst[k * dstStride + j] = (int16_t)((acc + 64) >> 7);
        }
}

/* enc@0x448c40 H265_2dIDst4x4_c, enc@0x448f60.. H265_2dIDct{4,8,16,32}_c — SURVEY.md B.4.
 * T = clip16((M^T C + 64) >> 7); R = (T M + 2048) >> 12; dst = clip8(pred + R). lastX/lastY only let the
 * reference skip known-zero columns/rows; the arithmetic result is the full transform. */
void ks265o_inv_transform(int idx, const int16_t *coef, uint8_t *dst, const uint8_t *pred, int coefStride, int dstStride,
                          int predStride, int16_t *tmp, int lastX, int lastY)
{
    (void)lastX; (void)lastY;
    int M[32][32], n;
    load_matrix(idx, M, &n);
    for (int x = 0; x < n; ++x)          /* column x of coef */
        for (int y = 0; y < n; ++y) {
            int acc = 0;
            for (int k = 0; k < n; ++k) acc += M[k][y] * coef[k * coefStride + x];
            tmp[y * n + x] = clip16((acc + 64) >> 7);
        }
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            int acc = 0;
            for (int k = 0; k < n; ++k) acc += tmp[y * n + k] * M[k][x];
            dst[y * dstStride + x] = clip8((int)pred[y * predStride + x] + ((acc + 2048) >> 12));
        }
}

/* ------------------------------------------------------------------ quant / dequant */

static const int kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};  /* g_quantScales */
static const int kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};                  /* g_invQuantScales */

/* enc@0x4a9c90 H265_GetBaseQuantParam — SURVEY.md B.5 / Appendix C */
void ks265o_get_base_quant_param(int qp, int sliceType, ks265o_quant_param *p)
{
    p->scale = kQuantScales[qp % 6];
    p->qbits = 21 + qp / 6;
    p->offF = sliceType == 2 ? 171 : 85;
    p->dq = kInvQuantScales[qp % 6] << (qp / 6);
    p->minus1 = -1;
    p->per = qp / 6;
}

/* enc@0x4a9cf0 H265QuantBlock_c (via H265Quant{4,8,16,32}_c enc@0x4a9de0..) — SURVEY.md a7 / B.5 */
int ks265o_quant(const int16_t *coef, int16_t *lvl, int stride, int scale, int off, int qbits, int16_t *deltaU, int n)
{
    int nz = 0;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            int c = coef[y * stride + x];
            int a = iabs(c) * scale;
            int l = (a + off) >> qbits;
            deltaU[y * stride + x] = (int16_t)((a - (l << qbits)) >> (qbits - 8));
            if (l > 32767) l = 32767;
            if (l) ++nz;
            lvl[y * stride + x] = (int16_t)(c < 0 ? -l : l);
        }
    return nz;
}

/* enc@0x4aa150 signBitHidingHDQ (called by postQuant enc@0x4ace80 after scanSigFlags enc@0x4a9f00 when sign-data hiding is on and the block
 * has more than one level): per 4x4 coefficient group whose first and last non-zero scan positions are more than 3 apart, the sign of the
 * first level is not coded but inferred from the parity of the group's level sum; if the parity is wrong, the level whose change costs
 * least (deltaU = quantisation remainder of ks265o_quant) moves by one.  The HM-lineage algorithm (HM TComTrQuant::signBitHidingHDQ), read
 * against the disassembly: candidates n from 15 (the last group: from its last level) down to 0; a non-zero level or a zero between the
 * ends costs -|deltaU| and moves towards the remainder's sign; the first level may not drop from 1 to 0; a zero before the first level costs
 * -deltaU and only qualifies if the coefficient's sign equals the hidden sign; strictly smaller cost wins; +-32767 can only shrink.
 * lvl / coef / deltaU: N x N with `stride`; scan_idx 0 diagonal, 1 horizontal, 2 vertical.  Returns the number of non-zero levels. */
static void sbh_scan(int scan_idx, int log2, int *pos /* N*N raster positions y * N + x in coding order */)
{
    const int n = 1 << log2, nsb = n >> 2;
    int i = 0;
    int sbx[64], sby[64], px[16], py[16], k = 0;
    /* sub-blocks and positions inside a sub-block follow the same pattern (H.265 6.5.3 .. 6.5.5) */
    for (int pass = 0; pass < 2; ++pass) {
        const int size = pass ? 4 : nsb;
        int *ox = pass ? px : sbx, *oy = pass ? py : sby;
        k = 0;
        if (scan_idx == 0) {
            int x = 0, y = 0, stop = 0;
            while (!stop) {
                while (y >= 0) { if (x < size && y < size) { ox[k] = x; oy[k] = y; ++k; } --y; ++x; }
                y = x; x = 0;
                if (k >= size * size) stop = 1;
            }
        } else if (scan_idx == 1) { for (int y = 0; y < size; ++y) for (int x = 0; x < size; ++x) { ox[k] = x; oy[k] = y; ++k; } }
        else { for (int x = 0; x < size; ++x) for (int y = 0; y < size; ++y) { ox[k] = x; oy[k] = y; ++k; } }
    }
    /* 8x8 with a horizontal / vertical scan: the reference's g_iScanIdx8 tables walk the four sub-blocks in that same order */
    for (int s = 0; s < nsb * nsb; ++s)
        for (int q = 0; q < 16; ++q) pos[i++] = (sby[s] * 4 + py[q]) * n + sbx[s] * 4 + px[q];
}
int ks265o_sign_bit_hiding(int16_t *lvl, const int16_t *coef, const int16_t *deltaU, int stride, int log2, int scan_idx)
{
    const int n = 1 << log2, ncg = (n * n) >> 4;
    int pos[1024];
    sbh_scan(scan_idx, log2, pos);
#define AT(a, p) (a)[((p) / n) * stride + ((p) % n)]
    int last_cg = -1;
    for (int cg = ncg - 1; cg >= 0 && last_cg < 0; --cg)
        for (int q = 0; q < 16; ++q) if (AT(lvl, pos[cg * 16 + q])) { last_cg = cg; break; }
    for (int cg = last_cg; cg >= 0; --cg) {
        const int *sp = pos + cg * 16;
        int first = 16, last = -1, sum = 0;
        for (int q = 15; q >= 0; --q) if (AT(lvl, sp[q])) { last = q; break; }
        for (int q = 0; q < 16; ++q) if (AT(lvl, sp[q])) { first = q; break; }
        if (last - first < 4) continue;
        for (int q = first; q <= last; ++q) sum += AT(lvl, sp[q]);
        const int signbit = AT(lvl, sp[first]) > 0 ? 0 : 1;
        if (signbit == (sum & 1)) continue;
        int min_cost = 0x7fffffff, min_pos = -1, final_change = 0;
        for (int q = (cg == last_cg ? last : 15); q >= 0; --q) {
            const int p = sp[q], l = AT(lvl, p), du = AT(deltaU, p);
            int cost, change;
            if (l != 0) {
                if (du > 0) { cost = -du; change = 1; }
                else if (q == first && (l == 1 || l == -1)) { cost = 0x7fffffff; change = 0; }
                else { cost = du; change = -1; }
            } else if (q < first) {
                const int this_sign = AT(coef, p) >= 0 ? 0 : 1;
                if (this_sign != signbit) { cost = 0x7fffffff; change = 0; }
                else { cost = -du; change = 1; }
            } else { cost = -du; change = 1; }
            if (cost < min_cost) { min_cost = cost; final_change = change; min_pos = p; }
        }
        if (min_pos < 0) continue;
        if (AT(lvl, min_pos) == 32767 || AT(lvl, min_pos) == -32768) final_change = -1;
        if (AT(coef, min_pos) >= 0) AT(lvl, min_pos) = (int16_t)(AT(lvl, min_pos) + final_change);
        else AT(lvl, min_pos) = (int16_t)(AT(lvl, min_pos) - final_change);
    }
    int nz = 0;
    for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) nz += lvl[y * stride + x] != 0;
#undef AT
    return nz;
}

/* enc@0x439210 H265DeQuantBlock_c — SURVEY.md a8 / B.5 */
void ks265o_dequant(const int16_t *lvl, int16_t *coef, int stride, int scale, int add, int shift, int lastX, int lastY)
{
    int cols = (lastX + 4) & ~3;
    for (int y = 0; y <= lastY; ++y)
        for (int x = 0; x < cols; ++x) coef[y * stride + x] = clip16((lvl[y * stride + x] * scale + add) >> shift);
}

/* ------------------------------------------------------------------ deblocking */

/* uiTCTable enc@0x4dc3e0, uiBetaTable enc@0x4dc3a0 (normative HEVC tables 8-12) */
const uint8_t ks265o_tc_table[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                     2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
const uint8_t ks265o_beta_table[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                       16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54,
                                       56, 58, 60, 62, 64};

/* one 4-line luma segment; xs = step across the edge, ys = step along it */
static void luma_segment(uint8_t *pix, long xs, long ys, int beta, int tc, int filterP, int filterQ)
{
#define P(i, l) ((int)pix[(l) * ys - ((i) + 1) * xs])
#define Q(i, l) ((int)pix[(l) * ys + (i) * xs])
    int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
    int dq0 = iabs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = iabs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
    int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
    if (d >= beta) return;
    int s0 = (2 * dpq0 < (beta >> 2)) && (iabs(P(3, 0) - P(0, 0)) + iabs(Q(0, 0) - Q(3, 0)) < (beta >> 3)) &&
             (iabs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1));
    int s3 = (2 * dpq3 < (beta >> 2)) && (iabs(P(3, 3) - P(0, 3)) + iabs(Q(0, 3) - Q(3, 3)) < (beta >> 3)) &&
             (iabs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1));
    int side = (beta + (beta >> 1)) >> 3;
    int dEp = dp < side, dEq = dq < side;
    for (int l = 0; l < 4; ++l) {
        int p0 = P(0, l), p1 = P(1, l), p2 = P(2, l), p3 = P(3, l);
        int q0 = Q(0, l), q1 = Q(1, l), q2 = Q(2, l), q3 = Q(3, l);
        uint8_t *pp = pix + l * ys;
        if (s0 && s3) {
            if (filterP) {
                pp[-1 * xs] = clip8(clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
                pp[-2 * xs] = clip8(clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2));
                pp[-3 * xs] = clip8(clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
            }
            if (filterQ) {
                pp[0] = clip8(clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
                pp[1 * xs] = clip8(clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2));
                pp[2 * xs] = clip8(clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
            }
        } else {
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (iabs(delta) < 10 * tc) {
                delta = clip3(-tc, tc, delta);
                if (filterP) {
                    pp[-1 * xs] = clip8(p0 + delta);
                    if (dEp) pp[-2 * xs] = clip8(p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
                }
                if (filterQ) {
                    pp[0] = clip8(q0 - delta);
                    if (dEq) pp[1 * xs] = clip8(q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
                }
            }
        }
    }
#undef P
#undef Q
}

/* enc@0x403630 EdgeFilterLumaVer_c — SURVEY.md B.6: vertical edge, lines advance by stride */
void ks265o_edge_filter_luma_ver(uint8_t *pix, int stride, int beta, int tc, int length, int filterP, int filterQ)
{
    for (int s = 0; s < length / 4; ++s) luma_segment(pix + (long)s * 4 * stride, 1, stride, beta, tc, filterP, filterQ);
}

/* enc@0x4038c0 EdgeFilterLumaHor_c — horizontal edge, lines advance by 1 */
void ks265o_edge_filter_luma_hor(uint8_t *pix, int stride, int beta, int tc, int length, int filterP, int filterQ)
{
    for (int s = 0; s < length / 4; ++s) luma_segment(pix + s * 4, stride, 1, beta, tc, filterP, filterQ);
}

static void chroma_line(uint8_t *pix, long xs, int tc, int filterP, int filterQ)
{
    int p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
    int delta = clip3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3);
    if (filterP) pix[-xs] = clip8(p0 + delta);
    if (filterQ) pix[0] = clip8(q0 - delta);
}

/* enc@0x403c50 PixelFilterChromaVer_c(pix, stride, tc, length, filterP, filterQ) */
void ks265o_pixel_filter_chroma_ver(uint8_t *pix, int stride, int tc, int length, int filterP, int filterQ)
{
    for (int l = 0; l < length; ++l) chroma_line(pix + (long)l * stride, 1, tc, filterP, filterQ);
}

/* enc@0x403d10 PixelFilterChromaHor_c */
void ks265o_pixel_filter_chroma_hor(uint8_t *pix, int stride, int tc, int length, int filterP, int filterQ)
{
    for (int l = 0; l < length; ++l) chroma_line(pix + l, stride, tc, filterP, filterQ);
}

/* ------------------------------------------------------------------ interpolation */

static const int8_t kLumaTaps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0},
                                       {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t kChromaTaps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                         {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

#define INTERP_BODY(SRC_T, TAPS, NT, STEP, EXPR)                                             \
    for (int y = 0; y < h; ++y)                                                              \
        for (int x = 0; x < w; ++x) {                                                        \
            int sum = 0;                                                                     \
            for (int i = 0; i < NT; ++i) sum += TAPS[frac][i] * (int)src[y * srcStride + x + (i - (NT / 2 - 1)) * (STEP)]; \
            dst[y * dstStride + x] = EXPR;                                                   \
        }

/* enc@0x40e4f0 interpLumaHor8to8_c — SURVEY.md B.7 */
void ks265o_interp_luma_hor_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kLumaTaps, 8, 1, clip8((sum + 32) >> 6)) }
/* enc@0x40f0c0 interpLumaVer8to8_c */
void ks265o_interp_luma_ver_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kLumaTaps, 8, srcStride, clip8((sum + 32) >> 6)) }
/* enc@0x40eb80 interpLumaHor8to16_c: 14-bit intermediate = raw tap sum (NO -8192 offset, unlike HM) */
void ks265o_interp_luma_hor_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kLumaTaps, 8, 1, (int16_t)sum) }
/* enc@0x40f950 interpLumaVer8to16_c */
void ks265o_interp_luma_ver_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kLumaTaps, 8, srcStride, (int16_t)sum) }
/* enc@0x4100b0 interpLumaVer16to8_c: second stage, (sum + 2^11) >> 12 */
void ks265o_interp_luma_ver_16to8(uint8_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(int16_t, kLumaTaps, 8, srcStride, clip8((sum + 2048) >> 12)) }
/* enc@0x4109b0 interpLumaVer16to16_c: second stage kept at 14 bit, sum >> 6 */
void ks265o_interp_luma_ver_16to16(int16_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(int16_t, kLumaTaps, 8, srcStride, (int16_t)(sum >> 6)) }
/* enc@0x4111c0.. interpChroma* — 4-tap, 1/8-sample */
void ks265o_interp_chroma_hor_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kChromaTaps, 4, 1, clip8((sum + 32) >> 6)) }
void ks265o_interp_chroma_ver_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kChromaTaps, 4, srcStride, clip8((sum + 32) >> 6)) }
void ks265o_interp_chroma_hor_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kChromaTaps, 4, 1, (int16_t)sum) }
void ks265o_interp_chroma_ver_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(uint8_t, kChromaTaps, 4, srcStride, (int16_t)sum) }
void ks265o_interp_chroma_ver_16to8(uint8_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(int16_t, kChromaTaps, 4, srcStride, clip8((sum + 2048) >> 12)) }
void ks265o_interp_chroma_ver_16to16(int16_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac)
{ INTERP_BODY(int16_t, kChromaTaps, 4, srcStride, (int16_t)(sum >> 6)) }

/* ------------------------------------------------------------------ SAO */

/* enc@0x43e4e0 SaoApplyOffsetBo_c — SURVEY.md B.9.  Read from the disassembly and pinned by the fixtures:
 * columns are processed in groups of 4 (width rounded UP to a multiple of 4), and offsets[k] goes to
 * table[bandPosition + k] without the normative "& 31" wrap (entries past 31 fall off the 32-entry table;
 * the reference encoder never selects bandPosition > 28). */
void ks265o_sao_apply_bo(const int8_t *offsets, uint8_t *rec, int stride, int height, int width, int bandPosition)
{
    int8_t table[32];
    memset(table, 0, sizeof table);
    for (int k = 0; k < 4; ++k)
        if (bandPosition + k < 32) table[bandPosition + k] = offsets[k];
    int cols = (width + 3) & ~3;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < cols; ++x) rec[y * stride + x] = clip8((int)rec[y * stride + x] + table[rec[y * stride + x] >> 3]);
}

/* enc@0x4ae9c0 statSaoBoEo01_c — SURVEY.md B.10: packed (sum<<12 | count) accumulators */
void ks265o_stat_sao_bo_eo01(int *eoJoint, int *bo, const uint8_t *org, const uint8_t *rec, int recStride, int orgStride,
                             int width, int height, int rowStep)
{
    for (int y = 0; y < height; y += rowStep)
        for (int x = 0; x < width; ++x) {
            const uint8_t *r = rec + y * recStride + x;
            int d = (int8_t)(org[y * orgStride + x] - r[0]);
            int v = (int)(((unsigned)d << 12) | 1u);
            int c0 = 2 + sgn((int)r[0] - (int)r[-1]) + sgn((int)r[0] - (int)r[1]);
            int c1 = 2 + sgn((int)r[0] - (int)r[-recStride]) + sgn((int)r[0] - (int)r[recStride]);
            bo[r[0] >> 3] += v;
            eoJoint[(c1 << 3) | c0] += v;
        }
}

/* enc@0x43e650 SaoApplyOffsetEo0_c / 0x43e970 Eo1 / 0x43edc0 Eo2 / 0x43ef70 Eo3 — "plain" mode (trailing
 * flag arguments 0: every neighbour is read from the picture itself, un-SAO'd).  `offsets` is the 5-entry
 * table indexed by the raw edge index 2 + sign(c-a) + sign(c-b) (read from the disassembly: movsbl (%rdi,idx)).
 * The saved-line modes of the reference only exist because its CTU pipeline filters in place; the frame-level
 * kernels are out-of-place, which is what this model computes.  All four classes are pinned: EO0/EO1 with their
 * flag arguments 0, EO2/EO3 (which always read a saved row/column and take a centred offset pointer) with the saved
 * row/column pointed into the picture itself (oracle/ref_probe/gen_golden.py). */
void ks265o_sao_apply_eo(int cls, const int8_t *offsets, uint8_t *rec, int stride, int height, int width)
{
    static const int dx[4] = {1, 0, 1, -1}, dy[4] = {0, 1, 1, 1};
    uint8_t *copy = (uint8_t *)malloc((size_t)(height + 2) * (size_t)(width + 2));
    int cs = width + 2;
    for (int y = -1; y <= height; ++y)
        for (int x = -1; x <= width; ++x) copy[(y + 1) * cs + x + 1] = rec[y * stride + x];
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            int c = copy[(y + 1) * cs + x + 1];
            int a = copy[(y + 1 - dy[cls]) * cs + x + 1 - dx[cls]], b = copy[(y + 1 + dy[cls]) * cs + x + 1 + dx[cls]];
            rec[y * stride + x] = clip8(c + offsets[2 + sgn(c - a) + sgn(c - b)]);
        }
    free(copy);
}

/* ------------------------------------------------------------------ bi-prediction helpers (second wave, EncInterMeBi* / ComInterPrediction*)
 * enc@0x435160 DefaultWeightedBi_c(dst, p0, p1, dstStride, srcStride, width, height): average of two 14-bit predictions.
 * The reference's 14-bit intermediates carry no -8192 offset (see interpLumaHor8to16_c), so the normative
 * (a + b + 2^14 + 64) >> 7 becomes (a + b + 64) >> 7.  Pinned by tests/golden/bipred.npz. */
void ks265o_default_weighted_bi(uint8_t *dst, const int16_t *p0, const int16_t *p1, int dstStride, int srcStride, int width, int height)
{
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) dst[y * dstStride + x] = clip8(((int)p0[y * srcStride + x] + (int)p1[y * srcStride + x] + 64) >> 7);
}

/* CEncSao::estIterOffset enc@0x4adbe0: the reference's choice of one SAO offset (HM lineage: TEncSampleAdaptiveOffset::estIterOffset in integers).  Starting at
 * *offset (the rounded mean difference of the class, clipped by the caller) the candidate walks towards zero, zero excluded; a candidate costs
 * count * off^2 - 2 * off * diffSum + ((lambda_q8 * (rate_base + |off| + 1) + 128) >> 8); whenever that is strictly below *best_cost it becomes the answer.
 * *offset leaves as 0 if nothing beat the preset *best_cost.  32-bit arithmetic as in the binary.  Pinned by tests/golden/sao_iter.npz.
 * (The frame stage's own SAO decision - sao_offset / sao_eval of the pipeline oracle - is a different, simpler rule; this is the reference's, for the day it replaces it.) */
void ks265o_sao_est_iter_offset(int lambda_q8, int rate_base, int32_t *offset, int count, int diff_sum, int32_t *best_cost)
{
    int off = *offset;
    const int step = off <= 0 ? 1 : -1;
    *offset = 0;
    for (; off != 0; off += step) {
        const int32_t dist = (int32_t)((uint32_t)off * ((uint32_t)count * (uint32_t)off - 2u * (uint32_t)diff_sum));
        const int32_t rate = (int32_t)(((uint32_t)lambda_q8 * (uint32_t)(rate_base + iabs(off) + 1) + 128u)) >> 8;
        const int32_t cost = (int32_t)((uint32_t)dist + (uint32_t)rate);
        if (cost < *best_cost) { *offset = off; *best_cost = cost; }
    }
}

/* CEncSao::BoTypeDistEstimation enc@0x4adc70 / CEncSao::EoTypeDistEstimation enc@0x4adf60: the reference's SAO offsets and type costs from the class statistics
 * (count[k] = samples of class k, sum[k] = sum of (source - reconstruction) over them).  Per class: no samples -> sum is cleared, offset 0; else the start value is
 * the mean rounded to nearest (half away from zero) and clipped to [-3, 3], refined by estIterOffset against the price of one bin ((lambda + 128) >> 8, which is
 * also what an unused class costs); edge categories 0, 1 only take positive offsets, 2, 3 only negative ones (else 0).  Band offset: rate base 1, the band
 * position is the first of the 28 windows of four consecutive bands with the smallest cost sum.  Edge offset: rate base 0, returns the cost sum of the four
 * categories.  Pinned by tests/golden/sao_type.npz.  Their callers add the type's own rate and keep the cheapest (read from the binary, not probed: calcRDcostBoY
 * enc@0x4ade40: window cost + ((7 lambda + 128) >> 8); calcRDcostEoY enc@0x4ae290: category cost + ((4 lambda + 128) >> 8); checkRDCostY enc@0x4ad810: strictly cheaper wins). */
static int32_t sao_start_offset(int32_t d, int32_t count, int sign_half)   /* (d + sign * count >> 1) / count (C division), clipped */
{
    int32_t v = (d + ((sign_half * count) >> 1)) / count;
    return v > 3 ? 3 : (v < -3 ? -3 : v);
}
int32_t ks265o_sao_bo_type_estimation(int lambda_q8, int32_t *count /*32*/, int32_t *sum /*32*/, int32_t *band, int32_t *offsets /*32*/)   /* returns the chosen window's cost (eax of the original) */
{
    const int32_t zero = (lambda_q8 + 128) >> 8;
    int32_t cost[32];
    for (int k = 0; k < 32; ++k) {
        if (!count[k]) { sum[k] = 0; cost[k] = zero; offsets[k] = 0; continue; }
        int32_t off = sao_start_offset(sum[k], count[k], (sum[k] > 0) - (sum[k] < 0)), best = zero;
        ks265o_sao_est_iter_offset(lambda_q8, 1, &off, count[k], sum[k], &best);
        offsets[k] = off; cost[k] = best;
    }
    int32_t bc = 0xffff000;
    for (int k = 0; k < 28; ++k) {
        const int32_t c = cost[k] + cost[k + 1] + cost[k + 2] + cost[k + 3];
        if (c < bc) { bc = c; *band = k; }
    }
    return bc;
}
int32_t ks265o_sao_eo_type_estimation(int lambda_q8, const int32_t *count /*4*/, int32_t *sum /*4*/, int32_t *offsets /*4*/)
{
    const int32_t zero = (lambda_q8 + 128) >> 8;
    int32_t total = 0;
    for (int k = 0; k < 4; ++k) {
        offsets[k] = 0;
        if (!count[k]) { sum[k] = 0; total += zero; continue; }
        const int positive = k < 2;
        if (positive ? sum[k] <= 0 : sum[k] >= 0) { total += zero; continue; }
        int32_t off = sao_start_offset(sum[k], count[k], positive ? 1 : -1), best = zero;
        ks265o_sao_est_iter_offset(lambda_q8, 0, &off, count[k], sum[k], &best);
        offsets[k] = off; total += best;
    }
    return total;
}

/* CEncSao::modeDecisionCtu enc@0x4af690 on its `-sao 4` path (also what an I slice takes at level 3), restated from the disassembly and pinned on calls recorded inside real
 * encodes (oracle/ref_probe/sao_shim.c, tests/golden/sao_decision.npz):
 *   stats = the object's statistics, 312 words: counts - band offset Y [0..32), U [32..64), V [64..96), edge classes Y at 96 + 5 class, U at 116 + 5 class, V at 136 + 5 class
 *           (four categories each) - then the sums in the same layout from word 156 on;
 *   records (32 bytes): [0] luma type (-1 off, 0 / 1 = edge class 0 / 90 degrees, 4 = band offset), [1] chroma type, [2] luma band, [3] / [4] Cb / Cr band, [5..8] luma offsets,
 *           [0xa..0xd] Cb, [0xf..0x12] Cr, [0x14] merge left, [0x15] merge up.
 * modeDecisionBoEo01 enc@0x4af300: luma EO 0, EO 1, BO, then chroma EO 0, EO 1, BO - calcRDcostEoY enc@0x4ae290 (+ (4 lambda + 128) >> 8), calcRDcostBoY enc@0x4ade40 (7 lambda),
 * calcRDcostEoUV enc@0x4ae2e0 (both components + 4 lambda_c), calcRDcostBoUV enc@0x4adeb0 (12 lambda_c, a band per component); checkRDCostY / UV enc@0x4ad810 / 0x4ad860: strictly
 * cheaper than the best so far, which starts at one bin ((lambda + 128) >> 8).  Then the merge candidates: the sum of the two bests + ((lambda_avg if an upper CTU exists) + 128) >> 8
 * against checkMerge enc@0x4ae7f0 of the left CTU's FINAL record (the distortion its offsets give on this CTU's statistics; 0xfffffff when it uses a type outside the masks) and of
 * the upper one's + (lambda_avg + 128) >> 8; strictly cheaper copies the neighbour's record (24 bytes) and sets the flag.  32-bit arithmetic as in the binary. */
static int32_t sao_merge_dist(const int32_t *st, const int8_t *p, int mask_y, int mask_uv)
{
    const int ty = p[0], tuv = p[1];
    if (ty != -1 && !((mask_y >> ty) & 1)) return 0xfffffff;
    if (tuv != -1 && !((mask_uv >> tuv) & 1)) return 0xfffffff;
    uint32_t d = 0;
    if (ty != -1) {
        const int32_t *cnt = ty == 4 ? st + p[2] : st + 96 + 5 * ty, *sum = ty == 4 ? st + 156 + p[2] : st + 156 + 96 + 5 * ty;
        for (int i = 0; i < 4; ++i) { const int32_t o = p[5 + i]; d += (uint32_t)(((uint32_t)cnt[i] * (uint32_t)o - 2u * (uint32_t)sum[i]) * (uint32_t)o); }
    }
    if (tuv != -1)
        for (int c = 0; c < 2; ++c) {
            const int32_t *cnt = tuv == 4 ? st + 32 + 32 * c + p[3 + c] : st + 116 + 20 * c + 5 * tuv, *sum = tuv == 4 ? st + 156 + 32 + 32 * c + p[3 + c] : st + 156 + 116 + 20 * c + 5 * tuv;
            for (int i = 0; i < 4; ++i) { const int32_t o = p[0xa + 5 * c + i]; d += (uint32_t)(((uint32_t)cnt[i] * (uint32_t)o - 2u * (uint32_t)sum[i]) * (uint32_t)o); }
        }
    return (int32_t)d;
}
void ks265o_sao_mode_decision(const int32_t *stats /*312*/, int lam_y, int lam_c, int left_avail, int up_avail, const int8_t *left /*32 or NULL*/, const int8_t *up, int mask_y, int mask_uv,
                              int8_t *out /*32: only the bytes the function writes are touched*/, int32_t *best /*2*/)
{
    const int32_t lam_avg = (lam_y + lam_c + 1) >> 1;
    out[0] = out[1] = -1; out[0x14] = out[0x15] = 0;
    int32_t by = (lam_y + 128) >> 8, buv = (lam_c + 128) >> 8;
    for (int cls = 0; cls < 2; ++cls) {
        int32_t cnt[4], sum[4], off[4];
        for (int k = 0; k < 4; ++k) { cnt[k] = stats[96 + 5 * cls + k]; sum[k] = stats[156 + 96 + 5 * cls + k]; }
        const int32_t cost = ks265o_sao_eo_type_estimation(lam_y, cnt, sum, off) + ((4 * lam_y + 128) >> 8);
        if (by > cost) { by = cost; out[0] = (int8_t)cls; out[2] = 0; for (int k = 0; k < 4; ++k) out[5 + k] = (int8_t)off[k]; }
    }
    {
        int32_t cnt[32], sum[32], off[32], band = 0;
        memcpy(cnt, stats, sizeof cnt); memcpy(sum, stats + 156, sizeof sum);
        const int32_t cost = ks265o_sao_bo_type_estimation(lam_y, cnt, sum, &band, off) + ((7 * lam_y + 128) >> 8);
        if (by > cost) { by = cost; out[0] = 4; out[2] = (int8_t)band; for (int k = 0; k < 4; ++k) out[5 + k] = (int8_t)off[band + k]; }
    }
    for (int cls = 0; cls < 2; ++cls) {
        int32_t off[2][4], cost = (4 * lam_c + 128) >> 8;
        for (int c = 0; c < 2; ++c) {
            int32_t cnt[4], sum[4];
            for (int k = 0; k < 4; ++k) { cnt[k] = stats[116 + 20 * c + 5 * cls + k]; sum[k] = stats[156 + 116 + 20 * c + 5 * cls + k]; }
            cost += ks265o_sao_eo_type_estimation(lam_c, cnt, sum, off[c]);
        }
        if (buv > cost) { buv = cost; out[1] = (int8_t)cls; out[3] = out[4] = 0; for (int k = 0; k < 4; ++k) { out[0xa + k] = (int8_t)off[0][k]; out[0xf + k] = (int8_t)off[1][k]; } }
    }
    {
        int32_t off[2][32], band[2] = {0, 0}, cost = (12 * lam_c + 128) >> 8;
        for (int c = 0; c < 2; ++c) {
            int32_t cnt[32], sum[32];
            memcpy(cnt, stats + 32 + 32 * c, sizeof cnt); memcpy(sum, stats + 156 + 32 + 32 * c, sizeof sum);
            cost += ks265o_sao_bo_type_estimation(lam_c, cnt, sum, &band[c], off[c]);
        }
        if (buv > cost) { buv = cost; out[1] = 4; out[3] = (int8_t)band[0]; out[4] = (int8_t)band[1]; for (int k = 0; k < 4; ++k) { out[0xa + k] = (int8_t)off[0][band[0] + k]; out[0xf + k] = (int8_t)off[1][band[1] + k]; } }
    }
    if (best) { best[0] = by; best[1] = buv; }
    int32_t tot = by + buv + (((up_avail ? lam_avg : 0) + 128) >> 8);
    if (left_avail && left) {
        const int32_t c = sao_merge_dist(stats, left, mask_y, mask_uv);
        if (tot > c) { tot = c; memcpy(out, left, 24); out[0x15] = 0; out[0x14] = 1; }
    }
    if (up_avail && up) {
        const int32_t c = sao_merge_dist(stats, up, mask_y, mask_uv) + ((lam_avg + 128) >> 8);
        if (tot > c) { memcpy(out, up, 24); out[0x15] = 1; out[0x14] = 0; }
    }
}

/* CalcBsInterP enc@0x402960 / CalcBsInterB enc@0x4029d0 (TNborData *p, TNborData *q, int transform edge): boundary strength of the edge between two blocks
 * (H.265 8.7.2.4 in the reference's data layout).  A block record is three words: word 0 - bits 2..3 lists used (0 = intra), bits 16..19 / 20..23 the reference
 * PICTURE id of list 0 / 1 (ids compare across lists), bit 24 coded residual; bytes 4..7 the list-0 vector, 8..11 the list-1 vector (quarter samples).
 * 2 if p is intra (the callers pass the pair both ways round); 1 on a transform edge with residual on either side; 1 if the blocks use different reference
 * pictures or a different number of vectors, or any paired vector component differs by 4 or more; for two bi-predictive blocks on the same two pictures both
 * pairings count when the two pictures are one.  The P variant looks at list 0 only.  Pinned by tests/golden/bs.npz (800 cases). */
static int bs_far(const int16_t *a, const int16_t *b) { return iabs(a[0] - b[0]) > 3 || iabs(a[1] - b[1]) > 3; }
int ks265o_calc_bs(const int32_t *p, const int32_t *q, int tu_edge, int is_b)
{
    const uint32_t wp = (uint32_t)p[0], wq = (uint32_t)q[0];
    const int16_t *mp = (const int16_t *)p + 2, *mq = (const int16_t *)q + 2;          /* [0..1] list 0, [2..3] list 1 */
    if (!(wp & 0xc)) return 2;
    if ((((wp | wq) >> 24) & 1) && tu_edge) return 1;
    if (!is_b) return ((wp ^ wq) & 0xf0000) ? 1 : bs_far(mp, mq);
    const unsigned lp = (wp >> 2) & 3, lq = (wq >> 2) & 3;
    if ((lp == 3) != (lq == 3)) return 1;                                              /* one vector against two */
    if (lp != 3) {                                                                      /* one vector each, whichever list it lives in */
        const int ip = lp >> 1, iq = lq >> 1;
        if (((wp >> (16 + 4 * ip)) & 15) != ((wq >> (16 + 4 * iq)) & 15)) return 1;
        return bs_far(mp + 2 * ip, mq + 2 * iq);
    }
    const unsigned p0 = (wp >> 16) & 15, p1 = (wp >> 20) & 15, q0 = (wq >> 16) & 15, q1 = (wq >> 20) & 15;
    const int straight = p0 == q0 && p1 == q1, crossed = p0 == q1 && p1 == q0;
    if (!straight && !crossed) return 1;
    const int far_straight = bs_far(mp, mq) || bs_far(mp + 2, mq + 2), far_crossed = bs_far(mp, mq + 2) || bs_far(mp + 2, mq);
    if (p0 == p1) return far_straight && far_crossed;                                   /* both pictures are one: either pairing may match */
    return straight ? far_straight : far_crossed;
}

/* enc@0x46a8a0 estBitRdoq(TEstBitsSbac &, log2 size, is luma, context states) = estCBFBit enc@0x46a480 + estSignificantCoeffGroupMapBit enc@0x46a4f0 +
 * estSignificantMapBit enc@0x46a540 + estSignificantCoefficientsBit enc@0x46a790: the bit-estimation tables rdoQuant enc@0x4aac50 prices its decisions with
 * (HM lineage: TComTrQuant::xRateDistOptQuant / TEncSbac::estBit).  ctx = the encoder's CABAC context states, one byte each (pStateIdx << 1 | valMps) in the
 * reference's own order: cbf at 0x0d (+5 for chroma), coded_sub_block_flag at 0x1d (+2), sig_coeff_flag at 0x21 (luma) / 0x3c (chroma), last_sig_coeff prefix at
 * 0x4b (x) / 0x69 (y), greater1 at 0x87 (luma, 16) / 0x97 (chroma, 8), greater2 at 0x9f (luma, 4) / 0xa3 (chroma, 2), rqt_root_cbf at 0xaa.  entropy = the 128
 * entries of g_iEntroyBits enc@0x4e0040 (bits x 2^15 of coding bin b in state s: entropy[s ^ b]) - data, passed in: the tests take it from the fixtures.
 * out = TEstBitsSbac as 180 words: [0..3] group flag [2][2]; [4..45] sig flag = 0 per context, [46..87] sig flag = 1; [88..97] last-x prefix bits, [98..107]
 * last-y; [108..139] greater1 [16][2]; [156..163] greater2 [4][2]; [168..177] cbf [5][2]; [178..179] root cbf.  Words not listed keep their value.
 * Pinned by tests/golden/estbits.npz. */
void ks265o_est_bit_rdoq(int32_t *out, int log2, int luma, const uint8_t *ctx, const int32_t *entropy)
{
    const uint8_t *c = ctx + 0x0d + (luma ? 0 : 5);
    for (int i = 0; i < 5; ++i) { out[168 + 2 * i] = entropy[c[i]]; out[169 + 2 * i] = entropy[c[i] ^ 1]; }
    out[178] = entropy[ctx[0xaa]]; out[179] = entropy[ctx[0xaa] ^ 1];
    c = ctx + 0x1d + (luma ? 0 : 2);
    for (int i = 0; i < 2; ++i) { out[2 * i] = entropy[c[i]]; out[2 * i + 1] = entropy[c[i] ^ 1]; }
    /* sig_coeff_flag: context 0 (DC) and the size's own range */
    c = ctx + (luma ? 0x21 : 0x3c);
    int first, end;
    if (log2 > 3) { first = luma ? 21 : 12; end = luma ? 27 : 15; }
    else if (log2 == 3) { first = 9; end = luma ? 21 : 12; }
    else { first = 1; end = 9; }
    out[4] = entropy[c[0]]; out[46] = entropy[c[0] ^ 1];
    for (int i = first; i < end; ++i) { out[4 + i] = entropy[c[i]]; out[46 + i] = entropy[c[i] ^ 1]; }
    /* last_sig_coeff_{x,y}_prefix: bits of "k ones then a zero", the last entry all ones */
    const int off = luma ? 3 * log2 - 6 + ((log2 - 1) >> 2) : 15, shift = luma ? (log2 + 1) >> 2 : log2 - 2, n = 2 * log2 - 1;
    for (int d = 0; d < 2; ++d) {
        const uint8_t *l = ctx + 0x4b + 0x1e * d;
        int32_t bits = 0;
        for (int k = 0; k < n; ++k) {
            const uint8_t s = l[off + (k >> shift)];
            out[88 + 10 * d + k] = bits + entropy[s];
            bits += entropy[s ^ 1];
        }
        out[88 + 10 * d + n] = bits;
    }
    if (luma) {
        for (int i = 0; i < 16; ++i) { out[108 + 2 * i] = entropy[ctx[0x87 + i]]; out[109 + 2 * i] = entropy[ctx[0x87 + i] ^ 1]; }
        for (int i = 0; i < 4; ++i) { out[156 + 2 * i] = entropy[ctx[0x9f + i]]; out[157 + 2 * i] = entropy[ctx[0x9f + i] ^ 1]; }
    } else {
        for (int i = 0; i < 8; ++i) { out[108 + 2 * i] = entropy[ctx[0x97 + i]]; out[109 + 2 * i] = entropy[ctx[0x97 + i] ^ 1]; }
        for (int i = 0; i < 2; ++i) { out[156 + 2 * i] = entropy[ctx[0xa3 + i]]; out[157 + 2 * i] = entropy[ctx[0xa3 + i] ^ 1]; }
    }
}

/* enc@0x4896d0 interMeBiFull_c / enc@0x4897e0 interMeBiHadFull_c (best, org, ref, orgStride, refStride, mvcost, h, log2w): the integer step of the joint
 * bi-prediction refinement (g_interMeBiFull_func / g_interMeBiHadFull_func; caller interMeBiFull_opt enc@0x4898e0).  `org` is the search target
 * clip8(2 org - pred_other) of calcBiMeOrg, `ref` the top-left corner of an 8 x 8 window of integer positions; position (x, y) costs
 * SAD (or HAD) + mvcost[x] + mvcost[8 + y], rows outside, columns inside, a later position wins only if strictly cheaper (unsigned compare against
 * 0xfffffff at the start).  Returns the cost, *best = (y << 16) | x.  Pinned by tests/golden/bifull.npz. */
uint32_t ks265o_inter_me_bi_full(int32_t *best, const uint8_t *org, const uint8_t *ref, int orgStride, int refStride, const uint16_t *mvcost, int h, int log2w, int use_had)
{
    uint32_t bc = 0xfffffffu;
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            const uint8_t *r = ref + (long)y * refStride + x;
            uint32_t c = (use_had ? ks265o_had(org, r, orgStride, refStride, h, 1 << log2w) : ks265o_sad(org, r, orgStride, refStride, h, 1 << log2w))
                         + mvcost[x] + mvcost[8 + y];
            if (c < bc) { bc = c; *best = (y << 16) | x; }
        }
    return bc;
}

/* enc@0x434510 ExplicitWeightedP_c / enc@0x434460 ExplicitWeightedBi_c: explicit weighted prediction (H.265 8.5.3.3.4.3) on the 14-bit intermediates, which in
 * this codec carry no -8192 offset.  wp = WeightParams as the two functions read it: {shift, w0, o0, -, w1, o1}; shift = log2WD of the standard (>= 1).
 *   uni: clip8(((p * w0 + 2^(shift-1)) >> shift) + o0)        bi: clip8((p0 * w0 + p1 * w1 + ((o0 + o1 + 1) << shift)) >> (shift + 1))
 * Pinned by tests/golden/wpred.npz.  The frame stages do not use weighted prediction (the encoder host never signals it). */
void ks265o_explicit_weighted_p(uint8_t *dst, const int16_t *p0, int dstStride, int srcStride, int width, int height, const int32_t *wp)
{
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) dst[y * dstStride + x] = clip8((((int)p0[y * srcStride + x] * wp[1] + (1 << (wp[0] - 1))) >> wp[0]) + wp[2]);
}
void ks265o_explicit_weighted_bi(uint8_t *dst, const int16_t *p0, const int16_t *p1, int dstStride, int srcStride, int width, int height, const int32_t *wp)
{
    const int rnd = (int)((uint32_t)(wp[2] + wp[5] + 1) << wp[0]);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) dst[y * dstStride + x] = clip8(((int)p0[y * srcStride + x] * wp[1] + (int)p1[y * srcStride + x] * wp[4] + rnd) >> (wp[0] + 1));
}

/* enc@0x47b1a0 calcBiMeOrg_c(dst, pred, org, stride, height, width): the bi-pred search target dst = clip8(2 org - pred)
 * (g_calcBiMeOrgFuncs); returns the clipping loss sum |2 org - pred - dst|. */
uint32_t ks265o_calc_bi_me_org(uint8_t *dst, const uint8_t *pred, const uint8_t *org, int stride, int height, int width)
{
    uint32_t loss = 0;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            int v = 2 * (int)org[y * stride + x] - (int)pred[y * stride + x];
            uint8_t c = clip8(v);
            dst[y * stride + x] = c;
            loss += (uint32_t)iabs(v - (int)c);
        }
    return loss;
}
