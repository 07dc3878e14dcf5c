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

