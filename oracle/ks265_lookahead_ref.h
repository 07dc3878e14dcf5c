/* ks265_lookahead_ref.h - TEST INFRASTRUCTURE (oracle): the reference's lookahead decisions, see ks265_lookahead_ref.c */
#ifndef KS265_LOOKAHEAD_REF_H
#define KS265_LOOKAHEAD_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* calcFrameAdaptQuant enc@0x4653c0 (mode 1): per 16x16 block the QP offset (double) and its inverse qscale factor (fix 8).  Y / U / V: packed planes of 16 nx x 16 ny and
 * 8 nx x 8 ny samples; count = the divisor of the mean (the reference's block count word, nx * ny in every recorded call) */
void kso_ref_frame_adapt_quant(const uint8_t *Y, const uint8_t *U, const uint8_t *V, int nx, int ny, int count, double strength, double *qp_off, uint16_t *inv_qscale);
/* cuTreePropagate enc@0x47d460: one step of the propagation from picture b into its references (ref0 / ref1: their propagate-cost planes, updated).
 * lg: the function's first argument; n = nx * ny blocks; list bits: 2 bits per block, 4 blocks per byte; mv: x = low 16 bits (signed), y = high 16 */
void kso_ref_cutree_propagate(int lg, int nx, int ny, const uint16_t *intra, const uint16_t *inv_qscale, const uint16_t *own, const uint16_t *inter, const uint8_t *list_bits,
                              const int32_t *mv0, const int32_t *mv1, uint16_t *ref0, uint16_t *ref1);
/* scenecut enc@0x47e9d0 once calcFrameCost has filled the costs: pcost = cost of the picture predicted from the previous one, icost / prev_icost = intra costs (prev -1: none),
 * blocks = nx * ny, lg = cfg+0x3c0, thr = cfg+0x390 (-scenecut), keyint = cfg+0x50, poc = the picture's number, last_key = cfg+0x6e0 */
int kso_ref_scenecut(int pcost, int icost, int prev_icost, int blocks, int lg, int thr, int keyint, int poc, int last_key);
void kso_aq_ctu_map(const double *off, int nx, int ny, int base_qp, int lo, int hi, int8_t *map);
void kso_qoff_ctu_map(const double *off, int nx, int ny, int lg, int cols, int rows, int base_qp, int lo, int hi, int8_t *map);
double kso_ref_log2(uint32_t x);
int kso_ref_exp2fix8(double x);

/* calcFrameCost enc@0x4a7410: the lookahead's cost of coding picture `cur` from ref0 (d0 pictures back) and ref1 (d1 pictures ahead) on the half-size pictures, per 8x8 block:
 * list-0 / list-1 diamond search from the neighbours' vectors (meInitPoint + interMeDia), the bi-predictive average, seven intra modes + refinement; the picture sums and the
 * motion statistics the slice-type decision, cuTree and the rate control read.  Fields are the words the function reads / writes (L = TInputPic+0x50, the half-size layer). */
typedef struct {
    const uint8_t *cur, *ref0, *ref1; int stride;      /* sample (0, 0) of the three planes (L+0x28), one stride (L+4); the planes are readable margin_x / margin_y beyond the picture */
    int w, h, nx, ny, cnt;                             /* L+0, +8, +0xc, +0x10, +0x14 */
    int d0, d1, flag, slice_type;                      /* arguments 5, 6, 7; cur+0x20 (2: the motion statistics are not taken) */
    int merange, lg, zero_thr, fast_intra, scenecut, preset, p8, aq, b_intra, f3a8, f36c, f538, f3b4;   /* TEncParam +0x710 +0x3c0 +0x3a0 +0x3a4 +0x390 +0xc +0x8 +0x378 +0x388 +0x3a8 +0x36c +0x538 +0x3b4 */
    const uint16_t *lambda_tab;                        /* TEncParam+0x720: u16[52], the integer lambda of every QP (the lookahead searches with entry 12; the others only through table overruns) */
    int do_list[2];                                    /* the list's vectors are not there yet (first word of the vector plane = 0x7fff): search; else the stored vectors and costs are used */
    int intra_done;                                    /* L+0x18, in / out */
    uint16_t *intra; uint8_t *imode; const uint16_t *invq; uint16_t *inter; uint8_t *bits;   /* L+0x30, +0x38, +0x48, +0x50[idx], +0x2d8[idx] (idx = 9 d0 + d1) */
    int32_t *mv[2], *cost[2];                          /* L+0x560[d0 - 1] / +0x5a0[d1 - 1] (x = low 16 bits, quarter pel), L+0x5e0[d0 - 1] / +0x620[d1 - 1] */
    int32_t intra_wins, sum_intra, sum_intra_aq, sum, sum_aq, stats[4];   /* L+0x660[d0], +0x684[0], +0x7c8[0], +0x684[idx], +0x7c8[idx], L+0x90c+16 d0 (4 words); in / out */
    int ret;
    int margin_x, margin_y, oob, table_oob;            /* not reference fields: a block read beyond the margin sets oob; an mvd cost more than 1024 entries from the table's centre sets table_oob */
} kso_cfc;
void kso_ref_calc_frame_cost(kso_cfc *c);
/* the cuTree finish inlined in CInputPicManage::updateQueue (enc@0x480964..0x480a54): per block off = aq_off - 1.8 (log2(propagate + intra') - log2(intra')) with the reference's table log2, intra' = (intra x inv_qscale + 128) >> 8,
 * clipped to [-15, 20]; blocks with intra' = 0 keep what `out` holds.  dbl = the propagate cost counts twice (TEncParam+0x35c set and the picture's +0x68 = 0) */
void kso_ref_cutree_finish(int cnt, const uint16_t *intra, const uint16_t *inv_qscale, const uint16_t *propagate, const double *aq_off, int dbl, double *out);
#ifdef __cplusplus
}
#endif
#endif
