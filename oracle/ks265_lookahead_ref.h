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
double kso_ref_log2(uint32_t x);
int kso_ref_exp2fix8(double x);
#ifdef __cplusplus
}
#endif
#endif
