/* ks265_rdoq_ref.h - TEST INFRASTRUCTURE (oracle): the reference's rate-distortion optimised quantisation, see ks265_rdoq_ref.c */
#ifndef KS265_RDOQ_REF_H
#define KS265_RDOQ_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* rdoQuant enc@0x4aac50.  lvl: N x N quantised levels (in: H265QuantBlock's output, out: the decision, signs restored); coef: N x N transform coefficients;
 * dq / per: QuantParam+0xc / +0x14; lam / lam_sdh: the two integer lambdas ((int64)(cfg multiplier x table[qp] + 0.5): cfg+0x43c / +0x434 for luma, +0x438 / +0x430 for chroma);
 * T: the 180-word bit table of estBitRdoq enc@0x46a8a0 (ks265o_est_bit_rdoq); tu5: TTransUnit+5; last_pos: TTransUnit+0x40[comp] (scan position of the last level the
 * quantiser left); sigmask: TTransUnit+0x68[comp][sub-block scan index], bit 15 - position = level non-zero (updated); flag_a4c0: TCtuInfo+0xa4c0; sdh: cfg+0x3e0.
 * Returns the number of non-zero levels; *out_last = the new last scan position (-1: none). */
int kso_ref_rdo_quant(int16_t *lvl, const int16_t *coef, int log2, int scan_idx, int comp, int dq, int per, int64_t lam, int64_t lam_sdh, const int32_t *T,
                      int tu5, int last_pos, uint16_t *sigmask, int flag_a4c0, int sdh, int32_t *out_last, uint64_t *out_cgmask);
/* the inputs rdoQuant takes from the quantiser: significance masks per sub-block, returns the last scan position */
int kso_rdoq_scan_flags(const int16_t *lvl, int log2, int scan_idx, uint16_t *sigmask);
#ifdef __cplusplus
}
#endif
#endif
