/* ks265_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the pixel kernels on the hot path of the KSC265 encoder
 * (ksvc/ks265codec v2.6.1.3, binary-only reference).  Every function keeps the exact
 * C signature of the entry it restates in the reference's operator tables (SURVEY.md
 * §2.3 / §8b "B3"), and cites the reference symbol as `enc@0xADDR name`
 * (= /root/reference/ubuntu_x64/appencoder, non-PIE, addresses from `nm -C -n`).
 *
 * Pinning: every function here is checked bit-for-bit against outputs of the reference
 * binary's own `_c` kernels (oracle/ref_probe/, fixtures in tests/golden/ (npz files),
 * test: tests/test_oracle_golden.py), and slotted behind the reference's own operator tables they leave its .265 byte-identical
 * (oracle/ref_probe/seam_harness.py, tests/test_seam.py).  Parity is therefore PINNED at kernel level.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#ifndef KS265_ORACLE_H
#define KS265_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1/a2/a4: distortion kernels (enc@0x47ae30 sad_c, 0x47ae90 sad4_c, 0x47b060 sad3_c,
 *      0x4cee30 sad4blk_8x8_c, 0x47b230.. sse_c<N>, 0x47b680 had_c) ---- */
uint32_t ks265o_sad(const uint8_t *a, const uint8_t *b, long sa, long sb, long h, long w);
void ks265o_sad4(const uint8_t *fenc, const uint8_t *ref, long sFenc, long sRef, long h, uint32_t out[4], long w);
void ks265o_sad3(const uint8_t *fenc, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2, long sFenc, long sRef,
                 long h, uint32_t out[3], long w);
void ks265o_sad4blk_8x8(const uint8_t *a, const uint8_t *b, long sa, long sb, uint32_t out[4]);
uint32_t ks265o_sse(const uint8_t *a, const uint8_t *b, int sa, int sb, int n);
uint32_t ks265o_had(const uint8_t *a, const uint8_t *b, long sa, long sb, long h, long w);

/* ---- a10: residual (enc@0x4345f0 H265_CalResidual<N>) ---- */
void ks265o_calc_residual(int16_t *res, const uint8_t *org, const uint8_t *pred, int strideOrg, int stridePred,
                          int strideRes, int n);

/* ---- a6: forward transform (enc@0x4c2250 H265_2dDst4x4_c, 0x4c2210/0x4c2290/0x4c22d0/0x4c2310 H265_2dDctNxN_c)
 *      idx: 0 = DST4, 1 = DCT4, 2 = DCT8, 3 = DCT16, 4 = DCT32 (order of g_H265_2dDct_Func enc@0x707ca0) ---- */
void ks265o_fwd_transform(int idx, const int16_t *src, int16_t *dst, int srcStride, int dstStride, int16_t *tmp);

/* ---- a7: quant (enc@0x4a9cf0 H265QuantBlock_c via H265Quant{4,8,16,32}_c), QuantParam (enc@0x4a9c90) ---- */
typedef struct {
    int scale;   /* +0x00 g_quantScales[qp%6]                       */
    int qbits;   /* +0x04 21 + qp/6 (before the -log2N of the call) */
    int offF;    /* +0x08 171 (I slice) / 85 (P,B)                  */
    int dq;      /* +0x0c g_invQuantScales[qp%6] << (qp/6)          */
    int minus1;  /* +0x10 -1                                        */
    int per;     /* +0x14 qp/6                                      */
} ks265o_quant_param;
void ks265o_get_base_quant_param(int qp, int sliceType /*2 = I*/, ks265o_quant_param *p);
int ks265o_quant(const int16_t *coef, int16_t *lvl, int stride, int scale, int off, int qbits, int16_t *deltaU, int n);

/* ---- a8: dequant (enc@0x439210 H265DeQuantBlock_c) ---- */
/* postQuant enc@0x4ace80 -> signBitHidingHDQ enc@0x4aa150 (sign-data hiding on the quantised levels) */
int ks265o_sign_bit_hiding(int16_t *lvl, const int16_t *coef, const int16_t *deltaU, int stride, int log2, int scan_idx);
void ks265o_dequant(const int16_t *lvl, int16_t *coef, int stride, int scale, int add, int shift, int lastX, int lastY);

/* ---- a9: inverse transform + pred add + clip (enc@0x448c40 H265_2dIDst4x4_c, 0x448f60.. H265_2dIDctNxN_c)
 *      idx as for the forward table (g_H265_2dIDct_Func enc@0x707060) ---- */
void ks265o_inv_transform(int idx, const int16_t *coef, uint8_t *dst, const uint8_t *pred, int coefStride, int dstStride,
                          int predStride, int16_t *tmp, int lastX, int lastY);

/* ---- a11: deblocking edge filters (enc@0x403630 EdgeFilterLumaVer_c, 0x4038c0 EdgeFilterLumaHor_c,
 *      0x403c50 PixelFilterChromaVer_c, 0x403d10 PixelFilterChromaHor_c), tables uiTCTable/uiBetaTable ---- */
void ks265o_edge_filter_luma_ver(uint8_t *pix, int stride, int beta, int tc, int length, int filterP, int filterQ);
void ks265o_edge_filter_luma_hor(uint8_t *pix, int stride, int beta, int tc, int length, int filterP, int filterQ);
void ks265o_pixel_filter_chroma_ver(uint8_t *pix, int stride, int tc, int length, int filterP, int filterQ);
void ks265o_pixel_filter_chroma_hor(uint8_t *pix, int stride, int tc, int length, int filterP, int filterQ);
extern const uint8_t ks265o_tc_table[54];
extern const uint8_t ks265o_beta_table[52];

/* ---- a5: fractional-sample interpolation (enc@0x40e4f0 interpLumaHor8to8_c ... 0x411850 interpChromaVer16to16_c) ---- */
void ks265o_interp_luma_hor_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_luma_ver_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_luma_hor_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_luma_ver_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_luma_ver_16to8(uint8_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_luma_ver_16to16(int16_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_chroma_hor_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_chroma_ver_8to8(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_chroma_hor_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_chroma_ver_8to16(int16_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_chroma_ver_16to8(uint8_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac);
void ks265o_interp_chroma_ver_16to16(int16_t *dst, int dstStride, const int16_t *src, int srcStride, int w, int h, int frac);

/* ---- a12: SAO (enc@0x43e4e0 SaoApplyOffsetBo_c, 0x43e650.. SaoApplyOffsetEo{0..3}_c, 0x4ae9c0 statSaoBoEo01_c) ---- */
void ks265o_sao_apply_bo(const int8_t *offsets, uint8_t *rec, int stride, int height, int width, int bandPosition);
void ks265o_sao_apply_eo(int cls, const int8_t *offsets, uint8_t *rec, int stride, int height, int width);
void ks265o_stat_sao_bo_eo01(int *eoJoint, int *bo, const uint8_t *org, const uint8_t *rec, int recStride, int orgStride,
                             int width, int height, int rowStep);

/* ---- bi-prediction helpers, pinned now for the B-picture row (enc@0x435160 DefaultWeightedBi_c, enc@0x47b1a0 calcBiMeOrg_c) ---- */
void ks265o_default_weighted_bi(uint8_t *dst, const int16_t *p0, const int16_t *p1, int dstStride, int srcStride, int width, int height);
void ks265o_sao_est_iter_offset(int lambda_q8, int rate_base, int32_t *offset, int count, int diff_sum, int32_t *best_cost);
int32_t ks265o_sao_bo_type_estimation(int lambda_q8, int32_t *count, int32_t *sum, int32_t *band, int32_t *offsets);
int32_t ks265o_sao_eo_type_estimation(int lambda_q8, const int32_t *count, int32_t *sum, int32_t *offsets);
void ks265o_sao_mode_decision(const int32_t *stats, int lam_y, int lam_c, int left_avail, int up_avail, const int8_t *left, const int8_t *up, int mask_y, int mask_uv, int8_t *out, int32_t *best);
int ks265o_calc_bs(const int32_t *p /*3 words*/, const int32_t *q, int tu_edge, int is_b);
void ks265o_est_bit_rdoq(int32_t *out /*180 words*/, int log2, int luma, const uint8_t *ctx, const int32_t *entropy /*128*/);
uint32_t ks265o_inter_me_bi_full(int32_t *best, const uint8_t *org, const uint8_t *ref, int orgStride, int refStride, const uint16_t *mvcost, int h, int log2w, int use_had);
void ks265o_explicit_weighted_p(uint8_t *dst, const int16_t *p0, int dstStride, int srcStride, int width, int height, const int32_t *wp);
void ks265o_explicit_weighted_bi(uint8_t *dst, const int16_t *p0, const int16_t *p1, int dstStride, int srcStride, int width, int height, const int32_t *wp);
uint32_t ks265o_calc_bi_me_org(uint8_t *dst, const uint8_t *pred, const uint8_t *org, int stride, int height, int width);

/* ---- intra prediction (ks265_intra_oracle.c): g_IntraPredFunction enc@0x7070a0 family, IntraPredFilterRef_c enc@0x424110.
 * `ref` / `src` / `dst` point at the corner sample of a linear reference array: [1 + x] = top, [-1 - y] = left. */
void ks265o_intra_pred(uint8_t *dst, int stride, const uint8_t *ref, int mode, int log2, int edge_filter);
void ks265o_intra_filter_ref(const uint8_t *src, uint8_t *dst, int size, int strong_enabled);

/* ---- lookahead leaf kernels: downsample_c enc@0x4a6a60, weightBi_sad_c enc@0x4a7170, acEnergyPlane_c enc@0x4650e0 */
void ks265o_downsample(uint8_t *dst, const uint8_t *src, int dstStride, int srcStride, int w, int h);
uint32_t ks265o_weight_bi_sad(const uint8_t *org, unsigned orgStride, const uint8_t *ref0, const uint8_t *ref1, unsigned stride0, unsigned stride1, int w, int h);
uint32_t ks265o_ac_energy_plane(const uint8_t *src, int stride, int log2);

#ifdef __cplusplus
}
#endif
#endif
