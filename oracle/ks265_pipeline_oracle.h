/* ks265_pipeline_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the whole-frame stages of include/ks265_hip.h §3, composed ONLY from the pinned
 * kernel restatements of ks265_oracle.h (each pinned against the reference binary).  The sequencing
 * follows the reference's per-CTU pipeline (motionSearchOneRef enc@0x483f40 ->
 * interMeDia enc@0x48fbe0 -> subMeSquare enc@0x4b5660 -> reconstruct enc@0x481da0 ->
 * CLoopFilterCtu::Execute enc@0x49dd30) restructured frame-wide (SURVEY.md §7.3 "granularity
 * inversion"): the CU decisions are the build's own because the reference's RDO is closed code
 * (SURVEY.md §7.1), so stage-level parity is GPU == this file, bit for bit, on the same inputs.
 *
 * Struct layouts are identical to include/ks265_hip.h so the same numpy dtypes serve both.
 */
#ifndef KS265_PIPELINE_ORACLE_H
#define KS265_PIPELINE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t width, height, qp, lambda_q4, me_range, me_method, subme, deblock, sao, beta_offset_div2, tc_offset_div2, bframes, refs, me_hex_thr, sdh, pre_search, merge, bi_refine, decimate, rdo, intra_inter, propagate,
            sub_satd, sub_thr, sub_flat, sub_cap, sub_cap_step, sub_diag_fast,     /* the sub-pel refinement's knobs (Stage B, ks265_pipeline_oracle.c) */
            part,                                                            /* -part 1: 2NxN / Nx2N partitions of 64 / 32 / 16 CUs in P / B pictures (kso_cu_decide_part[_b]) */
            tu_inter,                                                        /* -intertu 1 (tuInter: veryslow, placebo): a 2Nx2N inter CU of 32 / 16 samples may carry four transform units (split_transform_flag) */
            skip_rd;                                                         /* stage D2 after the reconstruction: CUs whose merge candidate without residual is the cheaper coding drop their residual (kso_skip_pass); 1 = B pictures, 2 = P pictures too */
} kso_frame_cfg;

typedef struct {
    int32_t pad_y, pad_c, stride_y, stride_c, rows_y, rows_c;
    int64_t bytes_y, bytes_c;
    int32_t ctu_cols, ctu_rows, pu_per_ctu;
    int64_t bytes_pu, bytes_cu8, bytes_sao;
} kso_frame_geom;

typedef struct { int16_t mvx, mvy, mvpx, mvpy; uint32_t cost, dist; } kso_pu;
typedef struct { int16_t mvx, mvy, mv1x, mv1y; uint8_t log2_cu, cbf, pred_mode, inter_dir; } kso_cu8;   /* pred_mode: 0 inter, 1 flat intra, 2 intra (mvx = luma mode); inter_dir: 1 = L0, 2 = L1, 3 = Bi */
typedef struct { int16_t mvx, mvy, mv1x, mv1y; uint32_t cost; uint32_t inter_dir; } kso_pu_b;
typedef struct { int8_t type, band, offset[4], rsv[2]; } kso_sao_param;
typedef struct { uint8_t *y, *u, *v; } kso_pic;

int kso_frame_geometry(const kso_frame_cfg *cfg, kso_frame_geom *g);
void kso_pad_picture(const kso_frame_cfg *cfg, kso_pic pic);
void kso_load_i420(const kso_frame_cfg *cfg, const uint8_t *i420, kso_pic dst);
void kso_store_i420(const kso_frame_cfg *cfg, kso_pic src, uint8_t *i420);
void kso_ref_planes(const kso_frame_cfg *cfg, kso_pic ref, uint8_t *planes);
void kso_presearch(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, int16_t *field /* ceil(W/16) x ceil(H/16) x {mvx, mvy}, integer pel */,
                   int16_t *ctu_off /* ctu_cols x ctu_rows x {ox, oy}: window offset of every CTU */);
void kso_ctu_mv_limits(const kso_frame_cfg *cfg, int cx, int cy, int ox, int oy, int lim[4]);
void kso_me_integer(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const kso_pu *prev_pu, kso_pu *pu);
void kso_me_integer_ex(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const kso_pu *prev_pu, kso_pu *pu, int16_t *off_out /* 2 per CTU, may be NULL */);
/* cfg->propagate: one round of vector propagation between neighbouring PUs of the same size (stage A2); in != out */
void kso_me_propagate(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const int16_t *ctu_off, const kso_pu *in, kso_pu *out);
void kso_me_subpel(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes, kso_pu *pu);
void kso_cu_decide(const kso_frame_cfg *cfg, const kso_pu *pu, kso_cu8 *cu8);
void kso_cu_flat_intra(const kso_frame_cfg *cfg, kso_cu8 *cu8);
void kso_cu_decide_part(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes, const kso_pu *pu, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8);
/* multi-reference B pictures: the lists' pictures of the B picture being coded (kso_set_mref(NULL) ends it); kso_ref_pick: per PU the cheapest picture of one list */
typedef struct { int n0, n1; const uint8_t *planes0[4], *planes1[4]; kso_pic pic0[4], pic1[4]; const uint8_t *idx0, *idx1; } kso_mref;
void kso_set_mref(const kso_mref *m);
void kso_ref_pick(const kso_frame_cfg *cfg, int nref, const kso_pu *const *pu, kso_pu *out, uint8_t *idx);
void kso_cu_decide_part_b(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0, const uint8_t *planes1, const kso_pu *pu0, const kso_pu *pu1, const kso_pu_b *pub,
                          const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8);
/* cfg->intra_inter: the CU trees with intra candidates (icost / imode: 85 per CTU from kso_intra_candidates; NULL = none), and the intra CUs' reconstruction pass */
void kso_cu_decide_ii(const kso_frame_cfg *cfg, const kso_pu *pu, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8);
void kso_cu_decide_b_ii(const kso_frame_cfg *cfg, const kso_pu_b *pub, const uint32_t *icost, const uint8_t *imode, kso_cu8 *cu8);
void kso_intra_candidates(const kso_frame_cfg *cfg, kso_pic src, const void *pu_records, uint32_t *cost_out, uint8_t *mode_out);
void kso_intra_inter_reconstruct(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon);
void kso_reconstruct(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref, const uint8_t *planes, kso_pic ref1, const uint8_t *planes1,
                     kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon);
/* B pictures: per-PU choice among L0, L1 and the bi-predictive average (interMeBi* lineage), then the CU quadtree on it */
void kso_bi_decide(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0, const uint8_t *planes1, const kso_pu *pu0, const kso_pu *pu1,
                   kso_pu_b *pub);
/* cfg->bi_refine == 2: the joint refinement after the CU decision, for the 2N x 2N inter CUs it chose (in front of the merge pass) */
void kso_bi_refine_chosen(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0, const uint8_t *planes1, const kso_pu *pu0, const kso_pu *pu1,
                          kso_pu_b *pub, kso_cu8 *cu8);
void kso_cu_decide_b(const kso_frame_cfg *cfg, const kso_pu_b *pub, kso_cu8 *cu8);
/* stage C2 (cfg->merge): every CU may adopt the motion of one of its spatial merge neighbours (or the zero vector); pu for P pictures, pub for B pictures (the other NULL) */
void kso_merge_pass(const kso_frame_cfg *cfg, kso_pic src, const uint8_t *planes0, const uint8_t *planes1, const kso_pu *pu, const kso_pu_b *pub,
                    const kso_cu8 *cu_in, kso_cu8 *cu_out);
/* stage D2 (cfg->skip_rd): after kso_reconstruct* - every inter CU with residual may become a CU without, carrying a merge candidate's motion; cu_in (the map with its coded-block
 * flags) -> cu_out; levels and reconstruction updated in place; ref1.y = NULL for P pictures;
 * several pictures per list: inside kso_set_mref */
void kso_skip_pass(const kso_frame_cfg *cfg, kso_pic src, kso_pic ref0, kso_pic ref1, const kso_cu8 *cu_in, kso_cu8 *cu_out, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon);
void kso_experiment_skip(const int *v /*[8]*/);
/* intra pictures: mode pre-selection on source neighbours + CU quadtree, then the sequential reconstruction.
 * Intra CU in cu8: pred_mode = 2, mvx = luma mode (0 planar, 1 DC, 2..34 angular), chroma = the luma mode (DM). */
void kso_intra_decide(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8);
void kso_intra_decide_ex(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, uint32_t *cost_out /* nctu x 85, PU indexing */);
/* lookahead frame cost on low-resolution pictures: out = { sum intra, sum inter, sum min, blocks | intra-cheaper blocks << 32 } over the 8x8 blocks */
void kso_lookahead_reduce(const kso_frame_cfg *cfg, const uint32_t *intra_cost, const kso_pu *pu, uint64_t out[4]);
void kso_intra_reconstruct(const kso_frame_cfg *cfg, kso_pic src, kso_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, kso_pic recon);
/* multi-reference P pictures: the per-PU choice among nref list-0 pictures (inter_dir = 1 | idx << 4), and the reconstruction from them */
void kso_ref_decide(const kso_frame_cfg *cfg, int nref, const kso_pu *const *pu, kso_pu_b *pub);
void kso_reconstruct_mref(const kso_frame_cfg *cfg, kso_pic src, int nref, const kso_pic *refs, const uint8_t *const *planes, kso_cu8 *cu8, int16_t *lvl_y,
                          int16_t *lvl_u, int16_t *lvl_v, kso_pic recon);
void kso_deblock(const kso_frame_cfg *cfg, const kso_cu8 *cu8, kso_pic recon);
void kso_sao(const kso_frame_cfg *cfg, kso_pic src, kso_pic deblocked, kso_sao_param *sao, kso_pic dst);

/* QP per CTU (cu_qp_delta, quantisation group = CTU): see the comment in ks265_pipeline_oracle.c */
void kso_set_qp_map(const int8_t *map);
void kso_effective_qp(const kso_frame_cfg *cfg, const kso_cu8 *cu8, uint8_t *eff);
#ifdef __cplusplus
}
#endif
#endif
