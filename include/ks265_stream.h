/* ks265_stream.h — host side of the pixel path: from the records the HIP stages leave (CU map, levels, SAO parameters) to a
 * conforming HEVC (H.265 Main profile) Annex-B bitstream.  Plain C, CPU only, no allocation inside the writers.
 *
 * This is the SURVEY.md §8(f) rank 4 row ("bitstream / wire formats") and the part of §8(b) B2 that the reference keeps on the CPU
 * (CEncOutputBs::onFrameEncoded enc@0x4a03f0, write_ParamSet<SPS> enc@0x4a52b0, write_slice_segment_header enc@0x4b14d0,
 * CCtuSbac::processCtuSbac enc@0x475880; `enc@` = symbol in /root/reference/ubuntu_x64/appencoder).  The reference's encoder
 * decisions are closed code, so the stream is not byte-identical to appencoder's; the contract is conformance: the reference's own
 * decoder (ubuntu_x64/appdecoder) must decode it to exactly the reconstruction the HIP path produced (tests/test_stream.py).
 *
 * Tool set written (what the pixel path produces): 64x64 CTBs, CUs 64..8, 2Nx2N partitions, TU = CU up to 32x32 (64x64 CUs carry four
 * 32x32 transform units), intra 35 modes with DM chroma, inter uni- and bi-prediction with explicit (AMVP) vectors or, where the chosen motion equals a merge candidate, merge / skip signalling,
 * no temporal MVP, sign-data hiding as configured, no transform skip, one slice per picture, flat quantisation, deblocking and SAO as signalled.
 */
#ifndef KS265_STREAM_H
#define KS265_STREAM_H
#include <stddef.h>
#include <stdint.h>
#include "ks265_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t width, height;                 /* luma, multiples of 8 (= MinCbSizeY)                                                   */
    int32_t fps_num, fps_den;              /* only informative (no VUI is written unless > 0)                                       */
    int32_t sao, deblock;                  /* sample_adaptive_offset_enabled_flag / !pps_deblocking_filter_disabled_flag            */
    int32_t beta_offset_div2, tc_offset_div2;
    int32_t max_dec_pic_buffering;         /* pictures the decoder must hold (references + current), >= 1                           */
    int32_t max_num_reorder;               /* pictures that may precede a picture in decoding order and follow it in output order   */
    int32_t log2_max_poc_lsb;              /* 4..16                                                                                 */
    int32_t sdh;                           /* sign_data_hiding_enabled_flag: the levels were produced with ks265_frame_cfg.sdh = 1                  */
    int32_t wpp;                           /* entropy_coding_sync_enabled_flag: every CTU row is a substream with an entry point (the reference's WPP) */
    int32_t list_mod;                      /* lists_modification_present_flag: l0_poc / l1_poc may name the used pictures of the RPS in any order (7.3.6.2)  */
    int32_t cu_qp_delta;                   /* cu_qp_delta_enabled_flag with diff_cu_qp_delta_depth = 0 (quantisation group = CTU): slices carry a QP per CTU (qp_map)     */
    int32_t tu_inter;                      /* max_transform_hierarchy_depth_inter = 1 (-intertu 1): inter CUs of 32 / 16 / 8 samples code split_transform_flag; ks265_cu8.log2_cu
                                              bits 4..5 = 3 marks a 2Nx2N CU with four transform units (ks265_frame_cfg.tu_inter)                                          */
} ks265_stream_cfg;

enum { KS265_SLICE_B = 0, KS265_SLICE_P = 1, KS265_SLICE_I = 2 };
enum { KS265_NAL_TRAIL_N = 0, KS265_NAL_TRAIL_R = 1, KS265_NAL_IDR_W_RADL = 19, KS265_NAL_IDR_N_LP = 20, KS265_NAL_CRA = 21,
       KS265_NAL_VPS = 32, KS265_NAL_SPS = 33, KS265_NAL_PPS = 34 };

/* one picture = one slice segment */
typedef struct {
    int32_t nal_type;                      /* KS265_NAL_*                                                                            */
    int32_t slice_type;                    /* KS265_SLICE_*                                                                          */
    int32_t poc;                           /* picture order count of this picture (0 for IDR)                                        */
    int32_t qp;                            /* SliceQpY                                                                               */
    /* reference picture set: every picture that stays in the DPB after this one is decoded, as POCs; used[i] = 1 if THIS picture
     * predicts from it.  Entries may come in any order. */
    int32_t num_rps;
    int32_t rps_poc[16];
    uint8_t rps_used[16];
    /* the reference lists; without cfg.list_mod they must come in the order the default construction of H.265 8.3.4 yields them (the
     * writer checks this and refuses anything else), with it any order of used pictures is signalled through ref_pic_lists_modification();
     * cu8.inter_dir >> 4 indexes list 0 for P pictures (multi-reference search) */
    int32_t num_l0, num_l1;
    int32_t l0_poc[4], l1_poc[4];
    /* records of the HIP stages (HOST copies): CU map (W/8 x H/8), levels (W x H, W/2 x H/2 x 2, TU in place), SAO (3 per CTU) */
    const ks265_cu8 *cu8;
    const int16_t *lvl[3];
    const ks265_sao_param *sao;            /* NULL = SAO off for this picture                                                        */
    const int8_t *qp_map;                  /* cfg.cu_qp_delta: the QP each CTU's residual was quantised with (raster order); NULL = the slice QP everywhere.
                                            * cu_qp_delta goes out with the first coded residual of a CTU, predicted from the previous CTU of the row (8.6.1)  */
} ks265_slice_in;

/* Parameter sets as Annex-B NAL units (start code 00 00 00 01 included).  Return the number of bytes written, < 0 on error
 * (KS265_* codes of ks265_hip.h: KS265_POINTER, KS265_NOTSUPPORTED = buffer too small or value out of range). */
long ks265_write_vps(const ks265_stream_cfg *cfg, uint8_t *out, size_t cap);
long ks265_write_sps(const ks265_stream_cfg *cfg, uint8_t *out, size_t cap);
long ks265_write_pps(const ks265_stream_cfg *cfg, uint8_t *out, size_t cap);
/* One coded picture (slice segment header + CABAC slice data) as one Annex-B NAL unit.  Thread-safe (no shared state): pictures of a
 * GOP can be written concurrently by different host threads, each into its own buffer.  `scratch` must hold ks265_slice_scratch_bytes()
 * bytes. */
size_t ks265_slice_scratch_bytes(const ks265_stream_cfg *cfg);
long ks265_write_slice(const ks265_stream_cfg *cfg, const ks265_slice_in *in, void *scratch, uint8_t *out, size_t cap);
/* the entropy coder's context states at the end of the slice `scratch` wrote last (pStateIdx << 1 | valMps; with substreams: those saved after the second CTU of the last CTU
 * row) and where the residual-coding groups start (layout[10]: cbf_luma, cbf_chroma, coded_sub_block_flag, sig_coeff_flag, last x, last y, greater1, greater2, rqt_root_cbf,
 * count) - what a host snapshots into the bit tables of rdoQuant (estBitRdoq enc@0x46a8a0) for the pictures that follow.  Returns the number of states. */
int ks265_slice_final_contexts(const ks265_stream_cfg *cfg, const void *scratch, uint8_t *out, int cap, int *layout);
/* -rdoq 1 (round 6): the eight bit tables [4 sizes][luma, chroma][180] rdoQuant enc@0x4aac50 prices with (estBitRdoq enc@0x46a8a0), from the context states a slice ended with
 * (states = ks265_slice_final_contexts' output: the tables follow the stream) or, states = NULL, from the initial states of a slice of slice_type at qp */
int ks265_rdoq_tables(const ks265_stream_cfg *cfg, const uint8_t *states, int slice_type, int qp, int32_t *tables);

/* cfg->wpp = 1: the same picture row by row, so that SEVERAL host threads can write one picture (the reference's WPP tasks, qy265executeEncCtuTaskWpp
 * enc@0x475d20).  ks265_wpp_begin prepares the job in `mem` (ks265_wpp_bytes), ks265_wpp_code_row codes one CTU row into its substream - thread-safe for
 * different rows; rows must be handed out in ascending order, a row waits while the row above is less than two CTUs ahead - and ks265_wpp_finish, after all
 * rows, assembles slice header (entry points), substreams and emulation prevention into the NAL unit.  ks265_write_slice does exactly this on one thread. */
size_t ks265_wpp_bytes(const ks265_stream_cfg *cfg);
int ks265_wpp_begin(const ks265_stream_cfg *cfg, const ks265_slice_in *in, void *mem);
int ks265_wpp_rows(const void *mem);
int ks265_wpp_code_row(void *mem, int row);
long ks265_wpp_finish(void *mem, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
