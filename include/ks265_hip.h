/* ks265_hip.h — C ABI of the MI355X-native HEVC encode pixel-kernel path (libks265hip.so).
 *
 * Drop-in boundary for the data-parallel hot path of the KSC265 encoder (ksvc/ks265codec v2.6.1.3,
 * binary-only).  The reference reaches this path ONLY through writable function-pointer tables
 * (SURVEY.md §8b "B3": g_sad_Function enc@0x707c60, g_sad4_Function enc@0x707be0, g_sad3_Function
 * enc@0x707b60, g_sse_Function enc@0x707ba0, g_had_Function enc@0x707b40, g_H265_2dDct_Func enc@0x707ca0,
 * g_QuantFuncs enc@0x707ce0, g_DeQuantFuncs enc@0x707990, g_H265_2dIDct_Func enc@0x707060,
 * g_calc_residual_funcs enc@0x706fe0, g_EdgeFilterLuma{Ver,Hor}Func, g_PixelFilterChroma{Ver,Hor}Func,
 * g_fSaoApplyOffset*, g_statBoEo01_funcs, g_interp{Luma,Chroma}*_func), called per block from inside
 * CCtuEnc::processOneCtu enc@0x46f320.  A GPU cannot be fed one 8x8 block per call, so every table entry
 * is exposed here in BATCHED form (section 2: same arithmetic, arrays of block descriptors instead of one
 * pointer pair), and the per-CTU sequencing of those tables is exposed as whole-frame
 * stages (section 3).  `enc@0xADDR` = symbol in /root/reference/ubuntu_x64/appencoder.
 *
 * Conventions: plain C, no torch types.  Every `dev` pointer is DEVICE memory (HBM) owned by the caller;
 * descriptor arrays are device memory too.  All calls are asynchronous on the context's HIP stream and
 * return 0 or a negative ks265_status; no call allocates after ks265_create/ks265_frame_create.
 * All arithmetic is integer and bit-exact with the reference kernels (tests/golden/).
 */
#ifndef KS265_HIP_H
#define KS265_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ 1. context */

typedef struct ks265_ctx ks265_ctx;

/* error codes follow the reference's qy265def.h:7-22 spirit (QY_OK = 0, failures negative) */
enum ks265_status {
    KS265_OK = 0,
    KS265_FAIL = -1,          /* HIP runtime error (see ks265_last_error)            */
    KS265_OUTOFMEMORY = -2,   /* QY_OUTOFMEMORY                                       */
    KS265_POINTER = -3,       /* QY_POINTER: NULL argument                            */
    KS265_NOTSUPPORTED = -4,  /* QY_NOTSUPPORTED: size / mode outside the HEVC tool set */
    KS265_NO_DEVICE = -5      /* no gfx950 device visible: the product never falls back to a CPU path */
};

int ks265_create(ks265_ctx **out, int device);      /* creates its own HIP stream                      */
/* the same with the device's highest stream priority when high_priority != 0 (a host's long, narrow kernels - the key picture's intra wavefront - underneath wide ones) */
int ks265_create_prio(ks265_ctx **out, int device, int high_priority);
void ks265_destroy(ks265_ctx *ctx);
int ks265_set_stream(ks265_ctx *ctx, void *hip_stream); /* adopt a caller stream (e.g. torch's current) */
/* waits for the context's stream; also reads (and clears) the device-side error word that kernels set when they could not complete
 * correctly (today: the intra wavefront's bounded wait timing out) -> KS265_FAIL with ks265_last_error() naming the condition; the
 * pictures encoded since the previous ks265_synchronize must then be re-encoded */
int ks265_synchronize(ks265_ctx *ctx);
/* reads and clears the same error word WITHOUT waiting for the stream: a pipelined host calls it once a picture's own completion event has fired (the word is
 * per context = per stream; kernels of later pictures may already have run, so a set bit means "this picture or a later one") -> KS265_OK / KS265_FAIL */
int ks265_take_device_error(ks265_ctx *ctx);
/* test hook: KS265_DBG_WAVEFRONT_SPINS = the number of polls a CTU row of the intra wavefront waits for the row above (value < 0
 * restores the default, about one second); 0 forces the timeout path */
enum { KS265_DBG_WAVEFRONT_SPINS = 1 };
int ks265_debug_set(ks265_ctx *ctx, int what, int value);
const char *ks265_last_error(ks265_ctx *ctx);
const char *ks265_version(void);                    /* cf. strLibQy265Version, qy265enc.h              */
/* Memory and stream-ordered copies for hosts written in plain C (the reference's host is C/C++ without any GPU runtime; a host above this
 * ABI needs no HIP header): device memory, pinned host memory, copies enqueued on the context's stream, and events to learn from another
 * host thread that everything enqueued before the record has finished. */
#include <stddef.h>
int ks265_dev_malloc(ks265_ctx *, void **dev, size_t bytes);
int ks265_dev_free(ks265_ctx *, void *dev);
int ks265_host_malloc(ks265_ctx *, void **host, size_t bytes);
int ks265_host_free(ks265_ctx *, void *host);
/* memory of the application pinned and mapped IN PLACE (hipHostRegister) so that ks265_memcpy_h2d_async can read it by DMA: the encoder host uploads the caller's picture planes
 * without a copy into pinned memory of its own.  A range is registered once (a second registration of overlapping memory fails); failure is not an error of the context */
int ks265_host_register(ks265_ctx *, void *host, size_t bytes);
int ks265_host_unregister(ks265_ctx *, void *host);
int ks265_memcpy_h2d_async(ks265_ctx *, void *dev, const void *host, size_t bytes);
int ks265_memcpy_h2d_sync(ks265_ctx *, void *dev, const void *host, size_t bytes);     /* on no stream of the library's; returns when the data is on the device */
int ks265_memcpy_d2h_async(ks265_ctx *, void *host, const void *dev, size_t bytes);
int ks265_memcpy_d2d_async(ks265_ctx *, void *dev_dst, const void *dev_src, size_t bytes);
/* device -> pinned host memory of ks265_host_malloc as a kernel of 32 work-groups on the context's stream (stores straight over PCIe): unlike a runtime
 * copy that may run as a machine-filling shader, it leaves the compute units to whatever the other streams launch */
int ks265_copy_out_async(ks265_ctx *, void *pinned_host, const void *dev, size_t bytes);
int ks265_memset_async(ks265_ctx *, void *dev, int value, size_t bytes);
int ks265_event_create(ks265_ctx *, void **ev);
int ks265_event_record(ks265_ctx *, void *ev);
int ks265_event_wait(ks265_ctx *, void *ev);
int ks265_event_query(ks265_ctx *, void *ev, int *done);          /* *done = 1 when everything in front of the event's last record has run; never blocks */
/* everything enqueued on this context's stream after the call waits for the event (recorded on ANOTHER context's stream of the same device):
 * copy-in / compute / copy-out streams of a pipelined host hand pictures over without blocking a host thread */
int ks265_stream_wait_event(ks265_ctx *, void *ev);
int ks265_event_destroy(ks265_ctx *, void *ev);
/* Launch sequences as graphs: between ks265_capture_begin and ks265_capture_end everything enqueued on this context's stream by THIS thread (kernel launches,
 * async memsets of the stage functions) is recorded instead of run (hipStreamBeginCapture, relaxed mode: other threads may go on using the runtime);
 * ks265_capture_end returns an executable graph, ks265_graph_launch enqueues all of it with one runtime call.  A host whose pictures repeat the same
 * launch sequence with the same device pointers (IPPP: a handful of buffer rotations) replays it instead of issuing ~20 launches per picture.  Nothing that
 * synchronises or allocates may run inside a capture; event hand-overs to other streams stay outside. */
int ks265_capture_begin(ks265_ctx *);
int ks265_capture_end(ks265_ctx *, void **graph_exec);
int ks265_graph_launch(ks265_ctx *, void *graph_exec);
int ks265_graph_destroy(ks265_ctx *, void *graph_exec);
/* HIP-event timing on the context's stream (bench.py roofline leg) */
/* a one-thread kernel named ks265_marker_kernel on the context's stream: brackets a region of interest in a kernel trace (profiling aid) */
int ks265_marker(ks265_ctx *, int id);
int ks265_timer_start(ks265_ctx *ctx);
int ks265_timer_stop_ms(ks265_ctx *ctx, float *ms);

/* ------------------------------------------------------------------ 2. batched operator tables (B3) */

/* one block of a distortion batch: byte offsets from the two plane base pointers (may be negative
 * relative to an interior origin); replaces the (u8* a, u8* b) pointer pair of sad_c enc@0x47ae30 */
typedef struct { int32_t a_off, b_off; int32_t w, h; } ks265_blk;
/* one source block against three arbitrary reference positions (sad3_c enc@0x47b060) */
typedef struct { int32_t a_off, b_off[3]; int32_t w, h; } ks265_blk3;

/* g_sad_Function / g_sad_Function_A: out[i] = SAD */
int ks265_sad_batch(ks265_ctx *, const uint8_t *dev_a, int sa, const uint8_t *dev_b, int sb,
                    const ks265_blk *dev_blks, int n, uint32_t *dev_out);
/* g_sad4_Function: out[4i..] = {SAD(b-sb), SAD(b+sb), SAD(b-1), SAD(b+1)} << 4 (sad4_c enc@0x47ae90) */
int ks265_sad4_batch(ks265_ctx *, const uint8_t *dev_fenc, int sFenc, const uint8_t *dev_ref, int sRef,
                     const ks265_blk *dev_blks, int n, uint32_t *dev_out);
/* g_sad3_Function: out[3i..] plain SADs */
int ks265_sad3_batch(ks265_ctx *, const uint8_t *dev_fenc, int sFenc, const uint8_t *dev_ref, int sRef,
                     const ks265_blk3 *dev_blks, int n, uint32_t *dev_out);
/* g_sad4blk_8x8_func (sad4blk_8x8_c enc@0x4cee30): four 8x8 quadrants of 16x16 blocks; w,h ignored */
int ks265_sad4blk_8x8_batch(ks265_ctx *, const uint8_t *dev_a, int sa, const uint8_t *dev_b, int sb,
                            const ks265_blk *dev_blks, int n, uint32_t *dev_out);
/* g_sse_Function (sse_c<N> enc@0x47b230..): w == h in {4,8,16,32,64} */
int ks265_sse_batch(ks265_ctx *, const uint8_t *dev_a, int sa, const uint8_t *dev_b, int sb,
                    const ks265_blk *dev_blks, int n, uint32_t *dev_out);
/* g_had_Function (had_c enc@0x47b680): 8x8 tiles (+2)>>2 | 4x4 tiles (+1)>>1 | 2x2 raw */
int ks265_had_batch(ks265_ctx *, const uint8_t *dev_a, int sa, const uint8_t *dev_b, int sb,
                    const ks265_blk *dev_blks, int n, uint32_t *dev_out);

/* g_calc_residual_funcs: res (packed NxN s16 per block) = org - pred */
int ks265_residual_batch(ks265_ctx *, const uint8_t *dev_org, int so, const uint8_t *dev_pred, int sp,
                         const ks265_blk *dev_blks, int n, int16_t *dev_res);

/* g_H265_2dDct_Func[idx], idx 0 = DST4, 1 = DCT4, 2 = DCT8, 3 = DCT16, 4 = DCT32.
 * src/dst: nblk packed NxN s16 blocks (the reference passes strides; batches are packed). */
int ks265_fwd_transform_batch(ks265_ctx *, int idx, const int16_t *dev_src, int16_t *dev_dst, int nblk);
/* g_QuantFuncs[log2N-2] (H265QuantBlock_c enc@0x4a9cf0): per-batch scalar parameters as in the reference
 * call; dev_nz[i] = number of non-zero levels of block i (the reference's return value). */
int ks265_quant_batch(ks265_ctx *, int n, const int16_t *dev_coef, int16_t *dev_lvl, int16_t *dev_deltaU,
                      int32_t *dev_nz, int scale, int off, int qbits, int nblk);
/* postQuant enc@0x4ace80 -> signBitHidingHDQ enc@0x4aa150 (the step between g_QuantFuncs and g_DeQuantFuncs when sign-data hiding is on): per 4x4
 * coefficient group whose first and last level are more than 3 scan positions apart the parity of the level sum is made to carry the
 * first level's sign; packed N x N blocks, levels in place; scan_idx 0 diagonal, 1 horizontal, 2 vertical (N <= 8 only, as in H.265).
 * Blocks with fewer than two levels are left alone (postQuant does not call the function for them). */
int ks265_sign_hiding_batch(ks265_ctx *, int n, int scan_idx, int16_t *dev_lvl, const int16_t *dev_coef, const int16_t *dev_deltaU, int nblk);
/* rdoQuant enc@0x4aac50 (EncQuant.cpp; `rdoq`, qy265enc.h:129: on from -preset fast upward) - the reference's rate-distortion optimised
 * quantisation of one transform block, batched: one wave per block.  Per block: levels (in: H265QuantBlock's output rounded at 1/2, out: the decision with signs) and
 * transform coefficients, N x N packed at `off`; dq / per = QuantParam+0xc / +0x14; lam / lam_sdh = the two integer lambdas ((int64)(weight x 0.85 x 2^((qp - 12) / 3) + 0.5),
 * weights -rdoql / -rdoqls, chroma -rdoqc / -rdoqcs); tab = index of the block's 180-word bit table in dev_tables (estBitRdoq enc@0x46a8a0's output for the block's size and
 * component: a host snapshot of the entropy coder's context states); last_pos / dev_sigmask (64 words per block, group scan order, bit 15 - position) = what the quantiser's
 * bookkeeping hands over (scanSigFlags enc@0x4a9b00 lineage), updated; tu5 = TTransUnit+5, flag_a4c0 = TCtuInfo+0xa4c0 (which coded-block-flag words price "all zero");
 * sdh = sign-data hiding (cfg+0x3e0).  dev_out[2 i] = number of non-zero levels, dev_out[2 i + 1] = new last scan position (-1: none); dev_hidden[i] = groups that hide a sign.
 * Pinned on calls recorded inside appencoder runs: tests/golden/rdoq.npz, tests/test_gpu_rdoq.py. */
typedef struct { int32_t off, tab, dq, last_pos; int64_t lam, lam_sdh; int8_t log2, scan_idx, comp, per, tu5, flag_a4c0, sdh, rsv; } ks265_rdoq_tu;
int ks265_rdoq_batch(ks265_ctx *, const ks265_rdoq_tu *dev_tus, int n, int16_t *dev_lvl, const int16_t *dev_coef, const int32_t *dev_tables, uint16_t *dev_sigmask,
                     int32_t *dev_out, uint64_t *dev_hidden);
/* g_DeQuantFuncs (H265DeQuantBlock_c enc@0x439210): full-block form (lastX = lastY = N-1) */
int ks265_dequant_batch(ks265_ctx *, int n, const int16_t *dev_lvl, int16_t *dev_coef, int scale, int add,
                        int shift, int nblk);
/* the same function with its lastX / lastY arguments (position of the last significant level): only rows 0..lastY and columns
 * 0..round_up(lastX + 1, 4) - 1 of each block are written, the rest of dev_coef keeps what it held; blocks of n rows x `stride` */
int ks265_dequant_rect_batch(ks265_ctx *, int n, int stride, const int16_t *dev_lvl, int16_t *dev_coef, int scale, int add,
                             int shift, int lastX, int lastY, int nblk);
/* g_H265_2dIDct_Func[idx]: coef (packed NxN) + pred (packed NxN u8) -> recon (packed NxN u8) */
int ks265_inv_transform_batch(ks265_ctx *, int idx, const int16_t *dev_coef, const uint8_t *dev_pred,
                              uint8_t *dev_dst, int nblk);

/* one deblocking edge: pix_off = byte offset of q0 of the first line (EdgeFilterLumaVer_c enc@0x403630) */
typedef struct { int32_t pix_off; int16_t beta, tc; int16_t length; uint8_t dir /*0 ver,1 hor*/, flags /*bit0 filterP, bit1 filterQ*/; } ks265_edge;
/* g_EdgeFilterLuma{Ver,Hor}Func: edges of one batch must not overlap (true for one direction of one picture) */
int ks265_edge_filter_luma_batch(ks265_ctx *, uint8_t *dev_plane, int stride, const ks265_edge *dev_edges, int n);
/* g_PixelFilterChroma{Ver,Hor}Func */
int ks265_edge_filter_chroma_batch(ks265_ctx *, uint8_t *dev_plane, int stride, const ks265_edge *dev_edges, int n);

/* g_interp{Luma,Chroma}{Hor,Ver}{8to8,8to16,16to8,16to16}_func on whole rectangles.
 * kind: bit0 chroma, bit1 vertical, bits 2-3: 0 = 8to8, 1 = 8to16, 2 = 16to8, 3 = 16to16. Strides in elements. */
int ks265_interp_rect(ks265_ctx *, int kind, void *dev_dst, int dstStride, const void *dev_src, int srcStride,
                      int w, int h, int frac);

/* g_fSaoApplyOffsetBo (SaoApplyOffsetBo_c enc@0x43e4e0) on a rectangle, in place; width is rounded up to 4
 * and bands past 31 are dropped exactly like the reference */
int ks265_sao_apply_bo_rect(ks265_ctx *, const int8_t offsets[4], uint8_t *dev_rec, int stride, int height,
                            int width, int bandPosition);
/* SaoApplyOffsetEo{0..3}_c enc@0x43e650.. in their plain mode, out of place (dst may not alias src):
 * offsets[5] indexed by 2 + sign(c-a) + sign(c-b) */
int ks265_sao_apply_eo_rect(ks265_ctx *, int eoClass, const int8_t offsets[5], const uint8_t *dev_src,
                            uint8_t *dev_dst, int stride, int height, int width);
/* g_statBoEo01_funcs (statSaoBoEo01_c enc@0x4ae9c0): packed accumulators (sum << 12 | count), eoJoint[64], bo[32];
 * one launch accumulates nrect rectangles (x, y, w, h) each into its own 96-int record (eoJoint then bo) */
typedef struct { int32_t org_off, rec_off, w, h; } ks265_sao_rect;
int ks265_sao_stats_batch(ks265_ctx *, const uint8_t *dev_org, int orgStride, const uint8_t *dev_rec, int recStride,
                          const ks265_sao_rect *dev_rects, int nrect, int rowStep, int32_t *dev_out /*nrect x 96*/);

/* g_IntraPredFunction enc@0x7070a0 (35 modes x sizes).  The reference kernels
 * IntraPredPlanar_0_c enc@0x425af0 / IntraPredDC_1_c enc@0x425d80 / IntraPredChromeDC_1_c enc@0x425c60 /
 * IntraPredAng{Hor,Ver}*_c enc@0x425f60..0x426ce0 all take (u8 *dst, int dstStride, u8 *ref, int mode, int log2Size, bool edgeFilter)
 * with `ref` pointing at the corner sample p[-1][-1] of a linear array: ref[1 + x] = top and top-right (x < 2N),
 * ref[-1 - y] = left and bottom-left (y < 2N).  One descriptor = one such call; ref_off / dst_off are byte offsets of the
 * corner sample / the block's first sample in dev_ref / dev_dst.  mode 0 planar, 1 DC, 2..34 angular; log2 2..5;
 * edge_filter = the boundary smoothing of DC / 10 / 26 (luma; chroma DC passes 0 = IntraPredChromeDC_1_c). */
typedef struct { int32_t ref_off, dst_off; int16_t dst_stride; uint8_t mode, log2, edge_filter, rsv[3]; } ks265_intra_blk;
int ks265_intra_pred_batch(ks265_ctx *, const uint8_t *dev_ref, uint8_t *dev_dst, const ks265_intra_blk *dev_blks, int n);
/* g_IntraPredFilterRefFunc enc@0x706d48 -> IntraPredFilterRef_c enc@0x424110 (src, dst, size, strongEnabled): [1 2 1]/4 smoothing of
 * the 4*size + 1 samples around the corner (ends copied); size 32 with strong_enabled tests the flatness condition itself and
 * writes the bi-linear array instead.  Offsets address the corner sample. */
typedef struct { int32_t src_off, dst_off, size, strong_enabled; } ks265_intra_ref;
int ks265_intra_filter_ref_batch(ks265_ctx *, const uint8_t *dev_src, uint8_t *dev_dst, const ks265_intra_ref *dev_refs, int n);

/* Lookahead / pre-analysis leaf kernels (the frame-level cost logic calcFrameCost enc@0x4a7410 / scenecut
 * enc@0x47e9d0 stays host code - a lowres search is stage A of section 3 run on a half-size frame object).
 * g_downsampleFunc enc@0x707b10 -> downsample_c enc@0x4a6a60 (dst, src, dstStride, srcStride, w, h): 2:1 both ways, w x h = OUTPUT size */
int ks265_downsample_rect(ks265_ctx *, const uint8_t *dev_src, int srcStride, uint8_t *dev_dst, int dstStride, int w, int h);
/* the same with the source in pinned host memory (ks265_host_malloc; pointer and stride multiples of 8, dev_dst and dstStride multiples of 4): a kernel of 64 work-groups that
 * reads the picture over PCIe - no copy-engine transfer that would queue behind a pipelined host's uploads, no machine-filling grid waiting for PCIe */
int ks265_downsample_from_host(ks265_ctx *, const uint8_t *pinned_src, int srcStride, uint8_t *dev_dst, int dstStride, int w, int h);
/* weightBi_sad_c enc@0x4a7170 (org, orgStride, ref0, ref1, stride0, stride1, w, h): SAD against the rounded average of two references;
 * descriptor: a_off = org offset, b_off[0] / b_off[1] = offsets in ref0 / ref1 (b_off[2] unused) */
int ks265_weight_bi_sad_batch(ks265_ctx *, const uint8_t *dev_org, int orgStride, const uint8_t *dev_ref0, int stride0, const uint8_t *dev_ref1, int stride1,
                              const ks265_blk3 *dev_blks, int n, uint32_t *dev_out);
/* g_interMeBiFull_func / g_interMeBiHadFull_func (interMeBiFull_c enc@0x4896d0, interMeBiHadFull_c enc@0x4897e0): the 8 x 8 integer window of the joint
 * bi-prediction refinement.  Block i: a_off = target block (clip8(2 org - pred) of calcBiMeOrg), b_off = top-left window position in the reference plane,
 * w x h the block (had: multiples of 4); mvcost[16 i ..] = 8 column costs then 8 row costs.  out[2 i] = minimum of SAD|HAD + mvcost[x] + mvcost[8 + y]
 * (first in scan order, rows outside), out[2 i + 1] = (y << 16) | x. */
int ks265_bi_full_batch(ks265_ctx *, int use_had, const uint8_t *dev_org, int so, const uint8_t *dev_ref, int sr, const ks265_blk *dev_blks,
                        const uint16_t *dev_mvcost, int n, uint32_t *dev_out);
/* g_acEnergyPlaneFunc enc@0x707b20 -> acEnergyPlane_c enc@0x4650e0 (src, stride, log2Size): ssd - (sum^2 >> 2 log2N) in 32-bit unsigned
 * arithmetic (the wrap of sum^2 for bright 32x32 blocks is part of the contract); _batch: explicit block offsets, _map: every aligned
 * N x N block of a w x h plane in raster order (the per-block variance map of calcFrameAdaptQuant enc@0x4653c0) */
int ks265_ac_energy_batch(ks265_ctx *, const uint8_t *dev_src, int stride, int log2, const int32_t *dev_offs, int n, uint32_t *dev_out);
int ks265_ac_energy_map(ks265_ctx *, const uint8_t *dev_plane, int stride, int w, int h, int log2, uint32_t *dev_out);
/* calcFrameAdaptQuant enc@0x4653c0 (TInputPic*, mode = 1, strength) - adaptive quantisation by block variance (iAqMode / fAqStrength, qy265enc.h:145-146): for every 16 x 16
 * luma block (with its two 8 x 8 chroma blocks) v = log2(AC energy + 2)^2 (_log2 enc@0x4c3c20), then offset = (v - mean) x strength x mean / 6000 (doubles, the mean a
 * sequential sum divided by `count`) and its inverse qscale factor qy265_exp2fix8 enc@0x4c3c50.  dev_qp_off: nx * ny doubles (TInputPic lookahead block +0x9a8),
 * dev_inv_qscale: nx * ny u16 (+0x48), dev_scratch2: two doubles.  Bit-exact against oracle/ks265_lookahead_ref.c (pinned on recorded calls of the reference). */
int ks265_frame_adapt_quant(ks265_ctx *, const uint8_t *dev_y, int stride_y, const uint8_t *dev_u, const uint8_t *dev_v, int stride_c, int nx, int ny, int count,
                            double strength, double *dev_qp_off, uint16_t *dev_inv_qscale, double *dev_scratch2);
/* the QP of every CTU (raster, (nx + 3) / 4 x (ny + 3) / 4) from the offsets of ks265_frame_adapt_quant: base_qp + clip(round(mean of the CTU's 16 x 16 blocks), -12, 12), clipped to
 * [qp_lo, qp_hi] - the map ks265_frame_set_qp_map takes (this build's rule: quantisation group = CTU) */
int ks265_aq_ctu_map(ks265_ctx *, const double *dev_qp_off, int nx, int ny, int base_qp, int qp_lo, int qp_hi, int8_t *dev_map);
/* cuTreePropagate enc@0x47d460 (log2, frames, p0, p1, b): one step of the macroblock-tree propagation - every block of picture b hands
 * ((inv_qscale x intra + 128) >> 8) + own) x (intra - inter) / intra to the (up to) four blocks of each reference its vector points at, weighted by the overlap
 * (32nds at lg = 3), half each when both lists are used; reference costs saturate at 0xffff.  list bits: 2 per block, 4 blocks per byte; vectors: x = low 16 bits,
 * y = high 16; dev_ref0 == dev_ref1 when both references are one picture; dev_acc: 2 * nx * ny 64-bit words, ZERO on entry and left zero. */
int ks265_cutree_propagate(ks265_ctx *, int lg, int nx, int ny, const uint16_t *dev_intra, const uint16_t *dev_inv_qscale, const uint16_t *dev_own, const uint16_t *dev_inter,
                           const uint8_t *dev_list_bits, const int32_t *dev_mv0, const int32_t *dev_mv1, uint16_t *dev_ref0, uint16_t *dev_ref1, uint64_t *dev_acc);

/* calcFrameCost enc@0x4a7410 (TEncParam*, TInputPic* ref0, TInputPic* ref1, TInputPic* cur, int d0, int d1, int flag) - the lookahead's cost of coding picture `cur` from the
 * picture d0 back and the picture d1 ahead, on the HALF-SIZE pictures (downsample_c enc@0x4a6a60), per block of 8 x 8 (lg 3) or 16 x 16 (lg 4: 1080p and up, veryfast and below):
 * list-0 / list-1 diamond search started from the right / lower neighbours' vectors (meInitPoint enc@0x48af50 + interMeDia enc@0x48fbe0 with the lambda of QP 12), list cost + 4,
 * the bi-predictive average (weightBi_sad_c enc@0x4a7170) + 9 if its SAD + 5 is below the better list, intra = best SAD of {planar, DC, 26, 10, 18, 2, 34} refined by +-2, +-1,
 * + 9; per block min(cost, 0xffff), the lists used (2 bits), the vectors (quarter pel) and list costs; per picture the cost sums (L+0x684 / +0x7c8: plain and weighted by the
 * inverse qscale of calcFrameAdaptQuant; B pictures x 100 / 130) and the motion statistics (L+0x90c..0x918) that scenecut enc@0x47e9d0, the slice-type decision, cuTreePropagate
 * enc@0x47d460 and the rate control read.  Words of TEncParam by offset (names are this build's): merange +0x710, lg +0x3c0, zero_thr +0x3a0, fast_intra +0x3a4, scenecut +0x390,
 * preset +0xc, p8 +0x8, aq +0x378, b_intra +0x388 (cuTree: B pictures compare against intra too), f3a8 / f36c / f538 / f3b4 (+0x3a8 ..): switches of the motion statistics.
 * do_list[l]: search list l (else the stored vectors and costs of an earlier call with the same distance are used, as the reference does when the vector plane's first word is
 * not 0x7fff); intra_done: L+0x18 (the intra costs of `cur` exist).  The reference's "already computed" shortcut (L+0x684[9 d0 + d1] >= 0) is the caller's.
 * Planes: pointers to sample (0, 0), one stride, readable 40 samples beyond the picture on every side (edge-replicated like the reference's 32); the block grid may overhang the
 * picture's last rows (1080p: 34 x 16 = 544 > 540).  Bit-exact against oracle/ks265_lookahead_ref.c, which is pinned on 461 recorded calls of the reference. */
typedef struct {
    int32_t w, h, nx, ny, cnt, stride;
    int32_t d0, d1, flag, slice_type;
    int32_t merange, lg, zero_thr, fast_intra, scenecut, preset, p8, aq, b_intra, f3a8, f36c, f538, f3b4;
    int32_t do_list[2], intra_done;
    uint16_t lambda_tab[52];           /* TEncParam+0x720: the integer motion lambda of every QP (entry 12 is used; the others only where the reference overruns a table row) */
} ks265_cfc_params;
typedef struct { int32_t intra_wins, sum_intra, sum_intra_aq, sum, sum_aq, stats[4], ret, intra_done; } ks265_cfc_sums;   /* L+0x660[d0], +0x684[0], +0x7c8[0], +0x684[idx], +0x7c8[idx], +0x90c+16 d0, return, L+0x18: in / out */
size_t ks265_calc_frame_cost_workspace(int nx, int ny);
int ks265_calc_frame_cost(ks265_ctx *, const ks265_cfc_params *, const uint8_t *dev_cur, const uint8_t *dev_ref0, const uint8_t *dev_ref1, uint16_t *dev_intra, uint8_t *dev_imode,
                          const uint16_t *dev_inv_qscale, uint16_t *dev_inter, uint8_t *dev_list_bits, int32_t *dev_mv0, int32_t *dev_cost0, int32_t *dev_mv1, int32_t *dev_cost1,
                          ks265_cfc_sums *dev_sums, void *dev_ws);
/* the cuTree finish inlined in CInputPicManage::updateQueue (enc@0x480964..0x480a54): out[i] = clip(aq_off[i] - 1.8 (log2(propagate[i] (x 2 if dbl) + intra'[i]) - log2(intra'[i])), -15, 20)
 * with intra' = (intra x inv_qscale + 128) >> 8 and the reference's table log2 (_log2 enc@0x4c3c20); blocks with intra' = 0 keep what out holds */
int ks265_cutree_finish(ks265_ctx *, int cnt, const uint16_t *dev_intra, const uint16_t *dev_inv_qscale, const uint16_t *dev_propagate, const double *dev_aq_off, int dbl, double *dev_out);
/* helpers of the host's cuTree pass (host/ks265_enc.c ct_*): edge replication around a plane (pointer to sample (0, 0)); a constant into a u16 plane (the inverse qscale 256 of a
 * picture without adaptive quantisation); one QP per CTU from the offsets of the lookahead's blocks (2^(lg + 1) luma samples each: 4 x 4 or 2 x 2 per CTU) by ks265_aq_ctu_map's rule */
int ks265_pad_plane(ks265_ctx *, uint8_t *dev_p00, int stride, int w, int h, int pad);
int ks265_fill_u16(ks265_ctx *, uint16_t *dev, int n, int value);
int ks265_qoff_ctu_map(ks265_ctx *, const double *dev_off, int nx, int ny, int lg, int ctu_cols, int ctu_rows, int base_qp, int qp_lo, int qp_hi, int8_t *dev_map);

/* ------------------------------------------------------------------ 3. whole-frame stages (a13 sequencing) */

typedef struct ks265_frame ks265_frame;

typedef struct {
    int32_t width, height;     /* luma, multiples of 8                                            */
    int32_t qp;                /* slice QP actually used for this picture (after the reference's
                                  hidden hierarchy offset, SURVEY.md §5: I = Q, P = Q+1)          */
    int32_t lambda_q4;         /* motion lambda in Q4 fixed point (host-side float setup only)     */
    int32_t me_range;          /* integer search range in pels, <= 64 (all presets use 64)         */
    int32_t me_method;         /* -me: 0 = DIA (interMeDia enc@0x48fbe0), 1 = HEX (interMeHex enc@0x48fde0), 2 = UMH (interMeUMH enc@0x4907b0) */
    int32_t subme;             /* -subme (qy265enc.h:137): 0 = integer only; 1 = "fast", 2 = "square full": the reference's refinement per PU - getMvResolution
                                  enc@0x483ca0, subMeSquare enc@0x4b5660 (eight half-sample candidates, order 3 4 1 6 0 2 5 7; eight quarter-sample candidates, order
                                  1 6 3 0 5 4 2 7; strict '<' against the integer cost; fast = only candidates within +-2 quarter samples of the integer position and
                                  diagonals next to the running winner) - restated in oracle/ks265_subme_ref.c, pinned on recorded calls (tests/test_subme.py) */
    int32_t deblock;           /* -df                                                               */
    int32_t sao;               /* -sao: 0 off, 1 BO + EO0..3 (this build's rule), 2 = the reference's decision: BO + EO0 / EO1 by CEncSao::modeDecisionBoEo01 enc@0x4af300 */
    int32_t beta_offset_div2, tc_offset_div2;
    int32_t bframes;           /* > 0: allocate the second-list workspace (planes, PU records) for B pictures (-bframes) */
    int32_t refs;              /* list-0 reference pictures a P picture may search (-ref / -ref0), 0 or 1 = one, at most 4 */
    int32_t me_hex_thr;        /* tME+0x368 of motionSearchOneRef enc@0x483f40: with -me 2, a PU whose start-point SAD is below this many units
                                  per sample runs interMeHex instead of interMeUMH; 16 at -preset slow, 0 (= always UMH) at veryslow */
    int32_t sdh;               /* the postQuant seam (postQuant enc@0x4ace80 between H265QuantBlock and H265DeQuantBlock): 1 = sign-data hiding on the
                                  quantised levels (signBitHidingHDQ enc@0x4aa150; the reference's PPS has sign_data_hiding_enabled_flag = 1) */
    int32_t pre_search;        /* 1 = stage A evaluates one more start candidate per PU: the vector of an exhaustive search on a three-level pyramid
                                  (ks265_presearch; the reference's lookahead searches 2:1 pictures, downsample_c enc@0x4a6a60, and meInitPoint
                                  enc@0x48af50 picks the best of several start candidates) */
    int32_t merge;             /* 1 = stage C2 after the CU decision: every CU may adopt the motion of one of its spatial merge neighbours or the zero vector
                                  (ks265_merge_pass; the reference's merge / skip decision: GetMergeCandsFor*, skipFastDecision) - single reference per list */
    int32_t bi_refine;         /* 1 = B pictures: joint refinement of the bi-predictive pair inside ks265_bi_decide (motionSearchBI enc@0x484910): the cheaper
                                  list stays, the other one is searched again against clip8(2 org - pred) (calcBiMeOrg enc@0x47b1a0) over the 8 x 8 integer
                                  window of interMeBiFull enc@0x4896d0 / interMeBiFull_opt enc@0x4898e0, then over the sub-pel ring.
                                  2 (round 5; what the encoder host runs) = the same refinement AFTER the CU decision, for the 2N x 2N inter CUs it chose
                                  (ks265_bi_refine_chosen, in front of the merge pass): every picture area refined once instead of once per quadtree level */
    int32_t decimate;          /* K > 0: coefficient decimation at the postQuant seam of inter LUMA TUs - a block whose levels are all +-1 and at most 2K (8x8),
                                  3K (16x16), 4K (32x32) of them is dropped (prediction only, cbf 0); chroma is left alone; the encoder host uses 2.  Where the reference makes this kind of
                                  decision is inside its closed RD code (tuDecision enc@0x4825a0 lineage); measured effect: DESIGN.md 8 */
    int32_t rdo;               /* K > 0: coefficient-group pruning at the postQuant seam, before sign-data hiding - the sub-block decision of HM-lineage RDOQ (the reference's
                                  rdoQuant enc@0x4aac50 is closed code; -rdoq 1 at -preset slow) with STATIC bit costs: a 4x4 group of levels of an inter TU (luma and chroma;
                                  also the intra CUs of P / B pictures) is kept only if the distortion it removes, sum d (2 c - d) >> 2 (7 - log2 N) over its levels (c coefficient,
                                  d dequantised level), exceeds lambda_mode x K / 4 x its bits (quarter bits: 14 / 20 / 26 + 8 floor(log2(|l| - 1)) per level + 10 + the zeros
                                  around); lambda_mode = (lambda_q4 / 16)^2.  The encoder host uses K = 4 with the P / B lambda table; key pictures are never pruned */
    int32_t intra_inter;       /* 1 = P / B pictures may hold intra CUs (EncIntraMD.cpp lineage: decideLumaMode enc@0x49acc0): the pre-selection cost of ks265_intra_decide competes in
                                  the CU decision, intra CUs are reconstructed after the inter CUs from reconstructed neighbours (CTU wavefront) */
    int32_t propagate;         /* n > 0: n rounds of ks265_me_propagate between the integer search and the sub-pel step of every search of ks265_encode_picture[_b|_mref]
                                  (meInitPoint enc@0x48af50 starts from the coded neighbours' vectors; a frame-parallel search gets them this way).  0..4; the encoder host uses 1 */
    /* the sub-pel refinement's knobs = what the reference's presets put into its configuration (read inside real encodes, oracle/ref_probe/subme_shim.c) */
    int32_t sub_satd;          /* tME+0x64 / TPredUnit+0x40: 0 = candidates judged by SAD (every preset up to slower), 1 = by Hadamard, start cost recomputed (veryslow, placebo) */
    int32_t sub_thr;           /* cfg+0x464 = tME+0x3c0: > 0 = a PU is refined only if the steepest of the four integer neighbours exceeds the winner's SAD by W H thr / 32, and a
                                  flat half step skips the quarter step; 80 / 76 / 68 / 56 / 40 / 24 / 24 / 0 / 0 from ultrafast to placebo */
    int32_t sub_flat;          /* tME+0x3c4: flat = max SAD of the half step - the winner's SAD <= W H flat / 8; 40 / 36 / 16 / 14 / 10 / 8 / 8 / 8 / 8 */
    int32_t sub_cap, sub_cap_step;   /* cfg+0x498 / +0x49c: > 0 = no refinement above an integer cost of (cap + (6 - log2 H) step) W^2; 6,6 / 6,6 / 12,6 / 0.. */
    int32_t sub_diag_fast;     /* cfg+0x580: half step skips the diagonals unless a horizontal / vertical candidate won (ultrafast, superfast) */
    int32_t part;              /* -part (qy265enc.h:131; slower, veryslow, placebo): 1 = a CU of 64 / 32 / 16 samples of a P or B picture (one reference per list) may be coded as two 2NxN or Nx2N prediction units
                                  (ks265_cu8.log2_cu bits 4..5); each half is priced with the refined vectors of the CU and of its two quarter-size PUs (ks265_rect_decide), the CU then
                                  holds four transform units (interSplitFlag).  The reference searches such PUs on their own inside its RD loop (closed code) */
    int32_t tu_inter;          /* -intertu 1 (tuInter, qy265enc.h:133; veryslow, placebo: the residual quadtree of inter CUs one level deep): a 2Nx2N inter CU of 32 / 16 samples is coded with four
                                  transform units (split_transform_flag; ks265_cu8.log2_cu bits 4..5 = 3, set by ks265_reconstruct*) when its luma residual sits in part of it - the quarters'
                                  residual SADs under the CU's final motion: max > 4 x min + (N / 2)^2.  The reference decides this inside its RD loop (tuDecision enc@0x4825a0, closed code) */
    int32_t skip_rd;           /* round 6 - Stage D2, the skip pass (ks265_skip_pass; run by ks265_encode_picture_b / _b_mref - with 2 also by ks265_encode_picture / _mref: P pictures gain
                                  nothing measurable - after the reconstruction of the inter CUs): per CTU, top-down over the nodes of
                                  64 / 32 / 16 / 8 samples that hold inter CUs only, the node becomes ONE CU without residual carrying a merge candidate's motion when SSE(source, that
                                  prediction) + lambda x (1 + position) bits is below SSE(source, reconstruction) + lambda x (level + syntax bits) of what the node holds now - the decision on the
                                  coded distortion the reference takes in skipFastDecision enc@0x486090 / skipFullMergeDecision enc@0x482da0 (closed code).  Needs the spare CU map (cfg.merge or this) */
} ks265_frame_cfg;

/* geometry of the padded picture buffers the caller allocates (one call, no allocation) */
typedef struct {
    int32_t pad_y, pad_c;          /* border in samples (80 / 40)                  */
    int32_t stride_y, stride_c;    /* bytes per row                                */
    int32_t rows_y, rows_c;        /* rows incl. borders                           */
    int64_t bytes_y, bytes_c;      /* plane allocations = stride x (rows + 1): one slack row, so that aligned window loads of the last padded row stay inside the allocation */
    int32_t ctu_cols, ctu_rows;
    int32_t pu_per_ctu;            /* 85: 1 + 4 + 16 + 64                          */
    int64_t bytes_pu;              /* ks265_pu records                             */
    int64_t bytes_cu8;             /* ks265_cu8 records, (H/8) x (W/8)             */
    int64_t bytes_sao;             /* ks265_sao_param records, 3 per CTU           */
} ks265_frame_geom;

/* motion / cost record of one PU (level 0: 64x64 ... level 3: 8x8; raster order inside the CTU) */
typedef struct { int16_t mvx, mvy; int16_t mvpx, mvpy; uint32_t cost; uint32_t dist; } ks265_pu;   /* quarter-pel units; mvp = predictor used for the rate term; dist = SAD (stage A) or SATD (stage B) */
/* final coding decision per 8x8 luma block */
typedef struct { int16_t mvx, mvy; /* list 0 */ int16_t mv1x, mv1y; /* list 1 */ uint8_t log2_cu; uint8_t cbf; /* bit0 Y, bit1 Cb, bit2 Cr */
                 uint8_t pred_mode; /* 0 inter, 1 intra(flat 128 stand-in), 2 intra (mvx = luma mode) */ uint8_t inter_dir; /* 1 = L0, 2 = L1, 3 = bi */ } ks265_cu8;
/* B pictures: the per-PU winner among L0, L1 and bi-prediction */
typedef struct { int16_t mvx, mvy, mv1x, mv1y; uint32_t cost; uint32_t inter_dir; } ks265_pu_b;
/* SAO decision per CTU and component */
typedef struct { int8_t type; /* -1 off, 0 BO, 1..4 EO class 0..3 */ int8_t band; int8_t offset[4]; int8_t rsv[2]; } ks265_sao_param;

/* a padded YUV 4:2:0 picture in HBM (pointers to the first byte of each allocation) */
typedef struct { uint8_t *y, *u, *v; } ks265_pic;

int ks265_frame_geometry(const ks265_frame_cfg *cfg, ks265_frame_geom *geom);
int ks265_frame_create(ks265_ctx *ctx, const ks265_frame_cfg *cfg, ks265_frame **out);
/* frames keep a pointer to their context: destroy every frame BEFORE ks265_destroy(ctx) */
void ks265_frame_destroy(ks265_frame *f);
/* round 4: one QP per CTU (raster order; DEVICE memory that stays valid while pictures coded with it are in flight; NULL = cfg.qp everywhere) for the pictures coded from
 * here on: every CTU's residual is quantised with its entry (chroma through the table), the deblocking filter runs at the QpY the DECODER derives for each CU
 * (cu_qp_delta with the quantisation group = the CTU: H.265 8.6.1 - ks265_stream_cfg.cu_qp_delta, ks265_slice_in.qp_map carry it into the stream).  The decisions
 * (lambda) stay on the picture's QP.  GPU == oracle with random maps: tests/test_gpu_dqp.py; the decoder's verdict on the oracle: tests/test_dqp.py */
int ks265_frame_set_qp_map(ks265_frame *f, const int8_t *dev_qp_map);
int ks265_frame_set_qp(ks265_frame *f, int qp, int lambda_q4);
/* round 6 - tools per picture: a host that codes a whole pyramid on one frame object lowers tools for some of its pictures (host/ks265_enc.c: the B pictures nothing predicts
 * from run without intra candidates, without the joint refinement and without SAO - measured on the CPU mirror and on the MI355X: the bytes at equal PSNR-Y stay, a quarter
 * of such a picture's kernel time goes).  Each argument: -1 = the value the frame object was created with, else the value cfg.intra_inter / cfg.bi_refine / cfg.sao take for
 * the pictures coded from now on - 0 or the created value (the workspace is the creation's); me_method: -1 = as created, else 0 .. 2 (interMeDia / interMeHex / interMeUMH: the
 * search method needs no workspace of its own).  With cfg.sao = 0 a B picture is reconstructed and deblocked straight in
 * recon_out (no SAO launch, no copy); its SAO records are written as "off" and the host passes ks265_slice_in.sao = NULL (slice_sao_luma_flag = slice_sao_chroma_flag = 0). */
int ks265_frame_set_picture_tools(ks265_frame *f, int intra_inter, int bi_refine, int sao, int me_method);

/* expandPicture_c enc@0x4a6ae0: replicate the picture edge into the borders of all three planes */
int ks265_pad_picture(ks265_frame *f, ks265_pic pic);
/* copy an unpadded I420 frame (dev, W*H*3/2 bytes) into a padded picture and pad it */
int ks265_load_i420(ks265_frame *f, const uint8_t *dev_i420, ks265_pic dst);
/* the same on another context's stream (same device): a pipelined host unpacks and pads the NEXT source picture beside the picture being coded; it orders the
 * streams with events (ks265_event_record / ks265_stream_wait_event) */
int ks265_load_i420_on(ks265_ctx *cx, ks265_frame *f, const uint8_t *dev_i420, ks265_pic dst);
/* inverse: padded picture -> packed I420 */
int ks265_store_i420(ks265_frame *f, ks265_pic src, uint8_t *dev_i420);

/* (Rounds 1 - 2 had a stage "ks265_ref_planes" here: sixteen precomputed fractional-sample planes per reference picture.  Every stage below now interpolates the
 * samples it needs from the padded reference picture itself - the normative 8-tap filters of interpLuma* enc@0x40e4f0..0x4109b0, per candidate as
 * subMeHpel_RealInterp enc@0x4b4e90 does - so the stages take the reference PICTURE.) */
/* Stage A0 (cfg.pre_search; run by ks265_me_integer itself, exported for stage tests): exhaustive motion search on a pyramid built with
 * downsample_c enc@0x4a6a60 - L2 blocks of 8x8 over +-(range/4 - 1), L1 blocks +-2, 16x16 picture blocks +-1; cost SAD + |mx| + |my|.
 * Above them a 1/8-resolution level: every CTU (one 8x8 block) over +-range/4 = +-2 range samples; its vector x 8 is the CTU's WINDOW OFFSET - stage A searches
 * +-range around it, so a reference several pictures away is in reach (the reference searches around its predicted vector; its pictures have no window).
 * dev_field: ceil(W/16) x ceil(H/16) x {mvx, mvy} int16, integer pel (NULL = skip the stand-alone full-resolution step); dev_ctu_off: ctu_cols x ctu_rows x
 * {ox, oy} int16 (NULL = the frame's own buffer) */
int ks265_presearch(ks265_frame *f, ks265_pic src, ks265_pic ref, int16_t *dev_field, int16_t *dev_ctu_off);
/* Stage A: integer-pel motion search for every PU of every CTU (motionSearchOneRef enc@0x483f40 ->
 * interMeDia enc@0x48fbe0 over sad4_c); prev_pu = PU records of the previous picture (temporal
 * predictor) or NULL */
int ks265_me_integer(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *dev_prev_pu, ks265_pu *dev_pu);
/* Stage A2 (cfg.propagate; run by ks265_encode_picture[_b|_mref] itself, exported for stage tests): one round of vector propagation.  Every PU of dev_in (the
 * records of ks265_me_integer for this src / ref pair - the CTUs' window offsets of that call are used for the vector limits) tries the integer vectors of its
 * left / above / right / below neighbours of the same size (across CTU borders; skipping vectors outside its CTU's limits, its own and repeats): SAD + its own
 * vector rate, strict '<', in that order; all 85 records per CTU go to dev_out (!= dev_in: every PU reads the state before the round) */
int ks265_me_propagate(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *dev_in, ks265_pu *dev_out);
/* Stage B: 8 half-pel + 8 quarter-pel refinement with SATD (subMeSquare enc@0x4b5660 ->
 * subMeHpel_RealInterp / subMeQpel_8Sad_*_RealInterp + had_c) */
int ks265_me_subpel(ks265_frame *f, ks265_pic src, ks265_pic ref, ks265_pu *dev_pu);
/* cfg.part: the CU decision of a P picture with 2NxN / Nx2N partitions (prices the halves of every 64 / 32 / 16 CU first); dev_ibest as for ks265_cu_decide_ii (may be null) */
int ks265_cu_decide_part(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *dev_pu, const uint32_t *dev_ibest, ks265_cu8 *dev_cu8);
/* the same for a B picture (round 5): a half takes the MOTION - direction and vector(s) - of the CU's record or of one of its two quarter-size records in dev_pub, priced by the
 * Hadamard cost of its prediction (bi: rounded average) + the rate of every vector used against the CU's predictors (dev_pu0 / dev_pu1: the uni-directional searches' records),
 * a bi-predictive half at 31 / 32; ks265_cu8.inter_dir / mv1 then differ between the halves too */
int ks265_cu_decide_part_b(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *dev_pu0, const ks265_pu *dev_pu1, const ks265_pu_b *dev_pub,
                           const uint32_t *dev_ibest, ks265_cu8 *dev_cu8);
/* Stage C2 (cfg.merge; run by ks265_encode_picture[_b] itself, exported for stage tests): merge pass on the motion field of the CU decision.
 * Per CU the five spatial merge neighbours of H.265 8.5.3.2.3 (inside the picture, earlier in z-scan order, inter) and the zero vector are tried as the
 * CU's own motion: SATD of the prediction (bi = rounded average) + lambda x (position + 1) against the search cost + 2 lambda.
 * dev_pu for P pictures (ref1 = null picture), dev_pub for B pictures (the other NULL); cu_in and cu_out must differ (all CUs decide on the same input field). */
int ks265_merge_pass(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *dev_pu, const ks265_pu_b *dev_pub,
                     const ks265_cu8 *dev_cu_in, ks265_cu8 *dev_cu_out);
/* Stage D2 (cfg.skip_rd; run by ks265_encode_picture[_b|_mref|_b_mref] itself, exported for stage tests): after ks265_reconstruct* - dev_cu8 with its coded-block flags, the
 * level planes and the reconstruction are updated in place for the nodes that become one CU without residual; ref1 = null picture for P pictures; a multi-reference picture's
 * context (ks265_encode_picture_mref / _b_mref) supplies the pictures of every candidate.  Intra CUs are not touched (and keep a node from being looked at). */
int ks265_skip_pass(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, ks265_cu8 *dev_cu8, int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon);
/* Stage C: CU quadtree decision from the PU costs (the bottom-up compare of processTree enc@0x4722a0) */
int ks265_cu_decide(ks265_frame *f, const ks265_pu *dev_pu, ks265_cu8 *dev_cu8);
/* Stage C': key picture — every CU 32x32-TU "flat" intra (pred = 128); stands in for the out-of-scope
 * intra path so that a GOP has a reconstructed first picture */
int ks265_cu_flat_intra(ks265_frame *f, ks265_cu8 *dev_cu8);
/* Stage D: prediction (luma 8-tap, chroma 4-tap, both from the reference picture) -> residual -> DCT -> quant ->
 * dequant -> IDCT -> recon, the reconstruct() chain enc@0x481da0; levels are s16 planes of W x H (Y) and
 * W/2 x H/2 (Cb, Cr) coefficients stored TU-in-place; recon is a padded picture */
int ks265_reconstruct(ks265_frame *f, ks265_pic src, ks265_pic ref, ks265_cu8 *dev_cu8,
                      int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon);
/* the same for a B picture with its list-1 reference; bi-predicted blocks use the exact 14-bit average
 * (DefaultWeightedBi_c enc@0x435160 over interp*8to16 / 16to16) */
int ks265_reconstruct_b(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1,
                        ks265_cu8 *dev_cu8, int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon);
/* Stage B': B pictures - per PU the cheapest of L0, L1 (the two uni-directional searches) and the bi-predictive pair of
 * their winners (SATD against the rounded average; interMeBi* enc@0x486c10.. lineage, no joint refinement yet) */
int ks265_bi_decide(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *dev_pu0,
                    const ks265_pu *dev_pu1, ks265_pu_b *dev_pub);
/* cfg.bi_refine == 2: the joint refinement of motionSearchBI enc@0x484910 for the CUs the decision chose (dev_cu8 = what the CU decision wrote, in front of the merge
 * pass): a 2N x 2N inter CU whose refined pair is cheaper takes it - into its PU record (cost, vectors, direction) and into its 8 x 8 blocks.  One wave per CTU. */
int ks265_bi_refine_chosen(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, const ks265_pu *dev_pu0, const ks265_pu *dev_pu1,
                           ks265_pu_b *dev_pub, ks265_cu8 *dev_cu8);
int ks265_cu_decide_b(ks265_frame *f, const ks265_pu_b *dev_pub, ks265_cu8 *dev_cu8);
/* cfg.intra_inter - intra CUs in P / B pictures (EncIntraMD.cpp lineage: decideLumaMode enc@0x49acc0 is closed RD code).  ks265_intra_candidates: per 8x8 / 16x16 /
 * 32x32 block the pre-selection cost and best luma mode from source neighbours (the arithmetic of ks265_intra_decide, no CU tree), packed cost << 6 | mode in
 * dev_best (nctu x 85 uint32, PU indexing; 0xFFFFFFFF = no candidate).  ks265_cu_decide[_b]_ii: the CU tree in which a block's intra cost + lambda x 96 bits competes
 * with its inter cost (null = no candidates); an intra CU of the map has pred_mode 2, mvx = luma mode.  ks265_intra_inter_reconstruct: after ks265_reconstruct[_b]
 * has written every inter CU, the intra CUs in CTU wavefront order from reconstructed neighbours (inter and intra alike), with the slice's quantiser offset and
 * cfg.rdo. */
int ks265_intra_candidates(ks265_frame *f, ks265_pic src, const void *dev_pu_records /* ks265_pu or ks265_pu_b of the picture's inter search: a CTU is
                               evaluated only if one of its 8x8 PUs costs >= lambda x 96 bits */, uint32_t *dev_best);
int ks265_cu_decide_ii(ks265_frame *f, const ks265_pu *dev_pu, const uint32_t *dev_ibest, ks265_cu8 *dev_cu8);
int ks265_cu_decide_b_ii(ks265_frame *f, const ks265_pu_b *dev_pub, const uint32_t *dev_ibest, ks265_cu8 *dev_cu8);
int ks265_intra_inter_reconstruct(ks265_frame *f, ks265_pic src, ks265_cu8 *dev_cu8, int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon);
/* Intra pictures.  ks265_intra_decide: every 8x8 / 16x16 / 32x32 block tries all 35 luma modes of
 * g_IntraPredFunction on reference samples taken from the SOURCE picture (decideBestLumaModeBySadFast enc@0x499170 lineage),
 * cost = SATD (had_c) + lambda * mode bits, then the CU quadtree bottom-up; cu8 of an intra CU: pred_mode = 2, mvx = luma mode,
 * chroma = the luma mode.  ks265_intra_reconstruct: CTU wavefront (the reference's own WPP order,
 * CCtuEncWpp::waitForTopRightCtu enc@0x46f4a0: a CTU starts when its top-right neighbour is done), CUs in z-order, neighbours from the reconstructed picture
 * (H.265 6.4.1 availability, 8.4.4.2.2 substitution, 8.4.4.2.3 smoothing = IntraPredFilterRef_c enc@0x424110), then the
 * reconstruct() chain enc@0x481da0 per TU (TU = CU, at most 32x32). */
int ks265_intra_decide(ks265_frame *, ks265_pic src, ks265_cu8 *dev_cu8);
int ks265_intra_decide_ex(ks265_frame *, ks265_pic src, ks265_cu8 *dev_cu8, uint32_t *dev_cost /* optional: nctu x 85 pre-selection costs, PU indexing */);
/* Lookahead frame cost (calcFrameCost enc@0x4a7410 / scenecut enc@0x47e9d0 lineage - the reference's cost logic is
 * closed, this is the composition its kernels belong to): on HALF-RESOLUTION pictures (ks265_downsample_rect = downsample_c, then
 * ks265_pad_picture; a ks265_frame created at the half size), per 8x8 block the intra pre-selection cost against the integer-search cost
 * of the 8x8 PU.  dev_out[4] = { sum intra, sum inter, sum min(intra, inter), blocks | intra-cheaper blocks << 32 }; the host compares the
 * sums (scene cut / slice type), ks265_ac_energy_map gives the adaptive-quantisation variance map.
 * Size rule: the half-size frame obeys ks265_frame_geometry (multiples of 8), i.e. the SOURCE must be a multiple of 16 in both directions; for other
 * sources (1920x1080 -> 960x540) create the lookahead frame at the half size rounded DOWN to a multiple of 8 (960x536) and downsample that part of
 * the picture: the lookahead then ignores the last source rows / columns (< 16), which the cost sums tolerate.  ks265_frame_create(…540…) returns
 * KS265_NOTSUPPORTED. */
int ks265_lookahead_reduce(ks265_frame *, const uint32_t *dev_intra_cost, const ks265_pu *dev_pu, uint64_t *dev_out);
int ks265_lookahead_picture(ks265_frame *, ks265_pic cur_lowres, ks265_pic ref_lowres, uint32_t *dev_cost_ws /* nctu x 85 */, uint64_t *dev_out);
/* cur against another reference, with the intra costs the ks265_lookahead_picture call for cur left in dev_cost_ws (search + sums only).  dev_cost_ws = NULL: no intra
 * costs - dev_out[1] (the search's sum) as always, dev_out[0] = dev_out[2] = the same (the host's slice-type decision reads dev_out[1] alone; ks265_lookahead_reduce likewise) */
int ks265_lookahead_inter(ks265_frame *, ks265_pic cur_lowres, ks265_pic ref_lowres, const uint32_t *dev_cost_ws, uint64_t *dev_out);
int ks265_intra_reconstruct(ks265_frame *, ks265_pic src, ks265_cu8 *dev_cu8, int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon);
/* Stage E: in-place deblocking of a reconstructed picture (CalcBsInterP enc@0x402960, ctuDeblockFilterVer
 * enc@0x403de0, CtuDeblockFilterHorT enc@0x477200) */
int ks265_deblock(ks265_frame *f, const ks265_cu8 *dev_cu8, ks265_pic recon);
/* Stage F: SAO statistics + decision + apply (CEncSao::modeDecisionCtu enc@0x4af690, qy265SaoApplyOffset
 * enc@0x43fc00); dst becomes the next reference picture (borders padded) */
/* cfg.rdoq (round 6; the SDK's rdoq field, qy265enc.h:129): with tables set, ks265_reconstruct / _b / _mref send the luma transform blocks of inter CUs through the reference's
 * rdoQuant (the operator of ks265_rdoq_batch: levels q / q - 1 / 0, last position, all-zero groups, sign-data hiding by RD cost) between a front half (transform, levels rounded at
 * 1 / 2) and a back half (dequantisation, inverse transform) - in place of dead zone + coefficient-group pruning + sign-data hiding at the seam.  host_tables: [4 sizes][luma,
 * chroma][180] words of estBitRdoq enc@0x46a8a0; host_lam / host_lam_sdh: rdoQuant's two lambdas by QP [52].  All null: back to the default seam.  P / B pictures (intra CUs keep the seam) */
int ks265_frame_set_rdoq(ks265_frame *f, const int32_t *host_tables, const int64_t *host_lam, const int64_t *host_lam_sdh);
int ks265_sao(ks265_frame *f, ks265_pic src, ks265_pic deblocked, ks265_sao_param *dev_sao, ks265_pic dst);

/* Multi-reference P pictures (-ref / -ref0; motionSearchOneRef enc@0x483f40 runs once per reference picture): search every picture
 * with stages A0 / A / B, then ks265_ref_decide picks per PU the picture with the smallest cost + lambda * ref_idx bits (truncated
 * unary, ties to the nearest picture; pub.inter_dir = 1 | idx << 4, carried into cu8.inter_dir), ks265_cu_decide_b builds the CU tree,
 * ks265_reconstruct_mref predicts every CU from its own picture.  pu / refs: HOST arrays, nearest first. */
int ks265_ref_decide(ks265_frame *f, int nref, const ks265_pu *const *dev_pu, ks265_pu_b *dev_pub);
int ks265_reconstruct_mref(ks265_frame *f, ks265_pic src, int nref, const ks265_pic *refs, ks265_cu8 *dev_cu8,
                           int16_t *dev_lvl_y, int16_t *dev_lvl_u, int16_t *dev_lvl_v, ks265_pic recon);
/* a P picture with nref <= cfg.refs list-0 pictures (refs[0] = the nearest); nref == 1 is ks265_encode_picture(is_key = 0) */
int ks265_encode_picture_mref(ks265_frame *f, ks265_pic src, const ks265_pic *refs, int nref, ks265_pic recon_out);

/* the whole hot path for one picture: A0 (if is_key == 0) A B C D E F in stream order.
 * Workspace (planes, PU/CU/SAO records, levels) lives inside the frame object. */
int ks265_encode_picture(ks265_frame *f, ks265_pic src, ks265_pic ref, int is_key, ks265_pic recon_out);
/* a B picture: ref0 = list 0 (past), ref1 = list 1 (future); needs cfg.bframes > 0 at ks265_frame_create */
int ks265_encode_picture_b(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, ks265_pic recon_out);
/* a B picture with several reference pictures per list (round 5; cfg.refs > 1 at ks265_frame_create; -ref with B pictures: -preset veryslow = 4 / 4): refs0 = pictures BEFORE this one
 * in display order, nearest first (n0 <= cfg.refs), refs1 = pictures after it, nearest first; no picture in both lists.  One search per picture, ks265_ref_pick keeps per PU and list the
 * picture with the smallest cost + lambda x ref_idx bits (truncated unary; ties to the nearest), the bi-predictive decision pairs the lists' winners; ks265_cu8.inter_dir =
 * direction | idx0 << 4 | idx1 << 6 (an unused list's index is 0).  HOST arrays */
int ks265_encode_picture_b_mref(ks265_frame *f, ks265_pic src, const ks265_pic *refs0, int n0, const ks265_pic *refs1, int n1, ks265_pic recon_out);
/* one list's per-PU choice: dev_pu[r] = the records of the search in the list's picture r (HOST array of device pointers); out (may be dev_pu[0]) = the winner's record with the index
 * bits in its cost, dev_idx = its picture, one byte per record */
int ks265_ref_pick(ks265_frame *f, int nref, const ks265_pu *const *dev_pu, ks265_pu *dev_out, uint8_t *dev_idx);
/* in-situ stage timing: HIP events recorded on the context's stream between the stages of ks265_encode_picture;
 * ms[] = {me_integer, me_subpel, intra_candidates, cu_decide (+ merge pass), reconstruct, intra_pass, deblock, sao (+ padding)} of the last picture, -1 = not run
 * (a key picture: intra_candidates = the mode pre-selection, intra_pass = the wavefront reconstruction) */
int ks265_frame_set_profiling(ks265_frame *f, int enable);
int ks265_frame_stage_ms(ks265_frame *f, float ms[8]);
/* duration of the last me_int_kernel launch alone (the SAD kernel: stage me_integer also holds the pre-search and the propagation round); -1 = none */
int ks265_frame_me_int_ms(ks265_frame *f, float *ms);
/* accessors to the frame object's internal workspace (device pointers) */
int16_t *ks265_frame_levels(ks265_frame *f, int comp);
ks265_pu *ks265_frame_pu(ks265_frame *f);
ks265_cu8 *ks265_frame_cu8(ks265_frame *f);
uint32_t *ks265_frame_ibest(ks265_frame *f);            /* cfg.intra_inter: nctu x 85 packed intra candidates of the last P / B picture, else null */
ks265_sao_param *ks265_frame_sao(ks265_frame *f);
/* The records of the picture just coded as ONE contiguous block in HBM, so that a pipelined host needs one device-side copy and one D2H per
 * picture: off[0..5] = byte offsets of { CU map, levels Y, Cb, Cr, SAO records, 64 caller-defined bytes (e.g. the three SSE sums) }, each
 * aligned to 256, off[6] = size of the block.  ks265_frame_pack_records copies them there on the context's stream (dev_extra64 may be NULL). */
/* host-side state a P picture's launch sequence depends on (bit 0: which PU buffer it writes, bit 1: a previous P picture's vectors exist) and the
 * transition ks265_encode_picture(…, is_key = 0, …) makes - for hosts that replay a captured picture (ks265_graph_launch) instead of calling it */
int ks265_frame_p_state(ks265_frame *);
int ks265_frame_p_advance(ks265_frame *);
int ks265_frame_p_restore(ks265_frame *, int state);     /* back to a state ks265_frame_p_state returned (a capture that failed after the calls were made) */
/* A host may code key pictures on a second frame object (another context = another stream, concurrently with the P pictures of the previous GOP); the frame
 * object that continues with the P pictures must then forget its temporal predictors, as ks265_encode_picture(is_key) does itself */
int ks265_frame_reset_prediction(ks265_frame *f);
int ks265_frame_records_layout(ks265_frame *f, size_t off[7]);
int ks265_frame_pack_records(ks265_frame *f, void *dev_dst, const void *dev_extra64);
/* The same records in COMPACT form: the level planes cut into lines of 64 bytes (32 levels of a row; planes Y, Cb, Cr one after the other, each rounded up
 * to whole lines), only the lines that hold a level are stored.  off[0..6] = byte offsets of { CU map, SAO records, 64 caller bytes, header of four uint32
 * (unused, unused, data_lines, lines), chunk table (uint32 per 1024 lines: index in the data area of the chunk's first stored line), line bitmap (bit L of
 * the little-endian 64-bit words: line L is stored), data area }, off[7] = capacity of the block (data area sized for every line).  ks265_frame_pack_compact
 * builds the block in HBM on the frame's stream; ks265_copy_out_compact_async copies its fixed part and the data_lines stored lines to mapped pinned host
 * memory on ANOTHER context's stream (a 32-work-group kernel; the size is read on the device, the host learns it from the header). */
int ks265_frame_compact_layout(ks265_frame *f, size_t off[8]);
int ks265_frame_pack_compact(ks265_frame *f, void *dev_dst, const void *dev_extra64);
/* draining a picture beside the next one: pack (and the SSE) on another context's stream, and the event that ends the drain handed to the frame - the next
 * ks265_encode_picture* call waits for it right before its first kernel that writes a record (CU map, levels, SAO parameters), its search does not wait.
 * The fence is consumed by that call. */
int ks265_frame_pack_compact_on(ks265_ctx *cx, ks265_frame *f, void *dev_dst, const void *dev_extra64);
int ks265_frame_set_records_fence(ks265_frame *f, void *ev);
int ks265_copy_out_compact_async(ks265_ctx *copy_ctx, ks265_frame *f, void *pinned_host, const void *dev_block);
/* the same with the copy engine for the fixed part + the first data_bytes of the data area, the kernel for what lies beyond (0 .. all of it) */
int ks265_copy_out_compact_dma_async(ks265_ctx *, ks265_frame *f, void *pinned_host, const void *dev_block, size_t data_bytes);
/* luma SSE between two padded pictures (PSNR-Y of the bench line; CPSNR_I420::calcPSNR enc@0x4c4060) */
int ks265_sse_picture(ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *dev_sse3);
int ks265_sse_picture_on(ks265_ctx *cx, ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *dev_sse3);

#ifdef __cplusplus
}
#endif
#endif
