/* ks265_enc.h — the library boundary "B2" of SURVEY.md §8(b): the encoder API of the KSC265 SDK (qy265enc.h:196-233, qy265def.h:7-22,
 * 179-198 under /root/reference/Android_demo/prebuilt/include/), served by the MI355X pixel path + the host bitstream writer.
 *
 * A caller written against the SDK keeps including the SDK's own qy265enc.h / qy265def.h and links libks265enc.so instead of libqyencoder:
 * the entry points have the SDK's names and signatures, the structures below have the SDK's layout (field order and types; checked
 * against offsets computed from the SDK header by tests/test_enc_api.py).  This header exists so that the library, its CLI and its tests
 * compile without the SDK; its comments say what THIS implementation does with each field.
 *
 * Behaviour that differs from the SDK, all of it reported through the log callback at open:
 *   - the encoder's decisions are its own: the stream is a conforming HEVC stream, not appencoder's bytes;
 *   - rate control: rc = 0 is constant QP with the reference's own hidden ladders (read from its -psnr 2 lines): I = Q; IPPP P pictures
 *     Q + 1 + {0, 2, 1, 2}[position & 3]; hierarchical GOP anchors Q + 1, B layers Q + 2 / + 4 / + 4 (-bframes 3, a pyramid of 4: Q + 2 / + 3); P + n plain B: B = Q + 2 (-fixqp 1: one QP);
 *     rc = 3 (CRF) maps crf to that ladder; rc = 1 / 2 / 4 (bitrate targets) run a frame-level controller on top of it (one offset per
 *     mini-GOP from the pictures coded so far: deterministic streams); rc = 5 and VBV are not implemented;
 *   - subme 0 / 1 / 2 and the preset's thresholds run the reference's sub-pel refinement (include/ks265_hip.h ks265_frame_cfg.subme);
 *   - part = 1 codes 2NxN / Nx2N prediction units (CUs of 64 / 32 / 16 samples; priced with the vectors of the square search);
 *     iAqMode / fAqStrength (-aq / -aqs) give every CTU its own QP (cu_qp_delta) from the reference's calcFrameAdaptQuant arithmetic;
 *   - lookahead: with the default hierarchical GOP (bframes -1 / 7) the slice-type decision (a block of 8 pictures coded as 8 or as 4 + 4)
 *     runs by itself, so the GOP layout depends on the content unless lookahead = 0; lookahead N > 0 adds scene-cut key pictures;
 *   - refnum with the pyramid GOPs gives the B pictures up to 4 reference pictures per list (round 5); ref0 (round 6; every preset from superfast up resolves to 3) gives the
 *     anchors of a pyramid the last ref0 anchors of their GOP to search; tuInter >= 1 codes 2Nx2N inter CUs of 32 / 16 samples with four transform units where the residual sits
 *     in part of the CU (one level of the residual quadtree; deeper values run as 1);
 *   - rdoq (round 6): the presets' rdoq = 1 runs this build's own seam (dead zone, coefficient-group pruning, sign-data hiding); rdoq set BY NAME (QY265ConfigParse "rdoq" "1" =
 *     `-rdoq 1`, stored as 2) sends the luma transform blocks of inter CUs through the reference's rdoQuant with bit tables that follow the stream; "0" = the seam;
 *   - sao: every level > 0, the presets' 3 (veryfast, fast) included, = this build's rule over all four edge classes + band offset; sao 3 set BY NAME (QY265ConfigParse "sao" "3" =
 *     `-sao 3`, stored as 5) = the reference's decision on its -sao 4 path (band offset + the 0 / 90 degree edge classes, its estimation functions, rates and lambda table, no merge
 *     candidates) - measured on configs[0]: 5.5 % more bytes at equal PSNR-Y than the build's rule, which is why the presets do not select it;
 *   - transskip, tuIntra, vpp_*, 2-pass, long-term references, VBV / CVQ: accepted, ignored;
 *   - input pictures: the caller's planes are pinned in place and uploaded from where they lie inside QY265EncoderEncodeFrame; the caller may reuse its buffers when the call returns
 *     (the SDK requires them to stay valid until the frame is done).
 */
#ifndef KS265_ENC_H
#define KS265_ENC_H
#ifdef __cplusplus
extern "C" {
#endif

/* qy265def.h:7-22 */
enum { QY_OK = 0, QY_FAIL = (int)0x80000001, QY_OUTOFMEMORY = (int)0x80000002, QY_POINTER = (int)0x80000003, QY_NOTSUPPORTED = (int)0x80000004,
       QY_AUTH_INVALID = (int)0x80000005 };

typedef enum QY265Tune_tag { QY265TUNE_DEFAULT = 0, QY265TUNE_SELFSHOW, QY265TUNE_GAME, QY265TUNE_MOVIE, QY265TUNE_SCREEN } QY265Tune;
typedef enum QY265Preset_tag { QY265PRESET_ULTRAFAST = 0, QY265PRESET_SUPERFAST, QY265PRESET_VERYFAST, QY265PRESET_FAST, QY265PRESET_MEDIUM, QY265PRESET_SLOW,
                               QY265PRESET_SLOWER, QY265PRESET_VERYSLOW, QY265PRESET_PLACEBO } QY265Preset;
typedef enum QY265Latency_tag { QY265LATENCY_ZERO = 0, QY265LATENCY_LOWDELAY, QY265LATENCY_LIVESTREMING, QY265LATENCY_DEFAULT } QY265Latency;

/* qy265enc.h:51-148, same field order and types */
typedef struct QY265EncConfig {
    void *pAuth;                              /* ignored (no licence check) */
    QY265Tune tune; QY265Preset preset; QY265Latency latency;
    int profileId;                            /* 1 = Main (the only one written) */
    int bHeaderBeforeKeyframe;                /* VPS / SPS / PPS in front of every key picture */
    int picWidth, picHeight;                  /* multiples of 8 */
    double frameRate;
    int bframes;                              /* -1: preset / latency default (hierarchical GOP 8 at default latency, else 0); 0: IPPP; 3 / 7: pyramids of 4 / 8 as in the reference; other n: n non-reference B pictures per anchor */
    int temporalLayer;
    int vpp_denoise, vpp_edge, vpp_color, vpp_hdr; double vpp_hdr_strength; int vpp_hdr_iter; double vpp_hdr_sigma_s, vpp_hdr_sigma_r, vpp_recur_filter;
    int rc;                                   /* 0 CQP, 1 CBR, 2 ABR, 3 CRF, 4 CVBR, 5 CVQ */
    int bitrateInkbps, vbv_buffer_size, vbv_max_rate, vbv_min_rate;
    int qp, crf, visual_quality;
    int iIntraPeriod;                         /* key picture (IDR) period, -1 = only the first */
    int qpmin, qpmax, enFrameSkip;
    int enWavefront, enFrameParallel;         /* the GPU path is frame-wide; ignored */
    int threads;                              /* host threads writing slice data (one picture each), 0 = all cores */
    int vui_parameters_present_flag;
    struct { int video_signal_type_present_flag, video_format, video_full_range_flag, colour_description_present_flag, colour_primaries,
             transfer_characteristics, matrix_coeffs; } vui;
    int logLevel, lookahead, calcPsnr, calcSsim, shortLoadingForPlayer;
    int iPass; char statFileName[256]; double fRateTolerance;
    int rdoq, me, part, do64, tuInter, tuIntra, smooth, transskip, subme, satdInter, satdIntra, searchrange, refnum, ref0, sao, longTermRef, iAqMode;
    double fAqStrength;
    int rasl;
} QY265EncConfig;

/* qy265enc.h:160-184 */
typedef struct QY265YUV { int iWidth, iHeight; unsigned char *pData[3]; int iStride[3]; } QY265YUV;
typedef struct QY265Picture { int iSliceType; int poc; long long pts; long long dts; QY265YUV *yuv; } QY265Picture;
typedef struct QY265Nal { int naltype; int tid; int iSize; long long pts; unsigned char *pPayload; } QY265Nal;

typedef void (*QYLogPrintf)(const char *msg);
void QY265SetLogPrintf(QYLogPrintf cb);                       /* qy265def.h:188; NULL = stdout */
extern const char strLibQy265Version[];                        /* qy265def.h:198 */

void *QY265EncoderOpen(QY265EncConfig *pCfg, int *errorCode);  /* NULL + *errorCode on failure */
void QY265EncoderClose(void *pEncoder);
void QY265EncoderReconfig(void *pEncoder, QY265EncConfig *pCfg);          /* qp / bitrate / iIntraPeriod take effect at the next picture */
int QY265EncoderEncodeHeaders(void *pEncoder, QY265Nal **pNals, int *iNalCount);
/* pInpic == NULL flushes.  The NAL array and payloads belong to the encoder and stay valid until the next call.  Returns QY_OK or an error. */
int QY265EncoderEncodeFrame(void *pEncoder, QY265Nal **pNals, int *iNalCount, QY265Picture *pInpic, QY265Picture *pOutpic, int bForceLogo);
void QY265EncoderKeyFrameRequest(void *pEncoder);
int QY265EncoderDelayedFrames(void *pEncoder);
int QY265ConfigDefault(QY265EncConfig *pConfig, QY265Preset preset, QY265Tune tune, QY265Latency latency);
int QY265ConfigDefaultPreset(QY265EncConfig *pConfig, char *preset, char *tune, char *latency);
#define QY265_PARAM_BAD_NAME (-1)
#define QY265_PARAM_BAD_VALUE (-2)
int QY265ConfigParse(QY265EncConfig *p, const char *name, const char *value);

/* not in the SDK: totals of the session for the CLI's summary lines */
typedef struct { long frames; long long bytes; double sse[3]; double gpu_ms; double host_write_ms;
                 double in_copy_ms, submit_ms, output_ms;   /* calling thread: input copy to pinned memory, enqueueing GPU work, waiting for / copying output */
                 double lat_gpu_ms, lat_queue_ms;           /* summed per picture: enqueue -> records on the host; enqueue -> a writer thread picked the picture up */
                 double key_wall_ms, key_cpu_ms; long keys; /* key pictures: records on the host -> slice finished (wall), summed thread time of its rows */
                 long occ_samples, occ_ring, occ_gpu, occ_ready;   /* sampled at every submission: pictures in the ring, of them not yet through the GPU, of them waiting for a writer */
                 double submit_wait_ms;                     /* the part of submit_ms the scheduler thread spent WAITING for a free ring slot (not runtime calls) */
} ks265_enc_stats;
int ks265_enc_get_stats(void *pEncoder, ks265_enc_stats *out);
/* extension: closed GOPs coded concurrently by this handle ("GOP lanes": KS265_GOP_LANES = 2..4 with enFrameParallel, -rc 0, key period >= 32, any GOP structure;
 * default: 2 for the pyramid GOPs - the SDK's default GOP and -bframes 3 - on one GPU with -rc 0 / -rc 3, where lanes leave the stream as it is (round 5: their B pictures leave the device under-filled, two closed GOPs side by side
 * code 700 pictures/s where one codes 631 at 2160p), 1 otherwise; KS265_GOP_LANES=1 switches it off).  Output stays in stream order and is byte for byte the one-lane stream;
 * it lags the input by up to that many GOPs, and every lane buffers a GOP of input (pinned host memory + a device twin per picture, capped by KS265_PINNED_MB per lane). */
int ks265_enc_lanes(void *pEncoder);
/* extension: the switches of the reference CLI that QY265EncConfig has no field for - "df" (deblocking, default 1), "fixqp" (1 = no per-layer QP offsets: every
 * picture at -qp), "md5" (1 = log `POC n MD5 y,u,v` of every reconstructed picture, display order).  Process-wide defaults read by the next QY265EncoderOpen. */
int ks265_enc_set_default(const char *name, int value);
/* extension: zero-copy input.  Fills `yuv` with the planes of one of the encoder's pinned input buffers (packed I420, strides = width, width / 2); the caller writes the
 * next picture there and passes the same QY265YUV to QY265EncoderEncodeFrame, which then copies nothing (0.35 ms of the calling thread per 2160p picture otherwise).
 * Never blocks: QY_FAIL when no buffer is free at the moment - pass your own buffer then, it is copied as usual.  At most one buffer is out at a time.
 * CONTRACT: the buffer belongs to the caller only until the NEXT QY265EncoderEncodeFrame call on this handle, whatever picture that call hands in - if it hands in another
 * buffer and no free input slot is left, the encoder copies that picture into the acquired buffer (its last resort; anything the caller had written there is lost).  Acquire,
 * fill and hand in the same buffer, one picture at a time.  With the lookahead every input slot also has a twin in device memory (picture size each; logged at open). */
int ks265_enc_acquire_input(void *pEncoder, QY265YUV *yuv);
/* extension: write the reconstruction (I420, display order) to `path` - the reference CLI's `-o`; call between Open and the first picture */
int ks265_enc_set_recon_file(void *pEncoder, const char *path);

#ifdef __cplusplus
}
#endif
#endif
