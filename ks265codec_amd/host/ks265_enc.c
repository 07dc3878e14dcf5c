/* ks265_enc.c — the encoder behind the SDK's library API (include/ks265_enc.h; SURVEY.md §8(b) "B2", caller sequence §3.4):
 * QY265ConfigDefault* -> QY265EncoderOpen -> QY265EncoderEncodeFrame ... -> flush -> QY265EncoderClose.
 *
 * Plain C host over two C ABIs: libks265hip.so (the pixel path on the MI355X, include/ks265_hip.h) and the bitstream writer
 * (ks265_stream.c).  What the reference's host does per frame (CHevcEncode::encodeFrame enc@0x4b9930, §3.2) maps to:
 *   input picture -> pinned host copy -> H2D (copy-in stream) -> ks265_encode_picture[_b/_mref] (all pixel stages on the compute stream)
 *   -> records to a staging set (D2D) -> D2H of the CU map / levels / SAO records (copy-out stream, pinned) -> event; the three streams hand
 *      over by events, so the copies of neighbouring pictures run under the kernels of the current one
 *   -> a host thread waits for the event and writes the slice NAL (CABAC), one picture per thread, pictures of a GOP in parallel
 *   -> NAL units are handed out in coding order (the SDK's asynchronous contract: output lags input).
 * GOP structures: IPPP (optionally several list-0 pictures), anchor + n non-reference B, hierarchical-B mini-GOPs of 8 (the SDK's
 * default at default latency); every key picture is an IDR (closed GOPs: what GOP-sharding over GPUs needs).
 */
#define _GNU_SOURCE
#include "ks265_enc.h"
#include "ks265_stream.h"
#include <dirent.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

const char strLibQy265Version[] = "ks265enc 0.2 (MI355X pixel path + host CABAC; API of libqycodec V2.6.1.3)";

static QYLogPrintf g_log_cb;
void QY265SetLogPrintf(QYLogPrintf cb) { g_log_cb = cb; }
static void logf_(int level, int min_level, const char *fmt, ...)
{
    if (level < min_level) return;
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (g_log_cb) g_log_cb(buf); else { fputs(buf, stdout); fflush(stdout); }
}

/* ---- MD5 (RFC 1321) of the reconstructed planes: the reference CLI's `-md5 1` lines `POC n MD5 y,u,v` (README.md:61-66 conventions) */
static void md5_block(uint32_t st[4], const uint8_t *p)
{
    static const uint8_t sh[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                   4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    static const uint32_t K[64] = {   /* floor(2^32 |sin(i + 1)|): a constant table (it used to be filled lazily by whichever writer thread came first) */
        0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u, 0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u, 0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u, 0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au, 0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u, 0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u, 0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u, 0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};
    uint32_t w[16], a = st[0], b = st[1], c = st[2], d = st[3];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
    for (int i = 0; i < 64; ++i) {
        uint32_t f; int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        const uint32_t t = d; d = c; c = b;
        const uint32_t x = a + f + K[i] + w[g];
        b = b + ((x << sh[i]) | (x >> (32 - sh[i]))); a = t;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}
static void md5_hex(const uint8_t *data, size_t n, char out[33])
{
    uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    size_t i = 0;
    for (; i + 64 <= n; i += 64) md5_block(st, data + i);
    uint8_t tail[128]; size_t r = n - i;
    memcpy(tail, data + i, r); tail[r++] = 0x80;
    const size_t padded = r <= 56 ? 64 : 128;
    memset(tail + r, 0, padded - r);
    const uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; ++k) tail[padded - 8 + k] = (uint8_t)(bits >> (8 * k));
    md5_block(st, tail); if (padded == 128) md5_block(st, tail + 64);
    for (int k = 0; k < 16; ++k) snprintf(out + 2 * k, 3, "%02x", (unsigned)((st[k >> 2] >> (8 * (k & 3))) & 255u));
}

/* CLI-level switches of `appencoder` that the SDK's QY265EncConfig has no field for (-df, -fixqp, -md5): process-wide defaults a front end
 * sets before QY265EncoderOpen (ks265_enc_set_default) */
static struct { int df, fixqp, md5, scenecut, cutree; } g_cli = {1, 0, 0, 0, 1};
int ks265_enc_set_default(const char *name, int value)
{
    if (!name) return QY_POINTER;
    if (!strcmp(name, "df")) { if (value < 0 || value > 1) return QY265_PARAM_BAD_VALUE; g_cli.df = value; return QY_OK; }
    if (!strcmp(name, "fixqp")) { if (value < 0 || value > 1) return QY265_PARAM_BAD_VALUE; g_cli.fixqp = value; return QY_OK; }
    if (!strcmp(name, "md5")) { if (value < 0 || value > 1) return QY265_PARAM_BAD_VALUE; g_cli.md5 = value; return QY_OK; }
    if (!strcmp(name, "scenecut")) { if (value < 0 || value > 100) return QY265_PARAM_BAD_VALUE; g_cli.scenecut = value; return QY_OK; }   /* the reference's hidden -scenecut N */
    if (!strcmp(name, "cutree")) { if (value < 0 || value > 1) return QY265_PARAM_BAD_VALUE; g_cli.cutree = value; return QY_OK; }        /* the reference's hidden -cutree N (on by default, as there) */
    return QY265_PARAM_BAD_NAME;
}

/* motion lambda in Q4 per QP: round(16 * sqrt(0.57 * 2^((qp - 12) / 3))) - the host-side float setup of the pixel path, as a table so that
 * every host (this one, the Python test mirror ks265codec_amd/synth.py) uses identical integers */
static const int kLambdaQ4[52] = {3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 17, 19, 22, 24, 27, 30, 34, 38, 43, 48, 54, 61, 68, 77, 86, 97, 108, 122, 137, 153, 172,
                                  193, 217, 244, 273, 307, 344, 387, 434, 487, 547, 614, 689, 773, 868, 974, 1093};

/* P / B pictures: lambda_mode carries HM's factor for pictures that are not key pictures, clip(2, 4, (qp - 12) / 6) - round(16 * sqrt(0.57 * f * 2^((qp - 12) / 3))).
 * The reference's own integer motion lambda (read from its mvd cost tables, tests/test_me_search.py: 8 at qp 27, 9 at 28, 16 at 33, 18 at 34) is within 10 percent of it.
 * It prices vectors, CU splits, merge, SAO and the coefficient-group pruning (ks265_frame_cfg.rdo) of those pictures; measured effect: DESIGN.md */
static const int kLambdaInterQ4[52] = {4, 5, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 17, 19, 22, 24, 27, 30, 34, 38, 43, 48, 54, 61, 68, 80, 93, 108, 125, 145, 167, 193, 222, 256, 294, 337, 387, 434, 487, 547, 614, 689, 773, 868, 974, 1093, 1227, 1378, 1546, 1736, 1948, 2187};

/* ------------------------------------------------------------------ configuration (QY265ConfigDefault enc@0x4b7020 lineage: fillCfgs<Preset>) */
static const char *const kPresetNames[] = {"ultrafast", "superfast", "veryfast", "fast", "medium", "slow", "slower", "veryslow", "placebo", 0};
static const char *const kTuneNames[] = {"default", "selfshow", "game", "movie", "screen", 0};
static const char *const kLatencyNames[] = {"zerolatency", "lowdelay", "livestreaming", "default", 0};
/* tools per preset as the reference resolves them (measured from the CLI's config echo) */
static const struct { int me, subme, ref, ref0, part, tuinter, rdoq, sao; } kPresetTools[9] = {
    {1, 1, 1, 2, 0, 0, 0, 1}, {1, 1, 1, 3, 0, 0, 0, 1}, {1, 1, 1, 3, 0, 0, 0, 3}, {1, 1, 1, 3, 0, 0, 1, 3}, {1, 1, 1, 3, 0, 0, 1, 4},
    {2, 1, 1, 3, 0, 0, 1, 4}, {2, 1, 2, 4, 1, 0, 1, 4}, {2, 2, 4, 4, 1, 1, 1, 4}, {2, 2, 5, 5, 1, 2, 1, 4}};

/* what the reference's presets put into the configuration words its sub-pel refinement reads (cfg+0x464 = tME+0x3c0, tME+0x3c4, cfg+0x498, cfg+0x49c, cfg+0x580,
 * tME+0x64): measured inside real appencoder runs (oracle/ref_probe/subme_shim.c).  The public QY265EncConfig has no fields for them: they follow the preset.
 * ks265codec_amd/synth.py SUBME_PRESET holds the same table for the tests. */
static const struct { int thr, flat, cap, cap_step, diag_fast, satd; } kPresetSubme[9] = {
    {80, 40, 6, 6, 1, 0}, {76, 36, 6, 6, 1, 0}, {68, 16, 12, 6, 0, 0}, {56, 14, 0, 0, 0, 0}, {40, 10, 0, 0, 0, 0},
    {24, 8, 0, 0, 0, 0}, {24, 8, 0, 0, 0, 0}, {0, 8, 0, 0, 0, 1}, {0, 8, 0, 0, 0, 1}};

int QY265ConfigDefault(QY265EncConfig *c, QY265Preset preset, QY265Tune tune, QY265Latency latency)
{
    if (!c) return QY_POINTER;
    if ((int)preset < 0 || (int)preset > 8 || (int)tune < 0 || (int)tune > 4 || (int)latency < 0 || (int)latency > 3) return QY_NOTSUPPORTED;
    memset(c, 0, sizeof *c);
    c->tune = tune; c->preset = preset; c->latency = latency;
    c->profileId = 1; c->bHeaderBeforeKeyframe = 1; c->frameRate = 25.0;
    c->bframes = -1; c->rc = 2; c->bitrateInkbps = 1000; c->qp = 26; c->crf = 24; c->visual_quality = 95; c->iIntraPeriod = 256;
    c->qpmin = 0; c->qpmax = 51; c->enWavefront = 1; c->enFrameParallel = 1; c->threads = 0;
    c->logLevel = 0; c->lookahead = -1; c->fRateTolerance = 2.0;
    c->rdoq = kPresetTools[preset].rdoq; c->me = kPresetTools[preset].me; c->part = kPresetTools[preset].part; c->do64 = 1;
    c->tuInter = kPresetTools[preset].tuinter; c->tuIntra = kPresetTools[preset].tuinter; c->smooth = 1; c->transskip = 0;
    c->subme = kPresetTools[preset].subme; c->satdInter = preset >= QY265PRESET_SLOW; c->satdIntra = preset >= QY265PRESET_SLOW;
    c->searchrange = 64; c->refnum = kPresetTools[preset].ref; c->ref0 = kPresetTools[preset].ref0; c->sao = kPresetTools[preset].sao;
    c->iAqMode = 0; c->fAqStrength = 1.0; c->rasl = 1;
    return QY_OK;
}
static int find_name(const char *const *names, const char *s)
{
    if (!s) return -2;
    for (int i = 0; names[i]; ++i) if (!strcmp(names[i], s)) return i;
    return -1;
}
int QY265ConfigDefaultPreset(QY265EncConfig *c, char *preset, char *tune, char *latency)
{
    int p = find_name(kPresetNames, preset), t = find_name(kTuneNames, tune), l = find_name(kLatencyNames, latency);
    if (p == -2) p = QY265PRESET_MEDIUM;
    if (t == -2) t = QY265TUNE_DEFAULT;
    if (l == -2) l = QY265LATENCY_DEFAULT;
    if (p < 0 || t < 0 || l < 0) return QY_NOTSUPPORTED;
    return QY265ConfigDefault(c, (QY265Preset)p, (QY265Tune)t, (QY265Latency)l);
}
int QY265ConfigParse(QY265EncConfig *c, const char *name, const char *value)
{
    if (!c || !name || !value) return QY265_PARAM_BAD_NAME;
    char *end = NULL;
    const double dv = strtod(value, &end);
    const int num_ok = end && end != value && *end == 0, iv = (int)dv;
#define INTP(NAME, FIELD, LO, HI) if (!strcmp(name, NAME)) { if (!num_ok || iv < (LO) || iv > (HI)) return QY265_PARAM_BAD_VALUE; c->FIELD = iv; return 0; }
    INTP("wdt", picWidth, 8, 8192) INTP("hgt", picHeight, 8, 8192) INTP("bframes", bframes, -1, 15) INTP("rc", rc, 0, 5) INTP("br", bitrateInkbps, 1, 1000000)
    INTP("qp", qp, 0, 51) INTP("crf", crf, 0, 51) INTP("iper", iIntraPeriod, -1, 100000) INTP("qpmin", qpmin, 0, 51) INTP("qpmax", qpmax, 0, 51)
    INTP("threads", threads, 0, 1024) INTP("psnr", calcPsnr, 0, 2) INTP("ssim", calcSsim, 0, 2) INTP("log", logLevel, -1, 3) INTP("lookahead", lookahead, -1, 250)
    /* rdoq (qy265enc.h:129): every preset from `fast` up resolves to 1, which this build runs as its own seam (dead zone, coefficient-group pruning, sign-data hiding); asked for BY
     * NAME, rdoq 1 is stored as 2 = the reference's rdoQuant between the two halves of the reconstruction (ks265_frame_set_rdoq), so that the presets' streams stay what they are */
    if (!strcmp(name, "rdoq")) { if (!num_ok || iv < 0 || iv > 1) return QY265_PARAM_BAD_VALUE; c->rdoq = iv ? 2 : 0; return 0; }
    INTP("me", me, 0, 4) INTP("part", part, 0, 1) INTP("do64", do64, 0, 1) INTP("intertu", tuInter, -1, 3) INTP("intratu", tuIntra, -1, 3)
    INTP("sis", smooth, 0, 1) INTP("ts", transskip, 0, 1) INTP("subme", subme, 0, 2) INTP("merange", searchrange, 1, 512) INTP("ref", refnum, 1, 16) INTP("ref0", ref0, 1, 16)
    /* sao (qy265enc.h:143): veryfast and fast resolve to 3, which - like every level > 0 - runs this build's rule; asked for BY NAME, sao 3 is stored as 5 = the reference's own
     * decision restated (ks265_frame_cfg.sao = 2), so that the presets' streams do not pay for it (configs[0] on the MI355X: 5.5 % fewer bytes at equal PSNR-Y with the build's rule) */
    if (!strcmp(name, "sao")) { if (!num_ok || iv < 0 || iv > 4) return QY265_PARAM_BAD_VALUE; c->sao = iv == 3 ? 5 : iv; return 0; }
    INTP("wpp", enWavefront, 0, 1) INTP("fpp", enFrameParallel, 0, 1) INTP("vbv-maxrate", vbv_max_rate, 0, 10000000)
    INTP("aq", iAqMode, 0, 3)                                  /* the reference's hidden -aq (iAqMode, qy265enc.h:145) */
    INTP("vbv-bufsize", vbv_buffer_size, 0, 10000000) INTP("pass", iPass, 0, 2) INTP("tlayer", temporalLayer, 0, 1) INTP("frameskip", enFrameSkip, 0, 1)
#undef INTP
    if (!strcmp(name, "fr")) { if (!num_ok || dv <= 0 || dv > 1000) return QY265_PARAM_BAD_VALUE; c->frameRate = dv; return 0; }
    if (!strcmp(name, "aqs")) { if (!num_ok || dv < 0 || dv > 3.0) return QY265_PARAM_BAD_VALUE; c->fAqStrength = dv; return 0; }   /* -aqs (fAqStrength, qy265enc.h:146) */
    if (!strcmp(name, "ratetol")) { if (!num_ok || dv < 0) return QY265_PARAM_BAD_VALUE; c->fRateTolerance = dv; return 0; }
    if (!strcmp(name, "preset") || !strcmp(name, "latency") || !strcmp(name, "tune")) {
        const int k = find_name(name[0] == 'p' ? kPresetNames : name[0] == 'l' ? kLatencyNames : kTuneNames, value);
        if (k < 0) return QY265_PARAM_BAD_VALUE;
        if (name[0] == 'p') c->preset = (QY265Preset)k; else if (name[0] == 'l') c->latency = (QY265Latency)k; else c->tune = (QY265Tune)k;
        return 0;
    }
    return QY265_PARAM_BAD_NAME;
}

/* ------------------------------------------------------------------ encoder */
#define MAX_DPB 18
#define MAX_JOBS 128                                      /* upper bound of the ring of pictures in flight; the encoder sizes its ring (Enc::ring) by picture size */
typedef struct TopWake { pthread_mutex_t mu; pthread_cond_t cv; unsigned long seq; } TopWake;
#undef MAX_INPUT                                         /* (<dirent.h> brings the terminal limit of that name along) */
#define MAX_INPUT (MAX_JOBS + 32 + 256)                /* input slots: the ring + a mini-GOP (+ in a GOP lane: a whole GOP of the next round, Enc::nin) */

typedef struct Job {
    int used, done, error;
    int disp, poc, kind, qp, nal_type, is_ref;            /* kind: 'I' 'P' 'B' */
    int no_sao;                                           /* the picture was coded without SAO (lean B pictures): slice_sao_luma_flag = slice_sao_chroma_flag = 0 */
    long long pts;
    int nl0, nl1, l0[4], l1[4], nrps, rps_poc[16]; unsigned char rps_used[16];
    uint8_t *cmp;                                         /* pinned host copy of the GPU's records in compact form (ks265_frame_compact_layout) */
    uint8_t *lvlbuf; uint64_t *dirty;                     /* the level planes expanded from it (plain memory) and the lines the previous picture in this slot set */
    ks265_cu8 *cu8; int16_t *lvl[3]; ks265_sao_param *sao; uint64_t *sse;      /* cu8 / sao / sse point into cmp, lvl into lvlbuf */
    int ev_err;
    uint8_t fctx[192]; int fctx_ok; int32_t *rq_host;     /* -rdoq 1: the context states the slice ended with (ks265_slice_final_contexts); the tables + lambdas the picture was quantised with (pinned) */
    long sub_seq;                                         /* this picture's number in submission order (the dispatcher's sticky device error ends at a key picture submitted after the error was seen) */
    void *wpp; ks265_slice_in sin; int started, nrows, next_row, rows_done;   /* row-wise writing of the slice (ks265_wpp_*) */
    int8_t *qp_map;                                       /* -aq: the QP of every CTU this picture was coded with (pinned; NULL without) */
    uint8_t *recon;                                       /* pinned I420 copy of the reconstruction (only with ks265_enc_set_recon_file / -md5) */
    char md5[3][33];
    void *ev;                                             /* recorded after the D2H copies */
    uint8_t *nal; size_t nal_cap; long nal_len;
    int key_headers;                                      /* parameter sets go in front of this picture */
    int rc_delta;                                         /* the controller's QP offset this picture was coded with (rate control) */
    double rc_budget;                                     /* this picture's share of the bit budget: the bitrate in force when it was handed in / frame rate (QY265EncoderReconfig) */
    double t_write_ms, t_submit, t_event, t_taken, t_done;      /* wall-clock marks: enqueued, records on the host (seen by a writer), writer started */
} Job;

/* The caller's picture has to be copied (the API gives the buffer back at once): 12 MB at 2160p, ~0.9 ms for one thread - more than the GPU needs for the
 * picture.  A few helper threads share the copy with the calling thread. */
#define COPY_HELPERS 3
typedef struct CopyPool {
    pthread_t th[COPY_HELPERS]; int nth, quit;
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_done;
    struct { uint8_t *d; const uint8_t *s; size_t n; } task[COPY_HELPERS];
    unsigned posted, pending;                             /* bit k: task k waits for helper k / is not finished */
} CopyPool;
static void *copy_helper(void *arg)
{
    CopyPool *p = (CopyPool *)((void **)arg)[0]; const int k = (int)(intptr_t)((void **)arg)[1];
    free(arg);
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->quit && !(p->posted & (1u << k))) pthread_cond_wait(&p->cv_work, &p->mu);
        if (p->quit) break;
        p->posted &= ~(1u << k);
        pthread_mutex_unlock(&p->mu);
        memcpy(p->task[k].d, p->task[k].s, p->task[k].n);
        pthread_mutex_lock(&p->mu);
        p->pending &= ~(1u << k);
        if (!p->pending) pthread_cond_signal(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}
static CopyPool *copy_pool_create(void)
{
    CopyPool *p = (CopyPool *)calloc(1, sizeof *p);
    if (!p) return NULL;
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_work, NULL); pthread_cond_init(&p->cv_done, NULL);
    for (int k = 0; k < COPY_HELPERS; ++k) {
        void **a = (void **)malloc(2 * sizeof *a);
        if (!a) break;
        a[0] = p; a[1] = (void *)(intptr_t)k;
        if (pthread_create(&p->th[k], NULL, copy_helper, a)) { free(a); break; }
        ++p->nth;
    }
    return p;
}
static void copy_pool_destroy(CopyPool *p)
{
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->cv_work); pthread_mutex_unlock(&p->mu);
    for (int k = 0; k < p->nth; ++k) pthread_join(p->th[k], NULL);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_work); pthread_cond_destroy(&p->cv_done);
    free(p);
}
/* memcpy shared between the calling thread and the helpers (one caller at a time: the API's calling thread) */
static void copy_shared(CopyPool *p, uint8_t *d, const uint8_t *s, size_t n)
{
    if (!p || !p->nth || n < ((size_t)1 << 20)) { memcpy(d, s, n); return; }
    const size_t part = (n / (size_t)(p->nth + 1) + 4095) & ~(size_t)4095;
    size_t off = part;                                    /* the caller takes the first part */
    pthread_mutex_lock(&p->mu);
    for (int k = 0; k < p->nth && off < n; ++k, off += part) {
        p->task[k].d = d + off; p->task[k].s = s + off; p->task[k].n = n - off < part ? n - off : part;
        p->posted |= 1u << k; p->pending |= 1u << k;
    }
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    memcpy(d, s, part < n ? part : n);
    pthread_mutex_lock(&p->mu);
    while (p->pending) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

#define LA_RING 9                                                      /* half-size pictures kept for the analysis: the current one and eight back */
#define LA_QMAX 64                                                     /* pictures that may wait at the input for their analysis (a GOP lane runs a whole GOP ahead of its pixel path) */
#define LA_FLY 16                                                      /* analyses in flight = result areas */
typedef struct Input { int used, disp, key, base_qp, iper, mini4, kbps, la_what, la_p, la_buf, up_sync; long long pts; uint8_t *i420; uint8_t *dev; void *ev_up; } Input;   /* up_sync: the twin was filled by a host-synchronous copy (nothing to wait for) */   /* dev / ev_up: the slot's twin on the device, uploaded
                                                                                                 * when the picture is handed in (round 4), and the event behind that upload */     /* pinned; key: this picture starts a closed GOP whatever the period says (GOP lanes,
                                                                                                 * QY265EncoderKeyFrameRequest); base_qp: the QP in force when the picture was handed in (QY265EncoderReconfig) -
                                                                                                 * iper: the key period in force then - all three travel WITH the picture: the scheduler thread
                                                                                                 * may be several pictures behind the caller */

typedef struct Enc {
    QY265EncConfig cfg;
    int W, H, log_level;
    CopyPool *pool;                                       /* shared by the lanes of one handle (owned by it) */
    int me_method, hex_thr, subme, refs, use_sao, use_df, gop_b, hier;                  /* resolved tools */
    int lean_b;                                                                         /* B pictures nothing predicts from: no intra candidates, no joint refinement, no SAO (KS265_LEAN_B=0: as the others) */
    int refs0, anc_hist[4], n_anc;                                                      /* -ref0 (qy265enc.h:142, the reference's ActiveRefNumFrm0InGop): how many past anchors an anchor of the pyramid searches; the last anchors' POCs, nearest first */
    int base_qp, iper, nthreads;
    ks265_ctx *ctx; ks265_frame *frame; ks265_frame_geom geom; ks265_frame_cfg fcfg; ks265_stream_cfg scfg;
    /* device: three streams - copy-in (ctx_in), the pixel path (ctx), copy-out (ctx_out) - so that the H2D of picture n+1 and the D2H of
     * picture n-1 run under the kernels of picture n.  Hand-over by events only (no host thread blocks): input buffers and output staging
     * sets rotate over NPIPE slots. */
#define NPIPE 3
    ks265_ctx *ctx_in, *ctx_out;
    uint8_t *dev_in[NPIPE]; void *ev_h2d[NPIPE], *ev_loaded[NPIPE];
    void *spacer[64]; int nspacer;
    /* round 6: the caller's planes are pinned in place (ks265_host_register, cached by address) and uploaded by DMA straight into the slot's device twin - no copy into pinned memory of
     * the encoder's own (KS265_INPUT_COPY=1: the copying path; pictures under 1 MB always copy).  The upload is waited for before QY265EncoderEncodeFrame returns - the SDK's own callers
     * refill ONE buffer for every picture (encoderwrapper.c:367-379) - unless KS265_INPUT_HOLD=1 (the caller keeps every buffer until its picture has come out, qy265enc.h:153-156) */
    int direct_in, hold_in; ks265_ctx *ctx_upl; struct Input *pending_up;
    ks265_ctx *ctx_up;                                    /* with the lookahead: = ctx_la, the stream the caller's thread feeds with uploads (the moment a picture is handed in) and with the analysis; else NULL */
    uint8_t *stg[NPIPE]; size_t cmp_off[8];               /* staging blocks of the (compact) records on the device and their layout */
    void *ev_staged[NPIPE], *ev_drained[NPIPE];
    /* split pipeline (default): the source picture of slot k is unpacked and padded on the copy-in stream into srcq[k], and the picture's drain (SSE, packing of the
     * records) runs on the copy-out stream behind ev_coded[k]; the next picture's search does not wait for either - it waits for ev_packed[k] only where it first writes
     * a record (ks265_frame_set_records_fence).  Measured at 2160p IPPP: 120 us of a 1.11 ms picture period leave the critical path. */
    /* -aq N (iAqMode != 0): adaptive quantisation = the reference's calcFrameAdaptQuant enc@0x4653c0 on the source picture (ks265_frame_adapt_quant, pinned on recorded
     * calls), one QP per CTU from it (ks265_aq_ctu_map), the pixel path and the writer on that map (ks265_frame_set_qp_map, cu_qp_delta) */
    int aq_on, aq_nx, aq_ny; double *aq_off[2], *aq_scratch[2]; uint16_t *aq_inv[2]; int8_t *dev_qmap[NPIPE], *dev_qmap_key[2];   /* aq_*[1], dev_qmap_key: the key pictures' stream (its own maps, rotating with nkeys: a key picture runs beside the pictures of its rotation slot) */
    int zero_latency;                                     /* -latency zerolatency without B pictures / lookahead / lanes: every EncodeFrame call hands out its own picture */
    int copy_mb;                                          /* KS265_COPYOUT_MB = N: hipMemcpyAsync takes the fixed part + N MB of stored lines per P / B picture (a key picture: everything) and the copy kernel
                                                           * only what lies beyond; default -1 = the copy kernel alone.  On this runtime the D2H hipMemcpyAsync is itself a kernel (__amd_rocclr_copyBuffer,
                                                           * 110 us for 4 MB): no better neighbour than ours (50 us for the ~2 MB a picture really holds) - measured both ways, within +- 1.5 % */
    int split; ks265_pic srcq[NPIPE]; void *ev_coded[NPIPE], *ev_packed[NPIPE];
    long seq;                                             /* pictures submitted */
    /* single-reference P pictures and B pictures as graphs: the launch sequence of a picture (unpack, the pixel path, SSE, packing of the records: ~20 launches) only
     * depends on a few rotating device pointers, the QP and two bits of frame state; each combination is captured once and replayed with one runtime call */
#define MAX_GRAPHS 256
    int use_graph, ngraph; struct { uint64_t key[9]; void *exec; } graph[MAX_GRAPHS];
    int recon_fd, recon_on; uint8_t *dev_recon;           /* reconstruction dump (the CLI's -o) and / or plane MD5s (-md5 1): recon_on = the pictures come back to the host */
    int md5, fixqp, psnr_hdr; int md5_next; char (*md5_ring)[3][33]; unsigned char *md5_have;   /* MD5 lines go out in display order (ring of 256 by display index) */
    ks265_pic src; ks265_pic dpb[MAX_DPB]; int dpb_poc[MAX_DPB]; int ndpb; uint64_t *dev_sse;
    /* key pictures on their own stream and frame object: an intra picture keeps 34 of 256 compute units busy for ~26 ms (2160p); coded as soon as its
     * input arrives - the pixel path is tens of pictures behind the input - it runs underneath the P pictures of the previous GOP instead of between
     * two GOPs.  Its reconstruction goes to one of two DPB slots of its own; the first P picture of the GOP waits for ev_key. */
    int refs_b;                                           /* reference pictures per list of a pyramid's B pictures (1 .. 4) */
    /* round 5 - the ANCHOR LANE of the pyramid GOPs: a mini-GOP's anchor P picture predicts from the anchor before it and from nothing else, so the chain of anchors is coded on a
     * stream and frame object of its own (like the key pictures), ahead of the B pictures of the mini-GOPs behind it: its kernels - and the long intra chain pass of a picture
     * 4 or 8 away from its reference - run beside the B pictures' instead of in front of them.  Four DPB slots of its own, rotating; a slot is overwritten only after the B
     * pictures that read its previous content have run on the main stream (ev_mg: one mark per mini-GOP, recorded behind its last B picture) */
    int anc_on, nanch, last_on_anc; ks265_ctx *ctx_anc; ks265_frame *frame_anc; ks265_pic src_anc; uint64_t *dev_sse_anc; void *ev_anc, *ev_mg[8];
    int key_overlap, nkeys; ks265_ctx *ctx_key; ks265_frame *frame_key; ks265_pic src_key; uint64_t *dev_sse_key; void *ev_key, *ev_firstp[2];
    /* scheduling */
    Input in[MAX_INPUT]; int nin, next_disp, in_disp;          /* next_disp: pictures handed to the scheduler; in_disp: pictures taken in (= next_disp + the lookahead's queue la_q) */
    int gop_start;                                        /* display index of the last key picture */
    int coded_upto;                                       /* display index up to which everything is scheduled */
    int force_key;
    int gop_end, gop_end_seen;                            /* GOP lanes: display index of the last picture of a GOP that ended early (-1: none); what the scheduler has acted on */
    int multi;                                            /* one of several GOP lanes of a handle */
    /* scene-cut lookahead (-lookahead N > 0; SURVEY.md 8(f) rank 2, scenecut enc@0x47e9d0 lineage): a stream and a frame object of HALF the size of their own; every
     * input picture is compared with its predecessor before the scheduler sees it (ks265_lookahead_picture: per 8x8 block of the half-size picture the intra
     * pre-selection cost against the integer-search cost) - where prediction from the previous picture is not clearly cheaper than intra coding a closed GOP starts */
    int la_on, la_have_prev, la_last_key, la_w, la_h; long la_cuts, la_mini4;
    int mg_adapt, mg4_until;                                          /* slice-type decision (-lookahead N with the hierarchical GOP): a block of 8 pictures is coded as 8 or as 4 + 4; display index up to which 4 is in force */
    long long la_prev_icost;                                           /* -scenecut N: the previous picture's intra cost (-1: none yet) */
    unsigned long long la_c4_prev;                                     /* inter cost of the previous picture on the GOP's grid of 4 against the picture 4 back */
    int la_auto;                                                       /* no -lookahead given, hierarchical GOP: the slice-type decision alone (pictures on the GOP's grid of 4), no scene cuts - works in GOP lanes */
    double la_t_bp, la_t_take, la_t_wait; long la_n_wait, la_n_poll;       /* where the caller's time goes (log level 1) */
    pthread_mutex_t la_mu;                                             /* the queue below and the decisions' state: the caller's thread fills it, the caller's and the scheduler's threads empty it (lock order: la_mu, then mu) */
    struct Input *la_q[LA_QMAX]; int la_qn, la_flying, la_seq, la_keep; void *la_evs[LA_FLY];   /* pictures handed in and not yet with the scheduler (display order); analyses in flight; their events / result areas, round robin */
    ks265_ctx *ctx_la; ks265_frame *frame_la; ks265_frame_geom geom_la; ks265_pic la_pic[LA_RING];
    uint32_t *la_cost_ws; uint64_t *la_dev_out, *la_host_out;
    /* cuTree (-rc 3 with -cutree 1, the default; round 6): the reference's macroblock-tree over the lookahead window.  Before a key picture / a mini-GOP is
     * submitted the scheduler thread runs, on a stream of its own, the reference's calcFrameCost enc@0x4a7410 (ks265_calc_frame_cost) for every picture of the window against the
     * pictures it will be predicted from, cuTreePropagate enc@0x47d460 from the window's end back to the mini-GOP (reverse coding order, as CInputPicManage::updateQueue does), and the
     * finish (enc@0x480964: offset = AQ offset - 1.8 log2((propagated + intra') / intra')) for the pictures about to be coded; their QP per CTU = picture QP + the mean of the CTU's
     * block offsets (ks265_qoff_ctu_map), through the -aq plumbing (ks265_frame_set_qp_map, cu_qp_delta).  Costs are cached per picture and reference pair. */
#define CT_RING 128
    int ct_on, ct_lg, ct_nx, ct_ny, ct_w, ct_h, ct_stride, ct_pad, ct_depth, ct_preset, qmap_on, qmap_fd;
    struct CtPic { int disp, p0, p1, intra_done, used; uint8_t *blk, *low; uint16_t *intra, *invq, *prop, *inter; uint8_t *imode, *bits; int32_t *mv0, *c0, *mv1, *c1; double *aq_off, *qoff; ks265_cfc_sums *sums; void *ev_used; } ct[CT_RING];
    ks265_ctx *ctx_ct; void *ct_ws, *ct_ev; uint64_t *ct_acc; double *ct_scratch; uint8_t *ct_full;
    struct TopWake *wake;                                 /* lanes: the handle's caller sleeps here until a picture of ANY lane is finished */
    Job jobs[MAX_JOBS]; int ring, job_head, job_tail, njobs;   /* ring of `ring` pictures in coding order */
    /* workers */
    pthread_t th[64]; int nth; pthread_mutex_t mu; pthread_cond_t cv_work, cv_done; int quit;
    int next_work, npending;                              /* ring index of the next job to write, jobs whose records are on the host but not yet taken by a writer */
    pthread_t disp; int disp_on; pthread_cond_t cv_disp; int next_ready, nwait;   /* the one thread that waits for GPU events (in submission order) */
    /* the scheduler thread: GOP decisions + every GPU enqueue of a picture (some thirty runtime calls) happen here, not on the caller's thread */
    pthread_t sched; int sched_on; pthread_cond_t cv_sched, cv_sched_done; int sched_seen, sched_flush, sched_idle, sched_err;
    int dev_err; long dev_err_seq;                        /* sticky device error of the lane (dispatcher), and how many pictures had been submitted when it was seen */
    struct WorkerArg { struct Enc *e; int idx; } warg[64];
    /* output */
    QY265Nal nals[4 * MAX_JOBS + 8]; uint8_t *hdr; long hdr_len, hdr_part[3];
    uint8_t *outbuf; size_t outcap, outpos;               /* NAL payloads handed to the caller live here until the next call */
    ks265_enc_stats st;
    /* rate control (frame level, rc 1 / 2 / 4): the QP offset of a mini-GOP is a closed-form function of the pictures coded at least RC_LAG pictures earlier
     * (coding order) - the scheduler waits for exactly those, so the stream does not depend on thread timing.  rc_qp_delta belongs to the scheduler thread;
     * the sums are updated under mu in coding order (rc_account) */
#define RC_LAG 16
#define RC_HIST 512
    /* -rdoq 1 (the reference's rdoQuant in the pixel path, round 6): its bit tables follow the stream - a P / B picture is quantised with the tables built from the context states of
     * the latest picture of its kind that was coded at least RC_LAG + 1 pictures earlier (rq_hist: tables per accounted picture, like rc_hist: what a picture is coded with does
     * not depend on how fast the writers were), else from the initial states of its slice type at its QP; Job::rq_host: the picture's tables + lambdas, pinned */
#define RQ_HIST 64
    int rdoq_on; int32_t *rq_hist; char rq_hist_kind[RQ_HIST]; int64_t rq_lam[104]; long rq_gop_seq;   /* rq_gop_seq: the submission number of the current GOP's key picture - tables never cross a key picture (closed GOPs: the stream does not depend on how GOPs are dealt to lanes) */
    double rc_sum_norm, rc_hist[RC_HIST]; long rc_acc_seq, rc_sub; int rc_acc_idx; int rc_qp_delta;
    double rc_sum_budget, rc_bhist[RC_HIST], rc_sum_real, rc_rhist[RC_HIST];   /* the budget and the bits really produced, summed picture by picture like rc_sum_norm */
} Enc;

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static void rc_account(Enc *e);
static void lane_wake_top(Enc *e)
{
    TopWake *w = e->wake;
    if (!w) return;
    pthread_mutex_lock(&w->mu); ++w->seq; pthread_cond_broadcast(&w->cv); pthread_mutex_unlock(&w->mu);
}
static int lane_recon_on(Enc *e);
static int hip_rc(int r) { return r == 0 ? QY_OK : r == KS265_OUTOFMEMORY ? QY_OUTOFMEMORY : r == KS265_POINTER ? QY_POINTER : r == KS265_NOTSUPPORTED ? QY_NOTSUPPORTED : QY_FAIL; }

static int pic_alloc(Enc *e, ks265_pic *p)
{
    /* KS265_PIC_SPACER = bytes: an unused allocation in front of every picture (where pictures lie against each other in device memory decides what shares a memory channel:
     * round 4) */
    const char *sp = getenv("KS265_PIC_SPACER");
    if (sp && atol(sp) > 0 && e->nspacer < 64) { void *d = NULL; if (!ks265_dev_malloc(e->ctx, &d, (size_t)atol(sp))) e->spacer[e->nspacer++] = d; }
    int r = ks265_dev_malloc(e->ctx, (void **)&p->y, (size_t)e->geom.bytes_y);
    if (!r) r = ks265_dev_malloc(e->ctx, (void **)&p->u, (size_t)e->geom.bytes_c);
    if (!r) r = ks265_dev_malloc(e->ctx, (void **)&p->v, (size_t)e->geom.bytes_c);
    return r;
}
static void pic_free(Enc *e, ks265_pic *p) { ks265_dev_free(e->ctx, p->y); ks265_dev_free(e->ctx, p->u); ks265_dev_free(e->ctx, p->v); p->y = p->u = p->v = NULL; }

/* ---- the dispatcher: ONE host thread waits for the pictures' events, in submission order (the copy-out stream finishes them in that order), and
 *      releases them to the writers.  Writers that waited inside the GPU runtime themselves would contend with the calling thread's enqueues. */
static void *dispatcher(void *arg)
{
    Enc *e = (Enc *)arg;
    pthread_mutex_lock(&e->mu);
    for (;;) {
        while (!e->quit && e->nwait == 0) pthread_cond_wait(&e->cv_disp, &e->mu);
        if (e->quit) break;
        Job *j = &e->jobs[e->next_ready];
        pthread_mutex_unlock(&e->mu);
        int err = 0;
        err = ks265_event_wait(e->ctx_out, j->ev);
        /* the device error word of the streams that code pictures (a wavefront that timed out leaves a picture whose levels do not match its reconstruction): the
         * picture - this one or one enqueued after it - fails instead of going out as if nothing had happened */
        if (!err) err = ks265_take_device_error(e->ctx);
        if (!err && e->ctx_key) err = ks265_take_device_error(e->ctx_key);
        if (err) logf_(2, e->log_level, "ks265enc: device error at picture %d: %s\n", j->disp, "intra wavefront timeout or device error flag (ks265_take_device_error)");
        j->t_event = now_ms();
        pthread_mutex_lock(&e->mu);
        /* the error word does not say WHICH picture raised it: it may be one enqueued after this one (a key picture running ahead on its own stream), and every picture
         * predicted from a broken reconstruction is broken too.  So the error sticks to the lane: this picture and all that were already submitted when it was seen fail,
         * and so does everything after them up to the first key picture (IDR: a closed GOP) submitted later */
        if (err && !e->dev_err) { e->dev_err = err; e->dev_err_seq = e->rc_sub; }
        else if (!err && e->dev_err) { if (j->kind == 'I' && j->sub_seq >= e->dev_err_seq) e->dev_err = 0; else err = e->dev_err; }
        j->ev_err = err;
        e->next_ready = (e->next_ready + 1) % e->ring; --e->nwait; ++e->npending;
        pthread_cond_broadcast(&e->cv_work);
    }
    pthread_mutex_unlock(&e->mu);
    return NULL;
}

/* ---- slice writers.  Every picture is written as CTU-row substreams (entropy_coding_sync, the reference's WPP: ks265_wpp_*).  A writer thread takes
 *      the next picture and codes its rows one after the other - pictures of a GOP in parallel, no waiting inside a picture.  A key picture's slice is
 *      an order of magnitude longer than a P picture's and output is in coding order, so writers that look for work JOIN a key picture that is in
 *      progress before they start another picture: its rows
 *      are handed out in ascending order, each row runs at most two CTUs behind the row above (the wavefront of qy265executeEncCtuTaskWpp enc@0x475d20). */
/* the level planes of one CTU row of a picture from its compact records: clear the lines the previous picture in this job slot had set there, put this
 * picture's lines in place.  Done by the thread that writes the row (rows of a key picture by different threads); a line that straddles two CTU rows is
 * written twice with the same bytes.  ctu_row < 0: the whole picture. */
static void expand_levels(Enc *e, Job *j, int ctu_row)
{
    const size_t npx = (size_t)e->W * e->H, pb[3] = {npx * 2, npx / 2, npx / 2};
    uint8_t *plane[3] = {(uint8_t *)j->lvl[0], (uint8_t *)j->lvl[1], (uint8_t *)j->lvl[2]};
    size_t first[4]; first[0] = 0;
    for (int p = 0; p < 3; ++p) first[p + 1] = first[p] + (pb[p] + 63) / 64;
    const uint32_t *table = (const uint32_t *)(j->cmp + e->cmp_off[4]);
    const uint64_t *bm = (const uint64_t *)(j->cmp + e->cmp_off[5]);
    const uint8_t *data = j->cmp + e->cmp_off[6];
    for (int p = 0; p < 3; ++p) {
        /* byte range of the CTU row in this plane -> line range */
        const size_t rowb = (size_t)(p ? e->W / 2 : e->W) * 2, rows = (size_t)(p ? e->H / 2 : e->H), ch = p ? 32 : 64;
        size_t y0 = ctu_row < 0 ? 0 : (size_t)ctu_row * ch, y1 = ctu_row < 0 ? rows : y0 + ch;
        if (y1 > rows) y1 = rows;
        if (y0 >= y1) continue;
        const size_t l0 = first[p] + y0 * rowb / 64, l1 = first[p] + (y1 * rowb + 63) / 64;
        for (int pass = 0; pass < 2; ++pass)                           /* 0: clear the old lines, 1: place the new ones */
            for (size_t w = l0 / 64; w <= (l1 - 1) / 64; ++w) {
                uint64_t bits = pass ? bm[w] : j->dirty[w];
                if (w == l0 / 64) bits &= ~0ull << (l0 & 63);
                if (w == (l1 - 1) / 64 && (l1 & 63)) bits &= ~0ull >> (64 - (l1 & 63));
                if (!bits) continue;
                size_t k = 0;
                if (pass) {                                            /* rank of the word's first line in the data area: chunk base + lines of the chunk's earlier words */
                    k = (size_t)table[w / 16];
                    for (size_t q = w & ~(size_t)15; q < w; ++q) k += (size_t)__builtin_popcountll(bm[q]);
                    k += (size_t)__builtin_popcountll(bm[w] & ~bits & ((bits & (~bits + 1)) - 1));      /* lines of this word below the range */
                }
                while (bits) {
                    const int b = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    const size_t L = w * 64 + (size_t)b;
                    const size_t off = (L - first[p]) * 64, n = pb[p] - off < 64 ? pb[p] - off : 64;
                    if (pass) memcpy(plane[p] + off, data + (k++) * 64, n); else memset(plane[p] + off, 0, n);
                }
            }
    }
}

static Job *find_helpable(Enc *e)
{
    for (int i = 0, k = e->job_head; i < e->njobs; ++i, k = (k + 1) % e->ring) {
        Job *j = &e->jobs[k];
        if (j->used && j->started && !j->done && j->kind == 'I' && j->next_row < j->nrows) return j;
    }
    return NULL;
}
static void *worker(void *arg)
{
    Enc *e = ((struct WorkerArg *)arg)->e;
    pthread_mutex_lock(&e->mu);
    for (;;) {
        Job *j = NULL;
        int owner = 0;
        while (!e->quit) {
            /* a key picture in progress comes first: output is in coding order, everything coded after it waits for its slice */
            if ((j = find_helpable(e)) != NULL) break;
            if (e->npending > 0) { j = &e->jobs[e->next_work]; e->next_work = (e->next_work + 1) % e->ring; --e->npending; owner = 1; break; }
            pthread_cond_wait(&e->cv_work, &e->mu);
        }
        if (e->quit) break;
        if (owner) {
            pthread_mutex_unlock(&e->mu);
            j->t_taken = now_ms();
            int err = j->ev_err;                                 /* the dispatcher saw this picture's records reach the host */
            if (!err && e->recon_fd >= 0 && j->recon) {
                const size_t fsz = (size_t)e->W * e->H * 3 / 2;
                if (pwrite(e->recon_fd, j->recon, fsz, (off_t)j->disp * (off_t)fsz) != (ssize_t)fsz) err = KS265_FAIL;
            }
            if (!err && e->md5 && j->recon) {
                const size_t np = (size_t)e->W * e->H;
                md5_hex(j->recon, np, j->md5[0]); md5_hex(j->recon + np, np / 4, j->md5[1]); md5_hex(j->recon + np + np / 4, np / 4, j->md5[2]);
            }
            if (!err) {
                ks265_slice_in *s = &j->sin;
                memset(s, 0, sizeof *s);
                s->nal_type = j->nal_type; s->slice_type = j->kind == 'I' ? KS265_SLICE_I : j->kind == 'P' ? KS265_SLICE_P : KS265_SLICE_B;
                s->poc = j->poc; s->qp = j->qp; s->num_rps = j->nrps;
                memcpy(s->rps_poc, j->rps_poc, sizeof s->rps_poc); memcpy(s->rps_used, j->rps_used, sizeof s->rps_used);
                s->num_l0 = j->nl0; s->num_l1 = j->nl1; memcpy(s->l0_poc, j->l0, sizeof s->l0_poc); memcpy(s->l1_poc, j->l1, sizeof s->l1_poc);
                s->cu8 = j->cu8; s->lvl[0] = j->lvl[0]; s->lvl[1] = j->lvl[1]; s->lvl[2] = j->lvl[2]; s->sao = e->use_sao && !j->no_sao ? j->sao : NULL;
                s->qp_map = e->qmap_on ? j->qp_map : NULL;
                if (e->qmap_on && e->qmap_fd >= 0) {                     /* KS265_DUMP_QPMAP=file (tests): display index, kind, QP, CTU count, then the map - one record per picture, whole records only */
                    const int nct = e->geom.ctu_cols * e->geom.ctu_rows;
                    uint8_t *rec = (uint8_t *)malloc(16 + (size_t)nct);
                    if (rec) { const int32_t hd[4] = {j->disp, j->kind, j->qp, nct}; memcpy(rec, hd, 16); memcpy(rec + 16, j->qp_map, (size_t)nct); if (write(e->qmap_fd, rec, 16 + (size_t)nct) < 0) { /* diagnostics only */ } free(rec); }
                }
                err = ks265_wpp_begin(&e->scfg, s, j->wpp);
            }
            pthread_mutex_lock(&e->mu);
            j->t_write_ms = 0;
            if (err) {                                           /* nothing to write: the picture is finished (with its error) */
                j->error = err; j->done = 1;
                rc_account(e);
                pthread_cond_broadcast(&e->cv_done); lane_wake_top(e);
                continue;
            }
            j->nrows = ks265_wpp_rows(j->wpp); j->next_row = 0; j->rows_done = 0; j->started = 1;
            if (j->kind == 'I') pthread_cond_broadcast(&e->cv_work);   /* idle writers may join */
        }
        while (j->next_row < j->nrows) {
            const int row = j->next_row++;
            pthread_mutex_unlock(&e->mu);
            const double t0 = now_ms();
            expand_levels(e, j, row);
            (void)ks265_wpp_code_row(j->wpp, row);               /* a failure is kept in the job memory and reported by ks265_wpp_finish */
            const double dt = now_ms() - t0;
            pthread_mutex_lock(&e->mu);
            j->t_write_ms += dt;
            if (++j->rows_done == j->nrows) {                    /* the thread that finishes the last row assembles the NAL unit */
                pthread_mutex_unlock(&e->mu);
                const double t1 = now_ms();
                const long n = ks265_wpp_finish(j->wpp, j->nal, j->nal_cap);
                j->fctx_ok = e->rdoq_on && n >= 0 && ks265_slice_final_contexts(&e->scfg, j->wpp, j->fctx, (int)sizeof j->fctx, NULL) > 0;
                memcpy(j->dirty, j->cmp + e->cmp_off[5], e->cmp_off[6] - e->cmp_off[5]);      /* what the next picture in this slot has to clear */
                pthread_mutex_lock(&e->mu);
                j->t_write_ms += now_ms() - t1;
                j->nal_len = n; j->error = n < 0 ? (int)n : 0; j->done = 1; j->t_done = now_ms();
                rc_account(e);
                pthread_cond_broadcast(&e->cv_done); lane_wake_top(e);
            }
        }
    }
    pthread_mutex_unlock(&e->mu);
    return NULL;
}

/* rate control: account finished pictures in coding order (under mu; called where a picture becomes done and before its ring slot is freed).  A picture's bits
 * are normalised to the base QP with the 6-steps-per-octave rule (bits x 2^(delta / 6)), so the sums say what the stream would have cost without the
 * controller.  rc_hist keeps the running sum after every picture: the decision for picture n reads the entry of picture n - RC_LAG - 1 whatever has been
 * accounted since (pictures finish early or late; the decision must not depend on that). */
static void rc_account(Enc *e)
{
    while (e->rc_acc_seq < e->rc_sub) {
        Job *j = &e->jobs[e->rc_acc_idx];
        if (!j->used || !j->done) break;
        if (e->rdoq_on) {                                         /* the tables this picture's final states give, for the pictures of its kind RC_LAG and more pictures on */
            const int slot = (int)(e->rc_acc_seq % RQ_HIST);
            e->rq_hist_kind[slot] = 0;
            if (j->fctx_ok && !j->error && ks265_rdoq_tables(&e->scfg, j->fctx, 0, 0, e->rq_hist + (size_t)slot * 1440) == 0) e->rq_hist_kind[slot] = (char)j->kind;
        }
        e->rc_sum_norm += (double)(j->nal_len > 0 ? j->nal_len : 0) * 8.0 * exp2((double)j->rc_delta / 6.0);
        e->rc_hist[e->rc_acc_seq % RC_HIST] = e->rc_sum_norm;
        e->rc_sum_budget += j->rc_budget; e->rc_bhist[e->rc_acc_seq % RC_HIST] = e->rc_sum_budget;      /* per picture at the bitrate it was handed in with: a later
                                                                                                           * QY265EncoderReconfig does not re-price the past */
        e->rc_sum_real += (double)(j->nal_len > 0 ? j->nal_len : 0) * 8.0; e->rc_rhist[e->rc_acc_seq % RC_HIST] = e->rc_sum_real;
        e->rc_acc_idx = (e->rc_acc_idx + 1) % e->ring; ++e->rc_acc_seq;
    }
}
/* scheduler thread: the offset for the pictures submitted next.  Waits until every picture up to RC_LAG before the next one is written (the writers do not need
 * the caller for that), then delta = 6 log2(normalised bits / budget) over exactly those pictures, at most 4 steps away from the previous value. */
static int rc_decide(Enc *e)
{
    if (!(e->cfg.rc == 1 || e->cfg.rc == 2 || e->cfg.rc == 4)) return 0;
    pthread_mutex_lock(&e->mu);
    const long need = e->rc_sub - RC_LAG;                              /* pictures [0, need) decide */
    for (;;) {
        rc_account(e);
        if (e->rc_acc_seq >= need || e->quit) break;
        pthread_cond_wait(&e->cv_done, &e->mu);
    }
    int d = e->rc_qp_delta;
    if (need >= 4 && !e->quit) {
        const double spent = e->rc_hist[(need - 1) % RC_HIST], budget = e->rc_bhist[(need - 1) % RC_HIST], real = e->rc_rhist[(need - 1) % RC_HIST];
        if (spent > 0 && budget > 0) {
            /* model term: what the stream costs at the base QP against the budget, 6 QP steps per octave; feedback term: the bits really produced so far against the
             * budget so far (half weight, at most 2 steps) - where the content's rate does not halve every 6 steps the model term alone settles beside the target */
            double fb = need >= 32 && real > 0 ? 3.0 * log2(real / budget) : 0.0;
            fb = fb > 2.0 ? 2.0 : fb < -2.0 ? -2.0 : fb;
            int want = (int)lrint(6.0 * log2(spent / budget) + fb);
            if (want > d + 4) want = d + 4;
            if (want < d - 4) want = d - 4;
            d = want < -51 ? -51 : want > 51 ? 51 : want;
        }
    }
    pthread_mutex_unlock(&e->mu);
    return d;
}

static int dpb_find(Enc *e, int poc) { for (int i = 0; i < e->ndpb + 2 + 4; ++i) if (e->dpb_poc[i] == poc) return i; return -1; }   /* incl. the key pictures' and the anchor lane's own slots (unused ones hold no POC) */
static int dpb_free_slot(Enc *e, const int *keep, int nkeep)
{
    for (int i = 0; i < e->ndpb; ++i) {
        int k = 0;
        for (int q = 0; q < nkeep; ++q) if (e->dpb_poc[i] == keep[q]) k = 1;
        if (!k || e->dpb_poc[i] < 0) return i;
    }
    return -1;
}

/* ---- cuTree: the lookahead window's costs, propagation and finish (scheduler thread only; see Enc::ct_on) -------------------------------------------------------------------- */
static Input *input_at(Enc *e, int disp);
typedef struct CtNode { int b, p0, p1, is_ref; } CtNode;
/* the integer motion lambda of every QP (TEncParam+0x720 as recorded inside the reference: x264's table; the lookahead searches with entry 12 = 1, the others only matter where the
 * reference's cost table is overrun into its neighbouring rows) */
static const uint16_t kCtLambda[52] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 7, 8, 9, 11, 12, 13, 15, 17, 19, 21, 24, 27, 30, 34, 38, 42, 47, 53, 60, 67};
/* TEncParam words calcFrameCost reads, by preset (read inside the reference with every preset, -rc 3: zero_thr +0x3a0, fast_intra +0x3a4, +0x36c, +0x538, +0x3b4; +0x3a8 = 13 and
 * +0x8 = 0 throughout); blocks of 16 x 16 half-size samples from 1280 x 720 up and for ultrafast .. veryfast, else 8 x 8 */
static const struct { int zero_thr, fast_intra, f36c, f538, f3b4, lg4; } kCtPreset[9] = {{4, 4, 0, 120, 2, 1}, {4, 4, 0, 120, 2, 1}, {4, 1, 0, 0, 1, 1}, {0, 0, 1, 0, 1, 0}, {0, 0, 1, 0, 1, 0},
                                                                                              {0, 0, 1, 0, 1, 0}, {0, 0, 1, 0, 1, 0}, {0, 0, 1, 0, 1, 0}, {0, 0, 0, 0, 1, 0}};
static struct CtPic *ct_slot(Enc *e, int disp) { return &e->ct[((disp % CT_RING) + CT_RING) % CT_RING]; }
static int ct_open(Enc *e, int device)
{
    const int ps = e->ct_preset;
    e->ct_w = e->W / 2; e->ct_h = e->H / 2;
    e->ct_lg = (kCtPreset[ps].lg4 || (long)e->W * e->H >= 1280L * 720) ? 4 : 3;
    const int bs = 1 << e->ct_lg;
    e->ct_nx = (e->ct_w + bs - 1) / bs; e->ct_ny = (e->ct_h + bs - 1) / bs;
    e->ct_pad = 40 + bs;                                               /* the search leaves the picture by at most merange / 2 + 1 = 33; the last block row / column may overhang it */
    e->ct_stride = (e->ct_w + 2 * e->ct_pad + 63) & ~63;
    const size_t n = (size_t)e->ct_nx * e->ct_ny, lowsz = (size_t)e->ct_stride * (size_t)(e->ct_h + 2 * e->ct_pad);
    int r = ks265_create(&e->ctx_ct, device);
    /* one block of device memory per picture of the ring: the padded half-size plane, then the per-block arrays (64-byte aligned) */
    const size_t a64 = 63;
    size_t off[16], o = (lowsz + a64) & ~a64;
    const size_t sz[14] = {n * 2, n * 2, n * 2, n * 2, n, (n + 3) / 4, n * 4, n * 4, n * 4, n * 4, n * 8, n * 8, sizeof(ks265_cfc_sums), 0};
    for (int k = 0; k < 13; ++k) { off[k] = o; o = (o + sz[k] + a64) & ~a64; }
    for (int i = 0; i < CT_RING && !r; ++i) {
        struct CtPic *c = &e->ct[i];
        c->disp = -1000000;
        r = ks265_dev_malloc(e->ctx_ct, (void **)&c->blk, o);
        if (r) break;
        c->low = c->blk + (size_t)e->ct_pad * e->ct_stride + e->ct_pad;
        c->intra = (uint16_t *)(c->blk + off[0]); c->invq = (uint16_t *)(c->blk + off[1]); c->prop = (uint16_t *)(c->blk + off[2]); c->inter = (uint16_t *)(c->blk + off[3]);
        c->imode = c->blk + off[4]; c->bits = c->blk + off[5]; c->mv0 = (int32_t *)(c->blk + off[6]); c->c0 = (int32_t *)(c->blk + off[7]); c->mv1 = (int32_t *)(c->blk + off[8]);
        c->c1 = (int32_t *)(c->blk + off[9]); c->aq_off = (double *)(c->blk + off[10]); c->qoff = (double *)(c->blk + off[11]); c->sums = (ks265_cfc_sums *)(c->blk + off[12]);
        r = ks265_event_create(e->ctx_ct, &c->ev_used);
    }
    if (!r) r = ks265_dev_malloc(e->ctx_ct, &e->ct_ws, ks265_calc_frame_cost_workspace(e->ct_nx, e->ct_ny));
    if (!r) r = ks265_dev_malloc(e->ctx_ct, (void **)&e->ct_acc, n * 16);
    if (!r) r = ks265_memset_async(e->ctx_ct, e->ct_acc, 0, n * 16);
    if (!r) r = ks265_dev_malloc(e->ctx_ct, (void **)&e->ct_scratch, 16);
    if (!r && e->aq_on) r = ks265_dev_malloc(e->ctx_ct, (void **)&e->ct_full, (size_t)e->W * e->H * 3 / 2);
    if (!r) r = ks265_event_create(e->ctx_ct, &e->ct_ev);
    return r;
}
static void ct_close(Enc *e)
{
    if (!e->ctx_ct) return;
    ks265_synchronize(e->ctx_ct);
    for (int i = 0; i < CT_RING; ++i) { ks265_dev_free(e->ctx_ct, e->ct[i].blk); if (e->ct[i].ev_used) ks265_event_destroy(e->ctx_ct, e->ct[i].ev_used); }
    ks265_dev_free(e->ctx_ct, e->ct_ws); ks265_dev_free(e->ctx_ct, e->ct_acc); ks265_dev_free(e->ctx_ct, e->ct_scratch); ks265_dev_free(e->ctx_ct, e->ct_full);
    if (e->ct_ev) ks265_event_destroy(e->ctx_ct, e->ct_ev);
    ks265_destroy(e->ctx_ct); e->ctx_ct = NULL;
}
/* picture `disp` in the ring: its half-size picture (downsample_c enc@0x4a6a60, edges replicated), its AQ offsets / inverse qscale factors (calcFrameAdaptQuant enc@0x4653c0 on the
 * lookahead's block grid, as the reference calls it; without -aq: zero offsets, factor 256) */
static int ct_prepare(Enc *e, int disp)
{
    struct CtPic *c = ct_slot(e, disp);
    if (c->disp == disp) return 0;
    Input *in = input_at(e, disp);
    if (!in) return KS265_FAIL;
    ks265_ctx *cx = e->ctx_ct;
    const int n = e->ct_nx * e->ct_ny;
    int r = c->used ? ks265_stream_wait_event(cx, c->ev_used) : 0;       /* the picture that held this slot CT_RING pictures ago: its QP map has been taken */
    c->disp = disp; c->p0 = c->p1 = -1000000; c->intra_done = 0;
    const uint8_t *full = NULL;
    if (in->dev) { if (!r && !in->up_sync) r = ks265_stream_wait_event(cx, in->ev_up); full = in->dev; }
    else if (e->ct_full) { if (!r) r = ks265_memcpy_h2d_async(cx, e->ct_full, in->i420, (size_t)e->W * e->H * 3 / 2); full = e->ct_full; }
    if (!r) r = full ? ks265_downsample_rect(cx, full, e->W, c->low, e->ct_stride, e->ct_w, e->ct_h) : ks265_downsample_from_host(cx, in->i420, e->W, c->low, e->ct_stride, e->ct_w, e->ct_h);
    if (!r) r = ks265_pad_plane(cx, c->low, e->ct_stride, e->ct_w, e->ct_h, e->ct_pad);
    if (!r && e->aq_on) r = ks265_frame_adapt_quant(cx, full, e->W, full + (size_t)e->W * e->H, full + (size_t)e->W * e->H * 5 / 4, e->W / 2, e->ct_nx, e->ct_ny, n, e->cfg.fAqStrength, c->aq_off, c->invq, e->ct_scratch);
    else if (!r) { r = ks265_memset_async(cx, c->aq_off, 0, (size_t)n * 8); if (!r) r = ks265_fill_u16(cx, c->invq, n, 256); }
    if (!r) r = ks265_memset_async(cx, c->sums, 0xff, sizeof(ks265_cfc_sums));      /* L+0x684 / +0x7c8 = -1: nothing computed yet */
    return r;
}
/* calcFrameCost of picture b against (p0, p1) (display indices; p1 < 0: none; p0 < 0: the intra pass alone), unless that is what its arrays hold */
static int ct_cost(Enc *e, int b, int p0, int p1)
{
    struct CtPic *c = ct_slot(e, b);
    if (p0 < 0) { if (c->intra_done) return 0; }
    else if (c->p0 == p0 && c->p1 == p1) return 0;
    const int ps = e->ct_preset;
    ks265_cfc_params q; memset(&q, 0, sizeof q);
    q.w = e->ct_w; q.h = e->ct_h; q.nx = e->ct_nx; q.ny = e->ct_ny; q.cnt = e->ct_nx * e->ct_ny; q.stride = e->ct_stride;
    q.d0 = p0 < 0 ? 0 : b - p0; q.d1 = p0 < 0 || p1 < 0 ? 0 : p1 - b; q.flag = 0; q.slice_type = p0 < 0 ? 2 : q.d1 ? 0 : 1;
    q.merange = 64; q.lg = e->ct_lg; q.zero_thr = kCtPreset[ps].zero_thr; q.fast_intra = kCtPreset[ps].fast_intra; q.scenecut = g_cli.scenecut ? g_cli.scenecut : 30; q.preset = ps; q.p8 = 0;
    q.aq = e->aq_on; q.b_intra = 1; q.f3a8 = 13; q.f36c = kCtPreset[ps].f36c; q.f538 = kCtPreset[ps].f538; q.f3b4 = kCtPreset[ps].f3b4;
    q.do_list[0] = q.d0 > 0; q.do_list[1] = q.d1 > 0; q.intra_done = c->intra_done;
    memcpy(q.lambda_tab, kCtLambda, sizeof q.lambda_tab);
    const int r = ks265_calc_frame_cost(e->ctx_ct, &q, c->low, q.d0 ? ct_slot(e, p0)->low : NULL, q.d1 ? ct_slot(e, p1)->low : NULL, c->intra, c->imode, c->invq, c->inter, c->bits,
                                        c->mv0, c->c0, c->mv1, c->c1, c->sums, e->ct_ws);
    if (p0 >= 0) { c->p0 = p0; c->p1 = p1; }
    if (q.d1 == 0) c->intra_done = 1;                                  /* (the reference marks the intra costs done after a pass without a list-1 picture only, enc@0x4a8be5) */
    return r;
}
/* the pictures of (lo, end] in coding order as this scheduler will code them: mini-GOPs (lo, hi], (hi, hi + span], .. - the anchor P picture first, then the B pictures (pyramid:
 * breadth first, references = the interval's ends; else every B picture between the two anchors) */
static int ct_structure(const Enc *e, int lo, int hi, int end, CtNode *out, int cap)
{
    int n = 0;
    const int span = e->gop_b + 1;
    while (lo < end && n < cap) {
        if (hi > end) hi = end;
        out[n++] = (CtNode){hi, lo, -1, 1};
        if (hi - lo > 1) {
            if (e->hier && ((hi - lo) & (hi - lo - 1)) == 0) {
                struct { int lo, hi; } cur[8], nxt[8]; int nc = 1;
                cur[0].lo = lo; cur[0].hi = hi;
                while (nc) {
                    int nn = 0;
                    for (int i = 0; i < nc; ++i) {
                        if (cur[i].hi - cur[i].lo < 2) continue;
                        const int mid = (cur[i].lo + cur[i].hi) / 2;
                        if (n < cap) out[n++] = (CtNode){mid, cur[i].lo, cur[i].hi, (mid - cur[i].lo >= 2) || (cur[i].hi - mid >= 2)};
                        nxt[nn].lo = cur[i].lo; nxt[nn++].hi = mid; nxt[nn].lo = mid; nxt[nn++].hi = cur[i].hi;
                    }
                    memcpy(cur, nxt, sizeof cur); nc = nn;
                }
            } else for (int b = lo + 1; b < hi && n < cap; ++b) out[n++] = (CtNode){b, lo, hi, 0};
        }
        lo = hi; hi = lo + span;
    }
    return n;
}
/* the window's end for a mini-GOP that ends at display index a: ct_depth pictures further, inside this closed GOP, inside what has arrived.  -1: not all of it is there yet */
static int ct_window_end(Enc *e, int a, int iper, int have, int flush, int gop_end)
{
    int end = a + e->ct_depth;
    if (iper > 0 && end > e->gop_start + iper - 1) end = e->gop_start + iper - 1;
    if (gop_end >= a && end > gop_end) end = gop_end;
    for (int k = a + 1; k <= end && k < have; ++k) { const Input *ik = input_at(e, k); if (ik && ik->key) { end = k - 1; break; } }   /* a requested key picture closes the GOP in front of it */
    if (end >= have) { if (!flush) return -1; end = have - 1; }
    return end < a ? a : end;
}
/* costs + propagation over (first, end], then the offsets of the pictures in [fin_lo, fin_hi] (the key picture itself when key >= 0) */
static int ct_run(Enc *e, int key, int d, int a, int end)
{
    CtNode nodes[CT_RING];
    const int first = key >= 0 ? key : d;
    if (end - first >= CT_RING - 2) end = first + CT_RING - 3;
    const int nn = key >= 0 ? ct_structure(e, key, key + e->gop_b + 1, end, nodes, CT_RING) : ct_structure(e, d, a, end, nodes, CT_RING);
    const int n = e->ct_nx * e->ct_ny;
    int r = 0;
    for (int k = first; k <= end && !r; ++k) r = ct_prepare(e, k);
    for (int k = first; k <= end && !r; ++k) r = ks265_memset_async(e->ctx_ct, ct_slot(e, k)->prop, 0, (size_t)n * 2);       /* (the reference clears L+0x40 of the window before every pass, enc@0x480116) */
    if (!r && key >= 0) r = ct_cost(e, key, -1, -1);                  /* the key picture's intra costs (enc@0x480bb1) */
    for (int i = nn - 1; i >= 0 && !r; --i) {
        const CtNode *nd = &nodes[i];
        struct CtPic *c = ct_slot(e, nd->b), *r0 = ct_slot(e, nd->p0), *r1 = nd->p1 >= 0 ? ct_slot(e, nd->p1) : r0;
        r = ct_cost(e, nd->b, nd->p0, nd->p1);
        if (!r) r = ks265_cutree_propagate(e->ctx_ct, e->ct_lg, e->ct_nx, e->ct_ny, c->intra, c->invq, c->prop, c->inter, c->bits, c->mv0, nd->p1 >= 0 ? c->mv1 : c->mv0, r0->prop, r1->prop, e->ct_acc);
    }
    /* the finish for what is about to be coded: reference pictures get the tree's offsets, the others their AQ offsets alone (enc@0x480854..0x480a54) */
    for (int i = -1; i < nn && !r; ++i) {
        int b, is_ref, dbl = 0;
        if (i < 0) { if (key < 0) continue; b = key; is_ref = 1; }
        else { if (key >= 0 || nodes[i].b > a) continue; b = nodes[i].b; is_ref = nodes[i].is_ref; dbl = nodes[i].p1 < 0 && end == a; }   /* nothing of the window lies behind this mini-GOP: its anchor's
                                                                                                                                             * propagated cost counts twice (enc@0x480c6a, 0x4809d2) */
        struct CtPic *c = ct_slot(e, b);
        r = ks265_memcpy_d2d_async(e->ctx_ct, c->qoff, c->aq_off, (size_t)n * 8);
        if (!r && is_ref) r = ks265_cutree_finish(e->ctx_ct, n, c->intra, c->invq, c->prop, c->aq_off, dbl, c->qoff);
    }
    if (!r) r = ks265_event_record(e->ctx_ct, e->ct_ev);
    return r;
}

/* enqueue one picture: GPU work + copies on the stream, then hand it to the writers */
static int submit(Enc *e, Input *in, int kind, int poc, int qp, const int *l0, int nl0, const int *l1, int nl1, const int *keep_after, int nkeep, int is_ref, int key_headers)
{
    /* wait for a free job slot (the ring is full only if the consumer did not drain it: block on the oldest) */
    pthread_mutex_lock(&e->mu);
    const double tw0 = now_ms();
    while (e->njobs == e->ring && !e->quit) {
        pthread_cond_broadcast(&e->cv_sched_done);                      /* a caller waiting for this thread in lane_put goes on and collects output instead */
        pthread_cond_wait(&e->cv_done, &e->mu);                         /* take_output signals when it has freed ring slots */
    }
    e->st.submit_wait_ms += now_ms() - tw0;
    if (e->quit) { pthread_mutex_unlock(&e->mu); return QY_FAIL; }       /* drained by take_output() of the calling thread itself: never full here */
    Job *j = &e->jobs[e->job_tail];
    pthread_mutex_unlock(&e->mu);
    const size_t fsz = (size_t)e->W * e->H * 3 / 2;
    const int k = (int)(e->seq % NPIPE);
    const int recycled = e->seq >= NPIPE;
    /* copy-in stream: the slot's previous picture must have been unpacked before the buffer is overwritten */
    int r = recycled ? ks265_stream_wait_event(e->ctx_in, e->ev_loaded[k]) : 0;
    /* pixel path: the main stream / frame object, or the key pictures' own */
    const int on_key = kind == 'I' && e->key_overlap && (in->iper <= 0 || in->iper >= 32);
    const int on_anc = kind == 'P' && e->anc_on && e->key_overlap && nl0 == 1;      /* (key_overlap goes off with the reconstruction dump: everything on the main stream then) */
    const int split = e->split && !on_key && !on_anc;
    ks265_ctx *cx = on_key ? e->ctx_key : on_anc ? e->ctx_anc : e->ctx;
    ks265_frame *fr = on_key ? e->frame_key : on_anc ? e->frame_anc : e->frame;
    ks265_pic srcp = on_key ? e->src_key : on_anc ? e->src_anc : split ? e->srcq[k] : e->src;
    if (!r && split && recycled) r = ks265_stream_wait_event(e->ctx_in, e->ev_drained[k]);   /* srcq[k]'s last reader (the SSE of three pictures ago) is through */
    /* with the lookahead the picture is on the device already, or on its way: uploaded into the slot's twin when it was handed in (round 4); else an H2D copy here, on this
     * stream, behind the waits above.  Graph replay needs fixed addresses: a device-to-device copy into the rotation buffer */
    const uint8_t *din = e->dev_in[k];
    /* an anchor on its lane whose picture is on the device already waits for the upload itself: a mark on the copy-in stream would sit behind that stream's work for the
     * pictures in front of it - and, HIP streams sharing hardware queues (four by default), behind whatever else runs in that queue: measured, the anchor then starts when
     * the last B picture of the mini-GOP before it has left the device */
    const int direct = on_anc && in->dev && !e->use_graph;
    if (direct) { if (!r && !in->up_sync) r = ks265_stream_wait_event(cx, in->ev_up); din = in->dev; }
    else if (in->dev) {
        if (!r && !in->up_sync) r = ks265_stream_wait_event(e->ctx_in, in->ev_up);
        if (!r && e->use_graph) r = ks265_memcpy_d2d_async(e->ctx_in, e->dev_in[k], in->dev, fsz);
        else din = in->dev;
    } else if (!r) r = ks265_memcpy_h2d_async(e->ctx_in, e->dev_in[k], in->i420, fsz);
    if (!r && split) r = ks265_load_i420_on(e->ctx_in, fr, din, srcp);
    if (!r && !direct) r = ks265_event_record(e->ctx_in, e->ev_h2d[k]);
    uint64_t *dsse = on_key ? e->dev_sse_key : on_anc ? e->dev_sse_anc : e->dev_sse;
    if (!r && !direct) r = ks265_stream_wait_event(cx, e->ev_h2d[k]);
    /* graph path: a P picture with one reference on the main stream, once the first pictures have made every lazy allocation */
    const int graphable = e->use_graph && ((kind == 'P' && nl0 == 1) || (kind == 'B' && nl0 == 1 && nl1 == 1)) && !on_key && !on_anc && !e->recon_on && e->seq >= 8;
    if (!graphable && !split) {
        if (!r) r = ks265_load_i420(fr, din, srcp);
        if (!r) r = ks265_event_record(cx, e->ev_loaded[k]);
    }
    if (!r) r = ks265_frame_set_qp(fr, qp, kind == 'I' ? kLambdaQ4[qp] : kLambdaInterQ4[qp]);
    /* round 6: a B picture nothing predicts from (half the pictures of a pyramid of 8) runs without intra candidates, without the joint refinement of its bi-predictive CUs and
     * without SAO - on the CPU mirror and on the MI355X its bytes at equal PSNR-Y stay (the refinement even costs bytes at QP + 4), a quarter of its kernel time goes */
    /* ... and a B picture others predict from whose own references are at most two pictures away (the second-deepest layer of a pyramid) keeps the refinement but runs without intra
     * candidates and without SAO: neutral at equal PSNR-Y on the mirror's three clips (profiles/r06_lean_b.txt) */
    const int near = kind == 'B' && is_ref && nl0 > 0 && nl1 > 0 && poc - l0[0] <= 2 && l1[0] - poc <= 2;
    const int lean = !e->lean_b || kind != 'B' ? 0 : !is_ref ? 2 : near && e->lean_b != 3 ? 1 : 0;          /* (KS265_LEAN_B=3: the non-reference pictures alone) */
    if (!r) r = lean == 2 ? ks265_frame_set_picture_tools(fr, 0, 0, 0, e->lean_b == 2 && e->me_method == 2 ? 1 : -1) : lean == 1 ? ks265_frame_set_picture_tools(fr, 0, -1, 0, -1) : ks265_frame_set_picture_tools(fr, -1, -1, -1, -1);
    if (e->rdoq_on && kind == 'I') e->rq_gop_seq = e->rc_sub;
    if (!r && e->rdoq_on && kind != 'I') {
        /* -rdoq 1: wait until every picture up to RC_LAG before this one is accounted (as the rate controller does), then the latest tables of this picture's kind among them */
        int32_t *tb = j->rq_host;
        pthread_mutex_lock(&e->mu);
        const long s0 = e->rc_sub, need = s0 - RC_LAG;
        for (;;) { rc_account(e); if (e->rc_acc_seq >= need || e->quit) break; pthread_cond_wait(&e->cv_done, &e->mu); }
        int found = 0;
        for (long q = need - 1; q >= 0 && q > e->rq_gop_seq && q >= need - (RQ_HIST - RC_LAG - 8) && !found; --q)
            if (e->rq_hist_kind[q % RQ_HIST] == (char)kind) { memcpy(tb, e->rq_hist + (size_t)(q % RQ_HIST) * 1440, 1440 * sizeof(int32_t)); found = 1; }
        pthread_mutex_unlock(&e->mu);
        if (!found) r = ks265_rdoq_tables(&e->scfg, NULL, kind == 'P' ? KS265_SLICE_P : KS265_SLICE_B, qp, tb) ? KS265_FAIL : 0;
        memcpy(tb + 1440, e->rq_lam, sizeof e->rq_lam);
        if (!r) r = ks265_frame_set_rdoq(fr, tb, (const int64_t *)(tb + 1440), (const int64_t *)(tb + 1440) + 52);
    }
    if (!r && e->ct_on) {
        /* cuTree: the picture's block offsets were finished on the lookahead's stream when its mini-GOP was scheduled (ct_run): one QP per CTU around this picture's QP */
        ks265_ctx *ca = split ? e->ctx_in : cx;
        struct CtPic *cp = ct_slot(e, in->disp);
        int8_t *qm = on_key ? e->dev_qmap_key[e->nkeys & 1] : e->dev_qmap[k];
        if (cp->disp != in->disp) r = KS265_FAIL;
        if (!r) r = ks265_stream_wait_event(ca, e->ct_ev);
        if (!r) r = ks265_qoff_ctu_map(ca, cp->qoff, e->ct_nx, e->ct_ny, e->ct_lg, e->geom.ctu_cols, e->geom.ctu_rows, qp, e->cfg.qpmin, e->cfg.qpmax ? e->cfg.qpmax : 51, qm);
        if (!r) r = ks265_event_record(ca, cp->ev_used);
        cp->used = 1;
        if (!r) r = ks265_frame_set_qp_map(fr, qm);
        if (!r) r = ks265_memcpy_d2h_async(ca, j->qp_map, qm, (size_t)e->geom.ctu_cols * e->geom.ctu_rows);
        if (!r && split) { r = ks265_event_record(e->ctx_in, e->ev_h2d[k]); if (!r) r = ks265_stream_wait_event(cx, e->ev_h2d[k]); }
    } else if (!r && e->aq_on) {
        /* the source picture is on the device: block variances -> offsets (the reference's arithmetic) -> one QP per CTU around this picture's QP.  With the split pipeline
         * this runs on the copy-in stream right behind the unpack (off the pixel path's chain: the mean is a sequential sum, 0.15 ms at 2160p); the map stays in its rotation
         * slot while the picture's kernels and the copy-home run */
        ks265_ctx *ca = split ? e->ctx_in : cx;
        const size_t oy = (size_t)e->geom.pad_y * e->geom.stride_y + e->geom.pad_y, oc = (size_t)e->geom.pad_c * e->geom.stride_c + e->geom.pad_c;
        r = ks265_frame_adapt_quant(ca, srcp.y + oy, e->geom.stride_y, srcp.u + oc, srcp.v + oc, e->geom.stride_c, e->aq_nx, e->aq_ny, e->aq_nx * e->aq_ny, e->cfg.fAqStrength,
                                    e->aq_off[on_key], e->aq_inv[on_key], e->aq_scratch[on_key]);
        /* a key picture on its own stream runs beside the pictures of earlier GOPs - among them the one three submissions back, whose kernels still read dev_qmap[k]
         * (ADVICE r4): it gets a map of its own; everything that touches that one is on the key pictures' stream, in order */
        int8_t *qm = on_key ? e->dev_qmap_key[e->nkeys & 1] : e->dev_qmap[k];
        if (!r) r = ks265_aq_ctu_map(ca, e->aq_off[on_key], e->aq_nx, e->aq_ny, qp, e->cfg.rc ? e->cfg.qpmin : 0, e->cfg.rc && e->cfg.qpmax ? e->cfg.qpmax : 51, qm);
        if (!r) r = ks265_frame_set_qp_map(fr, qm);
        if (!r) r = ks265_memcpy_d2h_async(ca, j->qp_map, qm, (size_t)e->geom.ctu_cols * e->geom.ctu_rows);
        if (!r && split) { r = ks265_event_record(e->ctx_in, e->ev_h2d[k]); if (!r) r = ks265_stream_wait_event(cx, e->ev_h2d[k]); }   /* (recorded again behind the map: the pixel path waits for this one) */
    }
    int keep[32], nk = 0;
    for (int i = 0; i < nkeep; ++i) keep[nk++] = keep_after[i];
    for (int i = 0; i < nl0; ++i) keep[nk++] = l0[i];
    for (int i = 0; i < nl1; ++i) keep[nk++] = l1[i];
    int slot;
    if (on_key) {
        slot = e->ndpb + (e->nkeys & 1);
        /* everything enqueued on the main stream so far belongs to earlier GOPs: when this mark fires, the readers of the previous key picture are through.
         * The slot written now held the key picture two GOPs back: wait for the mark set when the previous key picture was submitted */
        if (!r && e->nkeys >= 1) r = ks265_event_record(e->ctx, e->ev_firstp[(e->nkeys - 1) & 1]);
        if (!r && e->nkeys >= 2) r = ks265_stream_wait_event(cx, e->ev_firstp[e->nkeys & 1]);
    } else if (on_anc) {
        slot = e->ndpb + 2 + (e->nanch & 3);
        /* the slot held the anchor four back, which closed mini-GOP nanch - 4; its readers on the main stream were the B pictures of mini-GOPs nanch - 4 and nanch - 3 */
        if (!r && e->nanch >= 4) r = ks265_stream_wait_event(cx, e->ev_mg[(e->nanch - 3) & 7]);
    } else slot = dpb_free_slot(e, keep, nk);
    if (slot < 0) return QY_FAIL;
    ks265_pic out = e->dpb[slot];
    if (!r && graphable) {
        const ks265_pic refp = e->dpb[dpb_find(e, l0[0])], ref1p = kind == 'B' ? e->dpb[dpb_find(e, l1[0])] : refp;
        const uint64_t key[9] = {(uint64_t)(uintptr_t)refp.y, (uint64_t)(uintptr_t)out.y, (uint64_t)(uintptr_t)e->dev_in[k], (uint64_t)(uintptr_t)e->stg[k],
                                 (uint64_t)qp, (uint64_t)ks265_frame_p_state(fr), (uint64_t)(e->cfg.calcPsnr != 0), (uint64_t)kind | ((uint64_t)lean << 8), (uint64_t)(uintptr_t)ref1p.y};
        void *exec = NULL;
        for (int i = 0; i < e->ngraph && !exec; ++i) if (!memcmp(e->graph[i].key, key, sizeof key)) exec = e->graph[i].exec;
        if (recycled) r = ks265_stream_wait_event(cx, e->ev_drained[k]);       /* the staging block of this rotation slot has been copied out */
        if (!r && exec) {
            r = ks265_graph_launch(cx, exec);
            if (!r && kind == 'P') r = ks265_frame_p_advance(fr);              /* a B picture leaves the P pictures' predictor chain alone */
        } else if (!r) {
            int keep_it = e->ngraph < MAX_GRAPHS;
            const int state0 = ks265_frame_p_state(fr);
            if (keep_it && ks265_capture_begin(cx)) { keep_it = 0; e->use_graph = 0; }   /* no capture on this runtime: launch by launch from here on */
            if (!r) r = ks265_load_i420(fr, e->dev_in[k], srcp);
            if (!r) r = kind == 'B' ? ks265_encode_picture_b(fr, srcp, refp, ref1p, out) : ks265_encode_picture(fr, srcp, refp, 0, out);
            if (!r && e->cfg.calcPsnr) r = ks265_sse_picture(fr, srcp, out, dsse);
            if (!r) r = ks265_frame_pack_compact(fr, e->stg[k], e->cfg.calcPsnr ? dsse : NULL);
            if (keep_it) {
                void *ex = NULL;
                const int rc = ks265_capture_end(cx, &ex);                      /* also ends a capture that failed half-way */
                if (!r && !rc) { memcpy(e->graph[e->ngraph].key, key, sizeof key); e->graph[e->ngraph].exec = ex; ++e->ngraph; r = ks265_graph_launch(cx, ex); }
                else if (!r) {                                                  /* recorded but not instantiated: nothing ran.  Do the picture launch by launch, graphs off */
                    e->use_graph = 0;
                    r = ks265_frame_p_restore(fr, state0);
                    if (!r) r = ks265_load_i420(fr, e->dev_in[k], srcp);
                    if (!r) r = kind == 'B' ? ks265_encode_picture_b(fr, srcp, refp, ref1p, out) : ks265_encode_picture(fr, srcp, refp, 0, out);
                    if (!r && e->cfg.calcPsnr) r = ks265_sse_picture(fr, srcp, out, dsse);
                    if (!r) r = ks265_frame_pack_compact(fr, e->stg[k], e->cfg.calcPsnr ? dsse : NULL);
                }
            }
        }
        if (!r) r = ks265_event_record(cx, e->ev_loaded[k]);                  /* the input buffer is free again (a little later than on the plain path) */
        if (!r) r = ks265_event_record(cx, e->ev_staged[k]);
        e->dpb_poc[slot] = poc;
    } else {
    if (!r) {
        if (kind == 'I') { r = ks265_encode_picture(fr, srcp, out, 1, out); if (!r && fr == e->frame) r = ks265_frame_p_restore(fr, 0); }   /* (see on_key below) */
        else if (kind == 'B' && (nl0 > 1 || nl1 > 1)) {
            ks265_pic r0[4], r1[4];
            for (int i = 0; i < nl0; ++i) r0[i] = e->dpb[dpb_find(e, l0[i])];
            for (int i = 0; i < nl1; ++i) r1[i] = e->dpb[dpb_find(e, l1[i])];
            r = ks265_encode_picture_b_mref(fr, srcp, r0, nl0, r1, nl1, out);
        }
        else if (kind == 'B') r = ks265_encode_picture_b(fr, srcp, e->dpb[dpb_find(e, l0[0])], e->dpb[dpb_find(e, l1[0])], out);
        else if (nl0 > 1) { ks265_pic refs[4]; for (int i = 0; i < nl0; ++i) refs[i] = e->dpb[dpb_find(e, l0[i])]; r = ks265_encode_picture_mref(fr, srcp, refs, nl0, out); }
        else r = ks265_encode_picture(fr, srcp, e->dpb[dpb_find(e, l0[0])], 0, out);
    }
    e->dpb_poc[slot] = poc;
    if (!r && e->cfg.calcPsnr && !split) r = ks265_sse_picture(fr, srcp, out, dsse);
    if (!r && e->recon_on) {
        r = ks265_store_i420(fr, out, e->dev_recon);
        if (!r) r = ks265_memcpy_d2h_async(cx, j->recon, e->dev_recon, fsz);
    }
    if (split) {
        /* the drain runs on the copy-out stream (in order with the earlier pictures' drains: the staging block of this slot has been copied out before it is written) */
        if (!r) r = ks265_event_record(cx, e->ev_coded[k]);
        if (!r) r = ks265_stream_wait_event(e->ctx_out, e->ev_coded[k]);
        if (!r && e->cfg.calcPsnr) r = ks265_sse_picture_on(e->ctx_out, fr, srcp, out, dsse);
        if (!r) r = ks265_frame_pack_compact_on(e->ctx_out, fr, e->stg[k], e->cfg.calcPsnr ? dsse : NULL);
        if (!r) r = ks265_event_record(e->ctx_out, e->ev_packed[k]);
        if (!r) r = ks265_event_record(e->ctx_out, e->ev_staged[k]);
        if (!r) r = ks265_frame_set_records_fence(fr, e->ev_packed[k]);   /* the next picture on this frame object: its search starts now, its first record waits */
    } else {
    /* the records leave the frame object's buffers for a staging set (device to device, a few microseconds), so that the next picture can start
     * while the copy-out stream drains this one */
    if (!r && recycled) r = ks265_stream_wait_event(cx, e->ev_drained[k]);
    if (!r) r = ks265_frame_pack_compact(fr, e->stg[k], e->cfg.calcPsnr ? dsse : NULL);
    if (!r) r = ks265_event_record(cx, e->ev_staged[k]);
    }
    }
    if (on_key) {                                                      /* everything coded after it on the main stream waits for the key picture; its temporal predictors start over */
        if (!r) r = ks265_event_record(cx, e->ev_key);
        if (!r) r = ks265_stream_wait_event(e->ctx, e->ev_key);
        if (!r) r = ks265_frame_p_restore(e->frame, 0);                 /* temporal predictors start over, and so does the ping-pong of the PU record buffers: the graph keys of a GOP's
                                                                         * pictures (which hold that state) are the same in every GOP - nothing is captured after the first one */
        if (!r && e->anc_on) { r = ks265_stream_wait_event(e->ctx_anc, e->ev_key); if (!r) r = ks265_frame_p_restore(e->frame_anc, 0); }   /* the first anchor predicts from it */
        ++e->nkeys;
    }
    if (kind == 'I' && !on_key && e->anc_on) {                         /* a key picture on the main stream (short intra period): the anchors' stream waits for it there */
        if (!r) r = ks265_event_record(e->ctx, e->ev_key);
        if (!r) r = ks265_stream_wait_event(e->ctx_anc, e->ev_key);
        if (!r) r = ks265_frame_p_restore(e->frame_anc, 0);
    }
    e->last_on_anc = on_anc;
    if (on_anc) {                                                      /* the B pictures in front of it (and everything else the main stream codes from here on) wait for the anchor */
        if (!r) r = ks265_event_record(cx, e->ev_anc);
        if (!r) r = ks265_stream_wait_event(e->ctx, e->ev_anc);
        ++e->nanch;
    }
    /* copy-out stream */
    if (!r) r = ks265_stream_wait_event(e->ctx_out, e->ev_staged[k]);
    /* the records go home: the fixed part + the stored lines only (~2 MB for a P picture at 2160p), by a kernel that reads the size on the device (or, KS265_COPYOUT_MB,
     * a fixed amount by hipMemcpyAsync and the rest by that kernel) */
    if (!r) r = e->copy_mb < 0 ? ks265_copy_out_compact_async(e->ctx_out, e->frame, j->cmp, e->stg[k])
                               : ks265_copy_out_compact_dma_async(e->ctx_out, e->frame, j->cmp, e->stg[k], kind == 'I' ? e->cmp_off[7] : (size_t)e->copy_mb << 20);
    if (!r) r = ks265_event_record(e->ctx_out, e->ev_drained[k]);
    if (!r) r = ks265_event_record(e->ctx_out, j->ev);
    if (r) return hip_rc(r);
    ++e->seq;
    j->no_sao = lean != 0;
    j->disp = in->disp; j->pts = in->pts; j->poc = poc; j->kind = kind; j->qp = qp; j->is_ref = is_ref; j->key_headers = key_headers; j->rc_delta = e->rc_qp_delta;
    j->rc_budget = (double)in->kbps * 1000.0 / (e->cfg.frameRate > 0 ? e->cfg.frameRate : 25.0);
    j->nal_type = kind == 'I' ? KS265_NAL_IDR_W_RADL : is_ref ? KS265_NAL_TRAIL_R : KS265_NAL_TRAIL_N;
    j->nl0 = nl0; j->nl1 = nl1;
    for (int i = 0; i < nl0; ++i) j->l0[i] = l0[i];
    for (int i = 0; i < nl1; ++i) j->l1[i] = l1[i];
    /* RPS: everything that must stay (keep_after) + what this picture uses */
    j->nrps = 0;
    if (kind != 'I')
        for (int i = 0; i < nk; ++i) {
            int dup = 0;
            for (int q = 0; q < j->nrps; ++q) if (j->rps_poc[q] == keep[i]) dup = 1;
            if (dup || keep[i] == poc) continue;
            int used = 0;
            for (int q = 0; q < nl0; ++q) if (l0[q] == keep[i]) used = 1;
            for (int q = 0; q < nl1; ++q) if (l1[q] == keep[i]) used = 1;
            j->rps_poc[j->nrps] = keep[i]; j->rps_used[j->nrps++] = (unsigned char)used;
        }
    j->t_submit = now_ms();
    pthread_mutex_lock(&e->mu);
    in->used = 2;                                                      /* released when the job's event has fired (output time); under the lock: the caller counts the pictures in flight */
    j->done = 0; j->error = 0; j->used = 1; j->started = 0; j->nrows = 0; j->next_row = 0; j->rows_done = 0;
    j->sub_seq = e->rc_sub;
    e->job_tail = (e->job_tail + 1) % e->ring; ++e->njobs; ++e->nwait; ++e->rc_sub;
    e->st.occ_samples++; e->st.occ_ring += e->njobs; e->st.occ_gpu += e->nwait; e->st.occ_ready += e->npending;
    pthread_cond_signal(&e->cv_disp);
    pthread_cond_broadcast(&e->cv_sched_done);                         /* progress: a caller waiting for the scheduler (input back-pressure, flush) looks again */
    pthread_mutex_unlock(&e->mu);
    return QY_OK;
}

/* the input slot of display index disp (scheduler thread; under the lock: the caller's thread marks slots while it fills them) */
static Input *input_at(Enc *e, int disp)
{
    Input *r = NULL;
    pthread_mutex_lock(&e->mu);
    for (int i = 0; i < MAX_INPUT && !r; ++i) if (e->in[i].used == 1 && e->in[i].disp == disp) r = &e->in[i];
    pthread_mutex_unlock(&e->mu);
    return r;
}
static int clampqp(Enc *e, int q) { int lo = e->cfg.rc ? e->cfg.qpmin : 0, hi = e->cfg.rc ? (e->cfg.qpmax ? e->cfg.qpmax : 51) : 51; return q < lo ? lo : q > hi ? hi : q; }

/* hierarchical-B mini-GOP: anchor `a` is coded, now the B pictures of (d, a) breadth first; POCs are relative to the GOP's key picture */
static int code_hier(Enc *e, int d, int a)
{
    typedef struct { int lo, hi; } Iv;
    Iv cur[8], nxt[8]; int nc = 1, layer = 1;
    cur[0].lo = d; cur[0].hi = a;
    /* every picture of the mini-GOP that is a reference stays until its interval is done; simplest exact rule: keep all already coded
     * pictures of [d, a] plus d and a themselves (at most 9 with GOP 8) */
    int coded[16], ncoded = 0;
    coded[ncoded++] = d - e->gop_start; coded[ncoded++] = a - e->gop_start;
    while (nc) {
        int nn = 0;
        for (int i = 0; i < nc; ++i) {
            if (cur[i].hi - cur[i].lo < 2) continue;
            const int mid = (cur[i].lo + cur[i].hi) / 2;
            Input *in = input_at(e, mid);
            if (!in) return QY_FAIL;                                   /* (the picture has left the input table: the encoder is being torn down after an error) */
            const int is_ref = (mid - cur[i].lo >= 2) || (cur[i].hi - mid >= 2);
            /* list 0: the nearest pictures before `mid` among those this mini-GOP keeps (all its reference pictures coded so far), nearest first; list 1: those after it.  The
             * interval's ends come first; -ref > 1 adds the next nearest ones */
            int l0[4], l1[4], nl0 = 0, nl1 = 0;
            {
                const int pm = mid - e->gop_start;
                for (int want = 0; want < e->refs_b; ++want) {
                    int b0 = -1000000, b1 = 1000000;
                    for (int q = 0; q < ncoded; ++q) {
                        if (coded[q] < pm && coded[q] > b0 && (nl0 == 0 || coded[q] < l0[nl0 - 1])) b0 = coded[q];
                        if (coded[q] > pm && coded[q] < b1 && (nl1 == 0 || coded[q] > l1[nl1 - 1])) b1 = coded[q];
                    }
                    if (b0 > -1000000) l0[nl0++] = b0;
                    if (b1 < 1000000) l1[nl1++] = b1;
                }
            }
            /* B pictures of the pyramid: + 2 / + 4 / + 4 on the key picture's QP by layer - the reference's own ladder (appencoder -qp 27 -psnr 2: anchors 28, B pictures 29 / 31 / 31; ours was
             * + 2 / + 3 / + 4 until the end of round 3).  Larger offsets keep paying (+ 3 / + 5 / + 6: 1.51 x -> 1.44 x the reference's bitrate at its PSNR-Y on the 1080p clip, every B picture within 0.1 dB
             * of the anchors - their quality comes from their references), but -qp would no longer mean what it means in the reference */
            static const int kHierLayerQp[4] = {0, 1, 3, 3}, kPyr4LayerQp[4] = {0, 1, 2, 2};                       /* (-bframes 3: + 2 / + 3) */
            /* (the adaptive GOP's blocks of 4 keep + 2 / + 4: the reference's + 2 / + 3 there cost 1.3 % more bytes for + 0.004 dB on the 2160p clip, measured on the GPU at the end of round 4) */
            const int *lq = e->gop_b == 3 ? kPyr4LayerQp : kHierLayerQp;
            /* -ref0: the anchors the NEXT anchor searches besides d and a stay in this picture's reference picture set (they are in no list of it: the lists above are built
             * from the mini-GOP's own pictures) */
            int keepx[24], nkx = 0;
            for (int q = 0; q < ncoded; ++q) keepx[nkx++] = coded[q];
            for (int q = 2; q < e->refs0 && q < e->n_anc; ++q) keepx[nkx++] = e->anc_hist[q];
            int r = submit(e, in, 'B', mid - e->gop_start, clampqp(e, in->base_qp + e->rc_qp_delta + (e->fixqp ? 0 : 1 + lq[layer < 3 ? layer : 3])), l0, nl0, l1, nl1, keepx, nkx, is_ref, 0);
            if (r) return r;
            if (is_ref) coded[ncoded++] = mid - e->gop_start;
            nxt[nn].lo = cur[i].lo; nxt[nn++].hi = mid; nxt[nn].lo = mid; nxt[nn++].hi = cur[i].hi;
        }
        memcpy(cur, nxt, sizeof cur); nc = nn; ++layer;
    }
    return QY_OK;
}

/* schedule whatever can be coded with the pictures received so far; flush = no more input will come */
static int schedule(Enc *e, int flush, int have /* pictures [0, have) have arrived */, int gop_end /* a GOP ends at this picture whatever follows, or -1 */)
{
    for (;;) {
        const int d = e->coded_upto;                                   /* last anchor / last coded display index; -1 before the first picture */
        if (d + 1 >= have) return QY_OK;
        const int nxt = d + 1;
        Input *in = input_at(e, nxt);
        const int iper = in ? in->iper : 0;                            /* the period in force when this picture was handed in (QY265EncoderReconfig) */
        const int key = d < 0 || (iper > 0 && nxt - e->gop_start >= iper) || (in && in->key);
        if (key && e->ct_on) {                                         /* cuTree: the key picture's offsets need the window behind it */
            const int gs = e->gop_start; e->gop_start = nxt;
            const int end = ct_window_end(e, nxt, iper, have, flush, gop_end);
            e->gop_start = gs;
            if (end < 0) return QY_OK;
            e->gop_start = nxt;
            const int rr = ct_run(e, nxt, nxt, nxt, end);
            e->gop_start = gs;
            if (rr) return hip_rc(rr);
        }
        if (key) {
            if (!in) return QY_FAIL;
            e->gop_start = nxt; e->mg4_until = -1;
            e->anc_hist[0] = 0; e->n_anc = 1;                           /* the key picture is the GOP's first anchor (POC 0) */
            e->rc_qp_delta = rc_decide(e);                             /* rate control: one offset per key picture / mini-GOP, decided when it is certain to be submitted */
            for (int i = 0; i < e->ndpb + 2 + 4; ++i) e->dpb_poc[i] = -1000000;
            int r = submit(e, in, 'I', 0, clampqp(e, in->base_qp + e->rc_qp_delta), NULL, 0, NULL, 0, NULL, 0, 1, 1);
            if (r) return r;
            pthread_mutex_lock(&e->mu); e->coded_upto = nxt; pthread_mutex_unlock(&e->mu);
            continue;
        }
        int span = e->gop_b + 1;                                       /* anchor distance */
        if (e->mg_adapt && span == 8) {                                /* slice-type decision (lane_put): this block of 8 as two mini-GOPs of 4 */
            if (d < e->mg4_until) span = 4;                            /* its second half */
            else {
                const Input *i8 = d + 8 < have ? input_at(e, d + 8) : NULL;   /* the decision travels with the block's last picture; not there yet: nothing is coded before it arrives (or a key picture / the flush cuts the block short) */
                if (i8 && i8->mini4 && !i8->key) { span = 4; e->mg4_until = d + 8; }
            }
        }
        int a = d + span;
        if (iper > 0 && a - e->gop_start >= iper) a = e->gop_start + iper - 1;   /* the mini-GOP in front of a key picture is shortened */
        for (int k = nxt + 1; k <= a && k < have; ++k) {                 /* a picture asked to be a key picture: the mini-GOP in front of it is shortened as well */
            const Input *ik = input_at(e, k);
            if (ik && ik->key) { a = k - 1; break; }
        }
        if (gop_end >= nxt && a > gop_end) a = gop_end;                 /* the GOP was closed behind this picture (its successor goes to another lane) */
        if (a >= have) { if (!flush) return QY_OK; a = have - 1; }
        if (e->ct_on) {                                                /* cuTree: costs and propagation over the window behind this mini-GOP, offsets for its pictures */
            const int end = ct_window_end(e, a, iper, have, flush, gop_end);
            if (end < 0) return QY_OK;
            const int rr = ct_run(e, -1, d, a, end);
            if (rr) return hip_rc(rr);
        }
        const int pd = d - e->gop_start, pa = a - e->gop_start;
        int l0[4], nl0 = 0, keep[8], nkeep = 0;
        if (span == 1) {                                               /* IPPP: the most recent pictures, nearest first */
            for (int i = 0; i < e->refs && pa - 1 - i >= 0; ++i) l0[nl0++] = pa - 1 - i;
            for (int i = 0; i < e->refs - 1 && pa - 1 - i >= 0; ++i) keep[nkeep++] = pa - 1 - i;   /* still needed by the next picture */
        } else if (e->refs0 > 1 && e->n_anc > 0 && e->anc_hist[0] == pd) {
            /* -ref0: the last anchors of this GOP, nearest first (the first one is the previous anchor); all of them but the oldest are the next anchor's too */
            for (int i = 0; i < e->refs0 && i < e->n_anc; ++i) l0[nl0++] = e->anc_hist[i];
            for (int i = 0; i < e->refs0 - 1 && i < e->n_anc; ++i) keep[nkeep++] = e->anc_hist[i];
        } else { l0[nl0++] = pd; keep[nkeep++] = pd; }
        Input *ina = input_at(e, a);
        if (!ina) return QY_FAIL;                                      /* (as above: an error elsewhere emptied the input table under the scheduler) */
        e->rc_qp_delta = rc_decide(e);
        /* the QP ladder of P pictures: + 1 on the key picture's; IPPP: the reference's own cascade over four pictures (appencoder -bframes 0 -qp 27 -psnr 2: 30 / 29 / 30 / 28 / 30 ..),
         * measured with the CPU mirror of this host: - 12 % bytes of the P pictures for - 0.09 dB */
        static const int kIpppCascade[4] = {0, 2, 1, 2};
        const int casc = e->gop_b == 0 ? kIpppCascade[pa & 3] : 0;
        int r = submit(e, ina, 'P', pa, clampqp(e, ina->base_qp + e->rc_qp_delta + (e->fixqp ? 0 : 1 + casc)), l0, nl0, NULL, 0, keep, nkeep, 1, 0);
        if (r) return r;
        if (span > 1) {                                                /* the anchors' history: this one in front */
            for (int i = 3; i > 0; --i) e->anc_hist[i] = e->anc_hist[i - 1];
            e->anc_hist[0] = pa; if (e->n_anc < 4) ++e->n_anc;
        }
        const int anchor_on_lane = e->last_on_anc;
        if (a - d > 1) {
            if (e->hier && ((a - d) & (a - d - 1)) == 0) { r = code_hier(e, d, a); if (r) return r; }
            else {
                int kp[6] = {pd, pa}, nkp = 2;
                for (int i = 2; i < e->refs0 && i < e->n_anc; ++i) kp[nkp++] = e->anc_hist[i];      /* (-ref0: what the next anchor still searches) */
                for (int b = d + 1; b < a; ++b) {
                    Input *inb = input_at(e, b);
                    if (!inb) return QY_FAIL;
                    r = submit(e, inb, 'B', b - e->gop_start, clampqp(e, inb->base_qp + e->rc_qp_delta + (e->fixqp ? 0 : 2)), &pd, 1, &pa, 1, kp, nkp, 0, 0);
                    if (r) return r;
                }
            }
        }
        if (e->anc_on && anchor_on_lane) {   /* every B picture that reads the anchors of this mini-GOP is on the main stream now */
            r = hip_rc(ks265_event_record(e->ctx, e->ev_mg[(e->nanch - 1) & 7]));
            if (r) return r;
        }
        pthread_mutex_lock(&e->mu); e->coded_upto = a; pthread_mutex_unlock(&e->mu);
    }
}

/* ---- the scheduler thread: runs schedule() whenever pictures have arrived (or a flush was asked for).  The caller's thread only copies the input
 *      picture and collects output; this thread takes the GOP decisions and enqueues the GPU work of every picture. */
static int la_drain(Enc *e, int keep);
static void reg_handle(ks265_ctx *c, int open);
static void *scheduler(void *arg)
{
    Enc *e = (Enc *)arg;
    pthread_mutex_lock(&e->mu);
    for (;;) {
        while (!e->quit && e->sched_seen == e->next_disp && !e->sched_flush && e->gop_end_seen == e->gop_end) {
            e->sched_idle = 1; pthread_cond_broadcast(&e->cv_sched_done);
            if (e->la_on && __atomic_load_n(&e->la_qn, __ATOMIC_ACQUIRE) > 0) {
                /* pictures sit at the input until their lookahead results have arrived, and the caller only looks when it hands in the next picture - it may be asleep (its
                 * back-pressure, the output ring): with nothing to schedule this thread looks as well */
                struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
                ts.tv_nsec += 200000; if (ts.tv_nsec >= 1000000000) { ts.tv_nsec -= 1000000000; ++ts.tv_sec; }
                pthread_cond_timedwait(&e->cv_sched, &e->mu, &ts);
                if (e->quit || e->sched_seen != e->next_disp || e->sched_flush) continue;
                pthread_mutex_unlock(&e->mu);
                (void)la_drain(e, LA_QMAX);                             /* (never waits: the queue's size) */
                pthread_mutex_lock(&e->mu);
                if (e->sched_seen != e->next_disp) ++e->la_n_poll;
            } else pthread_cond_wait(&e->cv_sched, &e->mu);
        }
        if (e->quit) break;
        e->sched_idle = 0;
        const int flush = e->sched_flush, have = e->next_disp, gop_end = e->gop_end;
        pthread_mutex_unlock(&e->mu);
        const double t0 = now_ms();
        const int r = schedule(e, flush, have, gop_end);
        const double dt = now_ms() - t0;
        pthread_mutex_lock(&e->mu);
        e->st.submit_ms += dt;
        if (r && !e->sched_err) e->sched_err = r;
        e->sched_seen = have; e->gop_end_seen = gop_end;
        if (flush && have == e->next_disp) e->sched_flush = 0;          /* everything that had arrived is scheduled */
    }
    e->sched_idle = 1;
    pthread_cond_broadcast(&e->cv_sched_done);
    pthread_mutex_unlock(&e->mu);
    return NULL;
}

/* move finished pictures (in coding order) to the output array; wait while more than `max_in_flight` pictures are queued
 * (0 = drain everything, MAX_JOBS = never wait) */
static int take_output(Enc *e, int max_in_flight, int max_pics, QY265Nal **pNals, int *n, QY265Picture *out, int *pics)
{
    int cnt = 0, err = QY_OK, taken = 0;
    e->outpos = 0;
    pthread_mutex_lock(&e->mu);
    while (e->njobs && taken < max_pics) {
        Job *j = &e->jobs[e->job_head];
        if (!j->done) { if (e->njobs <= max_in_flight) break; pthread_cond_wait(&e->cv_done, &e->mu); continue; }
        if (j->error) err = hip_rc(j->error);
        const size_t need = (size_t)(j->nal_len > 0 ? j->nal_len : 0);
        if (e->outpos + need > e->outcap) {                             /* the caller reads the payloads after this call: they cannot stay in the job (its slot is reused) */
            if (cnt) break;                                             /* hand out what fits, the rest next time */
            uint8_t *nb = (uint8_t *)realloc(e->outbuf, need + 65536);
            if (!nb) { err = QY_OUTOFMEMORY; break; }
            e->outbuf = nb; e->outcap = need + 65536;
        }
        if (j->key_headers && e->cfg.bHeaderBeforeKeyframe) {
            long off = 0;                                               /* the three parameter sets as separate NAL entries */
            for (int k = 0; k < 3; ++k) {
                e->nals[cnt].naltype = KS265_NAL_VPS + k; e->nals[cnt].tid = 0; e->nals[cnt].iSize = (int)e->hdr_part[k]; e->nals[cnt].pts = j->pts; e->nals[cnt].pPayload = e->hdr + off;
                ++cnt; off += e->hdr_part[k];
            }
        }
        memcpy(e->outbuf + e->outpos, j->nal, need);
        e->nals[cnt].naltype = j->nal_type; e->nals[cnt].tid = 0; e->nals[cnt].iSize = (int)need; e->nals[cnt].pts = j->pts; e->nals[cnt].pPayload = e->outbuf + e->outpos;
        e->outpos += need;
        ++cnt;
        if (out) { out->iSliceType = j->kind == 'I' ? 2 : j->kind == 'P' ? 1 : 0; out->poc = j->disp; out->pts = j->pts; out->dts = j->pts; }
        e->st.frames++; e->st.bytes += j->nal_len > 0 ? j->nal_len : 0; e->st.host_write_ms += j->t_write_ms;
        e->st.lat_gpu_ms += j->t_event - j->t_submit; e->st.lat_queue_ms += j->t_taken - j->t_submit;
        if (j->kind == 'I') { e->st.key_wall_ms += j->t_done - j->t_event; e->st.key_cpu_ms += j->t_write_ms; e->st.keys++; }
        if (j->key_headers && e->cfg.bHeaderBeforeKeyframe) e->st.bytes += e->hdr_len;
        if (e->cfg.calcPsnr) {
            for (int k = 0; k < 3; ++k) e->st.sse[k] += (double)j->sse[k];
            if (e->cfg.calcPsnr >= 2) {
                const double np[3] = {(double)e->W * e->H, (double)e->W * e->H / 4, (double)e->W * e->H / 4};
                double ps[3];
                for (int k = 0; k < 3; ++k) ps[k] = j->sse[k] ? 10.0 * log10(255.0 * 255.0 * np[k] / (double)j->sse[k]) : 99.0;
                /* appencoder's -psnr 2 table: `poc slice bits psnrY psnrU psnrV qp`, tab separated, one header line */
                if (!e->psnr_hdr) { e->psnr_hdr = 1; logf_(2, e->log_level, "poc\tslice\tbits\tpsnr\t\t\tqp\n"); }
                logf_(2, e->log_level, "%d\t%c\t%ld\t%.4f\t%.4f\t%.4f\t%d\n", j->disp, j->kind, (long)j->nal_len * 8, ps[0], ps[1], ps[2], j->qp);
            }
        }
        if (e->md5 && e->md5_ring) {                                    /* `POC n MD5 y,u,v` in display order, like the reference's reconstruction output */
            memcpy(e->md5_ring[j->disp & 255], j->md5, sizeof j->md5); e->md5_have[j->disp & 255] = 1;
            while (e->md5_have[e->md5_next & 255]) {
                char (*m)[33] = e->md5_ring[e->md5_next & 255];
                logf_(2, e->log_level, "POC %d MD5 %s,%s,%s\n", e->md5_next, m[0], m[1], m[2]);
                e->md5_have[e->md5_next & 255] = 0; ++e->md5_next;
            }
        }
        rc_account(e);                                                  /* rate control: this picture's bits are on the books before its slot is reused */
        for (int i = 0; i < MAX_INPUT; ++i) if (e->in[i].used == 2 && e->in[i].disp == j->disp) e->in[i].used = 0;
        j->used = 0;
        e->job_head = (e->job_head + 1) % e->ring; --e->njobs; ++taken;
        if (cnt >= (int)(sizeof e->nals / sizeof e->nals[0]) - 4) break;
    }
    if (taken) pthread_cond_broadcast(&e->cv_done);                    /* the scheduler thread may be waiting for ring space */
    pthread_mutex_unlock(&e->mu);
    *pNals = e->nals; *n = cnt;
    if (pics) *pics = taken;
    return err;
}

static void lane_close(Enc *e, int report)
{
    if (!e) return;
    if (e->nth || e->disp_on || e->sched_on) {
        pthread_mutex_lock(&e->mu); e->quit = 1; pthread_cond_broadcast(&e->cv_work); pthread_cond_broadcast(&e->cv_disp); pthread_cond_broadcast(&e->cv_sched); pthread_cond_broadcast(&e->cv_done); pthread_mutex_unlock(&e->mu);
        if (e->sched_on) pthread_join(e->sched, NULL);
        for (int i = 0; i < e->nth; ++i) pthread_join(e->th[i], NULL);
        if (e->disp_on) pthread_join(e->disp, NULL);
    }
    if (e->ctx) {
        if (e->ctx_in) ks265_synchronize(e->ctx_in);
        ks265_synchronize(e->ctx);
        if (e->ctx_out) ks265_synchronize(e->ctx_out);
        if (report && e->la_on) logf_(1, e->log_level, "ks265enc: lookahead: %ld scene cuts, %ld blocks of 8 pictures coded as 4 + 4\n", e->la_cuts, e->la_mini4);
        if (report && e->la_on && e->in_disp) logf_(1, e->log_level, "ks265enc: lookahead, caller's ms per picture: launches + queue %.3f, of it waiting for results %.3f (%ld waits); pictures handed on by the scheduler thread: %ld\n",
                                                    e->la_t_take / e->in_disp, e->la_t_wait / e->in_disp, e->la_n_wait, e->la_n_poll);
        if (report && e->in_disp) logf_(1, e->log_level, "ks265enc: caller's back-pressure wait %.3f ms per picture\n", e->la_t_bp / e->in_disp);
        if (report && e->cfg.calcPsnr && e->st.frames) {
            const double np[3] = {(double)e->W * e->H, (double)e->W * e->H / 4, (double)e->W * e->H / 4};
            double ps[3];
            for (int k = 0; k < 3; ++k) ps[k] = e->st.sse[k] > 0 ? 10.0 * log10(255.0 * 255.0 * np[k] * e->st.frames / e->st.sse[k]) : 99.0;
            logf_(2, e->log_level, "bitrate, psnr: %.4f\t%.4f\t%.4f\t%.4f\n", e->st.bytes * 8.0 * e->cfg.frameRate / e->st.frames / 1000.0, ps[0], ps[1], ps[2]);
        }
        for (int i = 0; i < MAX_JOBS; ++i) {
            Job *j = &e->jobs[i];
            ks265_host_free(e->ctx, j->cmp); ks265_host_free(e->ctx, j->recon); ks265_host_free(e->ctx, j->qp_map); ks265_host_free(e->ctx, j->rq_host); free(j->lvlbuf); free(j->dirty);
            if (j->ev) ks265_event_destroy(e->ctx, j->ev);
            free(j->nal);
        }
        for (int i = 0; i < MAX_INPUT; ++i) { ks265_host_free(e->ctx, e->in[i].i420); if (e->in[i].dev) ks265_dev_free(e->ctx, e->in[i].dev); if (e->in[i].ev_up) ks265_event_destroy(e->ctx_up, e->in[i].ev_up); }
        for (int i = 0; i < e->ndpb + 2; ++i) pic_free(e, &e->dpb[i]);
        for (int i = 0; i < e->nspacer; ++i) ks265_dev_free(e->ctx, e->spacer[i]);
        pic_free(e, &e->src_key); ks265_dev_free(e->ctx, e->dev_sse_key);
        if (e->ev_key) ks265_event_destroy(e->ctx, e->ev_key);
        for (int i = 0; i < 2; ++i) if (e->ev_firstp[i]) ks265_event_destroy(e->ctx, e->ev_firstp[i]);
        if (e->ctx_key) ks265_synchronize(e->ctx_key);
        if (e->frame_key) ks265_frame_destroy(e->frame_key);
        if (e->ctx_anc) ks265_synchronize(e->ctx_anc);
        if (e->frame_anc) ks265_frame_destroy(e->frame_anc);
        pic_free(e, &e->src_anc); ks265_dev_free(e->ctx, e->dev_sse_anc);
        if (e->ev_anc) ks265_event_destroy(e->ctx, e->ev_anc);
        for (int i = 0; i < 8; ++i) if (e->ev_mg[i]) ks265_event_destroy(e->ctx, e->ev_mg[i]);
        for (int i = 0; i < 4; ++i) pic_free(e, &e->dpb[e->ndpb + 2 + i]);
        pic_free(e, &e->src);
        for (int k = 0; k < NPIPE; ++k) {
            ks265_dev_free(e->ctx, e->dev_in[k]); ks265_dev_free(e->ctx, e->stg[k]);
            if (e->ev_h2d[k]) ks265_event_destroy(e->ctx, e->ev_h2d[k]);
            if (e->ev_loaded[k]) ks265_event_destroy(e->ctx, e->ev_loaded[k]);
            if (e->ev_staged[k]) ks265_event_destroy(e->ctx, e->ev_staged[k]);
            if (e->ev_drained[k]) ks265_event_destroy(e->ctx, e->ev_drained[k]);
            if (e->ev_coded[k]) ks265_event_destroy(e->ctx, e->ev_coded[k]);
            if (e->ev_packed[k]) ks265_event_destroy(e->ctx, e->ev_packed[k]);
            pic_free(e, &e->srcq[k]);
        }
        for (int i = 0; i < e->ngraph; ++i) ks265_graph_destroy(e->ctx, e->graph[i].exec);
        ks265_dev_free(e->ctx, e->dev_sse); ks265_dev_free(e->ctx, e->dev_recon);
        for (int q = 0; q < 2; ++q) { ks265_dev_free(e->ctx, e->aq_off[q]); ks265_dev_free(e->ctx, e->aq_inv[q]); ks265_dev_free(e->ctx, e->aq_scratch[q]); }
        for (int k = 0; k < NPIPE; ++k) ks265_dev_free(e->ctx, e->dev_qmap[k]);
        for (int q = 0; q < 2; ++q) ks265_dev_free(e->ctx, e->dev_qmap_key[q]);
        if (e->recon_fd >= 0) close(e->recon_fd);
        if (e->qmap_fd >= 0) close(e->qmap_fd);
        ct_close(e);
        if (e->ctx_la) {
            ks265_synchronize(e->ctx_la);
            for (int i = 0; i < LA_RING; ++i) { ks265_dev_free(e->ctx_la, e->la_pic[i].y); ks265_dev_free(e->ctx_la, e->la_pic[i].u); ks265_dev_free(e->ctx_la, e->la_pic[i].v); }
            ks265_dev_free(e->ctx_la, e->la_cost_ws); ks265_dev_free(e->ctx_la, e->la_dev_out); ks265_host_free(e->ctx_la, e->la_host_out);
            for (int i = 0; i < LA_FLY; ++i) if (e->la_evs[i]) ks265_event_destroy(e->ctx_la, e->la_evs[i]);
            if (e->frame_la) ks265_frame_destroy(e->frame_la);
            ks265_destroy(e->ctx_la);
        }
        if (e->ctx_up) reg_handle(e->ctx, 0);
        if (e->ctx_upl) { ks265_synchronize(e->ctx_upl); ks265_destroy(e->ctx_upl); }
        if (e->frame) ks265_frame_destroy(e->frame);
        if (e->ctx_key) ks265_destroy(e->ctx_key);
        if (e->ctx_anc) ks265_destroy(e->ctx_anc);
        if (e->ctx_in) ks265_destroy(e->ctx_in);
        if (e->ctx_out) ks265_destroy(e->ctx_out);
        ks265_destroy(e->ctx);
    } else { if (e->ctx_la) ks265_destroy(e->ctx_la); if (e->ctx_upl) ks265_destroy(e->ctx_upl); }                   /* created first (lane_open), before the main context failed: nothing else of the lookahead exists yet */
    for (int i = 0; i < MAX_JOBS; ++i) free(e->jobs[i].wpp);
    free(e->hdr); free(e->outbuf); free(e->md5_ring); free(e->md5_have); free(e->rq_hist);
    pthread_mutex_destroy(&e->la_mu);
    pthread_mutex_destroy(&e->mu); pthread_cond_destroy(&e->cv_work); pthread_cond_destroy(&e->cv_done); pthread_cond_destroy(&e->cv_disp); pthread_cond_destroy(&e->cv_sched); pthread_cond_destroy(&e->cv_sched_done);
    free(e);
}

static Enc *lane_open(QY265EncConfig *cfg, int device, int multi, int *err)
{
    int dummy; if (!err) err = &dummy;
    *err = QY_OK;
    if (!cfg) { *err = QY_POINTER; return NULL; }
    if (cfg->picWidth <= 0 || cfg->picHeight <= 0 || (cfg->picWidth & 7) || (cfg->picHeight & 7) || cfg->frameRate <= 0 || cfg->rc < 0 || cfg->rc > 5) { *err = QY_NOTSUPPORTED; return NULL; }
    Enc *e = (Enc *)calloc(1, sizeof *e);
    if (e) { e->recon_fd = -1; e->qmap_fd = -1; }
    if (!e) { *err = QY_OUTOFMEMORY; return NULL; }
    pthread_mutex_init(&e->la_mu, NULL);
    pthread_mutex_init(&e->mu, NULL); pthread_cond_init(&e->cv_work, NULL); pthread_cond_init(&e->cv_done, NULL); pthread_cond_init(&e->cv_disp, NULL); pthread_cond_init(&e->cv_sched, NULL); pthread_cond_init(&e->cv_sched_done, NULL);
    e->cfg = *cfg; e->W = cfg->picWidth; e->H = cfg->picHeight; e->log_level = cfg->logLevel;
    e->me_method = cfg->me < 0 ? 1 : cfg->me > 2 ? 2 : cfg->me;        /* EPZS / Cross (-me 3 / 4) are not built: UMH instead */
    e->hex_thr = (e->me_method == 2 && (cfg->preset == QY265PRESET_SLOW || cfg->preset == QY265PRESET_SLOWER)) ? 16 : 0;   /* tME+0x368, SURVEY-measured */
    e->subme = cfg->subme < 0 ? 1 : cfg->subme > 2 ? 2 : cfg->subme;      /* 0 off, 1 fast, 2 square full (qy265enc.h:137): the reference's refinement, ks265_frame_cfg.subme */
    e->refs = cfg->refnum < 1 ? 1 : cfg->refnum > 4 ? 4 : cfg->refnum;
    e->use_sao = cfg->sao > 0; e->use_df = g_cli.df; e->fixqp = g_cli.fixqp; e->md5 = g_cli.md5;
    e->gop_b = cfg->bframes < 0 ? (cfg->latency == QY265LATENCY_DEFAULT ? 7 : 0) : cfg->bframes;
    /* the pyramid: the SDK's default GOP (8), and - as in the reference, whose -psnr 2 lines show it - explicit -bframes 7 (the same) and -bframes 3: anchors 4 apart, the middle
     * picture a reference B at Q + 2, the two outer ones non-reference B at Q + 3 (appencoder -bframes 3 -qp 27: 28 / 29 / 30 / 30; until round 4 ours was P + 3 plain B at Q + 2).
     * -bframes 1 / 2: the reference codes both as anchors 2 apart with one B picture at Q + 2; ours: P + n plain B at Q + 2 */
    e->hier = e->gop_b == 7 || cfg->bframes == 3;
    /* with B pictures the anchors keep one reference; round 5: the B pictures of the pyramid search up to -ref pictures per list (ks265_encode_picture_b_mref: config 5 = -preset
     * veryslow resolves to 4 / 4) - of the pictures the mini-GOP keeps anyway (code_hier), so the reference picture sets do not change */
    e->refs_b = e->hier ? e->refs : 1;
    if (e->gop_b > 0) e->refs = 1;
    /* round 6: -ref0 (-preset slow resolves to ref 1 / ref0 3): the anchors of a pyramid search the last ref0 anchors of their GOP (ks265_encode_picture_mref, list 0
     * nearest first = the default list construction); the older anchors stay in every reference picture set in between.  KS265_REF0 overrides (experiments: 1 = round 5's anchors) */
    e->refs0 = e->hier ? (cfg->ref0 < 1 ? 1 : cfg->ref0 > 4 ? 4 : cfg->ref0) : 1;
    if (getenv("KS265_REF0") && e->hier) { const int v = atoi(getenv("KS265_REF0")); e->refs0 = v < 1 ? 1 : v > 4 ? 4 : v; }
    e->base_qp = cfg->rc == 3 ? cfg->crf : cfg->qp;
    if (e->base_qp < 0) e->base_qp = 0;
    if (e->base_qp > 51) e->base_qp = 51;
    e->iper = cfg->iIntraPeriod;
    e->coded_upto = -1; e->gop_end = e->gop_end_seen = -1;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    e->nthreads = cfg->threads > 0 ? cfg->threads : (int)(ncpu > 0 ? ncpu : 4);
    if (e->nthreads > 64) e->nthreads = 64;

    e->rdoq_on = cfg->rdoq == 2;                                      /* asked for by name (QY265ConfigParse); the presets' rdoq 1 = this build's seam */
    if (e->rdoq_on) logf_(2, e->log_level, "ks265enc: -rdoq 1: the luma transform blocks of inter CUs go through the reference's rdoQuant (bit tables from the context states of the stream, %d pictures behind)\n", RC_LAG + 1);
    if (cfg->transskip) logf_(1, e->log_level, "ks265enc: transskip is accepted but not implemented by the pixel path\n");
    if (cfg->tuInter > 1) logf_(1, e->log_level, "ks265enc: -intertu %d runs as -intertu 1 (the residual quadtree of inter CUs one level deep)\n", cfg->tuInter);
    if (cfg->tuIntra > 0) logf_(2, e->log_level, "ks265enc: -intratu is accepted but not implemented (intra CUs carry one transform unit)\n");
    if (cfg->iAqMode > 1) logf_(1, e->log_level, "ks265enc: -aq %d runs as -aq 1 (block variance, the mode of the reference's calcFrameAdaptQuant)\n", cfg->iAqMode);
    if (cfg->part && e->refs > 1) logf_(1, e->log_level, "ks265enc: -part 1 (2NxN / Nx2N partitions of 64 / 32 / 16 CUs) acts on P and B pictures with one reference picture per list; multi-reference P pictures keep 2Nx2N\n");
    /* options whose VALUE is narrowed (SURVEY.md 8(a) config 5 = -preset veryslow: subme 2, part 1, ref 4): said once, never silently */
    if (cfg->refnum > 4) logf_(1, e->log_level, "ks265enc: -ref %d runs as -ref 4\n", cfg->refnum);
    if (e->gop_b > 0 && cfg->refnum > 1 && !e->hier) logf_(1, e->log_level, "ks265enc: -ref %d with plain (non-pyramid) B pictures runs as one reference per list\n", cfg->refnum);
    if (e->hier && e->refs_b > 1) logf_(1, e->log_level, "ks265enc: B pictures search up to %d pictures per list (list 0: the nearest coded pictures before, list 1: after)\n", e->refs_b);
    if (e->hier && cfg->ref0 > 4) logf_(1, e->log_level, "ks265enc: -ref0 %d runs as -ref0 4\n", cfg->ref0);
    if (!e->hier && e->gop_b > 0 && cfg->ref0 > 1) logf_(1, e->log_level, "ks265enc: -ref0 %d with plain (non-pyramid) B pictures runs as one reference for the anchors\n", cfg->ref0);
    if (e->hier && e->refs0 > 1) logf_(2, e->log_level, "ks265enc: the anchors of the pyramid search the last %d anchors (-ref0)\n", e->refs0);
    if (cfg->rc == 5 || cfg->vbv_buffer_size) logf_(1, e->log_level, "ks265enc: CVQ / VBV are not implemented; running the plain controller\n");

    /* the SDK's config has no device field: the lane's GPU comes from the handle (KS265_DEVICE: one GPU, default 0; KS265_GPUS / KS265_DEVICES: closed GOPs dealt
     * to lanes on several GPUs, QY265EncoderOpen) */
    /* The ORDER in which the streams are created matters (round 4, measured at 2160p; DESIGN 6c, tools/hw_queues.hip): a process's normal-priority streams are served by a few
     * in-order hardware queues (four; three where another runtime user - torch in bench.py - holds one) and share them in pairs that depend on the creation order; a stream that
     * shares the pixel path's queue runs in turn with it.  With the lookahead there is a fifth stream: created fourth, in front of the key pictures' stream, it cost 25 % (hierarchical
     * B) / 31 % (IPPP) with not one lookahead kernel launched; created last 8 %; created FIRST nothing measurable - it then pairs with the copy-out stream.  (In the CLI's own
     * process the four normal streams have a queue each; the key pictures' stream is high-priority and has its own.) */
    const int la_wanted = e->cfg.lookahead > 0 || (e->cfg.lookahead < 0 && e->hier && e->gop_b == 7 && !getenv("KS265_NO_AUTO_LOOKAHEAD"));
    const int la_order = getenv("KS265_LA_ORDER") ? atoi(getenv("KS265_LA_ORDER")) : 0;      /* (experiments: 1 = fourth, 2 = last) */
    if (!getenv("KS265_INPUT_COPY") && getenv("KS265_UPL_ORDER") && atoi(getenv("KS265_UPL_ORDER")) == -1 && (size_t)e->W * e->H * 3 / 2 >= ((size_t)1 << 20)) { if (ks265_create(&e->ctx_upl, device)) e->ctx_upl = NULL; }
    if (la_wanted && la_order == 0 && ks265_create(&e->ctx_la, device)) e->ctx_la = NULL;
    /* (not with KS265_GRAPH: the host-synchronous copy out of the caller's memory runs on the runtime's null stream from the CALLING thread, and while another thread has a capture open
     * that fails - 4 of 6 runs of the 2160p default GOP ended in QY_FAIL, 6 of 6 pass with the copying path; found and fixed at the end of round 6) */
    e->direct_in = !getenv("KS265_INPUT_COPY") && !getenv("KS265_GRAPH") && (size_t)e->W * e->H * 3 / 2 >= ((size_t)1 << 20) && !(cfg->latency == QY265LATENCY_ZERO && !multi);
    e->hold_in = getenv("KS265_INPUT_HOLD") && atoi(getenv("KS265_INPUT_HOLD")) > 0;
    /* the uploads' stream: created first, like the lookahead's.  Its own even beside the lookahead's: the caller waits for the upload, and behind the analysis kernels of the picture
     * before (measured: 1.17 ms per call instead of the 0.3 ms the DMA takes) it would wait for those too */
    /* how the upload out of the caller's memory runs.  4 (default): host-synchronous, on no stream of the lane's (ks265_memcpy_h2d_sync): the call returns when the picture is on the device.
     * 0 / -1 / 1 / 2 (experiments): asynchronously on a stream of its own created first / in front of the lookahead's / after the copy streams / last, waited for at the end of the call.
     * Measured at 2160p (profiles/r06_input_upload.txt): ANY stream more costs the two-lane default GOP a third of its rate (754 -> 515 .. 600 pictures/s: hardware queues are shared in
     * creation order) although the caller's input time falls to 0.33 ms; without one the default GOP codes 777 (copying path 758) and IPPP 1 028 (1 025), input 0.88 / 0.26 ms */
    const int upl_order = getenv("KS265_UPL_ORDER") ? atoi(getenv("KS265_UPL_ORDER")) : 4;
    if (e->direct_in && upl_order == 0 && ks265_create(&e->ctx_upl, device)) { e->ctx_upl = NULL; e->direct_in = 0; }
    int r = ks265_create(&e->ctx, device);
    if (r) { *err = hip_rc(r); lane_close(e, 0); return NULL; }       /* KS265_NO_DEVICE -> QY_FAIL: there is no CPU fallback */
    memset(&e->fcfg, 0, sizeof e->fcfg);
    e->fcfg.width = e->W; e->fcfg.height = e->H; e->fcfg.qp = e->base_qp; e->fcfg.lambda_q4 = kLambdaQ4[e->base_qp];
    e->fcfg.me_range = cfg->searchrange < 1 ? 64 : cfg->searchrange > 64 ? 64 : cfg->searchrange;
    e->fcfg.me_method = e->me_method; e->fcfg.subme = e->subme; e->fcfg.deblock = e->use_df;
    /* -sao (qy265enc.h:143: 1 / 2 faster, 3 usual, 4 complex): 3 given BY NAME = the reference's own decision on its -sao 4 path - band offset + the 0 / 90 degree edge classes priced by its pinned
     * estimation functions, rates and lambda table (ks265_frame_cfg.sao = 2: CEncSao::modeDecisionBoEo01 enc@0x4af300 without the merge candidates); every other level > 0 = this build's
     * rule over all four edge classes + band offset (the presets' -sao 4: 2.7 - 4.6 % fewer bytes at equal PSNR-Y than level 3, which buys 0.8 - 1.9 dB of chroma: DESIGN.md) */
    e->fcfg.sao = cfg->sao == 5 ? 2 : e->use_sao;                       /* (5 = -sao 3 BY NAME, QY265ConfigParse; the presets' 3 is the build's rule) */
    {   /* the sub-pel refinement's knobs follow the preset, as in the reference */
        const int ps = (int)cfg->preset < 0 || (int)cfg->preset > 8 ? QY265PRESET_SLOW : (int)cfg->preset;
        e->fcfg.sub_satd = kPresetSubme[ps].satd; e->fcfg.sub_thr = kPresetSubme[ps].thr; e->fcfg.sub_flat = kPresetSubme[ps].flat;
        e->fcfg.sub_cap = kPresetSubme[ps].cap; e->fcfg.sub_cap_step = kPresetSubme[ps].cap_step; e->fcfg.sub_diag_fast = kPresetSubme[ps].diag_fast;
    }
    e->fcfg.bframes = e->gop_b; e->fcfg.refs = e->refs > e->refs_b ? e->refs : e->refs_b; if (e->refs0 > e->fcfg.refs) e->fcfg.refs = e->refs0; e->fcfg.me_hex_thr = e->hex_thr;
    e->fcfg.sdh = 1;                                                    /* the reference's streams have sign_data_hiding_enabled_flag = 1 at every preset */
    e->fcfg.pre_search = 1;                                             /* stage A0: pyramid pre-search vectors as start candidates of the integer search */
    e->fcfg.merge = 1;                                                  /* stage C2: merge pass on the motion field (pictures with one reference per list) */
    e->fcfg.propagate = getenv("KS265_PROPAGATE") ? atoi(getenv("KS265_PROPAGATE")) & 3 : 1;   /* stage A2: rounds of vector propagation between neighbouring PUs after every
                                                                         * integer search (measured with one round: - 21 .. - 23 % bytes of the P / B pictures; the variable is a measuring aid) */
    e->fcfg.intra_inter = 1;                                            /* P / B pictures may hold intra CUs (uncovered regions, occlusions); 2 = none of 8x8: measured + 1.6 % bits, no faster */
    e->fcfg.rdo = 4;                                                    /* coefficient-group pruning at lambda x 1 (ks265_frame_cfg.rdo): supersedes the coefficient decimation of round 2 */
    e->lean_b = getenv("KS265_LEAN_B") ? atoi(getenv("KS265_LEAN_B")) : 1;           /* non-reference B pictures without intra candidates / joint refinement / SAO (submit) */
    e->fcfg.skip_rd = getenv("KS265_SKIP_RD") ? atoi(getenv("KS265_SKIP_RD")) & 3 : 1;   /* stage D2 (round 6): after the reconstruction of a B picture, nodes whose merge candidate without
                                                                         * residual is the cheaper coding - on the coded distortion - become one CU (ks265_frame_cfg.skip_rd; 2 = P pictures
                                                                         * too, where it gains nothing measurable; the variable is a measuring aid) */
    e->fcfg.tu_inter = cfg->tuInter > 0 ? 1 : 0;                        /* -intertu N (tuInter; veryslow 1, placebo 2): the residual quadtree of inter CUs ONE level deep (ks265_frame_cfg.tu_inter); deeper values run as 1 */
    e->fcfg.part = cfg->part ? 1 : 0;                                   /* -part 1 (slower, veryslow, placebo): 2NxN / Nx2N prediction units in P and B pictures (ks265_frame_cfg.part) */
    e->fcfg.bi_refine = getenv("KS265_BI_REFINE") ? atoi(getenv("KS265_BI_REFINE")) : (cfg->preset >= QY265PRESET_SLOWER && cfg->preset <= QY265PRESET_PLACEBO ? 2 : 0);   /* 2 (round 5): after the CU decision, for the CUs it chose; 1 = for every PU of the quadtree.  End of round 6: up to -preset slow it is off - measured on the MI355X at 1080p and 2160p it COSTS 0.5 - 0.7 % bytes at equal PSNR-Y (it lowers the Hadamard cost of residuals the quantiser drops, and pays vector bits) and 134 us per B picture; the presets that trade speed for tools (slower .. placebo) keep it */                                              /* B pictures: joint refinement of the bi-predictive pair (motionSearchBI enc@0x484910) */
    r = ks265_frame_geometry(&e->fcfg, &e->geom);
    if (!r) r = ks265_frame_create(e->ctx, &e->fcfg, &e->frame);
    const size_t fsz = (size_t)e->W * e->H * 3 / 2, npx = (size_t)e->W * e->H;
    const int dev_id = device;
    if (!r) r = ks265_frame_compact_layout(e->frame, e->cmp_off);
    if (!r) r = ks265_create(&e->ctx_in, dev_id);
    if (!r) r = ks265_create(&e->ctx_out, dev_id);
    if (!r && e->direct_in && upl_order == 1) { if (ks265_create(&e->ctx_upl, dev_id)) { e->ctx_upl = NULL; e->direct_in = 0; } }
    if (!r && la_wanted && la_order == 1) r = ks265_create(&e->ctx_la, dev_id);
    e->split = getenv("KS265_NO_SPLIT") ? 0 : 1;
    e->copy_mb = getenv("KS265_COPYOUT_MB") ? atoi(getenv("KS265_COPYOUT_MB")) : -1;
    for (int k = 0; k < NPIPE && !r; ++k) {
        r = ks265_dev_malloc(e->ctx, (void **)&e->dev_in[k], fsz);
        if (!r) r = ks265_dev_malloc(e->ctx, (void **)&e->stg[k], e->cmp_off[7]);
        if (!r) r = ks265_memset_async(e->ctx, e->stg[k], 0, e->cmp_off[7]);   /* the header's running counters start at zero */
        if (!r) r = ks265_event_create(e->ctx, &e->ev_h2d[k]);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_loaded[k]);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_staged[k]);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_drained[k]);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_coded[k]);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_packed[k]);
        if (!r && e->split) r = pic_alloc(e, &e->srcq[k]);
    }
    if (!r) r = ks265_dev_malloc(e->ctx, (void **)&e->dev_sse, 64);
    e->aq_on = cfg->iAqMode != 0 && cfg->fAqStrength > 0;
    /* cuTree: -rc 3 (CRF) with the reference's -cutree 1 (its default), B pictures (with -bframes 0 the reference runs no tree: TEncParam+0x388 = 0, every offset it leaves is zero)
     * and a lookahead (not -lookahead 0, not zero latency): a QP per CTU from the lookahead window */
    e->ct_preset = (int)cfg->preset < 0 || (int)cfg->preset > 8 ? QY265PRESET_SLOW : (int)cfg->preset;
    e->ct_on = cfg->rc == 3 && g_cli.cutree && e->gop_b > 0 && cfg->lookahead != 0 && cfg->latency != QY265LATENCY_ZERO && e->W >= 64 && e->H >= 64 && !getenv("KS265_NO_CUTREE");
    {   /* how far behind a key picture / a mini-GOP's anchor the window reaches - measured inside the reference (the calcFrameCost calls of every batch, -bframes 3): veryfast 8,
         * fast / medium 12, slow / veryslow 60, -lookahead 20: 16 - i.e. a queue of L pictures (12 / 16 / 64 by preset, or -lookahead N) in whole mini-GOPs: ((L - 1) / span) span */
        const int L = cfg->lookahead > 0 ? cfg->lookahead : e->ct_preset <= 2 ? 12 : e->ct_preset <= 4 ? 16 : 64, span = e->gop_b + 1;
        e->ct_depth = ((L - 1) / span) * span;
    }
    if (e->ct_depth < e->gop_b + 1) e->ct_depth = e->gop_b + 1;
    if (e->ct_depth > CT_RING - 40) e->ct_depth = CT_RING - 40;
    e->qmap_on = e->aq_on || e->ct_on;
    if (e->qmap_on && getenv("KS265_DUMP_QPMAP")) e->qmap_fd = open(getenv("KS265_DUMP_QPMAP"), O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (!r && e->ct_on) {
        r = ct_open(e, dev_id);
        logf_(1, e->log_level, "ks265enc: -rc 3: cuTree over a lookahead of %d pictures (%d x %d blocks of %d half-size samples; -cutree 0 / -lookahead 0 switch it off): QP per CTU = picture QP + the tree's mean offset\n",
              e->ct_depth, e->ct_nx, e->ct_ny, 1 << e->ct_lg);
    }
    if (e->qmap_on) {
        for (int k = 0; k < NPIPE && !r; ++k) r = ks265_dev_malloc(e->ctx, (void **)&e->dev_qmap[k], (size_t)e->geom.ctu_cols * e->geom.ctu_rows);
        for (int q = 0; q < 2 && !r; ++q) r = ks265_dev_malloc(e->ctx, (void **)&e->dev_qmap_key[q], (size_t)e->geom.ctu_cols * e->geom.ctu_rows);
    }
    if (e->aq_on && !e->ct_on) {
        e->aq_nx = (e->W + 15) / 16; e->aq_ny = (e->H + 15) / 16;       /* blocks that hang over the picture read its padding (replicated edges) */
        const size_t nb = (size_t)e->aq_nx * e->aq_ny;
        for (int q = 0; q < 2 && !r; ++q) {
            r = ks265_dev_malloc(e->ctx, (void **)&e->aq_off[q], nb * 8);
            if (!r) r = ks265_dev_malloc(e->ctx, (void **)&e->aq_inv[q], nb * 2);
            if (!r) r = ks265_dev_malloc(e->ctx, (void **)&e->aq_scratch[q], 16);
        }
    }
    if (!r) r = pic_alloc(e, &e->src);
    e->ndpb = e->hier ? 10 : e->gop_b ? 4 : e->refs + 2;
    for (int i = 0; i < MAX_DPB; ++i) e->dpb_poc[i] = -1000000;
    for (int i = 0; i < e->ndpb && !r; ++i) r = pic_alloc(e, &e->dpb[i]);
    e->key_overlap = getenv("KS265_NO_KEY_OVERLAP") ? 0 : 1;
    e->use_graph = getenv("KS265_GRAPH") ? 1 : 0;                     /* opt-in since round 4: launch by launch is faster on this runtime (843 against 817 pictures/s, 2160p IPPP) and the split pipeline needs the launches apart */
    if (e->qmap_on) e->use_graph = 0;                                    /* (a captured picture would replay one map) */
    if (e->rdoq_on) {
        e->use_graph = 0;                                                /* (a captured picture would replay one set of tables) */
        e->rq_hist = (int32_t *)calloc((size_t)RQ_HIST * 1440, sizeof(int32_t));
        if (!e->rq_hist) r = KS265_OUTOFMEMORY;
        /* rdoQuant's two lambdas by QP (luma weights 256 / 256 = -rdoql / -rdoqls of the reference): (int64)(weight x 0.85 x 2^((qp - 12) / 3) + 0.5) */
        for (int q = 0; q < 52; ++q) { const double lam = 0.85 * pow(2.0, (q - 12) / 3.0); e->rq_lam[q] = e->rq_lam[52 + q] = (int64_t)(256 * lam + 0.5); }
    }
    if (e->use_graph) e->split = 0;
    if (e->key_overlap) {
        if (!r) r = ks265_create_prio(&e->ctx_key, dev_id, getenv("KS265_KEY_PRIO") ? atoi(getenv("KS265_KEY_PRIO")) : 1);   /* the key picture's wavefront must run underneath the P pictures, not behind them */
        if (!r) r = ks265_frame_create(e->ctx_key, &e->fcfg, &e->frame_key);
        if (!r) r = pic_alloc(e, &e->src_key);
        if (!r) r = ks265_dev_malloc(e->ctx, (void **)&e->dev_sse_key, 64);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_key);
        for (int i = 0; i < 2 && !r; ++i) { r = pic_alloc(e, &e->dpb[e->ndpb + i]); e->dpb_poc[e->ndpb + i] = -1000000; if (!r) r = ks265_event_create(e->ctx, &e->ev_firstp[i]); }
    }
    /* opt-in (KS265_ANCHOR_LANE=1): measured on the MI355X at 2160p, default GOP - 561 pictures/s without, 568 with the lane at normal stream priority, 440 at high priority
     * (and 382 before the anchor waited for its upload directly); with 8 hardware queues (GPU_MAX_HW_QUEUES) 449 without and 515 with.  The kernels of a picture fill the
     * device: an anchor running beside B pictures slows them by what it gains */
    e->anc_on = e->hier && e->refs0 == 1 && e->key_overlap && !e->qmap_on && !e->use_graph && getenv("KS265_ANCHOR_LANE") && atoi(getenv("KS265_ANCHOR_LANE")) > 0;
    if (e->anc_on) {
        if (!r) r = ks265_create_prio(&e->ctx_anc, dev_id, getenv("KS265_ANC_PRIO") ? atoi(getenv("KS265_ANC_PRIO")) : 0);
        if (!r) r = ks265_frame_create(e->ctx_anc, &e->fcfg, &e->frame_anc);
        if (!r) r = pic_alloc(e, &e->src_anc);
        if (!r) r = ks265_dev_malloc(e->ctx, (void **)&e->dev_sse_anc, 64);
        if (!r) r = ks265_event_create(e->ctx, &e->ev_anc);
        for (int i = 0; i < 8 && !r; ++i) r = ks265_event_create(e->ctx, &e->ev_mg[i]);
        for (int i = 0; i < 4 && !r; ++i) { r = pic_alloc(e, &e->dpb[e->ndpb + 2 + i]); e->dpb_poc[e->ndpb + 2 + i] = -1000000; }
    }
    /* ring of pictures in flight: a key picture's slice takes one writer thread many picture periods, and output is in coding order - the ring must
     * hold everything that is coded meanwhile, or the GPU idles behind it.  About 6 GB of records (pinned compact block + expanded level planes + pinned input), at least 24 and at most MAX_JOBS pictures. */
    /* (the lookahead's objects; its stream was created first - see the top of this function) */
    /* no -lookahead on the command line and the SDK's default GOP (hierarchical B, 8): the slice-type decision runs by itself (round 4: it costs a search of the half-size
     * picture every fourth picture, and the caller does not wait for it) - the reference's adaptive BiPredFrames is on by default as well.  -lookahead 0 switches it off. */
    const int la_auto = cfg->lookahead < 0 && e->hier && e->gop_b == 7 && !getenv("KS265_NO_AUTO_LOOKAHEAD");
    if (!r && (cfg->lookahead > 0 || la_auto)) {
        const int w = (e->W / 2) & ~7, h = (e->H / 2) & ~7;            /* the analysis sees the picture without its last columns / rows when half the size is no multiple of 8 */
        if (w < 16 || h < 16) { if (!la_auto) logf_(1, e->log_level, "ks265enc: -lookahead %d: the analysis needs a picture of at least 32 x 32: off\n", cfg->lookahead); }
        else {
            ks265_frame_cfg lc; memset(&lc, 0, sizeof lc);
            lc.width = w; lc.height = h; lc.qp = e->base_qp > 0 ? e->base_qp : 27; lc.lambda_q4 = kLambdaQ4[lc.qp < 52 ? lc.qp : 51]; lc.me_range = 32; lc.me_method = 1; lc.subme = 0;
            lc.bframes = 0; lc.refs = 1;
            r = e->ctx_la ? 0 : getenv("KS265_LA_PRIO") ? ks265_create_prio(&e->ctx_la, dev_id, atoi(getenv("KS265_LA_PRIO"))) : ks265_create(&e->ctx_la, dev_id);
            if (!r) r = ks265_frame_geometry(&lc, &e->geom_la);
            if (!r) r = ks265_frame_create(e->ctx_la, &lc, &e->frame_la);
            for (int i = 0; i < LA_RING && !r; ++i) {
                r = ks265_dev_malloc(e->ctx_la, (void **)&e->la_pic[i].y, (size_t)e->geom_la.bytes_y);
                if (!r) r = ks265_dev_malloc(e->ctx_la, (void **)&e->la_pic[i].u, (size_t)e->geom_la.bytes_c);
                if (!r) r = ks265_dev_malloc(e->ctx_la, (void **)&e->la_pic[i].v, (size_t)e->geom_la.bytes_c);
                if (!r) r = ks265_memset_async(e->ctx_la, e->la_pic[i].u, 128, (size_t)e->geom_la.bytes_c);     /* the analysis is luma only */
                if (!r) r = ks265_memset_async(e->ctx_la, e->la_pic[i].v, 128, (size_t)e->geom_la.bytes_c);
            }
            if (!r) r = ks265_dev_malloc(e->ctx_la, (void **)&e->la_cost_ws, (size_t)e->geom_la.ctu_cols * e->geom_la.ctu_rows * 85 * sizeof(uint32_t));
            if (!r) r = ks265_dev_malloc(e->ctx_la, (void **)&e->la_dev_out, LA_FLY * 128);
            if (!r) r = ks265_host_malloc(e->ctx_la, (void **)&e->la_host_out, LA_FLY * 128);
            for (int i = 0; i < LA_FLY && !r; ++i) r = ks265_event_create(e->ctx_la, &e->la_evs[i]);
            /* how many pictures may wait for their analysis before the caller waits for the oldest one: five with one lane (the caller is paced by the output anyway); a GOP lane
             * takes a whole GOP in ahead of its pixel path - its analyses queue up behind two lanes' kernels, and a caller that waited for them fed 770 pictures/s (round 5) */
            e->la_keep = getenv("KS265_LA_KEEP") && atoi(getenv("KS265_LA_KEEP")) > 0 && atoi(getenv("KS265_LA_KEEP")) < LA_QMAX - 8 ? atoi(getenv("KS265_LA_KEEP")) : multi ? 48 : 5;
            if (!r) { e->la_on = 1; e->la_auto = la_auto; e->la_last_key = -1000000; e->la_prev_icost = -1; e->la_w = w; e->la_h = h; e->mg_adapt = e->hier; e->mg4_until = -1; }
        }
    }
    e->ring = (int)(((size_t)6 << 30) / (e->cmp_off[7] + npx * 3 + fsz));
    if (e->ring > MAX_JOBS) e->ring = MAX_JOBS;
    if (e->ring < 24) e->ring = 24;
    if (e->nthreads > e->ring - 14) e->nthreads = e->ring - 14;               /* more writers than pictures that can be in flight would idle */
    const int njobs_alloc = e->ring;
    for (int i = 0; i < njobs_alloc && !r; ++i) {
        Job *j = &e->jobs[i];
        r = ks265_host_malloc(e->ctx, (void **)&j->cmp, e->cmp_off[7]);
        if (!r) {
            j->lvlbuf = (uint8_t *)calloc(npx * 3 + 64, 1);                 /* zero pages until a picture touches them */
            j->dirty = (uint64_t *)calloc((e->cmp_off[6] - e->cmp_off[5]) / 8 + 1, 8);
            if (!j->lvlbuf || !j->dirty) r = KS265_OUTOFMEMORY;
            j->cu8 = (ks265_cu8 *)(j->cmp + e->cmp_off[0]); j->sao = (ks265_sao_param *)(j->cmp + e->cmp_off[1]); j->sse = (uint64_t *)(j->cmp + e->cmp_off[2]);
            j->lvl[0] = (int16_t *)j->lvlbuf; j->lvl[1] = (int16_t *)(j->lvlbuf + npx * 2); j->lvl[2] = (int16_t *)(j->lvlbuf + npx * 2 + npx / 2);
        }
        if (!r && e->qmap_on) r = ks265_host_malloc(e->ctx, (void **)&j->qp_map, (size_t)e->geom.ctu_cols * e->geom.ctu_rows);
        if (!r && e->rdoq_on) r = ks265_host_malloc(e->ctx, (void **)&j->rq_host, 1440 * sizeof(int32_t) + 104 * sizeof(int64_t));
        if (!r) r = ks265_event_create(e->ctx, &j->ev);
        j->nal_cap = npx * 2 + 65536;
        j->nal = (uint8_t *)malloc(j->nal_cap);
        if (!j->nal) r = KS265_OUTOFMEMORY;
    }
    /* input slots: the ring + a mini-GOP.  A GOP lane takes a whole GOP more: the caller hands the GOPs out in stream order, so a lane that could not hold its next GOP
     * while it is still coding the current one would make the caller wait - and the OTHER lanes, whose next GOPs come after, run dry (measured: lanes idle a third of
     * the time with ring + 32 slots) */
    e->nin = e->ring + 32 + (e->la_on ? 8 : 0) + (e->ct_on ? e->ct_depth : 0) + (multi ? (cfg->iIntraPeriod < 256 ? cfg->iIntraPeriod : 256) : 0);
    if (getenv("KS265_INPUT_SLOTS")) e->nin = atoi(getenv("KS265_INPUT_SLOTS"));
    if (e->nin < e->ring + 32) e->nin = e->ring + 32;
    if (e->nin > MAX_INPUT) e->nin = MAX_INPUT;
    {   /* pinned host memory is a machine-wide resource: a lane's input slots stay under KS265_PINNED_MB (default 4096 MB; the ring + one mini-GOP is the floor).  At 2160p
         * a slot is 12.4 MB: ring + 32 = 132 slots = 1.6 GB; a GOP lane at -iper 128 asks for 260 = 3.2 GB; eight GPUs with two lanes each = 52 GB without the cap */
        const char *pm = getenv("KS265_PINNED_MB");
        const long long budget = (pm ? atoll(pm) : 4096) * 1048576LL;
        const int fit = (int)(budget / (long long)fsz);
        if (e->nin > fit && fit >= e->ring + 32) {
            logf_(1, e->log_level, "ks265enc: %d input slots of %.1f MB would pin %.1f GB per lane: %d slots (KS265_PINNED_MB=%lld)\n", e->nin, fsz / 1048576.0, e->nin * (double)fsz / 1073741824.0, fit, budget / 1048576LL);
            e->nin = fit;
        } else if (e->nin > fit) {
            logf_(1, e->log_level, "ks265enc: the ring + one mini-GOP (%d input slots, %.1f GB pinned) is above KS265_PINNED_MB=%lld and is the floor\n", e->ring + 32, (e->ring + 32) * (double)fsz / 1073741824.0, budget / 1048576LL);
            e->nin = e->ring + 32;
        }
    }
    for (int i = 0; i < e->nin && !r; ++i) r = ks265_host_malloc(e->ctx, (void **)&e->in[i].i420, fsz);
    /* with the lookahead every input slot has a twin on the device, uploaded on the lookahead's stream the moment the picture is handed in (nothing on that stream ever waits for
     * the pipeline: an upload enqueued on the copy-in stream, or anywhere on the copy engine behind that stream's uploads, sits behind the pipeline's back-pressure and the
     * analysis would see the picture 10 ms and more late); without it the picture is uploaded when it is scheduled, on the copy-in stream, as before.  One stream for both,
     * not a sixth: a stream more is a hardware queue shared with somebody (top of this function) */
    if (e->direct_in && upl_order == 2) { if (ks265_create(&e->ctx_upl, dev_id)) { e->ctx_upl = NULL; e->direct_in = 0; } }
    if (e->direct_in && upl_order == 4) e->direct_in = 2;              /* host-synchronous uploads: no stream of their own */
    if (e->direct_in == 1 && !e->ctx_upl) e->direct_in = 0;

    if ((e->la_on && e->ctx_la) || e->direct_in) {
        e->ctx_up = e->direct_in == 1 && e->ctx_upl ? e->ctx_upl : e->ctx_la ? e->ctx_la : e->ctx;
        reg_handle(e->ctx, 1);
        for (int i = 0; i < e->nin && !r; ++i) { r = ks265_dev_malloc(e->ctx, (void **)&e->in[i].dev, fsz); if (!r) r = ks265_event_create(e->ctx_up, &e->in[i].ev_up); }
        logf_(2, e->log_level, "ks265enc: %d input slots with a device twin each: %.2f GB of device memory per lane (KS265_INPUT_SLOTS / KS265_PINNED_MB bound the slot count)\n", e->nin, e->nin * (double)fsz / 1073741824.0);
    }
    if (r) { *err = hip_rc(r); lane_close(e, 0); return NULL; }
    memset(&e->scfg, 0, sizeof e->scfg);
    e->scfg.width = e->W; e->scfg.height = e->H; e->scfg.sao = e->use_sao; e->scfg.deblock = e->use_df;
    e->scfg.sdh = e->fcfg.sdh;
    e->zero_latency = cfg->latency == QY265LATENCY_ZERO && e->gop_b == 0 && !e->la_on && !multi;
    e->scfg.cu_qp_delta = e->qmap_on;
    e->scfg.tu_inter = e->fcfg.tu_inter;
    e->scfg.wpp = 1;                                                    /* CTU rows as substreams: what lets several writer threads share one picture */
    e->scfg.max_dec_pic_buffering = e->hier ? 10 : e->gop_b ? 4 : e->refs + 1; e->scfg.log2_max_poc_lsb = 16;
    /* pictures that precede a picture in decoding order and follow it in output order: the whole GOP for the hierarchy (7, as before), ONE (the anchor) for
     * P + n non-reference B whatever n is (-bframes 1..15) */
    e->scfg.max_num_reorder = e->hier ? e->gop_b : e->gop_b ? 1 : 0;
    e->hdr = (uint8_t *)malloc(512);
    if (e->hdr) {
        long a = ks265_write_vps(&e->scfg, e->hdr, 512), b = a > 0 ? ks265_write_sps(&e->scfg, e->hdr + a, 512 - (size_t)a) : -1, c = b > 0 ? ks265_write_pps(&e->scfg, e->hdr + a + b, 512 - (size_t)(a + b)) : -1;
        e->hdr_len = c > 0 ? a + b + c : -1;
        e->hdr_part[0] = a; e->hdr_part[1] = b; e->hdr_part[2] = c;
    }
    e->outcap = npx + 65536;
    e->outbuf = (uint8_t *)malloc(e->outcap);
    if (!e->hdr || e->hdr_len < 0 || !e->outbuf) { *err = QY_FAIL; lane_close(e, 0); return NULL; }
    for (int i = 0; i < e->ring; ++i) { e->jobs[i].wpp = malloc(ks265_wpp_bytes(&e->scfg)); if (!e->jobs[i].wpp) { *err = QY_OUTOFMEMORY; lane_close(e, 0); return NULL; } }   /* virtual until used */
    if (e->md5) {                                                       /* -md5 1: plane MD5s of every reconstructed picture */
        e->md5_ring = calloc(256, sizeof *e->md5_ring); e->md5_have = calloc(256, 1);
        if (!e->md5_ring || !e->md5_have || lane_recon_on(e)) { *err = QY_OUTOFMEMORY; lane_close(e, 0); return NULL; }
    }
    for (int i = 0; i < e->nthreads; ++i) { e->warg[i].e = e; e->warg[i].idx = i; if (pthread_create(&e->th[i], NULL, worker, &e->warg[i])) break; ++e->nth; }
    if (e->nth && !pthread_create(&e->disp, NULL, dispatcher, e)) e->disp_on = 1;
    if (e->disp_on && !pthread_create(&e->sched, NULL, scheduler, e)) e->sched_on = 1;
    if (!e->nth || !e->disp_on || !e->sched_on) { *err = QY_FAIL; lane_close(e, 0); return NULL; }
    logf_(0, e->log_level, "ks265enc: GPU %d: %dx%d %.2f fps, qp %d, -me %d (hex below %d), subme %d, refs %d, %s, sao %d, key period %d, %d slice writer threads, %s\n", device, e->W, e->H,
          cfg->frameRate, e->base_qp, e->me_method, e->hex_thr, e->subme, e->refs, e->hier ? (e->gop_b == 3 ? "hierarchical-B GOP 4" : "hierarchical-B GOP 8") : e->gop_b ? "P + non-reference B" : "IPPP", e->use_sao, e->iper,
          e->nth, ks265_version());
    return e;
}

static void lane_reconfig(Enc *e, QY265EncConfig *cfg)
{
    if (!e || !cfg) return;
    pthread_mutex_lock(&e->mu);                                        /* lane_put snapshots base_qp / iper into the input slot under this lock */
    e->cfg.qp = cfg->qp; e->cfg.crf = cfg->crf; e->cfg.bitrateInkbps = cfg->bitrateInkbps; e->cfg.iIntraPeriod = cfg->iIntraPeriod;
    e->base_qp = e->cfg.rc == 3 ? cfg->crf : cfg->qp; e->iper = cfg->iIntraPeriod;   /* the rate-control mode itself is not reconfigurable: the handle's own rc picks crf / qp */
    if (e->base_qp < 0) e->base_qp = 0;
    if (e->base_qp > 51) e->base_qp = 51;
    pthread_mutex_unlock(&e->mu);
}
static int lane_headers(Enc *e, QY265Nal **pNals, int *n)
{
    if (!e || !pNals || !n) return QY_POINTER;
    long off = 0;                                                       /* VPS, SPS, PPS as three entries (as in front of a key picture): callers build hvcC / extradata per entry */
    for (int k = 0; k < 3; ++k) {
        e->nals[k].naltype = KS265_NAL_VPS + k; e->nals[k].tid = 0; e->nals[k].iSize = (int)e->hdr_part[k]; e->nals[k].pts = 0; e->nals[k].pPayload = e->hdr + off;
        off += e->hdr_part[k];
    }
    *pNals = e->nals; *n = 3;
    return QY_OK;
}
static int lane_delayed(Enc *e)
{
    if (!e) return 0;
    int n = 0;
    pthread_mutex_lock(&e->mu);                                        /* one snapshot: a picture moves from "waiting" to "in flight" under this lock */
    for (int i = 0; i < MAX_INPUT; ++i) if (e->in[i].used == 1) ++n;
    n += e->njobs + e->la_qn;                                          /* (la_q: at the input, their analysis still running) */
    pthread_mutex_unlock(&e->mu);
    return n;
}

/* zero-copy input: hand the caller one of the lane's pinned input slots to write the next picture into; QY265EncoderEncodeFrame recognises the
 * pointer and copies nothing.  Never blocks: QY_FAIL when no slot is free right now (the caller then passes its own buffer, which is copied as usual). */
static int lane_acquire(Enc *e, QY265YUV *yuv)
{
    Input *slot = NULL;
    pthread_mutex_lock(&e->mu);
    for (int i = 0; i < e->nin && !slot; ++i) if (e->in[i].used == 4) slot = &e->in[i];          /* acquired before and not handed in yet: the same one again */
    for (int i = 0; i < e->nin && !slot; ++i) if (!e->in[i].used) slot = &e->in[i];
    if (slot) slot->used = 4;
    pthread_mutex_unlock(&e->mu);
    if (!slot) return QY_FAIL;
    yuv->iWidth = e->W; yuv->iHeight = e->H;
    yuv->pData[0] = slot->i420; yuv->pData[1] = slot->i420 + (size_t)e->W * e->H; yuv->pData[2] = yuv->pData[1] + (size_t)e->W * e->H / 4;
    yuv->iStride[0] = e->W; yuv->iStride[1] = e->W / 2; yuv->iStride[2] = e->W / 2;
    return QY_OK;
}

/* ---- the lookahead at the input (-lookahead N, or by itself with the SDK's default GOP).  Round 4: nobody waits for it.  A picture's analysis is launched on a stream of its own
 *      when the picture is handed in; the picture then sits in a short queue (la_q, display order) until its results have arrived - looked at, not waited for, whenever the caller
 *      hands in another picture - and only then goes to the scheduler.  The queue holds at most LA_KEEP pictures: a result that is still missing then is waited for (the analysis
 *      has had several picture times by then).  Same decisions as when the caller waited for every picture (round 3; the CPU tests did not change), a few pictures of delay at
 *      the input.  The caller's thread launches; the caller's thread and - when it has nothing to schedule - the scheduler thread look (la_mu). */
#define LA_KEEP (e->la_keep)

/* a picture whose analysis is through (or which needs none) goes to the scheduler */
static void la_publish(Enc *e, Input *slot, int cut, int mini4)
{
    const int nd = slot->disp;
    pthread_mutex_lock(&e->mu);
    if (!e->la_auto) {                                                 /* (auto: no scene cuts, the key positions are known when the picture is handed in - la_take keeps them) */
        const int periodic = slot->iper > 0 && e->la_last_key > -1000000 && nd - e->la_last_key >= slot->iper;    /* (the scheduler's own rule: key positions are the same there) */
        if (cut) ++e->la_cuts;
        if (cut || slot->key || nd == 0 || periodic) e->la_last_key = nd;
    }
    slot->mini4 = mini4; slot->key = slot->key || cut; slot->used = 1; e->next_disp = nd + 1;
    pthread_cond_signal(&e->cv_sched);
    pthread_mutex_unlock(&e->mu);
}

/* the results of one analysed picture: the scene-cut verdict and, with the hierarchical GOP, the slice types of its block of 8 */
static void la_decide(Enc *e, const Input *slot, int *cut_out, int *mini4_out)
{
    const int nd = slot->disp, w = e->la_w, h = e->la_h;
    const uint64_t *out = e->la_host_out + 16 * slot->la_buf;
    int cut = 0, mini4 = 0;
    if (slot->la_what & 1) {                                           /* scene cut: the picture against its predecessor */
        /* a cut: predicting the picture from its predecessor costs at least 0.7 of coding it intra (both sums over the 8x8 blocks of the half-size picture),
         * and the last key picture is at least eight pictures back */
        if (g_cli.scenecut > 0) {
            /* -scenecut N: the reference's verdict (scenecut enc@0x47e9d0, restated in oracle/ks265_lookahead_ref.c and pinned on recorded calls) on this lookahead's
             * frame costs: a change of flatness (intra cost below 4 per 8x8 block of the half-size picture) decides at once; else a cut is where predicting the
             * picture costs at least (1 - N / 100 x pictures since the key picture / min(key period, 320)) of coding it intra */
            const long long icost = (long long)out[0], pcost = (long long)out[1], prev = e->la_prev_icost;
            const long long T = (long long)(w / 8) * (h / 8) * 4;
            int verdict = -1;
            if (prev >= 0) {
                if (prev < T) { if (icost > T) verdict = 1; else if (icost < T) verdict = 0; }
                else if (prev > T && icost < T) verdict = 1;
            }
            if (verdict < 0) {
                const int keyint = slot->iper > 0 ? (slot->iper < 320 ? slot->iper : 320) : 256;
                const double bias = (double)(nd - (e->la_last_key > -1000000 ? e->la_last_key : 0)) * ((double)g_cli.scenecut / 100.0) / (double)keyint;
                verdict = (double)pcost >= (1.0 - bias) * (double)icost;
            }
            cut = verdict;
            e->la_prev_icost = icost;
        } else if (out[1] * 10 >= out[0] * 7 && nd - e->la_last_key >= 8) cut = 1;
    }
    /* slice types of the hierarchical GOP (the reference's adaptive BiPredFrames): the GOP is laid out in blocks of 8 pictures from its key picture; a block is
     * coded with its anchor 8 pictures after the previous one, or - when predicting that anchor from 8 pictures back costs more than the two anchors 4 apart cost
     * together (+ 1/12: the shorter structure pays more B-picture overhead) - as two mini-GOPs of 4.  Costs = the inter sums of the frame-cost kernels, this
     * picture against the pictures 4 and 8 back; decided at the block's last picture, carried to the scheduler in its input slot. */
    if ((slot->la_what & 2) && !cut) {                                 /* (launched before the verdict was known: a cut starts a GOP here, the grid's sums are not used) */
        const int p = slot->la_p;
        const unsigned long long c4 = out[5];
        if ((p & 7) == 0) {
            const unsigned long long c8 = out[9], two = c4 + e->la_c4_prev;
            mini4 = c8 * 12 > two * 13;
            if (mini4) ++e->la_mini4;
        }
        e->la_c4_prev = c4;
    }
    *cut_out = cut; *mini4_out = mini4;
}

/* hand the queue's pictures to the scheduler, oldest first, as far as their results are there; while more than `keep` pictures are queued the oldest one's results are waited for */
static int la_drain_locked(Enc *e, int keep)
{
    while (e->la_qn > 0) {
        Input *h = e->la_q[0];
        int cut = 0, mini4 = 0;
        if (h->la_what >= 0) {
            int r, done = 1;
            if (e->la_qn > keep) { const double t0 = now_ms(); r = ks265_event_wait(e->ctx_la, e->la_evs[h->la_buf]); e->la_t_wait += now_ms() - t0; ++e->la_n_wait; }
            else r = ks265_event_query(e->ctx_la, e->la_evs[h->la_buf], &done);
            if (r) { pthread_mutex_lock(&e->mu); for (int i = 0; i < e->la_qn; ++i) e->la_q[i]->used = 0; e->la_qn = 0; e->la_flying = 0; e->sched_err = hip_rc(r); pthread_mutex_unlock(&e->mu); return hip_rc(r); }
            if (!done) break;
            la_decide(e, h, &cut, &mini4);
            --e->la_flying;
        }
        la_publish(e, h, cut, mini4);
        memmove(e->la_q, e->la_q + 1, sizeof e->la_q[0] * (size_t)(e->la_qn - 1));
        __atomic_store_n(&e->la_qn, e->la_qn - 1, __ATOMIC_RELEASE);
    }
    return QY_OK;
}
static int la_drain(Enc *e, int keep)
{
    pthread_mutex_lock(&e->la_mu);
    const int r = la_drain_locked(e, keep);
    pthread_mutex_unlock(&e->la_mu);
    return r;
}

/* a picture handed in: its half-size picture and the frame-cost kernels it needs (against its predecessor: scene cut; on the GOP's grid of 4 against the pictures 4 and 8
 * back: slice types), results on their way to the host; into the queue */
static int la_take_locked(Enc *e, Input *slot)
{
    int r;
    /* -lookahead N: a scene-cut verdict moves the GOP's grid, so the next picture's launch needs every verdict before it; auto: nothing of the launch depends on results.
     * At most four analyses in flight (their result areas) */
    if ((r = la_drain_locked(e, e->la_auto ? LA_KEEP : 0))) return r;
    while (e->la_flying >= LA_FLY || e->la_qn >= LA_QMAX - 1) if ((r = la_drain_locked(e, e->la_qn - 1))) return r;
    const int nd = slot->disp, c = nd % LA_RING, w = e->la_w, h = e->la_h;
    const size_t org = (size_t)e->geom_la.pad_y * e->geom_la.stride_y + e->geom_la.pad_y;
    const int keynow = slot->key || nd == 0 || (slot->iper > 0 && e->la_last_key > -1000000 && nd - e->la_last_key >= slot->iper);   /* (but for a scene cut, not known yet) */
    const int p = keynow ? 0 : nd - e->la_last_key;                    /* position inside the GOP */
    if (e->la_auto && keynow) e->la_last_key = nd;
    int what = -1;
    r = 0;
    if (!e->la_auto || (p & 3) == 0) {                                 /* auto: pictures off the grid are neither analysed nor kept */
        const int buf = e->la_seq++ % LA_FLY;
        uint64_t *dout = e->la_dev_out + 16 * buf;
        what = 0;
        /* the half-size picture from the slot's twin on the device (uploaded a moment ago on this stream).  Measured in round 4 before that existed: an H2D copy of the
         * luma plane on this stream (round 3) queues on the copy engine BEHIND the uploads the scheduler had enqueued, which waited for the pipeline's buffers - the results
         * arrived 10 ms and more late; kernels reading the pinned picture over PCIe instead (ks265_downsample_from_host, 64 work-groups) cost the encoder 28 % */
        r = e->ctx_up != e->ctx_la && !slot->up_sync ? ks265_stream_wait_event(e->ctx_la, slot->ev_up) : 0;      /* (the upload runs on a stream of its own) */
        if (!r) r = ks265_downsample_rect(e->ctx_la, slot->dev, e->W, e->la_pic[c].y + org, e->geom_la.stride_y, w, h);
        if (!r) r = ks265_pad_picture(e->frame_la, e->la_pic[c]);
        if (!r && e->la_have_prev && !e->la_auto) {
            r = ks265_lookahead_picture(e->frame_la, e->la_pic[c], e->la_pic[(nd + LA_RING - 1) % LA_RING], e->la_cost_ws, dout);
            what |= 1;
        }
        if (!r && e->mg_adapt && p >= 4 && (p & 3) == 0) {
            /* (-lookahead N: the picture's intra costs are in la_cost_ws from the call above - search + sums only; auto: NO intra pass since round 5 - the slice-type decision
             *  reads the searches' sums alone (la_decide: out[5], out[9]), and the pass was 0.3 ms per grid picture = 6 % of the device's time with the default GOP) */
            const uint32_t *ws = e->la_auto ? NULL : e->la_cost_ws;
            r = ks265_lookahead_inter(e->frame_la, e->la_pic[c], e->la_pic[(nd - 4) % LA_RING], ws, dout + 4);
            if (!r && (p & 7) == 0) r = ks265_lookahead_inter(e->frame_la, e->la_pic[c], e->la_pic[(nd - 8) % LA_RING], ws, dout + 8);
            what |= 2;
        }
        if (!r && what) r = ks265_memcpy_d2h_async(e->ctx_la, e->la_host_out + 16 * buf, dout, 96);
        if (!r) r = ks265_event_record(e->ctx_la, e->la_evs[buf]);     /* (the slot's pinned picture is being read by the upload: the slot stays here until the event has passed) */
        if (r) return hip_rc(r);
        e->la_have_prev = 1; slot->la_buf = buf; ++e->la_flying;
    }
    slot->la_what = what; slot->la_p = p;
    e->la_q[e->la_qn] = slot;
    __atomic_store_n(&e->la_qn, e->la_qn + 1, __ATOMIC_RELEASE);
    return la_drain_locked(e, LA_KEEP);
}
static int la_take(Enc *e, Input *slot)
{
    pthread_mutex_lock(&e->la_mu);
    const int r = la_take_locked(e, slot);
    pthread_mutex_unlock(&e->la_mu);
    return r;
}

/* one picture into the lane: copy to a pinned slot, hand it to the scheduler thread.  key: it starts a closed GOP regardless of the period */
/* ---- the application's picture buffers, pinned in place once and remembered (process-wide: hipHostRegister refuses memory that is registered already) --------------------------- */
#define REG_MAX 160
static struct { pthread_mutex_t mu; struct { uint8_t *base; size_t n; unsigned long stamp; } ent[REG_MAX]; int n; unsigned long clock; uint8_t *bad[32]; int nbad; int handles; } g_reg = {.mu = PTHREAD_MUTEX_INITIALIZER};
static int reg_get(ks265_ctx *c, uint8_t *p, size_t n, int may_evict)
{
    int ok = 0;
    pthread_mutex_lock(&g_reg.mu);
    ++g_reg.clock;
    for (int i = 0; i < g_reg.n && !ok; ++i)
        if (p >= g_reg.ent[i].base && p + n <= g_reg.ent[i].base + g_reg.ent[i].n) { g_reg.ent[i].stamp = g_reg.clock; ok = 1; }
    for (int i = 0; i < g_reg.nbad && !ok; ++i) if (g_reg.bad[i] == p) { pthread_mutex_unlock(&g_reg.mu); return 0; }
    if (!ok) {
        for (int i = 0; i < g_reg.n; )                                 /* a buffer that overlaps an older registration: that memory has been given back and handed out again */
            if (p < g_reg.ent[i].base + g_reg.ent[i].n && g_reg.ent[i].base < p + n) { (void)ks265_host_unregister(c, g_reg.ent[i].base); g_reg.ent[i] = g_reg.ent[--g_reg.n]; } else ++i;
        if (g_reg.n == REG_MAX && may_evict) {                         /* least recently used */
            int lru = 0;
            for (int i = 1; i < g_reg.n; ++i) if (g_reg.ent[i].stamp < g_reg.ent[lru].stamp) lru = i;
            (void)ks265_host_unregister(c, g_reg.ent[lru].base); g_reg.ent[lru] = g_reg.ent[--g_reg.n];
        }
        if (g_reg.n < REG_MAX && ks265_host_register(c, p, n) == KS265_OK) { g_reg.ent[g_reg.n].base = p; g_reg.ent[g_reg.n].n = n; g_reg.ent[g_reg.n++].stamp = g_reg.clock; ok = 1; }
        else if (g_reg.nbad < 32) g_reg.bad[g_reg.nbad++] = p;         /* memory the runtime cannot pin: that buffer is copied from now on */
    }
    pthread_mutex_unlock(&g_reg.mu);
    return ok;
}
static void reg_handle(ks265_ctx *c, int open)
{
    pthread_mutex_lock(&g_reg.mu);
    g_reg.handles += open ? 1 : -1;
    if (!open && g_reg.handles == 0) { for (int i = 0; i < g_reg.n; ++i) (void)ks265_host_unregister(c, g_reg.ent[i].base); g_reg.n = 0; g_reg.nbad = 0; }   /* the last lane of the process lets go of the application's memory */
    pthread_mutex_unlock(&g_reg.mu);
}

static int lane_put(Enc *e, QY265Picture *in, int key)
{
    if (!in->yuv || !in->yuv->pData[0] || !in->yuv->pData[1] || !in->yuv->pData[2]) return QY_POINTER;
    if (in->yuv->iWidth != e->W || in->yuv->iHeight != e->H) return QY_NOTSUPPORTED;
    Input *slot = NULL;
    const double tc0 = now_ms();
    pthread_mutex_lock(&e->mu);
    /* back-pressure on the input side: at most 16 pictures wait for the scheduler thread (it may itself be waiting for ring space, which only the
     * caller's take_output frees - then go on and collect).  GOP lanes (multi): the input slots are the limit (lane_has_slot) - a lane takes a whole GOP in
     * while it is still coding the previous one, or the caller would wait here while the other lanes run dry */
    /* (cuTree: the scheduler itself holds a mini-GOP back until its window has arrived - that many more may wait) */
    while (!e->multi && !e->quit && !e->sched_err && e->next_disp - (e->coded_upto + 1) > 16 + (e->ct_on ? e->ct_depth + e->gop_b + 1 : 0) && e->njobs <= e->ring - 12) pthread_cond_wait(&e->cv_sched_done, &e->mu);
    e->la_t_bp += now_ms() - tc0;   /* a scheduler that failed makes no more progress */
    int own = 0;                                                       /* the caller wrote the picture into a slot it had acquired (ks265_enc_acquire_input): nothing to copy */
    for (int i = 0; i < e->nin && !slot; ++i) if (e->in[i].used == 4 && e->in[i].i420 == in->yuv->pData[0]) { slot = &e->in[i]; own = 1; }
    for (int i = 0; i < e->nin && !slot; ++i) if (!e->in[i].used) slot = &e->in[i];
    for (int i = 0; i < e->nin && !slot; ++i) if (e->in[i].used == 4) slot = &e->in[i];   /* last resort: a buffer the caller acquired and did not use for this picture - copy into it
                                                                                            * (the caller's pointer to it is dead from here on, as after any EncodeFrame call) */
    if (slot) { slot->used = 3; slot->up_sync = 0; }                   /* being filled */
    pthread_mutex_unlock(&e->mu);
    if (!slot) return QY_FAIL;                                         /* one lane: cannot happen (more input slots than pictures in flight + one mini-GOP); lanes: the caller checked lane_has_slot */
    uint8_t *u = slot->i420 + (size_t)e->W * e->H, *v = u + (size_t)e->W * e->H / 4;
    int direct = 0;
    if (e->direct_in && !own && slot->dev && in->yuv->iStride[0] == e->W && in->yuv->iStride[1] == e->W / 2 && in->yuv->iStride[2] == e->W / 2) {
        const size_t ny = (size_t)e->W * e->H, nc = ny / 4;
        uint8_t *p0 = in->yuv->pData[0], *p1 = in->yuv->pData[1], *p2 = in->yuv->pData[2];
        const int whole = p1 == p0 + ny && p2 == p1 + nc, evict = !e->hold_in;      /* (hold: an old registration may still be read by a DMA in flight - then the table only grows) */
        int ru = KS265_OK;
        if (whole ? reg_get(e->ctx_up, p0, ny + 2 * nc, evict) : (reg_get(e->ctx_up, p0, ny, evict) && reg_get(e->ctx_up, p1, nc, evict) && reg_get(e->ctx_up, p2, nc, evict))) {
            if (e->direct_in == 2) {                                    /* no stream: the call returns when the picture is on the device */
                if (whole) ru = ks265_memcpy_h2d_sync(e->ctx, slot->dev, p0, ny + 2 * nc);
                else { ru = ks265_memcpy_h2d_sync(e->ctx, slot->dev, p0, ny); if (!ru) ru = ks265_memcpy_h2d_sync(e->ctx, slot->dev + ny, p1, nc); if (!ru) ru = ks265_memcpy_h2d_sync(e->ctx, slot->dev + ny + nc, p2, nc); }
                slot->up_sync = 1;
            } else {
            if (whole) ru = ks265_memcpy_h2d_async(e->ctx_up, slot->dev, p0, ny + 2 * nc);
            else { ru = ks265_memcpy_h2d_async(e->ctx_up, slot->dev, p0, ny); if (!ru) ru = ks265_memcpy_h2d_async(e->ctx_up, slot->dev + ny, p1, nc); if (!ru) ru = ks265_memcpy_h2d_async(e->ctx_up, slot->dev + ny + nc, p2, nc); }
            if (!ru) ru = ks265_event_record(e->ctx_up, slot->ev_up);
            if (!e->hold_in) e->pending_up = slot;                      /* waited for at the end of the call, behind the output's hand-over */
            }
            if (ru) { e->pending_up = NULL; pthread_mutex_lock(&e->mu); slot->used = 0; e->sched_err = hip_rc(ru); pthread_mutex_unlock(&e->mu); return hip_rc(ru); }
            direct = 1;
        }
    }
    if (direct) { /* on its way from the caller's own memory */ }
    else if (own) { /* in place */ }
    else if (in->yuv->iStride[0] == e->W && in->yuv->iStride[1] == e->W / 2 && in->yuv->iStride[2] == e->W / 2) {    /* packed planes: three block copies */
        copy_shared(e->pool, slot->i420, in->yuv->pData[0], (size_t)e->W * e->H);
        copy_shared(e->pool, u, in->yuv->pData[1], (size_t)e->W * e->H / 4);
        copy_shared(e->pool, v, in->yuv->pData[2], (size_t)e->W * e->H / 4);
    } else {
        for (int y = 0; y < e->H; ++y) memcpy(slot->i420 + (size_t)y * e->W, in->yuv->pData[0] + (size_t)y * in->yuv->iStride[0], (size_t)e->W);
        for (int y = 0; y < e->H / 2; ++y) {
            memcpy(u + (size_t)y * (e->W / 2), in->yuv->pData[1] + (size_t)y * in->yuv->iStride[1], (size_t)e->W / 2);
            memcpy(v + (size_t)y * (e->W / 2), in->yuv->pData[2] + (size_t)y * in->yuv->iStride[2], (size_t)e->W / 2);
        }
    }
    if (slot->dev && !direct) {   /* on its way to the device at once, on the lookahead's stream (nothing there waits for the pipeline) */
        int ru = ks265_memcpy_h2d_async(e->ctx_up, slot->dev, slot->i420, (size_t)e->W * e->H * 3 / 2);
        if (!ru) ru = ks265_event_record(e->ctx_up, slot->ev_up);
        if (ru) { pthread_mutex_lock(&e->mu); slot->used = 0; e->sched_err = hip_rc(ru); pthread_mutex_unlock(&e->mu); return hip_rc(ru); }
    }
    /* the picture's own fields travel with it from now; with the lookahead it becomes visible to the scheduler when its results are there (la_drain) */
    pthread_mutex_lock(&e->mu);
    slot->mini4 = 0;
    slot->disp = e->in_disp++; slot->pts = in->pts; slot->key = key || e->force_key; slot->base_qp = e->base_qp; slot->iper = e->iper; slot->kbps = e->cfg.bitrateInkbps;
    e->force_key = 0;
    if (!e->la_on) { slot->used = 1; e->next_disp = e->in_disp; pthread_cond_signal(&e->cv_sched); }   /* the scheduler thread takes it from here */
    pthread_mutex_unlock(&e->mu);
    if (e->la_on) {
        const double t0 = now_ms();
        const int r = la_take(e, slot);
        e->la_t_take += now_ms() - t0;
        if (r) { pthread_mutex_lock(&e->mu); slot->used = 0; e->sched_err = r; pthread_mutex_unlock(&e->mu); return r; }
    }
    e->st.in_copy_ms += now_ms() - tc0;
    return QY_OK;
}

/* flush, first half: everything that has arrived gets scheduled (short mini-GOPs at the end).  Returns 1 when the scheduler is through, 0 when the ring
 * filled up first (the scheduler then waits for the caller to collect output: hand out what is ready and come back, as the SDK's flush loop does anyway -
 * EncodeFrame(NULL) while DelayedFrames() > 0) */
static int lane_flush_begin(Enc *e, int wait)
{
    if (e->la_on) (void)la_drain(e, 0);                                /* the pictures whose analysis is still running: waited for (an error is in sched_err) */
    pthread_mutex_lock(&e->mu);
    e->sched_flush = 1;
    pthread_cond_signal(&e->cv_sched);
    int all = 0;
    for (;;) {
        all = !e->sched_flush && e->sched_idle && e->sched_seen == e->next_disp;
        if (all || !wait || e->quit || e->njobs > e->ring - 12) break;   /* GOP lanes do not wait here: while the caller sat in one lane's flush nobody would collect from the others */
        pthread_cond_wait(&e->cv_sched_done, &e->mu);
    }
    pthread_mutex_unlock(&e->mu);
    return all;
}

static int lane_encode_frame(Enc *e, QY265Nal **pNals, int *iNalCount, QY265Picture *in, QY265Picture *out)
{
    *pNals = e->nals; *iNalCount = 0;
    int r = QY_OK;
    if (in) {
        r = lane_put(e, in, 0);
        if (r) return r;
        /* finished pictures (copied to the output buffer); when the ring of in-flight pictures is nearly full, wait for the oldest ones -
         * only as many as needed, the writers keep running */
        const double t0 = now_ms();
        int in_flight = e->ring - 12;
        if (e->zero_latency) {
            /* -latency zerolatency (QY265LATENCY_ZERO; the SDK's live-streaming mode): the call returns with THIS picture's NAL units - nothing stays behind, QY265EncoderDelayedFrames
             * is 0 between calls.  The picture goes through the same pipeline (scheduler thread, streams, writer threads sharing its rows); the caller waits for it */
            pthread_mutex_lock(&e->mu);
            while (!e->quit && !e->sched_err && (e->sched_seen != e->next_disp || !e->sched_idle)) pthread_cond_wait(&e->cv_sched_done, &e->mu);
            pthread_mutex_unlock(&e->mu);
            in_flight = 0;
        }
        r = take_output(e, in_flight, 1 << 30, pNals, iNalCount, out, NULL);
        e->st.output_ms += now_ms() - t0;
        if (e->pending_up) {                                           /* the caller's buffer is its own again when the call returns: the upload out of it has finished */
            const double tw = now_ms();
            const int ru = ks265_event_wait(e->ctx_up, e->pending_up->ev_up);
            e->pending_up = NULL;
            e->st.in_copy_ms += now_ms() - tw;
            if (ru && !r) r = hip_rc(ru);
        }
        return r ? r : e->sched_err;
    }
    /* flush: then every picture in flight is collected */
    const double t0 = now_ms();
    const int all = lane_flush_begin(e, 1);
    if (e->sched_err) return e->sched_err;
    r = take_output(e, all ? 0 : e->ring - 12, 1 << 30, pNals, iNalCount, out, NULL);
    e->st.output_ms += now_ms() - t0;
    return r;
}


/* extension: dump the encoder's reconstruction as I420, every picture at its display position (the reference CLI's -o).  Call right after
 * QY265EncoderOpen, before the first picture.  Costs one more D2H of W*H*3/2 bytes per picture: a checking aid, not part of the normal path. */
static int lane_recon_on(Enc *e)                                       /* the reconstruction of every picture comes back to the host (pinned, per job) */
{
    if (e->recon_on) return QY_OK;
    if (e->in_disp != 0) return QY_NOTSUPPORTED;
    e->key_overlap = 0;                                                /* the dump shares one device buffer: key pictures stay on the main stream */
    const size_t fsz = (size_t)e->W * e->H * 3 / 2;
    int r = ks265_dev_malloc(e->ctx, (void **)&e->dev_recon, fsz);
    for (int i = 0; i < e->ring && !r; ++i) r = ks265_host_malloc(e->ctx, (void **)&e->jobs[i].recon, fsz);
    if (r) return hip_rc(r);
    e->recon_on = 1;
    return QY_OK;
}
static int lane_set_recon_file(Enc *e, const char *path)
{
    if (!e || !path) return QY_POINTER;
    if (e->in_disp != 0 || e->recon_fd >= 0) return QY_NOTSUPPORTED;
    const int r = lane_recon_on(e);
    if (r) return r;
    e->recon_fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    return e->recon_fd >= 0 ? QY_OK : QY_FAIL;
}

/* ------------------------------------------------------------------ GOP lanes
 * One lane (Enc) keeps one picture pipeline busy: a P picture needs its predecessor, so a single closed GOP fills about half of the MI355X (one stream
 * of the hot path runs at half the rate of three, bench.py).  Closed GOPs are independent - the shards of SURVEY.md 8(e) - so the encoder codes several of
 * them AT ONCE on one GPU: the handle owns L lanes, every lane a complete pipeline (streams, workspace, DPB, scheduler / dispatcher / writer threads);
 * GOP k of the input goes to lane k mod L, each GOP's first picture marked as a key picture in band; output is handed out in GOP order (all of GOP k,
 * then GOP k + 1 from the next lane), so the stream is byte for byte the one a single lane writes (tests/test_gpu_enc_api.py).  The reference's
 * enFrameParallel (frames of one stream coded concurrently on CPU threads) is the switch: lanes run with enFrameParallel != 0, fixed QP (-rc 0) and a key
 * period of at least 32 pictures (the rate controllers carry state across GOPs).  Every GOP structure works: the scheduler closes a GOP in front of a key picture
 * (the mini-GOP there is shortened to end in a P picture), so nothing references across lanes.  Cost: output lags the input by up to L GOPs, and L pipelines' worth of buffers.
 * Round 5: TWO lanes by default for the pyramid GOPs on one GPU (top_lanes_wanted: 631 -> 700 pictures/s at 2160p, GPU-bound at both), one otherwise.  IPPP, measured at 2160p
 * on the round-2 box: two lanes reach 1.05x of one (1123 vs 1068 frames/s; round 5: 890 against 970) -
 * with twice the pictures in flight the slice writers, not the GPU, set the pace (their time per picture grows from 19 to 59 ms of thread time as the
 * threads spread over the host), DESIGN.md section 6. */
#define MAX_LANES 16
#define MAX_CHUNKS 64
/* stash: pictures of a GOP that is not yet at the head of the output order are taken out of their lane as they finish (their NAL units copied here), so that the
 * lane's ring of pictures in flight never fills up with finished work waiting for an earlier GOP of another lane */
typedef struct Chunk {
    int lane, closed; long count, delivered, base; int disp0; /* the lane's own display index of the GOP's first picture */
    long stashed; QY265Nal *snal; size_t *soff; int sn, sn_cap; uint8_t *sbuf; size_t scap, spos; QY265Picture sout;
} Chunk;
typedef struct Top {
    int nlanes; Enc *lane[MAX_LANES];
    int iper, key_request, cur_lane;
    long n_in, chunk_left;
    Chunk ch[MAX_CHUNKS]; int ch_head, ch_n;
    CopyPool *pool;
    QY265Nal *onals; size_t *ooff; int on, on_cap;                   /* output of the current call: NAL payloads copied out of the lanes */
    uint8_t *obuf; size_t ocap, opos;
    double output_ms;
    TopWake wake;
} Top;

/* is there room for the picture whose first plane is `data`?  A slot in the caller's hands (4: ks265_enc_acquire_input) counts only if THIS picture is the one that was
 * produced into it - a caller may acquire a buffer and then hand in its own, or the picture may be routed to another lane */
static int lane_has_slot(Enc *e, const uint8_t *data)
{
    int ok = 0;
    pthread_mutex_lock(&e->mu);
    for (int i = 0; i < e->nin && !ok; ++i) ok = !e->in[i].used || (e->in[i].used == 4 && e->in[i].i420 == data);
    pthread_mutex_unlock(&e->mu);
    return ok;
}

static int nal_append(QY265Nal **dn, size_t **doff, int *dcnt, int *dcap, uint8_t **dbuf, size_t *bcap, size_t *bpos, const QY265Nal *nals, int n)
{
    for (int i = 0; i < n; ++i) {
        if (*dcnt == *dcap) {
            const int nc = *dcap ? 2 * *dcap : 1024;
            QY265Nal *a = (QY265Nal *)realloc(*dn, (size_t)nc * sizeof *a);
            if (a) *dn = a;
            size_t *b = a ? (size_t *)realloc(*doff, (size_t)nc * sizeof *b) : NULL;
            if (!a || !b) return QY_OUTOFMEMORY;
            *doff = b; *dcap = nc;
        }
        const size_t need = (size_t)nals[i].iSize;
        if (*bpos + need > *bcap) {
            const size_t nc = (*bpos + need) * 2 + 65536;
            uint8_t *nb = (uint8_t *)realloc(*dbuf, nc);
            if (!nb) return QY_OUTOFMEMORY;
            *dbuf = nb; *bcap = nc;
        }
        memcpy(*dbuf + *bpos, nals[i].pPayload, need);
        (*dn)[*dcnt] = nals[i]; (*doff)[*dcnt] = *bpos;
        ++*dcnt; *bpos += need;
    }
    return QY_OK;
}
static int top_append(Top *t, const QY265Nal *nals, int n) { return nal_append(&t->onals, &t->ooff, &t->on, &t->on_cap, &t->obuf, &t->ocap, &t->opos, nals, n); }

/* move finished pictures to the call's output, GOP after GOP; block = wait until at least one picture has come */
static int top_collect_pass(Top *t, QY265Picture *out, long *progress, long *pending)
{
    while (t->ch_n) {
        Chunk *c = &t->ch[t->ch_head];
        if (c->sn) {                                                    /* what was taken out of the lane while earlier GOPs were still going out */
            for (int i = 0; i < c->sn; ++i) c->snal[i].pPayload = c->sbuf + c->soff[i];
            const int ra = top_append(t, c->snal, c->sn);
            if (ra) return ra;
            c->delivered += c->stashed; *progress += c->stashed; c->stashed = 0; c->sn = 0; c->spos = 0;
            if (out) { out->iSliceType = c->sout.iSliceType; out->pts = c->sout.pts; out->dts = c->sout.dts; out->poc = (int)(c->base + (c->sout.poc - c->disp0)); }
        }
        if (c->delivered == c->count) {
            if (!c->closed) break;                                      /* the GOP still receives input */
            free(c->snal); free(c->soff); free(c->sbuf); c->snal = NULL; c->soff = NULL; c->sbuf = NULL; c->sn_cap = 0; c->scap = 0;
            t->ch_head = (t->ch_head + 1) % MAX_CHUNKS; --t->ch_n;
            continue;
        }
        QY265Nal *nals; int n = 0, pics = 0;
        const int r = take_output(t->lane[c->lane], MAX_JOBS + 1, (int)(c->count - c->delivered), &nals, &n, out, &pics);
        /* (a picture that failed on the device has left the lane's ring like any other - without a payload: it is on the books before the error goes up, or this GOP
         *  would wait for it for ever) */
        const int ra = pics ? top_append(t, nals, n) : QY_OK;
        c->delivered += pics; *progress += pics;
        if (r) return r;
        if (ra) return ra;
        if (!pics) break;
        if (out) out->poc = (int)(c->base + (out->poc - c->disp0));    /* the lane reports its own display index */
    }
    /* the GOPs behind the head: whatever their lanes have finished goes into the stash (a lane hands its pictures out in its own order: only its oldest unfinished
     * GOP can be taken from) */
    unsigned busy = 0;                                                  /* lanes whose oldest unfinished GOP has been seen */
    for (int k = 0; k < t->ch_n; ++k) {
        Chunk *c = &t->ch[(t->ch_head + k) % MAX_CHUNKS];
        *pending += c->count - c->delivered - c->stashed;
        if (busy >> c->lane & 1u) continue;
        if (c->delivered + c->stashed == c->count && c->closed) continue;   /* all taken: the lane's next GOP is its oldest unfinished one */
        busy |= 1u << c->lane;
        if (k == 0) continue;                                           /* the head goes out directly (above) */
        for (;;) {
            const long left = c->count - c->delivered - c->stashed;
            if (left <= 0) break;
            QY265Nal *nals; int n = 0, pics = 0;
            const int r = take_output(t->lane[c->lane], MAX_JOBS + 1, (int)left, &nals, &n, &c->sout, &pics);
            const int ra = pics ? nal_append(&c->snal, &c->soff, &c->sn, &c->sn_cap, &c->sbuf, &c->scap, &c->spos, nals, n) : QY_OK;
            c->stashed += pics; *progress += pics; *pending -= pics;
            if (r) return r;
            if (ra) return ra;
            if (!pics) break;
        }
    }
    return QY_OK;
}

/* move finished pictures to the call's output, GOP after GOP (and, from the lanes behind, to their stash); block = do not come back before SOMETHING has moved
 * (a picture out, or into a stash: either frees a lane's buffers), unless nothing is on its way at all */
static int top_collect(Top *t, int block, QY265Picture *out)
{
    for (;;) {
        pthread_mutex_lock(&t->wake.mu);
        const unsigned long seen = t->wake.seq;
        pthread_mutex_unlock(&t->wake.mu);
        long progress = 0, pending = 0;
        const int r = top_collect_pass(t, out, &progress, &pending);
        if (r || !block || progress || !pending) return r;
        for (int i = 0; i < t->nlanes; ++i) {
            Enc *e = t->lane[i];
            pthread_mutex_lock(&e->mu);
            const int bad = e->quit ? QY_FAIL : e->sched_err;
            pthread_mutex_unlock(&e->mu);
            if (bad) return bad;
        }
        pthread_mutex_lock(&t->wake.mu);
        if (t->wake.seq == seen) {                                      /* a writer thread of any lane finishing a picture wakes this up; the time limit covers a failing lane */
            struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_nsec += 20 * 1000000L; if (ts.tv_nsec >= 1000000000L) { ts.tv_nsec -= 1000000000L; ++ts.tv_sec; }
            pthread_cond_timedwait(&t->wake.cv, &t->wake.mu, &ts);
        }
        pthread_mutex_unlock(&t->wake.mu);
    }
}

/* no more input for the newest GOP.  early: it ends before its period is over (a key-picture request) - with B pictures the lane's scheduler is still waiting
 * for the rest of a mini-GOP, so it is told where the GOP ends (short mini-GOP there, as in front of a key picture); a GOP of full length ends by itself */
static void top_close_chunk(Top *t, int early)
{
    Chunk *c = &t->ch[(t->ch_head + t->ch_n - 1) % MAX_CHUNKS];
    c->closed = 1;
    if (t->lane[c->lane]->la_on) (void)la_drain(t->lane[c->lane], 0);  /* the GOP's last pictures do not wait for the lane's next ones (an error is in sched_err) */
    if (early) {
        Enc *e = t->lane[c->lane];
        pthread_mutex_lock(&e->mu);
        e->gop_end = e->in_disp - 1;                                  /* the lane's own index of the GOP's last picture */
        pthread_cond_signal(&e->cv_sched);
        pthread_mutex_unlock(&e->mu);
    }
}

/* The GPUs of this handle: KS265_DEVICES = "0,2,5" (an explicit list) or KS265_GPUS = N (devices 0 .. N - 1; the CLI's -gpus N), else KS265_DEVICE / 0.
 * Closed GOPs are independent ("frames / GOPs shard naturally across the GPUs of one node", no data-path collective), so a GPU is simply another
 * GOP lane: lane i runs on device dev[i mod ndev], KS265_GOP_LANES lanes per GPU.  The stream is byte for byte the one-GPU stream. */
static int top_devices(int dev[MAX_LANES])
{
    int n = 0;
    const char *list = getenv("KS265_DEVICES"), *cnt = getenv("KS265_GPUS"), *one = getenv("KS265_DEVICE");
    if (list && *list) {
        const char *p = list;
        while (*p && n < MAX_LANES) {
            char *end; const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v >= 0 && v < 1024) dev[n++] = (int)v;
            p = *end == ',' ? end + 1 : end;
        }
    } else if (cnt && atoi(cnt) > 0) { for (int i = 0; i < atoi(cnt) && i < MAX_LANES; ++i) dev[n++] = i; }
    if (!n) dev[n++] = one ? atoi(one) : 0;
    return n;
}
static int top_lanes_wanted(const QY265EncConfig *cfg, int ndev)
{
    /* one lane per GPU unless asked (round 6): two closed GOPs side by side fill what the B pictures of a pyramid GOP leave of the device (2160p: 700 pictures/s
     * against 631), but a lane runs a whole GOP behind its input - twice the lag and twice the pinned input of one - and needs more than the runtime's four hardware queues
     * (GPU_MAX_HW_QUEUES=8, exported BEFORE the process first touches the GPU; with four: 561).  Both are the application's decisions: the CLI and bench.py take them
     * (KS265_GOP_LANES=2 + GPU_MAX_HW_QUEUES=8 for the pyramid GOPs under -rc 0 / 3), the library never changes the environment. */
    const char *env = getenv("KS265_GOP_LANES");
    int per = env ? atoi(env) : 1;
    if (per > 1) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (!q || atoi(q) < 8) logf_(1, cfg->logLevel, "ks265enc: %d GOP lanes with %s hardware queues: export GPU_MAX_HW_QUEUES=8 before the process first uses the GPU (measured: two lanes on four queues are slower than one)\n", per, q ? q : "the runtime's default four");
    }
    if (per < 1) per = 1;
    int n = per * ndev;
    if (n > MAX_LANES) n = MAX_LANES;
    /* closed GOPs share no pixel; what a rate-control mode shares across GOPs decides whether they can be dealt to lanes:
     *   rc 0 (constant QP) and rc 3 (CRF = a constant ladder on crf): nothing - the lanes' stream is byte for byte the one-lane stream;
     *   rc 1 / 2 / 4 (bitrate targets): the controller's state.  With lanes every lane runs its OWN controller on the GOPs dealt to it, each picture carrying the same
     *   budget share (bitrate / frame rate, rc_account): the host-side bit-budget split of SURVEY.md 8(e) - a scalar per picture, no exchange between GPUs.  The
     *   stream is deterministic for a given lane count but not the one-lane stream (a lane's offset follows the GOPs it has seen), which the open logs;
     *   rc 5 / VBV are not implemented at all.  -md5 lines are in one display order: one lane. */
    if (!cfg->enFrameParallel || cfg->rc < 0 || cfg->rc > 4 || cfg->iIntraPeriod < 32 || g_cli.md5) n = 1;
    if (cfg->lookahead > 0) n = 1;                                   /* a scene cut restarts the key period: GOP boundaries are not known when the pictures are dealt to lanes */
    return n;
}

void *QY265EncoderOpen(QY265EncConfig *cfg, int *err)
{
    int dummy; if (!err) err = &dummy;
    *err = QY_OK;
    if (!cfg) { *err = QY_POINTER; return NULL; }
    Top *t = (Top *)calloc(1, sizeof *t);
    if (!t) { *err = QY_OUTOFMEMORY; return NULL; }
    int dev[MAX_LANES];
    const int ndev = top_devices(dev);
    t->nlanes = top_lanes_wanted(cfg, ndev);
    if (ndev > 1 && t->nlanes == 1) logf_(2, cfg->logLevel, "ks265enc: %d GPUs asked for, but GOP sharding needs enFrameParallel, -rc 0..4, no -lookahead / -md5 and a key period >= 32: one GPU\n", ndev);
    if (t->nlanes > 1 && (cfg->rc == 1 || cfg->rc == 2 || cfg->rc == 4))
        logf_(1, cfg->logLevel, "ks265enc: -rc %d over %d GOP lanes: every lane runs its own controller with the same per-picture bit budget (deterministic, not the one-lane stream)\n", cfg->rc, t->nlanes);
    t->iper = cfg->iIntraPeriod; t->cur_lane = -1;
    QY265EncConfig lc = *cfg;
    if (t->nlanes > 1) {                                                /* the writer threads are shared out: every lane sees 1 / L of the pictures */
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        /* the host's threads are the budget: every lane also runs a scheduler, a dispatcher (polling) and shares the copy helpers - three service threads apiece;
         * what is left is dealt to the lanes as slice writers, at most 32 each (one GPU's pictures keep about that many busy) */
        const int th = cfg->threads > 0 ? cfg->threads : (int)(ncpu > 0 ? ncpu : 4) - 3 * t->nlanes;    /* an explicit -threads is the total */
        lc.threads = th / t->nlanes;
        if (lc.threads > 32) lc.threads = 32;
        if (lc.threads < 2) lc.threads = 2;
    }
    for (int i = 0; i < t->nlanes; ++i) {
        if (i) lc.logLevel = cfg->logLevel > 2 ? cfg->logLevel : 3;     /* one start-up line */
        t->lane[i] = lane_open(&lc, dev[i % ndev], t->nlanes > 1, err);
        if (!t->lane[i]) {
            if (i == 0) { free(t); return NULL; }
            t->nlanes = i;                                              /* e.g. out of memory for another pipeline: go on with the lanes there are */
            *err = QY_OK;
            break;
        }
    }
    if ((size_t)cfg->picWidth * cfg->picHeight >= ((size_t)1 << 20)) t->pool = copy_pool_create();
    pthread_mutex_init(&t->wake.mu, NULL); pthread_cond_init(&t->wake.cv, NULL);
    for (int i = 0; i < t->nlanes; ++i) { t->lane[i]->pool = t->pool; t->lane[i]->multi = t->nlanes > 1; t->lane[i]->wake = t->nlanes > 1 ? &t->wake : NULL; }
    if (t->nlanes > 1) logf_(0, cfg->logLevel, "ks265enc: %d GOP lanes on %d GPU(s) (closed GOPs of %d pictures coded concurrently, output in GOP order)\n", t->nlanes, ndev < t->nlanes ? ndev : t->nlanes, t->iper);
    return t;
}

void QY265EncoderClose(void *h)
{
    Top *t = (Top *)h;
    if (!t) return;
    if (t->nlanes > 1) {                                                /* one summary line over all lanes */
        Enc *e0 = t->lane[0];
        ks265_enc_stats st; memset(&st, 0, sizeof st);
        for (int i = 0; i < t->nlanes; ++i) { const ks265_enc_stats *s = &t->lane[i]->st; st.frames += s->frames; st.bytes += s->bytes; for (int k = 0; k < 3; ++k) st.sse[k] += s->sse[k]; }
        if (e0->cfg.calcPsnr && st.frames) {
            const double np[3] = {(double)e0->W * e0->H, (double)e0->W * e0->H / 4, (double)e0->W * e0->H / 4};
            double ps[3];
            for (int k = 0; k < 3; ++k) ps[k] = st.sse[k] > 0 ? 10.0 * log10(255.0 * 255.0 * np[k] * st.frames / st.sse[k]) : 99.0;
            logf_(2, e0->log_level, "bitrate, psnr: %.4f\t%.4f\t%.4f\t%.4f\n", st.bytes * 8.0 * e0->cfg.frameRate / st.frames / 1000.0, ps[0], ps[1], ps[2]);
        }
    }
    for (int i = 0; i < t->nlanes; ++i) lane_close(t->lane[i], t->nlanes == 1);
    pthread_mutex_destroy(&t->wake.mu); pthread_cond_destroy(&t->wake.cv);
    copy_pool_destroy(t->pool);
    for (int i = 0; i < MAX_CHUNKS; ++i) { free(t->ch[i].snal); free(t->ch[i].soff); free(t->ch[i].sbuf); }
    free(t->onals); free(t->ooff); free(t->obuf);
    free(t);
}

void QY265EncoderReconfig(void *h, QY265EncConfig *cfg)
{
    Top *t = (Top *)h;
    if (!t || !cfg) return;
    for (int i = 0; i < t->nlanes; ++i) lane_reconfig(t->lane[i], cfg);
    if (t->nlanes > 1 && cfg->iIntraPeriod >= 1) t->iper = cfg->iIntraPeriod;
}
int QY265EncoderEncodeHeaders(void *h, QY265Nal **pNals, int *n)
{
    Top *t = (Top *)h;
    if (!t || !pNals || !n) return QY_POINTER;
    return lane_headers(t->lane[0], pNals, n);
}
void QY265EncoderKeyFrameRequest(void *h)
{
    Top *t = (Top *)h;
    if (!t) return;
    if (t->nlanes == 1) t->lane[0]->force_key = 1; else t->key_request = 1;   /* lanes: the next picture opens a new GOP on the next lane */
}
int QY265EncoderDelayedFrames(void *h)
{
    Top *t = (Top *)h;
    if (!t) return 0;
    int n = 0;
    for (int i = 0; i < t->nlanes; ++i) n += lane_delayed(t->lane[i]);
    for (int k = 0; k < t->ch_n; ++k) n += (int)t->ch[(t->ch_head + k) % MAX_CHUNKS].stashed;   /* taken out of their lane, waiting for an earlier GOP to go out */
    return n;
}

int QY265EncoderEncodeFrame(void *h, QY265Nal **pNals, int *iNalCount, QY265Picture *in, QY265Picture *out, int bForceLogo)
{
    (void)bForceLogo;
    Top *t = (Top *)h;
    if (!t || !pNals || !iNalCount) return QY_POINTER;
    if (t->nlanes == 1) return lane_encode_frame(t->lane[0], pNals, iNalCount, in, out);
    *pNals = NULL; *iNalCount = 0;
    t->on = 0; t->opos = 0;
    int r = QY_OK;
    if (in) {
        int first = 0;
        if (t->chunk_left <= 0 || t->key_request) {                     /* a new GOP: the next lane */
            if (t->ch_n) top_close_chunk(t, t->chunk_left > 0);
            while (t->ch_n == MAX_CHUNKS && !r) r = top_collect(t, 1, out);
            if (r) return r;
            t->cur_lane = (t->cur_lane + 1) % t->nlanes;
            Chunk *c = &t->ch[(t->ch_head + t->ch_n) % MAX_CHUNKS];
            memset(c, 0, sizeof *c);
            c->lane = t->cur_lane; c->base = t->n_in; c->disp0 = t->lane[t->cur_lane]->in_disp;
            ++t->ch_n;
            t->chunk_left = t->iper > 0 ? t->iper : (1L << 40);
            t->key_request = 0; first = 1;
        }
        Enc *e = t->lane[t->cur_lane];
        const double t0 = now_ms();
        while (!lane_has_slot(e, in->yuv ? in->yuv->pData[0] : NULL)) {                                     /* this lane is as far ahead as its buffers allow: finish older GOPs first */
            r = top_collect(t, 1, out);
            if (r) return r;
        }
        t->output_ms += now_ms() - t0;
        r = lane_put(e, in, first);
        if (r) return r;
        ++t->ch[(t->ch_head + t->ch_n - 1) % MAX_CHUNKS].count; ++t->n_in; --t->chunk_left;
        const double t1 = now_ms();
        r = top_collect(t, 0, out);
        t->output_ms += now_ms() - t1;
        if (e->pending_up) {                                            /* the caller's buffer is its own again when the call returns (lane_encode_frame) */
            const double tw = now_ms();
            const int ru = ks265_event_wait(e->ctx_up, e->pending_up->ev_up);
            e->pending_up = NULL;
            e->st.in_copy_ms += now_ms() - tw;
            if (ru && !r) r = hip_rc(ru);
        }
        if (!r) for (int i = 0; i < t->nlanes && !r; ++i) r = t->lane[i]->sched_err;
    } else {
        const double t0 = now_ms();
        if (t->ch_n) top_close_chunk(t, 0);                              /* the lanes are flushed below */
        t->chunk_left = 0;
        for (int i = 0; i < t->nlanes; ++i) {
            if (lane_delayed(t->lane[i])) lane_flush_begin(t->lane[i], 0);
            if (t->lane[i]->sched_err) return t->lane[i]->sched_err;
        }
        r = top_collect(t, 1, out);
        t->output_ms += now_ms() - t0;
    }
    for (int i = 0; i < t->on; ++i) t->onals[i].pPayload = t->obuf + t->ooff[i];
    *pNals = t->onals; *iNalCount = t->on;
    return r;
}

int ks265_enc_get_stats(void *h, ks265_enc_stats *out)
{
    Top *t = (Top *)h;
    if (!t || !out) return QY_POINTER;
    *out = t->lane[0]->st;
    for (int i = 1; i < t->nlanes; ++i) {                               /* sums over the lanes (per-picture averages divide by frames as before) */
        const ks265_enc_stats *s = &t->lane[i]->st;
        out->frames += s->frames; out->bytes += s->bytes;
        for (int k = 0; k < 3; ++k) out->sse[k] += s->sse[k];
        out->gpu_ms += s->gpu_ms; out->host_write_ms += s->host_write_ms; out->in_copy_ms += s->in_copy_ms; out->submit_ms += s->submit_ms;
        out->lat_gpu_ms += s->lat_gpu_ms; out->lat_queue_ms += s->lat_queue_ms; out->key_wall_ms += s->key_wall_ms; out->key_cpu_ms += s->key_cpu_ms; out->keys += s->keys;
        out->submit_wait_ms += s->submit_wait_ms; out->occ_samples += s->occ_samples; out->occ_ring += s->occ_ring; out->occ_gpu += s->occ_gpu; out->occ_ready += s->occ_ready;
    }
    if (t->nlanes > 1) out->output_ms = t->output_ms;
    return QY_OK;
}

int ks265_enc_lanes(void *h) { const Top *t = (const Top *)h; return t ? t->nlanes : 0; }

int ks265_enc_acquire_input(void *h, QY265YUV *yuv)
{
    Top *t = (Top *)h;
    if (!t || !yuv) return QY_POINTER;
    /* the lane the NEXT picture goes to (QY265EncoderEncodeFrame: a new GOP starts on the next lane) */
    const int lane = t->nlanes == 1 ? 0 : (t->chunk_left <= 0 || t->key_request) ? (t->cur_lane + 1) % t->nlanes : t->cur_lane;
    return lane_acquire(t->lane[lane], yuv);
}

int ks265_enc_set_recon_file(void *h, const char *path)
{
    Top *t = (Top *)h;
    if (!t || !path) return QY_POINTER;
    if (t->n_in) return QY_NOTSUPPORTED;
    for (int i = 1; i < t->nlanes; ++i) lane_close(t->lane[i], 0);     /* the dump is one file in display order: one lane */
    t->nlanes = 1;
    return lane_set_recon_file(t->lane[0], path);
}
