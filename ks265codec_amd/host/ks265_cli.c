/* ks265_cli.c — command-line front end with the flag surface of the SDK's `appencoder` (boundary "B1" of SURVEY.md §8(b): /root/reference/README.md:8-82
 * plus the tool flags the v2.6.1.3 binary accepts), driving the library API exactly like the SDK's own callers do
 * (encoderwrapper.c:327-427: ConfigDefaultPreset -> set fields -> Open -> EncodeFrame per picture -> flush while DelayedFrames -> Close).
 * Output conventions kept: Annex-B stream to -b, stdout lines `Total Frames: N, test time: X ms, FPS: F` and `bitrate, psnr: kbps Y U V`
 * (the Android demo parses the latter), `H265 encoder passed!!!` at the end. */
#define _GNU_SOURCE
#include "ks265_enc.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

/* the input file is read one picture ahead on a thread of its own: at 2160p a picture is 12 MB, ~0.6 ms from the page cache - as long as the encoder's share
 * of the calling thread (input copy, output) */
typedef struct {
    FILE *f; size_t fsz; unsigned char *buf[2];
    pthread_mutex_t mu; pthread_cond_t cv;
    int state[2];                                   /* 0 = free, 1 = filled, 2 = end of file */
    long limit, nread; int quit;
} Reader;
static void *reader_main(void *arg)
{
    Reader *r = (Reader *)arg;
    for (int k = 0;; k ^= 1) {
        pthread_mutex_lock(&r->mu);
        while (r->state[k] != 0 && !r->quit) pthread_cond_wait(&r->cv, &r->mu);
        const int stop = r->quit;
        pthread_mutex_unlock(&r->mu);
        if (stop) return NULL;
        const int ok = (r->limit < 0 || r->nread < r->limit) && fread(r->buf[k], 1, r->fsz, r->f) == r->fsz;
        pthread_mutex_lock(&r->mu);
        r->state[k] = ok ? 1 : 2;
        if (ok) ++r->nread;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
        if (!ok) return NULL;
    }
}
/* the next picture (NULL at the end of the input); the buffer stays valid until the following call */
static unsigned char *reader_next(Reader *r, int *cur)
{
    pthread_mutex_lock(&r->mu);
    if (*cur >= 0) { r->state[*cur] = 0; pthread_cond_broadcast(&r->cv); }     /* the previous picture's buffer may be refilled */
    const int k = *cur < 0 ? 0 : *cur ^ 1;
    while (r->state[k] == 0) pthread_cond_wait(&r->cv, &r->mu);
    const int st = r->state[k];
    pthread_mutex_unlock(&r->mu);
    *cur = k;
    return st == 1 ? r->buf[k] : NULL;
}

static void usage(void)
{
    puts("usage: ks265enc -i in.yuv -wdt W -hgt H [-fr FPS] [-preset ultrafast..placebo] [-latency zerolatency|lowdelay|livestreaming|default] [-tune T]\n"
         "                [-rc 0..5] [-qp Q] [-crf C] [-br KBPS] [-iper N] [-bframes N] [-frms N] [-threads N] [-psnr 0|1|2] [-b out.265] [-o recon.yuv]\n"
         "                [-me 0|1|2] [-subme 0|1|2] [-merange R] [-ref N] [-sao 0..4] [-df 0|1] [-fixqp 0|1] [-md5 0|1] [-scenecut N (with -lookahead: the reference's scene-cut rule at threshold N)] [-cutree 0|1 (-rc 3: the reference's macroblock tree over the lookahead, a QP per CTU; default 1)] [-aq 0|1 -aqs S (adaptive quantisation: a QP per CTU from the reference's block-variance rule)] [-c config_file] [-gpus N] [-v]\n"
         "  I420 8-bit input; width and height multiples of 8.  Needs one MI355X (gfx950): there is no CPU fallback.");
}

/* -c file: a text file of further options, `-name value` or `name value` / `name = value` / `name : value` per line, `#` starts a comment; its
 * options are spliced into the command line where -c stood (later options override earlier ones, as on the command line) */
static int splice_config(int *pargc, char ***pargv, int depth)
{
    int argc = *pargc; char **argv = *pargv;
    if (depth > 8) { fprintf(stderr, "-c nests deeper than 8 config files (a file that names itself?)\n"); return -1; }
    for (int i = 1; i + 1 < argc; ++i) {
        if (strcmp(argv[i], "-c")) continue;
        FILE *f = fopen(argv[i + 1], "r");
        if (!f) { fprintf(stderr, "cannot read the config file %s\n", argv[i + 1]); return -1; }
        const int cap = argc + 2048;                                     /* room for every argument of the command line + 1024 options from the file */
        char **nv = (char **)malloc(sizeof(char *) * (size_t)cap);
        int n = 0;
        if (!nv) { fclose(f); return -1; }
        for (int k = 0; k < i; ++k) nv[n++] = argv[k];
        char line[1024];
        while (fgets(line, sizeof line, f) && n + 2 + (argc - i - 2) <= cap) {    /* the arguments behind -c file still have to fit */
            char *h = strchr(line, '#'); if (h) *h = 0;
            char *name = strtok(line, " \t\r\n=:"), *val = name ? strtok(NULL, " \t\r\n=:") : NULL;
            if (!name || !val) continue;
            char *a = (char *)malloc(strlen(name) + 2);
            if (!a) break;
            sprintf(a, "%s%s", name[0] == '-' ? "" : "-", name);
            nv[n++] = a; nv[n++] = strdup(val);
        }
        fclose(f);
        for (int k = i + 2; k < argc; ++k) nv[n++] = argv[k];
        *pargc = n; *pargv = nv;
        return splice_config(pargc, pargv, depth + 1);                   /* a second -c (or one inside the file) */
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (splice_config(&argc, &argv, 0)) return 2;
    const char *in_path = NULL, *out_path = NULL, *preset = "medium", *latency = "default", *tune = "default";
    const char *rec_path = NULL;
    int frames = -1;
    /* two passes over the arguments: preset / latency / tune first (they reset every field), then the explicit settings */
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-v")) { printf("%s\n", strLibQy265Version); return 0; }
        if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) { usage(); return 0; }
        if (i + 1 < argc) {
            if (!strcmp(argv[i], "-preset")) preset = argv[++i];
            else if (!strcmp(argv[i], "-latency")) latency = argv[++i];
            else if (!strcmp(argv[i], "-tune")) tune = argv[++i];
        }
    }
    QY265EncConfig cfg;
    if (QY265ConfigDefaultPreset(&cfg, (char *)preset, (char *)tune, (char *)latency) != QY_OK) { fprintf(stderr, "bad -preset / -tune / -latency\n"); return 2; }
    cfg.rc = 0; cfg.calcPsnr = 1;
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        if (a[0] != '-') { fprintf(stderr, "unexpected argument %s\n", a); return 2; }
        if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a); return 2; }
        const char *v = argv[++i];
        if (!strcmp(a, "-i")) in_path = v;
        else if (!strcmp(a, "-b")) out_path = v;
        else if (!strcmp(a, "-o")) rec_path = v;
        else if (!strcmp(a, "-frms")) frames = atoi(v);
        else if (!strcmp(a, "-preset") || !strcmp(a, "-latency") || !strcmp(a, "-tune")) continue;
        else if (!strcmp(a, "-df") || !strcmp(a, "-fixqp") || !strcmp(a, "-md5") || !strcmp(a, "-scenecut") || !strcmp(a, "-cutree")) {       /* CLI switches without a QY265EncConfig field */
            if (ks265_enc_set_default(a + 1, atoi(v)) != QY_OK) { fprintf(stderr, "bad value for %s: %s\n", a, v); return 2; }
        }
        else if (!strcmp(a, "-gpus")) setenv("KS265_GPUS", v, 1);                            /* closed GOPs dealt to N GPUs behind this one handle (ks265_enc.h) */
        else {
            const int r = QY265ConfigParse(&cfg, a + 1, v);
            if (r == QY265_PARAM_BAD_NAME) { fprintf(stderr, "unknown option %s\n", a); return 2; }
            if (r == QY265_PARAM_BAD_VALUE) { fprintf(stderr, "bad value for %s: %s\n", a, v); return 2; }
        }
    }
    if (!in_path || cfg.picWidth <= 0 || cfg.picHeight <= 0) { usage(); return 2; }
    FILE *fi = fopen(in_path, "rb"), *fo = out_path ? fopen(out_path, "wb") : NULL;
    if (!fi || (out_path && !fo)) { perror("open"); return 1; }
    {   /* this process is the encoder's alone and has not touched the GPU yet: for the pyramid GOPs under -rc 0 / 3 it opts into two GOP lanes on eight hardware queues (DESIGN.md 6e;
         * the library itself never changes the environment and defaults to one lane).  KS265_GOP_LANES / GPU_MAX_HW_QUEUES set by the user stay. */
        const int gop_b = cfg.bframes < 0 ? (cfg.latency == QY265LATENCY_DEFAULT ? 7 : 0) : cfg.bframes;
        if ((gop_b == 7 || gop_b == 3) && (cfg.rc == 0 || cfg.rc == 3) && cfg.enFrameParallel && cfg.iIntraPeriod >= 32) { setenv("KS265_GOP_LANES", "2", 0); setenv("GPU_MAX_HW_QUEUES", "8", 0); }
    }
    int err = 0;
    void *h = QY265EncoderOpen(&cfg, &err);
    if (!h) { fprintf(stderr, "QY265EncoderOpen failed: 0x%08x\n", (unsigned)err); return 1; }
    if (rec_path && ks265_enc_set_recon_file(h, rec_path) != QY_OK) { fprintf(stderr, "cannot write the reconstruction to %s\n", rec_path); return 1; }
    const size_t luma = (size_t)cfg.picWidth * cfg.picHeight, fsz = luma * 3 / 2;
    Reader rd;
    memset(&rd, 0, sizeof rd);
    rd.f = fi; rd.fsz = fsz; rd.limit = frames;
    rd.buf[0] = (unsigned char *)malloc(fsz); rd.buf[1] = (unsigned char *)malloc(fsz);
    if (!rd.buf[0] || !rd.buf[1]) { fprintf(stderr, "out of memory\n"); return 1; }
    pthread_mutex_init(&rd.mu, NULL); pthread_cond_init(&rd.cv, NULL);
    pthread_t rth;
    if (pthread_create(&rth, NULL, reader_main, &rd)) { fprintf(stderr, "cannot start the reader thread\n"); return 1; }
    QY265YUV yuv = {cfg.picWidth, cfg.picHeight, {NULL, NULL, NULL}, {cfg.picWidth, cfg.picWidth / 2, cfg.picWidth / 2}};
    QY265Picture pic, outp;
    memset(&pic, 0, sizeof pic); memset(&outp, 0, sizeof outp);
    pic.yuv = &yuv;
    QY265Nal *nal; int nnal;
    long n = 0;
    const double t0 = now_ms();
    double t_io = 0;
    int cur = -1;
    for (;; ++n) {
        const double ta = now_ms();
        unsigned char *buf = reader_next(&rd, &cur);                    /* waits only if the reader is behind */
        t_io += now_ms() - ta;
        if (!buf) break;
        yuv.pData[0] = buf; yuv.pData[1] = buf + luma; yuv.pData[2] = buf + luma + luma / 4;
        pic.pts = n;
        err = QY265EncoderEncodeFrame(h, &nal, &nnal, &pic, &outp, 0);
        if (err) { fprintf(stderr, "EncodeFrame failed: 0x%08x\n", (unsigned)err); return 1; }
        for (int k = 0; k < nnal && fo; ++k) fwrite(nal[k].pPayload, 1, (size_t)nal[k].iSize, fo);
    }
    pthread_mutex_lock(&rd.mu); rd.quit = 1; pthread_cond_broadcast(&rd.cv); pthread_mutex_unlock(&rd.mu);
    pthread_join(rth, NULL);
    while (QY265EncoderDelayedFrames(h)) {
        err = QY265EncoderEncodeFrame(h, &nal, &nnal, NULL, &outp, 0);
        if (err) { fprintf(stderr, "EncodeFrame (flush) failed: 0x%08x\n", (unsigned)err); return 1; }
        for (int k = 0; k < nnal && fo; ++k) fwrite(nal[k].pPayload, 1, (size_t)nal[k].iSize, fo);
    }
    const double t1 = now_ms();
    ks265_enc_stats st;
    ks265_enc_get_stats(h, &st);
    printf("Total Frames: %ld, test time: %.0fms, FPS: %.4f\n", n, t1 - t0, n * 1000.0 / (t1 - t0));                 /* appencoder's two summary lines, same text */
    printf("Total Frames: %ld, pure encoding time: %.0fms, %.4f fps\n", n, t1 - t0 - t_io, n * 1000.0 / (t1 - t0 - t_io > 1e-3 ? t1 - t0 - t_io : 1e-3));
    printf("waiting for the input reader %.0f ms, slice writing %.1f ms per picture per thread\n", t_io, st.frames ? st.host_write_ms / st.frames : 0.0);
    printf("calling thread: input copy %.0f ms, enqueueing GPU work %.0f ms, output (wait + copy) %.0f ms\n", st.in_copy_ms, st.submit_ms, st.output_ms);
    printf("per picture: enqueue -> records on the host %.2f ms, enqueue -> writer pick-up %.2f ms\n", st.frames ? st.lat_gpu_ms / st.frames : 0.0, st.frames ? st.lat_queue_ms / st.frames : 0.0);
    QY265EncoderClose(h);                                            /* prints "bitrate, psnr: ..." */
    puts("H265 encoder passed!!!");
    free(rd.buf[0]); free(rd.buf[1]); fclose(fi); if (fo) fclose(fo);
    return 0;
}
