/* stub of the encoder API for testing the CLI's reader thread without a GPU: checks that pictures arrive complete and in order */
#include "ks265_enc.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
const char strLibQy265Version[] = "stub";
static long g_n; static int g_w, g_h;
int QY265ConfigDefaultPreset(QY265EncConfig *c, char *p, char *t, char *l) { (void)p; (void)t; (void)l; memset(c, 0, sizeof *c); return 0; }
int QY265ConfigParse(QY265EncConfig *c, const char *name, const char *value) { if (!strcmp(name, "wdt")) c->picWidth = atoi(value); else if (!strcmp(name, "hgt")) c->picHeight = atoi(value); return 0; }
void *QY265EncoderOpen(QY265EncConfig *c, int *err) { g_w = c->picWidth; g_h = c->picHeight; *err = 0; return &g_n; }
int QY265EncoderEncodeFrame(void *h, QY265Nal **n, int *nn, QY265Picture *in, QY265Picture *out, int f)
{
    (void)h; (void)out; (void)f; *nn = 0; *n = NULL;
    if (in) {
        const unsigned char *p = in->yuv->pData[0];
        /* every picture of the test file starts with its index in the first 4 bytes and is filled with index & 255 */
        unsigned idx = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24);
        if ((long)idx != g_n || in->pts != g_n) { fprintf(stderr, "picture %ld arrived as %u (pts %lld)\n", g_n, idx, in->pts); exit(9); }
        const size_t fsz = (size_t)g_w * g_h * 3 / 2;
        for (size_t i = 4; i < fsz; i += 997) if (p[i] != (unsigned char)(g_n & 255)) { fprintf(stderr, "picture %ld torn at %zu\n", g_n, i); exit(9); }
        if (in->yuv->pData[1] != p + (size_t)g_w * g_h) exit(8);
        ++g_n;
    }
    return 0;
}
int QY265EncoderDelayedFrames(void *h) { (void)h; return 0; }
void QY265EncoderClose(void *h) { (void)h; printf("stub saw %ld pictures\n", g_n); }
int ks265_enc_get_stats(void *h, ks265_enc_stats *o) { (void)h; memset(o, 0, sizeof *o); o->frames = g_n; return 0; }
int ks265_enc_set_recon_file(void *h, const char *p) { (void)h; (void)p; return 0; }
int ks265_enc_set_default(const char *n, int v) { (void)n; (void)v; return 0; }
