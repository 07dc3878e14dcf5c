/* TEST INFRASTRUCTURE: a CPU stand-in for the subset of libks265hip.so the encoder host (ks265codec_amd/host/ks265_enc.c) calls, so that the host's threads,
 * rings, GOP scheduling, GOP lanes, graph cache and bitstream writer can be exercised by `-m "not gpu"` tests.  It is NOT a CPU fallback of the product (nothing in
 * ks265codec_amd/ knows it) and it does not encode: every "picture" gets a fixed, valid set of records - 8x8 CUs, DC intra in key pictures, otherwise one or two vectors
 * per CU, and a few levels in every fifth block, all derived from hashes of the source and reference pictures (packed into the compact record format like
 * the device does) - so that the stream the writer produces depends on which pictures met in which
 * order, which is what the host tests look at.  Streams are synchronous (everything runs inside the call), events are always complete; a captured "graph" is the
 * list of the recorded calls, replayed by ks265_graph_launch. */
#include "ks265_hip.h"
#include "../oracle/ks265_pipeline_oracle.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ks265_ctx { int capturing; struct Op *ops; int nops; };
typedef struct Op { int kind; ks265_frame *f; const void *a; ks265_pic p0, p1, p2, p3; int i0; void *dst; } Op;
typedef struct Graph { Op *ops; int n; } Graph;
struct ks265_frame {
    ks265_ctx *ctx; ks265_frame_cfg cfg; ks265_frame_geom g;
    int cur_pu, have_prev;
    ks265_cu8 *cu8; ks265_sao_param *sao; uint64_t kind_hash, rq_hash;
    int me0, tools0[3];                                            /* intra_inter, bi_refine, sao as created (ks265_frame_set_picture_tools) */
    int16_t *lvl[3];                                          /* level planes, W x H and two W/2 x H/2, packed */
};

const char *ks265_version(void) { return "ks265hip CPU stub (tests only)"; }
const char *ks265_last_error(ks265_ctx *c) { (void)c; return "stub"; }
int ks265_create(ks265_ctx **out, int device);
int ks265_create_prio(ks265_ctx **out, int device, int high_priority) { (void)high_priority; return ks265_create(out, device); }
int ks265_create(ks265_ctx **out, int device)
{
    const char *nd = getenv("KS265_STUB_DEVICES");                     /* how many GPUs the stand-in "has" (default 8) */
    if (getenv("KS265_STUB_NO_DEVICE") || device < 0 || device >= (nd ? atoi(nd) : 8)) return KS265_NO_DEVICE;
    if (getenv("KS265_STUB_LOG_DEVICES")) fprintf(stderr, "stub: context on device %d\n", device);
    *out = (ks265_ctx *)calloc(1, sizeof **out);
    return *out ? KS265_OK : KS265_OUTOFMEMORY;
}
void ks265_destroy(ks265_ctx *c) { if (c) { free(c->ops); free(c); } }
int ks265_synchronize(ks265_ctx *c) { (void)c; return KS265_OK; }
int ks265_take_device_error(ks265_ctx *c)                              /* KS265_STUB_DEVERR_AT = k: the k-th call finds the device error word set (once), like a wavefront time-out */
{
    static int n, at = -2;
    (void)c;
    if (at == -2) { const char *e = getenv("KS265_STUB_DEVERR_AT"); at = e ? atoi(e) : -1; }
    return at >= 0 && __atomic_fetch_add(&n, 1, __ATOMIC_RELAXED) == at ? KS265_FAIL : KS265_OK;
}
int ks265_dev_malloc(ks265_ctx *c, void **p, size_t n) { (void)c; *p = calloc(1, n ? n : 1); return *p ? KS265_OK : KS265_OUTOFMEMORY; }
int ks265_dev_free(ks265_ctx *c, void *p) { (void)c; free(p); return KS265_OK; }
int ks265_host_malloc(ks265_ctx *c, void **p, size_t n) { return ks265_dev_malloc(c, p, n); }
int ks265_host_register(ks265_ctx *c, void *p, size_t n) { (void)c; (void)p; (void)n; return getenv("KS265_STUB_NO_REGISTER") ? KS265_FAIL : KS265_OK; }   /* (every byte of the stand-in's host is "DMA-able") */
int ks265_host_unregister(ks265_ctx *c, void *p) { (void)c; (void)p; return KS265_OK; }
int ks265_memcpy_h2d_sync(ks265_ctx *c, void *d, const void *s, size_t n) { (void)c; memcpy(d, s, n); return KS265_OK; }
int ks265_host_free(ks265_ctx *c, void *p) { return ks265_dev_free(c, p); }
static int stub_fast(void);
int ks265_memcpy_h2d_async(ks265_ctx *c, void *d, const void *s, size_t n) { (void)c; if (n > (1u << 20) && stub_fast()) return KS265_OK;   /* (KS265_STUB_FAST: a picture's upload is the copy engine's time, not the caller's) */
    memcpy(d, s, n); return KS265_OK; }
int ks265_memcpy_d2d_async(ks265_ctx *c, void *d, const void *s, size_t n) { (void)c; memcpy(d, s, n); return KS265_OK; }
int ks265_memcpy_d2h_async(ks265_ctx *c, void *d, const void *s, size_t n) { (void)c; memcpy(d, s, n); return KS265_OK; }
int ks265_memset_async(ks265_ctx *c, void *d, int v, size_t n) { (void)c; memset(d, v, n); return KS265_OK; }
int ks265_event_create(ks265_ctx *c, void **ev) { (void)c; *ev = malloc(4); return *ev ? KS265_OK : KS265_OUTOFMEMORY; }
int ks265_event_record(ks265_ctx *c, void *ev) { (void)c; *(int *)ev = 0; return KS265_OK; }
int ks265_event_wait(ks265_ctx *c, void *ev) { (void)c; (void)ev; return KS265_OK; }
/* KS265_STUB_EVENT_LAG = n: an event is reported done only at the n-th query after its record (the device stand-in runs everything at once: this is how the host's
 * "not ready yet" paths get exercised) */
int ks265_event_query(ks265_ctx *c, void *ev, int *done)
{
    (void)c;
    static int lag = -2147483647;
    if (lag == -2147483647) lag = getenv("KS265_STUB_EVENT_LAG") ? atoi(getenv("KS265_STUB_EVENT_LAG")) : 0;
    int *n = (int *)ev;
    if (lag >= 0) *done = *n >= lag;
    else {                                                              /* negative: a seed - every query is a coin toss (about one in three says done), so that the host's two
                                                                         * threads that look at the lookahead's queue meet it in every state */
        static unsigned long long st = 0;
        if (!st) st = (unsigned long long)(-lag) * 0x9E3779B97F4A7C15ull + 1;
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;                  /* (races on st between threads only add to the noise) */
        *done = (st % 3) == 0 || *n > 40;
    }
    ++*n;
    return KS265_OK;
}
int ks265_stream_wait_event(ks265_ctx *c, void *ev) { (void)c; (void)ev; return KS265_OK; }
int ks265_event_destroy(ks265_ctx *c, void *ev) { (void)c; free(ev); return KS265_OK; }

int ks265_frame_geometry(const ks265_frame_cfg *cfg, ks265_frame_geom *geom)
{
    kso_frame_cfg oc; kso_frame_geom og;
    memset(&oc, 0, sizeof oc);
    oc.width = cfg->width; oc.height = cfg->height; oc.qp = cfg->qp; oc.me_range = cfg->me_range;
    if (kso_frame_geometry(&oc, &og)) return KS265_NOTSUPPORTED;
    geom->pad_y = og.pad_y; geom->pad_c = og.pad_c; geom->stride_y = og.stride_y; geom->stride_c = og.stride_c; geom->rows_y = og.rows_y; geom->rows_c = og.rows_c;
    geom->bytes_y = og.bytes_y; geom->bytes_c = og.bytes_c; geom->ctu_cols = og.ctu_cols; geom->ctu_rows = og.ctu_rows; geom->pu_per_ctu = og.pu_per_ctu;
    geom->bytes_pu = og.bytes_pu; geom->bytes_cu8 = og.bytes_cu8; geom->bytes_sao = og.bytes_sao;
    return KS265_OK;
}
int ks265_frame_create(ks265_ctx *ctx, const ks265_frame_cfg *cfg, ks265_frame **out)
{
    ks265_frame *f = (ks265_frame *)calloc(1, sizeof *f);
    if (!f) return KS265_OUTOFMEMORY;
    f->ctx = ctx; f->cfg = *cfg;
    f->me0 = cfg->me_method; f->tools0[0] = cfg->intra_inter; f->tools0[1] = cfg->bi_refine; f->tools0[2] = cfg->sao;
    if (ks265_frame_geometry(cfg, &f->g)) { free(f); return KS265_NOTSUPPORTED; }
    f->cu8 = (ks265_cu8 *)calloc(1, (size_t)f->g.bytes_cu8); f->sao = (ks265_sao_param *)calloc(1, (size_t)f->g.bytes_sao);
    const size_t npx = (size_t)cfg->width * cfg->height;
    f->lvl[0] = (int16_t *)calloc(npx, 2); f->lvl[1] = (int16_t *)calloc(npx / 4 + 1, 2); f->lvl[2] = (int16_t *)calloc(npx / 4 + 1, 2);
    *out = f;
    return KS265_OK;
}
void ks265_frame_destroy(ks265_frame *f) { if (f) { free(f->cu8); free(f->sao); free(f->lvl[0]); free(f->lvl[1]); free(f->lvl[2]); free(f); } }
int ks265_frame_set_qp(ks265_frame *f, int qp, int l) { f->cfg.qp = qp; f->cfg.lambda_q4 = l; return KS265_OK; }
/* tools per picture: the stand-in's pictures do not depend on them; KS265_STUB_TOOLS_LOG = file: one line per inter picture handed in - kind, the three values - so that a
 * host test sees which pictures the host lowered them for */
int ks265_frame_set_picture_tools(ks265_frame *f, int ii, int br, int so, int me)
{
    if (me > 2) return KS265_NOTSUPPORTED;
    f->cfg.me_method = me < 0 ? f->me0 : me;
    const int v[3] = {ii, br, so};
    for (int i = 0; i < 3; ++i) {
        const int x = v[i] < 0 ? f->tools0[i] : v[i];
        if (x && x != f->tools0[i]) return KS265_NOTSUPPORTED;
    }
    f->cfg.intra_inter = ii < 0 ? f->tools0[0] : ii; f->cfg.bi_refine = br < 0 ? f->tools0[1] : br; f->cfg.sao = so < 0 ? f->tools0[2] : so;
    return KS265_OK;
}
static void tools_log(const ks265_frame *f, char kind)
{
    const char *p = getenv("KS265_STUB_TOOLS_LOG");
    if (!p) return;
    FILE *fp = fopen(p, "a");
    if (!fp) return;
    fprintf(fp, "%c %d %d %d %d\n", kind, f->cfg.intra_inter, f->cfg.bi_refine, f->cfg.sao, f->cfg.me_method);
    fclose(fp);
}
int ks265_frame_p_state(ks265_frame *f) { return (f->cur_pu & 1) | (f->have_prev ? 2 : 0); }
int ks265_frame_p_advance(ks265_frame *f) { f->cur_pu ^= 1; f->have_prev = 1; return KS265_OK; }
int ks265_frame_p_restore(ks265_frame *f, int s) { f->cur_pu = s & 1; f->have_prev = (s >> 1) & 1; return KS265_OK; }
int ks265_frame_reset_prediction(ks265_frame *f) { f->have_prev = 0; return KS265_OK; }

static uint64_t hash_bytes(const uint8_t *p, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; } return h; }
static uint8_t *luma0(const ks265_frame *f, ks265_pic p) { return p.y + (size_t)f->g.pad_y * f->g.stride_y + f->g.pad_y; }
/* a picture is identified by the first two rows of its luma plane (load_i420 puts the source there, the "encode" below writes its own identity there) */
static uint64_t pic_id(const ks265_frame *f, ks265_pic p) { return hash_bytes(luma0(f, p), (size_t)f->cfg.width) ^ (hash_bytes(luma0(f, p) + f->g.stride_y, (size_t)f->cfg.width) << 1); }

static void do_load(ks265_frame *f, const uint8_t *i420, ks265_pic dst)
{
    const int W = f->cfg.width, H = f->cfg.height;
    for (int y = 0; y < H; ++y) memcpy(luma0(f, dst) + (size_t)y * f->g.stride_y, i420 + (size_t)y * W, (size_t)W);
    const uint8_t *u = i420 + (size_t)W * H, *v = u + (size_t)W * H / 4;
    for (int y = 0; y < H / 2; ++y) {
        memcpy(dst.u + (size_t)(f->g.pad_c + y) * f->g.stride_c + f->g.pad_c, u + (size_t)y * (W / 2), (size_t)W / 2);
        memcpy(dst.v + (size_t)(f->g.pad_c + y) * f->g.stride_c + f->g.pad_c, v + (size_t)y * (W / 2), (size_t)W / 2);
    }
}
static int stub_fast(void) { static int fast = -1; if (fast < 0) fast = getenv("KS265_STUB_FAST") ? 1 : 0; return fast; }
static void do_encode(ks265_frame *f, int kind /*0 key, 1 P, 2 B*/, ks265_pic src, ks265_pic r0, ks265_pic r1, ks265_pic out, int state_at_capture)
{
    const int w8 = f->cfg.width / 8, h8 = f->cfg.height / 8;
    const uint64_t hs = pic_id(f, src), h0 = kind ? pic_id(f, r0) : 0, h1 = kind == 2 ? pic_id(f, r1) : 0;
    /* of the frame state only "a previous P picture exists" may show in the result (the PU ping-pong buffer is an implementation detail: two lanes are at different
     * parities for the same picture) */
    const uint64_t mix = hs ^ (h0 * 3) ^ (h1 * 5) ^ ((uint64_t)f->cfg.qp << 40) ^ ((uint64_t)((state_at_capture >> 1) & 1) << 50) ^ (kind ? f->rq_hash * 11 : 0);
    /* KS265_STUB_FAST (tools/caller_ceiling.py): a "device" that costs the host next to nothing - every block a skipped 8x8 CU, no levels, no reconstruction - so that at
     * 2160p with eight lanes the CALLING thread becomes the limit and its ceiling can be read off */
    if (stub_fast()) {
        ks265_cu8 c; memset(&c, 0, sizeof c);
        c.log2_cu = 3; if (kind == 0) { c.pred_mode = 2; c.mvx = 1; } else c.inter_dir = 1;
        for (int i = 0; i < w8 * h8; ++i) f->cu8[i] = c;
        for (int i = 0; i < f->g.ctu_cols * f->g.ctu_rows * 3; ++i) { memset(&f->sao[i], 0, sizeof f->sao[i]); f->sao[i].type = -1; }
        f->kind_hash = mix;
        return;
    }
    for (int i = 0; i < w8 * h8; ++i) {
        ks265_cu8 c; memset(&c, 0, sizeof c);
        c.log2_cu = 3;
        if (kind == 0) { c.pred_mode = 2; c.mvx = 1; }                                   /* DC */
        else {
            const uint64_t m = mix ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull);
            c.inter_dir = kind == 2 ? (uint8_t)(1 + (m >> 7) % 3) : 1;
            c.mvx = (int16_t)((m & 15) - 8); c.mvy = (int16_t)(((m >> 4) & 7) - 4);
            c.mv1x = (int16_t)(((m >> 8) & 15) - 8); c.mv1y = (int16_t)(((m >> 12) & 7) - 4);
        }
        /* a few levels in about every fifth block (luma 8x8, chroma 4x4), the coded-block flags to match: the host's expansion of the compact records and the
         * writer's residual coding get something to do; which blocks and what values depends on the pictures that met, like the vectors */
        const uint64_t r = mix ^ ((uint64_t)i * 0xD1B54A32D192ED03ull);
        const int W = f->cfg.width, bx = i % w8, by = i / w8;
        int16_t *ly = f->lvl[0] + (size_t)by * 8 * W + bx * 8, *lu = f->lvl[1] + (size_t)by * 4 * (W / 2) + bx * 4, *lv = f->lvl[2] + (size_t)by * 4 * (W / 2) + bx * 4;
        for (int y = 0; y < 8; ++y) memset(ly + (size_t)y * W, 0, 16);
        for (int y = 0; y < 4; ++y) { memset(lu + (size_t)y * (W / 2), 0, 8); memset(lv + (size_t)y * (W / 2), 0, 8); }
        if ((r >> 20) % 5 == 0) {
            const int n = 1 + (int)((r >> 24) & 3);
            for (int k = 0; k < n; ++k) { const int pos = (int)((r >> (28 + 6 * k)) & 63), v = (int)((r >> (52 + 2 * k)) & 3) + 1; ly[(size_t)(pos >> 3) * W + (pos & 7)] = (int16_t)((k & 1) ? -v : v); }
            c.cbf |= 1;
            if ((r >> 60) & 1) { lu[(size_t)((r >> 8) & 3) * (W / 2) + ((r >> 10) & 3)] = (int16_t)(1 + ((r >> 12) & 1)); c.cbf |= 2; }
            if ((r >> 61) & 1) { lv[(size_t)((r >> 14) & 3) * (W / 2) + ((r >> 16) & 3)] = (int16_t)-(1 + ((r >> 18) & 1)); c.cbf |= 4; }
        }
        f->cu8[i] = c;
    }
    for (int i = 0; i < f->g.ctu_cols * f->g.ctu_rows * 3; ++i) { memset(&f->sao[i], 0, sizeof f->sao[i]); f->sao[i].type = -1; }
    /* the "reconstruction": the source with the identity of this coding step stamped into the first two rows */
    const int W = f->cfg.width, H = f->cfg.height;
    for (int y = 0; y < H; ++y) memmove(luma0(f, out) + (size_t)y * f->g.stride_y, luma0(f, src) + (size_t)y * f->g.stride_y, (size_t)W);
    for (int x = 0; x < W; ++x) { luma0(f, out)[x] = (uint8_t)(mix >> (8 * (x & 7))); luma0(f, out)[f->g.stride_y + x] = (uint8_t)((mix * 7) >> (8 * (x & 7))); }
    f->kind_hash = mix;
}
int ks265_frame_compact_layout(ks265_frame *f, size_t off[8]);
static void do_pack(ks265_frame *f, uint8_t *dst, const uint64_t *extra)
{
    size_t off[8];
    ks265_frame_compact_layout(f, off);
    const size_t npx = (size_t)f->cfg.width * f->cfg.height, pb[3] = {npx * 2, npx / 2, npx / 2};
    size_t first[4]; first[0] = 0;
    for (int p = 0; p < 3; ++p) first[p + 1] = first[p] + (pb[p] + 63) / 64;
    const size_t nlines = first[3], nchunk = (nlines + 1023) / 1024;
    memcpy(dst + off[0], f->cu8, (size_t)f->g.bytes_cu8); memcpy(dst + off[1], f->sao, (size_t)f->g.bytes_sao);
    if (extra) memcpy(dst + off[2], extra, 64); else memset(dst + off[2], 0, 64);
    uint32_t *hdr = (uint32_t *)(dst + off[3]), *table = (uint32_t *)(dst + off[4]);
    uint64_t *bm = (uint64_t *)(dst + off[5]);
    uint8_t *data = dst + off[6];
    memset(bm, 0, nchunk * 128);
    uint32_t stored = 0;
    if (stub_fast()) memset(table, 0, nchunk * 4);               /* (no levels: nothing stored, nothing to look through) */
    else
    for (size_t L = 0; L < nlines; ++L) {                        /* the stored lines in line order; the device packs chunk by chunk in any order, the table says where */
        if ((L & 1023) == 0) table[L >> 10] = stored;
        const int p = L >= first[2] ? 2 : L >= first[1] ? 1 : 0;
        const size_t o = (L - first[p]) * 64, n = pb[p] - o < 64 ? pb[p] - o : 64;
        const uint8_t *src = (const uint8_t *)f->lvl[p] + o;
        int nz = 0;
        for (size_t k = 0; k < n; ++k) nz |= src[k];
        if (!nz) continue;
        bm[L >> 6] |= 1ull << (L & 63);
        memset(data + (size_t)stored * 64, 0, 64); memcpy(data + (size_t)stored * 64, src, n);
        ++stored;
    }
    memset(hdr, 0, 64); hdr[2] = stored; hdr[3] = (uint32_t)nlines;
}
int ks265_frame_compact_layout(ks265_frame *f, size_t off[8])
{
    const size_t npx = (size_t)f->cfg.width * f->cfg.height, pb[3] = {npx * 2, npx / 2, npx / 2};
    size_t nlines = 0;
    for (int i = 0; i < 3; ++i) nlines += (pb[i] + 63) / 64;
    const size_t nchunk = (nlines + 1023) / 1024;
    const size_t sz[7] = {(size_t)f->g.bytes_cu8, (size_t)f->g.bytes_sao, 64, 64, nchunk * 4, nchunk * 128, nlines * 64};
    size_t o = 0;
    for (int i = 0; i < 7; ++i) { off[i] = o; o += (sz[i] + 255) & ~(size_t)255; }
    off[7] = o;
    return KS265_OK;
}

/* ---- calls that a capture records */
enum { OP_LOAD, OP_ENC, OP_SSE, OP_PACK };
static int run_op(const Op *o)
{
    switch (o->kind) {
    case OP_LOAD: if (!stub_fast()) do_load(o->f, (const uint8_t *)o->a, o->p0); break;
    case OP_ENC: do_encode(o->f, o->i0 & 3, o->p0, o->p1, o->p2, o->p3, o->i0 >> 2); break;
    case OP_SSE: { uint64_t *d = (uint64_t *)o->dst; d[0] = o->f->kind_hash & 0xFFFFF; d[1] = 17; d[2] = 23; break; }
    case OP_PACK: do_pack(o->f, (uint8_t *)o->dst, (const uint64_t *)o->a); break;
    }
    return KS265_OK;
}
static int issue(ks265_ctx *c, Op o)
{
    if (!c->capturing) return run_op(&o);
    Op *n = (Op *)realloc(c->ops, (size_t)(c->nops + 1) * sizeof *n);
    if (!n) return KS265_OUTOFMEMORY;
    c->ops = n; c->ops[c->nops++] = o;
    return KS265_OK;
}
int ks265_capture_begin(ks265_ctx *c) { if (getenv("KS265_STUB_NO_CAPTURE")) return KS265_FAIL; c->capturing = 1; c->nops = 0; return KS265_OK; }
int ks265_capture_end(ks265_ctx *c, void **exec)
{
    *exec = NULL;
    if (!c->capturing) return KS265_FAIL;
    c->capturing = 0;
    if (getenv("KS265_STUB_NO_INSTANTIATE")) { c->nops = 0; return KS265_FAIL; }
    Graph *g = (Graph *)malloc(sizeof *g);
    g->ops = (Op *)malloc((size_t)(c->nops ? c->nops : 1) * sizeof(Op)); memcpy(g->ops, c->ops, (size_t)c->nops * sizeof(Op)); g->n = c->nops; c->nops = 0;
    *exec = g;
    return KS265_OK;
}
int ks265_graph_launch(ks265_ctx *c, void *exec) { (void)c; const Graph *g = (const Graph *)exec; for (int i = 0; i < g->n; ++i) run_op(&g->ops[i]); return KS265_OK; }
int ks265_graph_destroy(ks265_ctx *c, void *exec) { (void)c; Graph *g = (Graph *)exec; if (g) { free(g->ops); free(g); } return KS265_OK; }

int ks265_load_i420(ks265_frame *f, const uint8_t *i420, ks265_pic dst) { Op o = {OP_LOAD, f, i420, dst, dst, dst, dst, 0, NULL}; return issue(f->ctx, o); }
int ks265_load_i420_on(ks265_ctx *c, ks265_frame *f, const uint8_t *i420, ks265_pic dst) { (void)c; return ks265_load_i420(f, i420, dst); }   /* the stand-in runs every call at once: streams do not exist */
int ks265_frame_set_records_fence(ks265_frame *f, void *ev) { (void)f; (void)ev; return KS265_OK; }
int ks265_frame_set_qp_map(ks265_frame *f, const int8_t *m) { (void)f; (void)m; return KS265_OK; }
/* -rdoq 1: the stand-in has no quantiser; what the tables hold shows in the "coded" picture, so that a test sees whether the tables a picture gets depend on thread timing */
int ks265_frame_set_rdoq(ks265_frame *f, const int32_t *t, const int64_t *lam, const int64_t *lam_sdh)
{
    f->rq_hash = 0;
    if (t && lam && lam_sdh) f->rq_hash = hash_bytes((const uint8_t *)t, 1440 * 4) ^ (hash_bytes((const uint8_t *)lam, 52 * 8) << 1) ^ (hash_bytes((const uint8_t *)lam_sdh, 52 * 8) << 2) ^ 1;
    return KS265_OK;
}
/* -aq: the offsets by the oracle's restatement of calcFrameAdaptQuant on packed copies of the planes, the CTU map by the oracle's rule (the stand-in's pictures do not use it) */
#include "../oracle/ks265_lookahead_ref.h"
int ks265_frame_adapt_quant(ks265_ctx *c, const uint8_t *y, int sy, const uint8_t *u, const uint8_t *v, int sc, int nx, int ny, int count, double strength, double *off, uint16_t *inv, double *scratch)
{
    (void)c; (void)scratch;
    uint8_t *Y = malloc((size_t)nx * ny * 256), *U = malloc((size_t)nx * ny * 64), *V = malloc((size_t)nx * ny * 64);
    for (int r = 0; r < ny * 16; ++r) memcpy(Y + (size_t)r * nx * 16, y + (long)r * sy, (size_t)nx * 16);
    for (int r = 0; r < ny * 8; ++r) { memcpy(U + (size_t)r * nx * 8, u + (long)r * sc, (size_t)nx * 8); memcpy(V + (size_t)r * nx * 8, v + (long)r * sc, (size_t)nx * 8); }
    kso_ref_frame_adapt_quant(Y, U, V, nx, ny, count, strength, off, inv);
    free(Y); free(U); free(V);
    return KS265_OK;
}
int ks265_aq_ctu_map(ks265_ctx *c, const double *off, int nx, int ny, int base, int lo, int hi, int8_t *map) { (void)c; kso_aq_ctu_map(off, nx, ny, base, lo, hi, map); return KS265_OK; }
/* cuTree (-rc 3): the oracle's restatements of calcFrameCost, cuTreePropagate and the finish ARE the stand-in's "device" operators, so the host's pass over the lookahead window can
 * be held against the Python mirror of that pass picture for picture (tests/test_host_pipeline_cpu.py) */
size_t ks265_calc_frame_cost_workspace(int nx, int ny) { (void)nx; (void)ny; return 64; }
int ks265_calc_frame_cost(ks265_ctx *c, const ks265_cfc_params *q, const uint8_t *cur, const uint8_t *ref0, const uint8_t *ref1, uint16_t *intra, uint8_t *imode, const uint16_t *invq,
                          uint16_t *inter, uint8_t *bits, int32_t *mv0, int32_t *c0, int32_t *mv1, int32_t *c1, ks265_cfc_sums *s, void *ws)
{
    (void)c; (void)ws;
    kso_cfc k; memset(&k, 0, sizeof k);
    k.cur = cur; k.ref0 = ref0; k.ref1 = ref1; k.stride = q->stride; k.w = q->w; k.h = q->h; k.nx = q->nx; k.ny = q->ny; k.cnt = q->cnt;
    k.d0 = q->d0; k.d1 = q->d1; k.flag = q->flag; k.slice_type = q->slice_type;
    k.merange = q->merange; k.lg = q->lg; k.zero_thr = q->zero_thr; k.fast_intra = q->fast_intra; k.scenecut = q->scenecut; k.preset = q->preset; k.p8 = q->p8; k.aq = q->aq;
    k.b_intra = q->b_intra; k.f3a8 = q->f3a8; k.f36c = q->f36c; k.f538 = q->f538; k.f3b4 = q->f3b4; k.lambda_tab = q->lambda_tab;
    k.do_list[0] = q->d0 ? q->do_list[0] : 0; k.do_list[1] = q->d1 ? q->do_list[1] : 0; k.intra_done = q->intra_done;
    k.intra = intra; k.imode = imode; k.invq = invq; k.inter = inter; k.bits = bits; k.mv[0] = mv0; k.mv[1] = mv1; k.cost[0] = c0; k.cost[1] = c1;
    k.intra_wins = s->intra_wins; k.sum_intra = s->sum_intra; k.sum_intra_aq = s->sum_intra_aq; k.sum = -1; k.sum_aq = -1; memcpy(k.stats, s->stats, sizeof k.stats);
    if (q->d0 + q->d1 == 0) k.sum_intra = -1;                        /* (the operator always computes: the "already there" shortcut is the caller's) */
    k.margin_x = k.margin_y = 40;
    kso_ref_calc_frame_cost(&k);
    s->intra_wins = k.intra_wins; s->sum_intra = k.sum_intra; s->sum_intra_aq = k.sum_intra_aq; s->sum = q->d0 + q->d1 ? k.sum : k.sum_intra; s->sum_aq = q->d0 + q->d1 ? k.sum_aq : k.sum_intra_aq;
    memcpy(s->stats, k.stats, sizeof k.stats); s->ret = k.ret; s->intra_done = k.intra_done;
    return k.oob ? KS265_FAIL : KS265_OK;
}
int ks265_cutree_propagate(ks265_ctx *c, int lg, int nx, int ny, const uint16_t *intra, const uint16_t *invq, const uint16_t *own, const uint16_t *inter, const uint8_t *bits, const int32_t *mv0,
                           const int32_t *mv1, uint16_t *ref0, uint16_t *ref1, uint64_t *acc)
{
    (void)c; (void)acc;
    kso_ref_cutree_propagate(lg, nx, ny, intra, invq, own, inter, bits, mv0, mv1, ref0, ref1);
    return KS265_OK;
}
int ks265_cutree_finish(ks265_ctx *c, int cnt, const uint16_t *intra, const uint16_t *invq, const uint16_t *prop, const double *aq, int dbl, double *out) { (void)c; kso_ref_cutree_finish(cnt, intra, invq, prop, aq, dbl, out);
    return KS265_OK; }
int ks265_pad_plane(ks265_ctx *c, uint8_t *p, int stride, int w, int h, int pad)
{
    (void)c;
    for (int y = -pad; y < h + pad; ++y)
        for (int x = -pad; x < w + pad; ++x)
            if (x < 0 || x >= w || y < 0 || y >= h) p[(long)y * stride + x] = p[(long)(y < 0 ? 0 : y >= h ? h - 1 : y) * stride + (x < 0 ? 0 : x >= w ? w - 1 : x)];
    return KS265_OK;
}
int ks265_fill_u16(ks265_ctx *c, uint16_t *d, int n, int v) { (void)c; for (int i = 0; i < n; ++i) d[i] = (uint16_t)v; return KS265_OK; }
int ks265_qoff_ctu_map(ks265_ctx *c, const double *off, int nx, int ny, int lg, int cols, int rows, int base, int lo, int hi, int8_t *map) { (void)c; kso_qoff_ctu_map(off, nx, ny, lg, cols, rows, base, lo, hi, map); return KS265_OK; }
int ks265_store_i420(ks265_frame *f, ks265_pic src, uint8_t *i420)
{
    const int W = f->cfg.width, H = f->cfg.height;
    for (int y = 0; y < H; ++y) memcpy(i420 + (size_t)y * W, luma0(f, src) + (size_t)y * f->g.stride_y, (size_t)W);
    memset(i420 + (size_t)W * H, 128, (size_t)W * H / 2);
    return KS265_OK;
}
static int fail_now(void)                                      /* KS265_STUB_FAIL_AT = k: the k-th picture enqueued in this process fails like a launch error */
{
    static int n, at = -2;
    if (at == -2) { const char *e = getenv("KS265_STUB_FAIL_AT"); at = e ? atoi(e) : -1; }
    return at >= 0 && __atomic_fetch_add(&n, 1, __ATOMIC_RELAXED) == at;
}
int ks265_encode_picture(ks265_frame *f, ks265_pic src, ks265_pic ref, int is_key, ks265_pic out)
{
    if (fail_now()) return KS265_FAIL;
    tools_log(f, is_key ? 'I' : 'P');
    Op o = {OP_ENC, f, NULL, src, ref, ref, out, (is_key ? 0 : 1) | (is_key && getenv("KS265_STUB_B_STATELESS") ? 0 : ks265_frame_p_state(f) << 2), NULL};   /* (a key picture on the main stream: see below) */
    const int r = issue(f->ctx, o);
    if (!is_key) { f->cur_pu ^= 1; f->have_prev = 1; } else f->have_prev = 0;
    return r;
}
int ks265_encode_picture_b(ks265_frame *f, ks265_pic src, ks265_pic r0, ks265_pic r1, ks265_pic out)
{
    /* (the P chain's state goes into the record so that a replayed graph with the wrong state shows; KS265_STUB_B_STATELESS leaves it out - the real B pictures do not depend
     * on it, and the host's anchor lane moves the P chain to another frame object) */
    tools_log(f, 'B');
    Op o = {OP_ENC, f, NULL, src, r0, r1, out, 2 | (getenv("KS265_STUB_B_STATELESS") ? 0 : ks265_frame_p_state(f) << 2), NULL};
    return issue(f->ctx, o);
}
int ks265_encode_picture_mref(ks265_frame *f, ks265_pic src, const ks265_pic *refs, int nref, ks265_pic out) { return ks265_encode_picture(f, src, refs[nref - 1], 0, out); }
int ks265_encode_picture_b_mref(ks265_frame *f, ks265_pic src, const ks265_pic *refs0, int n0, const ks265_pic *refs1, int n1, ks265_pic out) { (void)n0; (void)n1; return ks265_encode_picture_b(f, src, refs0[0], refs1[0], out); }
int ks265_sse_picture(ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *sse3) { Op o = {OP_SSE, f, NULL, a, b, b, b, 0, sse3}; return issue(f->ctx, o); }
int ks265_frame_pack_compact(ks265_frame *f, void *dst, const void *extra) { Op o = {OP_PACK, f, extra, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, 0, dst}; return issue(f->ctx, o); }
int ks265_copy_out_compact_dma_async(ks265_ctx *c, ks265_frame *f, void *host, const void *dev, size_t n) { (void)n; return ks265_copy_out_compact_async(c, f, host, dev); }
int ks265_sse_picture_on(ks265_ctx *c, ks265_frame *f, ks265_pic a, ks265_pic b, uint64_t *sse3) { (void)c; return ks265_sse_picture(f, a, b, sse3); }
int ks265_frame_pack_compact_on(ks265_ctx *c, ks265_frame *f, void *dst, const void *extra) { (void)c; return ks265_frame_pack_compact(f, dst, extra); }
int ks265_copy_out_compact_async(ks265_ctx *c, ks265_frame *f, void *host, const void *dev)
{
    (void)c;
    size_t off[8];
    ks265_frame_compact_layout(f, off);
    memcpy(host, dev, off[6] + (size_t)((const uint32_t *)((const uint8_t *)dev + off[3]))[2] * 64);     /* the fixed part + the stored lines */
    return KS265_OK;
}

/* ---- scene-cut lookahead (host: -lookahead N): the half-size picture is real (2x2 averages), the two frame costs are stand-ins with the right behaviour - "intra" = how
 * far the samples are from mid-grey, "inter" = how far they are from the reference picture's co-located samples (no search), growing faster than linearly with that distance */
int ks265_downsample_rect(ks265_ctx *c, const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h);
void ks265o_downsample(uint8_t *dst, const uint8_t *src, int dstStride, int srcStride, int w, int h);
int ks265_downsample_from_host(ks265_ctx *c, const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h) { return ks265_downsample_rect(c, src, ss, dst, ds, w, h); }
int ks265_downsample_rect(ks265_ctx *c, const uint8_t *src, int ss, uint8_t *dst, int ds, int w, int h)
{
    (void)c;
    if (stub_fast()) return KS265_OK;                           /* (a real device takes these launches asynchronously: nothing of them is the calling thread's time) */
    ks265o_downsample(dst, src, ds, ss, w, h);                  /* downsample_c enc@0x4a6a60, as the device operator */
    return KS265_OK;
}
int ks265_pad_picture(ks265_frame *f, ks265_pic pic) { (void)f; (void)pic; return KS265_OK; }
int ks265_lookahead_picture(ks265_frame *f, ks265_pic cur, ks265_pic ref, uint32_t *ws, uint64_t *out)
{
    (void)ws;
    if (stub_fast()) { out[0] = 1000; out[1] = 100; out[2] = 100; out[3] = (uint64_t)f->cfg.width * f->cfg.height / 64; return KS265_OK; }
    const long org = (long)f->g.pad_y * f->g.stride_y + f->g.pad_y;
    uint64_t intra = 0, inter = 0;
    for (int y = 0; y < f->cfg.height; ++y)
        for (int x = 0; x < f->cfg.width; ++x) {
            const int a = cur.y[org + (long)y * f->g.stride_y + x], b = ref.y[org + (long)y * f->g.stride_y + x];
            intra += (uint64_t)(a > 128 ? a - 128 : 128 - a); inter += (uint64_t)(a > b ? a - b : b - a);
        }
    inter += inter * inter / ((uint64_t)f->cfg.width * f->cfg.height * 8);      /* like a search with a window: a change twice as large costs more than twice as much */
    out[0] = intra; out[1] = inter; out[2] = intra < inter ? intra : inter; out[3] = (uint64_t)f->cfg.width * f->cfg.height / 64;
    return KS265_OK;
}
int ks265_lookahead_inter(ks265_frame *f, ks265_pic cur, ks265_pic ref, const uint32_t *ws, uint64_t *out) { return ks265_lookahead_picture(f, cur, ref, (uint32_t *)ws, out); }
