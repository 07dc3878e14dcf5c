/* TEST INFRASTRUCTURE - builder container only (needs the reference binary; never runs on the GPU box, never linked into the product).
 *
 * Records the reference encoder's lookahead cost function on real encodes: inline hooks on the non-PIE ELF
 *   h265_codec::calcFrameCost(TEncParam*, TInputPic* ref0, TInputPic* ref1, TInputPic* cur, int d0, int d1, int flag)   enc@0x4a7410   kind 4
 *   h265_codec::CInputPicManage::addPicTobeEncoded(int count)                                                            enc@0x47f9a0   kind 5
 * kind 4: one record per call that COMPUTES (the function returns a stored sum when L+0x684[d0 * 9 + d1] >= 0): the words of TEncParam it reads, the three half-size planes
 * with a margin, every per-block array and sum it reads or writes, before and after.  kind 5: called right after the inlined cuTree finish (enc@0x480964..0x480a54) with the
 * pictures of the mini-GOP that leaves the lookahead: per picture the arrays that loop read (intra cost, inverse qscale, propagate cost, AQ offsets) and the offsets it left.
 * 64 int32 header (h[0] magic, h[1] kind, h[2] payload bytes) + payload.  gen_cfc_traces.py checks that the hooks leave the stream byte-identical.
 * Nothing of the reference is stored: inputs and outputs of the functions only. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#define RD(T, base, off) (*(T *)((uint8_t *)(base) + (off)))
typedef int (*fn_cfc)(uint8_t *cfg, uint8_t *r0, uint8_t *r1, uint8_t *cur, int d0, int d1, int flag);
typedef void (*fn_add)(uint8_t *self, int count);
static fn_cfc g_cfc; static fn_add g_add;
static FILE *g_dump;
static uint8_t *g_arena; static size_t g_used;
static void *take(size_t n) { void *p = g_arena + g_used; g_used += (n + 63) & ~(size_t)63; if (g_used > ((size_t)512 << 20)) _exit(6); return p; }
static void *snap(const void *src, size_t n) { void *p = take(n ? n : 1); if (src && n) memcpy(p, src, n); else if (n) memset(p, 0, n); return p; }

#define MX 36
#define MY 36
#define TABH 1024
#define TABW (2 * TABH + 1)
typedef struct { const void *p; size_t n; } part;
static void put(int kind, int32_t *h, const part *parts, int n)
{
    size_t tot = 0; for (int i = 0; i < n; ++i) tot += parts[i].n;
    h[0] = 0x43464331; h[1] = kind; h[2] = (int32_t)tot;
    fwrite(h, 4, 64, g_dump);
    for (int i = 0; i < n; ++i) if (parts[i].n) fwrite(parts[i].p, 1, parts[i].n, g_dump);
}
/* the plane of a picture's half-size layer with the margin: (h + 2 MY) rows of (w + 2 MX) samples */
static uint8_t *grab_plane(uint8_t *L, int w, int h)
{
    const int stride = RD(int32_t, L, 4);
    const uint8_t *p0 = RD(uint8_t *, L, 0x28);
    uint8_t *out = take((size_t)(w + 2 * MX) * (h + 2 * MY));
    for (int r = -MY; r < h + MY; ++r) memcpy(out + (size_t)(r + MY) * (w + 2 * MX), p0 + (long)r * stride - MX, (size_t)(w + 2 * MX));
    return out;
}
typedef struct { uint16_t *intra; uint8_t *imode; uint16_t *invq, *inter; uint8_t *bits; int32_t *mv0, *c0, *mv1, *c1; } arrays;
static arrays grab_arrays(uint8_t *L, int n, int d0, int d1)
{
    const int idx = d0 * 9 + d1;
    arrays a;
    a.intra = snap(RD(void *, L, 0x30), (size_t)n * 2); a.imode = snap(RD(void *, L, 0x38), (size_t)n); a.invq = snap(RD(void *, L, 0x48), (size_t)n * 2);
    a.inter = snap(d0 + d1 ? RD(void *, L, 0x50 + idx * 8) : NULL, (size_t)n * 2);
    a.bits = snap(RD(void *, L, 0x2d8 + idx * 8), (size_t)(n + 3) / 4);
    a.mv0 = snap(d0 ? RD(void *, L, 0x560 + (d0 - 1) * 8) : NULL, (size_t)n * 4); a.c0 = snap(d0 ? RD(void *, L, 0x5e0 + (d0 - 1) * 8) : NULL, (size_t)n * 4);
    a.mv1 = snap(d1 ? RD(void *, L, 0x5a0 + (d1 - 1) * 8) : NULL, (size_t)n * 4); a.c1 = snap(d1 ? RD(void *, L, 0x620 + (d1 - 1) * 8) : NULL, (size_t)n * 4);
    return a;
}
static int add_arrays(part *ps, int k, const arrays *a, int n)
{
    ps[k++] = (part){a->intra, (size_t)n * 2}; ps[k++] = (part){a->imode, (size_t)n}; ps[k++] = (part){a->invq, (size_t)n * 2}; ps[k++] = (part){a->inter, (size_t)n * 2};
    ps[k++] = (part){a->bits, (size_t)(n + 3) / 4}; ps[k++] = (part){a->mv0, (size_t)n * 4}; ps[k++] = (part){a->c0, (size_t)n * 4}; ps[k++] = (part){a->mv1, (size_t)n * 4};
    ps[k++] = (part){a->c1, (size_t)n * 4};
    return k;
}

/* kind 4 header: h[3] d0, h[4] d1, h[5] flag, h[6] return; h[7] w, h[8] h, h[9] nx, h[10] ny, h[11] L+0x14, h[12] / h[13] L+0x18 before / after; h[14..17] poc of cur / ref0 / ref1, cur+0x20;
 * h[18..30] TEncParam +0x710 +0x3c0 +0x3a0 +0x3a4 +0x390 +0xc +0x8 +0x378 +0x388 +0x3a8 +0x36c +0x538 +0x3b4; h[31] lambda (u16 at (cfg+0x720)+0x18); h[32] / h[33] list 0 / 1 is searched in
 * this call; h[34] MX, h[35] MY; h[36..40] before: L+0x660[d0], L+0x684[0], L+0x7c8[0], L+0x684[idx], L+0x7c8[idx]; h[41..45] the same after; h[46..49] / h[50..53] L+0x90c+16 d0 (4 words)
 * before / after; h[54] / h[55] ref0 / ref1 given, h[56] row length of the mvd cost table.  payload: u16[52] lambda table (TEncParam+0x720), u16[2049] mvd cost table around tME+0x10 (-1024..1024); planes cur, ref0, ref1 (absent ones as one zero byte); the nine arrays before;
 * the nine arrays after (intra u16, intra mode u8, inverse qscale u16, inter cost u16, list bits u8[(n + 3) / 4], list-0 vectors / costs i32, list-1 vectors / costs i32) */
static int hook_cfc(uint8_t *cfg, uint8_t *r0, uint8_t *r1, uint8_t *cur, int d0, int d1, int flag)
{
    uint8_t *L = RD(uint8_t *, cur, 0x50);
    const int idx = d0 * 9 + d1;
    if (!g_dump || RD(int32_t, L, 0x684 + idx * 4) >= 0) return g_cfc(cfg, r0, r1, cur, d0, d1, flag);
    const int w = RD(int32_t, L, 0), h = RD(int32_t, L, 8), nx = RD(int32_t, L, 0xc), ny = RD(int32_t, L, 0x10), n = nx * ny;
    int32_t hd[64] = {0};
    g_used = 0;
    hd[3] = d0; hd[4] = d1; hd[5] = flag; hd[7] = w; hd[8] = h; hd[9] = nx; hd[10] = ny; hd[11] = RD(int32_t, L, 0x14); hd[12] = RD(int32_t, L, 0x18);
    hd[14] = RD(int32_t, cur, 0x28); hd[15] = r0 ? RD(int32_t, r0, 0x28) : -1; hd[16] = r1 ? RD(int32_t, r1, 0x28) : -1; hd[17] = RD(int32_t, cur, 0x20);
    static const int kCfg[13] = {0x710, 0x3c0, 0x3a0, 0x3a4, 0x390, 0xc, 0x8, 0x378, 0x388, 0x3a8, 0x36c, 0x538, 0x3b4};
    for (int i = 0; i < 13; ++i) hd[18 + i] = RD(int32_t, cfg, kCfg[i]);
    hd[31] = RD(uint16_t, RD(uint8_t *, cfg, 0x720), 0x18);
    hd[32] = d0 && RD(int32_t, RD(uint8_t *, L, 0x560 + (d0 - 1) * 8), 0) == 0x7fff;
    hd[33] = d1 && RD(int32_t, RD(uint8_t *, L, 0x5a0 + (d1 - 1) * 8), 0) == 0x7fff;
    hd[34] = MX; hd[35] = MY;
    hd[36] = RD(int32_t, L, 0x660 + d0 * 4); hd[37] = RD(int32_t, L, 0x684); hd[38] = RD(int32_t, L, 0x7c8); hd[39] = RD(int32_t, L, 0x684 + idx * 4); hd[40] = RD(int32_t, L, 0x7c8 + idx * 4);
    memcpy(&hd[46], L + 0x90c + d0 * 16, 16);
    hd[54] = r0 != NULL; hd[55] = r1 != NULL;
    uint16_t *tab = take(TABW * 2), *lam = take(52 * 2);
    {   /* tME+0x10 as the function forms it (enc@0x4a7622..0x4a7663): row 12 of the u16 table [52][m] (m = 8 merange + 33), centre; the search indexes it without a range test, so the
         * neighbouring rows are recorded with it (entries outside the allocation as 0xffff) */
        uint8_t *ct = RD(uint8_t *, cfg, 0x778);
        const int m = RD(int32_t, ct, 0x28);
        const uint16_t *t0 = RD(uint16_t *, ct, 8);
        const long c0 = 12L * m + (m >> 1), tot = 52L * m;
        for (int d = -TABH; d <= TABH; ++d) tab[d + TABH] = (c0 + d >= 0 && c0 + d < tot) ? t0[c0 + d] : 0xffff;
        memcpy(lam, RD(uint8_t *, cfg, 0x720), 52 * 2);
        hd[56] = m;
    }
    static uint8_t zero1[1];
    uint8_t *pc = grab_plane(L, w, h), *p0 = r0 ? grab_plane(RD(uint8_t *, r0, 0x50), w, h) : zero1, *p1 = r1 ? grab_plane(RD(uint8_t *, r1, 0x50), w, h) : zero1;
    const size_t psz = (size_t)(w + 2 * MX) * (h + 2 * MY);
    const arrays before = grab_arrays(L, n, d0, d1);
    const int ret = g_cfc(cfg, r0, r1, cur, d0, d1, flag);
    const arrays after = grab_arrays(L, n, d0, d1);
    hd[6] = ret; hd[13] = RD(int32_t, L, 0x18);
    hd[41] = RD(int32_t, L, 0x660 + d0 * 4); hd[42] = RD(int32_t, L, 0x684); hd[43] = RD(int32_t, L, 0x7c8); hd[44] = RD(int32_t, L, 0x684 + idx * 4); hd[45] = RD(int32_t, L, 0x7c8 + idx * 4);
    memcpy(&hd[50], L + 0x90c + d0 * 16, 16);
    part ps[32]; int k = 0;
    ps[k++] = (part){lam, 52 * 2}; ps[k++] = (part){tab, TABW * 2}; ps[k++] = (part){pc, psz}; ps[k++] = (part){p0, r0 ? psz : 1}; ps[k++] = (part){p1, r1 ? psz : 1};
    k = add_arrays(ps, k, &before, n); k = add_arrays(ps, k, &after, n);
    put(4, hd, ps, k);
    return ret;
}

/* kind 5, one record per picture handed on: h[3] poc, h[4] cur+0x20, h[5] cur+0x38 (byte), h[6] cur+0x68, h[7] cur+0x6c, h[8] cur+0x121 (byte), h[9] nx, h[10] ny, h[11] L+0x14,
 * h[12] TEncParam+0x378, h[13] +0x388, h[14] +0x35c, h[15] position in the batch, h[16] batch size, h[17] L+0x684[0]; payload: intra u16[n], inverse qscale u16[n], propagate cost u16[n] (L+0x40),
 * AQ offsets f64[cnt] (L+0x9a8), final offsets f64[cnt] (L+0x9b0) */
static void hook_add(uint8_t *self, int count)
{
    if (g_dump) {
        uint8_t *cfg = RD(uint8_t *, self, 8);
        uint8_t **pics = RD(uint8_t **, self, 0x30);
        for (int i = 0; i < count; ++i) {
            uint8_t *pic = pics[i], *L = RD(uint8_t *, pic, 0x50);
            const int nx = RD(int32_t, L, 0xc), ny = RD(int32_t, L, 0x10), n = nx * ny, cnt = RD(int32_t, L, 0x14);
            if (!RD(void *, L, 0x9b0) || n <= 0) continue;
            int32_t hd[64] = {0};
            hd[3] = RD(int32_t, pic, 0x28); hd[4] = RD(int32_t, pic, 0x20); hd[5] = RD(uint8_t, pic, 0x38); hd[6] = RD(int32_t, pic, 0x68); hd[7] = RD(int32_t, pic, 0x6c); hd[8] = RD(uint8_t, pic, 0x121);
            hd[9] = nx; hd[10] = ny; hd[11] = cnt; hd[12] = RD(int32_t, cfg, 0x378); hd[13] = RD(int32_t, cfg, 0x388); hd[14] = RD(int32_t, cfg, 0x35c); hd[15] = i; hd[16] = count; hd[17] = RD(int32_t, L, 0x684);
            static uint8_t zeros[1 << 20];
            const part ps[5] = {{RD(void *, L, 0x30) ? RD(void *, L, 0x30) : (void *)zeros, (size_t)n * 2}, {RD(void *, L, 0x48) ? RD(void *, L, 0x48) : (void *)zeros, (size_t)n * 2},
                                {RD(void *, L, 0x40) ? RD(void *, L, 0x40) : (void *)zeros, (size_t)n * 2}, {RD(void *, L, 0x9a8) ? RD(void *, L, 0x9a8) : (void *)zeros, (size_t)cnt * 8},
                                {RD(void *, L, 0x9b0), (size_t)cnt * 8}};
            put(5, hd, ps, 5);
        }
    }
    g_add(self, count);
}

static uint8_t *g_tramp;
static void *install(uintptr_t addr, int displaced, const void *hook, int slot)
{
    uint8_t *fn = (uint8_t *)addr, *t = g_tramp + 64 * slot;
    const long page = sysconf(_SC_PAGESIZE);
    memcpy(t, fn, (size_t)displaced);
    { uint8_t *j = t + displaced; const void *back = fn + displaced; j[0] = 0xff; j[1] = 0x25; j[2] = j[3] = j[4] = j[5] = 0; memcpy(j + 6, &back, 8); }
    if (mprotect((uint8_t *)((uintptr_t)fn & ~(uintptr_t)(page - 1)), 2 * page, PROT_READ | PROT_WRITE | PROT_EXEC)) _exit(5);
    fn[0] = 0x48; fn[1] = 0xb8; memcpy(fn + 2, &hook, 8); fn[10] = 0xff; fn[11] = 0xe0;
    for (int i = 12; i < displaced; ++i) fn[i] = 0x90;
    return t;
}
static void finish(void) { if (g_dump) fclose(g_dump); }
__attribute__((constructor)) static void ctor(void)
{
    const char *dp = getenv("KS265_CFC_DUMP");
    if (!dp) return;
    g_dump = fopen(dp, "wb");
    g_arena = mmap(NULL, (size_t)512 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    setvbuf(g_dump, (char *)mmap(NULL, 1 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), _IOFBF, 1 << 20);
    g_tramp = mmap(NULL, 4096, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    /* displaced prologues (whole instructions, none rip-relative):
     *   calcFrameCost        push rbp; mov rbp,rsp; push r15; push r14; push r13; push r12                           = 12 bytes
     *   addPicTobeEncoded    see gen_cfc_traces.py (checked there against the bytes of the file)                     */
    const int mask = getenv("KS265_CFC_HOOKS") ? atoi(getenv("KS265_CFC_HOOKS")) : 3;
    const int add_len = getenv("KS265_CFC_ADDLEN") ? atoi(getenv("KS265_CFC_ADDLEN")) : 0;
    if (mask & 1) g_cfc = (fn_cfc)install(0x4a7410, 12, (const void *)hook_cfc, 0);
    if ((mask & 2) && add_len >= 12) g_add = (fn_add)install(0x47f9a0, add_len, (const void *)hook_add, 1);
    atexit(finish);
}
