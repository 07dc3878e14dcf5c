/* TEST INFRASTRUCTURE - builder container only (needs the reference binary; never runs on the GPU box, never linked into the product).
 *
 * Records the reference encoder's lookahead decisions on real encodes: inline hooks on the non-PIE ELF
 *   h265_codec::calcFrameAdaptQuant(TInputPic*, int mode, double strength)                         enc@0x4653c0   kind 1
 *   h265_codec::cuTreePropagate(int log2, TInputPic** frames, int p0, int p1, int b)   (.isra)     enc@0x47d460   kind 2
 *   h265_codec::scenecut(TEncParam*, TInputPic* prev, TInputPic* cur, int a, int b)                enc@0x47e9d0   kind 3
 * One binary record per call goes to KS265_LA_DUMP: 32 int32 header (h[0] magic, h[1] kind, h[2] payload bytes) + payload (see each hook).  gen_la_traces.py checks that
 * the hooks leave the stream byte-identical.  Nothing of the reference is stored: inputs and outputs of the functions only. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#define RD(T, base, off) (*(T *)((uint8_t *)(base) + (off)))
typedef void (*fn_aq)(uint8_t *pic, int mode, double strength);
typedef void (*fn_ct)(int lg, uint8_t **frames, int p0, int p1, int b);
typedef int (*fn_sc)(uint8_t *cfg, uint8_t *prev, uint8_t *cur, int a, int b);
static fn_aq g_aq; static fn_ct g_ct; static fn_sc g_sc;
static FILE *g_dump;
/* scratch for the copies: an arena mapped once - malloc / free inside the hooks would change what the encoder's own allocations find on the heap (a -rc 1 run came out different) */
static uint8_t *g_arena; static size_t g_used;
static void *take(size_t n) { void *p = g_arena + g_used; g_used += (n + 63) & ~(size_t)63; if (g_used > ((size_t)256 << 20)) _exit(6); return p; }

static void put(int kind, const int32_t *h32, const void *const *parts, const size_t *sizes, int n)
{
    int32_t h[32]; memcpy(h, h32, sizeof h);
    size_t tot = 0; for (int i = 0; i < n; ++i) tot += sizes[i];
    h[0] = 0x4c4f4f4b; h[1] = kind; h[2] = (int32_t)tot;
    fwrite(h, 4, 32, g_dump);
    for (int i = 0; i < n; ++i) fwrite(parts[i], 1, sizes[i], g_dump);
}

/* kind 1: header h[3] nx, h[4] ny, h[5] count, h[6] mode, h[8..9] strength (double), h[10..12] strides Y U V; payload: Y (16 ny rows x 16 nx), U, V (8 ny x 8 nx each; the V rows
 * are read at the U stride's offsets, as the function does), then the outputs: L+0x9a8 (count doubles), L+0x9b0 (count doubles), L+0x48 (count u16) */
static void hook_aq(uint8_t *pic, int mode, double strength)
{
    g_aq(pic, mode, strength);
    if (!g_dump || mode != 1) return;
    uint8_t *L = RD(uint8_t *, pic, 0x50), *fr = RD(uint8_t *, pic, 0);
    const int nx = RD(int32_t, L, 0xc), ny = RD(int32_t, L, 0x10), cnt = RD(int32_t, L, 0x14);
    const int sy = RD(int16_t, fr, 0x58), su = RD(int16_t, fr, 0x5a), sv = RD(int16_t, fr, 0x5c);
    if (nx <= 0 || ny <= 0) return;
    g_used = 0;
    uint8_t *Y = take((size_t)nx * ny * 256), *U = take((size_t)nx * ny * 64), *V = take((size_t)nx * ny * 64);
    for (int r = 0; r < ny * 16; ++r) memcpy(Y + (size_t)r * nx * 16, RD(uint8_t *, fr, 0x18) + (long)r * sy, (size_t)nx * 16);
    for (int r = 0; r < ny * 8; ++r) memcpy(U + (size_t)r * nx * 8, RD(uint8_t *, fr, 0x20) + (long)r * su, (size_t)nx * 8);
    /* V block (bx, by) starts at by * 8 * su + bx * 8 (the U stride), its rows advance by sv */
    for (int by = 0; by < ny; ++by) for (int r = 0; r < 8; ++r) memcpy(V + (size_t)(by * 8 + r) * nx * 8, RD(uint8_t *, fr, 0x28) + (long)by * 8 * su + (long)r * sv, (size_t)nx * 8);
    int32_t h[32] = {0}; h[3] = nx; h[4] = ny; h[5] = cnt; h[6] = mode; memcpy(&h[8], &strength, 8); h[10] = sy; h[11] = su; h[12] = sv;
    const void *parts[6] = {Y, U, V, RD(void *, L, 0x9a8), RD(void *, L, 0x9b0), RD(void *, L, 0x48)};
    const size_t sizes[6] = {(size_t)nx * ny * 256, (size_t)nx * ny * 64, (size_t)nx * ny * 64, (size_t)cnt * 8, (size_t)cnt * 8, (size_t)cnt * 2};
    put(1, h, parts, sizes, 6);
}

/* kind 2: header h[3] nx, h[4] ny, h[5] log2 argument, h[6] p0, h[7] p1, h[8] b, h[9] / h[10] = list 0 / 1 vectors present; payload (n = nx ny): intra u16[n] (L+0x30), inverse
 * qscale u16[n] (L+0x48), own propagate cost u16[n] (L+0x40), inter cost u16[n] (L+0x50[idx]), list bits u8[(n + 3) / 4] (L+0x2d8[idx]), vectors of list 0 / 1 int32[n] each
 * (zeros when absent), reference costs before: p0's and p1's L+0x40 u16[n] each; after: the same two */
static void hook_ct(int lg, uint8_t **frames, int p0, int p1, int b)
{
    if (!g_dump) { g_ct(lg, frames, p0, p1, b); return; }
    uint8_t *L = RD(uint8_t *, frames[b], 0x50), *L0 = RD(uint8_t *, frames[p0], 0x50), *L1 = RD(uint8_t *, frames[p1], 0x50);
    const int nx = RD(int32_t, L, 0xc), ny = RD(int32_t, L, 0x10), n = nx * ny;
    const int d0 = b - p0, d1 = p1 - b, idx = d1 + d0 * 9;
    int32_t h[32] = {0}; h[3] = nx; h[4] = ny; h[5] = lg; h[6] = p0; h[7] = p1; h[8] = b;
    g_used = 0;
    uint16_t *bef0 = take((size_t)n * 2), *bef1 = take((size_t)n * 2), *own = take((size_t)n * 2);
    int32_t *mv0 = take((size_t)n * 4), *mv1 = take((size_t)n * 4);
    memset(mv0, 0, (size_t)n * 4); memset(mv1, 0, (size_t)n * 4);
    memcpy(bef0, RD(void *, L0, 0x40), (size_t)n * 2); memcpy(bef1, RD(void *, L1, 0x40), (size_t)n * 2); memcpy(own, RD(void *, L, 0x40), (size_t)n * 2);
    if (d0 >= 1 && RD(void *, L, 0x560 + (d0 - 1) * 8)) { memcpy(mv0, RD(void *, L, 0x560 + (d0 - 1) * 8), (size_t)n * 4); h[9] = 1; }
    if (d1 >= 1 && RD(void *, L, 0x5a0 + (d1 - 1) * 8)) { memcpy(mv1, RD(void *, L, 0x5a0 + (d1 - 1) * 8), (size_t)n * 4); h[10] = 1; }
    uint16_t *intra = take((size_t)n * 2), *invq = take((size_t)n * 2), *inter = take((size_t)n * 2);
    uint8_t *bits = take((size_t)(n + 3) / 4);
    memcpy(intra, RD(void *, L, 0x30), (size_t)n * 2); memcpy(invq, RD(void *, L, 0x48), (size_t)n * 2);
    memcpy(inter, RD(void *, L, 0x50 + idx * 8), (size_t)n * 2); memcpy(bits, RD(void *, L, 0x2d8 + idx * 8), (size_t)(n + 3) / 4);
    g_ct(lg, frames, p0, p1, b);
    const void *parts[11] = {intra, invq, own, inter, bits, mv0, mv1, bef0, bef1, RD(void *, L0, 0x40), RD(void *, L1, 0x40)};
    const size_t sizes[11] = {(size_t)n * 2, (size_t)n * 2, (size_t)n * 2, (size_t)n * 2, (size_t)(n + 3) / 4, (size_t)n * 4, (size_t)n * 4, (size_t)n * 2, (size_t)n * 2, (size_t)n * 2, (size_t)n * 2};
    put(2, h, parts, sizes, 11);
}

/* kind 3: header h[3] return, h[4] a, h[5] b, h[6] cost of cur against prev (L+0x684 + 36 (b - a)), h[7] cur's intra cost (L+0x684), h[8] prev's intra cost, h[9] nx, h[10] ny,
 * h[11] cfg+0x3c0, h[12] cfg+0x390, h[13] cfg+0x50, h[14] cur+0x28, h[15] cfg+0x6e0; no payload */
static int hook_sc(uint8_t *cfg, uint8_t *prev, uint8_t *cur, int a, int b)
{
    const int ret = g_sc(cfg, prev, cur, a, b);
    if (g_dump) {
        uint8_t *L1 = RD(uint8_t *, cur, 0x50), *L0 = RD(uint8_t *, prev, 0x50);
        int32_t h[32] = {0};
        h[3] = ret; h[4] = a; h[5] = b; h[6] = RD(int32_t, L1, 0x684 + 36 * (b - a)); h[7] = RD(int32_t, L1, 0x684); h[8] = RD(int32_t, L0, 0x684);
        h[9] = RD(int32_t, L1, 0xc); h[10] = RD(int32_t, L1, 0x10); h[11] = RD(int32_t, cfg, 0x3c0); h[12] = RD(int32_t, cfg, 0x390); h[13] = RD(int32_t, cfg, 0x50);
        h[14] = RD(int32_t, cur, 0x28); h[15] = RD(int32_t, cfg, 0x6e0);
        put(3, h, NULL, NULL, 0);
    }
    return ret;
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
    const char *dp = getenv("KS265_LA_DUMP");
    if (!dp) return;
    g_dump = fopen(dp, "wb");
    g_arena = mmap(NULL, (size_t)256 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    setvbuf(g_dump, (char *)mmap(NULL, 1 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), _IOFBF, 1 << 20);
    g_tramp = mmap(NULL, 4096, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    /* displaced prologues (whole instructions, none rip-relative):
     *   calcFrameAdaptQuant  push r15; push r14; push r13; push r12; push rbp; push rbx; sub rsp,0x38                = 14 bytes
     *   cuTreePropagate      push r15; movsxd rax,r8d; add edi,2; push r14; push r13                                 = 12 bytes
     *   scenecut             push r13; xor r9d,r9d; mov r13,rsi; push r12; mov r12d,r8d                              = 13 bytes */
    const int mask = getenv("KS265_LA_HOOKS") ? atoi(getenv("KS265_LA_HOOKS")) : 7;
    if (mask & 1) g_aq = (fn_aq)install(0x4653c0, 14, (const void *)hook_aq, 0);
    if (mask & 2) g_ct = (fn_ct)install(0x47d460, 12, (const void *)hook_ct, 1);
    if (mask & 4) g_sc = (fn_sc)install(0x47e9d0, 13, (const void *)hook_sc, 2);
    atexit(finish);
}
