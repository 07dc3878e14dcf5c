/* TEST INFRASTRUCTURE - builder container only (needs the reference binary; never runs on the GPU box, never linked into the product).
 *
 * Records the reference encoder's rate-distortion optimised quantisation on real encodes: inline hooks on the non-PIE ELF
 *   h265_codec::rdoQuant(short* lvl, short* coef, int log2, int, TTransUnit*, int scanIdx, int comp, TCtuInfo*)   enc@0x4aac50
 *   h265_codec::estBitRdoq(TEstBitsSbac&, unsigned log2, bool luma, const unsigned char* ctx)                      enc@0x46a8a0   (called inside rdoQuant: its output table is kept)
 * The context object behind TCtuInfo+0x6678 ("needs faking") is not faked: the calls are recorded inside real `appencoder` runs, where it is real.
 * One binary record per rdoQuant call goes to KS265_RQ_DUMP (every KS265_RQ_EVERY-th call, default 1): 64 int32 header, the levels before and after (N x N s16 each), the
 * coefficients (N x N s16), the 180-word bit table estBitRdoq built for this call, the first 0x1e8 bytes of the TTransUnit before and after.  gen_rdoq_traces.py checks that the hooks
 * leave the stream byte-identical.  Nothing of the reference is stored: inputs and outputs of the function only. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
typedef int (*fn_rq)(int16_t *lvl, int16_t *coef, int log2, int a4, uint8_t *tu, int scan, int comp, uint8_t *ctu);
typedef void (*fn_eb)(int32_t *out, unsigned log2, int luma, const uint8_t *ctx);
static fn_rq g_orig_rq; static fn_eb g_orig_eb;
static FILE *g_dump; static unsigned long g_call, g_every = 1;
static int32_t g_tab[180]; static int g_tab_log2, g_tab_luma, g_tab_n;
#define RD(T, base, off) (*(T *)((base) + (off)))

static void hook_eb(int32_t *out, unsigned log2, int luma, const uint8_t *ctx)
{
    g_orig_eb(out, log2, luma, ctx);
    memcpy(g_tab, out, sizeof g_tab); g_tab_log2 = (int)log2; g_tab_luma = luma & 1; ++g_tab_n;
}
static int hook_rq(int16_t *lvl, int16_t *coef, int log2, int a4, uint8_t *tu, int scan, int comp, uint8_t *ctu)
{
    const unsigned long idx = g_call++;
    const int n = 1 << log2, want = g_dump && idx % g_every == 0;
    int16_t before[32 * 32]; uint8_t tu_in[0x1e8];
    if (want) { memcpy(before, lvl, sizeof(int16_t) * (size_t)(n * n)); memcpy(tu_in, tu, sizeof tu_in); }   /* the function updates the significance masks in the TTransUnit */
    const int tabn0 = g_tab_n;
    const int ret = g_orig_rq(lvl, coef, log2, a4, tu, scan, comp, ctu);
    if (want) {
        const uint8_t *cfg = RD(uint8_t *, ctu, 0);
        const uint8_t *qpar = ctu + 0x2c + (comp ? 28 : 0);
        int32_t h[64]; memset(h, 0, sizeof h);
        h[0] = 0x52444f51; h[1] = (int32_t)idx; h[2] = log2; h[3] = a4; h[4] = scan; h[5] = comp; h[6] = ret;
        h[7] = RD(int32_t, ctu, 0x18); h[8] = RD(int32_t, cfg, 0x434); h[9] = RD(int32_t, cfg, 0x43c); h[10] = RD(uint8_t, ctu, 0xa4c0);
        for (int k = 0; k < 7; ++k) h[11 + k] = RD(int32_t, qpar, 4 * k);                 /* the QuantParam of this component (H265_GetBaseQuantParam enc@0x4a9c90) */
        { const double lam = ((const double *)0x6ff460)[h[7]]; memcpy(&h[20], &lam, 8); } /* the lambda table entry the function multiplies cfg+0x434 / +0x43c with */
        { const double lamc = ((const double *)0x6ff460)[RD(int32_t, ctu, 0x1c)]; memcpy(&h[31], &lamc, 8); }   /* the chroma QP's entry (TCtuInfo+0x1c) */
        h[33] = RD(int32_t, cfg, 0x430); h[34] = RD(int32_t, cfg, 0x3e0);                                         /* chroma multiplier; sign-data hiding on */
        h[22] = g_tab_n - tabn0; h[23] = g_tab_log2; h[24] = g_tab_luma;
        h[25] = RD(int32_t, cfg, 0x438); h[26] = RD(int32_t, cfg, 0x440); h[27] = RD(int32_t, ctu, 0x1c); h[28] = RD(int32_t, ctu, 0x20); h[29] = RD(int32_t, ctu, 0x24); h[30] = RD(int32_t, ctu, 0x28);
        fwrite(h, 4, 64, g_dump);
        fwrite(before, 2, (size_t)(n * n), g_dump); fwrite(lvl, 2, (size_t)(n * n), g_dump); fwrite(coef, 2, (size_t)(n * n), g_dump);
        fwrite(g_tab, 4, 180, g_dump); fwrite(tu_in, 1, 0x1e8, g_dump); fwrite(tu, 1, 0x1e8, g_dump);
    }
    return ret;
}
static void put_jump(uint8_t *at, const void *target) { at[0] = 0x48; at[1] = 0xb8; memcpy(at + 2, &target, 8); at[10] = 0xff; at[11] = 0xe0; }
static uint8_t *g_tramp;
static void *install(uintptr_t addr, int displaced, const void *hook, int slot)
{
    uint8_t *fn = (uint8_t *)addr, *t = g_tramp + 64 * slot;
    const long page = sysconf(_SC_PAGESIZE);
    memcpy(t, fn, (size_t)displaced);
    { uint8_t *j = t + displaced; const void *back = fn + displaced; j[0] = 0xff; j[1] = 0x25; j[2] = j[3] = j[4] = j[5] = 0; memcpy(j + 6, &back, 8); }
    if (mprotect((uint8_t *)((uintptr_t)fn & ~(uintptr_t)(page - 1)), 2 * page, PROT_READ | PROT_WRITE | PROT_EXEC)) _exit(5);
    put_jump(fn, hook);
    for (int i = 12; i < displaced; ++i) fn[i] = 0x90;
    return t;
}
static void finish(void) { if (g_dump) fclose(g_dump); }
__attribute__((constructor)) static void ctor(void)
{
    const char *dp = getenv("KS265_RQ_DUMP"), *ev = getenv("KS265_RQ_EVERY");
    if (!dp) return;
    g_dump = fopen(dp, "wb");
    if (ev && atol(ev) > 0) g_every = (unsigned long)atol(ev);
    g_tramp = mmap(NULL, 4096, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    /* displaced prologues (whole instructions, none rip-relative): rdoQuant push rbp; mov rbp,rsp; push r15; push r14; push r13; push r12 = 12 bytes;
     * estBitRdoq push r13; mov r13d,esi; push r12; mov r12,rcx; push rbp; movzx ebp,dl = 14 bytes */
    g_orig_rq = (fn_rq)install(0x4aac50, 12, (const void *)hook_rq, 0);
    g_orig_eb = (fn_eb)install(0x46a8a0, 14, (const void *)hook_eb, 1);
    atexit(finish);
}
