/* TEST INFRASTRUCTURE - builder container only (needs the reference binary; never runs on the GPU box, never linked into the product).
 *
 * Records the reference encoder's SAO decision on real encodes (VERDICT r5 missing 5): an inline hook on the non-PIE ELF
 *   h265_codec::CEncSao::modeDecisionCtu(TFrameInfo*)   enc@0x4af690
 * The original runs first; one record per call = what the function read and what it left:
 *   h[3] sao level (TEncParam+0x3d4), h[4] slice type (TFrameInfo+0xc), h[5] TFrameInfo+0xe8 (frame switch), h[6] TFrameInfo+0xe9, h[7] / h[8] left / upper CTU available, h[9] CTU index,
 *   h[10] / h[11] the availability record's bytes +0x14 / +0x15, h[12] / h[13] this+0x534 / +0x538 (switch-off masks), h[14] / h[15] lambda luma / chroma (this+0x520 / +0x524),
 *   h[16] this+0x528, h[17] / h[18] type masks this+0x53c / +0x540, h[19] this+0x544, h[20] picture width in CTUs (TEncParam+0x6c8), h[21] / h[22] best costs this+0x52c / +0x530 after
 *   payload: the object's statistics this[0 .. 0x4e0) (counts: band offset Y / U / V 32 each, edge classes Y / U / V 4 x 5 each; then the sums in the same layout), the CTU's parameter
 *   record after the call (32 bytes), the left and the upper CTU's records (32 bytes each; zeros where not available).
 * 64 int32 header (h[0] magic, h[1] kind 6, h[2] payload bytes) + payload.  gen_sao_traces.py checks that the hook leaves the stream byte-identical.  Nothing of the reference is
 * stored: inputs and outputs of the function only. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#define RD(T, base, off) (*(T *)((uint8_t *)(base) + (off)))
typedef void (*fn_md)(uint8_t *self, uint8_t *frame);
static fn_md g_md;
static FILE *g_dump;
static void hook_md(uint8_t *self, uint8_t *frame)
{
    g_md(self, frame);
    if (!g_dump) return;
    int32_t h[64] = {0};
    uint8_t *prm = RD(uint8_t *, self, 0x4e0), *ctu = RD(uint8_t *, self, 0x4f0), *av = RD(uint8_t *, ctu, 0x10), *out = RD(uint8_t *, self, 0x518);
    const int w = RD(int32_t, prm, 0x6c8);
    h[0] = 0x53414f31; h[1] = 6; h[2] = 0x4e0 + 96;
    h[3] = RD(int32_t, prm, 0x3d4); h[4] = RD(int32_t, frame, 0xc); h[5] = RD(uint8_t, frame, 0xe8); h[6] = RD(uint8_t, frame, 0xe9);
    h[7] = RD(int32_t, av, 0); h[8] = RD(int32_t, av, 4); h[9] = RD(int32_t, av, 8); h[10] = RD(int8_t, av, 0x14); h[11] = RD(int8_t, av, 0x15);
    h[12] = RD(int32_t, self, 0x534); h[13] = RD(int32_t, self, 0x538); h[14] = RD(int32_t, self, 0x520); h[15] = RD(int32_t, self, 0x524); h[16] = RD(int32_t, self, 0x528);
    h[17] = RD(int32_t, self, 0x53c); h[18] = RD(int32_t, self, 0x540); h[19] = RD(int32_t, self, 0x544); h[20] = w; h[21] = RD(int32_t, self, 0x52c); h[22] = RD(int32_t, self, 0x530);
    uint8_t nb[64];
    memset(nb, 0, sizeof nb);
    if (h[7]) memcpy(nb, out - 0x20, 32);
    if (h[8]) memcpy(nb + 32, out - (size_t)0x20 * (size_t)w, 32);
    fwrite(h, 4, 64, g_dump); fwrite(self, 1, 0x4e0, g_dump); fwrite(out, 1, 32, g_dump); fwrite(nb, 1, 64, g_dump);
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
    const char *dp = getenv("KS265_SAO_DUMP");
    if (!dp) return;
    g_dump = fopen(dp, "wb");
    g_tramp = mmap(NULL, 4096, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    /* displaced prologue (whole instructions, none rip-relative): push r12; push rbp; push rbx; mov rax,[rdi+0x4f0]; mov rbx,rdi = 14 bytes (checked by gen_sao_traces.py) */
    g_md = (fn_md)install(0x4af690, 14, (const void *)hook_md, 0);
    atexit(finish);
}
