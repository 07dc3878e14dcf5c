/* TEST INFRASTRUCTURE - builder container only (needs the reference binary; never runs on the GPU box, never linked into the product).
 *
 * Records the reference encoder's sub-pel refinement on real encodes: inline hooks on the non-PIE ELF
 *   h265_codec::getMvResolution(TCtuInfo*, TPredUnit*, tME*)  enc@0x483ca0   - decides per PU whether the refinement runs (tME+0x3bc)
 *   h265_codec::subMeSquare(TCtuInfo*, TPredUnit*, tME*)      enc@0x4b5660   - half-sample step (subMeHpel_RealInterp enc@0x4b4e90), quarter-sample step
 *                                                                             (g_SubMeQpel_8Sad_RealInterpFunc enc@0x6ff6a0 -> enc@0x4b2bc0 / 0x4b3360 / 0x4b3b80 / 0x4b43a0),
 *                                                                             choice between the two AMVP predictors
 * The inner functions are NOT hooked (round 3: a hook below subMeSquare changes the encoder's stream; these two hooks leave it byte-identical -
 * gen_subme_traces.py checks that on every run).  One binary record per call goes to the file named by KS265_SM_DUMP; KS265_SM_SELECT (one byte per
 * subMeSquare call, as in me_trace_shim.c) limits the records with pixels to the selected calls.  Nothing of the reference is stored: inputs and
 * outputs of the two functions only (pixels of synthetic clips / of the encoder's reconstruction, vectors, costs, configuration words).
 *
 * Record = 64 int32 (fields below) [+ for kind 0: source block W*H, reference region (W+16)*(H+16) around the start position, 17 + 17 u16 rate table
 * entries around the start vector; for kind 1 with "searched" clear: source block and the (W+2)*(H+2) region the sad4 call reads]. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
typedef void (*fn3)(uint8_t *ctu, uint8_t *pu, uint8_t *me);
static fn3 g_orig_sq, g_orig_res;
static FILE *g_dump;
static uint8_t *g_sel; static long g_nsel;
static unsigned long g_call_sq, g_call_res;
#define RD(T, base, off) (*(T *)((base) + (off)))

/* rate of one vector component as subMeSquare computes it when tME+0x65 is set (enc@0x4b5b20..): table tME+0x10 around the predictor, beyond +-0x100 a
 * length formula times tME+0x80 */
static int32_t t10_rate(const uint8_t *me, int d)
{
    const uint16_t *t10 = RD(uint16_t *, me, 0x10);
    int a = d < 0 ? -d : d;
    if (a <= 0x100) return t10 ? t10[d] : 0;
    { int v = a * 2, n = 1; do { v >>= 1; n += 2; } while (v != 1); return (int32_t)((uint16_t)n * RD(int32_t, me, 0x80)); }
}

static void hook_sq(uint8_t *ctu, uint8_t *pu, uint8_t *me)
{
    const unsigned long idx = g_call_sq++;
    const uint8_t *cfg = RD(uint8_t *, ctu, 0);
    const int l2w = RD(int8_t, pu, 5), l2h = RD(int8_t, pu, 6), W = 1 << l2w, H = 1 << l2h;
    const uint8_t *fenc = RD(uint8_t *, me, 0x30);
    const int fstride = RD(int32_t, me, 0x38), stride = RD(int32_t, me, 0x50);
    const int mx = RD(int16_t, me, 0x54), my = RD(int16_t, me, 0x56), pidx = RD(int32_t, me, 0x58);
    const uint8_t *cur = RD(uint8_t *, me, 0x40);
    int32_t h[64]; memset(h, 0, sizeof h);
    const int want = g_dump && (!g_sel || ((long)idx < g_nsel && g_sel[idx]));
    uint8_t *fe = NULL, *reg = NULL; uint16_t cm[34];
    h[0] = 0x53554232; h[1] = 0; h[2] = (int32_t)idx; h[3] = W; h[4] = H; h[5] = mx; h[6] = my; h[7] = (int32_t)RD(uint32_t, me, 0x90);
    h[8] = RD(uint8_t, me, 0x64); h[9] = RD(uint8_t, me, 0x65); h[10] = RD(int32_t, me, 0x3bc); h[11] = RD(int32_t, me, 0x36c);
    h[12] = RD(int32_t, cfg, 0x464); h[13] = RD(int32_t, cfg, 0x580); h[14] = RD(int32_t, cfg, 0x568); h[15] = RD(int32_t, me, 0x3c4); h[16] = RD(int32_t, me, 0x60);
    h[17] = RD(int32_t, pu, 0x140);
    h[18] = RD(int16_t, pu, 0x1a0); h[19] = RD(int16_t, pu, 0x1a2); h[20] = RD(int16_t, pu, 0x1a4); h[21] = RD(int16_t, pu, 0x1a6);
    h[22] = pidx; h[23] = RD(int32_t, me, 0x2e0); h[24] = RD(int32_t, me, 0x2e4); h[25] = RD(int32_t, me, 0x80);
    h[26] = RD(void *, pu, 0x40) == RD(void *, pu, 0x38);                   /* sub-pel measure == integer measure (SAD); else Hadamard (satdInter) */
    h[27] = RD(int32_t, pu, 0xf8); h[28] = RD(int32_t, pu, 0xfc);
    if (want) {
        fe = malloc((size_t)W * H); reg = malloc((size_t)(W + 16) * (H + 16));
        for (int y = 0; y < H; ++y) memcpy(fe + y * W, fenc + (long)y * fstride, W);
        for (int y = 0; y < H + 16; ++y) memcpy(reg + y * (W + 16), cur + (long)(y - 8) * stride - 8, (size_t)W + 16);
        if (!h[9]) { const uint16_t *cmx = RD(uint16_t *, me, 0x18), *cmy = RD(uint16_t *, me, 0x20);
                     for (int d = -8; d <= 8; ++d) { cm[8 + d] = cmx[mx + d]; cm[25 + d] = cmy[my + d]; } }
        else { const int px = RD(int16_t, pu, 0x1a0 + 4 * pidx), py = RD(int16_t, pu, 0x1a2 + 4 * pidx);       /* tME+0x65: rates from tME+0x10 around the predictor */
               for (int d = -8; d <= 8; ++d) { cm[8 + d] = (uint16_t)t10_rate(me, mx + d - px); cm[25 + d] = (uint16_t)t10_rate(me, my + d - py); } }
    }
    g_orig_sq(ctu, pu, me);
    {
        const int ox = RD(int16_t, me, 0x54), oy = RD(int16_t, me, 0x56), o = RD(int32_t, me, 0x58);
        h[32] = ox; h[33] = oy; h[34] = (int32_t)RD(uint32_t, me, 0x90); h[35] = RD(int32_t, me, 0x94); h[36] = RD(int32_t, me, 0x98); h[37] = o;
        /* the rate terms the final predictor choice reads (table tME+0x10 relative to each of the two predictors) */
        h[38] = t10_rate(me, ox - h[18]); h[39] = t10_rate(me, oy - h[19]); h[40] = t10_rate(me, ox - h[20]); h[41] = t10_rate(me, oy - h[21]);
        h[42] = want ? 1 : 0;
    }
    if (g_dump) {
        fwrite(h, 4, 64, g_dump);
        if (want) { fwrite(fe, 1, (size_t)W * H, g_dump); fwrite(reg, 1, (size_t)(W + 16) * (H + 16), g_dump); fwrite(cm, 2, 34, g_dump); }
    }
    free(fe); free(reg);
}

static void hook_res(uint8_t *ctu, uint8_t *pu, uint8_t *me)
{
    const unsigned long idx = g_call_res++;
    const uint8_t *cfg = RD(uint8_t *, ctu, 0), *c8 = RD(uint8_t *, ctu, 8);
    const int l2w = RD(int8_t, pu, 5), l2h = RD(int8_t, pu, 6), W = 1 << l2w, H = 1 << l2h;
    const int mx = RD(int16_t, me, 0x54), my = RD(int16_t, me, 0x56);
    int32_t h[64]; memset(h, 0, sizeof h);
    const int searched = RD(uint8_t, me, 0x3b8);
    h[0] = 0x53554232; h[1] = 1; h[2] = (int32_t)idx; h[3] = W; h[4] = H; h[5] = mx; h[6] = my; h[7] = (int32_t)RD(uint32_t, me, 0x90);
    h[8] = RD(int32_t, cfg, 0x498); h[9] = RD(int32_t, cfg, 0x49c); h[10] = RD(uint8_t, me, 0x3c9); h[11] = RD(uint8_t, me, 0x65); h[12] = RD(int32_t, cfg, 0x464);
    h[13] = searched; h[14] = (int32_t)RD(uint32_t, me, 0x3a8); h[15] = (int32_t)RD(uint32_t, me, 0x3ac); h[16] = (int32_t)RD(uint32_t, me, 0x3b0); h[17] = (int32_t)RD(uint32_t, me, 0x3b4);
    h[18] = RD(int8_t, c8, 0x21); h[19] = RD(int32_t, me, 0x3c0); h[20] = RD(int32_t, me, 0x60);
    if (!h[11]) h[21] = RD(uint16_t *, me, 0x18)[mx] + RD(uint16_t *, me, 0x20)[my];              /* rate of the start vector */
    h[22] = (int32_t)g_call_sq;                                                                      /* the subMeSquare call this decision belongs to (the next one) */
    uint8_t *fe = NULL, *reg = NULL;
    const int pix = g_dump && !searched && h[10] && !h[11] && h[12] && !(h[8] && (uint32_t)((((6 - l2h) * h[9] + h[8]) << (2 * l2w))) < (uint32_t)h[7]);
    if (pix) {                                                                                       /* the function will call sad4 itself: keep what it reads */
        const uint8_t *fenc = RD(uint8_t *, me, 0x30), *cur = RD(uint8_t *, me, 0x40);
        const int fstride = RD(int32_t, me, 0x38), stride = RD(int32_t, me, 0x50);
        fe = malloc((size_t)W * H); reg = malloc((size_t)(W + 2) * (H + 2));
        for (int y = 0; y < H; ++y) memcpy(fe + y * W, fenc + (long)y * fstride, W);
        for (int y = 0; y < H + 2; ++y) memcpy(reg + y * (W + 2), cur + (long)(y - 1) * stride - 1, (size_t)W + 2);
    }
    g_orig_res(ctu, pu, me);
    h[32] = RD(int32_t, me, 0x3bc); h[33] = RD(int32_t, me, 0x94); h[42] = pix ? 1 : 0;
    h[34] = (int32_t)RD(uint32_t, me, 0x3a8); h[35] = (int32_t)RD(uint32_t, me, 0x3ac); h[36] = (int32_t)RD(uint32_t, me, 0x3b0); h[37] = (int32_t)RD(uint32_t, me, 0x3b4);
    if (g_dump) {
        fwrite(h, 4, 64, g_dump);
        if (pix) { fwrite(fe, 1, (size_t)W * H, g_dump); fwrite(reg, 1, (size_t)(W + 2) * (H + 2), g_dump); }
    }
    free(fe); free(reg);
}

static void put_jump(uint8_t *at, const void *target) { at[0] = 0x48; at[1] = 0xb8; memcpy(at + 2, &target, 8); at[10] = 0xff; at[11] = 0xe0; }   /* mov rax, imm64; jmp rax */
static uint8_t *g_tramp;
static void *install(uintptr_t addr, int displaced, const void *hook, int slot)
{
    uint8_t *fn = (uint8_t *)addr, *t = g_tramp + 64 * slot;
    const long page = sysconf(_SC_PAGESIZE);
    memcpy(t, fn, (size_t)displaced);
    { uint8_t *j = t + displaced; const void *back = fn + displaced; j[0] = 0xff; j[1] = 0x25; j[2] = j[3] = j[4] = j[5] = 0; memcpy(j + 6, &back, 8); }   /* jmp [rip+0]: no register touched */
    if (mprotect((uint8_t *)((uintptr_t)fn & ~(uintptr_t)(page - 1)), 2 * page, PROT_READ | PROT_WRITE | PROT_EXEC)) _exit(5);
    put_jump(fn, hook);
    for (int i = 12; i < displaced; ++i) fn[i] = 0x90;
    return t;
}
static void finish(void) { if (g_dump) fclose(g_dump); }
__attribute__((constructor)) static void ctor(void)
{
    const char *dp = getenv("KS265_SM_DUMP"), *sp = getenv("KS265_SM_SELECT");
    if (!dp) return;
    g_dump = fopen(dp, "wb");
    if (sp) { FILE *f = fopen(sp, "rb"); if (f) { fseek(f, 0, SEEK_END); g_nsel = ftell(f); fseek(f, 0, SEEK_SET); g_sel = malloc((size_t)g_nsel + 1); if (fread(g_sel, 1, (size_t)g_nsel, f) != (size_t)g_nsel) _exit(6); fclose(f); } }
    g_tramp = mmap(NULL, 4096, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    /* displaced prologues (whole instructions, none rip-relative):
     * subMeSquare     push r15; mov rax,rdi; push r14; push r13; push r12; push rbp; push rbx = 13 bytes
     * getMvResolution push r13; push r12; mov r12,rdi; push rbp; mov rbp,rsi; push rbx = 12 bytes */
    g_orig_sq = (fn3)install(0x4b5660, 13, (const void *)hook_sq, 0);
    g_orig_res = (fn3)install(0x483ca0, 12, (const void *)hook_res, 1);
    atexit(finish);
}
