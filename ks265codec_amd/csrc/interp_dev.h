// interp_dev.h — luma fractional-sample prediction ON THE FLY from the reference picture: the normative 8-tap filters of
// interpLumaHor8to8_c enc@0x40e4f0 (fy = 0: (sum + 32) >> 6), interpLumaVer8to8_c enc@0x40f0c0 (fx = 0), and for both fractions
// interpLumaHor8to16_c enc@0x40eb80 (raw tap sums in 16 bits) followed by interpLumaVer16to8_c enc@0x4100b0 ((sum + 2048) >> 12) - what
// subMeHpel_RealInterp enc@0x4b4e90 / subMeQpel_8Sad_*_RealInterp compute per candidate.  Round 1 / 2 kept sixteen precomputed planes per
// reference picture (ks265_ref_planes: 159 MB written and 147 MB read per 2160p picture); every consumer now interpolates the 8 x 8 tile or
// the two tile rows it needs, from the padded picture itself (L2-resident), with the arithmetic the planes held - bit for bit.
#pragma once
#include "ks265_dev.h"

namespace ks265 {

// hipcc (ROCm 7.2) folds clip8(x >> s) pairs into gfx950's v_ashr_pk_u8_i32 and then ORs the packed pair with v_lshl_or_b32 assuming bits 31:16 of its result
// are zero; on MI355X they are not (measured in round 1: the upper two samples of every packed dword came back OR-contaminated).  An empty asm on the shifted
// value keeps the shift and the clamp apart, so the instruction is never selected.
__device__ __forceinline__ int ks_no_pk(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ int ks_pack4(const signed char *c)
{
    return (int)((unsigned)(unsigned char)c[0] | ((unsigned)(unsigned char)c[1] << 8) | ((unsigned)(unsigned char)c[2] << 16) | ((unsigned)(unsigned char)c[3] << 24));
}

// the luma taps of fraction f WITHOUT a table: a lane-varying index into kLumaTaps is a memory load per lane (and eight dependent byte loads for the vertical
// taps); selects between literals cost a few VALU operations and no latency.  packed: taps 0..3 / 4..7 as bytes (v_dot4 operands); single taps as integers.
__device__ __forceinline__ int ks_sel4(int f, int a, int b, int c, int d) { return f == 0 ? a : f == 1 ? b : f == 2 ? c : d; }
__device__ __forceinline__ void luma_taps_packed(int f, int &tl, int &th)
{
    tl = ks_sel4(f, 0x40000000, 0x3af604ff, 0x28f504ff, 0x11fb0100);     // {0 0 0 64} {-1 4 -10 58} {-1 4 -11 40} {0 1 -5 17}
    th = ks_sel4(f, 0x00000000, 0x0001fb11, (int)0xff04f528u, (int)0xff04f63au);     // {0 0 0 0} {17 -5 1 0} {40 -11 4 -1} {58 -10 4 -1}
}
__device__ __forceinline__ void luma_taps(int f, int (&c)[8])
{
    c[0] = ks_sel4(f, 0, -1, -1, 0); c[1] = ks_sel4(f, 0, 4, 4, 1); c[2] = ks_sel4(f, 0, -10, -11, -5); c[3] = ks_sel4(f, 64, 58, 40, 17);
    c[4] = ks_sel4(f, 0, 17, 40, 58); c[5] = ks_sel4(f, 0, -5, -11, -10); c[6] = ks_sel4(f, 0, 1, 4, 4); c[7] = ks_sel4(f, 0, 0, -1, -1);
}

// raw horizontal tap sums of 8 adjacent samples of one row: h[i] = sum_k taps[k] * row[i - 3 + k] (scale 64, fits 16 bits).
// v_dot4_i32_i8 on (sample - 128): the + 8192 restores the bias (the taps sum to 64).  tl / th = taps 0..3 / 4..7 packed.
// In two steps so that a caller can have the loads of its NEXT rows in flight while it filters these: luma_hrow8_load fetches the five aligned dwords that
// cover samples -3 .. +12 of the row, luma_hrow8_calc (sh = the address's low two bits, as luma_hrow8_shift gives them) does the arithmetic.
__device__ __forceinline__ unsigned luma_hrow8_shift(const uint8_t *row) { return (unsigned)((uintptr_t)(row - 3) & 3); }
__device__ __forceinline__ void luma_hrow8_load(const uint8_t *row, unsigned (&a)[5])
{
    const uint8_t *q = row - 3;
    const unsigned *p = (const unsigned *)(q - ((uintptr_t)q & 3));
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = p[i];
}
__device__ __forceinline__ void luma_hrow8_calc(const unsigned (&a)[5], unsigned sh, int tl, int th, int (&h)[8])
{
    const unsigned a0 = a[0] ^ 0x80808080u, a1 = a[1] ^ 0x80808080u, a2 = a[2] ^ 0x80808080u, a3 = a[3] ^ 0x80808080u, a4 = a[4] ^ 0x80808080u;
    unsigned w[4];
    w[0] = align_bytes(a1, a0, sh); w[1] = align_bytes(a2, a1, sh); w[2] = align_bytes(a3, a2, sh); w[3] = align_bytes(a4, a3, sh);   // bytes -3..0, 1..4, 5..8, 9..12
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = i >> 2, s = i & 3;
        const unsigned lo = s ? align_bytes(w[k + 1], w[k], s) : w[k], hi = s ? align_bytes(w[k + 2], w[k + 1], s) : w[k + 1];
        h[i] = __builtin_amdgcn_sdot4((int)hi, th, __builtin_amdgcn_sdot4((int)lo, tl, 8192, false), false);
    }
}
__device__ __forceinline__ void luma_hrow8(const uint8_t *row, int tl, int th, int (&h)[8])
{
    unsigned a[5];
    luma_hrow8_load(row, a);
    luma_hrow8_calc(a, luma_hrow8_shift(row), tl, th, h);
}

// The half-sample step around an INTEGER position x0 (me_subpel_kernel's phase H): the candidates at x0 - 1/2 and x0 + 1/2 are the same nine half samples, read one apart -
// h[j] = the raw tap sum of the half sample between x0 - 1 + j and x0 + j (j = 0 .. 8), from the five dwords luma_hrow8_load(row = x0 - 1) fetched; g[i] = 64 x the plain sample
// x0 + i (what the tap set {0 0 0 64 0 0 0 0} makes of it).  The same values luma_hrow8_calc returns for the three x positions, for 9 instead of 24 filtered samples.
__device__ __forceinline__ void luma_hrow9_half(const unsigned (&a)[5], unsigned sh, int (&h)[9], int (&g)[8])
{
    const unsigned a0 = a[0] ^ 0x80808080u, a1 = a[1] ^ 0x80808080u, a2 = a[2] ^ 0x80808080u, a3 = a[3] ^ 0x80808080u, a4 = a[4] ^ 0x80808080u;
    unsigned w[4];
    w[0] = align_bytes(a1, a0, sh); w[1] = align_bytes(a2, a1, sh); w[2] = align_bytes(a3, a2, sh); w[3] = align_bytes(a4, a3, sh);   // bytes -3..0, 1..4, 5..8, 9..12 of the row pointer = samples x0 - 4 .. x0 + 11
    const int tl = 0x28f504ff, th = (int)0xff04f528u;                                                                                       // {-1 4 -11 40} {40 -11 4 -1}
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = i >> 2, s = i & 3;
        const unsigned lo = s ? align_bytes(w[k + 1], w[k], s) : w[k], hi = s ? align_bytes(w[k + 2], w[k + 1], s) : w[k + 1];
        h[i] = __builtin_amdgcn_sdot4((int)hi, th, __builtin_amdgcn_sdot4((int)lo, tl, 8192, false), false);
    }
    h[8] = __builtin_amdgcn_sdot4((int)w[3], th, __builtin_amdgcn_sdot4((int)w[2], tl, 8192, false), false);
    const unsigned u1 = w[1] ^ 0x80808080u, u2 = w[2] ^ 0x80808080u;                                                                        // samples x0 .. x0 + 3, x0 + 4 .. x0 + 7
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = (int)(((u1 >> (8 * i)) & 255u) << 6); g[4 + i] = (int)(((u2 >> (8 * i)) & 255u) << 6); }
}

// 8 samples of one row, unfiltered (fx = 0)
__device__ __forceinline__ void luma_row8(const uint8_t *row, int (&h)[8])
{
    uint2 v;
    __builtin_memcpy(&v, row, 8);                                    // byte-aligned 8-byte load (global_load_dwordx2)
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = (int)((v.x >> (8 * i)) & 255u); h[4 + i] = (int)((v.y >> (8 * i)) & 255u); }
}

__device__ __forceinline__ uint2 ks_pack_row8(const int (&p)[8])
{
    return make_uint2((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24),
                      (unsigned)p[4] | ((unsigned)p[5] << 8) | ((unsigned)p[6] << 16) | ((unsigned)p[7] << 24));
}

// NR adjacent rows x 8 samples of the prediction whose first sample lies at integer position p[0] + (fx / 4, fy / 4); out[r] = row r packed.
// Input rows -3 .. NR + 3 are filtered horizontally once each and feed every output row that taps them.
template <int NR>
__device__ __forceinline__ void luma_pred_rows(const uint8_t *p, long stride, int fx, int fy, uint2 (&out)[NR])
{
    if (!(fx | fy)) {
#pragma unroll
        for (int r = 0; r < NR; ++r) __builtin_memcpy(&out[r], p + r * stride, 8);
        return;
    }
    int tl, th;
    luma_taps_packed(fx, tl, th);
    if (!fy) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            int h[8];
            luma_hrow8(p + r * stride, tl, th, h);
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = clip8(ks_no_pk((h[i] + 32) >> 6));
            out[r] = ks_pack_row8(h);
        }
        return;
    }
    int cy[8];
    luma_taps(fy, cy);
    int acc[NR][8];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r][i] = 0;
#pragma unroll
    for (int ir = 0; ir < NR + 7; ++ir) {                            // input row ir - 3
        int h[8];
        if (fx) {
            luma_hrow8(p + (ir - 3) * stride, tl, th, h);
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = (int)(short)h[i];     // the reference's 16-bit intermediate
        } else luma_row8(p + (ir - 3) * stride, h);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = ir - r;                                    // tap of output row r that lands on this input row
            if (t < 0 || t > 7) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[r][i] += cy[t] * h[i];
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        int px[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) px[i] = fx ? clip8(ks_no_pk((acc[r][i] + 2048) >> 12)) : clip8(ks_no_pk((acc[r][i] + 32) >> 6));
        out[r] = ks_pack_row8(px);
    }
}

// the 8 x 8 prediction tile of the block whose first sample is org[0] (a sample of the padded reference picture), displaced by the quarter-sample
// vector (qx, qy): 16 packed dwords, row r = t[2 r], t[2 r + 1]
__device__ __forceinline__ void luma_pred_tile8(const uint8_t *org, long stride, int qx, int qy, unsigned (&t)[16])
{
    const uint8_t *p = org + (long)(qy >> 2) * stride + (qx >> 2);
    const int fx = qx & 3, fy = qy & 3;
    // two halves of four rows: 32 accumulators live instead of 64 (the callers hold whole tiles in registers); the integer position is eight plain loads
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint2 rows[4];
        luma_pred_rows<4>(p + (long)(4 * half) * stride, stride, fx, fy, rows);
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[8 * half + 2 * r] = rows[r].x; t[8 * half + 2 * r + 1] = rows[r].y; }
    }
}

}  // namespace ks265
