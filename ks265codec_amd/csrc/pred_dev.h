// pred_dev.h — the normative fractional-sample prediction of 4 adjacent samples (frame_recon.hip's reconstruction and frame_skip.hip's skip pass form
// their predictions with the same code: what the skip pass writes as a node's reconstruction is what a decoder predicts)
#pragma once
#include "recon_dev.h"

namespace ks265 {

// ---- fractional-sample prediction of 4 adjacent samples as RAW sums, shared by uni- and bi-prediction.
// kind 0: v = sample; kind 1: one fraction, v = tap sum (scale 64); kind 2: both fractions, v = vertical taps over the
// (int16) horizontal tap sums (scale 4096).  Horizontal taps via v_dot4_i32_i8 on (p - 128): + 8192 restores the bias.
//   uni-prediction (interp*8to8 / 16to8):      0: v          1: clip8((v + 32) >> 6)   2: clip8((v + 2048) >> 12)
//   14-bit for bi  (interp*8to16 / 16to16):    0: v << 6     1: v                      2: v >> 6
__device__ __forceinline__ int pack_taps4(const signed char *c)
{
    return (int)((unsigned)(unsigned char)c[0] | ((unsigned)(unsigned char)c[1] << 8) | ((unsigned)(unsigned char)c[2] << 16) | ((unsigned)(unsigned char)c[3] << 24));
}
// chroma (interpChroma* enc@0x4111c0..): 4 taps, 1/8 sample; rp -> sample 0 at the integer position
__device__ __forceinline__ int chroma_raw4(const uint8_t *rp, long stride, int fx, int fy, int (&v)[4])
{
    auto hrow = [&](const uint8_t *row, int (&h)[4]) {            // bytes -1 .. 5
        const uint8_t *q = row - 1;
        const unsigned sh = (unsigned)((uintptr_t)q & 3);
        const unsigned *a = (const unsigned *)(q - sh);
        const unsigned a0 = a[0] ^ 0x80808080u, a1 = a[1] ^ 0x80808080u, a2 = a[2] ^ 0x80808080u;
        const unsigned w0 = align_bytes(a1, a0, sh), w1 = align_bytes(a2, a1, sh);   // bytes -1..2, 3..6
        const int taps = pack_taps4(kChromaTaps[fx]);
        h[0] = __builtin_amdgcn_sdot4((int)w0, taps, 8192, false);
        h[1] = __builtin_amdgcn_sdot4((int)align_bytes(w1, w0, 1), taps, 8192, false);
        h[2] = __builtin_amdgcn_sdot4((int)align_bytes(w1, w0, 2), taps, 8192, false);
        h[3] = __builtin_amdgcn_sdot4((int)align_bytes(w1, w0, 3), taps, 8192, false);
    };
    if (!fy) {
        if (!fx) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rp[i];
            return 0;
        }
        hrow(rp, v);
        return 1;
    }
    v[0] = v[1] = v[2] = v[3] = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cy = kChromaTaps[fy][r];
        if (!fx) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += cy * (int)rp[(r - 1) * stride + i];
        } else {
            int h[4]; hrow(rp + (r - 1) * stride, h);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += cy * (int)(short)h[i];
        }
    }
    return fx ? 2 : 1;
}
// luma (interpLuma* enc@0x40e4f0..): 8 taps, 1/4 sample
__device__ __forceinline__ int luma_raw4(const uint8_t *rp, long stride, int fx, int fy, int (&v)[4])
{
    auto hrow = [&](const uint8_t *row, int (&h)[4]) {            // bytes -3 .. 7
        const uint8_t *q = row - 3;
        const unsigned sh = (unsigned)((uintptr_t)q & 3);
        const unsigned *a = (const unsigned *)(q - sh);
        const unsigned a0 = a[0] ^ 0x80808080u, a1 = a[1] ^ 0x80808080u, a2 = a[2] ^ 0x80808080u, a3 = a[3] ^ 0x80808080u;
        const unsigned w0 = align_bytes(a1, a0, sh), w1 = align_bytes(a2, a1, sh), w2 = align_bytes(a3, a2, sh);   // bytes -3..0, 1..4, 5..8
        const int tl = pack_taps4(kLumaTaps[fx]), th = pack_taps4(kLumaTaps[fx] + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned lo = i ? align_bytes(w1, w0, i) : w0, hi = i ? align_bytes(w2, w1, i) : w1;
            h[i] = __builtin_amdgcn_sdot4((int)hi, th, __builtin_amdgcn_sdot4((int)lo, tl, 8192, false), false);
        }
    };
    if (!fy) {
        if (!fx) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rp[i];
            return 0;
        }
        hrow(rp, v);
        return 1;
    }
    v[0] = v[1] = v[2] = v[3] = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int cy = kLumaTaps[fy][r];
        if (!fx) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += cy * (int)rp[(r - 3) * stride + i];
        } else {
            int h[4]; hrow(rp + (r - 3) * stride, h);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += cy * (int)(short)h[i];
        }
    }
    return fx ? 2 : 1;
}
__device__ __forceinline__ int uni_round(int kind, int v) { return kind == 0 ? v : kind == 1 ? clip8((v + 32) >> 6) : clip8((v + 2048) >> 12); }
__device__ __forceinline__ int to14(int kind, int v) { return kind == 0 ? v << 6 : kind == 1 ? v : v >> 6; }

}  // namespace ks265
