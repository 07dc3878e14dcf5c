// recon_dev.h — LDS transform plumbing shared by the inter (frame_recon.hip) and intra (frame_intra.hip) reconstruction kernels:
// matrices of all four sizes, row-times-row dot products with v_dot2c_i32_i16 (see the header comment of frame_recon.hip).
#pragma once
#include "frame_common.h"

namespace ks265 {

typedef short s16x2 __attribute__((ext_vector_type(2)));
#define RP 36                      // LDS row pitch of the sample/coefficient tiles, in shorts

// matrices of all four sizes, row pitch n + 4 shorts; offset of size n (4, 8, 16, 32)
__device__ __forceinline__ int mat_off(int log2n) { return log2n == 2 ? 0 : log2n == 3 ? 32 : log2n == 4 ? 128 : 448; }   // 4*8, 8*12, 16*20, 32*36
#define MAT_SHORTS (448 + 32 * 36)

__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false); }

// out[i] = shared[0..n) . rows[i * pitch + 0..n), i = 0..3 ; all pointers 8-byte aligned, n a multiple of 4
__device__ __forceinline__ void quad_dot(const short *shared, const short *rows, int pitch, int n, int (&out)[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int x = 0; x < n; x += 4) {
        const uint2 s = *(const uint2 *)(shared + x);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint2 r = *(const uint2 *)(rows + i * pitch + x);
            out[i] = dot2(s.x, r.x, out[i]);
            out[i] = dot2(s.y, r.y, out[i]);
        }
    }
}

// build the forward (Mf) and transposed (Mt) DCT matrices of all four sizes in LDS (nthreads threads cooperate)
__device__ __forceinline__ void build_matrices(short *Mf, short *Mt, int tid, int nthreads)
{
    for (int l2 = 2; l2 <= 5; ++l2) {
        const int n = 1 << l2, base = mat_off(l2), mp = n + 4;
        for (int i = tid; i < n * n; i += nthreads) {
            const int k = i / n, x = i % n, v = dct_coef(n, k, x);
            Mf[base + k * mp + x] = (short)v;
            Mt[base + x * mp + k] = (short)v;
        }
    }
}

}  // namespace ks265
