// intra_dev.h — intra prediction of one sample from a linear reference array (shared by the batched operator and the
// frame stages).  Restates g_IntraPredFunction enc@0x7070a0 (IntraPredPlanar_0_c enc@0x425af0, IntraPredDC_1_c enc@0x425d80,
// IntraPredAng*_c enc@0x425f60..0x426ce0) and IntraPredFilterRef_c enc@0x424110; pinned through tests/golden/intra.npz.
// Reference layout: r points at the corner p[-1][-1]; r[1 + x] = top / top-right, r[-1 - y] = left / bottom-left.
#pragma once
#include "ks265_dev.h"

namespace ks265 {

__device__ __forceinline__ int intra_angle(int mode)
{
    // 32 26 21 17 13 9 5 2 0 for modes 2..10 and mirrored; negative for 11..25
    const int d = mode <= 18 ? (mode <= 10 ? 10 - mode : mode - 10) : (mode <= 26 ? 26 - mode : mode - 26);   // distance to pure hor / ver: 0..8
    const int a = (int)((0x2069544D245080ull >> (6 * d)) & 63ull);         // 0 2 5 9 13 17 21 26 32, six bits each
    const bool neg = (mode > 10 && mode < 26);
    return neg ? -a : a;
}
__device__ __forceinline__ int intra_inv_angle(int mode)
{
    // modes 11..25: 256 * 32 / |angle|, negative
    const int d = mode <= 18 ? mode - 10 : 26 - mode;              // 1..8
    const unsigned long long w = d <= 4 ? 0x276038E06661000ull : 0x100013B018601E2ull;   // 4096 1638 910 630 | 482 390 315 256
    return -(int)((w >> (16 * ((d - 1) & 3))) & 0xFFFFull);
}

// luma reference smoothing decision of H.265 8.4.4.2.3 (filterFlag), n = 8, 16, 32 (never for 4 and never for DC)
__device__ __forceinline__ bool intra_filter_flag(int mode, int n)
{
    if (mode == 1 || n <= 4) return false;
    const int dist = min(abs(mode - 26), abs(mode - 10));
    const int thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
    return dist > thr;
}

// one predicted sample.  dc = the DC value (only read for mode 1).  edge = boundary smoothing of DC / 10 / 26 (luma, n < 32).
__device__ __forceinline__ int intra_sample(const uint8_t *r, int mode, int log2, int x, int y, int dc, bool edge)
{
    const int n = 1 << log2;
    if (mode == 0)
        return ((n - 1 - x) * r[-1 - y] + (x + 1) * r[1 + n] + (n - 1 - y) * r[1 + x] + (y + 1) * r[-1 - n] + n) >> (log2 + 1);
    if (mode == 1) {
        if (edge && n < 32) {
            if (x == 0 && y == 0) return (r[-1] + 2 * dc + r[1] + 2) >> 2;
            if (y == 0) return (r[1 + x] + 3 * dc + 2) >> 2;
            if (x == 0) return (r[-1 - y] + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    const bool ver = mode >= 18;
    const int ang = intra_angle(mode);
    const int i = ver ? x : y, j = ver ? y : x;                    // j walks away from the main reference side
    if (ang == 0) {
        if (edge && n < 32 && i == 0) return clip8((ver ? r[1] : r[-1]) + (((ver ? r[-1 - j] : r[1 + j]) - r[0]) >> 1));
        return ver ? r[1 + i] : r[-1 - i];
    }
    const int idx = ((j + 1) * ang) >> 5, fact = ((j + 1) * ang) & 31;
    const int inv = ang < 0 ? intra_inv_angle(mode) : 0;
    auto at = [&](int k) -> int {                                  // main reference, extended to negative k by projecting the side
        if (k >= 0) return ver ? r[k] : r[-k];
        const int s = (k * inv + 128) >> 8;
        return ver ? r[-s] : r[s];
    };
    const int a = at(i + idx + 1);
    if (!fact) return a;
    return ((32 - fact) * a + fact * at(i + idx + 2) + 16) >> 5;
}

// Four adjacent samples of one row (x0 .. x0 + 3, y): vertical angular modes share the row's index / fraction and five reference
// reads; everything else goes through intra_sample.
__device__ __forceinline__ void intra_quad(const uint8_t *r, int mode, int log2, int x0, int y, int dc, bool edge, int (&o)[4])
{
    const int ang = (mode >= 2 && mode != 10 && mode != 26) ? intra_angle(mode) : 0;
    if (ang == 0) {                                                // planar, DC, pure horizontal / vertical: the generic path
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = intra_sample(r, mode, log2, x0 + c, y, dc, edge);
        return;
    }
    const int inv = ang < 0 ? intra_inv_angle(mode) : 0;
    if (mode < 18) {
        // horizontal angular modes (round 6: through intra_sample they were 931 instructions per quad - half of the prediction phase of the intra chain): index / fraction
        // depend on the column, the main reference is the left column (r[-k]), extended upwards by projecting the top row (r[(k inv + 128) >> 8])
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = (x0 + c + 1) * ang, idx = t >> 5, f = t & 31, k = y + idx + 1;
            const int a = k >= 0 ? r[-k] : r[(k * inv + 128) >> 8], b = k + 1 >= 0 ? r[-(k + 1)] : r[((k + 1) * inv + 128) >> 8];
            o[c] = f ? ((32 - f) * a + f * b + 16) >> 5 : a;
        }
        return;
    }
    const int t = (y + 1) * ang, idx = t >> 5, f = t & 31;
    int v[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int k = x0 + idx + 1 + c;
        v[c] = k >= 0 ? r[k] : r[-((k * inv + 128) >> 8)];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = f ? ((32 - f) * v[c] + f * v[c + 1] + 16) >> 5 : v[c];
}

// Two rows x eight columns of one block at once (rows y0, y0 + 1, columns x0 .. x0 + 7) as four packed dwords (row 0 low / high,
// row 1 low / high) - the unit an MFMA operand column of the SATD needs.  Same arithmetic as intra_sample; for the angular modes
// the per-row (vertical) or per-column (horizontal) index / fraction is computed once and neighbouring samples share their
// reference reads, which cuts the instruction count per sample about four-fold against sixteen independent intra_sample calls.
__device__ __forceinline__ void intra_rows2x8(const uint8_t *r, int mode, int log2, int x0, int y0, int dc, bool edge, unsigned (&w)[4])
{
    const int ang = mode >= 2 ? intra_angle(mode) : 0;
    if (mode < 2 || ang == 0) {                                    // planar, DC, pure horizontal / vertical: the generic path
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned v = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) v |= (unsigned)intra_sample(r, mode, log2, x0 + 4 * (q & 1) + c, y0 + (q >> 1), dc, edge) << (8 * c);
            w[q] = v;
        }
        return;
    }
    const bool ver = mode >= 18;
    const int inv = ang < 0 ? intra_inv_angle(mode) : 0;
    auto at = [&](int k) -> int {                                  // main reference, extended to negative k by projecting the side
        if (k >= 0) return ver ? r[k] : r[-k];
        const int s = (k * inv + 128) >> 8;
        return ver ? r[-s] : r[s];
    };
    if (ver) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int t = (y0 + rr + 1) * ang, idx = t >> 5, f = t & 31;
            int v[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) v[c] = at(x0 + idx + 1 + c);
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                lo |= (unsigned)(f ? ((32 - f) * v[c] + f * v[c + 1] + 16) >> 5 : v[c]) << (8 * c);
                hi |= (unsigned)(f ? ((32 - f) * v[c + 4] + f * v[c + 5] + 16) >> 5 : v[c + 4]) << (8 * c);
            }
            w[2 * rr] = lo; w[2 * rr + 1] = hi;
        }
    } else {
        unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; ++c) {                              // column x0 + c: index / fraction depend on x only
            const int t = (x0 + c + 1) * ang, idx = t >> 5, f = t & 31;
            const int a = at(y0 + idx + 1), b = at(y0 + idx + 2), d = at(y0 + idx + 3);
            const int p0 = f ? ((32 - f) * a + f * b + 16) >> 5 : a, p1 = f ? ((32 - f) * b + f * d + 16) >> 5 : b;
            o[c >> 2] |= (unsigned)p0 << (8 * (c & 3));
            o[2 + (c >> 2)] |= (unsigned)p1 << (8 * (c & 3));
        }
        w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
    }
}

// IntraPredFilterRef_c enc@0x424110: sample k (-2 size .. 2 size) of the smoothed array; `bilinear` = the size-32 strong filter
// applies (decided once per array by intra_strong_flat)
__device__ __forceinline__ bool intra_strong_flat(const uint8_t *s)
{
    const int c = s[0];
    return abs(c + s[-64] - 2 * s[-32]) < 8 && abs(c + s[64] - 2 * s[32]) < 8;
}
__device__ __forceinline__ int intra_filtered(const uint8_t *s, int size, int k, bool bilinear)
{
    const int n2 = 2 * size;
    if (k == -n2 || k == n2) return s[k];
    if (bilinear) {
        if (k == 0) return s[0];
        const int a = abs(k);
        return ((n2 - a) * s[0] + a * s[k < 0 ? -n2 : n2] + 32) >> 6;
    }
    return (s[k - 1] + 2 * s[k] + s[k + 1] + 2) >> 2;
}

}  // namespace ks265
