// ks265_dev.h — device-side building blocks shared by the batched operator kernels and the whole-frame
// stage kernels (gfx950 / CDNA4, wave64).  Every arithmetic routine cites the reference kernel it is
// bit-exact with (enc@0xADDR = /root/reference/ubuntu_x64/appencoder).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ks265 {

// ------------------------------------------------------------------ scalar helpers
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
__device__ __forceinline__ int clip8(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }

// ------------------------------------------------------------------ cross-lane (wave64) primitives
// DPP control words (gfx9): quad_perm = perm bits, row_shr:n = 0x110+n, row_mirror = 0x140, row_half_mirror = 0x141
#define KS265_DPP_QUAD_XOR1 0xB1  // quad_perm [1,0,3,2]
#define KS265_DPP_QUAD_XOR2 0x4E  // quad_perm [2,3,0,1]
#define KS265_DPP_ROW_HALF_MIRROR 0x141
#define KS265_DPP_ROW_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}

// value held by lane (lane ^ MASK); DPP for 1,2; ds_swizzle (bit mode, within 32 lanes) for 4,8,16
template <int MASK>
__device__ __forceinline__ int lane_xor(int v)
{
    if constexpr (MASK == 1) return dpp_mov<KS265_DPP_QUAD_XOR1>(v);
    else if constexpr (MASK == 2) return dpp_mov<KS265_DPP_QUAD_XOR2>(v);
    else if constexpr (MASK < 32) return __builtin_amdgcn_ds_swizzle(v, 0x1F | (MASK << 10));
    else return __shfl_xor(v, 32, 64);
}

// sum over aligned groups of GROUP lanes (8, 16, 32, 64); every lane of the group gets the total.
// 1,2: quad permutes; 4: row_half_mirror; 8: row_mirror (valid because the partial sums are already
// uniform inside each quad / octet); 16: ds_swizzle; 32: bpermute.
template <int GROUP>
__device__ __forceinline__ unsigned group_sum(unsigned v)
{
    v += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR1>((int)v);
    v += (unsigned)dpp_mov<KS265_DPP_QUAD_XOR2>((int)v);
    if constexpr (GROUP >= 8) v += (unsigned)dpp_mov<KS265_DPP_ROW_HALF_MIRROR>((int)v);
    if constexpr (GROUP >= 16) v += (unsigned)dpp_mov<KS265_DPP_ROW_MIRROR>((int)v);
    if constexpr (GROUP >= 32) v += (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (16 << 10));
    if constexpr (GROUP >= 64) v += (unsigned)__shfl_xor((int)v, 32, 64);
    return v;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) { return group_sum<64>(v); }

// ------------------------------------------------------------------ byte SAD on packed dwords
// v_sad_u8: sum of 4 absolute byte differences + accumulator
__device__ __forceinline__ unsigned sad_u8x4(unsigned a, unsigned b, unsigned acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
// dword starting `sh` bytes into the 8-byte pair {hi:lo}
__device__ __forceinline__ unsigned align_bytes(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }

// ------------------------------------------------------------------ transform matrices
// HEVC core-transform magnitudes |cos(j*pi/64)| (g_uiTr32 enc@0x4e06a0 is the expanded 32x32 matrix)
__device__ __constant__ const signed char kMag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                                      61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
// DST4x4_COEFF enc@0x4e0aa0
__device__ __constant__ const signed char kDst4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};

// coefficient M[k][x] of the N-point DCT (N = 4..32)
__device__ __forceinline__ int dct_coef(int n, int k, int x)
{
    int m = (k * (32 / n) * (2 * x + 1)) & 127;
    if (m > 64) m = 128 - m;
    return m > 32 ? -(int)kMag[64 - m] : (int)kMag[m];
}

// fill an LDS matrix (row-major NxN ints stored as short) for table index idx (0 = DST4, 1..4 = DCT4..32)
__device__ __forceinline__ void load_matrix(short *M, int idx, int n, int tid, int nthreads)
{
    for (int i = tid; i < n * n; i += nthreads) M[i] = (short)(idx == 0 ? (int)kDst4[i] : dct_coef(n, i / n, i % n));
}

// ------------------------------------------------------------------ quant tables (g_quantScales / g_invQuantScales)
__device__ __constant__ const int kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};
__device__ __constant__ const int kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};

// H265QuantBlock_c enc@0x4a9cf0, one coefficient
__device__ __forceinline__ int quant_one(int c, int scale, int off, int qbits, int &deltaU)
{
    int a = abs(c) * scale;
    int l = (a + off) >> qbits;
    deltaU = (a - (l << qbits)) >> (qbits - 8);
    l = min(l, 32767);
    return c < 0 ? -l : l;
}
// H265DeQuantBlock_c enc@0x439210, one level
__device__ __forceinline__ int dequant_one(int l, int scale, int add, int shift) { return clip16((l * scale + add) >> shift); }

// Work-group barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt(0)) and joins the barrier, but
// leaves global stores in flight.  __syncthreads() also drains vmcnt, i.e. every barrier behind a store to HBM stalls for the
// full store latency (measured: ~3 us per CU in the intra wavefront).  Only for code whose later phases never read back, through
// global memory, what an earlier phase of the same work-group stored.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);          // vmcnt = 63 (no wait), expcnt = 7 (no wait), lgkmcnt = 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------ deblocking (normative, EdgeFilterLuma*_c enc@0x403630/0x4038c0)
__device__ __constant__ const unsigned char kTcTable[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                                            2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__device__ __constant__ const unsigned char kBetaTable[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                                              16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54,
                                                              56, 58, 60, 62, 64};

// One 4-line luma segment held in registers: px[l][0..7] = p3 p2 p1 p0 q0 q1 q2 q3 of line l.
// Filters in place; returns a bit mask of modified columns is not needed — callers store columns 1..6.
__device__ __forceinline__ void deblock_luma_segment(int px[4][8], int beta, int tc, bool filterP, bool filterQ)
{
    auto d2 = [&](int l, int a, int b, int c) { return abs(px[l][a] - 2 * px[l][b] + px[l][c]); };
    int dp0 = d2(0, 1, 2, 3), dp3 = d2(3, 1, 2, 3), dq0 = d2(0, 6, 5, 4), dq3 = d2(3, 6, 5, 4);
    int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
    if (d >= beta) return;
    bool s0 = (2 * dpq0 < (beta >> 2)) && (abs(px[0][0] - px[0][3]) + abs(px[0][4] - px[0][7]) < (beta >> 3)) &&
              (abs(px[0][3] - px[0][4]) < ((5 * tc + 1) >> 1));
    bool s3 = (2 * dpq3 < (beta >> 2)) && (abs(px[3][0] - px[3][3]) + abs(px[3][4] - px[3][7]) < (beta >> 3)) &&
              (abs(px[3][3] - px[3][4]) < ((5 * tc + 1) >> 1));
    int side = (beta + (beta >> 1)) >> 3;
    bool dEp = dp < side, dEq = dq < side;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        int p3 = px[l][0], p2 = px[l][1], p1 = px[l][2], p0 = px[l][3], q0 = px[l][4], q1 = px[l][5], q2 = px[l][6], q3 = px[l][7];
        if (s0 && s3) {
            if (filterP) {
                px[l][3] = clip8(clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
                px[l][2] = clip8(clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2));
                px[l][1] = clip8(clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
            }
            if (filterQ) {
                px[l][4] = clip8(clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
                px[l][5] = clip8(clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2));
                px[l][6] = clip8(clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
            }
        } else {
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta) < 10 * tc) {
                delta = clip3(-tc, tc, delta);
                if (filterP) {
                    px[l][3] = clip8(p0 + delta);
                    if (dEp) px[l][2] = clip8(p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
                }
                if (filterQ) {
                    px[l][4] = clip8(q0 - delta);
                    if (dEq) px[l][5] = clip8(q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
                }
            }
        }
    }
}

// PixelFilterChroma{Ver,Hor}_c enc@0x403c50/0x403d10, one line: p1 p0 q0 q1 -> new p0,q0
__device__ __forceinline__ void deblock_chroma_line(int p1, int &p0, int &q0, int q1, int tc, bool filterP, bool filterQ)
{
    int delta = clip3(-tc, tc, (((q0 - p0) << 2) + p1 - q1 + 4) >> 3);
    int np = clip8(p0 + delta), nq = clip8(q0 - delta);
    if (filterP) p0 = np;
    if (filterQ) q0 = nq;
}

// ------------------------------------------------------------------ interpolation taps (interpLuma*/interpChroma* enc@0x40e4f0..)
__device__ __constant__ const signed char kLumaTaps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0},
                                                             {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__device__ __constant__ const signed char kChromaTaps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                                               {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

// chroma QP mapping for 4:2:0 (HEVC table 8-10), cQpOffset = 0
__device__ __forceinline__ int chroma_qp(int qp)
{
    const int tab[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
    return qp < 30 ? qp : (qp >= 44 ? qp - 6 : tab[qp - 30]);
}

// ------------------------------------------------------------------ motion-vector rate (createMvdCostTable enc@0x48b850 analogue)
// signed exp-Golomb length of a quarter-pel mvd component; cost = (lambda_q4 * bits) >> 4
__device__ __forceinline__ int se_bits(int v)
{
    unsigned u = (unsigned)(v <= 0 ? -2 * v : 2 * v - 1) + 1u;   // ue code number + 1
    return 2 * (31 - __clz((int)u)) + 1;
}
__device__ __forceinline__ int mv_cost(int mvx, int mvy, int px, int py, int lambda_q4)
{
    return (lambda_q4 * (se_bits(mvx - px) + se_bits(mvy - py))) >> 4;
}

// ------------------------------------------------------------------ Hadamard (had_c enc@0x47b680, xCalcHADs8x8 enc@0x47b3b0)
// 64 lanes hold the 8x8 difference tile (lane = y*8 + x); returns sum |H8 D H8^T| in every lane
__device__ __forceinline__ unsigned had8x8_abs_sum(int v, int lane)
{
    int p;
    p = lane_xor<1>(v);  v = (lane & 1) ? p - v : v + p;
    p = lane_xor<2>(v);  v = (lane & 2) ? p - v : v + p;
    p = lane_xor<4>(v);  v = (lane & 4) ? p - v : v + p;
    p = lane_xor<8>(v);  v = (lane & 8) ? p - v : v + p;
    p = lane_xor<16>(v); v = (lane & 16) ? p - v : v + p;
    p = lane_xor<32>(v); v = (lane & 32) ? p - v : v + p;
    return wave_sum((unsigned)abs(v));
}
// 16-lane groups each hold a 4x4 difference tile (lane&15 = y*4 + x); returns the group's sum |H4 D H4^T|
__device__ __forceinline__ unsigned had4x4_abs_sum16(int v, int lane)
{
    int p;
    p = lane_xor<1>(v); v = (lane & 1) ? p - v : v + p;
    p = lane_xor<2>(v); v = (lane & 2) ? p - v : v + p;
    p = lane_xor<4>(v); v = (lane & 4) ? p - v : v + p;
    p = lane_xor<8>(v); v = (lane & 8) ? p - v : v + p;
    return group_sum<16>((unsigned)abs(v));
}

// ------------------------------------------------------------------ core transforms in LDS
// Forward (H265_2dDct*_c enc@0x4c2210..): X (NxN, row-major s16) -> X. Stage shifts (2*log2N-2, 7).
// Caller: M loaded, X loaded, __syncthreads() done. T is scratch. Ends without a trailing barrier.
template <int N>
__device__ __forceinline__ void fwd_transform_lds(const short *M, short *X, short *T, int tid, int nthreads)
{
    constexpr int LOG2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    constexpr int S1 = 2 * LOG2N - 2, R1 = 1 << (S1 - 1);
    for (int t = tid; t < N * N; t += nthreads) {
        int k = t / N, j = t % N, acc = 0;
#pragma unroll 8
        for (int x = 0; x < N; ++x) acc += (int)M[k * N + x] * (int)X[j * N + x];
        T[k * N + j] = (short)((acc + R1) >> S1);
    }
    __syncthreads();
    for (int t = tid; t < N * N; t += nthreads) {
        int k = t / N, j = t % N, acc = 0;
#pragma unroll 8
        for (int x = 0; x < N; ++x) acc += (int)M[k * N + x] * (int)T[j * N + x];
        X[k * N + j] = (short)((acc + 64) >> 7);
    }
}
// Inverse (H265_2dIDct*_c enc@0x448f60..): X = coefficients -> X = residual (before pred add).
template <int N>
__device__ __forceinline__ void inv_transform_lds(const short *M, short *X, short *T, int tid, int nthreads)
{
    for (int t = tid; t < N * N; t += nthreads) {
        int y = t / N, x = t % N, acc = 0;
#pragma unroll 8
        for (int k = 0; k < N; ++k) acc += (int)M[k * N + y] * (int)X[k * N + x];
        T[y * N + x] = (short)clip16((acc + 64) >> 7);
    }
    __syncthreads();
    for (int t = tid; t < N * N; t += nthreads) {
        int y = t / N, x = t % N, acc = 0;
#pragma unroll 8
        for (int k = 0; k < N; ++k) acc += (int)T[y * N + k] * (int)M[k * N + x];
        X[y * N + x] = (short)((acc + 2048) >> 12);
    }
}

}  // namespace ks265
