/* ks265_intra_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's intra prediction kernels, i.e. the functions behind
 * g_IntraPredFunction enc@0x7070a0 and g_IntraPredFilterRefFunc enc@0x706d48:
 *   IntraPredPlanar_0_c enc@0x425af0, IntraPredDC_1_c enc@0x425d80, IntraPredChromeDC_1_c enc@0x425c60,
 *   IntraPredAngHorPlus_2_c enc@0x425f60, IntraPredAngHorPlus_3_9_c enc@0x4260e0, IntraPredAngHor0_10_c enc@0x426300,
 *   IntraPredAngHorMinus_11_17_c enc@0x4264c0, IntraPredAngVerMinus_18_c enc@0x426720,
 *   IntraPredAngVerMinus_19_25_c enc@0x4267e0, IntraPredAngVer0_26_c enc@0x4269d0,
 *   IntraPredAngVerPlus_27_33_c enc@0x426bb0, IntraPredAngVerPlus_34_c enc@0x426ce0,
 *   IntraPredFilterRef_c enc@0x424110.
 * All share the signature (u8 *dst, int dstStride, u8 *ref, int mode, int log2Size, bool edgeFilter); `ref` points at the
 * top-left corner sample p[-1][-1] of a linear reference array: ref[1 + x] = p[x][-1] (top, then top-right, x < 2N),
 * ref[-1 - y] = p[-1][y] (left, then bottom-left, y < 2N).  The arithmetic is the normative HEVC process (H.265 8.4.4.2.4-6),
 * restated here from the standard and pinned against the reference binary by tests/golden/intra.npz
 * (oracle/ref_probe/gen_golden.py: gen_intra).
 */
#include "ks265_oracle.h"

static inline uint8_t iclip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static const int8_t kIntraAngle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                       -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
static const int16_t kInvAngle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};   /* modes 11..25 */

/* mode 0 = planar, 1 = DC, 2..34 angular.  edge_filter enables the DC / horizontal (10) / vertical (26) boundary smoothing
 * (the reference passes it for luma blocks; it is ignored for N = 32 as in the standard). */
void ks265o_intra_pred(uint8_t *dst, int stride, const uint8_t *ref, int mode, int log2, int edge_filter)
{
    const int n = 1 << log2;
    if (mode == 0) {
        const int tr = ref[1 + n], bl = ref[-1 - n];
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x)
                dst[y * stride + x] = (uint8_t)(((n - 1 - x) * ref[-1 - y] + (x + 1) * tr + (n - 1 - y) * ref[1 + x] + (y + 1) * bl + n) >> (log2 + 1));
        return;
    }
    if (mode == 1) {
        int sum = n;
        for (int i = 0; i < n; ++i) sum += ref[1 + i] + ref[-1 - i];
        const int dc = sum >> (log2 + 1);
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) dst[y * stride + x] = (uint8_t)dc;
        if (edge_filter && n < 32) {
            dst[0] = (uint8_t)((ref[-1] + 2 * dc + ref[1] + 2) >> 2);
            for (int x = 1; x < n; ++x) dst[x] = (uint8_t)((ref[1 + x] + 3 * dc + 2) >> 2);
            for (int y = 1; y < n; ++y) dst[y * stride] = (uint8_t)((ref[-1 - y] + 3 * dc + 2) >> 2);
        }
        return;
    }
    const int ver = mode >= 18, ang = kIntraAngle[mode];
    /* main[] = the side the mode projects from (top for vertical modes, left for horizontal), index 0 = corner; side[] = the other */
    uint8_t buf[3 * 64 + 1];
    uint8_t *m = buf + 64;
    for (int i = 0; i <= 2 * n; ++i) m[i] = ver ? ref[i] : ref[-i];
    if (ang < 0) {
        const int last = (n * ang) >> 5, inv = kInvAngle[mode - 11];
        for (int i = -1; i >= last; --i) {
            const int k = (i * inv + 128) >> 8;                 /* 1-based position on the side array */
            m[i] = ver ? ref[-k] : ref[k];
        }
    }
    for (int j = 0; j < n; ++j) {                              /* j walks away from the main side */
        const int idx = ((j + 1) * ang) >> 5, fact = ((j + 1) * ang) & 31;
        for (int i = 0; i < n; ++i) {
            const int v = fact ? ((32 - fact) * m[i + idx + 1] + fact * m[i + idx + 2] + 16) >> 5 : m[i + idx + 1];
            if (ver) dst[j * stride + i] = (uint8_t)v;
            else dst[i * stride + j] = (uint8_t)v;
        }
    }
    if (ang == 0 && edge_filter && n < 32) {
        for (int k = 0; k < n; ++k) {
            if (ver) dst[k * stride] = iclip8(ref[1] + ((ref[-1 - k] - ref[0]) >> 1));          /* mode 26: first column */
            else dst[k] = iclip8(ref[-1] + ((ref[1 + k] - ref[0]) >> 1));                      /* mode 10: first row */
        }
    }
}

/* IntraPredFilterRef_c enc@0x424110 (src, dst, size, strongEnabled): the [1 2 1] / 4 smoothing of the 4N + 1 reference samples
 * (ends copied).  For size 32 with strongEnabled the function itself tests the flatness condition of H.265 8.4.4.2.3
 * (|corner + end - 2 * middle| < 8 on both sides) and, if it holds, writes the bi-linear replacement instead.
 * src / dst point at the corner sample. */
static inline int iabs_(int v) { return v < 0 ? -v : v; }
void ks265o_intra_filter_ref(const uint8_t *src, uint8_t *dst, int size, int strong_enabled)
{
    const int n2 = 2 * size;
    if (size == 32 && strong_enabled) {
        const int c = src[0], l = src[-n2], t = src[n2];
        if (iabs_(c + l - 2 * src[-size]) < 8 && iabs_(c + t - 2 * src[size]) < 8) {
            dst[0] = (uint8_t)c; dst[-n2] = (uint8_t)l; dst[n2] = (uint8_t)t;
            for (int i = 1; i < n2; ++i) {
                dst[-i] = (uint8_t)(((n2 - i) * c + i * l + 32) >> 6);
                dst[i] = (uint8_t)(((n2 - i) * c + i * t + 32) >> 6);
            }
            return;
        }
    }
    dst[-n2] = src[-n2]; dst[n2] = src[n2];
    for (int i = -n2 + 1; i < n2; ++i) dst[i] = (uint8_t)((src[i - 1] + 2 * src[i] + src[i + 1] + 2) >> 2);
}

/* ------------------------------------------------------------------ lookahead leaf kernels, pinned by tests/golden/lookahead.npz
 * downsample_c enc@0x4a6a60 (dst, src, dstStride, srcStride, w, h): 2:1 in both directions, w x h OUTPUT samples:
 *   out = (((s[2x] + s[2x + S] + 1) >> 1) + ((s[2x+1] + s[2x+1 + S] + 1) >> 1) + 1) >> 1   (vertical pairs first, then their mean) */
void ks265o_downsample(uint8_t *dst, const uint8_t *src, int dstStride, int srcStride, int w, int h)
{
    for (int y = 0; y < h; ++y, dst += dstStride, src += 2 * srcStride)
        for (int x = 0; x < w; ++x) {
            int a = (src[2 * x] + src[2 * x + srcStride] + 1) >> 1, b = (src[2 * x + 1] + src[2 * x + 1 + srcStride] + 1) >> 1;
            dst[x] = (uint8_t)((a + b + 1) >> 1);
        }
}
/* weightBi_sad_c enc@0x4a7170 (org, orgStride, ref0, ref1, stride0, stride1, w, h): SAD of org against the rounded average of two references */
uint32_t ks265o_weight_bi_sad(const uint8_t *org, unsigned orgStride, const uint8_t *ref0, const uint8_t *ref1, unsigned stride0, unsigned stride1, int w, int h)
{
    uint32_t s = 0;
    for (int y = 0; y < h; ++y, org += orgStride, ref0 += stride0, ref1 += stride1)
        for (int x = 0; x < w; ++x) {
            int p = (ref0[x] + ref1[x] + 1) >> 1, d = p - org[x];
            s += (uint32_t)(d < 0 ? -d : d);
        }
    return s;
}
/* acEnergyPlane_c enc@0x4650e0 (src, stride, log2Size): AC energy of an N x N block, ssd - (sum^2 >> 2 log2N), all in 32-bit
 * unsigned arithmetic as the reference computes it (sum^2 wraps for a bright 32x32 block; the wrap is part of the contract) */
uint32_t ks265o_ac_energy_plane(const uint8_t *src, int stride, int log2)
{
    const int n = 1 << log2;
    uint32_t sum = 0, ssd = 0;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) { uint32_t p = src[y * stride + x]; sum += p; ssd += p * p; }
    return ssd - ((sum * sum) >> (2 * log2));
}
