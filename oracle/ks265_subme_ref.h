/* ks265_subme_ref.h — TEST INFRASTRUCTURE (oracle): the reference's sub-pel refinement control, see ks265_subme_ref.c */
#ifndef KS265_SUBME_REF_H
#define KS265_SUBME_REF_H
#include <stdint.h>
#include "ks265_me_ref.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t (*kso_rate_fn)(void *ctx, int qx, int qy);     /* rate of a vector in quarter samples (tME+0x18[qx] + tME+0x20[qy] in the reference) */

/* the fields of tME / TPredUnit / the encoder configuration that subMeSquare enc@0x4b5660 and its callees read */
typedef struct {
    const uint8_t *fenc; int fstride;       /* tME+0x30 / +0x38                                                                             */
    const uint8_t *ref0; int stride;        /* tME+0x40 at entry: the reference sample under the integer start vector; tME+0x50             */
    int log2w, log2h;                       /* TPredUnit+5 / +6                                                                             */
    kso_dist_fn dist;                       /* TPredUnit+0x40: sad (satdInter 0: every preset up to slower) or had (veryslow, placebo)      */
    int recost;                             /* tME+0x64: the start cost is recomputed with `dist` first (set together with satdInter)       */
    int subme;                              /* tME+0x36c = QY265EncConfig.subme: 1 fast, 2 square full                                      */
    int mvres_thr;                          /* cfg+0x464: > 0 = the quarter step may be skipped (68 veryfast, 40 medium, 24 slow, 0 veryslow) */
    int hpel_diag_fast;                     /* cfg+0x580 (0 in every preset of v2.6.1.3)                                                    */
    int c568, flat_factor, flat_shift;      /* cfg+0x568, tME+0x3c4 (16 veryfast, 10 medium, 8 slow / veryslow), tME+0x60                    */
    uint32_t pu140;                         /* TPredUnit+0x140                                                                              */
    kso_rate_fn rate; void *rate_ctx;
    int do_subpel;                          /* tME+0x3bc, the verdict of getMvResolution                                                    */
    int mx, my; uint32_t cost;              /* tME+0x54 / +0x56 (quarter samples), tME+0x90: in = integer result, out = refined             */
    /* out */
    int hpel_idx, qpel_idx, qpel_ran;       /* winners of the two steps (-1 none), whether the quarter step ran                             */
    uint32_t rate_out, dist_out;            /* rate of the result (before the predictor-index term) and tME+0x98                            */
    int aliased;                            /* the call went through the path in which the reference's buffers overlap (see .c); result may differ */
} kso_subme;

void kso_ref_subme_square(kso_subme *s);

/* getMvResolution enc@0x483ca0 -> tME+0x3bc.  costs4 = tME+0x3a8.. (the four neighbour SADs << 4 of the integer search's last step), valid if `searched`
 * (tME+0x3b8); otherwise the function calls sad4 itself on fenc / ref0. */
int kso_ref_mv_resolution(int c498, int c49c, int log2w, int log2h, uint32_t cost, int me3c9, int me65, uint32_t rate0, int mvres_thr, int searched,
                          uint32_t costs4[4], int c8_21, int me3c0, int me60, const uint8_t *fenc, int fstride, const uint8_t *ref0, int stride);

/* trace replay (tests/test_subme.py): one recorded subMeSquare call.  hdr = the 64-word record header of oracle/ref_probe/subme_shim.c, fenc = W x H,
 * region = (W+16) x (H+16) around the start position, cm = 17 + 17 rate table entries around the start vector.  out[0..5] = mv x, y, cost (tME+0x90),
 * tME+0x94, tME+0x98, predictor index (tME+0x58); out[6] = aliased; out[7] = hpel_idx, out[8] = qpel_idx */
void kso_subme_replay(const int32_t *hdr, const uint8_t *fenc, const uint8_t *region, const uint16_t *cm, int32_t out[9]);
int kso_mvres_replay(const int32_t *hdr, const uint8_t *fenc, const uint8_t *region, int32_t out[2]);
#ifdef __cplusplus
}
#endif
#endif
