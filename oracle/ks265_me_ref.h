/* ks265_me_ref.h — TEST INFRASTRUCTURE (oracle): the reference's integer-pel search control, see ks265_me_ref.c */
#ifndef KS265_ME_REF_H
#define KS265_ME_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t (*kso_dist_fn)(const uint8_t *a, const uint8_t *b, long sa, long sb, long h, long w);   /* sad_c enc@0x47ae30 / had_c enc@0x47b680 */

typedef struct {
    const uint8_t *fenc; int fstride;                 /* tME+0x30 / +0x38                                                        */
    const uint8_t *ref0; int stride;                  /* tME+0x8 + PU position: the reference sample under mv (0,0); tME+0x50    */
    int log2w, log2h;                                 /* TPredUnit+5 / +6                                                        */
    const uint16_t *cmx, *cmy;                        /* tME+0x18 / +0x20, indexed by the quarter-pel mv (integer mv << 2)       */
    int merange;                                      /* tME+0x68                                                                */
    int range_shift;                                  /* TPredUnit+0x1f1 (interMeDia only)                                       */
    int mv_min_x, mv_max_x, mv_min_y, mv_max_y;       /* tME+0x6c / +0x6e / +0x70 / +0x72 (integer pel)                          */
    int skip_cross;                                   /* TPredUnit+3 (interMeUMH only)                                           */
    kso_dist_fn dist;                                 /* TPredUnit+0x38                                                          */
    int mx, my; uint32_t cost; int converged;         /* tME+0x54 / +0x56, +0x90, +0x3b8: in = start point and its cost, out = result */
    /* replay guard (not a reference field): when chk != 0 a block read at an mv outside [gx0, gx1] x [gy0, gy1] is not performed and oob is set */
    int chk, gx0, gx1, gy0, gy1, oob;
} kso_me;

/* createMvdCostTable enc@0x48b850: the byte table of code lengths the reference builds once (loop at enc@0x48b926..0x48b97d): for a quarter-pel
 * difference d, v = 2 d (d > 0) or 1 - 2 d (d <= 0), bits = 1 + 2 floor(log2 v) - the signed exp-Golomb length; the u16 cost tables are
 * lambda(qp) x bits for |d| <= 4 merange + 16 (enc@0x48b9e0..0x48ba01: imul, 16-bit store), and tME+0x18 / +0x20 point into them at -mvp, so that
 * p_cost_mvx[mv] = lambda x bits(mv - mvp).  Pinned on the table slices recorded inside the reference (tests/test_me_search.py). */
int kso_mvd_bits(int d);
/* out[i] = (uint16)(lambda x bits(4 (lo + i) - mvp_q)), i = 0 .. hi - lo: the slice at integer positions lo..hi the trace shim records */
void kso_mvd_cost_slice(int lambda, int mvp_q, int lo, int hi, uint16_t *out);

void kso_ref_me_dia(kso_me *m);
void kso_ref_me_hex(kso_me *m);
void kso_ref_me_umh(kso_me *m);

/* trace replay (tests/test_me_search.py): one recorded call of the reference.  plane = the recorded region of the reference plane,
 * (rx0, ry0) its first sample relative to the plane origin, rw x rh its size; cmx / cmy = the recorded table slices, cmx[0] = entry of
 * integer mv xlo.  Returns 0, or -1 if the replay would read outside the recorded region (out[] untouched). */
int kso_me_replay(int method, const uint8_t *fenc, int log2w, int log2h, const uint8_t *plane, int rx0, int ry0, int rw, int rh,
                  int pux, int puy, const uint16_t *cmx, int xlo, int xhi, const uint16_t *cmy, int ylo, int yhi,
                  int merange, int range_shift, const int lim[4], int skip_cross, int use_had, int sx, int sy, uint32_t cost0, int32_t out[4]);
#ifdef __cplusplus
}
#endif
#endif
