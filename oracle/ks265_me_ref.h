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

/* meInitPoint enc@0x48af50 (+ checkLayerMv enc@0x48ad80): the start point of the integer search.  In: the PU, its two AMVP candidates and the cost of coding
 * each index, the mv limits, merange, the mvd cost table and lambda, the look-ahead's vector for the block, the two extra vectors the PU carries.
 * Out: the predictor index chosen, the start mv and its cost / SAD, the search window, the table offsets. */
typedef struct {
    int log2w, log2h, pux, puy, stride;               /* TPredUnit+5 / +6 / +0xf8 / +0xfc, tME+0x50                                */
    int mvp[2][2];                                    /* TPredUnit+0x1a0 / +0x1a4 (quarter pel)                                    */
    int lim[4];                                       /* tME+0x74 / +0x76 / +0x78 / +0x7a: integer mv limits x0 x1 y0 y1           */
    int merange, lambda;                              /* tME+0x68, +0x80                                                           */
    uint32_t idx_cost[2];                             /* tME+0x2e0 / +0x2e4: cost of mvp_idx 0 / 1                                 */
    int cand_on[2], cand[2][2];                       /* TPredUnit+0x110+l / +0x114+4l and +0x11c+l / +0x120+4l                    */
    int layer_enabled, layer_on, layer_mv[2];         /* TEncParam+0x4a0; the look-ahead coded the block inter; its half-res vector */
    int prev_on, prev[2];                             /* TPredUnit+0x1f2+l / +0x1f4+4l before the call                             */
    const uint16_t *base;                             /* tME+0x10: centre of the mvd cost table, valid for [-256, 256] at least    */
    uint32_t (*dist)(void *user, long off); void *user;   /* TPredUnit+0x38 on the block at plane offset off                        */
    int mvp_idx, mx, my; uint32_t cost, sad;          /* tME+0x58, +0x54 / +0x56, +0x90, TPredUnit+0x150                           */
    int zero_tried, outside, win[4]; long off;        /* tME+0x5c, +0x65, +0x6c..+0x72, +0x40 - plane                              */
    int prev_on_out, prev_out[2], cmx_off, cmy_off;   /* TPredUnit+0x1f2+l / +0x1f4+4l after; tME+0x18 / +0x20 - tME+0x10          */
} kso_me_init;
void kso_ref_me_init_point(kso_me_init *m);
/* the cost of a quarter-pel difference d as meInitPoint prices a start point outside its window: base[d] for |d| <= 256, else lambda x (3 + 2 floor(log2 |d|)) */
uint32_t kso_ref_mvd_cost_far(const uint16_t *base, int lambda, int d);
/* trace replay (tests/test_me_init.py): h = the 64-word record oracle/ref_probe/init_shim.c writes, tab513 = base[-256..256].  The block comparisons are
 * answered from the record, in order; -1 if the restatement asks for one the reference did not make (or fewer), else 0 and out[] = the results. */
int kso_me_init_replay(const int32_t h[64], const uint16_t *tab513, int32_t out[20]);
#ifdef __cplusplus
}
#endif
#endif
