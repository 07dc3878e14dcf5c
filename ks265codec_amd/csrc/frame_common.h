// frame_common.h — whole-frame stage plumbing shared by frame_*.hip (not part of the C ABI)
#pragma once
#include "ks265_internal.h"

#define KS_PAD_Y 80
#define KS_PAD_C 40
#define KS_COST_INVALID 0xFFFFFFFFu
#define KS_NSTAGE 8                   // me_integer, me_subpel, intra_candidates, cu_decide (+ merge pass), reconstruct, intra_pass, deblock, sao (+ padding)

// geometry handed to kernels by value
// cfg.part: what a CU of 64 / 32 / 16 samples costs in two halves (ks265_rect_decide): [0] = 2NxN (top, bottom), [1] = Nx2N (left, right); cost KS_COST_INVALID = not considered
struct KsRect { unsigned cost[2]; short mv[2][2][2]; short mv1[2][2][2]; unsigned char dir[2][2]; };   // B pictures: mv1 / dir = the half's list-1 vector and direction (ks265_cu_decide_part_b)

// Multi-reference B pictures (round 5; -preset veryslow = config 5: ref 4 / 4).  A block's pictures come from its record: inter_dir = direction | idx0 << 4 | idx1 << 6 (an
// index of a list the block does not use is 0: equal bytes = equal pictures, which is what the boundary strength compares).  Kernels pick a list's picture with selects - an
// indexed pointer array would lose the global address space.  idx0 / idx1: per PU record (85 per CTU) the picture ks265_ref_pick chose for the list; bits: lambda x ref_idx bits.
struct KsRefList { const uint8_t *p[4]; };
__device__ __forceinline__ const uint8_t *ks_pick(const KsRefList &r, int i) { return i == 0 ? r.p[0] : (i == 1 ? r.p[1] : (i == 2 ? r.p[2] : r.p[3])); }
struct KsMrefB { KsRefList y0, y1; const uint8_t *idx0, *idx1; int bits0[4], bits1[4]; };

struct KsGeom {
    int W, H;                 // luma size
    int sy, sc;               // strides
    long bytes_y, bytes_c;
    int ctu_cols, ctu_rows;
    int w8, h8;
    long org_y, org_c;        // byte offset of sample (0,0) inside a padded plane
};

struct ks265_frame {
    ks265_ctx *ctx = nullptr;
    ks265_frame_cfg cfg{};
    ks265_frame_cfg cfg0{};             // the tools the frame object was created with (ks265_frame_set_picture_tools lowers and restores cfg's)
    ks265_frame_geom geom{};
    KsGeom g{};
    // workspace (device)
    // (no fractional planes: every consumer interpolates from the reference picture itself, interp_dev.h)
    ks265_pu *pu_x[3] = {nullptr, nullptr, nullptr};
    ks265_pu *pu1 = nullptr;
    ks265_pu *pu_s = nullptr;            // cfg.propagate: the other side of the propagation rounds
    ks265_pu_b *pub = nullptr;
    ks265_pu *pu[2] = {nullptr, nullptr};
    int cur_pu = 0;
    bool have_prev = false;
    ks265_cu8 *cu8 = nullptr;
    ks265_cu8 *cu8_tmp = nullptr;        // cfg.merge: the CU decision's map, input of the merge pass
    ks265_sao_param *sao = nullptr;
    int16_t *lvl[3] = {nullptr, nullptr, nullptr};
    uint8_t *deb[3] = {nullptr, nullptr, nullptr};   // reconstructed picture before SAO (padded geometry)
    unsigned long long *sse = nullptr;
    unsigned long long *sse_acc = nullptr;   // ks265_sse_picture: three running sums + finished work-groups (zero between calls)
    short *mats = nullptr;              // forward + transposed DCT matrices of all sizes in the kernels' LDS layout (2 x MAT_SHORTS)
    int *progress = nullptr;            // intra wavefront: CTUs finished per CTU row
    const int8_t *qp_map = nullptr;      // round 4: one QP per CTU (ks265_frame_set_qp_map; null = cfg.qp everywhere), device memory of the host
    uint8_t *qp_eff = nullptr;           // ... and the QpY of every 8x8 block as the decoder derives it (deblocking), w8 * h8 bytes, allocated on first use
    void *rec_fence = nullptr;               // event the next picture waits for before it writes a record (ks265_frame_set_records_fence); consumed by that picture
    void *rect = nullptr;                                  // cfg.part: the 2NxN / Nx2N records of the P picture being coded (KsRect, 21 per CTU)
    int *ic_work = nullptr;                                // ... the CTUs that passed the candidates' gate: [0] = count, [4 ..] = CTU indices (round 5)
    uint32_t *icost = nullptr;                             // cfg.intra_inter: intra candidates of the P / B picture being coded (85 per CTU: cost << 6 | mode)
    uint8_t *pyr[10] = {};              // pre-search (cfg.pre_search): L1 / L2 of the source, L1 / L2 of the reference, L2 / L1 vectors, the 16x16 field, L3 of both, CTU window offsets
    // round 5: the two uni-directional searches of a B picture are independent chains of latency-bound kernels - list 1's runs on a side stream beside list 0's
    // (ks265_encode_picture_b).  Its own pre-search workspace (the source pyramid [0] [1] [7] is shared and built once, before the fork), propagation scratch, stream, fork / join events
    uint8_t *pyr2[10] = {};
    ks265_pu *pu_s2 = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // heavy-first dispatch of the integer search (frame_me_int.hip me_order_kernel): rounds per CTU wave of the previous search, the order built from them; [1] = the side stream's
    unsigned *me_work = nullptr, *me_work_all[2] = {nullptr, nullptr};
    int *me_order = nullptr, *me_order_all[2] = {nullptr, nullptr};
    int me_order_off = 0;               // KS265_ME_ORDER_OFF: experiments
    // multi-reference B picture being coded (ks265_encode_picture_b_mref sets it for the duration of the picture: ks265_bi_decide, ks265_merge_pass, ks265_cu_decide_part_b and
    // ks265_reconstruct_b then take every block's pictures from its record); host-side copies of the lists + the per-PU index arrays and the extra list-1 PU records
    // cfg.rdoq (round 6; -rdoq 1 = the SDK's rdoq field, qy265enc.h:129): the luma transform blocks of inter CUs go through the reference's rdoQuant (rdoq_ops.hip) - front half of
    // ks265_reconstruct* (coefficients + levels rounded at 1 / 2 to planes), rdoq_prep_kernel (the blocks listed, packed, their significance masks and last positions), rdoq_kernel,
    // rdoq_unpack_kernel, back half (dequantisation, inverse transform, reconstruction).  Workspace allocated by ks265_frame_set_rdoq; tables + lambdas from the host per picture
    int16_t *rq_coef = nullptr, *rq_pack_lvl = nullptr, *rq_pack_coef = nullptr;
    ks265_rdoq_tu *rq_tus = nullptr;
    int *rq_pos = nullptr, *rq_ctr = nullptr, *rq_out = nullptr;
    int32_t *rq_tab = nullptr;
    long long *rq_lam = nullptr;
    uint16_t *rq_sigmask = nullptr;
    unsigned long long *rq_hidden = nullptr;
    bool rq_ready = false;
    bool mrefb = false;
    bool mr_pslice = false;             // round 6: the context is a multi-reference P picture's (ks265_encode_picture_mref: two-list records, one list in the slice - the merge pass's zero candidate is uni-directional)
    int mr_n[2] = {1, 1};
    ks265_pic mr_pic[2][4] = {};
    uint8_t *ridx[2] = {nullptr, nullptr};
    ks265_pu *pu1_x[3] = {nullptr, nullptr, nullptr};
    bool src_pyr_ready = false;         // ks265_presearch: the source picture's pyramid is in place (skip its pyr_down launch)
    int b_parallel = 1;                 // 0: the two searches one after the other on the context's stream (graph capture, experiments: KS265_B_SERIAL)
    // optional in-situ stage timing (HIP events on the context's stream, between the stages of ks265_encode_picture)
    bool profiling = false;
    hipEvent_t ev[KS_NSTAGE + 1] = {};
    hipEvent_t ev_k[2] = {}; bool ev_k_valid = false;      // profiling: around the me_int_kernel launch alone (the SAD kernel of the roofline figure: stage me_integer also holds pre-search + propagation)
    bool ev_valid[KS_NSTAGE + 1] = {};
};

static inline ks265_pic ks_deb_pic(ks265_frame *f) { return ks265_pic{f->deb[0], f->deb[1], f->deb[2]}; }

__device__ __forceinline__ const uint8_t *ks_org_y(const KsGeom &g, const uint8_t *p) { return p + g.org_y; }
__device__ __forceinline__ uint8_t *ks_org_y(const KsGeom &g, uint8_t *p) { return p + g.org_y; }
__device__ __forceinline__ const uint8_t *ks_org_c(const KsGeom &g, const uint8_t *p) { return p + g.org_c; }
__device__ __forceinline__ uint8_t *ks_org_c(const KsGeom &g, uint8_t *p) { return p + g.org_c; }

// PU indexing inside a CTU: level l (0: 64x64 .. 3: 8x8), raster
__device__ __forceinline__ int ks_level_base(int l) { return l == 0 ? 0 : l == 1 ? 1 : l == 2 ? 5 : 21; }
__device__ __forceinline__ int ks_pu_index(int l, int px, int py) { return ks_level_base(l) + py * (1 << l) + px; }
__device__ __forceinline__ bool ks_pu_inside(const KsGeom &g, int cx, int cy, int l, int px, int py)
{
    int s = 64 >> l, x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
    return x0 + s <= g.W && y0 + s <= g.H;
}

// XCD-aware block -> work-item map (cdna_hip_programming.md T1): block b runs on XCD b % 8; give every XCD a contiguous
// raster range of CTUs so that horizontally / vertically adjacent CTUs (which share reference rows and 128-byte lines)
// meet in the same 4 MiB L2.  Bijective for any n; speed only, never correctness.
__device__ __forceinline__ int ks_xcd_swizzle(int b, int n)
{
    const int per = n >> 3, rem = n & 7;                 // XCD x owns per + (x < rem) items
    const int x = b & 7, j = b >> 3;
    return x * per + min(x, rem) + j;
}

// legal vectors of the PUs of a CTU around its window offset (pre-search, else zero): +-range around the offset, and never so far that a block of the CTU
// leaves the 64-sample margin of the padded planes
__device__ __forceinline__ void ctu_mv_limits(const KsGeom &g, int range, int cx, int cy, int ox, int oy, int &lox, int &hix, int &loy, int &hiy)
{
    const int xe = min(cx * 64 + 64, g.W), ye = min(cy * 64 + 64, g.H);
    lox = max(ox - range, -64 - cx * 64); hix = min(ox + range, g.W + 64 - xe);
    loy = max(oy - range, -64 - cy * 64); hiy = min(oy + range, g.H + 64 - ye);
}

int ks265_frame_build_matrices(ks265_frame *f);      // frame_recon.hip
int ks265_presearch_source(ks265_frame *f, ks265_pic src);   // frame_presearch.hip
int ks265_sao_off(ks265_frame *f, ks265_sao_param *sao, ks265_pic dst);   // frame_loop.hip: the tail of a picture coded without SAO, in place

// every frame-level entry point: the calling thread's current device is the frame's (a host with one encoder lane per GPU drives several devices from several
// threads; kernel launches go to the CURRENT device's streams only)
#define KS_FRAME_CHECK(f) do { if (!(f) || !(f)->ctx) return KS265_POINTER; ks_use_device((f)->ctx); } while (0)
