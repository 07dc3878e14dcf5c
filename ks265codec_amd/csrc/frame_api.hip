// frame_api.hip — ks265_frame lifetime and the per-picture sequencing of the stages (include/ks265_hip.h §3).
// The order is the reference's per-CTU order hoisted to whole pictures: ME -> sub-pel -> CU decision ->
// reconstruct -> deblock -> SAO; every stage is one or two launches on the context's stream, no host synchronisation.
#include "frame_common.h"

static int dev_alloc(ks265_ctx *ctx, void **p, size_t bytes, bool zero)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return ks265_hip(ctx, e);
    if (zero) {
        e = hipMemsetAsync(*p, 0, bytes, ctx->stream);
        if (e != hipSuccess) return ks265_hip(ctx, e);
    }
    return KS265_OK;
}

extern "C" {

int ks265_frame_create(ks265_ctx *ctx, const ks265_frame_cfg *cfg, ks265_frame **out)
{
    if (!ctx || !cfg || !out) return KS265_POINTER;
    *out = nullptr;
    ks265_frame_geom geom;
    int r = ks265_frame_geometry(cfg, &geom);
    if (r) return r;
    if (cfg->me_method < 0 || cfg->me_method > 2) return KS265_NOTSUPPORTED;   /* 0 = DIA, 1 = HEX, 2 = UMH (-me); EPZS / Cross not built */
    if (cfg->refs > 4 || cfg->propagate < 0 || cfg->propagate > 4) return KS265_NOTSUPPORTED;
    if ((long long)geom.bytes_y >= (1ll << 31)) return KS265_NOTSUPPORTED;         /* stage B addresses a luma plane with 32-bit offsets */
    ks265_frame *f = new ks265_frame();                                            /* every validation above: nothing to undo on those returns */
    f->ctx = ctx; f->cfg = *cfg; f->cfg0 = *cfg; f->geom = geom;
    f->me_order_off = getenv("KS265_ME_ORDER_OFF") ? 1 : 0;
    KsGeom &g = f->g;
    g.W = cfg->width; g.H = cfg->height; g.sy = geom.stride_y; g.sc = geom.stride_c; g.bytes_y = geom.bytes_y; g.bytes_c = geom.bytes_c;
    g.ctu_cols = geom.ctu_cols; g.ctu_rows = geom.ctu_rows; g.w8 = cfg->width / 8; g.h8 = cfg->height / 8;
    g.org_y = (long)KS_PAD_Y * g.sy + KS_PAD_Y; g.org_c = (long)KS_PAD_C * g.sc + KS_PAD_C;
    (void)hipSetDevice(ctx->device);
    const size_t npx = (size_t)g.W * g.H;
    r = KS265_OK;
    for (int i = 0; i < 2 && !r; ++i) r = dev_alloc(ctx, (void **)&f->pu[i], (size_t)geom.bytes_pu, true);
    if (!r && cfg->propagate) r = dev_alloc(ctx, (void **)&f->pu_s, (size_t)geom.bytes_pu, true);
    if (!r && cfg->bframes > 0) {
        r = dev_alloc(ctx, (void **)&f->pu1, (size_t)geom.bytes_pu, true);
        if (!r && cfg->propagate) r = dev_alloc(ctx, (void **)&f->pu_s2, (size_t)geom.bytes_pu, true);
        if (!r) r = ks265_hip(ctx, hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
        if (!r) r = ks265_hip(ctx, hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming));
        if (!r) r = ks265_hip(ctx, hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming));
        f->b_parallel = getenv("KS265_B_SERIAL") ? 0 : 1;
        if (!r) r = dev_alloc(ctx, (void **)&f->pub, (size_t)geom.ctu_cols * geom.ctu_rows * 85 * sizeof(ks265_pu_b), true);
    }
    for (int x = 0; x + 1 < cfg->refs && !r; ++x) {              /* list-0 pictures 1..refs-1 of multi-reference P pictures */
        r = dev_alloc(ctx, (void **)&f->pu_x[x], (size_t)geom.bytes_pu, true);
    }
    if (cfg->refs > 1 && !f->pub && !r) r = dev_alloc(ctx, (void **)&f->pub, (size_t)geom.ctu_cols * geom.ctu_rows * 85 * sizeof(ks265_pu_b), true);
    if (cfg->refs > 1 && cfg->bframes > 0) {                       /* multi-reference B pictures: list 1's pictures 1 .. refs - 1, and per list the index of every PU's picture */
        for (int x = 0; x + 1 < cfg->refs && !r; ++x) r = dev_alloc(ctx, (void **)&f->pu1_x[x], (size_t)geom.bytes_pu, true);
        for (int l = 0; l < 2 && !r; ++l) r = dev_alloc(ctx, (void **)&f->ridx[l], (size_t)geom.ctu_cols * geom.ctu_rows * 85, true);
    }
    if (!r && cfg->intra_inter) {                                  /* intra candidates of P / B pictures: cost and mode of every block */
        r = dev_alloc(ctx, (void **)&f->icost, (size_t)geom.ctu_cols * geom.ctu_rows * 85 * sizeof(uint32_t), true);
    }
    if (!r && cfg->part) r = dev_alloc(ctx, &f->rect, (size_t)geom.ctu_cols * geom.ctu_rows * 21 * sizeof(KsRect), true);
    if (!r) r = dev_alloc(ctx, (void **)&f->cu8, (size_t)geom.bytes_cu8, true);
    if (!r && (cfg->merge || cfg->skip_rd)) r = dev_alloc(ctx, (void **)&f->cu8_tmp, (size_t)geom.bytes_cu8, true);      /* the CU decision's map in front of the merge pass; the skip pass's snapshot */
    if (!r) r = dev_alloc(ctx, (void **)&f->sao, (size_t)geom.bytes_sao, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->lvl[0], npx * 2, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->lvl[1], npx / 2, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->lvl[2], npx / 2, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->deb[0], (size_t)g.bytes_y, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->deb[1], (size_t)g.bytes_c, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->deb[2], (size_t)g.bytes_c, true);
    if (!r) r = dev_alloc(ctx, (void **)&f->sse, 3 * sizeof(unsigned long long), true);
    if (!r) r = dev_alloc(ctx, (void **)&f->sse_acc, 4 * sizeof(unsigned long long), true);
    if (!r) r = dev_alloc(ctx, (void **)&f->progress, sizeof(int) * (size_t)g.ctu_rows * g.ctu_cols, true);   /* intra wavefront flags: per CTU row (key pictures), per CTU (cfg.intra_inter) */
    if (!r) r = dev_alloc(ctx, (void **)&f->mats, sizeof(short) * 2 * 1600, true);      /* 2 x MAT_SHORTS (recon_dev.h) */
    if (!r) r = ks265_frame_build_matrices(f);
    if (r) { ks265_frame_destroy(f); return r; }
    *out = f;
    return KS265_OK;
}

void ks265_frame_destroy(ks265_frame *f)
{
    if (!f) return;
    if (f->ctx) { (void)hipSetDevice(f->ctx->device); (void)hipStreamSynchronize(f->ctx->stream); }
    for (int i = 0; i <= KS_NSTAGE; ++i)
        if (f->ev[i]) (void)hipEventDestroy(f->ev[i]);
    if (f->side) { (void)hipStreamSynchronize(f->side); (void)hipStreamDestroy(f->side); }
    for (int i = 0; i < 2; ++i) { if (f->me_work_all[i]) (void)hipFree(f->me_work_all[i]); if (f->me_order_all[i]) (void)hipFree(f->me_order_all[i]); }
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    if (f->ev_join) (void)hipEventDestroy(f->ev_join);
    for (int i = 0; i < 10; ++i)
        if (f->pyr2[i] && f->pyr2[i] != f->pyr[i]) (void)hipFree(f->pyr2[i]);
    void *ptrs[] = {f->rq_coef, f->rq_pack_lvl, f->rq_pack_coef, f->rq_tus, f->rq_pos, f->rq_ctr, f->rq_out, f->rq_tab, f->rq_lam, f->rq_sigmask, f->rq_hidden, f->ic_work, f->pu1_x[0], f->pu1_x[1], f->pu1_x[2], f->ridx[0], f->ridx[1], f->pu_s2, f->pu1, f->pu_s, f->pub, f->pu[0], f->pu[1], f->cu8, f->sao, f->lvl[0], f->lvl[1], f->lvl[2], f->deb[0], f->deb[1], f->deb[2], f->sse, f->sse_acc, f->cu8_tmp, f->progress, f->mats, f->icost, f->rect, f->pu_x[0], f->pu_x[1], f->pu_x[2]};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (uint8_t *p : f->pyr)
        if (p) (void)hipFree(p);
    if (f->qp_eff) (void)hipFree(f->qp_eff);
    delete f;
}

/* one QP per CTU (raster order, device memory of the host: it must stay valid until the pictures coded with it have run) for every picture coded from here on; null = cfg.qp
 * everywhere.  The residual of a CTU is quantised with its entry, the deblocking filter reads the QpY the decoder derives (cu_qp_delta, quantisation group = CTU) */
int ks265_frame_set_qp_map(ks265_frame *f, const int8_t *dev_qp_map)
{
    KS_FRAME_CHECK(f);
    f->qp_map = dev_qp_map;
    return KS265_OK;
}
int ks265_frame_set_qp(ks265_frame *f, int qp, int lambda_q4)
{
    KS_FRAME_CHECK(f);
    if (qp < 0 || qp > 51 || lambda_q4 < 0) return KS265_NOTSUPPORTED;
    f->cfg.qp = qp; f->cfg.lambda_q4 = lambda_q4;
    return KS265_OK;
}

int ks265_frame_set_picture_tools(ks265_frame *f, int intra_inter, int bi_refine, int sao, int me_method)
{
    KS_FRAME_CHECK(f);
    const int ii = intra_inter < 0 ? f->cfg0.intra_inter : intra_inter, br = bi_refine < 0 ? f->cfg0.bi_refine : bi_refine, so = sao < 0 ? f->cfg0.sao : sao;
    if ((ii && ii != f->cfg0.intra_inter) || (br && br != f->cfg0.bi_refine) || (so && so != f->cfg0.sao)) return KS265_NOTSUPPORTED;   /* off, or what the workspace was made for */
    if (me_method > 2) return KS265_NOTSUPPORTED;
    f->cfg.intra_inter = ii; f->cfg.bi_refine = br; f->cfg.sao = so; f->cfg.me_method = me_method < 0 ? f->cfg0.me_method : me_method;
    return KS265_OK;
}

/* stage A (+ A2): the integer search of src in one reference picture, then cfg.propagate rounds of vector propagation; the records end up in pu */
static int me_search(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *prev, ks265_pu *pu)
{
    const int n = f->cfg.propagate;
    ks265_pu *a = (n & 1) ? f->pu_s : pu, *b = (n & 1) ? pu : f->pu_s;      /* n rounds swap sides n times */
    int r = ks265_me_integer(f, src, ref, prev, a);
    for (int i = 0; i < n && !r; ++i) {
        r = ks265_me_propagate(f, src, ref, a, b);
        ks265_pu *t = a; a = b; b = t;
    }
    return r;
}

/* the records of the previous picture (CU map, levels, SAO parameters) may still be read on another stream (ks265_frame_pack_compact_on): the host names the event that
 * ends that, and the picture waits for it right before its first kernel that writes a record - its search runs beside the previous picture's drain */
int ks265_frame_set_records_fence(ks265_frame *f, void *ev)
{
    KS_FRAME_CHECK(f);
    f->rec_fence = ev;
    return KS265_OK;
}
static int records_fence(ks265_frame *f)
{
    if (!f->rec_fence) return KS265_OK;
    void *ev = f->rec_fence;
    f->rec_fence = nullptr;
    return ks265_hip(f->ctx, hipStreamWaitEvent(f->ctx->stream, (hipEvent_t)ev, 0));
}
int ks265_encode_picture(ks265_frame *f, ks265_pic src, ks265_pic ref, int is_key, ks265_pic recon_out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !recon_out.y) return KS265_POINTER;
    int r;
    ks265_pu *pu = f->pu[f->cur_pu];
    int evi = 0;
    auto mark = [&](int stage_done) {                      // stage boundary: event stage_done closes stage (stage_done - 1)
        if (!f->profiling) return;
        while (evi < stage_done) f->ev_valid[evi++] = false;   // skipped stages of a key picture
        (void)hipEventRecord(f->ev[stage_done], f->ctx->stream);
        f->ev_valid[stage_done] = true;
        evi = stage_done + 1;
    };
    ks265_pic deb = ks_deb_pic(f);
    if (is_key) {
        /* intra picture: mode pre-selection + CU tree on the source, then the wavefront reconstruction */
        mark(2);
        if ((r = records_fence(f))) return r;
        if ((r = ks265_intra_decide(f, src, f->cu8))) return r;
        mark(3);
        mark(5);
        if ((r = ks265_intra_reconstruct(f, src, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
        f->have_prev = false;
    } else {
        if (!ref.y) return KS265_POINTER;
        mark(0);
        if ((r = me_search(f, src, ref, f->have_prev ? f->pu[f->cur_pu ^ 1] : nullptr, pu))) return r;
        mark(1);
        if (f->cfg.subme && (r = ks265_me_subpel(f, src, ref, pu))) return r;
        mark(2);
        const bool ii = f->cfg.intra_inter != 0;              /* intra CUs may compete: their candidates first */
        if (ii && (r = ks265_intra_candidates(f, src, pu, f->icost))) return r;
        mark(3);
        if ((r = records_fence(f))) return r;
        const bool part = f->cfg.part != 0;                    /* -part 1: the halves of every 64 / 32 / 16 CU priced first, the CU decision then picks among 2Nx2N / 2NxN / Nx2N / split */
        if (f->cfg.merge) {                                    /* stage C2: the CU decision goes to the spare map, the merge pass writes the final one */
            if ((r = part ? ks265_cu_decide_part(f, src, ref, pu, ii ? f->icost : nullptr, f->cu8_tmp) : ks265_cu_decide_ii(f, pu, ii ? f->icost : nullptr, f->cu8_tmp))) return r;
            if ((r = ks265_merge_pass(f, src, ref, ks265_pic{nullptr, nullptr, nullptr}, pu, nullptr, f->cu8_tmp, f->cu8))) return r;
        } else if ((r = part ? ks265_cu_decide_part(f, src, ref, pu, ii ? f->icost : nullptr, f->cu8) : ks265_cu_decide_ii(f, pu, ii ? f->icost : nullptr, f->cu8))) return r;
        mark(4);
        if ((r = ks265_reconstruct(f, src, ref, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
        if (f->cfg.skip_rd >= 2 && (r = ks265_skip_pass(f, src, ref, ks265_pic{nullptr, nullptr, nullptr}, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;   /* stage D2: on the coded distortion (1: B pictures only) */
        mark(5);
        if (ii && (r = ks265_intra_inter_reconstruct(f, src, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
    }
    mark(6);
    if (f->cfg.deblock && (r = ks265_deblock(f, f->cu8, deb))) return r;
    mark(7);
    if ((r = ks265_sao(f, src, deb, f->sao, recon_out))) return r;
    mark(8);
    if (!is_key) { f->cur_pu ^= 1; f->have_prev = true; }
    return KS265_OK;
}

/* P picture with several list-0 pictures: one search per picture, the per-PU choice, the CU tree, then the common back end */
int ks265_encode_picture_mref(ks265_frame *f, ks265_pic src, const ks265_pic *refs, int nref, ks265_pic recon_out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !refs || !recon_out.y) return KS265_POINTER;
    if (nref < 1 || nref > (f->cfg.refs > 1 ? f->cfg.refs : 1)) return KS265_NOTSUPPORTED;
    if (nref == 1) return ks265_encode_picture(f, src, refs[0], 0, recon_out);
    int r;
    ks265_pu *pu0 = f->pu[f->cur_pu];
    const ks265_pu *pus[4] = {pu0, f->pu_x[0], f->pu_x[1], f->pu_x[2]};
    for (int i = 0; i < nref; ++i) if (!refs[i].y) return KS265_POINTER;
    /* one picture's chain: pre-search, integer search, propagation, sub-pel refinement; the temporal predictor (previous picture's vectors) belongs to the nearest picture only */
    auto chain = [&](int i) -> int {
        ks265_pu *pu = i == 0 ? pu0 : f->pu_x[i - 1];
        int rr = me_search(f, src, refs[i], i == 0 && f->have_prev ? f->pu[f->cur_pu ^ 1] : nullptr, pu);
        if (!rr && f->cfg.subme) rr = ks265_me_subpel(f, src, refs[i], pu);
        return rr;
    };
    const bool par = f->b_parallel && !f->ctx->capturing && f->side && f->pyr2[0] && f->cfg.pre_search && (!f->cfg.propagate || f->pu_s2);
    if (par) {
        /* round 6: the chains of the farther pictures on the side stream beside the nearest picture's (the B pictures' two lists do the same, encode_b_lists): chains of
         * latency-bound kernels that leave most of the device idle - an anchor with three pictures takes two chains' time instead of three */
        if ((r = ks265_presearch_source(f, src))) return r;
        if ((r = ks265_hip(f->ctx, hipEventRecord(f->ev_fork, f->ctx->stream)))) return r;
        if ((r = ks265_hip(f->ctx, hipStreamWaitEvent(f->side, f->ev_fork, 0)))) return r;
        hipStream_t mainst = f->ctx->stream;
        auto swap_ws = [&]() {
            for (int i = 0; i < 10; ++i) { uint8_t *t = f->pyr[i]; f->pyr[i] = f->pyr2[i]; f->pyr2[i] = t; }
            ks265_pu *t = f->pu_s; f->pu_s = f->pu_s2; f->pu_s2 = t;
            if (f->me_work) { const bool side = f->me_work == f->me_work_all[0]; f->me_work = f->me_work_all[side]; f->me_order = f->me_order_all[side]; }
        };
        f->src_pyr_ready = true;
        f->ctx->stream = f->side; swap_ws();
        r = 0;
        for (int i = 1; i < nref && !r; ++i) r = chain(i);
        const int rj = ks265_hip(f->ctx, hipEventRecord(f->ev_join, f->side));      /* (joined whatever happened after the fork: see encode_b_lists) */
        f->ctx->stream = mainst; swap_ws();
        if (!r) r = chain(0);
        f->src_pyr_ready = false;
        const int rw = rj ? ks265_hip(f->ctx, hipStreamSynchronize(f->side)) : ks265_hip(f->ctx, hipStreamWaitEvent(f->ctx->stream, f->ev_join, 0));
        if (r) return r;
        if (rw) return rw;
    } else
        for (int i = 0; i < nref; ++i) if ((r = chain(i))) return r;
    if ((r = ks265_ref_decide(f, nref, pus, f->pub))) return r;
    /* round 6 (-ref0: the anchors of the pyramid GOPs search several past anchors): the two-list records go through the stages a one-reference P picture has - intra candidates
     * against them (cfg.intra_inter), the CU tree, the merge pass on the records' pictures (cfg.merge), the intra CUs' pass; -part 1 stays with one reference picture */
    const bool ii = f->cfg.intra_inter != 0 && f->icost;
    if (ii && (r = ks265_intra_candidates(f, src, f->pub, f->icost))) return r;
    if ((r = records_fence(f))) return r;
    const bool mg = f->cfg.merge && f->cu8_tmp;
    if ((r = ks265_cu_decide_b_ii(f, f->pub, ii ? f->icost : nullptr, mg ? f->cu8_tmp : f->cu8))) return r;
    /* the merge pass and the skip pass take a candidate's picture from its record (the multi-reference form); list 1 does not exist: mr_pslice makes the zero candidate uni-directional */
    auto mr_scope = [&](auto &&body) -> int {
        struct MrScope { ks265_frame *f; ~MrScope() { f->mrefb = false; f->mr_pslice = false; } } scope{f};
        f->mrefb = true; f->mr_pslice = true; f->mr_n[0] = nref; f->mr_n[1] = 1;
        for (int i = 0; i < 4; ++i) { f->mr_pic[0][i] = refs[i < nref ? i : nref - 1]; f->mr_pic[1][i] = refs[0]; }
        return body();
    };
    if (mg && (r = mr_scope([&] { return ks265_merge_pass(f, src, refs[0], refs[0], nullptr, f->pub, f->cu8_tmp, f->cu8); }))) return r;
    ks265_pic deb = ks_deb_pic(f);
    if ((r = ks265_reconstruct_mref(f, src, nref, refs, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
    if (f->cfg.skip_rd >= 2 && f->cu8_tmp && (r = mr_scope([&] { return ks265_skip_pass(f, src, refs[0], ks265_pic{nullptr, nullptr, nullptr}, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb); }))) return r;
    if (ii && (r = ks265_intra_inter_reconstruct(f, src, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
    if (f->cfg.deblock && (r = ks265_deblock(f, f->cu8, deb))) return r;
    if ((r = ks265_sao(f, src, deb, f->sao, recon_out))) return r;
    f->cur_pu ^= 1; f->have_prev = true;                          /* the nearest picture's vectors seed the next picture */
    return KS265_OK;
}

/* B picture: both uni-directional searches, the bi candidate, the CU tree, then the common back end.
 * The temporal predictor chain of the P pictures (pu ping-pong) is left untouched. */
static int encode_b_lists(ks265_frame *f, ks265_pic src, const ks265_pic *refs0, int n0, const ks265_pic *refs1, int n1, ks265_pic recon_out);
int ks265_encode_picture_b(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, ks265_pic recon_out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref0.y || !ref1.y || !recon_out.y) return KS265_POINTER;
    if (!f->pu1) return KS265_NOTSUPPORTED;                   /* created with cfg.bframes == 0 */
    return encode_b_lists(f, src, &ref0, 1, &ref1, 1, recon_out);
}
/* a B picture with n0 / n1 <= cfg.refs pictures per list (round 5; -ref with B pictures: config 5 = -preset veryslow resolves to 4 / 4): list 0 = pictures BEFORE this one in
 * display order, nearest first; list 1 = pictures after it, nearest first; no picture in both lists (the boundary strength compares list indices).  Every picture of a list is
 * searched on its own (list 1's chain on the side stream beside list 0's), ks265_ref_pick keeps per PU and list the cheapest (+ lambda x ref_idx bits), the bi-predictive decision
 * pairs the winners; ks265_cu8.inter_dir = direction | idx0 << 4 | idx1 << 6 */
int ks265_encode_picture_b_mref(ks265_frame *f, ks265_pic src, const ks265_pic *refs0, int n0, const ks265_pic *refs1, int n1, ks265_pic recon_out)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !refs0 || !refs1 || !recon_out.y) return KS265_POINTER;
    if (!f->pu1) return KS265_NOTSUPPORTED;
    const int cap = f->cfg.refs > 1 ? f->cfg.refs : 1;
    if (n0 < 1 || n1 < 1 || n0 > cap || n1 > cap || n0 > 4 || n1 > 4) return KS265_NOTSUPPORTED;
    if ((n0 > 1 || n1 > 1) && (!f->ridx[0] || !f->pu_x[0])) return KS265_NOTSUPPORTED;
    for (int i = 0; i < n0; ++i) if (!refs0[i].y) return KS265_POINTER;
    for (int i = 0; i < n1; ++i) if (!refs1[i].y) return KS265_POINTER;
    return encode_b_lists(f, src, refs0, n0, refs1, n1, recon_out);
}
static int encode_b_lists(ks265_frame *f, ks265_pic src, const ks265_pic *refs0, int n0, const ks265_pic *refs1, int n1, ks265_pic recon_out)
{
    const ks265_pic ref0 = refs0[0], ref1 = refs1[0];
    const bool mr = n0 > 1 || n1 > 1;
    int r;
    ks265_pu *pu0 = f->pu[f->cur_pu];                         /* scratch: the next P picture overwrites it */
    /* one list's chain on the stream the context holds at the moment: every picture of the list searched and refined, then (several pictures) the per-PU choice */
    auto chain = [&](int l) -> int {
        const ks265_pic *refs = l ? refs1 : refs0; const int n = l ? n1 : n0;
        const ks265_pu *pus[4];
        int rr = 0;
        for (int i = 0; i < n && !rr; ++i) {
            ks265_pu *pu = i == 0 ? (l ? f->pu1 : pu0) : (l ? f->pu1_x[i - 1] : f->pu_x[i - 1]);
            pus[i] = pu;
            rr = me_search(f, src, refs[i], nullptr, pu);
            if (!rr && f->cfg.subme) rr = ks265_me_subpel(f, src, refs[i], pu);
        }
        if (!rr && mr) rr = ks265_ref_pick(f, n, pus, l ? f->pu1 : pu0, f->ridx[l]);
        return rr;
    };
    const bool par = f->b_parallel && !f->ctx->capturing && f->side && f->cfg.pre_search && (!f->cfg.propagate || f->pu_s2);   /* (a picture being captured as a graph stays on the one stream: the fork's host-side workspace swaps are not part of a graph) */
    if (par) {
        /* the two searches are independent chains (pre-search, integer search, propagation, sub-pel refinement: ~ 0.4 ms each at 2160p, each kernel with a long tail of
         * half-empty CUs): list 1's is enqueued on the side stream with its own workspace and runs beside list 0's; the join is in front of the bi-predictive decision */
        if ((r = ks265_presearch_source(f, src))) return r;                        /* the source pyramid both use, once, before the fork */
        if ((r = ks265_hip(f->ctx, hipEventRecord(f->ev_fork, f->ctx->stream)))) return r;
        if ((r = ks265_hip(f->ctx, hipStreamWaitEvent(f->side, f->ev_fork, 0)))) return r;
        hipStream_t mainst = f->ctx->stream;
        auto swap_ws = [&]() {
            for (int i = 0; i < 10; ++i) { uint8_t *t = f->pyr[i]; f->pyr[i] = f->pyr2[i]; f->pyr2[i] = t; }
            ks265_pu *t = f->pu_s; f->pu_s = f->pu_s2; f->pu_s2 = t;
            if (f->me_work) { const bool side = f->me_work == f->me_work_all[0]; f->me_work = f->me_work_all[side]; f->me_order = f->me_order_all[side]; }
        };
        f->src_pyr_ready = true;
        f->ctx->stream = f->side; swap_ws();
        r = chain(1);
        /* the join is recorded and waited for whatever happened after the fork: what the side stream has been handed keeps reading src / refs and writing this frame's
         * workspace, and the host recycles those on its error path */
        const int rj = ks265_hip(f->ctx, hipEventRecord(f->ev_join, f->side));
        f->ctx->stream = mainst; swap_ws();
        if (!r) r = chain(0);
        f->src_pyr_ready = false;
        const int rw = rj ? ks265_hip(f->ctx, hipStreamSynchronize(f->side)) : ks265_hip(f->ctx, hipStreamWaitEvent(f->ctx->stream, f->ev_join, 0));
        if (r) return r;
        if (rw) return rw;
    } else {
    if ((r = chain(0))) return r;
    if ((r = chain(1))) return r;
    }
    /* several pictures per list: from here to the reconstruction every stage takes a block's pictures from its record */
    struct MrScope { ks265_frame *f; ~MrScope() { f->mrefb = false; } } scope{f};
    if (mr) {
        f->mrefb = true; f->mr_n[0] = n0; f->mr_n[1] = n1;
        for (int i = 0; i < 4; ++i) { f->mr_pic[0][i] = refs0[i < n0 ? i : n0 - 1]; f->mr_pic[1][i] = refs1[i < n1 ? i : n1 - 1]; }
    }
    if ((r = ks265_bi_decide(f, src, ref0, ref1, pu0, f->pu1, f->pub))) return r;
    const bool ii = f->cfg.intra_inter != 0;
    if (ii && (r = ks265_intra_candidates(f, src, f->pub, f->icost))) return r;
    if ((r = records_fence(f))) return r;
    const bool part = f->cfg.part != 0;                        /* -part 1: the halves of every 64 / 32 / 16 CU priced with the motions of the CU and of its quarters (round 5: B pictures too) */
    ks265_cu8 *cud = f->cfg.merge ? f->cu8_tmp : f->cu8;
    if ((r = part ? ks265_cu_decide_part_b(f, src, ref0, ref1, pu0, f->pu1, f->pub, ii ? f->icost : nullptr, cud) : ks265_cu_decide_b_ii(f, f->pub, ii ? f->icost : nullptr, cud))) return r;
    if (f->cfg.bi_refine == 2 && (r = ks265_bi_refine_chosen(f, src, ref0, ref1, pu0, f->pu1, f->pub, cud))) return r;   /* the joint refinement, for the CUs the decision chose (round 5) */
    if (f->cfg.merge && (r = ks265_merge_pass(f, src, ref0, ref1, nullptr, f->pub, f->cu8_tmp, f->cu8))) return r;
    const bool no_sao = f->cfg.sao == 0;                       /* without SAO the picture is reconstructed and deblocked where it is handed over: no SAO launch, no copy */
    ks265_pic deb = no_sao ? recon_out : ks_deb_pic(f);
    if ((r = ks265_reconstruct_b(f, src, ref0, ref1, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
    if (f->cfg.skip_rd && (r = ks265_skip_pass(f, src, ref0, ref1, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
    if (ii && (r = ks265_intra_inter_reconstruct(f, src, f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], deb))) return r;
    if (f->cfg.deblock && (r = ks265_deblock(f, f->cu8, deb))) return r;
    if (no_sao) return ks265_sao_off(f, f->sao, recon_out);
    return ks265_sao(f, src, deb, f->sao, recon_out);
}

/* forget the temporal predictors (the next P picture starts like the first one after a key picture): for hosts that code key pictures on another frame object */
int ks265_frame_reset_prediction(ks265_frame *f)
{
    KS_FRAME_CHECK(f);
    f->have_prev = false;
    return KS265_OK;
}

int ks265_frame_p_state(ks265_frame *f) { return f ? (f->cur_pu & 1) | (f->have_prev ? 2 : 0) : -1; }
int ks265_frame_p_restore(ks265_frame *f, int state)
{
    KS_FRAME_CHECK(f);
    if (state < 0 || state > 3) return KS265_NOTSUPPORTED;
    f->cur_pu = state & 1; f->have_prev = (state & 2) != 0;
    return KS265_OK;
}
int ks265_frame_p_advance(ks265_frame *f)
{
    KS_FRAME_CHECK(f);
    f->cur_pu ^= 1; f->have_prev = true;
    return KS265_OK;
}

int ks265_frame_set_profiling(ks265_frame *f, int enable)
{
    KS_FRAME_CHECK(f);
    if (enable && !f->ev[0])
        for (int i = 0; i <= KS_NSTAGE; ++i)
            if (hipEventCreate(&f->ev[i]) != hipSuccess) return KS265_FAIL;
    if (enable && !f->ev_k[0])
        for (int i = 0; i < 2; ++i)
            if (hipEventCreate(&f->ev_k[i]) != hipSuccess) return KS265_FAIL;
    f->profiling = enable != 0;
    return KS265_OK;
}

/* the duration of the last me_int_kernel launch alone (HIP events on the frame's stream around that one launch; -1 = none since profiling was switched on): the SAD kernel
 * of bench.py's roofline figure - stage 0 of ks265_frame_stage_ms also holds the pyramid pre-search and the propagation round */
int ks265_frame_me_int_ms(ks265_frame *f, float *ms)
{
    KS_FRAME_CHECK(f);
    if (!ms) return KS265_POINTER;
    if (!f->profiling) return KS265_NOTSUPPORTED;
    int r = ks265_hip(f->ctx, hipStreamSynchronize(f->ctx->stream));
    if (r) return r;
    *ms = -1.0f;
    if (f->ev_k_valid && hipEventElapsedTime(ms, f->ev_k[0], f->ev_k[1]) != hipSuccess) *ms = -1.0f;
    return KS265_OK;
}

/* elapsed milliseconds of the stages of the LAST ks265_encode_picture call (synchronises the stream); -1 = stage not run */
int ks265_frame_stage_ms(ks265_frame *f, float ms[8])
{
    KS_FRAME_CHECK(f);
    if (!ms) return KS265_POINTER;
    if (!f->profiling) return KS265_NOTSUPPORTED;
    int r = ks265_hip(f->ctx, hipStreamSynchronize(f->ctx->stream));
    if (r) return r;
    for (int s = 0; s < KS_NSTAGE; ++s) {
        ms[s] = -1.0f;
        if (f->ev_valid[s] && f->ev_valid[s + 1] && hipEventElapsedTime(&ms[s], f->ev[s], f->ev[s + 1]) != hipSuccess) ms[s] = -1.0f;
    }
    return KS265_OK;
}

int16_t *ks265_frame_levels(ks265_frame *f, int comp) { return f && comp >= 0 && comp < 3 ? f->lvl[comp] : nullptr; }
ks265_pu *ks265_frame_pu(ks265_frame *f) { return f ? f->pu[f->cur_pu ^ (f->have_prev ? 1 : 0)] : nullptr; }   /* records of the last coded P picture */
ks265_cu8 *ks265_frame_cu8(ks265_frame *f) { return f ? f->cu8 : nullptr; }
uint32_t *ks265_frame_ibest(ks265_frame *f) { return f ? f->icost : nullptr; }        /* cfg.intra_inter: the intra candidates of the last P / B picture (null without) */
ks265_sao_param *ks265_frame_sao(ks265_frame *f) { return f ? f->sao : nullptr; }

// ------------------------------------------------------------------ the records of one picture as ONE contiguous block (hosts: one copy-out per picture)
struct PackSegs { const uint8_t *src[6]; unsigned long long off[6], bytes[6]; };
__global__ __launch_bounds__(256) void pack_records_kernel(PackSegs sg, uint8_t *dst)
{
    const int seg = blockIdx.y;
    const uint8_t *s = sg.src[seg];
    if (!s) return;
    uint8_t *d = dst + sg.off[seg];
    const unsigned long long n16 = sg.bytes[seg] >> 4, n4 = (sg.bytes[seg] & 15) >> 2;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256)
        ((uint4 *)d)[i] = ((const uint4 *)s)[i];
    if (blockIdx.x == 0 && threadIdx.x < n4) ((unsigned *)(d + (n16 << 4)))[threadIdx.x] = ((const unsigned *)(s + (n16 << 4)))[threadIdx.x];
}
static void records_layout(const ks265_frame *f, size_t off[7])
{
    const size_t npx = (size_t)f->g.W * f->g.H;
    const size_t sz[6] = {(size_t)f->geom.bytes_cu8, npx * 2, npx / 2, npx / 2, (size_t)f->geom.bytes_sao, 64};
    size_t o = 0;
    for (int i = 0; i < 6; ++i) { off[i] = o; o += (sz[i] + 255) & ~(size_t)255; }
    off[6] = o;
}
int ks265_frame_records_layout(ks265_frame *f, size_t off[7])
{
    if (!f || !off) return KS265_POINTER;
    records_layout(f, off);
    return KS265_OK;
}
int ks265_frame_pack_records(ks265_frame *f, void *dev_dst, const void *dev_extra64)
{
    KS_FRAME_CHECK(f);
    if (!dev_dst) return KS265_POINTER;
    size_t off[7];
    records_layout(f, off);
    const size_t npx = (size_t)f->g.W * f->g.H;
    PackSegs sg;
    const void *src[6] = {f->cu8, f->lvl[0], f->lvl[1], f->lvl[2], f->sao, dev_extra64};
    const size_t sz[6] = {(size_t)f->geom.bytes_cu8, npx * 2, npx / 2, npx / 2, (size_t)f->geom.bytes_sao, 64};
    for (int i = 0; i < 6; ++i) { sg.src[i] = (const uint8_t *)src[i]; sg.off[i] = off[i]; sg.bytes[i] = sz[i]; }
    const unsigned gx = (unsigned)((npx * 2 / 16 + 256 * 8 - 1) / (256 * 8));               // the largest segment: 8 x 16 bytes per thread
    hipLaunchKernelGGL(pack_records_kernel, dim3(gx ? gx : 1, 6), dim3(256), 0, f->ctx->stream, sg, (uint8_t *)dev_dst);
    return ks265_check_launch(f->ctx);
}
// ------------------------------------------------------------------ compact records: the level planes without their all-zero 64-byte lines
// A P / B picture's levels are almost all zero; what leaves the GPU per picture is then ~2 MB instead of 26 MB (2160p).  The three level planes are cut
// into lines of 64 bytes (32 levels of one row); a line bitmap says which lines hold a level, those lines follow back to back.  A work-group handles a
// chunk of 1024 lines: ballots give the bitmap words and the line's rank inside the chunk, one atomic add reserves the chunk's room in the data area and
// the chunk table records where (chunks may land in any order; inside a chunk the lines keep their order).  The last work-group to finish publishes the
// total and clears the running counters for the next picture (no memset launch).
#define KS_CL_LINE 64
#define KS_CL_CHUNK 1024
struct CompactArgs { const uint8_t *plane[3]; unsigned long long plane_bytes[3]; unsigned first_line[4]; unsigned nchunk; };
__global__ __launch_bounds__(256) void pack_compact_kernel(CompactArgs a, unsigned *hdr /* run_lines, done, data_lines, nlines */, unsigned *table, unsigned long long *bitmap, uint4 *data)
{
    __shared__ unsigned cnt[16], base_sh;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned chunk = blockIdx.x, nlines = a.first_line[3];
    uint4 v[4][4];
    bool nz[4];
    unsigned rank_in_wave[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned L = chunk * KS_CL_CHUNK + p * 256 + t;
        nz[p] = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[p][q] = make_uint4(0, 0, 0, 0);
        if (L < nlines) {
            const int pl = L >= a.first_line[2] ? 2 : L >= a.first_line[1] ? 1 : 0;
            const unsigned long long off = (unsigned long long)(L - a.first_line[pl]) * KS_CL_LINE;
            const uint8_t *src = a.plane[pl] + off;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (off + 16 * q + 16 <= a.plane_bytes[pl]) v[p][q] = *(const uint4 *)(src + 16 * q);       // planes are multiples of 16 bytes long
            unsigned o = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) o |= v[p][q].x | v[p][q].y | v[p][q].z | v[p][q].w;
            nz[p] = o != 0;
        }
        const unsigned long long b = __ballot(nz[p]);
        rank_in_wave[p] = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) {
            cnt[p * 4 + wave] = (unsigned)__popcll(b);
            const unsigned word = chunk * (KS_CL_CHUNK / 64) + p * 4 + wave;
            if (word * 64 < nlines) bitmap[word] = b;
        }
    }
    __syncthreads();
    if (t == 0) {
        unsigned tot = 0;
        for (int i = 0; i < 16; ++i) { const unsigned c = cnt[i]; cnt[i] = tot; tot += c; }
        const unsigned base = tot ? atomicAdd(&hdr[0], tot) : 0u;
        table[chunk] = base;
        base_sh = base;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if (nz[p]) {
            uint4 *d = data + (unsigned long long)(base_sh + cnt[p * 4 + wave] + rank_in_wave[p]) * (KS_CL_LINE / 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = v[p][q];
        }
    __syncthreads();
    if (t == 0) {
        __threadfence();
        if (atomicAdd(&hdr[1], 1u) == a.nchunk - 1u) {
            __threadfence();
            hdr[2] = atomicExch(&hdr[0], 0u);
            hdr[3] = nlines;
            atomicExch(&hdr[1], 0u);
        }
    }
}
// copy-out of a compact block: the fixed part, then as many data lines as the header says (read on the device: the host does not know the size yet)
__global__ __launch_bounds__(256) void copy_out_compact_kernel(uint4 *dst, const uint4 *src, unsigned long long fixed16, const unsigned *hdr, unsigned long long first16)
{
    const unsigned long long n16 = fixed16 + (unsigned long long)hdr[2] * (KS_CL_LINE / 16);
    for (unsigned long long i = first16 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) dst[i] = src[i];
}
static void compact_layout(const ks265_frame *f, size_t off[8])
{
    const size_t npx = (size_t)f->g.W * f->g.H;
    const size_t pb[3] = {npx * 2, npx / 2, npx / 2};
    size_t nlines = 0;
    for (int i = 0; i < 3; ++i) nlines += (pb[i] + KS_CL_LINE - 1) / KS_CL_LINE;
    const size_t nchunk = (nlines + KS_CL_CHUNK - 1) / KS_CL_CHUNK;
    const size_t sz[7] = {(size_t)f->geom.bytes_cu8, (size_t)f->geom.bytes_sao, 64, 64, nchunk * 4, nchunk * (KS_CL_CHUNK / 8), nlines * KS_CL_LINE};
    size_t o = 0;
    for (int i = 0; i < 7; ++i) { off[i] = o; o += (sz[i] + 255) & ~(size_t)255; }
    off[7] = o;
}
int ks265_frame_compact_layout(ks265_frame *f, size_t off[8])
{
    if (!f || !off) return KS265_POINTER;
    compact_layout(f, off);
    return KS265_OK;
}
int ks265_frame_pack_compact(ks265_frame *f, void *dev_dst, const void *dev_extra64) { return f ? ks265_frame_pack_compact_on(f->ctx, f, dev_dst, dev_extra64) : KS265_POINTER; }
int ks265_frame_pack_compact_on(ks265_ctx *cx, ks265_frame *f, void *dev_dst, const void *dev_extra64)
{
    KS_FRAME_CHECK(f);
    if (!dev_dst || !cx) return KS265_POINTER;
    ks_use_device(cx);
    size_t off[8];
    compact_layout(f, off);
    const size_t npx = (size_t)f->g.W * f->g.H;
    PackSegs sg = {};
    const void *src[3] = {f->cu8, f->sao, dev_extra64};
    const size_t sz[3] = {(size_t)f->geom.bytes_cu8, (size_t)f->geom.bytes_sao, 64};
    for (int i = 0; i < 3; ++i) { sg.src[i] = (const uint8_t *)src[i]; sg.off[i] = off[i]; sg.bytes[i] = sz[i]; }
    const unsigned gx = (unsigned)((sz[0] / 16 + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(pack_records_kernel, dim3(gx ? gx : 1, 3), dim3(256), 0, cx->stream, sg, (uint8_t *)dev_dst);
    CompactArgs a;
    const size_t pb[3] = {npx * 2, npx / 2, npx / 2};
    unsigned fl = 0;
    for (int i = 0; i < 3; ++i) { a.plane[i] = (const uint8_t *)f->lvl[i]; a.plane_bytes[i] = pb[i]; a.first_line[i] = fl; fl += (unsigned)((pb[i] + KS_CL_LINE - 1) / KS_CL_LINE); }
    a.first_line[3] = fl; a.nchunk = (fl + KS_CL_CHUNK - 1) / KS_CL_CHUNK;
    uint8_t *d = (uint8_t *)dev_dst;
    hipLaunchKernelGGL(pack_compact_kernel, dim3(a.nchunk), dim3(256), 0, cx->stream, a, (unsigned *)(d + off[3]), (unsigned *)(d + off[4]), (unsigned long long *)(d + off[5]), (uint4 *)(d + off[6]));
    return ks265_check_launch(cx);
}
int ks265_copy_out_compact_async(ks265_ctx *ctx, ks265_frame *f, void *pinned_host, const void *dev_block)
{
    if (!ctx || !f || !pinned_host || !dev_block) return KS265_POINTER;
    ks_use_device(ctx);
    size_t off[8];
    compact_layout(f, off);
    hipLaunchKernelGGL(copy_out_compact_kernel, dim3(32), dim3(256), 0, ctx->stream, (uint4 *)pinned_host, (const uint4 *)dev_block, (unsigned long long)(off[6] / 16),
                       (const unsigned *)((const uint8_t *)dev_block + off[3]), 0ull);
    return ks265_check_launch(ctx);
}
/* the copy engine takes the fixed part and the first `data_bytes` of the data area (one hipMemcpyAsync: no kernel writes host memory - a kernel that does slowed the
 * kernels running beside it by ~30 us per picture at 2160p), a kernel only what lies beyond - normally nothing: it reads the header and leaves */
int ks265_copy_out_compact_dma_async(ks265_ctx *ctx, ks265_frame *f, void *pinned_host, const void *dev_block, size_t data_bytes)
{
    if (!ctx || !f || !pinned_host || !dev_block) return KS265_POINTER;
    ks_use_device(ctx);
    size_t off[8];
    compact_layout(f, off);
    data_bytes = (data_bytes + 255) & ~(size_t)255;
    if (data_bytes > off[7] - off[6]) data_bytes = off[7] - off[6];
    int r = ks265_hip(ctx, hipMemcpyAsync(pinned_host, dev_block, off[6] + data_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (r) return r;
    if (data_bytes < off[7] - off[6])
        hipLaunchKernelGGL(copy_out_compact_kernel, dim3(32), dim3(256), 0, ctx->stream, (uint4 *)pinned_host, (const uint4 *)dev_block, (unsigned long long)(off[6] / 16),
                           (const unsigned *)((const uint8_t *)dev_block + off[3]), (unsigned long long)((off[6] + data_bytes) / 16));
    return ks265_check_launch(ctx);
}

}  // extern "C"
