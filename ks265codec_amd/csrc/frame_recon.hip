// frame_recon.hip — Stage D: prediction -> residual -> forward transform -> quant -> dequant -> inverse transform -> recon
// (the reconstruct() chain enc@0x481da0: calc_residual -> g_H265_2dDct_Func -> g_QuantFuncs -> g_DeQuantFuncs ->
// g_H265_2dIDct_Func).  One workgroup per 32x32 luma region (+ its two 16x16 chroma regions); the region holds TUs of
// 8..32 (luma) / 4..16 (chroma) samples.  All four 1-D passes are row-times-row dot products out of LDS:
//   fwd 1  T[k][j] = M[k] . X[j]        fwd 2  C[k][j] = M[k] . T[j]      (H265_2dDct*_c enc@0x4c2210.., shifts 2log2N-2, 7)
//   inv 1  T[y][x] = Mt[y] . Ct[x]      inv 2  R[y][x] = T[y] . Mt[x]     (H265_2dIDct*_c enc@0x448f60.., shifts 7, 12)
// (Ct = dequantised coefficients stored transposed, Mt = transposed matrix), so every operand is read along a row with
// 8-byte LDS reads and multiplied with v_dot2c_i32_i16 (all intermediates fit int16 exactly as in the reference's
// `short` buffers; accumulation is int32).  Row pitch 36 shorts = 18 dwords makes the row-per-lane ds_read_b64 walks
// bank-conflict free.  Each thread produces 4 adjacent outputs that share one of the two operand rows.
#include "frame_common.h"
#include "recon_dev.h"
#include "pred_dev.h"

using namespace ks265;

// further list-0 reference pictures (multi-reference P pictures): this component's planes of pictures 1..3
struct KsCompRefs { const uint8_t *r[3]; const uint8_t *s[3]; };   // s: list 1's pictures 1 .. 3 (multi-reference B pictures, round 5)

template <int RS /*region size in samples: 32 luma, 16 chroma*/, bool MREF>
__device__ __forceinline__ void code_region(const KsGeom &g, int comp, int qp, int rx8, int ry8, ks265_cu8 *blk /*LDS [16]*/,
                                            unsigned char *tu_log2 /*LDS [16]: log2 of TU size in 8x8 blocks*/, const short *Mf, const short *Mt,
                                            short *X, short *T, unsigned char *P, int *nzcnt /*LDS [16]*/, const uint8_t *src, const uint8_t *ref,
                                            const uint8_t *ref1, int16_t *lvl, uint8_t *rec, int tid, const KsCompRefs xr,
                                            bool sdh, short *LV, short *DU, short *CF, int *lastcg /*LDS [16]*/, int dec_k, long long rdo_lam2k /* lambda_q4^2 x cfg.rdo, 0 = off */,
                                            bool tu_split = false /* cfg.tu_inter, the luma call: decide which 2Nx2N inter CUs of 32 / 16 carry four transform units */,
                                            int rq_mode = 0 /* cfg.rdoq, the luma call: 1 = front half (coefficients + levels rounded at 1 / 2 to the planes, nothing else), 2 = back half (levels from the plane) */,
                                            int16_t *coefp = nullptr)
{
    constexpr int UNIT = RS / 4;                      // samples per 8x8-luma block along one axis
    constexpr int NQ = RS * RS / 4;                   // quads (4 adjacent samples of one row)
    const long stride = comp == 0 ? g.sy : g.sc;
    const int lstride = comp == 0 ? g.W : g.W / 2;
    const int X0 = rx8 * UNIT, Y0 = ry8 * UNIT;       // region origin in this component's samples
    const uint8_t *S = comp == 0 ? ks_org_y(g, src) : ks_org_c(g, src);
    uint8_t *Rc = comp == 0 ? ks_org_y(g, rec) : ks_org_c(g, rec);
    const int qx = (tid % (RS / 4)) * 4, qy = tid / (RS / 4);      // this thread's quad: samples (qx..qx+3, qy)
    const bool has_quad = tid < NQ;
    const int b = has_quad ? (qy / UNIT) * 4 + qx / UNIT : 0;
    const ks265_cu8 c = blk[b];
    const bool coded = has_quad && c.log2_cu != 0 && c.pred_mode != 2;     // block inside the picture; an intra CU of a P / B picture (pred_mode 2) is coded afterwards (ks265_intra_inter_reconstruct)
    if (tid < 16) { nzcnt[tid] = 0; lastcg[tid] = 0; }
    if (tu_split) __syncthreads();                                 // (the residual stage adds into lastcg)
    // ---- prediction + residual
    if (has_quad) {
        int pred[4] = {128, 128, 128, 128}, res[4] = {0, 0, 0, 0};
        if (coded) {
            if (c.pred_mode == 0) {
                // ref / ref1 are the planes of THIS component (luma for comp 0: only used by bi-prediction)
                const int dir = c.inter_dir & 3;
                if (MREF) {                                         // list-0 picture of this CU: inter_dir >> 4 (selects, not an indexed array: keeps the global address space)
                    const int ri = (c.inter_dir >> 4) & 3, ri1 = (c.inter_dir >> 6) & 3;      // (B pictures: list 1's picture in bits 6 .. 7)
                    ref = ri == 0 ? ref : (ri == 1 ? xr.r[0] : (ri == 2 ? xr.r[1] : xr.r[2]));
                    ref1 = ri1 == 0 ? ref1 : (ri1 == 1 ? xr.s[0] : (ri1 == 2 ? xr.s[1] : xr.s[2]));
                }
                if (dir == 3) {
                    // bi-prediction: exact 14-bit average of the two lists (DefaultWeightedBi_c enc@0x435160)
                    int v0[4], v1[4];
                    const uint8_t *o0 = comp == 0 ? ks_org_y(g, ref) : ks_org_c(g, ref), *o1 = comp == 0 ? ks_org_y(g, ref1) : ks_org_c(g, ref1);
                    int k0, k1;
                    if (comp == 0) {
                        k0 = luma_raw4(o0 + (long)(Y0 + qy + (c.mvy >> 2)) * g.sy + X0 + qx + (c.mvx >> 2), g.sy, c.mvx & 3, c.mvy & 3, v0);
                        k1 = luma_raw4(o1 + (long)(Y0 + qy + (c.mv1y >> 2)) * g.sy + X0 + qx + (c.mv1x >> 2), g.sy, c.mv1x & 3, c.mv1y & 3, v1);
                    } else {
                        k0 = chroma_raw4(o0 + (long)(Y0 + qy + (c.mvy >> 3)) * g.sc + X0 + qx + (c.mvx >> 3), g.sc, c.mvx & 7, c.mvy & 7, v0);
                        k1 = chroma_raw4(o1 + (long)(Y0 + qy + (c.mv1y >> 3)) * g.sc + X0 + qx + (c.mv1x >> 3), g.sc, c.mv1x & 7, c.mv1y & 7, v1);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) pred[i] = clip8(((int)(short)to14(k0, v0[i]) + (int)(short)to14(k1, v1[i]) + 64) >> 7);
                } else {
                    const int ux = dir == 2 ? c.mv1x : c.mvx, uy = dir == 2 ? c.mv1y : c.mvy;
                    if (comp == 0) {
                        // luma: the normative 8-tap filters on the reference picture itself (round 2 read one of sixteen precomputed planes here)
                        const uint8_t *rp = ks_org_y(g, dir == 2 ? ref1 : ref) + (long)(Y0 + qy + (uy >> 2)) * g.sy + X0 + qx + (ux >> 2);
                        int v[4];
                        const int k = luma_raw4(rp, g.sy, ux & 3, uy & 3, v);
#pragma unroll
                        for (int i = 0; i < 4; ++i) pred[i] = uni_round(k, v[i]);
                    } else {
                        const uint8_t *rp = ks_org_c(g, dir == 2 ? ref1 : ref) + (long)(Y0 + qy + (uy >> 3)) * g.sc + X0 + qx + (ux >> 3);
                        int v[4];
                        const int k = chroma_raw4(rp, g.sc, ux & 7, uy & 7, v);
#pragma unroll
                        for (int i = 0; i < 4; ++i) pred[i] = uni_round(k, v[i]);
                    }
                }
            }
            const unsigned sv = *(const unsigned *)(S + (long)(Y0 + qy) * stride + X0 + qx);
#pragma unroll
            for (int i = 0; i < 4; ++i) res[i] = (int)((sv >> (8 * i)) & 255) - pred[i];
            if (tu_split && c.pred_mode == 0) atomicAdd(&lastcg[b], abs(res[0]) + abs(res[1]) + abs(res[2]) + abs(res[3]));      // residual SAD of the 8x8 block (lastcg is free until the sign-hiding phase)
        }
        *(unsigned *)(P + qy * RS + qx) = (unsigned)pred[0] | ((unsigned)pred[1] << 8) | ((unsigned)pred[2] << 16) | ((unsigned)pred[3] << 24);
        *(uint2 *)(X + qy * RP + qx) = make_uint2(((unsigned)res[0] & 0xFFFF) | ((unsigned)res[1] << 16), ((unsigned)res[2] & 0xFFFF) | ((unsigned)res[3] << 16));
    }
    __syncthreads();
    if (tu_split) {
        // cfg.tu_inter (-intertu 1; the oracle's rule in reconstruct_impl): a 2Nx2N inter CU of 32 / 16 samples gets four transform units when its luma residual is concentrated in
        // part of it - the quarters' residual SADs: max > 4 x min + (N / 2)^2.  The thread of the CU's first block decides; log2_cu bits 4..5 = 3 mark the CU for every later reader
        if (tid < 16) {
            const ks265_cu8 q = blk[tid];
            const int l2 = q.log2_cu & 15, n8 = l2 >= 3 ? 1 << (l2 - 3) : 1, bx = tid & 3, by = tid >> 2;
            if (q.log2_cu != 0 && q.pred_mode == 0 && !(q.log2_cu >> 4) && (n8 == 2 || n8 == 4) && !(bx & (n8 - 1)) && !(by & (n8 - 1))) {
                const int h = n8 >> 1, N = n8 * 8;
                int mn = 0x7fffffff, mx = 0;
                for (int k = 0; k < 4; ++k) {
                    int sq = 0;
                    for (int yy = 0; yy < h; ++yy)
                        for (int xx = 0; xx < h; ++xx) sq += lastcg[(by + (k >> 1) * h + yy) * 4 + bx + (k & 1) * h + xx];
                    mn = min(mn, sq); mx = max(mx, sq);
                }
                if (mx > 4 * mn + (N / 2) * (N / 2))
                    for (int yy = 0; yy < n8; ++yy)
                        for (int xx = 0; xx < n8; ++xx) { const int o = (by + yy) * 4 + bx + xx; blk[o].log2_cu = (uint8_t)(q.log2_cu | 0x30); tu_log2[o] = (unsigned char)(l2 - 4); }
            }
        }
        __syncthreads();
        if (tid < 16) lastcg[tid] = 0;
        __syncthreads();
    }
    // TU geometry of the quad
    const int t8 = 1 << tu_log2[b], tbx = (b & 3) & ~(t8 - 1), tby = (b >> 2) & ~(t8 - 1), tb = tby * 4 + tbx;
    const int ox = tbx * UNIT, oy = tby * UNIT, n = t8 * UNIT, log2n = tu_log2[b] + (RS == 32 ? 3 : 2);
    const short *mf = Mf + mat_off(log2n), *mt = Mt + mat_off(log2n);
    const int mp = n + 4;                                          // matrix row pitch
    if (rq_mode == 2) {
        // ---- cfg.rdoq, back half: the levels rdoQuant left in the plane -> dequantised (transposed) tile, coded-block counts; then the inverse passes below
        if (has_quad && coded) {
            const int k = qy - oy, j = qx - ox, shift = log2n - 1, dqs = kInvQuantScales[qp % 6] << (qp / 6);
            const uint2 lw = *(const uint2 *)(lvl + (long)(Y0 + qy) * lstride + X0 + qx);
            int nz = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w = i < 2 ? lw.x : lw.y;
                const int l = (int)(short)((i & 1) ? (w >> 16) : (w & 0xFFFFu));
                nz += l != 0;
                X[(oy + j + i) * RP + ox + k] = (short)dequant_one(l, dqs, 1 << (shift - 1), shift);
            }
            if (nz) atomicAdd(&nzcnt[tb], nz);
        }
        __syncthreads();
    } else {
    // ---- forward pass 1: T[k][j] = rnd(M[k] . X[j], 2 log2N - 2)
    if (has_quad) {
        const int k = qy - oy, j = qx - ox, s1 = 2 * log2n - 2;
        int acc[4];
        quad_dot(mf + k * mp, X + (oy + j) * RP + ox, RP, n, acc);
        unsigned short o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned short)(short)((acc[i] + (1 << (s1 - 1))) >> s1);
        *(uint2 *)(T + qy * RP + qx) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
    __syncthreads();
    // ---- forward pass 2 + quant + dequant; coefficient (k, j) belongs to element (oy + k, ox + j); the dequantised
    //      value is stored transposed, at (oy + j, ox + k), for the inverse passes
    if (has_quad) {
        const int k = qy - oy, j = qx - ox;
        int acc[4];
        quad_dot(mf + k * mp, T + (oy + j) * RP + ox, RP, n, acc);
        const int qp6 = qp / 6, scale = kQuantScales[qp % 6], dqs = kInvQuantScales[qp % 6] << qp6;
        const int qbits = 21 + qp6 - log2n, off = (c.pred_mode != 0 ? 171 : 85) << (qbits - 9), shift = log2n - 1;
        unsigned short lv[4];
        int nz = 0;
        if (rq_mode == 1) {
            // ---- cfg.rdoq, front half: the transform coefficients and the quantiser's levels rounded at 1 / 2 (what rdoQuant enc@0x4aac50 is handed) go to their planes
            unsigned short cf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int coef = (short)((acc[i] + 64) >> 7), a = coef < 0 ? -coef : coef;
                int q = coded ? (int)(((long long)a * scale + (1ll << (qbits - 1))) >> qbits) : 0;
                if (q > 32767) q = 32767;
                lv[i] = (unsigned short)(short)(coef < 0 ? -q : q); cf[i] = (unsigned short)(short)(coded ? coef : 0);
            }
            if (coded) {
                *(uint2 *)(lvl + (long)(Y0 + qy) * lstride + X0 + qx) = make_uint2(lv[0] | ((unsigned)lv[1] << 16), lv[2] | ((unsigned)lv[3] << 16));
                *(uint2 *)(coefp + (long)(Y0 + qy) * lstride + X0 + qx) = make_uint2(cf[0] | ((unsigned)cf[1] << 16), cf[2] | ((unsigned)cf[3] << 16));
            }
        } else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int coef = (short)((acc[i] + 64) >> 7);
            int l = 0, dqv = 0, du = 0;
            if (coded) {
                l = quant_one(coef, scale, off, qbits, du);
                dqv = dequant_one(l, dqs, 1 << (shift - 1), shift);
                nz += l != 0;
            }
            lv[i] = (unsigned short)(short)l;
            if (sdh || rdo_lam2k) { const int o = (oy + k) * RP + ox + j + i; LV[o] = (short)l; DU[o] = (short)du; CF[o] = (short)coef; }
            X[(oy + j + i) * RP + ox + k] = (short)dqv;
        }
        if (coded && rq_mode != 1) {
            *(uint2 *)(lvl + (long)(Y0 + qy) * lstride + X0 + qx) = make_uint2(lv[0] | ((unsigned)lv[1] << 16), lv[2] | ((unsigned)lv[3] << 16));
            if (nz) {
                atomicAdd(&nzcnt[tb], nz);
                if (dec_k) {                                       // largest magnitude of the TU (lastcg is free until the sign-data hiding phase)
                    int mx = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) mx = max(mx, abs((int)(short)lv[i]));
                    atomicMax(&lastcg[tb], mx);
                }
            }
        }
    }
    __syncthreads();
    if (rq_mode == 1) return;                                      // (uniform: the front half ends here - no levels counted, no reconstruction)
    if (dec_k) {
        // ---- cfg.decimate (luma only: the chroma calls pass 0): a TU holding nothing but a few +-1 levels is dropped (levels zeroed in HBM, no residual, cbf 0)
        if (tid < 16) {
            const int d8 = 1 << tu_log2[tid], dtb = ((tid >> 2) & ~(d8 - 1)) * 4 + ((tid & 3) & ~(d8 - 1));
            const int dl2 = tu_log2[tid] + (RS == 32 ? 3 : 2);
            if (dtb == tid && blk[tid].log2_cu != 0 && blk[tid].pred_mode == 0 && nzcnt[tid] > 0 && lastcg[tid] <= 1 && nzcnt[tid] <= dec_k * (dl2 - 1)) nzcnt[tid] = -1;
        }
        __syncthreads();
        if (coded && nzcnt[tb] < 0) *(uint2 *)(lvl + (long)(Y0 + qy) * lstride + X0 + qx) = make_uint2(0u, 0u);
        __syncthreads();
        if (tid < 16) { if (nzcnt[tid] < 0) nzcnt[tid] = 0; lastcg[tid] = 0; }
        __syncthreads();
    }
    if (rdo_lam2k) {
        // ---- cfg.rdo: coefficient-group pruning (recon_dev.h), one lane per 4x4 coefficient group of the region; inter blocks only
        constexpr int NCG = RS / 4;
        if (tid < NCG * NCG) {
            const int gx = tid % NCG, gy = tid / NCG, gb = (gy * 4 / UNIT) * 4 + gx * 4 / UNIT;
            const int g8 = 1 << tu_log2[gb], gtb = ((gb >> 2) & ~(g8 - 1)) * 4 + ((gb & 3) & ~(g8 - 1));
            const int gox = (gtb & 3) * UNIT, goy = (gtb >> 2) * UNIT, r0 = gy * 4, c0 = gx * 4, cbase = r0 * RP + c0;
            if (blk[gb].log2_cu != 0 && blk[gb].pred_mode == 0 && nzcnt[gtb] > 0) {
                const int l2n = 31 - __clz(g8 * UNIT);
                const int cnt = rdo_group_prune(LV, CF, cbase, kInvQuantScales[qp % 6] << (qp / 6), l2n, rdo_lam2k);
                if (cnt) {
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        *(uint2 *)(LV + cbase + y * RP) = make_uint2(0u, 0u);
                        *(uint2 *)(lvl + (long)(Y0 + r0 + y) * lstride + X0 + c0) = make_uint2(0u, 0u);
                        *(uint2 *)(X + (goy + (c0 + y - gox)) * RP + gox + (r0 - goy)) = make_uint2(0u, 0u);      // the dequantised tile is stored transposed
                    }
                    atomicSub(&nzcnt[gtb], cnt);
                }
            }
        }
        __syncthreads();
    }
    if (sdh) {
        // ---- the postQuant seam (postQuant enc@0x4ace80): sign-data hiding, one lane per 4x4 coefficient group of the region
        constexpr int NCG = RS / 4;                                   // groups per region row
        unsigned survey = 0;
        int cbase = 0, gtb = 0, gorder = 0, gn = 0, goy = 0, gox = 0;
        SbhRegs sr;
        const bool owner = tid < NCG * NCG;
        if (owner) {
            const int gx = tid % NCG, gy = tid / NCG;                // group coordinates in the region
            const int gb = (gy * 4 / UNIT) * 4 + gx * 4 / UNIT;      // its 8x8-luma block
            const int g8 = 1 << tu_log2[gb], gtbx = (gb & 3) & ~(g8 - 1), gtby = (gb >> 2) & ~(g8 - 1);
            gtb = gtby * 4 + gtbx;
            gn = g8 * UNIT; goy = gtby * UNIT; gox = gtbx * UNIT;
            cbase = gy * 4 * RP + gx * 4;
            if (blk[gb].log2_cu != 0 && nzcnt[gtb] > 1) {
                sbh_load(LV, DU, CF, cbase, sr);
                survey = sbh_survey_r<0>(sr);
                gorder = sbh_group_order(0, gn / 4, gx - gtbx * UNIT / 4, gy - gtby * UNIT / 4) + 1;
                if (survey >> 17) atomicMax(&lastcg[gtb], gorder);
            }
        }
        __syncthreads();
        if (owner && survey) {
            // the one level that moves: patched where the quantiser left it - the level plane in HBM and the dequantised (transposed) tile
            int nl = 0;
            const int pos = sbh_apply_r<0>(sr, survey, lastcg[gtb] == gorder, nl);
            if (pos >= 0) {
                const int row = cbase / RP + (pos >> 2), col = cbase % RP + (pos & 3), l2n = 31 - __clz(gn), shift = l2n - 1;
                const int dqs = kInvQuantScales[qp % 6] << (qp / 6);
                lvl[(long)(Y0 + row) * lstride + X0 + col] = (int16_t)nl;
                X[(goy + (col - gox)) * RP + gox + (row - goy)] = (short)dequant_one(nl, dqs, 1 << (shift - 1), shift);
            }
        }
        __syncthreads();
    }
    }   // rq_mode != 2
    const bool live = has_quad && nzcnt[tb] != 0;
    // ---- inverse pass 1: T[y][x] = clip16((Mt[y] . Ct[x] + 64) >> 7)
    if (has_quad) {
        const int y = qy - oy, x = qx - ox;
        int acc[4] = {0, 0, 0, 0};
        if (live) quad_dot(mt + y * mp, X + (oy + x) * RP + ox, RP, n, acc);
        unsigned short o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned short)(short)clip16((acc[i] + 64) >> 7);
        *(uint2 *)(T + qy * RP + qx) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
    __syncthreads();
    // ---- inverse pass 2 + prediction add: R[y][x] = (T[y] . Mt[x] + 2048) >> 12
    if (coded) {
        const int x = qx - ox;
        int acc[4] = {0, 0, 0, 0};
        if (live) quad_dot(T + qy * RP + ox, mt + x * mp, mp, n, acc);
        const unsigned pv = *(const unsigned *)(P + qy * RS + qx);
        unsigned o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = live ? (acc[i] + 2048) >> 12 : 0;
            o |= (unsigned)clip8((int)((pv >> (8 * i)) & 255) + r) << (8 * i);
        }
        *(unsigned *)(Rc + (long)(Y0 + qy) * stride + X0 + qx) = o;
    }
    __syncthreads();
}

// the forward and transposed DCT matrices of all four sizes, LDS layout (recon_dev.h), built once per frame object
__global__ __launch_bounds__(256) void build_matrices_kernel(short *mats) { build_matrices(mats, mats + MAT_SHORTS, threadIdx.x, 256); }
int ks265_frame_build_matrices(ks265_frame *f)
{
    hipLaunchKernelGGL(build_matrices_kernel, dim3(1), dim3(256), 0, f->ctx->stream, f->mats);
    return ks265_check_launch(f->ctx);
}

struct KsRefExtra { KsCompRefs y, u, v; };
static inline long long ks_rdo_lam2k(const ks265_frame *f) { return (long long)f->cfg.lambda_q4 * f->cfg.lambda_q4 * (f->cfg.rdo > 0 ? f->cfg.rdo : 0); }

// list-1 pointers are null for I / P pictures (no block carries inter_dir 2 or 3 there)
template <bool MREF>
__global__ __launch_bounds__(256) void reconstruct_kernel(KsGeom g, int qp, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v,
                                                          const uint8_t *ref_y, const uint8_t *ref_u, const uint8_t *ref_v,
                                                          const uint8_t *ref1_y, const uint8_t *ref1_u, const uint8_t *ref1_v, ks265_cu8 *cu8,
                                                          int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, const short *mats, const KsRefExtra xr, int sdh_on, int dec_k, long long rdo_lam2k, const int8_t *qp_map,
                                                          int tu_inter, int rq_mode, int16_t *coef_y)
{
    __shared__ __attribute__((aligned(16))) short Mf[MAT_SHORTS];
    __shared__ __attribute__((aligned(16))) short Mt[MAT_SHORTS];
    __shared__ __attribute__((aligned(16))) short X[32 * RP];
    __shared__ __attribute__((aligned(16))) short T[32 * RP];
    __shared__ __attribute__((aligned(16))) unsigned char P[32 * 32];
    __shared__ ks265_cu8 blk[16];
    __shared__ unsigned char tu_log2[16];
    __shared__ int nzcnt[16];
    __shared__ int cbf[16];
    __shared__ int lastcg[16];
    __shared__ __attribute__((aligned(16))) short LV[32 * RP], DU[32 * RP], CF[32 * RP];      // levels / remainders / coefficients of the region (sign-data hiding)
    const int tid = threadIdx.x;
    const bool sdh = sdh_on != 0;
    // XCD-aware mapping (T1): each XCD gets a contiguous raster range of regions, so the four regions that share a 128-byte
    // line of the source / prediction / level rows meet in one L2 instead of four (measured: FETCH_SIZE 3.9x the algorithmic reads)
    const int nrx = (g.W + 31) / 32, nreg = nrx * ((g.H + 31) / 32);
    const int reg = ks_xcd_swizzle(blockIdx.x, nreg), rx = reg % nrx, ry = reg / nrx;
    if (qp_map) qp = qp_map[(ry >> 1) * g.ctu_cols + (rx >> 1)];        // a 32x32 region lies in one CTU: its QP (cu_qp_delta, quantisation group = CTU)
    // transform matrices: built once per frame object (ks265_frame_create), 6.4 KB copied from L2 with 16-byte loads
    for (int i = tid; i < MAT_SHORTS / 8; i += 256) {
        ((uint4 *)Mf)[i] = ((const uint4 *)mats)[i];
        ((uint4 *)Mt)[i] = ((const uint4 *)(mats + MAT_SHORTS))[i];
    }
    if (tid < 16) {
        const int bx = rx * 4 + (tid & 3), by = ry * 4 + (tid >> 2);
        ks265_cu8 c;
        c.mvx = 0; c.mvy = 0; c.mv1x = 0; c.mv1y = 0; c.log2_cu = 0; c.cbf = 0; c.pred_mode = 0; c.inter_dir = 0;
        if (bx < g.w8 && by < g.h8) c = cu8[(long)by * g.w8 + bx];
        blk[tid] = c;
        tu_log2[tid] = (unsigned char)(c.log2_cu ? min((int)(c.log2_cu & 15) - 3 - (c.log2_cu >> 4 ? 1 : 0), 2) : 0);     // TU = min(CU, 32); a CU in two partitions (cfg.part): four TUs (interSplitFlag)
        cbf[tid] = 0;
    }
    __syncthreads();
    const int qpc = chroma_qp(qp);
    code_region<32, MREF>(g, 0, qp, rx * 4, ry * 4, blk, tu_log2, Mf, Mt, X, T, P, nzcnt, src_y, ref_y, ref1_y, lvl_y, rec_y, tid, xr.y, sdh, LV, DU, CF, lastcg, dec_k, rdo_lam2k, tu_inter != 0 && rq_mode != 2, rq_mode, coef_y);
    if (tid < 16 && blk[tid].log2_cu) {
        const int t8 = 1 << tu_log2[tid], tb = ((tid >> 2) & ~(t8 - 1)) * 4 + ((tid & 3) & ~(t8 - 1));
        if (nzcnt[tb]) cbf[tid] |= 1;
    }
    if (rq_mode == 2) {                                            // cfg.rdoq, back half: luma only - the chroma bits of the front half stay
        if (tid < 16 && blk[tid].log2_cu && blk[tid].pred_mode != 2) {
            const int bx = rx * 4 + (tid & 3), by = ry * 4 + (tid >> 2);
            cu8[(long)by * g.w8 + bx].cbf = (uint8_t)((blk[tid].cbf & 6) | (cbf[tid] & 1));
        }
        return;
    }
    __syncthreads();
    code_region<16, MREF>(g, 1, qpc, rx * 4, ry * 4, blk, tu_log2, Mf, Mt, X, T, P, nzcnt, src_u, ref_u, ref1_u, lvl_u, rec_u, tid, xr.u, sdh, LV, DU, CF, lastcg, 0, 0ll);      // chroma: no decimation, no group pruning (2 % of the bytes for 2 dB of chroma PSNR: the oracle's code_tu)
    if (tid < 16 && blk[tid].log2_cu) {
        const int t8 = 1 << tu_log2[tid], tb = ((tid >> 2) & ~(t8 - 1)) * 4 + ((tid & 3) & ~(t8 - 1));
        if (nzcnt[tb]) cbf[tid] |= 2;
    }
    __syncthreads();
    code_region<16, MREF>(g, 2, qpc, rx * 4, ry * 4, blk, tu_log2, Mf, Mt, X, T, P, nzcnt, src_v, ref_v, ref1_v, lvl_v, rec_v, tid, xr.v, sdh, LV, DU, CF, lastcg, 0, 0ll);
    if (tid < 16 && blk[tid].log2_cu) {
        const int t8 = 1 << tu_log2[tid], tb = ((tid >> 2) & ~(t8 - 1)) * 4 + ((tid & 3) & ~(t8 - 1));
        if (nzcnt[tb]) cbf[tid] |= 4;
        const int bx = rx * 4 + (tid & 3), by = ry * 4 + (tid >> 2);
        cu8[(long)by * g.w8 + bx].cbf = (uint8_t)cbf[tid];
        if (tu_inter) cu8[(long)by * g.w8 + bx].log2_cu = blk[tid].log2_cu;      // (bits 4..5 = 3: four transform units)
    }
}

// ------------------------------------------------------------------ cfg.rdoq: the reference's rdoQuant between the two halves of the reconstruction
int ks265_rdoq_listed(ks265_ctx *ctx, const ks265_rdoq_tu *dev_tus, const int *dev_n, int max_n, int16_t *dev_lvl, const int16_t *dev_coef, const int32_t *dev_tables, uint16_t *dev_sigmask,
                      int32_t *dev_out, uint64_t *dev_hidden);      // rdoq_ops.hip
// One wave per 8x8 block position; the wave of a luma transform block's first block (inter CU: TU = min(CU, 32), four TUs for a CU in two partitions / with split transform units)
// packs the block's levels and coefficients (planes, row pitch W) into contiguous N x N arrays, builds what rdoQuant is handed - per 4x4 group in scan order the mask of the
// positions the quantiser left non-zero (bit 15 - k = scan position k: scanSigFlags enc@0x4a9b00 lineage) and the last significant scan position -
// and appends the descriptor.  A block without a level is not listed (its plane is zero already).  ctr[0] = blocks listed, ctr[1] = packed elements.
__global__ __launch_bounds__(256) void rdoq_prep_kernel(KsGeom g, int qp, const int8_t *qp_map, int sdh, const ks265_cu8 *cu8, const int16_t *lvl, const int16_t *coef, int16_t *pack_lvl,
                                                        int16_t *pack_coef, ks265_rdoq_tu *tus, int *pos, int *ctr, unsigned short *sigmask, const long long *lam)
{
    __shared__ unsigned mask_s[4][64];
    __shared__ int last_s[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + wave;
    if (b >= g.w8 * g.h8) return;                                   // (whole waves leave: no work-group barrier below)
    const int bx = b % g.w8, by = b / g.w8;
    const ks265_cu8 c = cu8[b];
    if (c.log2_cu == 0 || c.pred_mode == 2) return;
    const int tl = min((int)(c.log2_cu & 15) - 3 - (c.log2_cu >> 4 ? 1 : 0), 2), t8 = 1 << tl;
    if ((bx & (t8 - 1)) || (by & (t8 - 1))) return;
    const int n = 8 * t8, log2n = 3 + tl, w = n >> 2, NN = n * n, x0 = bx * 8, y0 = by * 8;
    mask_s[wave][lane] = 0;
    if (lane == 0) last_s[wave] = -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // lane = (row chunk): every lane walks its share of the NN elements in rows of four
    for (int e = lane * 4; e < NN; e += 256) {
        const int y = e / n, x = e - y * n;
        const uint2 lw = *(const uint2 *)(lvl + (long)(y0 + y) * g.W + x0 + x);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned wd = i < 2 ? lw.x : lw.y;
            const int l = (int)(short)((i & 1) ? (wd >> 16) : (wd & 0xFFFFu));
            if (l) {
                const int xx = x + i, gidx = sbh_group_order(0, w, xx >> 2, y >> 2);
                const int k = (int)((0xfda6eb73c8419520ull >> (4 * ((y & 3) * 4 + (xx & 3)))) & 15ull);        // scan position of (x, y) inside its group (up-right diagonal)
                atomicOr(&mask_s[wave][gidx], 1u << (15 - k));
                atomicMax(&last_s[wave], gidx * 16 + k);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int last = last_s[wave];
    if (last < 0) return;
    int ti = 0, off = 0;
    if (lane == 0) { ti = atomicAdd(&ctr[0], 1); off = atomicAdd(&ctr[1], NN); }
    ti = __builtin_amdgcn_readfirstlane(ti); off = __builtin_amdgcn_readfirstlane(off);
    for (int e = lane * 4; e < NN; e += 256) {
        const int y = e / n, x = e - y * n;
        *(uint2 *)(pack_lvl + off + e) = *(const uint2 *)(lvl + (long)(y0 + y) * g.W + x0 + x);
        *(uint2 *)(pack_coef + off + e) = *(const uint2 *)(coef + (long)(y0 + y) * g.W + x0 + x);
    }
    sigmask[(long)ti * 64 + lane] = (unsigned short)mask_s[wave][lane];
    if (lane == 0) {
        const int q = qp_map ? qp_map[(by >> 3) * g.ctu_cols + (bx >> 3)] : qp;
        ks265_rdoq_tu d;
        d.off = off; d.tab = (log2n - 2) * 2; d.dq = kInvQuantScales[q % 6] << (q / 6); d.last_pos = last; d.lam = lam[q]; d.lam_sdh = lam[52 + q];
        d.log2 = (int8_t)log2n; d.scan_idx = 0; d.comp = 0; d.per = (int8_t)(q / 6); d.tu5 = 1; d.flag_a4c0 = 1; d.sdh = (int8_t)(sdh != 0); d.rsv = 0;
        tus[ti] = d;
        pos[2 * ti] = x0; pos[2 * ti + 1] = y0;
    }
}
// the levels rdoQuant decided, back to the level plane (one wave per listed block)
__global__ __launch_bounds__(256) void rdoq_unpack_kernel(KsGeom g, const ks265_rdoq_tu *tus, const int *pos, const int *ctr, const int16_t *pack_lvl, int16_t *lvl)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ti = blockIdx.x * 4 + wave;
    if (ti >= ctr[0]) return;
    const ks265_rdoq_tu d = tus[ti];
    const int n = 1 << d.log2, NN = n * n, x0 = pos[2 * ti], y0 = pos[2 * ti + 1];
    for (int e = lane * 4; e < NN; e += 256) {
        const int y = e / n, x = e - y * n;
        *(uint2 *)(lvl + (long)(y0 + y) * g.W + x0 + x) = *(const uint2 *)(pack_lvl + d.off + e);
    }
}

/* cfg.rdoq: the workspace (first call) and, per picture, the bit tables + lambdas the host built: tables = [4 sizes][luma, chroma][180] words of estBitRdoq enc@0x46a8a0 (the
 * host keeps them adaptive: from the context states the slice writer ended the previous picture of the kind with), lam / lam_sdh = [52] by QP (rdoQuant's two lambdas) */
extern "C" int ks265_frame_set_rdoq(ks265_frame *f, const int32_t *host_tables, const int64_t *host_lam, const int64_t *host_lam_sdh)
{
    KS_FRAME_CHECK(f);
    if (!host_tables || !host_lam || !host_lam_sdh) { f->rq_ready = false; return host_tables || host_lam || host_lam_sdh ? KS265_POINTER : KS265_OK; }     // all null: back to the default seam
    if (f->ctx->capturing) return KS265_NOTSUPPORTED;
    const size_t npx = (size_t)f->g.W * f->g.H, nb = (size_t)f->g.w8 * f->g.h8;
    if (!f->rq_coef) {
        hipError_t e = hipSuccess;
        auto al = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
        al((void **)&f->rq_coef, npx * 2); al((void **)&f->rq_pack_lvl, npx * 2); al((void **)&f->rq_pack_coef, npx * 2); al((void **)&f->rq_tus, nb * sizeof(ks265_rdoq_tu));
        al((void **)&f->rq_pos, nb * 8); al((void **)&f->rq_ctr, 16); al((void **)&f->rq_out, nb * 8); al((void **)&f->rq_tab, 8 * 180 * 4); al((void **)&f->rq_lam, 104 * 8);
        al((void **)&f->rq_sigmask, nb * 64 * 2); al((void **)&f->rq_hidden, nb * 8);
        if (e != hipSuccess) return ks265_hip(f->ctx, e);
    }
    int r = ks265_hip(f->ctx, hipMemcpyAsync(f->rq_tab, host_tables, 8 * 180 * 4, hipMemcpyHostToDevice, f->ctx->stream));
    if (!r) r = ks265_hip(f->ctx, hipMemcpyAsync(f->rq_lam, host_lam, 52 * 8, hipMemcpyHostToDevice, f->ctx->stream));
    if (!r) r = ks265_hip(f->ctx, hipMemcpyAsync(f->rq_lam + 52, host_lam_sdh, 52 * 8, hipMemcpyHostToDevice, f->ctx->stream));
    f->rq_ready = !r;
    return r;
}

// every reconstruction entry point ends here: `launch(rq_mode)` enqueues reconstruct_kernel in that mode
template <typename L>
static int reconstruct_sequence(ks265_frame *f, ks265_cu8 *cu8, int16_t *lvl_y, L launch)
{
    if (!f->rq_ready) { launch(0); return ks265_check_launch(f->ctx); }
    int r = ks265_hip(f->ctx, hipMemsetAsync(f->rq_ctr, 0, 16, f->ctx->stream));
    if (r) return r;
    launch(1);
    const int nb = f->g.w8 * f->g.h8;
    hipLaunchKernelGGL(rdoq_prep_kernel, dim3((nb + 3) / 4), dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, f->qp_map, f->cfg.sdh, cu8, lvl_y, f->rq_coef, f->rq_pack_lvl, f->rq_pack_coef,
                       f->rq_tus, f->rq_pos, f->rq_ctr, f->rq_sigmask, f->rq_lam);
    if ((r = ks265_check_launch(f->ctx))) return r;
    if ((r = ks265_rdoq_listed(f->ctx, f->rq_tus, f->rq_ctr, nb, f->rq_pack_lvl, f->rq_pack_coef, f->rq_tab, f->rq_sigmask, f->rq_out, (uint64_t *)f->rq_hidden))) return r;
    hipLaunchKernelGGL(rdoq_unpack_kernel, dim3((nb + 3) / 4), dim3(256), 0, f->ctx->stream, f->g, f->rq_tus, f->rq_pos, f->rq_ctr, f->rq_pack_lvl, lvl_y);
    launch(2);
    return ks265_check_launch(f->ctx);
}

static int launch_reconstruct(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1, ks265_cu8 *cu8,
                              int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, ks265_pic recon)
{
    dim3 grid(((f->g.W + 31) / 32) * ((f->g.H + 31) / 32));
    return reconstruct_sequence(f, cu8, lvl_y, [&](int rq_mode) {
    hipLaunchKernelGGL(reconstruct_kernel<false>, grid, dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, src.y, src.u, src.v, ref0.y, ref0.u, ref0.v, ref1.y,
                       ref1.u, ref1.v, cu8, lvl_y, lvl_u, lvl_v, recon.y, recon.u, recon.v, f->mats, KsRefExtra{}, f->cfg.sdh, f->cfg.decimate, ks_rdo_lam2k(f), f->qp_map, f->cfg.tu_inter, rq_mode, f->rq_coef);
    });
}

// multi-reference P pictures (-ref / -ref0): list 0 holds nref <= 4 pictures, the CU's picture is refs[inter_dir >> 4]
extern "C" int ks265_reconstruct_mref(ks265_frame *f, ks265_pic src, int nref, const ks265_pic *refs, ks265_cu8 *cu8, int16_t *lvl_y,
                                      int16_t *lvl_u, int16_t *lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !refs || !cu8 || !lvl_y || !lvl_u || !lvl_v || !recon.y) return KS265_POINTER;
    if (nref < 1 || nref > 4) return KS265_NOTSUPPORTED;
    KsRefExtra xr{};
    for (int r = 1; r < nref; ++r) {
        if (!refs[r].y) return KS265_POINTER;
        xr.y.r[r - 1] = refs[r].y; xr.u.r[r - 1] = refs[r].u; xr.v.r[r - 1] = refs[r].v;
    }
    dim3 grid(((f->g.W + 31) / 32) * ((f->g.H + 31) / 32));
    return reconstruct_sequence(f, cu8, lvl_y, [&](int rq_mode) {
    hipLaunchKernelGGL(reconstruct_kernel<true>, grid, dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, src.y, src.u, src.v, refs[0].y, refs[0].u, refs[0].v,
                       (const uint8_t *)nullptr, (const uint8_t *)nullptr, (const uint8_t *)nullptr, cu8, lvl_y, lvl_u, lvl_v, recon.y, recon.u,
                       recon.v, f->mats, xr, f->cfg.sdh, f->cfg.decimate, ks_rdo_lam2k(f), f->qp_map, f->cfg.tu_inter, rq_mode, f->rq_coef);
    });
}

extern "C" int ks265_reconstruct(ks265_frame *f, ks265_pic src, ks265_pic ref, ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u,
                                 int16_t *lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !cu8 || !lvl_y || !lvl_u || !lvl_v || !recon.y) return KS265_POINTER;
    return launch_reconstruct(f, src, ref, ks265_pic{nullptr, nullptr, nullptr}, cu8, lvl_y, lvl_u, lvl_v, recon);
}

extern "C" int ks265_reconstruct_b(ks265_frame *f, ks265_pic src, ks265_pic ref0, ks265_pic ref1,
                                   ks265_cu8 *cu8, int16_t *lvl_y, int16_t *lvl_u, int16_t *lvl_v, ks265_pic recon)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref0.y || !ref1.y || !cu8 || !lvl_y || !lvl_u || !lvl_v || !recon.y) return KS265_POINTER;
    if (f->mrefb) {                                                   // several pictures per list: every CU's pictures from its record (the lists of ks265_encode_picture_b_mref)
        KsRefExtra xr{};
        for (int r = 1; r < 4; ++r) {
            const ks265_pic a = f->mr_pic[0][r < f->mr_n[0] ? r : f->mr_n[0] - 1], b = f->mr_pic[1][r < f->mr_n[1] ? r : f->mr_n[1] - 1];
            xr.y.r[r - 1] = a.y; xr.u.r[r - 1] = a.u; xr.v.r[r - 1] = a.v; xr.y.s[r - 1] = b.y; xr.u.s[r - 1] = b.u; xr.v.s[r - 1] = b.v;
        }
        const ks265_pic a0 = f->mr_pic[0][0], b0 = f->mr_pic[1][0];
        dim3 grid(((f->g.W + 31) / 32) * ((f->g.H + 31) / 32));
        return reconstruct_sequence(f, cu8, lvl_y, [&](int rq_mode) {
        hipLaunchKernelGGL(reconstruct_kernel<true>, grid, dim3(256), 0, f->ctx->stream, f->g, f->cfg.qp, src.y, src.u, src.v, a0.y, a0.u, a0.v, b0.y, b0.u, b0.v, cu8, lvl_y, lvl_u, lvl_v,
                           recon.y, recon.u, recon.v, f->mats, xr, f->cfg.sdh, f->cfg.decimate, ks_rdo_lam2k(f), f->qp_map, f->cfg.tu_inter, rq_mode, f->rq_coef);
        });
    }
    return launch_reconstruct(f, src, ref0, ref1, cu8, lvl_y, lvl_u, lvl_v, recon);
}
