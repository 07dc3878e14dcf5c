// frame_me.hip — motion estimation stages.
//   Stage A  ks265_me_integer : one workgroup per CTU; the 64x64 source block and the +-64 reference window are staged
//            in LDS once (every window byte leaves HBM once per CTU), then all 85 PUs (64x64 .. 8x8) run the reference's
//            diamond loop (interMeDia enc@0x48fbe0) coarse to fine.  One lane = one row segment of one PU;
//            v_sad_u8 on packed dwords, v_alignbyte for the unaligned window reads, DPP/ds_swizzle group reductions.
//   Stage B  ks265_me_subpel  : 8 half-pel + 8 quarter-pel SATD candidates per PU on the precomputed fractional planes
//            (subMeSquare enc@0x4b5660; had_c enc@0x47b680).  8 lanes per 8x8 tile: horizontal Hadamard in registers,
//            vertical Hadamard across lanes.
//   Stage C  ks265_cu_decide  : bottom-up quadtree compare.
#include "frame_common.h"

using namespace ks265;

#define WIN_XL 80                 // window column 0 is picture x = ctu_x*64 - 80
#define WIN_YT 65                 // window row 0 is picture y = ctu_y*64 - 65
#define WIN_W 224                 // loaded bytes per row (x in [-80, 144))
#define WIN_ROWS 194              // y in [-65, 129)
#define WIN_STRIDE 228            // 57 dwords (odd): row-per-lane ds_read_b32 is bank-conflict free
#define FENC_STRIDE 68            // 17 dwords (odd)

__device__ __forceinline__ unsigned lds_u32(const uint8_t *p) { return *(const unsigned *)p; }

// motion predictor (see oracle pu_predictor): nearest valid ancestor's integer MV, else temporal / zero
__device__ __forceinline__ void pu_predictor(const KsGeom &g, int range, const int *pmv, const ks265_pu *prev_ctu, int cx, int cy, int l, int px,
                                             int py, int &mx, int &my, bool &root)
{
    for (int a = l - 1; a >= 0; --a) {
        int ax = px >> (l - a), ay = py >> (l - a);
        if (ks_pu_inside(g, cx, cy, a, ax, ay)) {
            int v = pmv[ks_pu_index(a, ax, ay)];
            mx = (int)(short)(v & 0xFFFF); my = v >> 16; root = false;
            return;
        }
    }
    root = true; mx = 0; my = 0;
    if (prev_ctu && prev_ctu[0].cost != KS_COST_INVALID) {
        mx = clip3(-range, range, ((int)prev_ctu[0].mvx + 2) >> 2);
        my = clip3(-range, range, ((int)prev_ctu[0].mvy + 2) >> 2);
    }
}

template <int LEVEL>
__device__ __forceinline__ void me_level(const KsGeom &g, int cx, int cy, int range, int lam, const uint8_t *win, const uint8_t *fenc, int *pmv,
                                         const ks265_pu *prev_ctu, ks265_pu *out_ctu, int tid)
{
    constexpr int S = 64 >> LEVEL;
    constexpr int G = LEVEL <= 1 ? 64 : (LEVEL == 2 ? 16 : 8);     // lanes per PU
    constexpr int D = LEVEL == 0 ? 16 : (LEVEL == 3 ? 2 : 4);      // dwords (4 pixels) per lane
    constexpr int NPU = 1 << (2 * LEVEL);
    constexpr int PER_PASS = 256 / G;
    for (int pass = 0; pass * PER_PASS < NPU; ++pass) {
        const int pu = pass * PER_PASS + tid / G;
        const bool exists = pu < NPU;
        const int px = exists ? (pu & ((1 << LEVEL) - 1)) : 0, py = exists ? (pu >> LEVEL) : 0;
        const int gl = tid % G;
        const int row = LEVEL == 1 ? (gl & 31) : gl, xoff = LEVEL == 1 ? (gl >> 5) * 16 : 0;
        const bool valid = exists && ks_pu_inside(g, cx, cy, LEVEL, px, py);
        if (exists && !valid && gl == 0) {                          // PU not (completely) inside the picture: marked, never searched
            ks265_pu o; o.mvx = o.mvy = o.mvpx = o.mvpy = 0; o.cost = KS_COST_INVALID; o.dist = KS_COST_INVALID;
            out_ctu[ks_pu_index(LEVEL, px, py)] = o;
        }
        if (!__any(valid)) continue;                                // wave-uniform
        const int bx0 = px * S + xoff, by0 = py * S + row;          // this lane's segment inside the CTU
        unsigned f[D];
#pragma unroll
        for (int j = 0; j < D; ++j) f[j] = lds_u32(fenc + by0 * FENC_STRIDE + bx0 + 4 * j);

        int pmx = 0, pmy = 0; bool root = true;
        if (valid) pu_predictor(g, range, pmv, prev_ctu, cx, cy, LEVEL, px, py, pmx, pmy, root);

        // SAD of this lane's segment at integer displacement (dx, dy), partial (before the group reduction)
        auto seg_sad = [&](int dx, int dy) -> unsigned {
            int wx = bx0 + dx + WIN_XL, wy = by0 + dy + WIN_YT;
            const uint8_t *p = win + wy * WIN_STRIDE + (wx & ~3);
            unsigned sh = wx & 3, acc = 0, lo = lds_u32(p);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                unsigned hi = lds_u32(p + 4 * (j + 1));
                acc = sad_u8x4(f[j], align_bytes(hi, lo, sh), acc);
                lo = hi;
            }
            return acc;
        };

        int mx = pmx, my = pmy;
        unsigned bcost = group_sum<G>(seg_sad(mx, my)) + (unsigned)mv_cost(mx << 2, my << 2, pmx << 2, pmy << 2, lam);
        if (__any(valid && root && (pmx | pmy))) {                  // second start candidate: the zero vector
            unsigned c0 = group_sum<G>(seg_sad(0, 0)) + (unsigned)mv_cost(0, 0, pmx << 2, pmy << 2, lam);
            if (root && (pmx | pmy) && c0 < bcost) { bcost = c0; mx = 0; my = 0; }
        }
        const int iters = root ? range : max(range >> 2, 1);
        int it = 0;
        bool active = valid;
        bcost <<= 4;
        while (__any(active)) {
            // four neighbours; left / right share the centre row reads
            unsigned s_up = seg_sad(mx, my - 1), s_dn = seg_sad(mx, my + 1), s_lf, s_rt;
            {
                int wx = bx0 + mx - 1 + WIN_XL, wy = by0 + my + WIN_YT;
                const uint8_t *p = win + wy * WIN_STRIDE + (wx & ~3);
                unsigned sl = wx & 3, sr = sl + 2;                  // right = left + 2 bytes
                bool carry = sr >= 4;
                sr &= 3;
                unsigned q0 = lds_u32(p), q1 = lds_u32(p + 4);
                s_lf = 0; s_rt = 0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    unsigned q2 = lds_u32(p + 4 * (j + 2));
                    s_lf = sad_u8x4(f[j], align_bytes(q1, q0, sl), s_lf);
                    s_rt = sad_u8x4(f[j], align_bytes(carry ? q2 : q1, carry ? q1 : q0, sr), s_rt);
                    q0 = q1; q1 = q2;
                }
            }
            unsigned c[4] = {group_sum<G>(s_up), group_sum<G>(s_dn), group_sum<G>(s_lf), group_sum<G>(s_rt)};
            if (active) {
                const int dx[4] = {0, 0, -1, 1}, dy[4] = {-1, 1, 0, 0};
                const unsigned code[4] = {1, 3, 4, 12};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int nx = mx + dx[k], ny = my + dy[k];
                    if (abs(nx) > range || abs(ny) > range) continue;
                    unsigned v = (c[k] << 4) + ((unsigned)mv_cost(nx << 2, ny << 2, pmx << 2, pmy << 2, lam) << 4) + code[k];
                    bcost = min(bcost, v);
                }
                if (!(bcost & 15)) active = false;
                else {
                    mx -= (int)((int)(bcost << 28) >> 30);
                    my -= (int)((int)(bcost << 30) >> 30);
                    bcost &= ~15u;
                    if (++it >= iters) active = false;
                }
            }
        }
        bcost >>= 4;
        if (valid && gl == 0) {
            int idx = ks_pu_index(LEVEL, px, py);
            pmv[idx] = (mx & 0xFFFF) | (my << 16);
            ks265_pu o;
            o.mvx = (int16_t)(mx << 2); o.mvy = (int16_t)(my << 2); o.mvpx = (int16_t)(pmx << 2); o.mvpy = (int16_t)(pmy << 2);
            o.cost = bcost; o.dist = bcost - (unsigned)mv_cost(mx << 2, my << 2, pmx << 2, pmy << 2, lam);
            out_ctu[idx] = o;
        }
    }
}

__global__ __launch_bounds__(256) void me_int_kernel(KsGeom g, int range, int lam, const uint8_t *src, const uint8_t *ref, const ks265_pu *prev,
                                                     ks265_pu *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[WIN_ROWS * WIN_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t fenc[64 * FENC_STRIDE];
    __shared__ int pmv[85];
    const int tid = threadIdx.x, ctu = blockIdx.x, cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const uint8_t *R = ks_org_y(g, ref), *Sp = ks_org_y(g, src);
    // reference window: 16-byte global loads (x0 - 80 is 16-byte aligned), dword LDS stores
    for (int i = tid; i < WIN_ROWS * (WIN_W / 16); i += 256) {
        int r = i / (WIN_W / 16), c = i - r * (WIN_W / 16);
        int yy = min(cy * 64 - WIN_YT + r, g.H + KS_PAD_Y - 1);     // rows past the border are never used by a valid PU
        uint4 v = *(const uint4 *)(R + (long)yy * g.sy + cx * 64 - WIN_XL + c * 16);
        unsigned *d = (unsigned *)(win + r * WIN_STRIDE + c * 16);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    {
        int r = tid >> 2, c = tid & 3;
        uint4 v = *(const uint4 *)(Sp + (long)(cy * 64 + r) * g.sy + cx * 64 + c * 16);
        unsigned *d = (unsigned *)(fenc + r * FENC_STRIDE + c * 16);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const ks265_pu *prev_ctu = prev ? prev + (long)ctu * 85 : nullptr;
    ks265_pu *out_ctu = out + (long)ctu * 85;
    me_level<0>(g, cx, cy, range, lam, win, fenc, pmv, prev_ctu, out_ctu, tid);
    __syncthreads();
    me_level<1>(g, cx, cy, range, lam, win, fenc, pmv, prev_ctu, out_ctu, tid);
    __syncthreads();
    me_level<2>(g, cx, cy, range, lam, win, fenc, pmv, prev_ctu, out_ctu, tid);
    __syncthreads();
    me_level<3>(g, cx, cy, range, lam, win, fenc, pmv, prev_ctu, out_ctu, tid);
}

extern "C" int ks265_me_integer(ks265_frame *f, ks265_pic src, ks265_pic ref, const ks265_pu *prev_pu, ks265_pu *pu)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !ref.y || !pu) return KS265_POINTER;
    if (f->cfg.me_method != 0) return KS265_NOTSUPPORTED;
    hipLaunchKernelGGL(me_int_kernel, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(256), 0, f->ctx->stream, f->g, f->cfg.me_range, f->cfg.lambda_q4,
                       src.y, ref.y, prev_pu, pu);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ Stage B: sub-pel SATD refinement
// One workgroup per CTU.  For each PU level the 64 8x8 tiles of the CTU are evaluated 32 at a time (8 lanes per tile,
// lane = tile row): per candidate each tile adds its (sum|H8 d H8^T| + 2) >> 2 to its PU's LDS accumulator
// (xCalcHADs8x8 enc@0x47b3b0 normalisation); a PU-owner thread then walks the candidates in the reference's order.
__device__ __forceinline__ unsigned tile_satd8(const unsigned f0, const unsigned f1, const uint8_t *pred_row, int lane)
{
    // 8 pixels of the prediction row at arbitrary byte alignment (global memory)
    const uint8_t *pa = (const uint8_t *)((uintptr_t)pred_row & ~(uintptr_t)3);
    unsigned sh = (unsigned)((uintptr_t)pred_row & 3);
    unsigned a0 = *(const unsigned *)pa, a1 = *(const unsigned *)(pa + 4), a2 = *(const unsigned *)(pa + 8);
    unsigned p0 = align_bytes(a1, a0, sh), p1 = align_bytes(a2, a1, sh);
    int d[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        d[i] = (int)((f0 >> (8 * i)) & 255) - (int)((p0 >> (8 * i)) & 255);
        d[4 + i] = (int)((f1 >> (8 * i)) & 255) - (int)((p1 >> (8 * i)) & 255);
    }
    // horizontal 8-point Hadamard in registers
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i += 2 * len)
#pragma unroll
            for (int j = i; j < i + len; ++j) { int u = d[j], v = d[j + len]; d[j] = u + v; d[j + len] = u - v; }
    // vertical 8-point Hadamard across the 8 lanes of the tile
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int v = d[i], p;
        p = lane_xor<1>(v); v = (lane & 1) ? p - v : v + p;
        p = lane_xor<2>(v); v = (lane & 2) ? p - v : v + p;
        p = lane_xor<4>(v); v = (lane & 4) ? p - v : v + p;
        acc += (unsigned)abs(v);
    }
    return (group_sum<8>(acc) + 2) >> 2;
}

__global__ __launch_bounds__(256) void me_subpel_kernel(KsGeom g, int lam, const uint8_t *src, const uint8_t *planes, ks265_pu *pus)
{
    __shared__ unsigned cost[64][9];          // per PU of the current level x candidate (0 = centre, 1..8 = ring)
    __shared__ int cmv[64];                   // current centre of each PU (packed qpel mv)
    __shared__ unsigned best[64][2];          // best cost / dist so far
    const int tid = threadIdx.x, lane = tid & 63, ctu = blockIdx.x, cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    ks265_pu *cp = pus + (long)ctu * 85;
    const uint8_t *Sp = ks_org_y(g, src);
    const int ox[9] = {0, -1, 0, 1, -1, 1, -1, 0, 1}, oy[9] = {0, -1, -1, -1, 0, 0, 1, 1, 1};
    for (int l = 0; l < 4; ++l) {
        const int npu = 1 << (2 * l), sh8 = 3 - l;               // tiles per PU side = 1 << sh8
        for (int i = tid; i < npu; i += 256) {
            const ks265_pu p = cp[ks_level_base(l) + i];
            cmv[i] = ((int)p.mvx & 0xFFFF) | ((int)p.mvy << 16);
            best[i][0] = KS_COST_INVALID; best[i][1] = KS_COST_INVALID;
        }
        for (int phase = 0; phase < 2; ++phase) {               // 0: centre + half-pel ring, 1: quarter-pel ring
            const int step = phase == 0 ? 2 : 1, k0 = phase == 0 ? 0 : 1;
            for (int i = tid; i < 64 * 9; i += 256) cost[i / 9][i % 9] = 0;
            __syncthreads();
            for (int tp = 0; tp < 2; ++tp) {
                const int tile = tp * 32 + (tid >> 3), tx = tile & 7, ty = tile >> 3, r = tid & 7;
                const int x0 = cx * 64 + tx * 8, y0 = cy * 64 + ty * 8;
                const int ppx = tx >> sh8, ppy = ty >> sh8, pi = ppy * (1 << l) + ppx;
                const bool valid = x0 < g.W && y0 < g.H && cp[ks_level_base(l) + pi].cost != KS_COST_INVALID;
                if (!__any(valid)) continue;
                const uint8_t *frow = Sp + (long)(y0 + r) * g.sy + x0;
                const unsigned f0 = valid ? *(const unsigned *)frow : 0, f1 = valid ? *(const unsigned *)(frow + 4) : 0;
                const int c = cmv[pi], bx = (int)(short)(c & 0xFFFF), by = c >> 16;
                for (int k = k0; k < 9; ++k) {
                    int qx = bx + ox[k] * step, qy = by + oy[k] * step;
                    if (!valid) { qx = 0; qy = 0; }
                    const uint8_t *pl = planes + (long)((qy & 3) * 4 + (qx & 3)) * g.bytes_y + g.org_y;
                    const uint8_t *prow = pl + (long)((valid ? y0 + r : 0) + (qy >> 2)) * g.sy + (valid ? x0 : 0) + (qx >> 2);
                    unsigned s = tile_satd8(f0, f1, prow, lane);
                    if (valid && r == 0) atomicAdd(&cost[pi][k], s);
                }
            }
            __syncthreads();
            for (int i = tid; i < npu; i += 256) {
                const ks265_pu p = cp[ks_level_base(l) + i];
                if (p.cost == KS_COST_INVALID) continue;
                const int c = cmv[i], cx0 = (int)(short)(c & 0xFFFF), cy0 = c >> 16;
                unsigned bc = best[i][0], bd = best[i][1];
                int bx = cx0, by = cy0;
                for (int k = k0; k < 9; ++k) {
                    int qx = cx0 + ox[k] * step, qy = cy0 + oy[k] * step;
                    unsigned d = cost[i][k], cc = d + (unsigned)mv_cost(qx, qy, p.mvpx, p.mvpy, lam);
                    if (k == 0 || cc < bc) { bc = cc; bd = d; bx = qx; by = qy; }
                }
                best[i][0] = bc; best[i][1] = bd;
                cmv[i] = (bx & 0xFFFF) | (by << 16);
            }
            __syncthreads();
        }
        for (int i = tid; i < npu; i += 256) {
            ks265_pu p = cp[ks_level_base(l) + i];
            if (p.cost == KS_COST_INVALID) continue;
            const int c = cmv[i];
            p.mvx = (int16_t)(c & 0xFFFF); p.mvy = (int16_t)(c >> 16); p.cost = best[i][0]; p.dist = best[i][1];
            cp[ks_level_base(l) + i] = p;
        }
        __syncthreads();
    }
}

extern "C" int ks265_me_subpel(ks265_frame *f, ks265_pic src, const uint8_t *planes, ks265_pu *pu)
{
    KS_FRAME_CHECK(f);
    if (!src.y || !planes || !pu) return KS265_POINTER;
    hipLaunchKernelGGL(me_subpel_kernel, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(256), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, src.y, planes, pu);
    return ks265_check_launch(f->ctx);
}

// ------------------------------------------------------------------ Stage C: CU quadtree (64 threads per CTU)
__global__ __launch_bounds__(64) void cu_decide_kernel(KsGeom g, int lam, const ks265_pu *pus, ks265_cu8 *cu8)
{
    __shared__ unsigned bestc[85];
    __shared__ unsigned char split[85];
    const int t = threadIdx.x, ctu = blockIdx.x, cx = ctu % g.ctu_cols, cy = ctu / g.ctu_cols;
    const ks265_pu *cp = pus + (long)ctu * 85;
    const unsigned pen = (unsigned)((lam * 12) >> 4);
    for (int l = 3; l >= 0; --l) {
        const int n = 1 << l, s = 64 >> l;
        for (int i = t; i < n * n; i += 64) {
            const int px = i & (n - 1), py = i >> l, idx = ks_level_base(l) + i;
            const int x0 = cx * 64 + px * s, y0 = cy * 64 + py * s;
            unsigned own = cp[idx].cost, res; unsigned char sp = 0;
            if (x0 >= g.W || y0 >= g.H) res = 0;
            else if (l == 3) res = own;
            else {
                unsigned long long sum = pen;
                for (int k = 0; k < 4; ++k) sum += bestc[ks_pu_index(l + 1, px * 2 + (k & 1), py * 2 + (k >> 1))];
                if (own != KS_COST_INVALID && (unsigned long long)own <= sum) res = own;
                else { sp = 1; res = sum > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)sum; }
            }
            bestc[idx] = res; split[idx] = sp;
        }
        __syncthreads();
    }
    // emit: thread t = 8x8 block (bx, by) of the CTU; walk down from the root
    const int bx = t & 7, by = t >> 3, X = cx * 64 + bx * 8, Y = cy * 64 + by * 8;
    if (X >= g.W || Y >= g.H) return;
    int l = 0;
    while (l < 3 && split[ks_pu_index(l, bx >> (3 - l), by >> (3 - l))]) ++l;
    const ks265_pu p = cp[ks_pu_index(l, bx >> (3 - l), by >> (3 - l))];
    ks265_cu8 c;
    c.mvx = p.mvx; c.mvy = p.mvy; c.log2_cu = (uint8_t)(6 - l); c.cbf = 0; c.pred_mode = 0; c.rsv = 0;
    cu8[(long)(Y >> 3) * g.w8 + (X >> 3)] = c;
}

__global__ __launch_bounds__(256) void cu_flat_intra_kernel(KsGeom g, ks265_cu8 *cu8)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.w8 * g.h8) return;
    int bx = i % g.w8, by = i / g.w8, lg = 3;
    for (int t = 5; t > 3; --t) {
        int n = 1 << (t - 3), ax = bx / n * n, ay = by / n * n;
        if (ax + n <= g.w8 && ay + n <= g.h8) { lg = t; break; }
    }
    ks265_cu8 c;
    c.mvx = 0; c.mvy = 0; c.log2_cu = (uint8_t)lg; c.cbf = 0; c.pred_mode = 1; c.rsv = 0;
    cu8[i] = c;
}

extern "C" int ks265_cu_decide(ks265_frame *f, const ks265_pu *pu, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!pu || !cu8) return KS265_POINTER;
    hipLaunchKernelGGL(cu_decide_kernel, dim3(f->g.ctu_cols * f->g.ctu_rows), dim3(64), 0, f->ctx->stream, f->g, f->cfg.lambda_q4, pu, cu8);
    return ks265_check_launch(f->ctx);
}

extern "C" int ks265_cu_flat_intra(ks265_frame *f, ks265_cu8 *cu8)
{
    KS_FRAME_CHECK(f);
    if (!cu8) return KS265_POINTER;
    hipLaunchKernelGGL(cu_flat_intra_kernel, dim3((f->g.w8 * f->g.h8 + 255) / 256), dim3(256), 0, f->ctx->stream, f->g, cu8);
    return ks265_check_launch(f->ctx);
}
