/* ks265_stream.c — HEVC (H.265 Main) bitstream writer for the records of the HIP pixel path: parameter sets, slice segment header,
 * CABAC slice data.  See include/ks265_stream.h for the contract.  Written from the H.265 specification (clause numbers in the
 * comments); the reference keeps this stage on the CPU too (CCtuSbac::processCtuSbac enc@0x475880, CEncCabacEngine, EncParameterSetWrite).
 * The context initialisation values and the range table are normative constants of H.265 (Tables 9-5..9-37, 9-46); they were checked
 * byte for byte against the tables in the reference binary's rodata (enc@0x4dc120 / 0x4dc1e0 / 0x4dc2a0 per slice type, enc@0x4e03a0).
 */
#include "ks265_stream.h"
#include <string.h>

/* ------------------------------------------------------------------ raw bit writer (RBSP) */
typedef struct { uint8_t *p; size_t cap, pos; uint32_t acc; int nacc; int overflow; } BitW;

static void bw_init(BitW *b, uint8_t *p, size_t cap) { b->p = p; b->cap = cap; b->pos = 0; b->acc = 0; b->nacc = 0; b->overflow = 0; }
static void bw_byte(BitW *b, unsigned v) { if (b->pos < b->cap) b->p[b->pos++] = (uint8_t)v; else b->overflow = 1; }
static void bw_put(BitW *b, uint32_t v, int n)
{
    for (int i = n - 1; i >= 0; --i) {
        b->acc = (b->acc << 1) | ((v >> i) & 1u);
        if (++b->nacc == 8) { bw_byte(b, b->acc & 0xFF); b->acc = 0; b->nacc = 0; }
    }
}
static void bw_ue(BitW *b, uint32_t v)
{
    uint32_t x = v + 1;
    int n = 0;
    while ((x >> n) > 1) ++n;
    bw_put(b, 0, n);
    bw_put(b, x, n + 1);
}
static void bw_se(BitW *b, int32_t v) { bw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_trailing(BitW *b) { bw_put(b, 1, 1); while (b->nacc) bw_put(b, 0, 1); }

/* Annex B: start code + NAL header + RBSP with emulation prevention (7.4.2) */
static long nal_wrap(int nal_type, const uint8_t *rbsp, size_t n, uint8_t *out, size_t cap)
{
    size_t o = 0;
    int zeros = 0;
    if (cap < 6) return KS265_NOTSUPPORTED;
    out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = 1;
    out[o++] = (uint8_t)(nal_type << 1);             /* forbidden_zero_bit, nal_unit_type, nuh_layer_id (high bit) */
    out[o++] = 1;                                    /* nuh_layer_id (low 5 bits) = 0, nuh_temporal_id_plus1 = 1 */
    for (size_t i = 0; i < n; ++i) {
        if (zeros >= 2 && rbsp[i] <= 3) { if (o >= cap) return KS265_NOTSUPPORTED; out[o++] = 3; zeros = 0; }
        if (o >= cap) return KS265_NOTSUPPORTED;
        out[o++] = rbsp[i];
        zeros = rbsp[i] == 0 ? zeros + 1 : 0;
    }
    return (long)o;
}

/* ------------------------------------------------------------------ parameter sets (7.3.2) */
static void profile_tier_level(BitW *b)
{
    bw_put(b, 0, 2);                     /* general_profile_space */
    bw_put(b, 0, 1);                     /* general_tier_flag */
    bw_put(b, 1, 5);                     /* general_profile_idc = Main */
    bw_put(b, 0x60000000u, 32);          /* general_profile_compatibility_flag[1] (Main) and [2] (Main 10) */
    bw_put(b, 1, 1);                     /* general_progressive_source_flag */
    bw_put(b, 0, 1);                     /* general_interlaced_source_flag */
    bw_put(b, 0, 1);                     /* general_non_packed_constraint_flag */
    bw_put(b, 1, 1);                     /* general_frame_only_constraint_flag */
    bw_put(b, 0, 32); bw_put(b, 0, 12);  /* general_reserved_zero_44bits */
    bw_put(b, 183, 8);                   /* general_level_idc = 6.1 (no level limit is enforced by the pixel path) */
}

long ks265_write_vps(const ks265_stream_cfg *cfg, uint8_t *out, size_t cap)
{
    if (!cfg || !out) return KS265_POINTER;
    uint8_t buf[64];
    BitW b; bw_init(&b, buf, sizeof buf);
    bw_put(&b, 0, 4);                    /* vps_video_parameter_set_id */
    bw_put(&b, 3, 2);                    /* vps_base_layer_internal_flag, vps_base_layer_available_flag */
    bw_put(&b, 0, 6);                    /* vps_max_layers_minus1 */
    bw_put(&b, 0, 3);                    /* vps_max_sub_layers_minus1 */
    bw_put(&b, 1, 1);                    /* vps_temporal_id_nesting_flag */
    bw_put(&b, 0xFFFF, 16);              /* vps_reserved_0xffff_16bits */
    profile_tier_level(&b);
    bw_put(&b, 1, 1);                    /* vps_sub_layer_ordering_info_present_flag */
    bw_ue(&b, (uint32_t)(cfg->max_dec_pic_buffering - 1));
    bw_ue(&b, (uint32_t)cfg->max_num_reorder);
    bw_ue(&b, 0);                        /* vps_max_latency_increase_plus1 */
    bw_put(&b, 0, 6);                    /* vps_max_layer_id */
    bw_ue(&b, 0);                        /* vps_num_layer_sets_minus1 */
    bw_put(&b, 0, 1);                    /* vps_timing_info_present_flag */
    bw_put(&b, 0, 1);                    /* vps_extension_flag */
    bw_trailing(&b);
    return b.overflow ? KS265_NOTSUPPORTED : nal_wrap(KS265_NAL_VPS, buf, b.pos, out, cap);
}

static int cfg_ok(const ks265_stream_cfg *c)
{
    return c->width > 0 && c->height > 0 && !(c->width & 7) && !(c->height & 7) && c->width <= 8192 && c->height <= 8192 && c->max_dec_pic_buffering >= 1 &&
           c->max_dec_pic_buffering <= 16 && c->max_num_reorder >= 0 && c->max_num_reorder < c->max_dec_pic_buffering && c->log2_max_poc_lsb >= 4 && c->log2_max_poc_lsb <= 16;
}

long ks265_write_sps(const ks265_stream_cfg *cfg, uint8_t *out, size_t cap)
{
    if (!cfg || !out) return KS265_POINTER;
    if (!cfg_ok(cfg)) return KS265_NOTSUPPORTED;
    uint8_t buf[96];
    BitW b; bw_init(&b, buf, sizeof buf);
    bw_put(&b, 0, 4);                    /* sps_video_parameter_set_id */
    bw_put(&b, 0, 3);                    /* sps_max_sub_layers_minus1 */
    bw_put(&b, 1, 1);                    /* sps_temporal_id_nesting_flag */
    profile_tier_level(&b);
    bw_ue(&b, 0);                        /* sps_seq_parameter_set_id */
    bw_ue(&b, 1);                        /* chroma_format_idc = 4:2:0 */
    bw_ue(&b, (uint32_t)cfg->width);
    bw_ue(&b, (uint32_t)cfg->height);
    bw_put(&b, 0, 1);                    /* conformance_window_flag */
    bw_ue(&b, 0); bw_ue(&b, 0);          /* bit_depth_luma_minus8, bit_depth_chroma_minus8 */
    bw_ue(&b, (uint32_t)(cfg->log2_max_poc_lsb - 4));
    bw_put(&b, 1, 1);                    /* sps_sub_layer_ordering_info_present_flag */
    bw_ue(&b, (uint32_t)(cfg->max_dec_pic_buffering - 1));
    bw_ue(&b, (uint32_t)cfg->max_num_reorder);
    bw_ue(&b, 0);                        /* sps_max_latency_increase_plus1 */
    bw_ue(&b, 0);                        /* log2_min_luma_coding_block_size_minus3: 8 */
    bw_ue(&b, 3);                        /* log2_diff_max_min_luma_coding_block_size: 64 */
    bw_ue(&b, 0);                        /* log2_min_luma_transform_block_size_minus2: 4 */
    bw_ue(&b, 3);                        /* log2_diff_max_min_luma_transform_block_size: 32 */
    bw_ue(&b, cfg->tu_inter ? 1 : 0);    /* max_transform_hierarchy_depth_inter */
    bw_ue(&b, 0);                        /* max_transform_hierarchy_depth_intra */
    bw_put(&b, 0, 1);                    /* scaling_list_enabled_flag */
    bw_put(&b, 0, 1);                    /* amp_enabled_flag */
    bw_put(&b, cfg->sao ? 1 : 0, 1);     /* sample_adaptive_offset_enabled_flag */
    bw_put(&b, 0, 1);                    /* pcm_enabled_flag */
    bw_ue(&b, 0);                        /* num_short_term_ref_pic_sets: every slice carries its own */
    bw_put(&b, 0, 1);                    /* long_term_ref_pics_present_flag */
    bw_put(&b, 0, 1);                    /* sps_temporal_mvp_enabled_flag */
    bw_put(&b, 1, 1);                    /* strong_intra_smoothing_enabled_flag */
    bw_put(&b, 0, 1);                    /* vui_parameters_present_flag */
    bw_put(&b, 0, 1);                    /* sps_extension_present_flag */
    bw_trailing(&b);
    return b.overflow ? KS265_NOTSUPPORTED : nal_wrap(KS265_NAL_SPS, buf, b.pos, out, cap);
}

long ks265_write_pps(const ks265_stream_cfg *cfg, uint8_t *out, size_t cap)
{
    if (!cfg || !out) return KS265_POINTER;
    uint8_t buf[64];
    BitW b; bw_init(&b, buf, sizeof buf);
    bw_ue(&b, 0);                        /* pps_pic_parameter_set_id */
    bw_ue(&b, 0);                        /* pps_seq_parameter_set_id */
    bw_put(&b, 0, 1);                    /* dependent_slice_segments_enabled_flag */
    bw_put(&b, 0, 1);                    /* output_flag_present_flag */
    bw_put(&b, 0, 3);                    /* num_extra_slice_header_bits */
    bw_put(&b, cfg->sdh ? 1 : 0, 1);     /* sign_data_hiding_enabled_flag */
    bw_put(&b, 0, 1);                    /* cabac_init_present_flag */
    bw_ue(&b, 0); bw_ue(&b, 0);          /* num_ref_idx_l0 / l1_default_active_minus1 */
    bw_se(&b, 0);                        /* init_qp_minus26 */
    bw_put(&b, 0, 1);                    /* constrained_intra_pred_flag */
    bw_put(&b, 0, 1);                    /* transform_skip_enabled_flag */
    bw_put(&b, cfg->cu_qp_delta ? 1 : 0, 1);   /* cu_qp_delta_enabled_flag */
    if (cfg->cu_qp_delta) bw_ue(&b, 0);  /* diff_cu_qp_delta_depth: one quantisation group per CTU */
    bw_se(&b, 0); bw_se(&b, 0);          /* pps_cb_qp_offset, pps_cr_qp_offset */
    bw_put(&b, 0, 1);                    /* pps_slice_chroma_qp_offsets_present_flag */
    bw_put(&b, 0, 1);                    /* weighted_pred_flag */
    bw_put(&b, 0, 1);                    /* weighted_bipred_flag */
    bw_put(&b, 0, 1);                    /* transquant_bypass_enabled_flag */
    bw_put(&b, 0, 1);                    /* tiles_enabled_flag */
    bw_put(&b, cfg->wpp ? 1 : 0, 1);     /* entropy_coding_sync_enabled_flag */
    bw_put(&b, 0, 1);                    /* pps_loop_filter_across_slices_enabled_flag */
    bw_put(&b, 1, 1);                    /* deblocking_filter_control_present_flag */
    bw_put(&b, 0, 1);                    /* deblocking_filter_override_enabled_flag */
    bw_put(&b, cfg->deblock ? 0 : 1, 1); /* pps_deblocking_filter_disabled_flag */
    if (cfg->deblock) { bw_se(&b, cfg->beta_offset_div2); bw_se(&b, cfg->tc_offset_div2); }
    bw_put(&b, 0, 1);                    /* pps_scaling_list_data_present_flag */
    bw_put(&b, cfg->list_mod ? 1 : 0, 1);/* lists_modification_present_flag */
    bw_ue(&b, 0);                        /* log2_parallel_merge_level_minus2 */
    bw_put(&b, 0, 1);                    /* slice_segment_header_extension_present_flag */
    bw_put(&b, 0, 1);                    /* pps_extension_present_flag */
    bw_trailing(&b);
    return b.overflow ? KS265_NOTSUPPORTED : nal_wrap(KS265_NAL_PPS, buf, b.pos, out, cap);
}

/* ------------------------------------------------------------------ CABAC encoder (9.3.4.4 .. 9.3.4.6) */
static const uint8_t kRangeTabLps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166},
    {95, 116, 137, 158}, {90, 110, 130, 150}, {85, 104, 123, 142}, {81, 99, 117, 135}, {77, 94, 111, 128}, {73, 89, 105, 122}, {69, 85, 100, 116}, {66, 80, 95, 110},
    {62, 76, 90, 104}, {59, 72, 86, 99}, {56, 69, 81, 94}, {53, 65, 77, 89}, {51, 62, 73, 85}, {48, 59, 69, 80}, {46, 56, 66, 76}, {43, 53, 63, 72},
    {41, 50, 59, 69}, {39, 48, 56, 65}, {37, 45, 54, 62}, {35, 43, 51, 59}, {33, 41, 48, 56}, {32, 39, 46, 53}, {30, 37, 43, 50}, {29, 35, 41, 48},
    {27, 33, 39, 45}, {26, 31, 37, 43}, {24, 30, 35, 41}, {23, 28, 33, 39}, {22, 27, 32, 37}, {21, 26, 30, 35}, {20, 24, 29, 33}, {19, 23, 27, 31},
    {18, 22, 26, 30}, {17, 21, 25, 28}, {16, 20, 23, 27}, {15, 19, 22, 25}, {14, 18, 21, 24}, {14, 17, 20, 23}, {13, 16, 19, 22}, {12, 15, 18, 21},
    {12, 14, 17, 20}, {11, 14, 16, 19}, {11, 13, 15, 18}, {10, 12, 15, 17}, {10, 12, 14, 16}, {9, 11, 13, 15}, {9, 11, 12, 14}, {8, 10, 12, 14},
    {8, 9, 11, 13}, {7, 9, 11, 12}, {7, 9, 10, 12}, {7, 8, 10, 11}, {6, 8, 9, 11}, {6, 7, 9, 10}, {6, 7, 8, 9}, {2, 2, 2, 2}};
static const uint8_t kTransIdxLps[64] = {0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                         24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

/* context layout */
enum {
    CX_SAO_MERGE = 0, CX_SAO_TYPE = 1, CX_SPLIT_CU = 2, CX_SKIP = 5, CX_PRED_MODE = 8, CX_PART_MODE = 9, CX_PREV_INTRA = 13, CX_CHROMA_PRED = 14,
    CX_MERGE_FLAG = 15, CX_MERGE_IDX = 16, CX_INTER_DIR = 17, CX_REF_IDX = 22, CX_MVP = 24, CX_MVD = 25, CX_ROOT_CBF = 27, CX_SPLIT_TU = 28,
    CX_CBF_LUMA = 31, CX_CBF_CHROMA = 33, CX_LAST_X = 37, CX_LAST_Y = 55, CX_CSBF = 73, CX_SIG = 77, CX_G1 = 119, CX_G2 = 143, CX_DQP = 149 /* cu_qp_delta_abs: first bin, later bins */, CX_COUNT = 151
};
/* initValue per context for initType 0 (I), 1 (P), 2 (B): Tables 9-5 .. 9-37.  154 where a syntax element does not occur. */
static const uint8_t kInit[3][CX_COUNT] = {
    {153, 200, 139, 141, 157, 154, 154, 154, 154, 184, 154, 154, 154, 184, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 153, 138, 138,
     111, 141, 94, 138, 182, 154,
     110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
     91, 171, 134, 141,
     111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,
     140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
     138, 153, 136, 167, 152, 152,
     154, 154},
    {153, 185, 107, 139, 126, 197, 185, 201, 149, 154, 139, 154, 154, 154, 152, 110, 122, 95, 79, 63, 31, 31, 153, 153, 168, 140, 198, 79, 124, 138, 94,
     153, 111, 149, 107, 167, 154,
     125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108, 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
     121, 140, 61, 154,
     155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
     154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
     107, 167, 91, 122, 107, 167,
     154, 154},
    {153, 160, 107, 139, 126, 197, 185, 201, 134, 154, 139, 154, 154, 183, 152, 154, 137, 95, 79, 63, 31, 31, 153, 153, 168, 169, 198, 79, 224, 167, 122,
     153, 111, 149, 92, 167, 154,
     125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93, 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
     121, 140, 61, 154,
     170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140,
     154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182,
     107, 167, 91, 107, 107, 167,
     154, 154}};

typedef struct {
    uint8_t *p; size_t cap, pos; int overflow;
    uint32_t low, range; int bits_left, num_buffered; unsigned buffered_byte;
    uint8_t state[CX_COUNT];               /* pStateIdx << 1 | valMps */
#ifdef KS265_BIT_STATS
    int cat;                               /* diagnostic build only (scratch): information content of the bins by syntax category */
#endif
} Cabac;
#ifdef KS265_BIT_STATS
#include <math.h>
enum { CAT_CU, CAT_MERGE, CAT_MOTION, CAT_COEF, CAT_SAO, CAT_INTRA, CAT_N };
static __thread double g_bit_stats[CAT_N];
void ks265_bit_stats(double *out, int reset) { for (int i = 0; i < CAT_N; ++i) { out[i] = g_bit_stats[i]; if (reset) g_bit_stats[i] = 0; } }
#define CAT(c, k) ((c)->cat = (k))
#define CAT_GET(c) ((c)->cat)
#else
#define CAT(c, k) ((void)0)
#define CAT_GET(c) 0
#endif

static void cb_out(Cabac *c, unsigned v) { if (c->pos < c->cap) c->p[c->pos++] = (uint8_t)v; else c->overflow = 1; }
static void cb_init(Cabac *c, uint8_t *p, size_t cap, int init_type, int qp)
{
    c->p = p; c->cap = cap; c->pos = 0; c->overflow = 0;
    c->low = 0; c->range = 510; c->bits_left = 23; c->num_buffered = 0; c->buffered_byte = 0xFF;
    CAT(c, 0);
    for (int i = 0; i < CX_COUNT; ++i) {                           /* 9.3.2.2 */
        const int iv = kInit[init_type][i], slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
        int pre = ((slope * (qp < 0 ? 0 : qp > 51 ? 51 : qp)) >> 4) + offset;
        pre = pre < 1 ? 1 : pre > 126 ? 126 : pre;
        const int mps = pre > 63;
        c->state[i] = (uint8_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps);
    }
}
static void cb_write_out(Cabac *c)
{
    const unsigned lead = c->low >> (24 - c->bits_left);
    c->bits_left += 8;
    c->low &= 0xFFFFFFFFu >> c->bits_left;
    if (lead == 0xFF) { ++c->num_buffered; return; }
    if (c->num_buffered > 0) {
        const unsigned carry = lead >> 8;
        cb_out(c, c->buffered_byte + carry);
        c->buffered_byte = lead & 0xFF;
        const unsigned fill = (0xFF + carry) & 0xFF;
        while (c->num_buffered > 1) { cb_out(c, fill); --c->num_buffered; }
    } else {
        c->num_buffered = 1;
        c->buffered_byte = lead;
    }
}
static const uint8_t kRenorm[32] = {6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
static inline void cb_bin(Cabac *c, int ctx, int bin)
{
    uint8_t s = c->state[ctx];
    const unsigned lps = kRangeTabLps[s >> 1][(c->range >> 6) & 3];
#ifdef KS265_BIT_STATS
    g_bit_stats[c->cat] += log2((double)c->range / (double)(((bin & 1) != (s & 1)) ? lps : c->range - lps));
#endif
    c->range -= lps;
    if ((bin & 1) != (s & 1)) {
        const int nb = kRenorm[lps >> 3];                               /* lps in 6 .. 240: shifts until bit 8 is set */
        c->low = (c->low + c->range) << nb;
        c->range = lps << nb;
        if ((s >> 1) == 0) s ^= 1;
        c->state[ctx] = (uint8_t)((kTransIdxLps[s >> 1] << 1) | (s & 1));
        c->bits_left -= nb;
    } else {
        if ((s >> 1) < 62) c->state[ctx] = (uint8_t)(s + 2);
        if (c->range >= 256) return;
        c->low <<= 1; c->range <<= 1;
        --c->bits_left;
    }
    if (c->bits_left < 12) cb_write_out(c);
}
static inline void cb_bypass(Cabac *c, int bin)
{
#ifdef KS265_BIT_STATS
    g_bit_stats[c->cat] += 1.0;
#endif
    c->low <<= 1;
    if (bin) c->low += c->range;
    if (--c->bits_left < 12) cb_write_out(c);
}
/* n bypass bins at once, most significant first (9.3.4.3.4 n times: low = 2 low + bin x range, written out after at most 8 of them) */
static void cb_bypass_bits(Cabac *c, uint32_t v, int n)
{
#ifdef KS265_BIT_STATS
    g_bit_stats[c->cat] += n;
#endif
    while (n > 0) {
        const int k = n > 8 ? 8 : n;
        n -= k;
        c->low = (c->low << k) + c->range * ((v >> n) & ((1u << k) - 1));
        c->bits_left -= k;
        if (c->bits_left < 12) cb_write_out(c);
    }
}
static void cb_terminate(Cabac *c, int bin)
{
    c->range -= 2;
    if (bin) {
        c->low += c->range;
        c->low <<= 7; c->range = 2 << 7;
        c->bits_left -= 7;
    } else if (c->range >= 256) return;
    else { c->low <<= 1; c->range <<= 1; --c->bits_left; }
    if (c->bits_left < 12) cb_write_out(c);
}
/* 9.3.4.5 flushing at the end of the slice segment; the rbsp_slice_segment_trailing_bits follow */
static void cb_finish(Cabac *c)
{
    if (c->low >> (32 - c->bits_left)) {
        cb_out(c, c->buffered_byte + 1);
        while (c->num_buffered > 1) { cb_out(c, 0x00); --c->num_buffered; }
        c->low -= 1u << (32 - c->bits_left);
    } else {
        if (c->num_buffered > 0) cb_out(c, c->buffered_byte);
        while (c->num_buffered > 1) { cb_out(c, 0xFF); --c->num_buffered; }
    }
    /* the remaining 24 - bits_left bits of low >> 8, then the stop bit and alignment */
    const int n = 24 - c->bits_left;
    uint32_t v = c->low >> 8;
    uint32_t acc = 0; int na = 0;
    for (int i = n - 1; i >= 0; --i) { acc = (acc << 1) | ((v >> i) & 1u); if (++na == 8) { cb_out(c, acc); acc = 0; na = 0; } }
    acc = (acc << 1) | 1u; ++na;                                    /* rbsp_stop_one_bit */
    while (na < 8) { acc <<= 1; ++na; }
    cb_out(c, acc);
}

/* ------------------------------------------------------------------ scans (6.5.3 .. 6.5.5) */
typedef struct { uint8_t x, y; } XY;
static void build_scan(int scan_idx, int size, XY *out)
{
    int i = 0;
    if (scan_idx == 0) {                                            /* up-right diagonal */
        int x = 0, y = 0, stop = 0;
        while (!stop) {
            while (y >= 0) { if (x < size && y < size) { out[i].x = (uint8_t)x; out[i].y = (uint8_t)y; ++i; } --y; ++x; }
            y = x; x = 0;
            if (i >= size * size) stop = 1;
        }
    } else if (scan_idx == 1) { for (int y = 0; y < size; ++y) for (int x = 0; x < size; ++x) { out[i].x = (uint8_t)x; out[i].y = (uint8_t)y; ++i; } }
    else { for (int x = 0; x < size; ++x) for (int y = 0; y < size; ++y) { out[i].x = (uint8_t)x; out[i].y = (uint8_t)y; ++i; } }
}
static const uint8_t kCtxIdxMap4x4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
typedef struct {
    XY pos4[3][16];          /* positions inside a 4x4 sub-block, per scanIdx */
    XY sb[3][4][64];         /* sub-block order for log2 size 2..5 (1, 4, 16, 64 sub-blocks), per scanIdx */
    /* sig_coeff_flag context (9.3.4.2.5) of scan position n of a sub-block, offset from CX_SIG: [log2 - 2][chroma][scanIdx][sub-block is not the first][csbf right + 2 below][n] */
    uint8_t sigctx[4][2][3][2][4][16];
} Scans;
static void scans_init(Scans *s)
{
    for (int k = 0; k < 3; ++k) {
        build_scan(k, 4, s->pos4[k]);
        for (int l = 0; l < 4; ++l) build_scan(k, 1 << l, s->sb[k][l]);
    }
    for (int log2 = 2; log2 <= 5; ++log2)
        for (int ch = 0; ch < 2; ++ch)
            for (int k = 0; k < 3; ++k)
                for (int nf = 0; nf < 2; ++nf)
                    for (int pc = 0; pc < 4; ++pc)
                        for (int n = 0; n < 16; ++n) {
                            const int xp = s->pos4[k][n].x, yp = s->pos4[k][n].y;
                            int sc;
                            if (log2 == 2) sc = kCtxIdxMap4x4[(yp << 2) + xp];
                            else if (!nf && xp + yp == 0) sc = 0;
                            else {
                                if (pc == 0) sc = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
                                else if (pc == 1) sc = (yp == 0) ? 2 : (yp == 1) ? 1 : 0;
                                else if (pc == 2) sc = (xp == 0) ? 2 : (xp == 1) ? 1 : 0;
                                else sc = 2;
                                if (!ch) { if (nf) sc += 3; sc += log2 == 3 ? (k == 0 ? 9 : 15) : 21; }
                                else sc += log2 == 3 ? 9 : 12;
                            }
                            s->sigctx[log2 - 2][ch][k][nf][pc][n] = (uint8_t)(ch ? 27 + sc : sc);
                        }
}

/* ------------------------------------------------------------------ slice state */
typedef struct {
    const ks265_stream_cfg *cfg;
    const ks265_slice_in *in;
    int W, H, w8, h8, ctb_cols, ctb_rows;
    Cabac c;
    Scans scans;
    uint8_t *skip;                         /* cu_skip_flag of every 8x8 block coded so far (context of the neighbours) */
    int qp_prev, qp_want, dqp_coded;       /* cfg.cu_qp_delta: QpY of the previous quantisation group's last CU, this CTU's QP, IsCuQpDeltaCoded */
} Enc;

size_t ks265_wpp_bytes(const ks265_stream_cfg *cfg);
size_t ks265_slice_scratch_bytes(const ks265_stream_cfg *cfg)
{
    /* Enc + an RBSP buffer generous enough for any picture: the levels are 16-bit, worst case about 3 bytes per sample */
    const size_t whole = sizeof(Enc) + (size_t)cfg->width * (size_t)cfg->height * 4 + 65536 + (size_t)(cfg->width >> 3) * (size_t)(cfg->height >> 3);
    const size_t rows = ks265_wpp_bytes(cfg);                         /* cfg->wpp: the row-wise writer's job memory */
    return whole > rows ? whole : rows;
}

static inline const ks265_cu8 *cu_at(const Enc *e, int x, int y) { return &e->in->cu8[(long)(y >> 3) * e->w8 + (x >> 3)]; }
static inline int is_intra(const ks265_cu8 *c) { return c->pred_mode != 0; }
/* z-scan order address of the 8x8 block containing luma sample (x, y): CTB raster address, then Morton order inside the CTB (6.4.1) */
static int zaddr(const Enc *e, int x, int y)
{
    const int bx = (x >> 3) & 7, by = (y >> 3) & 7;
    int m = 0;
    for (int b = 0; b < 3; ++b) m |= (((bx >> b) & 1) << (2 * b)) | (((by >> b) & 1) << (2 * b + 1));
    return (((y >> 6) * e->ctb_cols + (x >> 6)) << 6) | m;
}
/* 6.4.1: is the block at (xn, yn) available to the block at (xc, yc)?  (one slice, no tiles) */
static int avail_z(const Enc *e, int xc, int yc, int xn, int yn)
{
    if (xn < 0 || yn < 0 || xn >= e->W || yn >= e->H) return 0;
    return zaddr(e, xn, yn) <= zaddr(e, xc, yc);
}

/* 6.4.2: availability for a prediction block at (xp, yp) of the coding block (xc, yc, cs): a location inside the same coding block belongs to the partition decoded
 * before this one and is available (the N x N exception does not arise: inter CUs here are 2Nx2N, 2NxN or Nx2N); everything else by z-scan order from (xp, yp) */
static int avail_pb(const Enc *e, int xc, int yc, int cs, int xp, int yp, int xn, int yn)
{
    if (xn >= xc && xn < xc + cs && yn >= yc && yn < yc + cs) return !(xn >= xp && yn >= yp);      /* same coding block: the earlier partition only */
    return avail_z(e, xp, yp, xn, yn);
}

/* ------------------------------------------------------------------ SAO syntax (7.3.8.3) */
static void sao_offsets(Enc *e, const ks265_sao_param *p)
{
    Cabac *c = &e->c;
    for (int i = 0; i < 4; ++i) {                                   /* sao_offset_abs: TR, cMax = 7, bypass */
        const int a = p->offset[i] < 0 ? -p->offset[i] : p->offset[i];
        for (int k = 0; k < a; ++k) cb_bypass(c, 1);
        if (a < 7) cb_bypass(c, 0);
    }
    if (p->type == 0) {                                             /* band offset: signs of the non-zero offsets, then the band position */
        for (int i = 0; i < 4; ++i) if (p->offset[i]) cb_bypass(c, p->offset[i] < 0);
        cb_bypass_bits(c, (uint32_t)p->band, 5);
    }
}
static void sao_ctb(Enc *e, int rx, int ry)
{
    Cabac *c = &e->c;
    const ks265_sao_param *p = e->in->sao + (long)(ry * e->ctb_cols + rx) * 3;
    CAT(c, CAT_SAO);
    if (rx > 0) cb_bin(c, CX_SAO_MERGE, 0);                          /* sao_merge_left_flag */
    if (ry > 0) cb_bin(c, CX_SAO_MERGE, 0);                          /* sao_merge_up_flag */
    for (int ci = 0; ci < 3; ++ci) {
        const ks265_sao_param *q = p + ci;
        if (ci < 2) {                                               /* sao_type_idx_luma / _chroma (shared by Cb and Cr): 0 off, 1 band, 2 edge */
            if (q->type < 0) cb_bin(c, CX_SAO_TYPE, 0);
            else { cb_bin(c, CX_SAO_TYPE, 1); cb_bypass(c, q->type == 0 ? 0 : 1); }
        }
        if (q->type < 0) continue;
        sao_offsets(e, q);
        if (q->type > 0 && ci < 2) cb_bypass_bits(c, (uint32_t)(q->type - 1), 2);   /* sao_eo_class_luma / _chroma */
    }
    CAT(c, CAT_CU);
}

/* ------------------------------------------------------------------ residual_coding (7.3.8.11, 9.3.4.2.4 .. 9.3.4.2.7) */

static void code_last_prefix(Cabac *c, int ctx_base, int v, int log2, int cidx)
{
    int off, shift;
    if (cidx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); shift = (log2 + 1) >> 2; }
    else { off = 15; shift = log2 - 2; }
    const int cmax = (log2 << 1) - 1;
    for (int i = 0; i < v; ++i) cb_bin(c, ctx_base + off + (i >> shift), 1);
    if (v < cmax) cb_bin(c, ctx_base + off + (v >> shift), 0);
}
static void last_pos_bins(int pos, int *prefix, int *suffix, int *nsuf)
{
    if (pos < 4) { *prefix = pos; *nsuf = 0; *suffix = 0; return; }
    int p = 4;
    for (;; ++p) {                                                  /* pos = (1 << ((p >> 1) - 1)) * (2 + (p & 1)) + suffix */
        const int len = (p >> 1) - 1, base = (1 << len) * (2 + (p & 1));
        if (pos < base + (1 << len)) { *prefix = p; *nsuf = len; *suffix = pos - base; return; }
    }
}
static void code_remaining(Cabac *c, unsigned v, int rice)                 /* coeff_abs_level_remaining (9.3.3.11), bypass */
{
    if (v < (3u << rice)) {
        const unsigned len = v >> rice;
        cb_bypass_bits(c, (1u << (len + 1)) - 2, (int)len + 1);
        cb_bypass_bits(c, v & ((1u << rice) - 1), rice);
    } else {
        unsigned len = (unsigned)rice;
        v -= 3u << rice;
        while (v >= (1u << len)) { v -= 1u << len; ++len; }
        const int pre = (int)(3 + len + 1 - (unsigned)rice);
        for (int i = 0; i < pre - 1; ++i) cb_bypass(c, 1);
        cb_bypass(c, 0);
        cb_bypass_bits(c, v, (int)len);
    }
}

static void residual_coding(Enc *e, const int16_t *blk, int stride, int log2, int cidx, int scan_idx)
{
    Cabac *c = &e->c;
    const int cat0 = CAT_GET(c); (void)cat0;
    CAT(c, CAT_COEF);
    const int size = 1 << log2, nsb_log2 = log2 - 2, nsb = 1 << (2 * nsb_log2);
    const XY *sbs = e->scans.sb[scan_idx][nsb_log2], *p4 = e->scans.pos4[scan_idx];
    /* which 4x4 sub-blocks hold a level at all: four 8-byte reads per sub-block (most TUs of a P picture hold a handful of levels) */
    uint8_t sbnz[8][8];
    {
        const int ns = 1 << nsb_log2;
        for (int ys = 0; ys < ns; ++ys)
            for (int xs = 0; xs < ns; ++xs) {
                uint64_t a, o = 0;
                for (int r = 0; r < 4; ++r) { memcpy(&a, blk + (ys * 4 + r) * stride + xs * 4, 8); o |= a; }
                sbnz[ys][xs] = o != 0;
            }
    }
    /* last significant coefficient in scan order */
    int last_sb = -1, last_n = -1;
    for (int i = nsb - 1; i >= 0 && last_sb < 0; --i) {
        if (!sbnz[sbs[i].y][sbs[i].x]) continue;
        for (int n = 15; n >= 0; --n) {
            const int x = sbs[i].x * 4 + p4[n].x, y = sbs[i].y * 4 + p4[n].y;
            if (blk[y * stride + x]) { last_sb = i; last_n = n; break; }
        }
    }
    if (last_sb < 0) { CAT(c, cat0); return; }                                         /* cbf said otherwise: never reached */
    int lx = sbs[last_sb].x * 4 + p4[last_n].x, ly = sbs[last_sb].y * 4 + p4[last_n].y;
    if (scan_idx == 2) { const int t = lx; lx = ly; ly = t; }       /* vertical scan: the coordinates are swapped in the syntax */
    int px, sx, nx, py, sy, ny;
    last_pos_bins(lx, &px, &sx, &nx);
    last_pos_bins(ly, &py, &sy, &ny);
    code_last_prefix(c, CX_LAST_X, px, log2, cidx);
    code_last_prefix(c, CX_LAST_Y, py, log2, cidx);
    if (nx) cb_bypass_bits(c, (uint32_t)sx, nx);
    if (ny) cb_bypass_bits(c, (uint32_t)sy, ny);

    uint8_t csbf[8][8];
    memset(csbf, 0, sizeof csbf);
    int c1 = 1;                                                      /* greater1Ctx carried between sub-blocks (9.3.4.2.6) */
    (void)size;
    for (int i = last_sb; i >= 0; --i) {
        const int xs = sbs[i].x, ys = sbs[i].y;
        int absv[16], sign[16], npos[16], nsig = 0;
        /* the sub-block's levels in scan order, and which of them are significant */
        int16_t v16[16];
        uint16_t sigmask = 0;
        if (sbnz[ys][xs]) {
            const int16_t *b0 = blk + (ys * 4) * stride + xs * 4;
            for (int n = (i == last_sb ? last_n : 15); n >= 0; --n) {
                const int v = b0[p4[n].y * stride + p4[n].x];
                v16[n] = (int16_t)v;
                if (v) sigmask |= (uint16_t)(1u << n);
            }
        }
        const int right = xs + 1 < (1 << nsb_log2) ? csbf[ys][xs + 1] : 0, below = ys + 1 < (1 << nsb_log2) ? csbf[ys + 1][xs] : 0;
        int coded = sigmask != 0, infer_dc = 0;
        if (i < last_sb && i > 0) { cb_bin(c, CX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0), coded); infer_dc = 1; }
        else coded = 1;                                              /* first and last sub-block: inferred 1 */
        csbf[ys][xs] = (uint8_t)coded;
        if (!coded) continue;
        /* sig_coeff_flag */
        const uint8_t *sctx = e->scans.sigctx[log2 - 2][cidx != 0][scan_idx][(xs | ys) != 0][right + 2 * below];
        for (int n = (i == last_sb ? last_n - 1 : 15); n >= 0; --n) {
            const int sig = (sigmask >> n) & 1;
            if (n > 0 || !infer_dc) {
                cb_bin(c, CX_SIG + sctx[n], sig);
                if (sig) infer_dc = 0;
            }
        }
        /* levels of the sub-block in coding order (high scan position first) */
        for (unsigned m = sigmask; m; ) {
            const int n = 31 - __builtin_clz(m);
            m &= ~(1u << n);
            const int v = v16[n];
            absv[nsig] = v < 0 ? -v : v; sign[nsig] = v < 0; npos[nsig] = n; ++nsig;
        }
        if (!nsig) continue;                                         /* the inferred DC of a coded sub-block is always significant */
        int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
        if (c1 == 0) ++ctx_set;
        c1 = 1;
        const int n1 = nsig < 8 ? nsig : 8;
        int first_g2 = -1;
        for (int k = 0; k < n1; ++k) {
            const int g = absv[k] > 1;
            cb_bin(c, CX_G1 + (cidx ? 16 : 0) + 4 * ctx_set + c1, g);
            if (g) { c1 = 0; if (first_g2 < 0) first_g2 = k; }
            else if (c1 < 3 && c1 > 0) ++c1;
        }
        if (c1 == 0 && first_g2 >= 0) cb_bin(c, CX_G2 + (cidx ? 4 : 0) + ctx_set, absv[first_g2] > 2);
        /* coeff_sign_flag; with sign-data hiding the sign of the group's first coefficient in scan order (= the last one coded) is inferred from
         * the parity of the level sum when the first and the last significant position are more than 3 apart */
        const int hidden = e->cfg->sdh && npos[0] - npos[nsig - 1] > 3;
        for (int k = 0; k < nsig - hidden; ++k) cb_bypass(c, sign[k]);
        if (c1 == 0 || nsig > 8) {
            int first_coeff2 = 1, rice = 0;
            for (int k = 0; k < nsig; ++k) {
                const int base = k < 8 ? 2 + first_coeff2 : 1;
                if (absv[k] >= base) {
                    code_remaining(c, (unsigned)(absv[k] - base), rice);
                    if (absv[k] > 3 * (1 << rice)) rice = rice < 4 ? rice + 1 : 4;
                }
                if (absv[k] >= 2) first_coeff2 = 0;
            }
        }
    }
    CAT(c, cat0);
}

/* ------------------------------------------------------------------ motion vector prediction (8.5.3.2.6, 8.5.3.2.7; no temporal candidate) */
typedef struct { int avail; int mvx, mvy; } MvCand;

static int ref_poc(const Enc *e, int list, int idx) { return list ? e->in->l1_poc[idx] : e->in->l0_poc[idx]; }
/* motion of list X of the block at (x, y): pred flag, ref idx, mv */
static int blk_motion(const ks265_cu8 *b, int list, int *ref_idx, int *mvx, int *mvy)
{
    if (b->pred_mode != 0) return 0;
    if (list == 0) { if (!(b->inter_dir & 1)) return 0; *ref_idx = (b->inter_dir >> 4) & 3; *mvx = b->mvx; *mvy = b->mvy; return 1; }
    if (!(b->inter_dir & 2)) return 0;
    *ref_idx = (b->inter_dir >> 6) & 3; *mvx = b->mv1x; *mvy = b->mv1y;      /* round 5: several pictures in list 1 too (inter_dir = direction | idx0 << 4 | idx1 << 6) */
    return 1;
}
static int clip3i(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static void scale_mv(int cur_poc, int poc_cand_ref, int poc_target_ref, int *mvx, int *mvy)
{
    const int td = clip3i(-128, 127, cur_poc - poc_cand_ref), tb = clip3i(-128, 127, cur_poc - poc_target_ref);
    if (td == tb || td == 0) return;
    const int tx = (16384 + (td < 0 ? -td : td) / 2) / td;
    const int dsf = clip3i(-4096, 4095, (tb * tx + 32) >> 6);
    int v = dsf * *mvx; *mvx = clip3i(-32768, 32767, (v < 0 ? -1 : 1) * (((v < 0 ? -v : v) + 127) >> 8));
    v = dsf * *mvy; *mvy = clip3i(-32768, 32767, (v < 0 ? -1 : 1) * (((v < 0 ? -v : v) + 127) >> 8));
}
/* candidate from neighbour k for (list X, target POC): same-picture vectors first (X then Y); scaled = allow a different picture */
static int nb_cand(const Enc *e, const ks265_cu8 *b, int listx, int target_poc, int scaled, int *mvx, int *mvy)
{
    for (int t = 0; t < 2; ++t) {
        const int l = t ? !listx : listx;
        int ri, mx, my;
        if (!blk_motion(b, l, &ri, &mx, &my)) continue;
        const int p = ref_poc(e, l, ri);
        if (!scaled) { if (p == target_poc) { *mvx = mx; *mvy = my; return 1; } }
        else { scale_mv(e->in->poc, p, target_poc, &mx, &my); *mvx = mx; *mvy = my; return 1; }
    }
    return 0;
}
static void amvp(const Enc *e, int xc, int yc, int cs, int x, int y, int w, int h, int listx, int ref_idx, int cand[2][2])      /* (xc, yc, cs): the coding block; (x, y, w, h): the prediction block */
{
    const int target = ref_poc(e, listx, ref_idx);
    const int nbx[5] = {x - 1, x - 1, x + w, x + w - 1, x - 1}, nby[5] = {y + h, y + h - 1, y - 1, y - 1, y - 1};   /* A0 A1 B0 B1 B2 */
    int av[5];
    const ks265_cu8 *nb[5];
    for (int k = 0; k < 5; ++k) {
        av[k] = avail_pb(e, xc, yc, cs, x, y, nbx[k], nby[k]);
        nb[k] = av[k] ? cu_at(e, nbx[k], nby[k]) : NULL;
        if (av[k] && is_intra(nb[k])) av[k] = 0;
    }
    MvCand a = {0, 0, 0}, b = {0, 0, 0};
    const int is_scaled = av[0] || av[1];
    for (int k = 0; k < 2 && !a.avail; ++k) if (av[k]) a.avail = nb_cand(e, nb[k], listx, target, 0, &a.mvx, &a.mvy);
    for (int k = 0; k < 2 && !a.avail; ++k) if (av[k]) a.avail = nb_cand(e, nb[k], listx, target, 1, &a.mvx, &a.mvy);
    for (int k = 2; k < 5 && !b.avail; ++k) if (av[k]) b.avail = nb_cand(e, nb[k], listx, target, 0, &b.mvx, &b.mvy);
    if (!is_scaled && b.avail) a = b;
    if (!is_scaled) {
        b.avail = 0;
        for (int k = 2; k < 5 && !b.avail; ++k) if (av[k]) b.avail = nb_cand(e, nb[k], listx, target, 1, &b.mvx, &b.mvy);
    }
    int n = 0;
    if (a.avail) { cand[n][0] = a.mvx; cand[n][1] = a.mvy; ++n; }
    if (b.avail && !(a.avail && a.mvx == b.mvx && a.mvy == b.mvy)) { cand[n][0] = b.mvx; cand[n][1] = b.mvy; ++n; }
    for (; n < 2; ++n) { cand[n][0] = 0; cand[n][1] = 0; }
}


/* ------------------------------------------------------------------ merge mode (8.5.3.2.2 .. 8.5.3.2.5; no temporal candidate)
 * Pure signalling: the pixel path chose a vector per CU; when that motion equals one of the normative merge candidates the CU is written
 * with merge_idx instead of ref_idx / mvd / mvp (and as a skipped CU when it has no residual) - the decoder reconstructs the same samples. */
typedef struct { int pf[2], ri[2], mv[2][2]; } Motion;
static void blk_motion_all(const ks265_cu8 *b, Motion *m)
{
    memset(m, 0, sizeof *m);
    for (int l = 0; l < 2; ++l) { int ri = 0, x = 0, y = 0; m->pf[l] = blk_motion(b, l, &ri, &x, &y); if (m->pf[l]) { m->ri[l] = ri; m->mv[l][0] = x; m->mv[l][1] = y; } }
}
static int same_motion(const Motion *a, const Motion *b)
{
    for (int l = 0; l < 2; ++l) {
        if (a->pf[l] != b->pf[l]) return 0;
        if (a->pf[l] && (a->ri[l] != b->ri[l] || a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1])) return 0;
    }
    return 1;
}
#define MAX_MERGE 5
/* (x, y, w, h): the prediction block; part / part_idx: of a CU in two partitions the second one may not merge back into the first (8.5.3.2.3: A1 is out for Nx2N,
 * B1 for 2NxN) - merging both would be the 2Nx2N CU */
static int merge_candidates(const Enc *e, int xc, int yc, int cs, int x, int y, int w, int h, int part, int part_idx, Motion cand[MAX_MERGE])
{
    const int nx[5] = {x - 1, x + w - 1, x + w, x - 1, x - 1}, ny[5] = {y + h - 1, y - 1, y - 1, y + h, y - 1};   /* A1 B1 B0 A0 B2 */
    Motion m[5]; int av[5];
    for (int k = 0; k < 5; ++k) {
        av[k] = avail_pb(e, xc, yc, cs, x, y, nx[k], ny[k]);
        if (part_idx == 1 && ((part == 2 && k == 0) || (part == 1 && k == 1))) av[k] = 0;
        if (av[k]) { const ks265_cu8 *b = cu_at(e, nx[k], ny[k]); if (is_intra(b)) av[k] = 0; else blk_motion_all(b, &m[k]); }
    }
    /* pruning compares with the neighbouring BLOCK (available and inter), whether or not that block made it into the list itself */
    int fl[5];
    fl[0] = av[0];
    fl[1] = av[1] && !(av[0] && same_motion(&m[1], &m[0]));                               /* B1 vs A1 */
    fl[2] = av[2] && !(av[1] && same_motion(&m[2], &m[1]));                               /* B0 vs B1 */
    fl[3] = av[3] && !(av[0] && same_motion(&m[3], &m[0]));                               /* A0 vs A1 */
    fl[4] = av[4] && !(av[0] && same_motion(&m[4], &m[0])) && !(av[1] && same_motion(&m[4], &m[1])) && fl[0] + fl[1] + fl[2] + fl[3] != 4;
    int n = 0;
    for (int k = 0; k < 5 && n < MAX_MERGE; ++k) if (fl[k]) cand[n++] = m[k];
    const int isb = e->in->slice_type == KS265_SLICE_B;
    if (isb && n > 1 && n < MAX_MERGE) {                                                  /* combined bi-predictive candidates */
        static const int l0i[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1i[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
        const int norig = n;
        for (int c = 0; c < norig * (norig - 1) && n < MAX_MERGE; ++c) {
            const Motion *a = &cand[l0i[c]], *b = &cand[l1i[c]];
            if (a->pf[0] && b->pf[1] && (ref_poc(e, 0, a->ri[0]) != ref_poc(e, 1, b->ri[1]) || a->mv[0][0] != b->mv[1][0] || a->mv[0][1] != b->mv[1][1])) {
                Motion *o = &cand[n++];
                memset(o, 0, sizeof *o);
                o->pf[0] = 1; o->ri[0] = a->ri[0]; o->mv[0][0] = a->mv[0][0]; o->mv[0][1] = a->mv[0][1];
                o->pf[1] = 1; o->ri[1] = b->ri[1]; o->mv[1][0] = b->mv[1][0]; o->mv[1][1] = b->mv[1][1];
            }
        }
    }
    const int nref = isb ? (e->in->num_l0 < e->in->num_l1 ? e->in->num_l0 : e->in->num_l1) : e->in->num_l0;
    for (int z = 0; n < MAX_MERGE; ++z) {                                                 /* zero candidates */
        Motion *o = &cand[n++];
        memset(o, 0, sizeof *o);
        o->pf[0] = 1; o->ri[0] = z < nref ? z : 0;
        if (isb) { o->pf[1] = 1; o->ri[1] = z < nref ? z : 0; }
    }
    return n;
}

static int mvd_bits(int d)                                            /* bins of one mvd component (for the candidate choice only) */
{
    const int a = d < 0 ? -d : d;
    if (a == 0) return 1;
    if (a == 1) return 3;
    int v = a - 2, k = 1, n = 3;
    while (v >= (1 << k)) { v -= 1 << k; ++k; ++n; }
    return n + 1 + k;
}
static void mvd_coding(Cabac *c, int dx, int dy)                      /* 7.3.8.9 */
{
    const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    cb_bin(c, CX_MVD, ax > 0);
    cb_bin(c, CX_MVD, ay > 0);
    if (ax > 0) cb_bin(c, CX_MVD + 1, ax > 1);
    if (ay > 0) cb_bin(c, CX_MVD + 1, ay > 1);
    for (int k = 0; k < 2; ++k) {
        const int a = k ? ay : ax, d = k ? dy : dx;
        if (a > 0) {
            if (a > 1) {                                             /* abs_mvd_minus2: EG1, bypass */
                unsigned v = (unsigned)(a - 2); int kk = 1;
                while (v >= (1u << kk)) { cb_bypass(c, 1); v -= 1u << kk; ++kk; }
                cb_bypass(c, 0);
                cb_bypass_bits(c, v, kk);
            }
            cb_bypass(c, d < 0);
        }
    }
}

/* ------------------------------------------------------------------ coding quadtree */
static void transform_unit(Enc *e, int x, int y, int log2, int cbf_y, int cbf_cb, int cbf_cr, int intra, int luma_mode)
{
    if ((cbf_y | cbf_cb | cbf_cr) && e->cfg->cu_qp_delta && !e->dqp_coded) {
        /* cu_qp_delta_abs: prefix truncated unary (cMax 5; context 0 for the first bin, 1 for the others), suffix EG0 in bypass; then the sign (7.3.8.14, 9.3.3.10) */
        Cabac *c = &e->c;
        const int d = e->qp_want - e->qp_prev, a = d < 0 ? -d : d;
        if (d < -26 || d > 25) c->overflow = 1;                          /* outside CuQpDeltaVal's range (7.4.9.14): the slice is refused (KS265_NOTSUPPORTED) */
        const int pre = a < 5 ? a : 5;
        for (int i = 0; i < pre; ++i) cb_bin(c, CX_DQP + (i > 0), 1);
        if (pre < 5) cb_bin(c, CX_DQP + (pre > 0), 0);
        else {
            int v = a - 5, k = 0;
            while (v >= (1 << k)) { cb_bypass(c, 1); v -= 1 << k; ++k; }
            cb_bypass(c, 0);
            while (k--) cb_bypass(c, (v >> k) & 1);
        }
        if (a) cb_bypass(c, d < 0);
        e->dqp_coded = 1;
    }
    if (cbf_y) {
        int scan = 0;
        if (intra && (log2 == 2 || log2 == 3)) scan = (luma_mode >= 6 && luma_mode <= 14) ? 2 : (luma_mode >= 22 && luma_mode <= 30) ? 1 : 0;
        residual_coding(e, e->in->lvl[0] + (long)y * e->W + x, e->W, log2, 0, scan);
    }
    for (int ci = 1; ci < 3; ++ci) {
        if (!(ci == 1 ? cbf_cb : cbf_cr)) continue;
        int scan = 0;
        if (intra && log2 - 1 == 2) scan = (luma_mode >= 6 && luma_mode <= 14) ? 2 : (luma_mode >= 22 && luma_mode <= 30) ? 1 : 0;   /* chroma mode = luma mode (DM) */
        residual_coding(e, e->in->lvl[ci] + (long)(y >> 1) * (e->W >> 1) + (x >> 1), e->W >> 1, log2 - 1, ci, scan);
    }
}

static int coding_unit(Enc *e, int x, int y, int log2)
{
    Cabac *c = &e->c;
    const ks265_cu8 *cu = cu_at(e, x, y);
    const int size = 1 << log2, intra = is_intra(cu), st = e->in->slice_type;
    const int pm = intra ? 0 : (cu->log2_cu >> 4) & 3;
    const int part = pm == 3 ? 0 : pm;                               /* 0 = 2Nx2N, 1 = 2NxN, 2 = Nx2N (ks265_frame_cfg.part): two prediction units, four transform units */
    const int tsplit = pm == 3;                                      /* 2Nx2N with four transform units (ks265_frame_cfg.tu_inter: split_transform_flag = 1) */
    const int rqt = e->cfg->tu_inter && !intra && log2 <= 5;         /* max_transform_hierarchy_depth_inter = 1: split_transform_flag is coded at depth 0 (no interSplitFlag inference) */
    if (tsplit && (!e->cfg->tu_inter || log2 > 5 || log2 < 4)) return KS265_NOTSUPPORTED;
    if (cu->pred_mode == 1) return KS265_NOTSUPPORTED;               /* the flat-128 stand-in is not an HEVC prediction mode */
    if (intra && log2 > 5) return KS265_NOTSUPPORTED;
    if (part > 2 || (part && log2 < 4)) return KS265_NOTSUPPORTED;   /* (an 8x8 CU in two partitions would need vectors per 8x4 block) */
    int merge_idx = -1;
    if (st != KS265_SLICE_I) {
        int skip = 0;
        if (!intra && !part) {
            Motion cand[MAX_MERGE], mine;
            const int n = merge_candidates(e, x, y, size, x, y, size, size, 0, 0, cand);
            blk_motion_all(cu, &mine);
            for (int k = 0; k < n && merge_idx < 0; ++k) if (same_motion(&mine, &cand[k])) merge_idx = k;
            int any = cu->cbf & 7;
            if (log2 == 6 || tsplit) for (int k = 1; k < 4; ++k) any |= cu_at(e, x + (k & 1) * (size >> 1), y + (k >> 1) * (size >> 1))->cbf & 7;
            skip = merge_idx >= 0 && !any;
        }
        const int inc = (x > 0 && e->skip[(long)(y >> 3) * e->w8 + ((x - 1) >> 3)]) + (y > 0 && e->skip[(long)((y - 1) >> 3) * e->w8 + (x >> 3)]);
        CAT(c, CAT_CU);
        cb_bin(c, CX_SKIP + inc, skip);                               /* cu_skip_flag */
        if (skip) {
            CAT(c, CAT_MERGE);
            for (int by = 0; by < size >> 3; ++by) memset(e->skip + (long)((y >> 3) + by) * e->w8 + (x >> 3), 1, (size_t)(size >> 3));
            cb_bin(c, CX_MERGE_IDX, merge_idx > 0);                   /* merge_idx: TR, cMax = 4, first bin with context */
            for (int k = 1; k < MAX_MERGE - 1 && k <= merge_idx; ++k) cb_bypass(c, merge_idx > k);
            CAT(c, CAT_CU);
            return 0;
        }
        cb_bin(c, CX_PRED_MODE, intra);
    } else if (!intra) return KS265_NOTSUPPORTED;
    if (part) { cb_bin(c, CX_PART_MODE, 0); cb_bin(c, CX_PART_MODE + 1, part == 1); }   /* part_mode (9.3.3.7, no AMP, above the minimum CU size): 2NxN = 01, Nx2N = 00 */
    else if (!intra || log2 == 3) cb_bin(c, CX_PART_MODE, 1);        /* PART_2Nx2N */
    if (intra) {
        const int mode = cu->mvx;
        /* most probable modes (8.4.2) */
        int ca = 1, cb = 1;
        if (x > 0) { const ks265_cu8 *n = cu_at(e, x - 1, y); if (is_intra(n)) ca = n->mvx; }
        if (y > 0 && (y & 63)) { const ks265_cu8 *n = cu_at(e, x, y - 1); if (is_intra(n)) cb = n->mvx; }
        int mpm[3];
        if (ca == cb) {
            if (ca < 2) { mpm[0] = 0; mpm[1] = 1; mpm[2] = 26; }
            else { mpm[0] = ca; mpm[1] = 2 + ((ca + 29) % 32); mpm[2] = 2 + ((ca - 2 + 1) % 32); }
        } else {
            mpm[0] = ca; mpm[1] = cb;
            mpm[2] = (ca != 0 && cb != 0) ? 0 : (ca != 1 && cb != 1) ? 1 : 26;
        }
        int idx = -1;
        for (int k = 0; k < 3; ++k) if (mpm[k] == mode) idx = k;
        CAT(c, CAT_INTRA);
        cb_bin(c, CX_PREV_INTRA, idx >= 0);
        if (idx >= 0) { cb_bypass(c, idx > 0); if (idx > 0) cb_bypass(c, idx > 1); }
        else {
            if (mpm[0] > mpm[1]) { const int t = mpm[0]; mpm[0] = mpm[1]; mpm[1] = t; }
            if (mpm[0] > mpm[2]) { const int t = mpm[0]; mpm[0] = mpm[2]; mpm[2] = t; }
            if (mpm[1] > mpm[2]) { const int t = mpm[1]; mpm[1] = mpm[2]; mpm[2] = t; }
            int rem = mode;
            for (int k = 2; k >= 0; --k) if (rem > mpm[k]) --rem;
            cb_bypass_bits(c, (uint32_t)rem, 5);
        }
        cb_bin(c, CX_CHROMA_PRED, 0);                                /* intra_chroma_pred_mode = 4: derived from luma */
        CAT(c, CAT_CU);
    } else {
        for (int pi = 0; pi < (part ? 2 : 1); ++pi) {                /* prediction_unit(): 7.3.8.6 */
        const int xp = x + (part == 2 && pi ? size >> 1 : 0), yp = y + (part == 1 && pi ? size >> 1 : 0), wp = part == 2 ? size >> 1 : size, hp = part == 1 ? size >> 1 : size;
        const ks265_cu8 *pc = cu_at(e, xp, yp);
        int midx = merge_idx;
        if (part) {                                                  /* does this partition's motion equal one of ITS merge candidates? */
            Motion cand[MAX_MERGE], mine;
            const int n = merge_candidates(e, x, y, size, xp, yp, wp, hp, part, pi, cand);
            blk_motion_all(pc, &mine);
            midx = -1;
            for (int k = 0; k < n && midx < 0; ++k) if (same_motion(&mine, &cand[k])) midx = k;
        }
        CAT(c, CAT_MERGE);
        cb_bin(c, CX_MERGE_FLAG, midx >= 0);
        const int dir = pc->inter_dir & 3;
        if (midx >= 0) {
            cb_bin(c, CX_MERGE_IDX, midx > 0);
            for (int k = 1; k < MAX_MERGE - 1 && k <= midx; ++k) cb_bypass(c, midx > k);
        } else {
        CAT(c, CAT_MOTION);
        if (st == KS265_SLICE_B) {
            /* inter_pred_idc: nPbW + nPbH != 12 always (prediction blocks of at least 16x8 / 8x16 / 8x8) */
            const int depth = 6 - log2;
            cb_bin(c, CX_INTER_DIR + depth, dir == 3);
            if (dir != 3) cb_bin(c, CX_INTER_DIR + 4, dir == 2);
        } else if (dir != 1) return KS265_NOTSUPPORTED;
        for (int l = 0; l < 2; ++l) {
            if (!(dir & (1 << l))) continue;
            const int nact = l ? e->in->num_l1 : e->in->num_l0;
            const int ri = l ? (pc->inter_dir >> 6) & 3 : (pc->inter_dir >> 4) & 3;
            if (ri >= nact) return KS265_NOTSUPPORTED;
            if (nact > 1) {                                          /* ref_idx_lX: TR, cMax = nact - 1, two context bins then bypass */
                for (int k = 0; k < nact - 1; ++k) {
                    const int bin = k < ri;
                    if (k < 2) cb_bin(c, CX_REF_IDX + k, bin); else cb_bypass(c, bin);
                    if (!bin) break;
                }
            }
            int cand[2][2];
            amvp(e, x, y, size, xp, yp, wp, hp, l, ri, cand);
            const int mvx = l ? pc->mv1x : pc->mvx, mvy = l ? pc->mv1y : pc->mvy;
            const int b0 = mvd_bits(mvx - cand[0][0]) + mvd_bits(mvy - cand[0][1]), b1 = mvd_bits(mvx - cand[1][0]) + mvd_bits(mvy - cand[1][1]);
            const int pick = b1 < b0;
            mvd_coding(c, mvx - cand[pick][0], mvy - cand[pick][1]);
            cb_bin(c, CX_MVP, pick);
        }
        }
        }
        CAT(c, CAT_CU);
    }
    /* transform tree (7.3.8.8): max_transform_hierarchy_depth = 0 -> one TU per CU, except 64x64 CUs (four 32x32 TUs, split inferred: above the maximum TU size) and
     * inter CUs in two partitions (four TUs of half the size: interSplitFlag, split inferred as well) */
    if (log2 == 6 || part || tsplit) {
        const int hs = size >> 1;
        int any = 0, cby[4], ccb[4], ccr[4], acb = 0, acr = 0;
        for (int k = 0; k < 4; ++k) {
            const ks265_cu8 *q = cu_at(e, x + (k & 1) * hs, y + (k >> 1) * hs);
            cby[k] = q->cbf & 1; ccb[k] = (q->cbf >> 1) & 1; ccr[k] = (q->cbf >> 2) & 1;
            any |= q->cbf & 7; acb |= ccb[k]; acr |= ccr[k];
        }
        if (merge_idx < 0 || part) { cb_bin(c, CX_ROOT_CBF, any != 0); if (!any) return 0; }      /* rqt_root_cbf: inferred 1 only for a merged 2Nx2N CU (which has residual, else it was skipped) */
        if (rqt) cb_bin(c, CX_SPLIT_TU + 5 - log2, 1);               /* split_transform_flag (ctxInc = 5 - log2TrafoSize): explicit with the residual quadtree on, for partitioned CUs too */
        cb_bin(c, CX_CBF_CHROMA + 0, acb);
        cb_bin(c, CX_CBF_CHROMA + 0, acr);
        for (int k = 0; k < 4; ++k) {
            if (acb) cb_bin(c, CX_CBF_CHROMA + 1, ccb[k]);
            if (acr) cb_bin(c, CX_CBF_CHROMA + 1, ccr[k]);
            cb_bin(c, CX_CBF_LUMA + 0, cby[k]);                      /* trafoDepth 1 -> ctxInc 0 */
            transform_unit(e, x + (k & 1) * hs, y + (k >> 1) * hs, log2 - 1, cby[k], ccb[k], ccr[k], 0, 0);
        }
        return 0;
    }
    const int cbf_y = cu->cbf & 1, cbf_cb = (cu->cbf >> 1) & 1, cbf_cr = (cu->cbf >> 2) & 1;
    if (!intra && merge_idx < 0) {
        cb_bin(c, CX_ROOT_CBF, (cu->cbf & 7) != 0);
        if (!(cu->cbf & 7)) return 0;
    }
    if (rqt) cb_bin(c, CX_SPLIT_TU + 5 - log2, 0);                   /* one transform unit (8x8 CUs are never split: their chroma would sit at the parent) */
    cb_bin(c, CX_CBF_CHROMA + 0, cbf_cb);
    cb_bin(c, CX_CBF_CHROMA + 0, cbf_cr);
    if (intra || cbf_cb || cbf_cr) cb_bin(c, CX_CBF_LUMA + 1, cbf_y); /* trafoDepth 0 -> ctxInc 1; else inferred 1 */
    transform_unit(e, x, y, log2, cbf_y, cbf_cb, cbf_cr, intra, intra ? cu->mvx : 0);
    return 0;
}

static int coding_quadtree(Enc *e, int x, int y, int log2)
{
    Cabac *c = &e->c;
    const int size = 1 << log2;
    int split;
    if (x + size <= e->W && y + size <= e->H && log2 > 3) {
        const ks265_cu8 *cu = cu_at(e, x, y);
        if ((cu->log2_cu & 15) > log2 || (cu->log2_cu & 15) < 3) return KS265_NOTSUPPORTED;
        split = (cu->log2_cu & 15) < log2;
        const int depth = 6 - log2;
        int inc = 0;
        if (x > 0 && 6 - (cu_at(e, x - 1, y)->log2_cu & 15) > depth) ++inc;
        if (y > 0 && 6 - (cu_at(e, x, y - 1)->log2_cu & 15) > depth) ++inc;
        cb_bin(c, CX_SPLIT_CU + inc, split);
    } else split = log2 > 3;
    if (!split) return coding_unit(e, x, y, log2);
    const int h = size >> 1;
    for (int k = 0; k < 4; ++k) {
        const int xx = x + (k & 1) * h, yy = y + (k >> 1) * h;
        if (xx < e->W && yy < e->H) { const int r = coding_quadtree(e, xx, yy, log2 - 1); if (r) return r; }
    }
    return 0;
}

/* ------------------------------------------------------------------ slice segment header (7.3.6.1) + data */
/* 8.3.4: RefPicListTemp0 = StCurrBefore (closest first), StCurrAfter (closest first); list 1 the other way round.  Returns the number of
 * pictures the slice predicts from (NumPicTotalCurr) and the two temporary lists. */
static int temp_lists(const ks265_slice_in *in, int *t0, int *t1)
{
    int before[16], after[16], nb = 0, na = 0;
    for (int i = 0; i < in->num_rps; ++i) {
        if (!in->rps_used[i]) continue;
        if (in->rps_poc[i] < in->poc) before[nb++] = in->rps_poc[i]; else after[na++] = in->rps_poc[i];
    }
    for (int i = 0; i < nb; ++i) for (int j = i + 1; j < nb; ++j) if (before[j] > before[i]) { const int t = before[i]; before[i] = before[j]; before[j] = t; }
    for (int i = 0; i < na; ++i) for (int j = i + 1; j < na; ++j) if (after[j] < after[i]) { const int t = after[i]; after[i] = after[j]; after[j] = t; }
    for (int k = 0; k < nb + na; ++k) { t0[k] = k < nb ? before[k] : after[k - nb]; t1[k] = k < na ? after[k] : before[k - na]; }
    return nb + na;
}

/* list entries of ref_pic_lists_modification() (7.3.6.2) for one list: ent[i] = index into the temporary list; returns 0 = the default
 * construction already gives the list, 1 = entries needed, -1 = a picture of the list is not a used picture of the RPS */
static int list_entries(const int *want, int n, const int *temp, int tot, int *ent)
{
    int mod = 0;
    for (int i = 0; i < n; ++i) {
        int k = 0;
        while (k < tot && temp[k] != want[i]) ++k;
        if (k == tot) return -1;
        ent[i] = k;
        if (k != i % tot) mod = 1;
    }
    return mod;
}

static int check_lists(const ks265_stream_cfg *cfg, const ks265_slice_in *in)
{
    if (in->slice_type == KS265_SLICE_I) return 1;
    int t0[16], t1[16], ent[4];
    const int tot = temp_lists(in, t0, t1);
    if (tot == 0 || in->num_l0 < 1 || in->num_l0 > 4) return 0;
    int m = list_entries(in->l0_poc, in->num_l0, t0, tot, ent);
    if (m < 0 || (m && !cfg->list_mod)) return 0;
    if (in->slice_type == KS265_SLICE_B) {
        if (in->num_l1 < 1 || in->num_l1 > 4) return 0;
        m = list_entries(in->l1_poc, in->num_l1, t1, tot, ent);
        if (m < 0 || (m && !cfg->list_mod)) return 0;
    }
    return 1;
}

/* argument checks shared by the whole-slice and the row-wise entry points */
static int slice_args_ok(const ks265_stream_cfg *cfg, const ks265_slice_in *in)
{
    if (!cfg || !in || !in->cu8 || !in->lvl[0] || !in->lvl[1] || !in->lvl[2]) return KS265_POINTER;
    if (!cfg_ok(cfg) || in->qp < 0 || in->qp > 51 || in->num_rps < 0 || in->num_rps > 15) return KS265_NOTSUPPORTED;
    if (in->slice_type < 0 || in->slice_type > 2) return KS265_NOTSUPPORTED;
    const int idr = in->nal_type == KS265_NAL_IDR_W_RADL || in->nal_type == KS265_NAL_IDR_N_LP;
    if (idr && (in->slice_type != KS265_SLICE_I || in->poc != 0)) return KS265_NOTSUPPORTED;
    if (!idr && !check_lists(cfg, in)) return KS265_NOTSUPPORTED;
    return KS265_OK;
}

/* slice_segment_header() up to and including byte_alignment(); nentry > 0: entry points of the CTU-row substreams (entry_bytes[i] = coded size of
 * substream i INCLUDING its emulation prevention bytes, 7.4.7.1) */
static int write_slice_header(BitW *bp, const ks265_stream_cfg *cfg, const ks265_slice_in *in, int nentry, const long *entry_bytes)
{
    BitW b = *bp;
    const int idr = in->nal_type == KS265_NAL_IDR_W_RADL || in->nal_type == KS265_NAL_IDR_N_LP;
    const int sao_on = cfg->sao && in->sao != NULL;
    bw_put(&b, 1, 1);                                                /* first_slice_segment_in_pic_flag */
    if (in->nal_type >= 16 && in->nal_type <= 23) bw_put(&b, 0, 1);  /* no_output_of_prior_pics_flag */
    bw_ue(&b, 0);                                                    /* slice_pic_parameter_set_id */
    bw_ue(&b, (uint32_t)in->slice_type);
    if (!idr) {
        bw_put(&b, (uint32_t)in->poc & ((1u << cfg->log2_max_poc_lsb) - 1), cfg->log2_max_poc_lsb);
        bw_put(&b, 0, 1);                                            /* short_term_ref_pic_set_sps_flag */
        /* st_ref_pic_set(0) without inter RPS prediction (7.3.7) */
        int neg[16], pos[16], nused[16], pused[16], nn = 0, np = 0;
        for (int i = 0; i < in->num_rps; ++i) {
            if (in->rps_poc[i] < in->poc) { neg[nn] = in->rps_poc[i]; nused[nn++] = in->rps_used[i]; }
            else if (in->rps_poc[i] > in->poc) { pos[np] = in->rps_poc[i]; pused[np++] = in->rps_used[i]; }
            else return KS265_NOTSUPPORTED;
        }
        for (int i = 0; i < nn; ++i) for (int j = i + 1; j < nn; ++j) if (neg[j] > neg[i]) { int t = neg[i]; neg[i] = neg[j]; neg[j] = t; t = nused[i]; nused[i] = nused[j]; nused[j] = t; }
        for (int i = 0; i < np; ++i) for (int j = i + 1; j < np; ++j) if (pos[j] < pos[i]) { int t = pos[i]; pos[i] = pos[j]; pos[j] = t; t = pused[i]; pused[i] = pused[j]; pused[j] = t; }
        bw_ue(&b, (uint32_t)nn);
        bw_ue(&b, (uint32_t)np);
        int prev = in->poc;
        for (int i = 0; i < nn; ++i) { bw_ue(&b, (uint32_t)(prev - neg[i] - 1)); bw_put(&b, (uint32_t)nused[i], 1); prev = neg[i]; }
        prev = in->poc;
        for (int i = 0; i < np; ++i) { bw_ue(&b, (uint32_t)(pos[i] - prev - 1)); bw_put(&b, (uint32_t)pused[i], 1); prev = pos[i]; }
    }
    if (cfg->sao) { bw_put(&b, (uint32_t)sao_on, 1); bw_put(&b, (uint32_t)sao_on, 1); }   /* slice_sao_luma_flag, slice_sao_chroma_flag */
    if (in->slice_type != KS265_SLICE_I) {
        const int over = in->num_l0 != 1 || (in->slice_type == KS265_SLICE_B && in->num_l1 != 1);
        bw_put(&b, (uint32_t)over, 1);                               /* num_ref_idx_active_override_flag */
        if (over) { bw_ue(&b, (uint32_t)(in->num_l0 - 1)); if (in->slice_type == KS265_SLICE_B) bw_ue(&b, (uint32_t)(in->num_l1 - 1)); }
        int t0[16], t1[16], ent[4];
        const int tot = temp_lists(in, t0, t1);
        if (cfg->list_mod && tot > 1) {                              /* ref_pic_lists_modification(): list_entry_lX is u(Ceil(Log2(NumPicTotalCurr))) */
            int len = 0;
            while ((1 << len) < tot) ++len;
            for (int x = 0; x < (in->slice_type == KS265_SLICE_B ? 2 : 1); ++x) {
                const int n = x ? in->num_l1 : in->num_l0;
                const int m = list_entries(x ? in->l1_poc : in->l0_poc, n, x ? t1 : t0, tot, ent);
                bw_put(&b, (uint32_t)(m > 0), 1);                    /* ref_pic_list_modification_flag_lX */
                if (m > 0) for (int i = 0; i < n; ++i) bw_put(&b, (uint32_t)ent[i], len);
            }
        }
        if (in->slice_type == KS265_SLICE_B) bw_put(&b, 0, 1);       /* mvd_l1_zero_flag */
        bw_ue(&b, 5 - MAX_MERGE);                                    /* five_minus_max_num_merge_cand */
    }
    bw_se(&b, in->qp - 26);                                          /* slice_qp_delta */
    if (cfg->wpp) {                                                  /* entropy_coding_sync_enabled_flag: the CTU rows are substreams with entry points */
        bw_ue(&b, (uint32_t)nentry);                                 /* num_entry_point_offsets */
        if (nentry > 0) {
            long mx = 1;
            for (int i = 0; i < nentry; ++i) if (entry_bytes[i] > mx) mx = entry_bytes[i];
            int len = 1;
            while (len < 32 && ((mx - 1) >> len)) ++len;
            bw_ue(&b, (uint32_t)(len - 1));                          /* offset_len_minus1 */
            for (int i = 0; i < nentry; ++i) {                       /* entry_point_offset_minus1[i], u(len): possibly more than 25 bits -> two pieces */
                const uint32_t v = (uint32_t)(entry_bytes[i] - 1);
                if (len > 16) { bw_put(&b, v >> 16, len - 16); bw_put(&b, v & 0xFFFFu, 16); } else bw_put(&b, v, len);
            }
        }
    }
    bw_put(&b, 1, 1);                                                /* byte_alignment(): alignment_bit_equal_to_one, then zeros */
    while (b.nacc) bw_put(&b, 0, 1);
    *bp = b;
    return b.overflow ? KS265_NOTSUPPORTED : KS265_OK;
}

static void enc_setup(Enc *e, const ks265_stream_cfg *cfg, const ks265_slice_in *in, uint8_t *skip)
{
    e->cfg = cfg; e->in = in; e->W = cfg->width; e->H = cfg->height; e->w8 = e->W >> 3; e->h8 = e->H >> 3;
    e->ctb_cols = (e->W + 63) >> 6; e->ctb_rows = (e->H + 63) >> 6;
    e->skip = skip;
}

/* ------------------------------------------------------------------ CTU rows as substreams (entropy_coding_sync_enabled_flag = 1; the reference's WPP,
 * CCtuEncWpp::processOneCtu enc@0x46f5f0): every row is coded by its own arithmetic coder whose contexts start from the state the row above had
 * after its second CTU (9.3.2.2 / 9.3.2.4); rows of one picture can therefore be written by different host threads, each at most two CTUs behind the
 * row above.  Shared between the rows: the skip-flag map (context of cu_skip_flag) and the progress counters. */
#include <stdatomic.h>
#include <sched.h>
typedef struct {
    const ks265_stream_cfg *cfg; const ks265_slice_in *in;
    int rows, cols, error;
    size_t row_cap;
    uint8_t *skip, *rowbuf;
    long *row_len;
    atomic_int *progress;                  /* CTUs of the row that are finished (contexts of the second one saved before the count passes 2) */
    uint8_t *snap;                         /* rows x CX_COUNT */
    Scans scans;
} Wpp;

static size_t wpp_row_cap(const ks265_stream_cfg *cfg) { return (size_t)((cfg->width + 63) >> 6) * (64 * 64 * 3 / 2) * 3 + 4096; }
size_t ks265_wpp_bytes(const ks265_stream_cfg *cfg)
{
    const size_t rows = (size_t)((cfg->height + 63) >> 6);
    return ((sizeof(Wpp) + 63) & ~(size_t)63) + (size_t)(cfg->width >> 3) * (size_t)(cfg->height >> 3) + rows * (wpp_row_cap(cfg) + sizeof(long) + sizeof(atomic_int) + CX_COUNT) + 256;
}
int ks265_wpp_begin(const ks265_stream_cfg *cfg, const ks265_slice_in *in, void *mem)
{
    if (!mem) return KS265_POINTER;
    const int r = slice_args_ok(cfg, in);
    if (r) return r;
    if (!cfg->wpp) return KS265_NOTSUPPORTED;
    Wpp *w = (Wpp *)mem;
    uint8_t *p = (uint8_t *)mem + ((sizeof(Wpp) + 63) & ~(size_t)63);
    w->cfg = cfg; w->in = in; w->rows = (cfg->height + 63) >> 6; w->cols = (cfg->width + 63) >> 6; w->error = 0; w->row_cap = wpp_row_cap(cfg);
    w->row_len = (long *)p; p += sizeof(long) * (size_t)w->rows;
    w->progress = (atomic_int *)p; p += sizeof(atomic_int) * (size_t)w->rows;
    w->snap = p; p += (size_t)CX_COUNT * (size_t)w->rows;
    w->skip = p; p += (size_t)(cfg->width >> 3) * (size_t)(cfg->height >> 3);
    w->rowbuf = (uint8_t *)(((uintptr_t)p + 63) & ~(uintptr_t)63);
    memset(w->skip, 0, (size_t)(cfg->width >> 3) * (size_t)(cfg->height >> 3));
    for (int i = 0; i < w->rows; ++i) { atomic_init(&w->progress[i], 0); w->row_len[i] = 0; }
    scans_init(&w->scans);
    return KS265_OK;
}
int ks265_wpp_rows(const void *mem) { return mem ? ((const Wpp *)mem)->rows : 0; }

/* one CTU row.  Rows must be STARTED in order (row r - 1 before row r) by whoever distributes them; a row waits (yielding) while the row above is
 * less than two CTUs ahead.  Thread-safe for different rows of the same picture. */
int ks265_wpp_code_row(void *mem, int ry)
{
    Wpp *w = (Wpp *)mem;
    if (!w || ry < 0 || ry >= w->rows) return KS265_POINTER;
    Enc es, *e = &es;
    enc_setup(e, w->cfg, w->in, w->skip);
    e->scans = w->scans;
    const ks265_slice_in *in = w->in;
    const int sao_on = w->cfg->sao && in->sao != NULL;
    cb_init(&e->c, w->rowbuf + (size_t)ry * w->row_cap, w->row_cap, in->slice_type == KS265_SLICE_I ? 0 : in->slice_type == KS265_SLICE_P ? 1 : 2, in->qp);
    int rc = KS265_OK;
    for (int rx = 0; rx < w->cols; ++rx) {
        if (ry > 0) {
            const int need = rx + 2 < w->cols ? rx + 2 : w->cols;
            int spins = 0;
            while (atomic_load_explicit(&w->progress[ry - 1], memory_order_acquire) < need) { if (++spins > 64) { sched_yield(); spins = 0; } }
            if (rx == 0 && w->cols > 1) memcpy(e->c.state, w->snap + (size_t)(ry - 1) * CX_COUNT, CX_COUNT);     /* synchronisation process 9.3.2.4 */
        }
        if (!rc) {
            if (sao_on) sao_ctb(e, rx, ry);
            if (rx == 0) e->qp_prev = in->qp;                          /* first quantisation group of a CTB row under entropy_coding_sync: SliceQpY (8.6.1) */
            e->dqp_coded = 0; e->qp_want = in->qp_map ? in->qp_map[ry * w->cols + rx] : in->qp;
            rc = coding_quadtree(e, rx << 6, ry << 6, 6);
            if (e->dqp_coded) e->qp_prev = e->qp_want;
            const int last = ry == w->rows - 1 && rx == w->cols - 1;
            cb_terminate(&e->c, last);                               /* end_of_slice_segment_flag */
            if (!last && rx == w->cols - 1) cb_terminate(&e->c, 1);  /* end_of_subset_one_bit; byte_alignment() comes with the flush below */
        }
        if (rx == 1) memcpy(w->snap + (size_t)ry * CX_COUNT, e->c.state, CX_COUNT);                               /* storage process 9.3.2.3 */
        atomic_store_explicit(&w->progress[ry], rx + 1, memory_order_release);                                     /* also after an error: nobody may hang */
    }
    cb_finish(&e->c);
    if (e->c.overflow && !rc) rc = KS265_NOTSUPPORTED;
    w->row_len[ry] = (long)e->c.pos;
    if (rc) w->error = rc;
    return rc;
}

static long count_ep(const uint8_t *p, long n)
{
    long cnt = 0; int zeros = 0;
    for (long i = 0; i < n; ++i) {
        if (zeros >= 2 && p[i] <= 3) { ++cnt; zeros = 0; }
        zeros = p[i] == 0 ? zeros + 1 : 0;
    }
    return cnt;
}
/* after every row has been coded: slice header with the entry points, then the substreams; returns the NAL size */
long ks265_wpp_finish(void *mem, uint8_t *out, size_t cap)
{
    Wpp *w = (Wpp *)mem;
    if (!w || !out) return KS265_POINTER;
    if (w->error) return w->error;
    long entry[256];
    if (w->rows > 256) return KS265_NOTSUPPORTED;
    for (int i = 0; i + 1 < w->rows; ++i) entry[i] = w->row_len[i] + count_ep(w->rowbuf + (size_t)i * w->row_cap, w->row_len[i]);
    uint8_t hdr[2048];
    BitW b; bw_init(&b, hdr, sizeof hdr);
    const int r = write_slice_header(&b, w->cfg, w->in, w->rows - 1, entry);
    if (r) return r;
    /* Annex B wrapping of header + substreams as one RBSP (the emulation prevention state carries over the seams; a substream never ends in 00) */
    size_t o = 0; int zeros = 0;
    if (cap < 6) return KS265_NOTSUPPORTED;
    out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = 1; out[o++] = (uint8_t)(w->in->nal_type << 1); out[o++] = 1;
    for (int seg = -1; seg < w->rows; ++seg) {
        const uint8_t *p = seg < 0 ? hdr : w->rowbuf + (size_t)seg * w->row_cap;
        const size_t n = seg < 0 ? b.pos : (size_t)w->row_len[seg];
        if (o + n + n / 2 + 8 > cap) return KS265_NOTSUPPORTED;
        for (size_t i = 0; i < n; ++i) {
            if (zeros >= 2 && p[i] <= 3) { out[o++] = 3; zeros = 0; }
            out[o++] = p[i];
            zeros = p[i] == 0 ? zeros + 1 : 0;
        }
    }
    return (long)o;
}

/* the entropy coder's context states at the end of the slice `scratch` wrote last (without substreams: the whole slice's; with them: those saved after the second CTU of
 * the last CTU row, i.e. after two CTUs of every row) - what a host snapshots into rate tables for the pictures that follow (estBitRdoq enc@0x46a8a0 builds rdoQuant's bit
 * tables from exactly these states).  out: n <= cap bytes, pStateIdx << 1 | valMps each; layout[12] = first index of: cbf_luma, cbf_chroma, coded_sub_block_flag (luma 2,
 * chroma 2), sig_coeff_flag (luma 27, chroma 15), last x prefix (luma 15, chroma 3), last y prefix, greater1 (luma 16, chroma 8), greater2 (luma 4, chroma 2), rqt_root_cbf,
 * then the count.  Returns the number of states. */
int ks265_slice_final_contexts(const ks265_stream_cfg *cfg, const void *scratch, uint8_t *out, int cap, int *layout)
{
    if (!cfg || !scratch || !out) return KS265_POINTER;
    if (cap < CX_COUNT) return KS265_NOTSUPPORTED;
    if (cfg->wpp) { const Wpp *w = (const Wpp *)scratch; memcpy(out, w->snap + (size_t)(w->rows - 1) * CX_COUNT, CX_COUNT); }
    else memcpy(out, ((const Enc *)scratch)->c.state, CX_COUNT);
    if (layout) { const int l[10] = {CX_CBF_LUMA, CX_CBF_CHROMA, CX_CSBF, CX_SIG, CX_LAST_X, CX_LAST_Y, CX_G1, CX_G2, CX_ROOT_CBF, CX_COUNT}; memcpy(layout, l, sizeof l); }
    return CX_COUNT;
}

/* ---- the bit tables of the reference's rate-distortion optimised quantisation (-rdoq 1; round 6).  rdoQuant enc@0x4aac50 prices its decisions with the 180-word tables
 * estBitRdoq enc@0x46a8a0 builds from the entropy coder's context states (TEstBitsSbac: [0..3] coded_sub_block_flag [2][2]; [4..45] / [46..87] sig_coeff_flag = 0 / 1 per context;
 * [88..97] / [98..107] last x / y prefix; [108..139] greater1 [16][2]; [156..163] greater2 [4][2]; [168..177] cbf [5][2]; [178..179] rqt_root_cbf; bits x 2^15).  The states are
 * gathered in the order that function reads them (cbf at 0x0d, + 5 chroma; group flags at 0x1d, + 2; sig flags at 0x21 luma / 0x3c chroma; last x / y at 0x4b / 0x69; greater1 at
 * 0x87 luma / 0x97 chroma; greater2 at 0x9f / 0xa3; root cbf at 0xaa - entries this writer has no state for read state 0, as they do when the function is fed this writer's
 * states in the tests' mirror), the entropy values are the standard's fractional-bit table of the 64 probability states (kEntropyBits: bits x 2^15 of the less probable / more
 * probable symbol per state).  states = what ks265_slice_final_contexts returned for a slice (the tables then follow the stream), or NULL = the initial states of a slice of
 * `slice_type` at `qp`.  tables: [4 sizes 4 .. 32][luma, chroma][180] words, all written (words the function leaves alone are 0). */
static const int32_t kEntropyBits[128] = {
    0x7b23, 0x85f9, 0x74a0, 0x8cbc, 0x6ee4, 0x9354, 0x67f4, 0x9c1b, 0x60b0, 0xa62a, 0x5a9c, 0xaf5b, 0x548d, 0xb955, 0x4f56, 0xc2a9, 0x4a87, 0xcbf7, 0x45d6, 0xd5c3, 0x4144, 0xe01b, 0x3d88, 0xe937,
    0x39e0, 0xf2cd, 0x3663, 0xfc9e, 0x3347, 0x10600, 0x3050, 0x10f95, 0x2d4d, 0x11a02, 0x2ad3, 0x12333, 0x286e, 0x12cad, 0x2604, 0x136df, 0x2425, 0x13f48, 0x21f4, 0x149c4, 0x203e, 0x1527b,
    0x1e4d, 0x15d00, 0x1c99, 0x166de, 0x1b18, 0x17017, 0x19a5, 0x17988, 0x1841, 0x18327, 0x16df, 0x18d50, 0x15d9, 0x19547, 0x147c, 0x1a083, 0x138e, 0x1a8a3, 0x1251, 0x1b418, 0x1166, 0x1bd27,
    0x1068, 0x1c77b, 0xf7f, 0x1d18e, 0xeda, 0x1d91a, 0xe19, 0x1e254, 0xd4f, 0x1ec9a, 0xc90, 0x1f6e0, 0xc01, 0x1fef8, 0xb5f, 0x208b1, 0xab6, 0x21362, 0xa15, 0x21e46, 0x988, 0x2285d, 0x934,
    0x22ea8, 0x8a8, 0x239b2, 0x81d, 0x24577, 0x7c9, 0x24ce6, 0x763, 0x25663, 0x710, 0x25e8f, 0x6a0, 0x26a26, 0x672, 0x26f23, 0x5e8, 0x27ef8, 0x5ba, 0x284b5, 0x55e, 0x29057, 0x50c, 0x29bab,
    0x4c1, 0x2a674, 0x4a7, 0x2aa5e, 0x46f, 0x2b32f, 0x41f, 0x2c0ad, 0x3e7, 0x2ca8d, 0x3ba, 0x2d323, 0x10c, 0x3bfbb};
int ks265_rdoq_tables(const ks265_stream_cfg *cfg, const uint8_t *states, int slice_type, int qp, int32_t *tables)
{
    if (!cfg || !tables) return KS265_POINTER;
    uint8_t st[CX_COUNT];
    if (states) memcpy(st, states, CX_COUNT);
    else {
        Cabac tmp; uint8_t dummy[4];
        cb_init(&tmp, dummy, sizeof dummy, slice_type == KS265_SLICE_I ? 0 : slice_type == KS265_SLICE_P ? 1 : 2, qp);
        memcpy(st, tmp.state, CX_COUNT);
    }
    uint8_t c[256];
    memset(c, 0, sizeof c);
    memcpy(c + 0x0d, st + CX_CBF_LUMA, 2); memcpy(c + 0x12, st + CX_CBF_CHROMA, 4);
    memcpy(c + 0x1d, st + CX_CSBF, 4);
    memcpy(c + 0x21, st + CX_SIG, 42);
    memcpy(c + 0x4b, st + CX_LAST_X, 18); memcpy(c + 0x69, st + CX_LAST_Y, 18);
    memcpy(c + 0x87, st + CX_G1, 24); memcpy(c + 0x9f, st + CX_G2, 6);
    c[0xaa] = st[CX_ROOT_CBF];
    const int32_t *E = kEntropyBits;
    memset(tables, 0, sizeof(int32_t) * 8 * 180);
    for (int log2 = 2; log2 <= 5; ++log2)
        for (int ch = 0; ch < 2; ++ch) {
            int32_t *o = tables + ((log2 - 2) * 2 + ch) * 180;
            const int luma = !ch;
            const uint8_t *p = c + 0x0d + (luma ? 0 : 5);
            for (int i = 0; i < 5; ++i) { o[168 + 2 * i] = E[p[i]]; o[169 + 2 * i] = E[p[i] ^ 1]; }
            o[178] = E[c[0xaa]]; o[179] = E[c[0xaa] ^ 1];
            p = c + 0x1d + (luma ? 0 : 2);
            for (int i = 0; i < 2; ++i) { o[2 * i] = E[p[i]]; o[2 * i + 1] = E[p[i] ^ 1]; }
            p = c + (luma ? 0x21 : 0x3c);
            const int first = log2 > 3 ? (luma ? 21 : 12) : log2 == 3 ? 9 : 1, end = log2 > 3 ? (luma ? 27 : 15) : log2 == 3 ? (luma ? 21 : 12) : 9;
            o[4] = E[p[0]]; o[46] = E[p[0] ^ 1];
            for (int i = first; i < end; ++i) { o[4 + i] = E[p[i]]; o[46 + i] = E[p[i] ^ 1]; }
            const int off = luma ? 3 * log2 - 6 + ((log2 - 1) >> 2) : 15, shift = luma ? (log2 + 1) >> 2 : log2 - 2, n = 2 * log2 - 1;
            for (int d = 0; d < 2; ++d) {
                const uint8_t *l = c + 0x4b + 0x1e * d;
                int32_t bits = 0;
                for (int k = 0; k < n; ++k) { const uint8_t s = l[off + (k >> shift)]; o[88 + 10 * d + k] = bits + E[s]; bits += E[s ^ 1]; }
                o[88 + 10 * d + n] = bits;
            }
            const int g1 = luma ? 0x87 : 0x97, n1 = luma ? 16 : 8, g2 = luma ? 0x9f : 0xa3, n2 = luma ? 4 : 2;
            for (int i = 0; i < n1; ++i) { o[108 + 2 * i] = E[c[g1 + i]]; o[109 + 2 * i] = E[c[g1 + i] ^ 1]; }
            for (int i = 0; i < n2; ++i) { o[156 + 2 * i] = E[c[g2 + i]]; o[157 + 2 * i] = E[c[g2 + i] ^ 1]; }
        }
    return KS265_OK;
}

long ks265_write_slice(const ks265_stream_cfg *cfg, const ks265_slice_in *in, void *scratch, uint8_t *out, size_t cap)
{
    if (!scratch || !out) return KS265_POINTER;
    int r = slice_args_ok(cfg, in);
    if (r) return r;
    if (cfg->wpp) {                                                  /* the rows one after the other on this thread */
        r = ks265_wpp_begin(cfg, in, scratch);
        for (int ry = 0; !r && ry < ks265_wpp_rows(scratch); ++ry) r = ks265_wpp_code_row(scratch, ry);
        return r ? r : ks265_wpp_finish(scratch, out, cap);
    }
    Enc *e = (Enc *)scratch;
    uint8_t *rbsp = (uint8_t *)scratch + sizeof(Enc);
    const size_t skip_bytes = (size_t)(cfg->width >> 3) * (size_t)(cfg->height >> 3);
    const size_t rcap = sizeof(Enc) + (size_t)cfg->width * (size_t)cfg->height * 4 + 65536 - sizeof(Enc);
    enc_setup(e, cfg, in, rbsp + rcap);
    scans_init(&e->scans);
    memset(e->skip, 0, skip_bytes);
    BitW b; bw_init(&b, rbsp, rcap);
    const int sao_on = cfg->sao && in->sao != NULL;
    r = write_slice_header(&b, cfg, in, 0, NULL);
    if (r) return r;

    cb_init(&e->c, rbsp + b.pos, rcap - b.pos, in->slice_type == KS265_SLICE_I ? 0 : in->slice_type == KS265_SLICE_P ? 1 : 2, in->qp);
    for (int ry = 0; ry < e->ctb_rows; ++ry)
        for (int rx = 0; rx < e->ctb_cols; ++rx) {
            if (sao_on) sao_ctb(e, rx, ry);
            if (rx == 0 && ry == 0) e->qp_prev = in->qp;               /* (no entropy_coding_sync: the predictor runs on across the rows) */
            e->dqp_coded = 0; e->qp_want = in->qp_map ? in->qp_map[ry * e->ctb_cols + rx] : in->qp;
            r = coding_quadtree(e, rx << 6, ry << 6, 6);
            if (r) return r;
            if (e->dqp_coded) e->qp_prev = e->qp_want;
            cb_terminate(&e->c, ry == e->ctb_rows - 1 && rx == e->ctb_cols - 1);   /* end_of_slice_segment_flag */
        }
    cb_finish(&e->c);
    if (e->c.overflow) return KS265_NOTSUPPORTED;
    return nal_wrap(in->nal_type, rbsp, b.pos + e->c.pos, out, cap);
}
