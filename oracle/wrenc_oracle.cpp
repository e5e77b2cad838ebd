// wrenc_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see wrenc_oracle.h).
//
// Literal C++17 restatement of wrenc's per-CTU RD search + final pass for the
// live configuration: CTU 32, QT only (32/16/8/4), 8-bit 4:2:0,
// dep-quant on, DCT-2 only, CCLM on, fixed QP.  PARITY UNPINNED (no reference
// golden vectors exist; the Rust reference cannot be built in this image).
//
// Conventions: every function cites the reference lines it follows.  Integer
// types mirror the reference (i16 intermediates are re-narrowed with (int16_t)
// where the reference computes in i16; Rust release builds wrap, Cargo.toml:21).
// Build with -ffp-contract=off: the f32 cost arithmetic must not be fused.

#include "wrenc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// Tables
// ---------------------------------------------------------------------------

// DCT-2: integer cosine values c[j] ~ 64*sqrt(2)*cos(j*pi/128) as fixed by
// H.266 8.7.4.5 (same numbers as transformer.rs:934-1191, where row k of the
// 64-point matrix is c[(2n+1)k] with the usual cosine symmetries).
static const int kCos[65] = {
    64, 91, 90, 90, 90, 90, 90, 90, 89, 88, 88, 87, 87, 86, 85, 84, 83, 83, 82, 81, 80,
    79, 78, 77, 75, 73, 73, 71, 70, 69, 67, 65, 64, 62, 61, 59, 57, 56, 54, 52, 50, 48,
    46, 44, 43, 41, 38, 37, 36, 33, 31, 28, 25, 24, 22, 20, 18, 15, 13, 11, 9,  7,  4,
    2,  0};

static int16_t g_dct64[64][64];
// set when a table index reaches 1024: the reference panics there (block_splitter.rs:453,
// quantizer.rs:30); the oracle clamps the index and reports failure
static bool g_table_overflow = false;
static inline size_t tbl(size_t i) {
    if (i >= 1024) {
        g_table_overflow = true;
        return 1023;
    }
    return i;
}
static bool g_tables_ready = false;
// Test hook (wro_debug_perturb): a deliberate misreading of one constant, to prove that the independent
// spec decoder notices what the oracle + GPU pair cannot notice about themselves.  0 = none.
static int g_perturb = 0;

// intraPredAngle for predModeIntra -14..80 (H.266 Table 24; common.rs:145)
static const int kIntraAngle[95] = {
    512, 341, 256, 171, 128, 102, 86,  73,  64,  57,  51,  45,  39,  35,  0,   0,
    32,  29,  26,  23,  20,  18,  16,  14,  12,  10,  8,   6,   4,   3,   2,   1,
    0,   -1,  -2,  -3,  -4,  -6,  -8,  -10, -12, -14, -16, -18, -20, -23, -26, -29,
    -32, -29, -26, -23, -20, -18, -16, -14, -12, -10, -8,  -6,  -4,  -3,  -2,  -1,
    0,   1,   2,   3,   4,   6,   8,   10,  12,  14,  16,  18,  20,  23,  26,  29,
    32,  35,  39,  45,  51,  57,  64,  73,  86,  102, 128, 171, 256, 341, 512};

// fC interpolation filter (H.266 Table 25; common.rs:153)
static const int kFC[32][4] = {
    {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},
    {-2, 58, 10, -2}, {-3, 57, 12, -2}, {-4, 56, 14, -2}, {-4, 55, 15, -2},
    {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3},
    {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4},
    {-4, 36, 36, -4}, {-4, 33, 39, -4}, {-4, 30, 42, -4}, {-4, 29, 44, -5},
    {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3},
    {-2, 10, 58, -2}, {-1, 7, 60, -2},  {0, 4, 62, -2},   {0, 2, 63, -1}};

static int g_fg[32][4];         // fG (common.rs:188): {16-(p>>1), 32-(p>>1), 16+(p>>1), p>>1}
static int16_t g_pdpc_w[3][64]; // PDPSF_WEIGHTS intra_predictor.rs:36-52: 32 >> ((i<<1)>>nScale)

struct XY {
    int x, y;
};
static std::vector<XY> g_diag[5][5]; // [log2h][log2w], ctu.rs:14-81

static inline int ilog2(int v) {
    int r = 0;
    while (v > 1) {
        v >>= 1;
        ++r;
    }
    return r;
}

static void init_tables() {
    if (g_tables_ready) return;
    for (int k = 0; k < 64; ++k)
        for (int n = 0; n < 64; ++n) {
            int t = ((2 * n + 1) * k) % 256; // angle in units of pi/128
            if (t > 128) t = 256 - t;
            int v = (t > 64) ? -kCos[128 - t] : kCos[t];
            g_dct64[k][n] = (int16_t)v;
        }
    for (int p = 0; p < 32; ++p) {
        g_fg[p][0] = 16 - (p >> 1);
        g_fg[p][1] = 32 - (p >> 1);
        g_fg[p][2] = 16 + (p >> 1);
        g_fg[p][3] = p >> 1;
    }
    for (int s = 0; s < 3; ++s)
        for (int i = 0; i < 64; ++i) {
            int sh = (i << 1) >> s;
            g_pdpc_w[s][i] = (int16_t)(sh > 5 ? 0 : (32 >> sh));
        }
    // ctu.rs:54-77 up-right diagonal scan
    for (int lh = 0; lh <= 4; ++lh)
        for (int lw = 0; lw <= 4; ++lw) {
            int bw = 1 << lw, bh = 1 << lh;
            std::vector<XY>& o = g_diag[lh][lw];
            o.assign((size_t)bw * bh, XY{0, 0});
            int i = 0, x = 0, y = 0;
            bool stop = false;
            while (!stop) {
                while (y >= 0) {
                    if (x < bw && y < bh) {
                        o[i].x = x;
                        o[i].y = y;
                        ++i;
                    }
                    --y;
                    ++x;
                }
                y = x;
                x = 0;
                if (i >= bw * bh) stop = true;
            }
        }
    g_tables_ready = true;
}

static const int kQStateTrans[4][2] = {{0, 2}, {2, 0}, {1, 3}, {3, 1}}; // encoder_context.rs:339
static const int kLevelScale0[6] = {40, 45, 51, 57, 64, 72};           // quantizer.rs:8

// ---------------------------------------------------------------------------
// RD-model constants (defaults; block_splitter.rs / quantizer.rs)
// ---------------------------------------------------------------------------
struct RdConst {
    int64_t lv[1024]; // lv_dq_trellis_table, block_splitter.rs:51-52
    int64_t dq[1024]; // dq_table, quantizer.rs:20-21
    int64_t lambda_q; // quantizer.rs:683
    int qp;
    // f32 model terms (trellis + dep-quant variants)
    float non_planar_offset = 2.2153597f;   // :195
    float mpm_idx_offset = 1.3660221f;      // :211
    float mpm_remainder_mult = 0.5007182f;  // :227
    float mpm_remainder_offset = 2.2973304f; // :243
    float planar_offset = 0.9626864f;       // :260 (key "planer_offset_dq_trellis")
    float header_bits = 1.1772872f;         // :276
    float chroma_header_bits = 1.309252f;   // :634
    float qp_div = 4.4043665f;              // :292
    float lambda_mul = 1.1282581f;          // :308
    float cclm_pow = 0.4587651f;            // :318
    float mpm_idx_pow = 0.40271285f;        // :322
    float mpm_remainder_pow = 0.34385094f;  // :326
    float cclm_mode_idx_offset = 2.1f;      // :336
    float non_cclm_offset = 0.89f;          // :352
    float cclm_offset = 0.53f;              // :368
    float chroma_lambda_mul = 1.1282581f;   // :775-778: extra-param "a" replaces lambda_mul in the chroma cost
};

// --extra-params of the reference (main.rs:202-217): key -> text; read where the reference reads them
static std::map<std::string, std::string> g_extra;
static int g_extra_gen = 0; // bumped by wro_set_extra_params: cached constants are rebuilt
static double xp_f64(const char* key, double dflt) {
    auto it = g_extra.find(key);
    return it == g_extra.end() ? dflt : strtod(it->second.c_str(), nullptr);
}
static float xp_f32(const char* key, float dflt) {
    auto it = g_extra.find(key);
    return it == g_extra.end() ? dflt : strtof(it->second.c_str(), nullptr);
}

static void init_rd(RdConst& r, int qp) {
    r = RdConst();
    r.qp = qp;
    // block_splitter.rs:29-52, quantizer.rs:16-21
    const double lv_pow = xp_f64("lv_pow_dq_trellis", 0.48592678233563835);
    const double lv_off = xp_f64("lv_offset_dq_trellis", 0.15150746310196822);
    const double dq_pow = xp_f64("quant_lv_pow", 0.5004010166085378);
    for (int i = 0; i < 1024; ++i) {
        r.lv[i] = (int64_t)(std::pow((double)i + lv_off, lv_pow) * 16384.0);
        r.dq[i] = (int64_t)std::pow((double)(i * 16384), dq_pow);
    }
    // quantizer.rs:650-683  (2.0f64.powf(qp/qp_div) * lambda_mul) as i64 + lambda_offset
    const auto off = g_extra.find("quant_lambda_offset_trellis");
    r.lambda_q = (int64_t)(std::pow(2.0, (double)qp / xp_f64("quant_qp_div_trellis", 5.218413785332902)) *
                           xp_f64("quant_lambda_mul_trellis", 1.2709404305806742)) +
                 (off == g_extra.end() ? 11 : strtoll(off->second.c_str(), nullptr, 10));
    // block_splitter.rs:187-375,594-693: the dep-quant + trellis variants are the live ones
    r.non_planar_offset = xp_f32("non_planar_offset_dq_trellis", r.non_planar_offset);
    r.mpm_idx_offset = xp_f32("mpm_idx_offset_dq_trellis", r.mpm_idx_offset);
    r.mpm_remainder_mult = xp_f32("mpm_remainder_mult_dq_trellis", r.mpm_remainder_mult);
    r.mpm_remainder_offset = xp_f32("mpm_remainder_offset_dq_trellis", r.mpm_remainder_offset);
    r.planar_offset = xp_f32("planer_offset_dq_trellis", r.planar_offset);
    r.header_bits = xp_f32("header_bits_dq_trellis", r.header_bits);
    r.chroma_header_bits = xp_f32("chroma_header_bits_dq_trellis", r.chroma_header_bits);
    r.qp_div = xp_f32("qp_div_dq_trellis", r.qp_div);
    r.lambda_mul = xp_f32("lambda_mul_dq_trellis", r.lambda_mul);
    r.cclm_pow = xp_f32("cclm_pow", r.cclm_pow);
    r.mpm_idx_pow = xp_f32("mpm_idx_pow", r.mpm_idx_pow);
    r.mpm_remainder_pow = xp_f32("mpm_remainder_pow", r.mpm_remainder_pow);
    r.cclm_mode_idx_offset = xp_f32("cclm_mode_idx_offset_dq_trellis", r.cclm_mode_idx_offset);
    r.non_cclm_offset = xp_f32("non_cclm_offset_dq_trellis", r.non_cclm_offset);
    r.cclm_offset = xp_f32("cclm_offset_dq_trellis", r.cclm_offset);
    r.chroma_lambda_mul = xp_f32("a", r.lambda_mul);
}

// block_splitter.rs:472 / :775-778
static inline float rd_lambda(const RdConst& r) {
    return std::pow(2.0f, (float)r.qp / r.qp_div) * r.lambda_mul;
}
static inline float rd_lambda_chroma(const RdConst& r) {
    return std::pow(2.0f, (float)r.qp / r.qp_div) * r.chroma_lambda_mul;
}

// block_splitter.rs:377-406
static int64_t header_bits_luma(const RdConst& r, int tree, bool non_planar, bool mpm_flag,
                                int mpm_idx, int mpm_rem, bool cclm_flag, int cclm_idx) {
    float cclm_bits;
    if (cclm_flag)
        cclm_bits = r.cclm_offset + std::pow((float)cclm_idx + r.cclm_mode_idx_offset, r.cclm_pow);
    else if (tree == 1)
        cclm_bits = 0.0f;
    else
        cclm_bits = r.non_cclm_offset;
    float mode_bits;
    if (non_planar) {
        float t;
        if (mpm_flag)
            t = std::pow((float)mpm_idx + r.mpm_idx_offset, r.mpm_idx_pow);
        else
            t = r.mpm_remainder_mult *
                std::pow((float)mpm_rem + r.mpm_remainder_offset, r.mpm_remainder_pow);
        mode_bits = r.non_planar_offset + t;
    } else {
        mode_bits = r.planar_offset;
    }
    mode_bits = mode_bits + cclm_bits;
    float hb;
    if (tree == 0)
        hb = r.header_bits + mode_bits;
    else if (tree == 1)
        hb = r.header_bits / 3.0f + mode_bits;
    else
        hb = cclm_bits;
    return (int64_t)(hb * 16384.0f);
}

// block_splitter.rs:695-712
static int64_t header_bits_chroma(const RdConst& r, bool cclm_flag, int cclm_idx) {
    float mode_bits;
    if (cclm_flag)
        mode_bits = r.cclm_offset + std::pow((float)cclm_idx + r.cclm_mode_idx_offset, r.cclm_pow);
    else
        mode_bits = r.non_cclm_offset;
    return (int64_t)((r.chroma_header_bits + mode_bits) * 16384.0f);
}

// ---------------------------------------------------------------------------
// Transform (transformer.rs)
// ---------------------------------------------------------------------------

// transformer.rs:2040-2378, DCT-2 both directions (mts_idx == 0, :1894-1902)
static void fwd_dct(const int16_t* res, int log2n, int16_t* coef) {
    const int n = 1 << log2n;
    const int step = 64 >> log2n; // S[6-log2n][i][x] = B[i<<shift][x], :1212-1221
    std::vector<int32_t> h((size_t)n * n), t((size_t)n * n);
    // stage 1 (:2180-2188): h[y][i] = sum_x T[i][x]*r[y][x]
    for (int y = 0; y < n; ++y)
        for (int i = 0; i < n; ++i) {
            int32_t s = 0;
            for (int x = 0; x < n; ++x)
                s += (int32_t)g_dct64[i * step][x] * (int32_t)res[y * n + x];
            h[y * n + i] = s;
        }
    // :2201-2209
    {
        const int shift = log2n - 1;
        const int32_t d = 1 << (shift - 1);
        for (int k = 0; k < n * n; ++k) h[k] = (h[k] + d) >> shift;
    }
    // stage 2 (:2287-2295): t[i][x] = sum_y T[i][y]*h[y][x]
    for (int x = 0; x < n; ++x)
        for (int i = 0; i < n; ++i) {
            int32_t s = 0;
            for (int y = 0; y < n; ++y) s += (int32_t)g_dct64[i * step][y] * h[y * n + x];
            t[i * n + x] = s;
        }
    // :2309-2316
    {
        const int shift = log2n + 6;
        const int32_t d = 1 << (shift - 1);
        for (int k = 0; k < n * n; ++k) coef[k] = (int16_t)((t[k] + d) >> shift); // :2371 `as i16`
    }
}

// transformer.rs:2380-2737
static void inv_dct(const int16_t* deq, int log2n, int16_t* res) {
    const int n = 1 << log2n;
    const int step = 64 >> log2n; // I[shift][x][i] = B[i<<shift][x], :1222-1231
    std::vector<int32_t> v((size_t)n * n), it((size_t)n * n);
    // stage 1 vertical (:2545-2554): v[y][x] = sum_i T[i][y]*d[i][x]
    for (int x = 0; x < n; ++x)
        for (int y = 0; y < n; ++y) {
            int32_t s = 0;
            for (int i = 0; i < n; ++i)
                s += (int32_t)g_dct64[i * step][y] * (int32_t)deq[i * n + x];
            v[y * n + x] = s;
        }
    // :2569-2580
    for (int k = 0; k < n * n; ++k) {
        int32_t c = (v[k] + (g_perturb == 4 ? 63 : 64)) >> 7;
        v[k] = std::min(std::max(c, -32768), 32767);
    }
    // stage 2 horizontal (:2663-2671): it[y][x] = sum_i T[i][x]*v[y][i]
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            int32_t s = 0;
            for (int i = 0; i < n; ++i) s += (int32_t)g_dct64[i * step][x] * v[y * n + i];
            it[y * n + x] = s;
        }
    // :2687-2700  bd_shift = 20 - bit_depth = 12
    for (int k = 0; k < n * n; ++k) res[k] = (int16_t)((it[k] + 2048) >> 12);
}

// ---------------------------------------------------------------------------
// Quantiser (quantizer.rs)
// ---------------------------------------------------------------------------

// quantizer.rs:617-622 + derive_ls :326-333, flat m == 16 (:571-583)
static inline int32_t level_scale(int qp) {
    return (int32_t)((16 * (kLevelScale0[(qp + 1) % 6] + (g_perturb == 3 ? 1 : 0))) << ((qp + 1) / 6));
}
// quantizer.rs:558-569: bit_depth + rect(0) + (log2w+log2h)/2 - 5 + dep_quant(1)
static inline int quant_bd_shift(int log2n) { return 8 + log2n - 5 + 1; }

// ctu.rs:827-845 for square TBs >= 4: 4x4 sub-blocks
struct ScanGeom {
    int log2n, n, log2_sb = 2, num_sb_coeff = 16, num_sb;
    const std::vector<XY>* coeff_order;
    const std::vector<XY>* sb_order;
    explicit ScanGeom(int l2) : log2n(l2), n(1 << l2) {
        num_sb = 1 << (2 * l2 - 4);
        coeff_order = &g_diag[2][2];
        sb_order = &g_diag[l2 - 2][l2 - 2];
    }
    inline void pos(int sb, int sp, int& xc, int& yc) const {
        const XY s = (*sb_order)[sb];
        const XY c = (*coeff_order)[sp];
        xc = (s.x << 2) + c.x;
        yc = (s.y << 2) + c.y;
    }
};

struct TrellisEntry {
    size_t a;
    int16_t q;
    int64_t cost;
};

struct Trellis {
    const RdConst& rd;
    const int16_t* t;
    ScanGeom g;
    int32_t lsc;
    int bd_shift;
    int32_t bd_offset;
    int64_t lambda;
    std::vector<TrellisEntry> table; // [sb][pos][state]
    Trellis(const RdConst& r, const int16_t* coef, int log2n, int qp)
        : rd(r), t(coef), g(log2n) {
        lsc = level_scale(qp);
        bd_shift = quant_bd_shift(log2n);
        bd_offset = (1 << bd_shift) >> 1;
        lambda = r.lambda_q;
        table.assign((size_t)g.num_sb * 16 * 4, TrellisEntry{0, 0, INT64_MIN});
    }
    inline int64_t dq_cost(int64_t dist, int64_t bits) const { // quantizer.rs:29-31
        return 128 * dist + lambda * rd.dq[tbl((size_t)bits)];
    }
    // quantizer.rs:338-517 (memo key omits is_trailing_zeros: first visit wins)
    TrellisEntry search(int q_state, int last_scan_pos, int last_sub_block, size_t depth,
                        bool is_trailing_zeros) {
        TrellisEntry& slot = table[((size_t)last_sub_block * 16 + last_scan_pos) * 4 + q_state];
        if (slot.cost != INT64_MIN) return slot;
        int xc, yc;
        g.pos(last_sub_block, last_scan_pos, xc, yc);
        const int32_t tc = (int32_t)t[yc * g.n + xc];
        size_t a;
        int16_t q;
        int64_t cost;
        if (depth == 0 || (last_scan_pos == 0 && last_sub_block == 0)) {
            if (tc == 0) {
                cost = dq_cost(0, 1 - (int64_t)is_trailing_zeros);
                a = 0;
                q = 0;
            } else {
                const size_t delta = (q_state > 1) ? 1 : 0;
                int32_t s = (int32_t)((uint32_t)tc << bd_shift) - bd_offset;
                if (tc < 0) s = -s;
                const size_t a0 = (size_t)(s / lsc / 2);
                int16_t q0 = (int16_t)(2 * a0 - delta); // usize wrap, then `as i16` (:379)
                if (tc < 0) q0 = (int16_t)(-q0);
                const int32_t dq0 = ((int32_t)q0 * lsc + bd_offset) >> bd_shift;
                const int32_t d0 = std::abs(tc - dq0);
                const int64_t cost0 =
                    dq_cost(d0, (int64_t)(a0 + 1) * (int64_t)(a0 != 0 || !is_trailing_zeros));
                const size_t a1 = a0 + 1;
                int16_t q1 = (int16_t)(2 * a1 - delta);
                if (tc < 0) q1 = (int16_t)(-q1);
                const int32_t dq1 = ((int32_t)q1 * lsc + bd_offset) >> bd_shift;
                const int32_t d1 = std::abs(tc - dq1);
                const int64_t cost1 = dq_cost(d1, (int64_t)(a1 + 1));
                if (cost0 <= cost1) {
                    a = a0;
                    q = q0;
                    cost = cost0;
                } else {
                    a = a1;
                    q = q1;
                    cost = cost1;
                }
            }
        } else {
            const int* trans = kQStateTrans[q_state];
            int next_scan_pos, next_sub_block;
            if (last_scan_pos == 0) {
                next_scan_pos = g.num_sb_coeff - 1;
                next_sub_block = last_sub_block - 1;
            } else {
                next_scan_pos = last_scan_pos - 1;
                next_sub_block = last_sub_block;
            }
            if (tc == 0) {
                const int nq = trans[0];
                const TrellisEntry n =
                    search(nq, next_scan_pos, next_sub_block, depth - 1, is_trailing_zeros);
                cost = n.cost + dq_cost(0, 1 - (int64_t)is_trailing_zeros);
                a = 0;
                q = 0;
            } else {
                int32_t s = (int32_t)((uint32_t)tc << bd_shift) - bd_offset;
                if (tc < 0) s = -s;
                const int32_t delta = (q_state > 1) ? 1 : 0;
                const size_t a0 = (size_t)((s / lsc + delta) / 2);
                const int nq0 = trans[a0 & 1];
                int32_t q0 = a0 > 0 ? 2 * (int32_t)a0 - delta : 0;
                if (tc < 0) q0 = -q0;
                const int32_t dq0 = (q0 * lsc + bd_offset) >> bd_shift;
                const int32_t d0 = std::abs(tc - dq0);
                int64_t cost0 = (a0 == 0 && is_trailing_zeros) ? dq_cost(d0, 0)
                                                                 : dq_cost(d0, (int64_t)(a0 + 1));
                const TrellisEntry n0 = search(nq0, next_scan_pos, next_sub_block, depth - 1,
                                               is_trailing_zeros && a0 == 0);
                cost0 += n0.cost;
                const size_t a1 = a0 + 1;
                const int nq1 = trans[a1 & 1];
                int64_t q1 = 2 * (int64_t)a1 - (int64_t)(q_state > 1);
                if (tc < 0) q1 = -q1;
                const int32_t dq1 = ((int32_t)q1 * lsc + bd_offset) >> bd_shift;
                const int32_t d1 = std::abs(tc - dq1);
                int64_t cost1 = dq_cost(d1, (int64_t)(a1 + 1));
                const TrellisEntry n1 =
                    search(nq1, next_scan_pos, next_sub_block, depth - 1, false);
                cost1 += n1.cost;
                if (cost0 <= cost1) {
                    a = a0;
                    q = (int16_t)q0;
                    cost = cost0;
                } else {
                    a = a1;
                    q = (int16_t)q1;
                    cost = cost1;
                }
            }
        }
        if (last_scan_pos == 0 && is_trailing_zeros && a == 0) cost -= lambda * rd.dq[1]; // :512-514
        // (re-fetch: recursion may not reallocate, table is pre-sized)
        TrellisEntry& out = table[((size_t)last_sub_block * 16 + last_scan_pos) * 4 + q_state];
        out = TrellisEntry{a, q, cost};
        return out;
    }
};

// quantizer.rs:633-721 (dep-quant, trellis=true)
static void quantize(const RdConst& rd, const int16_t* coef, int log2n, int qp, int16_t* levels) {
    Trellis tr(rd, coef, log2n, qp);
    const ScanGeom& g = tr.g;
    int q_state = 0;
    int last_scan_pos = g.num_sb_coeff;
    int last_sub_block = g.num_sb - 1;
    bool is_not_first_sub_block = last_sub_block > 0;
    bool is_trailing_zeros = true;
    const size_t depth = (size_t)g.n * g.n;
    do {
        if (last_scan_pos == 0) {
            last_scan_pos = g.num_sb_coeff;
            last_sub_block -= 1;
            is_not_first_sub_block = last_sub_block > 0;
        }
        last_scan_pos -= 1;
        int xc, yc;
        g.pos(last_sub_block, last_scan_pos, xc, yc);
        const TrellisEntry e =
            tr.search(q_state, last_scan_pos, last_sub_block, depth, is_trailing_zeros);
        is_trailing_zeros = is_trailing_zeros && (e.a == 0);
        levels[yc * g.n + xc] = e.q;
        q_state = kQStateTrans[q_state][e.a & 1];
    } while (last_scan_pos > 0 || is_not_first_sub_block);
}

// Backward 4-state Viterbi equivalent of the memoised DFS above.
// First visit of node (i, s) in the DFS has trailing == (s == 0 && i <= istar)
// where positions i count in reverse scan order (0 = last scan position) and
// istar is the first i whose state-0 a0 is > 0 (all positions if none).
static void quantize_viterbi(const RdConst& rd, const int16_t* coef, int log2n, int qp,
                             int16_t* levels) {
    ScanGeom g(log2n);
    const int N = g.n * g.n;
    const int32_t lsc = level_scale(qp);
    const int bd_shift = quant_bd_shift(log2n);
    const int32_t bd_offset = (1 << bd_shift) >> 1;
    const int64_t lambda = rd.lambda_q;
    std::vector<int32_t> tc(N);
    std::vector<uint8_t> first_in_sb(N);
    std::vector<int> px(N), py(N);
    {
        int i = 0;
        for (int sb = g.num_sb - 1; sb >= 0; --sb)
            for (int sp = 15; sp >= 0; --sp, ++i) {
                int xc, yc;
                g.pos(sb, sp, xc, yc);
                px[i] = xc;
                py[i] = yc;
                tc[i] = coef[yc * g.n + xc];
                first_in_sb[i] = (sp == 0);
            }
    }
    auto sval = [&](int32_t t) {
        int32_t s = (int32_t)((uint32_t)t << bd_shift) - bd_offset;
        return t < 0 ? -s : s;
    };
    int istar = N; // exclusive bound: state-0 nodes with i <= istar are "trailing"
    for (int i = 0; i < N; ++i) {
        if (tc[i] == 0) continue;
        const int32_t a0 = (sval(tc[i]) / lsc) / 2; // delta == 0 in state 0; DC formula is the same
        if (a0 > 0) {
            istar = i;
            break;
        }
    }
    auto trailing = [&](int i, int s) { return s == 0 && i <= istar; };
    auto dqc = [&](int64_t dist, int64_t bits) { return 128 * dist + lambda * rd.dq[tbl((size_t)bits)]; };
    std::vector<int64_t> C((size_t)(N + 1) * 4, 0);
    std::vector<uint32_t> A((size_t)N * 4);
    std::vector<int16_t> Q((size_t)N * 4);
    for (int i = N - 1; i >= 0; --i)
        for (int s = 0; s < 4; ++s) {
            const bool tz = trailing(i, s);
            const int32_t t = tc[i];
            size_t a;
            int16_t q;
            int64_t cost;
            if (i == N - 1) {
                if (t == 0) {
                    cost = dqc(0, 1 - (int64_t)tz);
                    a = 0;
                    q = 0;
                } else {
                    const size_t delta = s > 1;
                    const size_t a0 = (size_t)(sval(t) / lsc / 2);
                    int16_t q0 = (int16_t)(2 * a0 - delta);
                    if (t < 0) q0 = (int16_t)-q0;
                    const int32_t d0 = std::abs(t - (((int32_t)q0 * lsc + bd_offset) >> bd_shift));
                    const int64_t c0 = dqc(d0, (int64_t)(a0 + 1) * (int64_t)(a0 != 0 || !tz));
                    const size_t a1 = a0 + 1;
                    int16_t q1 = (int16_t)(2 * a1 - delta);
                    if (t < 0) q1 = (int16_t)-q1;
                    const int32_t d1 = std::abs(t - (((int32_t)q1 * lsc + bd_offset) >> bd_shift));
                    const int64_t c1 = dqc(d1, (int64_t)(a1 + 1));
                    if (c0 <= c1) {
                        a = a0;
                        q = q0;
                        cost = c0;
                    } else {
                        a = a1;
                        q = q1;
                        cost = c1;
                    }
                }
            } else {
                const int* trans = kQStateTrans[s];
                if (t == 0) {
                    cost = C[(size_t)(i + 1) * 4 + trans[0]] + dqc(0, 1 - (int64_t)tz);
                    a = 0;
                    q = 0;
                } else {
                    const int32_t delta = s > 1;
                    const size_t a0 = (size_t)((sval(t) / lsc + delta) / 2);
                    int32_t q0 = a0 > 0 ? 2 * (int32_t)a0 - delta : 0;
                    if (t < 0) q0 = -q0;
                    const int32_t d0 = std::abs(t - ((q0 * lsc + bd_offset) >> bd_shift));
                    int64_t c0 = (a0 == 0 && tz) ? dqc(d0, 0) : dqc(d0, (int64_t)(a0 + 1));
                    c0 += C[(size_t)(i + 1) * 4 + trans[a0 & 1]];
                    const size_t a1 = a0 + 1;
                    int32_t q1 = 2 * (int32_t)a1 - delta;
                    if (t < 0) q1 = -q1;
                    const int32_t d1 = std::abs(t - ((q1 * lsc + bd_offset) >> bd_shift));
                    int64_t c1 = dqc(d1, (int64_t)(a1 + 1));
                    c1 += C[(size_t)(i + 1) * 4 + trans[a1 & 1]];
                    if (c0 <= c1) {
                        a = a0;
                        q = (int16_t)q0;
                        cost = c0;
                    } else {
                        a = a1;
                        q = (int16_t)q1;
                        cost = c1;
                    }
                }
            }
            if (first_in_sb[i] && tz && a == 0) cost -= lambda * rd.dq[1];
            C[(size_t)i * 4 + s] = cost;
            A[(size_t)i * 4 + s] = (uint32_t)a;
            Q[(size_t)i * 4 + s] = q;
        }
    int s = 0;
    for (int i = 0; i < N; ++i) {
        levels[py[i] * g.n + px[i]] = Q[(size_t)i * 4 + s];
        s = kQStateTrans[s][A[(size_t)i * 4 + s] & 1];
    }
}

// ---------------------------------------------------------------------------
// MODEL of the device quantiser's shortcuts (round 4), kept here so that they are proven against the literal DFS on the
// CPU before the kernel relies on them (tests/test_oracle.py).  Same backward Viterbi as above, plus two exits that
// skip the serial walk where its outcome can be PROVEN without it:
//
//  (H) the head.  Positions i < 16 * SB (SB = the 4x4 sub-block that holds istar, or the DC sub-block) all have
//      a0(state 0) == 0.  The forward trace starts in state 0 at i = 0 and stays there while state 0 decides "zero", so
//      only state 0's decisions matter in the head.  With V_i[s] the exact cost-to-go, G_i = min_{s != 0} V_i[s] - V_i[0],
//      alpha_i = c1(i) - c0tz(i) (state 0's margin; +inf for a zero coefficient) and beta_i = min over the branches of
//      states 1..3 that do not lead to state 0 of (cost - c0tz(i)):
//          state 0 decides zero at i      <=>  V_{i+1}[2] - V_{i+1}[0] >= -alpha_i   <=  G_{i+1} >= -alpha_i
//          G_i >= min(alpha_i, beta_i + G_{i+1}) + rebate_i        (state 1's odd branch is the only way into state 0,
//                                                                   and it costs c1(i), the same as state 0's)
//      so with beta_i >= 0 everywhere, alpha_min = min alpha_i >= 0 and G_{16 SB} >= -alpha_min (exact, from the walk so
//      far) induction gives G_i >= min(alpha_min, G_{16 SB}) >= -alpha_min for the whole head: all its levels are zero.
//  (Z) a sub-block strictly behind the head (every node non-trailing), not the DC one, whose 16 quotients are all zero:
//      c0 is the same in all four states, c1 depends on delta only.  If every decision is "zero" the state permutation
//      [0,2,1,3] is applied 16 times (identity) and V_top[s] = V_bottom[s] + sum c0.  The decisions at step j compare
//      m_j(delta) = c1 - c0 with V[t0(s)] - V[t1(s)], which under the hypothesis alternate between two patterns of the
//      bottom values: even j: |V0 - V2| <= m_j(0), |V1 - V3| <= m_j(1); odd j: |V0 - V1| <= m_j(0), |V2 - V3| <= m_j(1)
//      (tie rule cost0 <= cost1 -> zero, hence <=).  All 16 x 4 checks hold <=> the hypothesis is exact.
// ---------------------------------------------------------------------------
struct DqScStats {
    long long blocks, nz_blocks, sub_blocks, head_sb_skipped, head_tests, head_fail, z_eligible, z_pass, walked, seg_sb, seg_kept, whole_zero;
};
static DqScStats g_sc_stats[6]; // by log2n
static bool g_sc_stats_on = false;
static long long g_sc_mismatch = 0;

static void quantize_viterbi_sc(const RdConst& rd, const int16_t* coef, int log2n, int qp, int16_t* levels,
                                bool use_head, bool use_z, bool use_seg = false, bool use_whole = true) {
    ScanGeom g(log2n);
    const int N = g.n * g.n;
    const int32_t lsc = level_scale(qp);
    const int bd_shift = quant_bd_shift(log2n);
    const int32_t bd_offset = (1 << bd_shift) >> 1;
    const int64_t lambda = rd.lambda_q;
    std::vector<int32_t> tc(N), qd(N);
    std::vector<int> px(N), py(N);
    bool any = false;
    {
        int i = 0;
        for (int sb = g.num_sb - 1; sb >= 0; --sb)
            for (int sp = 15; sp >= 0; --sp, ++i) {
                int xc, yc;
                g.pos(sb, sp, xc, yc);
                px[i] = xc;
                py[i] = yc;
                tc[i] = coef[yc * g.n + xc];
                any = any || tc[i] != 0;
            }
    }
    auto sval = [&](int32_t t) {
        int32_t s = (int32_t)((uint32_t)t << bd_shift) - bd_offset;
        return t < 0 ? -s : s;
    };
    int istar = N;
    for (int i = 0; i < N; ++i) {
        qd[i] = tc[i] == 0 ? 0 : sval(tc[i]) / lsc;
        if (istar == N && tc[i] != 0 && qd[i] / 2 > 0) istar = i;
    }
    DqScStats& st = g_sc_stats[log2n];
    if (g_sc_stats_on) {
        st.blocks++;
        if (any) {
            st.nz_blocks++;
            st.sub_blocks += N / 16;
        }
    }
    auto dqc = [&](int64_t dist, int64_t bits) { return 128 * dist + lambda * rd.dq[tbl((size_t)bits)]; };
    // branch costs of node (i, s): level a0 -> (c0, parity), a0 + 1 -> c1; has1 = false for a zero coefficient
    struct Br {
        int64_t c0, c1;
        size_t a0;
        int16_t q0, q1;
        bool has1;
    };
    auto branches = [&](int i, int s) {
        Br b;
        const bool tz = (s == 0 && i <= istar);
        const int32_t t = tc[i];
        if (t == 0) {
            b.c0 = dqc(0, 1 - (int64_t)tz);
            b.c1 = 0;
            b.a0 = 0;
            b.q0 = b.q1 = 0;
            b.has1 = false;
            return b;
        }
        b.has1 = true;
        if (i == N - 1) {
            const size_t delta = s > 1;
            const size_t a0 = (size_t)(sval(t) / lsc / 2);
            int16_t q0 = (int16_t)(2 * a0 - delta);
            if (t < 0) q0 = (int16_t)-q0;
            const int32_t d0 = std::abs(t - (((int32_t)q0 * lsc + bd_offset) >> bd_shift));
            b.c0 = dqc(d0, (int64_t)(a0 + 1) * (int64_t)(a0 != 0 || !tz));
            int16_t q1 = (int16_t)(2 * (a0 + 1) - delta);
            if (t < 0) q1 = (int16_t)-q1;
            const int32_t d1 = std::abs(t - (((int32_t)q1 * lsc + bd_offset) >> bd_shift));
            b.c1 = dqc(d1, (int64_t)(a0 + 2));
            b.a0 = a0;
            b.q0 = q0;
            b.q1 = q1;
        } else {
            const int32_t delta = s > 1;
            const size_t a0 = (size_t)((sval(t) / lsc + delta) / 2);
            int32_t q0 = a0 > 0 ? 2 * (int32_t)a0 - delta : 0;
            if (t < 0) q0 = -q0;
            const int32_t d0 = std::abs(t - ((q0 * lsc + bd_offset) >> bd_shift));
            b.c0 = (a0 == 0 && tz) ? dqc(d0, 0) : dqc(d0, (int64_t)(a0 + 1));
            int32_t q1 = 2 * (int32_t)(a0 + 1) - delta;
            if (t < 0) q1 = -q1;
            const int32_t d1 = std::abs(t - ((q1 * lsc + bd_offset) >> bd_shift));
            b.c1 = dqc(d1, (int64_t)(a0 + 2));
            b.a0 = a0;
            b.q0 = (int16_t)q0;
            b.q1 = (int16_t)q1;
        }
        return b;
    };
    std::vector<int64_t> C((size_t)(N + 1) * 4, 0);
    std::vector<uint32_t> A((size_t)N * 4, 0);
    std::vector<int16_t> Q((size_t)N * 4, 0);
    if (!any) { // (the device leaves at once too)
        for (int i = 0; i < N; ++i) levels[py[i] * g.n + px[i]] = 0;
        return;
    }
    // alpha_i / beta_i of every position (see (H) above); the proof region is [0, 16 sb_star): the sub-blocks above the
    // first position that cannot be part of it (a0(state 0) > 0, alpha < 0 or beta < 0)
    std::vector<int64_t> alpha(N, INT64_MAX), beta(N, INT64_MAX);
    int kstar = N;
    for (int i = 0; i < N && i <= istar && kstar == N; ++i) {
        const Br b0 = branches(i, 0);
        const int64_t c0tz = b0.c0;
        int64_t am = INT64_MAX, bm = INT64_MAX;
        if (b0.has1) am = std::min(am, b0.c1 - c0tz);
        for (int s = 1; s < 4; ++s) {
            const Br b = branches(i, s);
            const int* trans = kQStateTrans[s];
            if (!b.has1) {
                bm = std::min(bm, b.c0 - c0tz); // (its target trans[s][0] is never 0 for s != 0)
            } else {
                if (trans[b.a0 & 1] != 0)
                    bm = std::min(bm, b.c0 - c0tz);
                else
                    am = std::min(am, b.c0 - c0tz); // a way into state 0
                if (trans[(b.a0 + 1) & 1] != 0)
                    bm = std::min(bm, b.c1 - c0tz);
                else
                    am = std::min(am, b.c1 - c0tz); // a way into state 0 (state 1, odd level)
            }
        }
        alpha[i] = am;
        beta[i] = bm;
        // (the DC position, with its own level formula and no successor, can be part of the region like any other: then
        // the region is the whole block)
        if (i == istar || (i == N - 1 && !use_whole) || am < 0 || bm < 0) kstar = i;
    }
    if (use_head && use_whole && kstar == N && lambda * rd.dq[1] >= 0) {
        // (W) nothing ends the region: with G = 0 behind the DC position the induction of (H) runs over the WHOLE block
        // -- every level is zero, nothing is walked
        if (g_sc_stats_on) st.whole_zero++;
        for (int i = 0; i < N; ++i) levels[py[i] * g.n + px[i]] = 0;
        return;
    }
    const int sb_star = std::min(kstar, N - 1) >> 4;
    // one position of a walk whose costs live in Cw (4 values: the costs at i + 1 on entry, at i on return)
    auto step_on = [&](int i, int64_t* Cw, uint32_t* Aw, int16_t* Qw) {
        int64_t nc[4];
        for (int s = 0; s < 4; ++s) {
            const bool tz = (s == 0 && i <= istar);
            const Br b = branches(i, s);
            const int* trans = kQStateTrans[s];
            size_t a;
            int16_t q;
            int64_t cost;
            if (!b.has1) {
                cost = b.c0 + Cw[trans[0]];
                a = 0;
                q = 0;
            } else {
                const int64_t k0 = b.c0 + Cw[trans[b.a0 & 1]], k1 = b.c1 + Cw[trans[(b.a0 + 1) & 1]];
                if (k0 <= k1) {
                    a = b.a0;
                    q = b.q0;
                    cost = k0;
                } else {
                    a = b.a0 + 1;
                    q = b.q1;
                    cost = k1;
                }
            }
            if ((i & 15) == 15 && tz && a == 0) cost -= lambda * rd.dq[1];
            nc[s] = cost;
            Aw[s] = (uint32_t)a;
            Qw[s] = q;
        }
        for (int s = 0; s < 4; ++s) Cw[s] = nc[s];
    };
    auto step = [&](int i) {
        for (int s = 0; s < 4; ++s) {
            const bool tz = (s == 0 && i <= istar);
            const Br b = branches(i, s);
            const int* trans = kQStateTrans[s];
            size_t a;
            int16_t q;
            int64_t cost;
            if (!b.has1) {
                cost = b.c0 + (i == N - 1 ? 0 : C[(size_t)(i + 1) * 4 + trans[0]]);
                a = 0;
                q = 0;
            } else {
                const int64_t k0 = b.c0 + (i == N - 1 ? 0 : C[(size_t)(i + 1) * 4 + trans[b.a0 & 1]]);
                const int64_t k1 = b.c1 + (i == N - 1 ? 0 : C[(size_t)(i + 1) * 4 + trans[(b.a0 + 1) & 1]]);
                if (k0 <= k1) {
                    a = b.a0;
                    q = b.q0;
                    cost = k0;
                } else {
                    a = b.a0 + 1;
                    q = b.q1;
                    cost = k1;
                }
            }
            if ((i & 15) == 15 && tz && a == 0) cost -= lambda * rd.dq[1];
            C[(size_t)i * 4 + s] = cost;
            A[(size_t)i * 4 + s] = (uint32_t)a;
            Q[(size_t)i * 4 + s] = q;
        }
    };
    int start = 0; // the forward trace starts here in state 0; everything before it is zero
    // (S) the long non-trailing part of a chain in FOUR SEGMENTS walked side by side.  Behind istar's sub-block the
    // recursion is (min, +)-linear in the costs it starts from (the rebate of :512-514 only touches trailing state-0
    // nodes).  The bottom segment starts from the true costs; each of the others is walked from the four vectors e_t = (0
    // for state t, BIG elsewhere) at once.  Every cost vector v with differences below BIG is min_t (v[t] + e_t), so the true
    // walk is the same combination of the four -- and where the four have MERGED (equal costs up to a constant at a
    // sub-block boundary) they are one walk: its decisions are the true ones from there on, and its costs (up to a
    // constant) the true costs the next segment starts from.  Only the sub-blocks between a segment's bottom and its
    // merge point are walked again, from the true costs.
    int seg_done_above = N / 16; // sub-blocks >= this index are decided already
    if (use_seg) {
        const int sb_tz = std::min(istar, N - 1) >> 4;         // sub-blocks <= sb_tz hold trailing nodes: not linear
        const int n_lin = N / 16 - 1 - sb_tz;
        if (n_lin >= 8) {
            const int64_t BIG = (int64_t)1 << 40;
            int bound[5]; // segment k = sub-blocks [bound[k], bound[k + 1]), k = 3 at the DC end
            for (int k = 0; k <= 4; ++k) bound[k] = sb_tz + 1 + (n_lin * k) / 4;
            std::vector<int64_t> Cw(16 * 4);                   // [seg * 4 + basis][state]
            std::vector<uint32_t> Ab((size_t)N * 4 * 4);       // basis walks' decisions [basis][i][s]
            std::vector<int16_t> Qb((size_t)N * 4 * 4);
            int merged_at[4] = {-1, -1, -1, -1};               // sub-block whose walk ended with the four merged
            int64_t top[4][4];                                 // a segment's costs at its top (differences matter)
            for (int k = 0; k < 4; ++k) {
                const int nb_ = k == 3 ? 1 : 4;
                for (int t = 0; t < nb_; ++t)
                    for (int s2 = 0; s2 < 4; ++s2) Cw[(k * 4 + t) * 4 + s2] = k == 3 ? 0 : (s2 == t ? 0 : BIG);
                for (int sb = bound[k + 1] - 1; sb >= bound[k]; --sb) {
                    for (int i = 16 * sb + 15; i >= 16 * sb; --i)
                        for (int t = 0; t < nb_; ++t)
                            step_on(i, &Cw[(k * 4 + t) * 4], &Ab[((size_t)t * N + i) * 4], &Qb[((size_t)t * N + i) * 4]);
                    for (int t = 0; t < nb_; ++t) { // renormalise, as the device does
                        int64_t* c4 = &Cw[(k * 4 + t) * 4];
                        const int64_t m = std::min(std::min(c4[0], c4[1]), std::min(c4[2], c4[3]));
                        for (int s2 = 0; s2 < 4; ++s2) c4[s2] -= m;
                    }
                    if (k < 3 && merged_at[k] < 0) {
                        bool same = true;
                        for (int t = 1; t < 4; ++t)
                            for (int s2 = 0; s2 < 4; ++s2) same = same && Cw[(k * 4 + t) * 4 + s2] == Cw[(k * 4) * 4 + s2];
                        if (same) merged_at[k] = sb;
                    }
                }
                for (int s2 = 0; s2 < 4; ++s2) top[k][s2] = Cw[(k * 4) * 4 + s2];
            }
            // true walks: segment 3 is one already; the others from the true costs at their bottom up to their merge point
            for (int i = 16 * bound[3]; i < N; ++i)
                for (int s2 = 0; s2 < 4; ++s2) {
                    A[(size_t)i * 4 + s2] = Ab[((size_t)0 * N + i) * 4 + s2];
                    Q[(size_t)i * 4 + s2] = Qb[((size_t)0 * N + i) * 4 + s2];
                }
            for (int k = 2; k >= 0; --k) {
                int64_t c4[4];
                for (int s2 = 0; s2 < 4; ++s2) c4[s2] = top[k + 1][s2];
                const int redo_to = merged_at[k] < 0 ? bound[k] : merged_at[k]; // sub-blocks [redo_to, bound[k + 1]) again
                for (int sb = bound[k + 1] - 1; sb >= redo_to; --sb) {
                    for (int i = 16 * sb + 15; i >= 16 * sb; --i) step_on(i, c4, &A[(size_t)i * 4], &Q[(size_t)i * 4]);
                    const int64_t m = std::min(std::min(c4[0], c4[1]), std::min(c4[2], c4[3]));
                    for (int s2 = 0; s2 < 4; ++s2) c4[s2] -= m;
                }
                for (int i = 16 * bound[k]; i < 16 * redo_to; ++i) // beyond the merge point: the merged walk's decisions
                    for (int s2 = 0; s2 < 4; ++s2) {
                        A[(size_t)i * 4 + s2] = Ab[((size_t)0 * N + i) * 4 + s2];
                        Q[(size_t)i * 4 + s2] = Qb[((size_t)0 * N + i) * 4 + s2];
                    }
                if (merged_at[k] < 0)
                    for (int s2 = 0; s2 < 4; ++s2) top[k][s2] = c4[s2]; // never merged: the true walk went all the way
                if (g_sc_stats_on) {
                    st.seg_sb += bound[k + 1] - bound[k];              // (statistics: sub-blocks of segments 0..2 ...
                    st.seg_kept += redo_to - bound[k];                 //  ... and how many were not walked again)
                }
            }
            for (int s2 = 0; s2 < 4; ++s2) C[(size_t)(16 * bound[0]) * 4 + s2] = top[0][s2];
            seg_done_above = bound[0];
        }
    }
    for (int sb = N / 16 - 1; sb >= 0; --sb) {
        const int base = 16 * sb;
        if (sb >= seg_done_above) continue;
        if (use_head && sb < sb_star) {
            // (H): the exact costs at 16 (sb + 1) are known; try to prove the rest of the head, [0, 16 (sb + 1)), zero
            // (first at the end of the head; after a failure again one sub-block further up, with the minima of what is left)
            const int head_end = 16 * (sb + 1);
            if (g_sc_stats_on) st.head_tests++;
            int64_t amin = INT64_MAX;
            for (int i = 0; i < head_end; ++i) amin = std::min(amin, alpha[i]);
            const int64_t* Cb = &C[(size_t)head_end * 4];
            const int64_t G = std::min(std::min(Cb[1], Cb[2]), Cb[3]) - Cb[0];
            const bool ok = lambda * rd.dq[1] >= 0 && (amin == INT64_MAX || G >= -amin); // (alpha, beta >= 0 in the region)
            if (ok) {
                if (g_sc_stats_on) st.head_sb_skipped += sb + 1;
                start = head_end;
                break;
            }
            if (g_sc_stats_on) st.head_fail++;
        }
        bool done = false;
        if (use_z && sb > sb_star && sb < N / 16 - 1) {
            bool elig = true;
            for (int j = 0; j < 16; ++j) elig = elig && qd[base + j] == 0;
            if (elig) {
                if (g_sc_stats_on) st.z_eligible++;
                const int64_t* Cb = &C[(size_t)(base + 16) * 4];
                int64_t sum = 0;
                bool pass = true;
                for (int i = base + 15, j = 0; i >= base; --i, ++j) {
                    const Br b0 = branches(i, 1), b1 = branches(i, 2);
                    sum += b0.c0;
                    if (!b0.has1) continue;
                    const int64_t m0 = b0.c1 - b0.c0, m1 = b1.c1 - b1.c0;
                    const int64_t x = (j & 1) ? Cb[1] : Cb[2], y = (j & 1) ? Cb[2] : Cb[1];
                    pass = pass && std::llabs(Cb[0] - x) <= m0 && std::llabs(y - Cb[3]) <= m1;
                }
                if (pass) {
                    if (g_sc_stats_on) st.z_pass++;
                    for (int s = 0; s < 4; ++s) C[(size_t)base * 4 + s] = Cb[s] + sum;
                    // (the costs inside the sub-block are not needed: A and Q stay zero, the state maps are [0,2,1,3])
                    done = true;
                }
            }
        }
        if (!done) {
            if (g_sc_stats_on) st.walked++;
            for (int i = base + 15; i >= base; --i) step(i);
        }
    }
    int s = 0;
    for (int i = 0; i < N; ++i) {
        if (i < start) {
            levels[py[i] * g.n + px[i]] = 0;
            continue;
        }
        levels[py[i] * g.n + px[i]] = Q[(size_t)i * 4 + s];
        s = kQStateTrans[s][A[(size_t)i * 4 + s] & 1];
    }
}

// quantizer.rs:1068-1077
static void dequantize(const int16_t* levels, int log2n, int qp, int16_t* deq) {
    const int n = 1 << log2n;
    const int32_t lsc = level_scale(qp);
    const int bd_shift = quant_bd_shift(log2n);
    const int32_t bd_offset = (1 << bd_shift) >> 1;
    for (int k = 0; k < n * n; ++k) {
        int32_t v = ((int32_t)levels[k] * lsc + bd_offset) >> bd_shift;
        deq[k] = (int16_t)std::min(std::max(v, -32768), 32767);
    }
}

// block_splitter.rs:415-460 (one component)
static int64_t level_cost(const RdConst& rd, const int16_t* levels, int log2n) {
    ScanGeom g(log2n);
    int64_t sum = 0;
    int q_state = 0;
    int last_scan_pos = g.num_sb_coeff;
    int last_sub_block = g.num_sb - 1;
    bool is_not_first_sub_block = last_sub_block > 0;
    bool is_trailing_zeros = true;
    do {
        if (last_scan_pos == 0) {
            last_scan_pos = g.num_sb_coeff;
            last_sub_block -= 1;
            is_not_first_sub_block = last_sub_block > 0;
        }
        last_scan_pos -= 1;
        int xc, yc;
        g.pos(last_sub_block, last_scan_pos, xc, yc);
        const size_t qc = (size_t)std::abs((int)levels[yc * g.n + xc]);
        if (qc == 0) {
            sum += is_trailing_zeros ? 0 : rd.lv[0];
            q_state = kQStateTrans[q_state][0];
        } else {
            const size_t a = (qc + (q_state > 1 ? 1 : 0)) / 2;
            sum += rd.lv[tbl(a)];
            q_state = kQStateTrans[q_state][a & 1];
        }
        is_trailing_zeros = is_trailing_zeros && (qc == 0);
    } while (last_scan_pos > 0 || is_not_first_sub_block);
    return sum;
}

// ---------------------------------------------------------------------------
// Data model (ctu.rs, tile.rs)
// ---------------------------------------------------------------------------
enum TreeType { SINGLE_TREE = 0, DUAL_TREE_LUMA = 1, DUAL_TREE_CHROMA = 2 };
enum { MODE_TYPE_ALL = 4, MODE_TYPE_INTRA = 6 };
enum { PLANAR = 0, DC = 1, LT_CCLM = 81, L_CCLM = 82, T_CCLM = 83 };

struct Node;

// CodingUnit + its single TransformTree/TransformUnit (ctu.rs:324-497,1190-1358)
struct CU {
    int x, y, w, h; // luma units
    TreeType tree;
    Node* parent;
    int ipm[3] = {PLANAR, PLANAR, PLANAR};    // CodingUnit.intra_pred_mode (:1328)
    int tu_ipm[3] = {PLANAR, PLANAR, PLANAR}; // TransformUnit.cu_intra_pred_mode (:361,:421)
    // TU coefficient buffers (ctu.rs:340-344), one per component, row-major n*n
    std::vector<int16_t> resid[3], coef[3], lev[3], deq[3], itr[3];
    bool active(int c) const { // ctu.rs:499-505
        if (tree == DUAL_TREE_LUMA) return c == 0;
        if (tree == DUAL_TREE_CHROMA) return c != 0;
        return true;
    }
    int csize(int c) const { return c == 0 ? w : w / 2; }
    int cx(int c) const { return c == 0 ? x : x / 2; }
    int cy(int c) const { return c == 0 ? y : y / 2; }
    bool cclm_flag() const { return ipm[1] >= LT_CCLM && ipm[1] <= T_CCLM; } // ctu.rs:1411-1416
    int cclm_idx() const { return cclm_flag() ? ipm[1] - LT_CCLM : 0; }      // :1418-1424
    // ctu.rs:1637-1741 with intra_chroma_pred_mode == 4 (:1298), no MIP/IBC/ACT/BDPCM
    int derived_chroma_mode() const {
        if (cclm_flag()) return LT_CCLM + cclm_idx();
        return ipm[0];
    }
    // ctu.rs:1372-1381
    void set_intra_pred_mode(const int m[3]) {
        ipm[0] = m[0];
        ipm[1] = m[1];
        ipm[2] = m[2];
        const int c = derived_chroma_mode();
        ipm[1] = c;
        ipm[2] = c;
        tu_ipm[0] = m[0];
        tu_ipm[1] = m[1];
        tu_ipm[2] = m[2];
    }
};

// CodingTree (ctu.rs:1793-1901)
struct Node {
    int x, y, w, h;
    int depth;
    TreeType tree;
    int mode_type;
    bool split_qt = false;
    std::vector<Node*> cts;
    std::vector<CU*> cus;
    Node* parent = nullptr;
};

struct Picture {
    int W, H, qp, max_depth;
    RdConst rd;
    std::vector<uint8_t> org[3], pred[3], rec[3];
    int stride[3];
    std::vector<std::unique_ptr<Node>> node_pool;
    std::vector<std::unique_ptr<CU>> cu_pool;
    std::vector<Node*> ctu_root; // ctu.ct[0] per CTU (raster)
    int ctu_cols, ctu_rows;
    long final_mismatch = 0;

    Node* new_node(int x, int y, int size, int depth, TreeType tree, int mode_type, Node* parent) {
        node_pool.emplace_back(new Node());
        Node* n = node_pool.back().get();
        n->x = x;
        n->y = y;
        n->w = n->h = size;
        n->depth = depth;
        n->tree = tree;
        n->mode_type = mode_type;
        n->parent = parent;
        // ctu.rs:1882-1899: every new CT owns one CU with one TU
        cu_pool.emplace_back(new CU());
        CU* cu = cu_pool.back().get();
        cu->x = x;
        cu->y = y;
        cu->w = cu->h = size;
        cu->tree = tree;
        cu->parent = n;
        for (int c = 0; c < 3; ++c) {
            const int s = (c == 0) ? size : size / 2;
            const size_t e = (size_t)s * s;
            cu->resid[c].assign(e, 0);
            cu->coef[c].assign(e, 0);
            cu->lev[c].assign(e, 0);
            cu->deq[c].assign(e, 0);
            cu->itr[c].assign(e, 0);
        }
        n->cus.push_back(cu);
        return n;
    }

    // tile.rs:63-83 -> ctu.rs:265-281 -> ctu.rs:2372-2396
    CU* get_cu(int x, int y) const {
        if (x < 0 || y < 0 || x >= W || y >= H) return nullptr;
        const Node* n = ctu_root[(y >> 5) * ctu_cols + (x >> 5)];
        for (;;) {
            if (!n->cts.empty()) {
                const Node* next = nullptr;
                for (const Node* c : n->cts)
                    if (x >= c->x && x < c->x + c->w && y >= c->y && y < c->y + c->h) {
                        next = c;
                        break;
                    }
                if (!next) abort();
                n = next;
            } else {
                for (CU* cu : n->cus)
                    if (x >= cu->x && x < cu->x + cu->w && y >= cu->y && y < cu->y + cu->h)
                        return cu;
                abort();
            }
        }
    }
};

// ctu.rs:2120-2188 (x_tile = y_tile = 0, one tile = picture)
static bool ct_above_right(const Picture& p, const Node* n) {
    if (n->x + n->w >= p.W) return false;
    const Node* ct = n->parent;
    if (ct) {
        if (n->w == ct->w && n->h == ct->h) return ct_above_right(p, ct);
        if (ct->cts.size() > 1) {
            // SPLIT_QT
            if (n->x == ct->x && n->y == ct->y) return 0 < n->y;
            if (n->y == ct->y) return ct_above_right(p, ct);
            if (n->x == ct->x) return true;
            return false;
        }
        return ct_above_right(p, ct);
    }
    return 0 < n->y && n->x + n->w < p.W;
}

// ctu.rs:2083-2118
static bool ct_below_left(const Picture& p, const Node* n) {
    if (n->y + n->h >= p.H) return false;
    const Node* ct = n->parent;
    if (ct) {
        if (ct->cts.size() > 1) {
            if (ct->x < n->x) return false;
            if (n->y + n->h < ct->y + ct->h) return 0 < n->x;
            return ct_below_left(p, ct);
        }
        return ct_below_left(p, ct);
    }
    return false;
}

// TU -> TT -> CU -> CT chains (ctu.rs:525-591,1077-1151,1426-1490): each level
// first applies the same picture-edge test, then defers (single TU/TT/CU).
static bool tu_above_right(const Picture& p, const CU* cu) {
    if (cu->x + cu->w >= p.W) return false;
    return ct_above_right(p, cu->parent);
}
static bool tu_below_left(const Picture& p, const CU* cu) {
    if (cu->y + cu->h >= p.H) return false;
    return ct_below_left(p, cu->parent);
}

// encoder_context.rs:918-956 (check_pred_mode_y = false, no WPP)
static bool nb_available(const Picture& p, int x_curr, int y_curr, int x_nb, int y_nb, int width,
                         int height, bool above_right, bool below_left) {
    return x_nb >= 0 && y_nb >= 0 && x_nb < p.W && y_nb < p.H &&
           ((x_nb >> 5) <= (x_curr >> 5) || (y_nb >> 5) < (y_curr >> 5)) &&
           (y_nb >> 5) < (y_curr >> 5) + 1 && (x_nb < x_curr + width || above_right) &&
           (y_nb < y_curr + height || below_left);
}

// ctu.rs:1498-1635
static void mpm_flag_idx_rem(const Picture& p, const CU* cu, bool& mpm_flag, int& mpm_idx,
                             int& mpm_rem) {
    if (cu->ipm[0] == PLANAR) {
        mpm_flag = true;
        mpm_idx = 0;
        mpm_rem = 0;
        return;
    }
    const CU* left_cu = p.get_cu(cu->x - 1, cu->y + cu->h - 1);
    const CU* above_cu = p.get_cu(cu->x + cu->w - 1, cu->y - 1);
    const int left = left_cu ? left_cu->ipm[0] : PLANAR;
    int above;
    if (above_cu) {
        if (cu->y - 1 < ((cu->y >> 5) << 5))
            above = PLANAR;
        else
            above = above_cu->ipm[0];
    } else {
        above = PLANAR;
    }
    int cand[5];
    if (left == above && left > DC) {
        const int m = left;
        cand[0] = m;
        cand[1] = 2 + (m + 61) % 64;
        cand[2] = 2 + (m - 1) % 64;
        cand[3] = 2 + (m + 60) % 64;
        cand[4] = 2 + m % 64;
    } else if (left != above && (left > DC || above > DC)) {
        const int mn = std::min(left, above), mx = std::max(left, above);
        if (mn > DC) {
            const int d = mx - mn;
            cand[0] = left;
            cand[1] = above;
            if (d == 1) {
                cand[2] = 2 + (mn + 61) % 64;
                cand[3] = 2 + (mx - 1) % 64;
                cand[4] = 2 + (mn + 60) % 64;
            } else if (d >= 62) {
                cand[2] = 2 + (mn - 1) % 64;
                cand[3] = 2 + (mx + 61) % 64;
                cand[4] = 2 + mn % 64;
            } else if (d == 2) {
                cand[2] = 2 + (mn - 1) % 64;
                cand[3] = 2 + (mn + 61) % 64;
                cand[4] = 2 + (mx - 1) % 64;
            } else {
                cand[2] = 2 + (mn + 61) % 64;
                cand[3] = 2 + (mn - 1) % 64;
                cand[4] = 2 + (mx + 61) % 64;
            }
        } else {
            cand[0] = mx;
            cand[1] = 2 + (mx + 61) % 64;
            cand[2] = 2 + (mx - 1) % 64;
            cand[3] = 2 + (mx + 60) % 64;
            cand[4] = 2 + mx % 64;
        }
    } else {
        cand[0] = DC;
        cand[1] = 50;
        cand[2] = 18;
        cand[3] = 46;
        cand[4] = 54;
    }
    const int mode = cu->ipm[0];
    for (int i = 0; i < 5; ++i)
        if (cand[i] == mode) {
            mpm_flag = true;
            mpm_idx = i;
            mpm_rem = 0;
            return;
        }
    std::sort(cand, cand + 5);
    mpm_flag = false;
    mpm_idx = 0;
    if (mode > cand[4])
        mpm_rem = mode - 6;
    else if (mode > cand[3])
        mpm_rem = mode - 5;
    else if (mode > cand[2])
        mpm_rem = mode - 4;
    else if (mode > cand[1])
        mpm_rem = mode - 3;
    else if (mode > cand[0])
        mpm_rem = mode - 2;
    else
        mpm_rem = mode - 1;
}

// ---------------------------------------------------------------------------
// Intra prediction (intra_predictor.rs)
// ---------------------------------------------------------------------------
struct Predictor {
    Picture& p;
    int16_t left_ref[130], above_ref[129];   // :13-14
    int16_t left_f[130] = {0}, above_f[129] = {0}; // filtered (:15-16), persistent
    int16_t ref_l[64][64], ref_t[64][64];    // :11-12
    explicit Predictor(Picture& pic) : p(pic) {}

    inline uint8_t& P(int c, int x, int y) { return p.pred[c][(size_t)y * p.stride[c] + x]; }
    inline uint8_t R(int c, int x, int y) const { return p.rec[c][(size_t)y * p.stride[c] + x]; }
    inline uint8_t O(int c, int x, int y) const { return p.org[c][(size_t)y * p.stride[c] + x]; }

    // intra_predictor.rs:146-353 (ref_idx = 0, no ISP)
    void set_refs(const CU* tu, int c) {
        const int xt = tu->cx(c), yt = tu->cy(c);
        const int nw = tu->csize(c), nh = nw;
        const int ref_w = nw * 2, ref_h = nh * 2;
        const int mode = tu->tu_ipm[c];
        const bool ref_filter_flag = (mode == 0 || mode == 2 || mode == 34 || mode == 66); // :185-188
        const int nl = ref_h + 1, na = ref_w;
        for (int i = 0; i < nl; ++i) left_ref[i] = -1;
        for (int i = 0; i < na; ++i) above_ref[i] = -1;
        const bool ar = tu_above_right(p, tu), bl = tu_below_left(p, tu);
        bool available = true;
        const int cs = (c != 0) ? 1 : 0;
        {
            const int x_nb_cmp = xt - 1;
            const int x_nb_y = x_nb_cmp * (1 << cs);
            for (int y = -1; y <= ref_h - 1; ++y) {
                const int y_nb_cmp = yt + y;
                const int y_nb_y = y_nb_cmp * (1 << cs);
                if (y == -1 || y % 4 == 0)
                    available = nb_available(p, tu->x, tu->y, x_nb_y, y_nb_y, tu->w, tu->h, ar, bl);
                if (available) left_ref[y + 1] = (int16_t)R(c, x_nb_cmp, y_nb_cmp);
            }
        }
        {
            const int y_nb_cmp = yt - 1;
            const int y_nb_y = y_nb_cmp * (1 << cs);
            const int row = y_nb_cmp >= 0 ? y_nb_cmp : 0;
            for (int x = 0; x <= ref_w - 1; ++x) {
                const int x_nb_cmp = xt + x;
                const int x_nb_y = x_nb_cmp * (1 << cs);
                if (x == 0 || x % 4 == 0)
                    available = nb_available(p, tu->x, tu->y, x_nb_y, y_nb_y, tu->w, tu->h, ar, bl);
                if (available) above_ref[x] = (int16_t)R(c, x_nb_cmp, row);
            }
        }
        // substitution :263-302
        bool left_all = true, above_all = true;
        for (int i = 0; i < nl; ++i) left_all = left_all && left_ref[i] < 0;
        for (int i = 0; i < na; ++i) above_all = above_all && above_ref[i] < 0;
        if (left_all && above_all) {
            for (int i = 0; i < nl; ++i) left_ref[i] = 128;
            for (int i = 0; i < na; ++i) above_ref[i] = 128;
        } else {
            if (left_ref[nl - 1] < 0) {
                bool found = false;
                for (int i = nl - 2; i >= 0; --i)
                    if (left_ref[i] >= 0) {
                        left_ref[nl - 1] = left_ref[i];
                        found = true;
                        break;
                    }
                if (!found)
                    for (int i = 0; i < na; ++i)
                        if (above_ref[i] >= 0) {
                            left_ref[nl - 1] = above_ref[i];
                            break;
                        }
            }
            for (int y = ref_h - 2; y >= -1; --y)
                if (left_ref[y + 1] < 0) left_ref[y + 1] = left_ref[y + 2];
        }
        if (above_ref[0] < 0) above_ref[0] = left_ref[0];
        for (int x = 1; x <= ref_w - 1; ++x)
            if (above_ref[x] < 0) above_ref[x] = above_ref[x - 1];
        // filtering :304-352
        const bool filter_flag = nw * nh > 32 && c == 0 && ref_filter_flag;
        if (filter_flag) {
            left_f[0] = (int16_t)((left_ref[1] + 2 * left_ref[0] + above_ref[0] + 2) >> 2);
            for (int y = 0; y < ref_h - 1; ++y)
                left_f[1 + y] = (int16_t)((left_ref[2 + y] + 2 * left_ref[1 + y] + left_ref[y] + 2) >> 2);
            left_f[ref_h] = left_ref[ref_h];
            above_f[0] = (int16_t)((left_ref[0] + 2 * above_ref[0] + above_ref[1] + 2) >> 2);
            for (int x = 0; x < ref_w - 2; ++x)
                above_f[1 + x] =
                    (int16_t)((above_ref[x] + 2 * above_ref[x + 1] + above_ref[x + 2] + 2) >> 2);
            above_f[ref_w - 1] = above_ref[ref_w - 1];
        } else {
            for (int i = 0; i < nl; ++i) left_f[i] = left_ref[i];
            for (int i = 0; i < na; ++i) above_f[i] = above_ref[i];
        }
    }

    // intra_predictor.rs:355-757 (scalar arm :745-755)
    void pdpc(const int16_t* above, const int16_t* left, int16_t alrs, const CU* tu, int c,
              int pred_mode, int inv_angle) {
        const int tw = tu->csize(c), th = tw;
        const int tx = tu->cx(c), ty = tu->cy(c);
        int n_scale;
        if (pred_mode > 50)
            n_scale = std::min(ilog2(th) - ilog2(3 * inv_angle - 2) + 8, 2);
        else if (pred_mode > 1 && pred_mode < 18)
            n_scale = std::min(ilog2(tw) - ilog2(3 * inv_angle - 2) + 8, 2);
        else
            n_scale = (ilog2(tw) + ilog2(th) - 2) >> 2;
        static const int16_t zero_w[64] = {0};
        const int16_t *w_l, *w_t;
        if (pred_mode < 2) {
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) {
                    ref_l[y][x] = left[y];
                    ref_t[y][x] = above[x];
                }
            w_l = g_pdpc_w[n_scale];
            w_t = g_pdpc_w[n_scale];
        } else if (pred_mode == 18 || pred_mode == 50) {
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) {
                    const int16_t pp = (int16_t)P(c, tx + x, ty + y);
                    ref_l[y][x] = (int16_t)(left[y] - alrs + pp);
                    ref_t[y][x] = (int16_t)(above[x] - alrs + pp);
                }
            w_l = pred_mode == 50 ? g_pdpc_w[n_scale] : zero_w;
            w_t = pred_mode == 18 ? g_pdpc_w[n_scale] : zero_w;
        } else if (pred_mode < 18 && n_scale >= 0) {
            for (int y = 0; y < th; ++y) {
                const int16_t dx_int = (int16_t)(((y + 1) * inv_angle + 256) >> 9);
                for (int x = 0; x < tw; ++x) {
                    ref_l[y][x] = 0;
                    ref_t[y][x] = (y < (3 << n_scale)) ? above[(int16_t)(x + dx_int)] : (int16_t)0;
                }
            }
            w_l = zero_w;
            w_t = g_pdpc_w[n_scale];
        } else if (pred_mode > 50 && n_scale >= 0) {
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) {
                    const int16_t dy_int = (int16_t)(((x + 1) * inv_angle + 256) >> 9);
                    ref_t[y][x] = 0;
                    ref_l[y][x] = (x < (3 << n_scale)) ? left[(int16_t)(y + dy_int)] : (int16_t)0;
                }
            w_l = g_pdpc_w[n_scale];
            w_t = zero_w;
        } else {
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) {
                    ref_l[y][x] = 0;
                    ref_t[y][x] = 0;
                }
            w_l = zero_w;
            w_t = zero_w;
        }
        for (int y = 0; y < th; ++y) {
            const int16_t w_ty = w_t[y];
            const int16_t neg_w_ty = (int16_t)(64 - w_ty);
            for (int x = 0; x < tw; ++x) {
                uint8_t& tp = P(c, tx + x, ty + y);
                // i16 arithmetic with release-mode wrap (:747-752)
                int16_t v = (int16_t)(ref_l[y][x] * w_l[x]);
                v = (int16_t)(v + (int16_t)(ref_t[y][x] * w_ty));
                v = (int16_t)(v + (int16_t)((int16_t)(neg_w_ty - w_l[x]) * (int16_t)tp));
                v = (int16_t)(v + (g_perturb == 1 ? 31 : 32));
                v = (int16_t)(v >> 6);
                tp = (uint8_t)std::min<int>(std::max<int>(v, 0), 255);
            }
        }
    }

    // intra_predictor.rs:759-1146 (square arm, scalar :929-940,:1090-1099)
    void predict_planar(const CU* tu, int c) {
        set_refs(tu, c);
        const int tw = tu->csize(c), th = tw;
        const int tx = tu->cx(c), ty = tu->cy(c);
        const int16_t alrs = left_f[0];
        const int16_t* lrs = left_f + 1;
        const int16_t* ars = above_f;
        const int16_t ars_r = ars[tw];
        const int16_t lrs_b = lrs[th];
        const int shift = ilog2(tw) + 1;
        for (int y = 0; y < th; ++y) {
            const int16_t rv = (int16_t)(th - 1 - y);
            const int16_t ry = (int16_t)((int16_t)(y + 1) * lrs_b);
            for (int x = 0; x < tw; ++x) {
                const int16_t rx = (int16_t)((int16_t)(x + 1) * ars_r);
                const int16_t pv = (int16_t)((int16_t)(rv * ars[x]) + ry);
                const int16_t ph = (int16_t)((int16_t)((int16_t)(tw - 1 - x) * lrs[y]) + rx);
                const int16_t val = (int16_t)((int16_t)((int16_t)(pv + ph) + (int16_t)tw) >> shift);
                P(c, tx + x, ty + y) = (uint8_t)val;
            }
        }
        pdpc(ars, lrs, alrs, tu, c, PLANAR, 0); // tw>=4 && th>=4 always (:1133)
    }

    // intra_predictor.rs:1148-1285
    void predict_dc(const CU* tu, int c) {
        set_refs(tu, c);
        const int tw = tu->csize(c), th = tw;
        const int tx = tu->cx(c), ty = tu->cy(c);
        const int16_t alrs = left_f[0];
        const int16_t* lrs = left_f + 1;
        const int16_t* ars = above_f;
        int16_t v = (int16_t)tw;
        int16_t sa = 0, sl = 0;
        for (int i = 0; i < tw; ++i) sa = (int16_t)(sa + ars[i]);
        for (int i = 0; i < th; ++i) sl = (int16_t)(sl + lrs[i]);
        v = (int16_t)(v + (int16_t)(sa + sl));
        const uint8_t dc_val = (uint8_t)(int16_t)(v >> (ilog2(tw) + 1));
        for (int y = 0; y < th; ++y)
            for (int x = 0; x < tw; ++x) P(c, tx + x, ty + y) = dc_val;
        pdpc(ars, lrs, alrs, tu, c, DC, 0);
    }

    // intra_predictor.rs:1287-1602 (square blocks: no wide-angle remap)
    void predict_angular(const CU* tu, int c) {
        const int mode = tu->tu_ipm[c];
        set_refs(tu, c);
        const int tw = tu->csize(c), th = tw;
        const int tx = tu->cx(c), ty = tu->cy(c);
        const int n_tb_w = tw, n_tb_h = th, ref_w = 2 * tw, ref_h = 2 * th;
        const int16_t* lrs = left_f; // index 0 = corner
        const int16_t alrs = lrs[0];
        const int16_t* ars = above_f;
        const int n_tb_s = (ilog2(tw) + ilog2(th)) >> 1;
        const bool ref_filter_flag = (mode == 0 || mode == 2 || mode == 34 || mode == 66);
        bool filter_flag;
        if (ref_filter_flag) {
            filter_flag = false;
        } else {
            const int md = std::min(std::abs(mode - 50), std::abs(mode - 18));
            int thr;
            switch (n_tb_s) {
            case 2: thr = 24; break;
            case 3: thr = 14; break;
            case 4: thr = 2; break;
            case 5: thr = 0; break;
            case 6: thr = 0; break;
            default: abort();
            }
            filter_flag = md > thr;
        }
        const int angle = kIntraAngle[14 + mode];
        int inv_angle;
        if (angle > 0)
            inv_angle = (512 * 32 + angle / 2) / angle;
        else if (angle < 0)
            inv_angle = -((512 * 32 + (-angle) / 2) / -angle);
        else
            inv_angle = 0;
        std::vector<int16_t> refx;
        if (mode >= 34) {
            refx.assign((size_t)tw + 2, 0);
            refx[0] = alrs;
            for (int x = 0; x <= tw; ++x) refx[x + 1] = ars[x];
            if (angle < 0) {
                for (int x = -n_tb_h; x <= -1; ++x)
                    refx.push_back(lrs[std::min((x * inv_angle + 256) >> 9, n_tb_h)]);
            } else {
                for (int x = n_tb_w + 2; x < ref_w; ++x) refx.push_back(ars[x - 1]);
                for (int k = 1; k <= 3; ++k) refx.push_back(ars[ref_w - 1]);
            }
            const int len = (int)refx.size();
            for (int y = 0; y < th; ++y) {
                const int i_idx = ((y + 1) * angle) >> 5;
                const int i_fact = ((y + 1) * angle) & 31;
                if (c == 0) {
                    const int* f = filter_flag ? g_fg[i_fact] : kFC[i_fact];
                    for (int x = 0; x < tw; ++x) {
                        long s = 0;
                        for (int i = 0; i <= 3; ++i) {
                            int idx = x + i_idx + i;
                            if (idx < 0) idx = len + idx;
                            s += (long)f[i] * (long)refx[idx];
                        }
                        P(c, tx + x, ty + y) = (uint8_t)std::min<long>(std::max<long>((s + 32) >> 6, 0), 255);
                    }
                } else if (i_fact != 0) {
                    for (int x = 0; x < tw; ++x) {
                        int idx0 = x + i_idx + 1;
                        if (idx0 < 0) idx0 = len + idx0;
                        int idx1 = x + i_idx + 2;
                        if (idx1 < 0) idx1 = len + idx1;
                        P(c, tx + x, ty + y) =
                            (uint8_t)(((32 - i_fact) * (long)refx[idx0] + i_fact * (long)refx[idx1] + 16) >> 5);
                    }
                } else {
                    for (int x = 0; x < tw; ++x) {
                        int idx = x + i_idx + 1;
                        if (idx < 0) idx = len + idx;
                        P(c, tx + x, ty + y) = (uint8_t)refx[idx];
                    }
                }
            }
        } else {
            refx.assign((size_t)n_tb_h + 2, 0);
            for (int x = 0; x <= n_tb_h + 1; ++x) refx[x] = lrs[x];
            if (angle < 0) {
                for (int x = -n_tb_w; x <= -1; ++x) {
                    const int idx = std::min((x * inv_angle + 256) >> 9, n_tb_w);
                    refx.push_back(idx == 0 ? alrs : ars[idx - 1]);
                }
            } else {
                for (int x = n_tb_h + 2; x <= ref_h; ++x) refx.push_back(lrs[x]);
                for (int k = 1; k <= 2; ++k) refx.push_back(lrs[ref_h]);
            }
            const int len = (int)refx.size();
            for (int x = 0; x < tw; ++x) {
                const int i_idx = ((x + 1) * angle) >> 5;
                const int i_fact = ((x + 1) * angle) & 31;
                if (c == 0) {
                    const int* f = filter_flag ? g_fg[i_fact] : kFC[i_fact];
                    for (int y = 0; y < th; ++y) {
                        long s = 0;
                        for (int i = 0; i <= 3; ++i) {
                            int idx = y + i_idx + i;
                            if (idx < 0) idx = len + idx;
                            s += (long)f[i] * (long)refx[idx];
                        }
                        P(c, tx + x, ty + y) = (uint8_t)std::min<long>(std::max<long>((s + 32) >> 6, 0), 255);
                    }
                } else if (i_fact != 0) {
                    for (int y = 0; y < th; ++y) {
                        int idx0 = y + i_idx + 1;
                        if (idx0 < 0) idx0 = len + idx0;
                        int idx1 = y + i_idx + 2;
                        if (idx1 < 0) idx1 = len + idx1;
                        P(c, tx + x, ty + y) =
                            (uint8_t)(((32 - i_fact) * (long)refx[idx0] + i_fact * (long)refx[idx1] + 16) >> 5);
                    }
                } else {
                    for (int y = 0; y < th; ++y) {
                        int idx = y + i_idx + 1;
                        if (idx < 0) idx = len + idx;
                        P(c, tx + x, ty + y) = (uint8_t)refx[idx];
                    }
                }
            }
        }
        if (mode <= 18 || (mode >= 50 && mode < LT_CCLM)) // :1571-1578
            pdpc(ars, lrs + 1, alrs, tu, c, mode, inv_angle);
    }

    // intra_predictor.rs:1604-2055 (4:2:0, not vertically collocated)
    void predict_cclm(const CU* tu, int c) {
        const int tw = tu->csize(c), th = tw;
        const int tx = tu->cx(c), ty = tu->cy(c);
        const int mode = tu->tu_ipm[c];
        const bool avail_l = nb_available(p, tu->x, tu->y, tu->x - 1, tu->y, tu->w, tu->h, false, false);
        const bool avail_t = nb_available(p, tu->x, tu->y, tu->x, tu->y - 1, tu->w, tu->h, false, false);
        int num_top_right = 0;
        if (mode == T_CCLM) {
            const bool ar = tu_above_right(p, tu), bl = tu_below_left(p, tu);
            bool avail_tr = true;
            for (int x = tw; x < 2 * tw; ++x) {
                if (!avail_tr) break;
                avail_tr = nb_available(p, tu->x, tu->y, tu->x + x * 2, tu->y - 1, tu->w, tu->h, ar, bl);
                if (avail_tr) ++num_top_right;
            }
        }
        int num_below_left = 0;
        if (mode == L_CCLM) {
            const bool ar = tu_above_right(p, tu), bl = tu_below_left(p, tu);
            bool avail_bl = true;
            for (int y = th; y < 2 * th; ++y) {
                if (!avail_bl) break;
                avail_bl = nb_available(p, tu->x, tu->y, tu->x - 1, tu->y + y * 2, tu->w, tu->h, ar, bl);
                if (avail_bl) ++num_below_left;
            }
        }
        int num_samp_t, num_samp_l;
        if (mode == LT_CCLM) {
            num_samp_t = avail_t ? tw : 0;
            num_samp_l = avail_l ? th : 0;
        } else {
            num_samp_t = (avail_t && mode == T_CCLM) ? tw + std::min(num_top_right, th) : 0;
            num_samp_l = (avail_l && mode == L_CCLM) ? th + std::min(num_below_left, tw) : 0;
        }
        const bool b_ctu_boundary = (tu->y & 31) == 0;
        const int num_is_4 = !(avail_t && avail_l && mode == LT_CCLM) ? 1 : 0;
        const int start_pos_t = num_samp_t >> (2 + num_is_4);
        const int pick_step_t = std::max(num_samp_t >> (1 + num_is_4), 1);
        int cnt_t = 0, pick_pos_t[4] = {0, 0, 0, 0};
        if (avail_t && (mode == LT_CCLM || mode == T_CCLM)) {
            cnt_t = std::min((1 + num_is_4) << 1, num_samp_t);
            for (int i = 0; i < cnt_t; ++i) pick_pos_t[i] = start_pos_t + i * pick_step_t;
        }
        const int start_pos_l = num_samp_l >> (2 + num_is_4);
        const int pick_step_l = std::max(num_samp_l >> (1 + num_is_4), 1);
        int cnt_l = 0, pick_pos_l[4] = {0, 0, 0, 0};
        if (avail_l && (mode == LT_CCLM || mode == L_CCLM)) {
            cnt_l = std::min((1 + num_is_4) << 1, num_samp_l);
            for (int i = 0; i < cnt_l; ++i) pick_pos_l[i] = start_pos_l + i * pick_step_l;
        }
        if (num_samp_l == 0 && num_samp_t == 0) {
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) P(c, tx + x, ty + y) = 128;
            return;
        }
        const int dim = tu->h + tu->w + 3;
        std::vector<long> win((size_t)dim * dim, 0);
        const int ox = 3, oy = 3;
        auto Wn = [&](int y, int x) -> long& { return win[(size_t)(y + oy) * dim + (x + ox)]; };
        for (int y = 0; y < tu->h; ++y)
            for (int x = 0; x < tu->w; ++x) Wn(y, x) = R(0, tu->x + x, tu->y + y);
        if (avail_l)
            for (int y = (avail_t ? -1 : 0); y < 2 * std::max(num_samp_l, th); ++y)
                for (int x = -3; x <= -1; ++x) Wn(y, x) = R(0, tu->x + x, tu->y + y);
        if (!avail_t)
            for (int y = -2; y <= -1; ++y)
                for (int x = -2; x < tu->w; ++x) Wn(y, x) = Wn(0, x);
        if (avail_t)
            for (int y = -3; y <= -1; ++y)
                for (int x = (avail_l ? -1 : 0); x < 2 * std::max(num_samp_t, tw); ++x)
                    Wn(y, x) = R(0, tu->x + x, tu->y + y);
        if (!avail_l)
            for (int y = -2; y < 2 * th; ++y) Wn(y, -1) = Wn(y, 0);
        std::vector<long> pds((size_t)tw * th);
        for (int y = 0; y < th; ++y)
            for (int x = 0; x < tw; ++x) {
                const int sx = 2 * x, sy = 2 * y;
                pds[(size_t)y * tw + x] = (Wn(sy, sx - 1) + Wn(sy + 1, sx - 1) + Wn(sy, sx) * 2 +
                                          Wn(sy + 1, sx) * 2 + Wn(sy, sx + 1) + Wn(sy + 1, sx + 1) + (g_perturb == 2 ? 3 : 4)) >> 3;
            }
        long sel_y[4] = {0, 0, 0, 0}, sel_c[4] = {0, 0, 0, 0};
        if (num_samp_t > 0) {
            for (int i = 0; i < cnt_t; ++i) sel_c[i] = R(c, tx + pick_pos_t[i], ty - 1);
            for (int i = 0; i < cnt_t; ++i) {
                const int sx = 2 * pick_pos_t[i];
                if (!b_ctu_boundary)
                    sel_y[i] = (Wn(-1, sx - 1) + Wn(-2, sx - 1) + Wn(-1, sx) * 2 + Wn(-2, sx) * 2 +
                                Wn(-1, sx + 1) + Wn(-2, sx + 1) + 4) >> 3;
                else
                    sel_y[i] = (Wn(-1, sx - 1) + Wn(-1, sx) * 2 + Wn(-1, sx + 1) + 2) >> 2;
            }
        }
        if (num_samp_l > 0) {
            for (int i = cnt_t; i < cnt_t + cnt_l; ++i) sel_c[i] = R(c, tx - 1, ty + pick_pos_l[i - cnt_t]);
            for (int i = cnt_t; i < cnt_t + cnt_l; ++i) {
                const int sx = -2;
                const int sy = 2 * pick_pos_l[i - cnt_t];
                sel_y[i] = (Wn(sy, sx - 1) + Wn(sy + 1, sx - 1) + Wn(sy, sx) * 2 + Wn(sy + 1, sx) * 2 +
                            Wn(sy, sx + 1) + Wn(sy + 1, sx + 1) + 4) >> 3;
            }
        }
        if (cnt_t + cnt_l == 2) abort(); // :1967-1972 unreachable for 4:2:0 >= 4x4
        int min_grp[2] = {0, 2}, max_grp[2] = {1, 3};
        if (sel_y[min_grp[0]] > sel_y[min_grp[1]]) std::swap(min_grp[0], min_grp[1]);
        if (sel_y[max_grp[0]] > sel_y[max_grp[1]]) std::swap(max_grp[0], max_grp[1]);
        if (sel_y[min_grp[0]] > sel_y[max_grp[1]]) {
            std::swap(min_grp[0], max_grp[0]);
            std::swap(min_grp[1], max_grp[1]);
        }
        if (sel_y[min_grp[1]] > sel_y[max_grp[0]]) std::swap(min_grp[1], max_grp[0]);
        const long max_y = (sel_y[max_grp[0]] + sel_y[max_grp[1]] + 1) >> 1;
        const long max_c = (sel_c[max_grp[0]] + sel_c[max_grp[1]] + 1) >> 1;
        const long min_y = (sel_y[min_grp[0]] + sel_y[min_grp[1]] + 1) >> 1;
        const long min_c = (sel_c[min_grp[0]] + sel_c[min_grp[1]] + 1) >> 1;
        const long diff = max_y - min_y;
        long a, b;
        int k;
        if (diff != 0) {
            const long diff_c = max_c - min_c;
            int x = ilog2((int)diff);
            const long norm_diff = ((diff << 4) >> x) & 15;
            x += (norm_diff != 0) ? 1 : 0;
            const int y = std::labs(diff_c) > 0 ? ilog2((int)std::labs(diff_c)) + 1 : 0;
            static const long div_sig[16] = {0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0};
            a = (diff_c == 0) ? 0 : (diff_c * (div_sig[norm_diff] | 8) + (1L << (y - 1))) >> y;
            if (3 + x - y < 1) {
                k = 1;
                a = a < 0 ? -15 : (a > 0 ? 15 : 0);
            } else {
                k = 3 + x - y;
            }
            b = min_c - ((a * min_y) >> k);
        } else {
            a = 0;
            k = 0;
            b = min_c;
        }
        for (int y = 0; y < th; ++y)
            for (int x = 0; x < tw; ++x) {
                const long v = ((pds[(size_t)y * tw + x] * a) >> k) + b;
                P(c, tx + x, ty + y) = (uint8_t)std::min<long>(std::max<long>(v, 0), 255);
            }
    }

    // intra_predictor.rs:56-144
    void predict(CU* tu, int c) {
        const int mode = tu->tu_ipm[c];
        if (mode == PLANAR)
            predict_planar(tu, c);
        else if (mode == DC)
            predict_dc(tu, c);
        else if (mode <= 66)
            predict_angular(tu, c);
        else
            predict_cclm(tu, c);
        const int tw = tu->csize(c), tx = tu->cx(c), ty = tu->cy(c);
        for (int y = 0; y < tw; ++y)
            for (int x = 0; x < tw; ++x)
                tu->resid[c][(size_t)y * tw + x] =
                    (int16_t)((int16_t)O(c, tx + x, ty + y) - (int16_t)P(c, tx + x, ty + y));
    }
};

// Optional trace of every candidate evaluation of the search (tests compare it with the GPU's):
// kind 0 get_intra_pred_aux_cost, 1 get_intra_pred_cost, 2 get_chroma_intra_pred_aux_cost,
// 3 get_chroma_intra_pred_cost; value = the f32 the function returns.
struct TraceRec {
    int32_t x, y, log2n, tree, kind, ml, mc;
    float value;
};
static bool g_trace_on = false;
static std::vector<TraceRec> g_trace;
static inline float trace_put(int x, int y, int w, int tree, int kind, int ml, int mc, float v) {
    if (g_trace_on) g_trace.push_back(TraceRec{x, y, ilog2(w), tree, kind, ml, mc, v});
    return v;
}

static unsigned long long g_dbg_ssd = 0;
static long long g_dbg_level = 0;

// ---------------------------------------------------------------------------
// RD search (block_splitter.rs)
// ---------------------------------------------------------------------------
struct Splitter {
    Picture& p;
    Predictor ip;
    explicit Splitter(Picture& pic) : p(pic), ip(pic) {}

    // predict -> T -> Q -> DQ -> IT for one component (block_splitter.rs:148-160)
    void code_component(CU* tu, int c) {
        const int log2n = ilog2(tu->csize(c));
        ip.predict(tu, c);
        fwd_dct(tu->resid[c].data(), log2n, tu->coef[c].data());
        quantize(p.rd, tu->coef[c].data(), log2n, p.qp, tu->lev[c].data());
        if (g_sc_stats_on) { // round-4 model check on the search's own blocks (tests / tools only)
            std::vector<int16_t> alt(tu->lev[c].size());
            quantize_viterbi_sc(p.rd, tu->coef[c].data(), log2n, p.qp, alt.data(), true, true, false);
            if (alt != tu->lev[c]) ++g_sc_mismatch;
            g_sc_stats_on = false; // (the segmented walk's statistics are its own two counters; the others count once)
            DqScStats& st_ = g_sc_stats[log2n];
            const DqScStats keep = st_;
            g_sc_stats_on = true;
            quantize_viterbi_sc(p.rd, tu->coef[c].data(), log2n, p.qp, alt.data(), false, false, true);
            if (alt != tu->lev[c]) ++g_sc_mismatch;
            const long long sb_ = st_.seg_sb, kept_ = st_.seg_kept;
            st_ = keep;
            st_.seg_sb = sb_;
            st_.seg_kept = kept_;
        }
        dequantize(tu->lev[c].data(), log2n, p.qp, tu->deq[c].data());
        inv_dct(tu->deq[c].data(), log2n, tu->itr[c].data());
    }

    // rec = clamp(pred + res) and SSD (block_splitter.rs:170-183)
    size_t reconstruct(CU* tu, int c, bool with_ssd) {
        const int tw = tu->csize(c), tx = tu->cx(c), ty = tu->cy(c);
        size_t ssd = 0;
        for (int y = 0; y < tw; ++y)
            for (int x = 0; x < tw; ++x) {
                const size_t o = (size_t)(ty + y) * p.stride[c] + tx + x;
                const int16_t v = (int16_t)((int16_t)p.pred[c][o] + tu->itr[c][(size_t)y * tw + x]);
                const uint8_t rec = (uint8_t)std::min<int>(std::max<int>(v, 0), 255);
                p.rec[c][o] = rec;
                if (with_ssd) {
                    const int d = (int)rec - (int)p.org[c][o];
                    ssd += (size_t)(d * d);
                }
            }
        return ssd;
    }

    size_t sad(CU* tu, int c) {
        const int tw = tu->csize(c), tx = tu->cx(c), ty = tu->cy(c);
        size_t s = 0;
        for (int y = 0; y < tw; ++y)
            for (int x = 0; x < tw; ++x) {
                const size_t o = (size_t)(ty + y) * p.stride[c] + tx + x;
                s += (size_t)std::abs((int)p.pred[c][o] - (int)p.org[c][o]);
            }
        return s;
    }

    // block_splitter.rs:64-108
    float get_intra_pred_aux_cost(const int mode[3], Node* ct) {
        CU* cu = ct->cus[0];
        cu->set_intra_pred_mode(mode);
        size_t s = 0;
        for (int c = 0; c < 3; ++c)
            if (cu->active(c)) {
                ip.predict(cu, c);
                s += sad(cu, c);
            }
        return trace_put(cu->x, cu->y, cu->w, ct->tree, 0, mode[0], mode[1], (float)s);
    }

    // block_splitter.rs:110-474
    float get_intra_pred_cost(const int mode[3], Node* ct) {
        CU* cu = ct->cus[0];
        const int tree_type = ct->tree;
        cu->set_intra_pred_mode(mode);
        const bool non_planar = cu->ipm[0] != PLANAR;
        bool mpm_flag;
        int mpm_idx, mpm_rem;
        mpm_flag_idx_rem(p, cu, mpm_flag, mpm_idx, mpm_rem);
        const bool cclm_flag = cu->cclm_flag();
        const int cclm_idx = cu->cclm_idx();
        size_t ssd = 0;
        for (int c = 0; c < 3; ++c)
            if (cu->active(c)) {
                code_component(cu, c);
                ssd += reconstruct(cu, c, true);
            }
        const int64_t hb = header_bits_luma(p.rd, tree_type, non_planar, mpm_flag, mpm_idx, mpm_rem,
                                            cclm_flag, cclm_idx);
        int64_t sum = 0;
        for (int c = 0; c < 3; ++c) {
            if (!cu->active(c)) continue;
            sum += level_cost(p.rd, cu->lev[c].data(), ilog2(cu->csize(c)));
        }
        const int64_t level = sum + hb;
        const float lambda = rd_lambda(p.rd);
        g_dbg_ssd = ssd;
        g_dbg_level = level;
        return trace_put(cu->x, cu->y, cu->w, tree_type, 1, mode[0], mode[1], (float)ssd + lambda * ((float)level / 16384.0f));
    }

    // block_splitter.rs:476-522
    float get_chroma_intra_pred_aux_cost(int m, Node* ct) {
        CU* cu = ct->cus[0];
        int mode[3] = {cu->ipm[0], m, m};
        cu->set_intra_pred_mode(mode);
        size_t s = 0;
        for (int c = 1; c < 3; ++c)
            if (cu->active(c)) {
                ip.predict(cu, c);
                s += sad(cu, c);
            }
        return trace_put(cu->x, cu->y, cu->w, ct->tree, 2, 0, m, (float)s);
    }

    // block_splitter.rs:524-780
    float get_chroma_intra_pred_cost(int m, Node* ct) {
        CU* cu = ct->cus[0];
        int mode[3] = {cu->ipm[0], m, m};
        cu->set_intra_pred_mode(mode);
        const bool cclm_flag = cu->cclm_flag();
        const int cclm_idx = cu->cclm_idx();
        size_t ssd = 0;
        for (int c = 1; c < 3; ++c)
            if (cu->active(c)) {
                code_component(cu, c);
                ssd += reconstruct(cu, c, true);
            }
        if (ct->tree == DUAL_TREE_LUMA) abort(); // :709
        const int64_t hb = header_bits_chroma(p.rd, cclm_flag, cclm_idx);
        int64_t sum = 0;
        for (int c = 1; c < 3; ++c) sum += level_cost(p.rd, cu->lev[c].data(), ilog2(cu->csize(1)));
        const int64_t level = sum + hb;
        const float lambda = rd_lambda_chroma(p.rd);
        return trace_put(cu->x, cu->y, cu->w, ct->tree, 3, 0, m, (float)ssd + lambda * ((float)level / 16384.0f));
    }

    void cache_reconsts(const Node* ct, int c0, int c1, std::vector<uint8_t> out[3]) {
        for (int c = c0; c < c1; ++c) {
            const int cx = c == 0 ? ct->x : ct->x / 2, cy = c == 0 ? ct->y : ct->y / 2;
            const int cw = c == 0 ? ct->w : ct->w / 2;
            out[c].resize((size_t)cw * cw);
            for (int y = 0; y < cw; ++y)
                for (int x = 0; x < cw; ++x)
                    out[c][(size_t)y * cw + x] = p.rec[c][(size_t)(cy + y) * p.stride[c] + cx + x];
        }
    }
    void restore_reconsts(const Node* ct, int c0, int c1, const std::vector<uint8_t> in[3]) {
        for (int c = c0; c < c1; ++c) {
            const int cx = c == 0 ? ct->x : ct->x / 2, cy = c == 0 ? ct->y : ct->y / 2;
            const int cw = c == 0 ? ct->w : ct->w / 2;
            for (int y = 0; y < cw; ++y)
                for (int x = 0; x < cw; ++x)
                    p.rec[c][(size_t)(cy + y) * p.stride[c] + cx + x] = in[c][(size_t)y * cw + x];
        }
    }

    static float fmin_fold(const float* v, int n) { // iter().fold(f32::MAX, |m, v| v.min(m))
        float m = 3.40282347e+38f;
        for (int i = 0; i < n; ++i) m = std::fmin(v[i], m);
        return m;
    }
    static int first_eq(const float* v, int n, float m) {
        for (int i = 0; i < n; ++i)
            if (v[i] == m) return i;
        abort();
    }

    // ctu.rs:1960-2064 (SPLIT_QT only)
    void split(Node* self) {
        self->split_qt = true;
        self->cus.clear();
        // get_mode_type_condition (ctu.rs:1923-1958): 1 iff 8x8 QT split of a MODE_TYPE_ALL node
        int mode_type_condition = 0;
        if (self->mode_type == MODE_TYPE_ALL && self->w * self->h == 64) mode_type_condition = 1;
        const int mode_type = mode_type_condition == 1 ? MODE_TYPE_INTRA : self->mode_type;
        TreeType tree_type = mode_type == MODE_TYPE_INTRA ? DUAL_TREE_LUMA : self->tree;
        if (self->w == 8 && self->h == 8 && tree_type == SINGLE_TREE) abort(); // ctu.rs:1995-1997
        self->cts.clear();
        for (int i = 0; i < 4; ++i)
            self->cts.push_back(p.new_node(self->x + (i % 2) * (self->w / 2),
                                           self->y + (i / 2) * (self->h / 2), self->w / 2,
                                           self->depth + 1, tree_type, mode_type, self));
        if (self->mode_type == MODE_TYPE_ALL && mode_type == MODE_TYPE_INTRA)
            self->cts.push_back(p.new_node(self->x, self->y, self->w, self->depth, DUAL_TREE_CHROMA,
                                           mode_type, self));
    }

    // block_splitter.rs:782-1154
    float split_ct(Node* ct, int max_depth) {
        if (max_depth == 0) {
            const TreeType tree_type = ct->tree;
            if (tree_type == DUAL_TREE_CHROMA) {
                Node* par = ct->parent;
                CU* luma_cu = nullptr;
                // parent.get_cu(x + w/2, y + h/2) through the parent's own children (:795-800)
                {
                    const int qx = par->x + par->w / 2, qy = par->y + par->h / 2;
                    const Node* n = par;
                    for (;;) {
                        if (!n->cts.empty()) {
                            const Node* next = nullptr;
                            for (const Node* ch : n->cts)
                                if (qx >= ch->x && qx < ch->x + ch->w && qy >= ch->y && qy < ch->y + ch->h) {
                                    next = ch;
                                    break;
                                }
                            if (!next) abort();
                            n = next;
                        } else {
                            luma_cu = n->cus[0];
                            break;
                        }
                    }
                }
                const int chroma_pred_mode = luma_cu->derived_chroma_mode(); // :801-805
                // cclm_enabled_flag == true (:806)
                const float cclm_lt = get_chroma_intra_pred_aux_cost(LT_CCLM, ct);
                const float cclm_t = get_chroma_intra_pred_aux_cost(T_CCLM, ct);
                const float cclm_l = get_chroma_intra_pred_aux_cost(L_CCLM, ct);
                int cclm_mode;
                if (cclm_lt <= cclm_t && cclm_lt <= cclm_l)
                    cclm_mode = LT_CCLM;
                else if (cclm_t <= cclm_l)
                    cclm_mode = T_CCLM;
                else
                    cclm_mode = L_CCLM;
                const float cclm_cost = get_chroma_intra_pred_cost(cclm_mode, ct);
                std::vector<uint8_t> cache[3];
                cache_reconsts(ct, 1, 3, cache);
                const float current_cost = get_chroma_intra_pred_cost(chroma_pred_mode, ct);
                const float cands[2] = {current_cost, cclm_cost};
                const float mn = fmin_fold(cands, 2);
                const int idx = first_eq(cands, 2, mn);
                if (idx == 1) {
                    const int m3[3] = {cclm_mode, cclm_mode, cclm_mode};
                    ct->cus[0]->set_intra_pred_mode(m3);
                    restore_reconsts(ct, 1, 3, cache);
                }
                return mn;
            }
            static const int cand_modes0[15] = {0, 1, 2, 7, 13, 18, 23, 29, 34, 39, 45, 50, 55, 60, 66};
            float cand_costs[15];
            for (int i = 0; i < 15; ++i) {
                const int m3[3] = {cand_modes0[i], cand_modes0[i], cand_modes0[i]};
                cand_costs[i] = cand_modes0[i] <= 1 ? get_intra_pred_cost(m3, ct)
                                                   : get_intra_pred_aux_cost(m3, ct);
            }
            const float min_dir_cost = fmin_fold(cand_costs + 2, 13);
            const int min_dir_idx = first_eq(cand_costs + 2, 13, min_dir_cost) + 2;
            auto step_search = [&](int current_mode, int step, float current_cost, bool aux,
                                   float& out_cost) -> int {
                if (!aux) {
                    const int m3[3] = {current_mode, current_mode, current_mode};
                    current_cost = get_intra_pred_cost(m3, ct);
                }
                while (step > 0) {
                    float cost0, cost1;
                    if (current_mode < 2 + step) {
                        cost0 = 3.40282347e+38f;
                    } else {
                        const int m = current_mode - step;
                        const int m3[3] = {m, m, m};
                        cost0 = aux ? get_intra_pred_aux_cost(m3, ct) : get_intra_pred_cost(m3, ct);
                    }
                    if (current_mode + step > 66) {
                        cost1 = 3.40282347e+38f;
                    } else {
                        const int m = current_mode + step;
                        const int m3[3] = {m, m, m};
                        cost1 = aux ? get_intra_pred_aux_cost(m3, ct) : get_intra_pred_cost(m3, ct);
                    }
                    const float min_cost = std::fmin(std::fmin(current_cost, cost0), cost1);
                    if (current_cost == min_cost) {
                    } else if (cost0 == min_cost) {
                        current_mode = current_mode - step;
                        current_cost = cost0;
                    } else {
                        current_mode = current_mode + step;
                        current_cost = cost1;
                    }
                    step /= 2;
                }
                out_cost = current_cost;
                return current_mode;
            };
            float tmp_cost, dir_cost;
            int dir_mode = step_search(cand_modes0[min_dir_idx], 2, min_dir_cost, true, tmp_cost);
            dir_mode = step_search(dir_mode, 1, min_dir_cost, false, dir_cost);
            const int cand_modes[3] = {0, 1, dir_mode};
            const float cc[3] = {cand_costs[0], cand_costs[1], dir_cost};
            float min_cost = fmin_fold(cc, 3);
            const int min_idx = first_eq(cc, 3, min_cost);
            CU* cu = ct->cus[0];
            const int mode = cand_modes[min_idx];
            {
                const int m3[3] = {mode, mode, mode};
                cu->set_intra_pred_mode(m3);
            }
            if (cu->active(0)) { // luma re-run :989-1037
                code_component(cu, 0);
                reconstruct(cu, 0, false);
            }
            if (tree_type != DUAL_TREE_LUMA) { // cclm_enabled_flag (:1039)
                const float current_cost = get_chroma_intra_pred_cost(mode, ct);
                const float cclm_lt = get_chroma_intra_pred_aux_cost(LT_CCLM, ct);
                const float cclm_t = get_chroma_intra_pred_aux_cost(T_CCLM, ct);
                const float cclm_l = get_chroma_intra_pred_aux_cost(L_CCLM, ct);
                int cclm_mode;
                if (cclm_lt <= cclm_t && cclm_lt <= cclm_l)
                    cclm_mode = LT_CCLM;
                else if (cclm_t <= cclm_l)
                    cclm_mode = T_CCLM;
                else
                    cclm_mode = L_CCLM;
                const float cclm_cost = get_chroma_intra_pred_cost(cclm_mode, ct);
                const float cands[2] = {current_cost, cclm_cost};
                const float mn = fmin_fold(cands, 2);
                const int idx = first_eq(cands, 2, mn);
                if (idx == 0) {
                    const int m3[3] = {mode, mode, mode};
                    cu->set_intra_pred_mode(m3);
                    min_cost = get_intra_pred_cost(m3, ct);
                } else {
                    const int m3[3] = {mode, cclm_mode, cclm_mode};
                    min_cost = get_intra_pred_cost(m3, ct);
                }
            } else if (mode <= 1) {
                const int m3[3] = {mode, mode, mode};
                min_cost = get_intra_pred_cost(m3, ct);
            }
            return min_cost;
        }
        const float no_split_cost = split_ct(ct, 0);
        // split_ct = Arc::new(Mutex::new(ct.clone())) : shallow clone (ctu.rs:1793 derive(Clone))
        p.node_pool.emplace_back(new Node(*ct));
        Node* sct = p.node_pool.back().get();
        std::vector<uint8_t> no_split[3];
        int c0 = 0, c1 = 3;
        if (ct->tree == DUAL_TREE_LUMA) c1 = 1;
        if (ct->tree == DUAL_TREE_CHROMA) c0 = 1;
        cache_reconsts(ct, c0, c1, no_split);
        split(sct);
        const size_t n = sct->cts.size();
        float split_cost = 0.0f;
        for (size_t i = 0; i < n; ++i) split_cost += split_ct(sct->cts[i], max_depth - 1);
        if (split_cost > no_split_cost) {
            restore_reconsts(ct, c0, c1, no_split);
            return no_split_cost;
        }
        // *ct = split_ct.clone()
        Node* keep_parent = ct->parent; // identical in the clone
        *ct = *sct;
        ct->parent = keep_parent;
        return split_cost;
    }

    // ctu_encoder.rs:1421-1461, in coding order (ctu_encoder.rs:448-465)
    void final_pass(Node* n) {
        if (!n->cts.empty()) {
            for (Node* c : n->cts) final_pass(c);
            return;
        }
        for (CU* cu : n->cus)
            for (int c = 0; c < 3; ++c)
                if (cu->active(c)) {
                    const int tw = cu->csize(c), tx = cu->cx(c), ty = cu->cy(c);
                    std::vector<uint8_t> before((size_t)tw * tw);
                    for (int y = 0; y < tw; ++y)
                        for (int x = 0; x < tw; ++x)
                            before[(size_t)y * tw + x] = p.rec[c][(size_t)(ty + y) * p.stride[c] + tx + x];
                    code_component(cu, c);
                    reconstruct(cu, c, false);
                    for (int y = 0; y < tw; ++y)
                        for (int x = 0; x < tw; ++x)
                            if (before[(size_t)y * tw + x] != p.rec[c][(size_t)(ty + y) * p.stride[c] + tx + x])
                                ++p.final_mismatch;
                }
    }
};

// Decoder-side reconstruction from an output record (tree + modes + levels): predict ->
// dequantize -> inverse transform -> clip, CU by CU in coding order.  This is what a VVC decoder
// does with the bitstream's contents, so "reconstruct(record) == rec planes" is the in-repo form of
// the reference's only end-to-end check (scripts/intergration_test.sh: decoded == reconstructed).
struct RecordView {
    const uint8_t* cu_log2;
    const uint8_t* luma_mode;
    const uint8_t* chroma_mode;
    const int16_t* lev[3];
};

static void recon_component(Picture& p, Splitter& sp, CU* cu, int c, const RecordView& rv) {
    const int tw = cu->csize(c), tx = cu->cx(c), ty = cu->cy(c);
    const int log2n = ilog2(tw);
    sp.ip.predict(cu, c);
    for (int y = 0; y < tw; ++y)
        for (int x = 0; x < tw; ++x)
            cu->lev[c][(size_t)y * tw + x] = rv.lev[c][(size_t)(ty + y) * p.stride[c] + tx + x];
    dequantize(cu->lev[c].data(), log2n, p.qp, cu->deq[c].data());
    inv_dct(cu->deq[c].data(), log2n, cu->itr[c].data());
    sp.reconstruct(cu, c, false);
}

static void recon_tree(Picture& p, Splitter& sp, Node* n, const RecordView& rv) {
    const int w4 = p.W / 4, w8 = p.W / 8;
    const int sz = rv.cu_log2[(n->y / 4) * w4 + n->x / 4];
    if (n->tree == SINGLE_TREE && sz < ilog2(n->w)) {
        sp.split(n);
        for (Node* c : n->cts) recon_tree(p, sp, c, rv);
        return;
    }
    CU* cu = n->cus[0];
    if (n->tree != DUAL_TREE_CHROMA) {
        const int ml = rv.luma_mode[(n->y / 4) * w4 + n->x / 4];
        const int mc = n->tree == SINGLE_TREE ? rv.chroma_mode[(n->y / 8) * w8 + n->x / 8] : ml;
        const int m3[3] = {ml, mc, mc};
        cu->set_intra_pred_mode(m3);
    } else {
        const int mc = rv.chroma_mode[(n->y / 8) * w8 + n->x / 8];
        const int m3[3] = {PLANAR, mc, mc};
        cu->set_intra_pred_mode(m3);
    }
    for (int c = 0; c < 3; ++c)
        if (cu->active(c)) recon_component(p, sp, cu, c, rv);
}

static long g_last_final_mismatch = 0;

static void export_tree(const Picture& p, const Node* n, wro_picture_out* out) {
    if (!n->cts.empty()) {
        for (const Node* c : n->cts) export_tree(p, c, out);
        return;
    }
    for (const CU* cu : n->cus) {
        if (cu->tree != DUAL_TREE_CHROMA) {
            const int w4 = p.W / 4;
            for (int y = cu->y / 4; y < (cu->y + cu->h) / 4; ++y)
                for (int x = cu->x / 4; x < (cu->x + cu->w) / 4; ++x) {
                    out->cu_log2_size[y * w4 + x] = (uint8_t)ilog2(cu->w);
                    out->luma_mode[y * w4 + x] = (uint8_t)cu->ipm[0];
                }
        }
        if (cu->tree != DUAL_TREE_LUMA) {
            const int w8 = p.W / 8;
            for (int y = cu->y / 8; y < (cu->y + cu->h) / 8; ++y)
                for (int x = cu->x / 8; x < (cu->x + cu->w) / 8; ++x)
                    out->chroma_mode[y * w8 + x] = (uint8_t)cu->tu_ipm[1];
        }
        for (int c = 0; c < 3; ++c)
            if (cu->active(c)) {
                int16_t* dst = c == 0 ? out->lev_y : (c == 1 ? out->lev_cb : out->lev_cr);
                const int tw = cu->csize(c), tx = cu->cx(c), ty = cu->cy(c);
                for (int y = 0; y < tw; ++y)
                    for (int x = 0; x < tw; ++x)
                        dst[(size_t)(ty + y) * p.stride[c] + tx + x] = cu->lev[c][(size_t)y * tw + x];
            }
    }
}

} // namespace

extern "C" {

int wro_encode_picture(const wro_params* prm, const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                       wro_picture_out* out) {
    init_tables();
    if (!prm || prm->width <= 0 || prm->height <= 0 || (prm->width & 31) || (prm->height & 31))
        return -1;
    if (prm->max_split_depth < 0 || prm->max_split_depth > 3) return -2;
    if (prm->qp < 0 || prm->qp > 63) return -3;
    g_table_overflow = false;
    Picture p;
    p.W = prm->width;
    p.H = prm->height;
    p.qp = prm->qp;
    p.max_depth = prm->max_split_depth;
    init_rd(p.rd, p.qp);
    p.stride[0] = p.W;
    p.stride[1] = p.stride[2] = p.W / 2;
    const uint8_t* src[3] = {y, cb, cr};
    for (int c = 0; c < 3; ++c) {
        const size_t n = (size_t)p.stride[c] * (c == 0 ? p.H : p.H / 2);
        p.org[c].assign(src[c], src[c] + n);
        p.pred[c].assign(n, 0); // tile.rs:49-58 zero-initialised planes
        p.rec[c].assign(n, 0);
    }
    p.ctu_cols = p.W / 32;
    p.ctu_rows = p.H / 32;
    // picture.rs:70-103 init_ctus: one SINGLE_TREE / MODE_TYPE_ALL root CT per CTU
    for (int r = 0; r < p.ctu_rows; ++r)
        for (int c = 0; c < p.ctu_cols; ++c)
            p.ctu_root.push_back(p.new_node(c * 32, r * 32, 32, 0, SINGLE_TREE, MODE_TYPE_ALL, nullptr));
    // slice_encoder.rs:352-379: CTUs in raster order; per CTU search then emit (final pass)
    for (int i = 0; i < p.ctu_cols * p.ctu_rows; ++i) {
        Splitter sp(p); // ctu_encoder.rs:53 BlockSplitter::new per CTU
        const float cost = sp.split_ct(p.ctu_root[i], p.max_depth);
        if (out && out->ctu_cost) out->ctu_cost[i] = cost;
        sp.final_pass(p.ctu_root[i]);
    }
    g_last_final_mismatch = p.final_mismatch;
    if (g_table_overflow) return -4; // a level reached 1024: the reference would have panicked
    if (out) {
        for (int c = 0; c < 3; ++c) {
            uint8_t* dst = c == 0 ? out->rec_y : (c == 1 ? out->rec_cb : out->rec_cr);
            if (dst) memcpy(dst, p.rec[c].data(), p.rec[c].size());
        }
        if (out->lev_y && out->lev_cb && out->lev_cr && out->cu_log2_size && out->luma_mode &&
            out->chroma_mode)
            for (Node* root : p.ctu_root) export_tree(p, root, out);
    }
    return 0;
}

int wro_reconstruct_from_record(const wro_params* prm, const wro_picture_out* rec, uint8_t* out_y, uint8_t* out_cb,
                                uint8_t* out_cr) {
    init_tables();
    if (!prm || !rec || (prm->width & 31) || (prm->height & 31) || prm->width <= 0 || prm->height <= 0) return -1;
    Picture p;
    p.W = prm->width;
    p.H = prm->height;
    p.qp = prm->qp;
    p.max_depth = prm->max_split_depth;
    init_rd(p.rd, p.qp);
    p.stride[0] = p.W;
    p.stride[1] = p.stride[2] = p.W / 2;
    for (int c = 0; c < 3; ++c) {
        const size_t n = (size_t)p.stride[c] * (c == 0 ? p.H : p.H / 2);
        p.org[c].assign(n, 0);
        p.pred[c].assign(n, 0);
        p.rec[c].assign(n, 0);
    }
    p.ctu_cols = p.W / 32;
    p.ctu_rows = p.H / 32;
    for (int r = 0; r < p.ctu_rows; ++r)
        for (int c = 0; c < p.ctu_cols; ++c)
            p.ctu_root.push_back(p.new_node(c * 32, r * 32, 32, 0, SINGLE_TREE, MODE_TYPE_ALL, nullptr));
    RecordView rv;
    rv.cu_log2 = rec->cu_log2_size;
    rv.luma_mode = rec->luma_mode;
    rv.chroma_mode = rec->chroma_mode;
    rv.lev[0] = rec->lev_y;
    rv.lev[1] = rec->lev_cb;
    rv.lev[2] = rec->lev_cr;
    Splitter sp(p);
    for (Node* root : p.ctu_root) recon_tree(p, sp, root, rv);
    memcpy(out_y, p.rec[0].data(), p.rec[0].size());
    memcpy(out_cb, p.rec[1].data(), p.rec[1].size());
    memcpy(out_cr, p.rec[2].data(), p.rec[2].size());
    return 0;
}

// Prediction of single blocks in the environment of a picture whose reconstruction planes are given
// (kernel-level parity of intra_predictor.rs:56-144 with every mode / size / availability pattern).
// An item is 6 ints: x, y (luma, picture coordinates, multiples of the size), log2 luma size, tree type
// (0 single, 1 dual luma [4x4], 2 dual chroma [8x8 luma area]), component, mode; `out` receives the
// component's predicted block (row-major), items back to back.  The CT chain from the CTU root down to
// the block is made by quad-tree splits, since the availability walkers (ctu.rs:2083-2188) follow it.
int wro_predict_blocks(const wro_params* prm, const uint8_t* rec_y, const uint8_t* rec_cb, const uint8_t* rec_cr,
                       int n_items, const int32_t* items, uint8_t* out) {
    init_tables();
    if (!prm || (prm->width & 31) || (prm->height & 31) || prm->width <= 0 || prm->height <= 0) return -1;
    Picture p;
    p.W = prm->width;
    p.H = prm->height;
    p.qp = prm->qp;
    p.max_depth = 3;
    init_rd(p.rd, p.qp);
    p.stride[0] = p.W;
    p.stride[1] = p.stride[2] = p.W / 2;
    const uint8_t* src[3] = {rec_y, rec_cb, rec_cr};
    for (int c = 0; c < 3; ++c) {
        const size_t n = (size_t)p.stride[c] * (c == 0 ? p.H : p.H / 2);
        p.org[c].assign(n, 0);
        p.pred[c].assign(n, 0);
        p.rec[c].assign(src[c], src[c] + n);
    }
    p.ctu_cols = p.W / 32;
    p.ctu_rows = p.H / 32;
    for (int r = 0; r < p.ctu_rows; ++r)
        for (int c = 0; c < p.ctu_cols; ++c) p.ctu_root.push_back(nullptr);
    Splitter sp(p);
    size_t at = 0;
    for (int it = 0; it < n_items; ++it) {
        const int32_t* q = items + 6 * it;
        const int x = q[0], y = q[1], lg = q[2], tree = q[3], c = q[4], mode = q[5];
        const int n = 1 << lg;
        if (lg < 2 || lg > 5 || x < 0 || y < 0 || x + n > p.W || y + n > p.H || (x & (n - 1)) || (y & (n - 1))) return -2;
        if ((tree == 1 && lg != 2) || (tree == 2 && lg != 3) || (tree == 0 && lg == 2)) return -2;
        Node* node = p.new_node(x & ~31, y & ~31, 32, 0, SINGLE_TREE, MODE_TYPE_ALL, nullptr);
        p.ctu_root[(size_t)(y >> 5) * p.ctu_cols + (x >> 5)] = node;
        const int target = tree == 2 ? 8 : n;
        while (node->w > target) {
            sp.split(node);
            Node* next = nullptr;
            for (int i = 0; i < 4; ++i) {
                Node* ch = node->cts[(size_t)i];
                if (x >= ch->x && x < ch->x + ch->w && y >= ch->y && y < ch->y + ch->h) next = ch;
            }
            node = next;
        }
        if (tree == 2) {
            sp.split(node);
            node = node->cts[4];
        }
        CU* cu = node->cus[0];
        if (!cu->active(c)) return -3;
        const int m[3] = {c == 0 ? mode : PLANAR, mode, mode};
        cu->tu_ipm[0] = m[0]; // prediction reads the TU array
        cu->tu_ipm[1] = m[1];
        cu->tu_ipm[2] = m[2];
        sp.ip.predict(cu, c);
        const int tw = cu->csize(c), tx = cu->cx(c), ty = cu->cy(c);
        for (int yy = 0; yy < tw; ++yy)
            for (int xx = 0; xx < tw; ++xx) out[at++] = p.pred[c][(size_t)(ty + yy) * p.stride[c] + tx + xx];
    }
    return 0;
}

long wro_last_final_pass_mismatches(void) { return g_last_final_mismatch; }

// main.rs:202-217: "K1=V1,K2=V2"; an empty string or NULL restores the defaults.  Returns 0, or -1 when an
// item is not KEY=VALUE (the reference prints "Invalid extra-params" and exits).
int wro_set_extra_params(const char* text) {
    ++g_extra_gen;
    g_extra.clear();
    if (!text || !*text) return 0;
    std::string t = text;
    size_t pos = 0;
    while (pos <= t.size()) {
        size_t end = t.find(',', pos);
        if (end == std::string::npos) end = t.size();
        const std::string item = t.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq == std::string::npos || item.find('=', eq + 1) != std::string::npos) {
            g_extra.clear();
            return -1;
        }
        g_extra[item.substr(0, eq)] = item.substr(eq + 1);
        pos = end + 1;
    }
    return 0;
}

float wro_lambda_rd_chroma(int qp) {
    RdConst rd;
    init_rd(rd, qp);
    return rd_lambda_chroma(rd);
}

void wro_debug_perturb(int which) { g_perturb = which; }

void wro_trace_enable(int on) {
    g_trace_on = on != 0;
    g_trace.clear();
}
// copies up to max records of 8 int32 words (x, y, log2n, tree, kind, ml, mc, value bits); returns the count
long wro_trace_read(int32_t* out, long max) {
    const long n = (long)g_trace.size() < max ? (long)g_trace.size() : max;
    if (n > 0) memcpy(out, g_trace.data(), (size_t)n * sizeof(TraceRec));
    return (long)g_trace.size();
}
void wro_debug_last_cost(unsigned long long* ssd, long long* level) {
    *ssd = g_dbg_ssd;
    *level = g_dbg_level;
}

void wro_fwd_dct(const int16_t* res, int log2n, int16_t* coef) {
    init_tables();
    fwd_dct(res, log2n, coef);
}
void wro_inv_dct(const int16_t* deq, int log2n, int16_t* res) {
    init_tables();
    inv_dct(deq, log2n, res);
}
void wro_quantize(const int16_t* coef, int log2n, int qp, int16_t* levels) {
    init_tables();
    static thread_local RdConst rd;
    static thread_local int rd_qp = -1, rd_gen = -1;
    if (rd_qp != qp || rd_gen != g_extra_gen) {
        init_rd(rd, qp);
        rd_qp = qp;
        rd_gen = g_extra_gen;
    }
    quantize(rd, coef, log2n, qp, levels);
}
void wro_quantize_viterbi(const int16_t* coef, int log2n, int qp, int16_t* levels) {
    init_tables();
    static thread_local RdConst rd;
    static thread_local int rd_qp = -1, rd_gen = -1;
    if (rd_qp != qp || rd_gen != g_extra_gen) {
        init_rd(rd, qp);
        rd_qp = qp;
        rd_gen = g_extra_gen;
    }
    quantize_viterbi(rd, coef, log2n, qp, levels);
}
void wro_quantize_viterbi_sc(const int16_t* coef, int log2n, int qp, int16_t* levels, int use_head, int use_z) {
    init_tables();
    static thread_local RdConst rd;
    static thread_local int rd_qp = -1, rd_gen = -1;
    if (rd_qp != qp || rd_gen != g_extra_gen) {
        init_rd(rd, qp);
        rd_qp = qp;
        rd_gen = g_extra_gen;
    }
    quantize_viterbi_sc(rd, coef, log2n, qp, levels, use_head != 0, (use_z & 1) != 0, (use_z & 2) != 0, (use_z & 4) == 0);
}
void wro_dq_sc_stats_enable(int on) {
    g_sc_stats_on = on != 0;
    if (on) {
        memset(g_sc_stats, 0, sizeof(g_sc_stats));
        g_sc_mismatch = 0;
    }
}
long long wro_dq_sc_stats_read(long long* out72) {
    for (int l = 0; l < 6; ++l) memcpy(out72 + 12 * l, &g_sc_stats[l], 12 * sizeof(long long));
    return g_sc_mismatch;
}
void wro_dequantize(const int16_t* levels, int log2n, int qp, int16_t* deq) {
    dequantize(levels, log2n, qp, deq);
}
int64_t wro_level_cost(const int16_t* levels, int log2n) {
    init_tables();
    static thread_local RdConst rd;
    static thread_local int rd_gen = -1;
    if (rd_gen != g_extra_gen) {
        init_rd(rd, 32);
        rd_gen = g_extra_gen;
    }
    return level_cost(rd, levels, log2n);
}
void wro_tables(int qp, int64_t* lv, int64_t* dq, int64_t* lambda_q, float* lambda_rd) {
    RdConst rd;
    init_rd(rd, qp);
    if (lv) memcpy(lv, rd.lv, sizeof(rd.lv));
    if (dq) memcpy(dq, rd.dq, sizeof(rd.dq));
    if (lambda_q) *lambda_q = rd.lambda_q;
    if (lambda_rd) *lambda_rd = rd_lambda(rd);
}
int64_t wro_header_bits(int tree, int non_planar, int mpm_flag, int mpm_idx, int mpm_rem,
                        int cclm_flag, int cclm_idx) {
    RdConst rd;
    init_rd(rd, 32);
    return header_bits_luma(rd, tree, non_planar != 0, mpm_flag != 0, mpm_idx, mpm_rem, cclm_flag != 0,
                            cclm_idx);
}
int64_t wro_chroma_header_bits(int cclm_flag, int cclm_idx) {
    RdConst rd;
    init_rd(rd, 32);
    return header_bits_chroma(rd, cclm_flag != 0, cclm_idx);
}
void wro_dct64(int16_t* m) {
    init_tables();
    memcpy(m, g_dct64, sizeof(g_dct64));
}

} // extern "C"
