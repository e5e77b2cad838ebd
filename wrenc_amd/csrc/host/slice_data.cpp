// slice_data.cpp -- CTU syntax of one all-intra picture, from the search record to CABAC bins.
//
// Restates, for the syntax that is live under the reference's parameter sets,
//   coding_tree           ctu_encoder.rs:227-438   (quadtree only; 8x8 -> 4x4 opens a local dual tree,
//                                                   ctu.rs:1960-2064)
//   coding_unit           ctu_encoder.rs:440-1323  (MPM signalling ctu.rs:1498-1635, chroma ctu.rs:1637-1741)
//   transform_unit        ctu_encoder.rs:1463-1784
//   residual_coding       ctu_encoder.rs:1786-2269 (dependent quantisation levels, three coding passes)
// with the context selection of bool_coder.rs:2053-2400,2659-2740 and the binarisations of
// bool_coder.rs:1133-1465.  The reference walks its CT/CU/TU object graph; here the same walk runs over
// the flat maps the device returns (include/wrenc_bitstream.h).
#include "slice_data.h"

#include <cstdlib>
#include <cstring>

namespace wrenc_host {

namespace {

enum Tree { SINGLE_TREE = 0, DUAL_TREE_LUMA = 1, DUAL_TREE_CHROMA = 2 };
enum { PLANAR = 0, DC = 1, ANG18 = 18, ANG46 = 46, ANG50 = 50, ANG54 = 54, LT_CCLM = 81 };

// 6.5.2 up-right diagonal scan of a (1 << lw) x (1 << lh) block (the reference's table ctu.rs:14-81)
struct Scan {
    uint8_t x[64], y[64];
};
void make_scan(int lw, int lh, Scan& s) {
    const int w = 1 << lw, h = 1 << lh;
    int i = 0, x = 0, y = 0;
    while (i < w * h) {
        while (y >= 0) {
            if (x < w && y < h) {
                s.x[i] = (uint8_t)x;
                s.y[i] = (uint8_t)y;
                ++i;
            }
            --y;
            ++x;
        }
        y = x;
        x = 0;
    }
}
struct Scans {
    Scan s[4]; // square blocks of side 1, 2, 4, 8
    Scans() {
        for (int l = 0; l < 4; ++l) make_scan(l, l, s[l]);
    }
};
const Scans kScans;

const int kQStateTrans[4][2] = {{0, 2}, {2, 0}, {1, 3}, {3, 1}}; // encoder_context.rs:339
const int kRiceParams[32] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2,
                             2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3}; // cabac_contexts.rs:919

} // namespace

class PictureCoder {
public:
    PictureCoder(int width, int height, int qp, const wrenc_bs_record& rec, BitWriter& bw)
        : W_(width), H_(height), qp_(qp), r_(rec), cabac_(bw) {}
    // the residual syntax as device-made tokens (include/wrenc_gpu.h, wrenc_gpu_download_tokens): the record then
    // carries the maps only
    PictureCoder(int width, int height, int qp, const wrenc_bs_record& maps, const wrenc_bs_tokens& tok, BitWriter& bw)
        : W_(width), H_(height), qp_(qp), r_(maps), cabac_(bw), tok_(&tok) {}

    // ctu_encoder.rs:38-47 (CABAC initialised at the picture's first CTU) + :172-201 (the CTU's coding tree)
    int encode_ctu(int x, int y) {
        if (x == 0 && y == 0) cabac_.start(qp_);
        if (tok_) { // this CTU's run of tokens
            tk_page_ = tok_->first_page[(size_t)(y >> 5) * (W_ >> 5) + (x >> 5)];
            tk_pos_ = 0;
        }
        return coding_tree(x, y, 5);
    }
    // slice_encoder.rs:388-394: end_of_slice_one_bit behind the last CTU
    void end_of_slice() { cabac_.finish(); }

private:
    int leaf_lg(int x, int y) const { return r_.cu_log2_size[(size_t)(y >> 2) * (W_ >> 2) + (x >> 2)]; }
    int luma_mode(int x, int y) const { return r_.luma_mode[(size_t)(y >> 2) * (W_ >> 2) + (x >> 2)]; }

    // ctu_encoder.rs:227-438
    int coding_tree(int x0, int y0, int lg) {
        const int leaf = leaf_lg(x0, y0);
        if (leaf > lg || leaf < 2) return WRENC_BS_EDATA;
        const bool split = leaf < lg;
        // split_cu_flag (:284-296): only the quadtree split is allowed (MTT depth 0), so ctxSetIdx is 0;
        // condL / condA compare the neighbouring leaf's size with this block's (bool_coder.rs:2712-2741)
        const int cond_l = x0 > 0 && leaf_lg(x0 - 1, y0) < lg;
        const int cond_a = y0 > 0 && leaf_lg(x0, y0 - 1) < lg;
        cabac_.encode(CTX_SPLIT_CU + cond_l + cond_a, split);
        if (lg == 5) qp_delta_coded_ = false; // quantisation group = CTU (:304-309, cu_qp_delta_subdiv 0)
        if (!split) return coding_unit(x0, y0, lg, SINGLE_TREE);
        if (lg > 3) {
            for (int i = 0; i < 4; ++i) {
                const int rc = coding_tree(x0 + ((i & 1) << (lg - 1)), y0 + ((i >> 1) << (lg - 1)), lg - 1);
                if (rc) return rc;
            }
            return WRENC_BS_OK;
        }
        // 8x8 -> four 4x4 luma blocks, then the 4x4 chroma block of the whole 8x8 (local dual tree,
        // modeTypeCondition 1: nothing is signalled for it; 4x4 luma and the chroma tree cannot split)
        if (leaf != 2) return WRENC_BS_EDATA;
        for (int i = 0; i < 4; ++i) {
            const int rc = coding_unit(x0 + ((i & 1) << 2), y0 + ((i >> 1) << 2), 2, DUAL_TREE_LUMA);
            if (rc) return rc;
        }
        return coding_unit(x0, y0, 3, DUAL_TREE_CHROMA);
    }

    // candModeList of 8.4.2 as ctu.rs:1498-1598 builds it
    void mpm_list(int x0, int y0, int size, int cand[5]) const {
        const int left = x0 > 0 ? luma_mode(x0 - 1, y0 + size - 1) : PLANAR;
        const int above = (y0 > 0 && y0 - 1 >= ((y0 >> 5) << 5)) ? luma_mode(x0 + size - 1, y0 - 1) : PLANAR;
        if (left == above && left > DC) {
            cand[0] = left;
            cand[1] = 2 + (left + 61) % 64;
            cand[2] = 2 + (left - 1) % 64;
            cand[3] = 2 + (left + 60) % 64;
            cand[4] = 2 + left % 64;
        } else if (left != above && (left > DC || above > DC)) {
            const int mn = left < above ? left : above, mx = left < above ? above : left;
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
            cand[1] = ANG50;
            cand[2] = ANG18;
            cand[3] = ANG46;
            cand[4] = ANG54;
        }
    }

    // ctu_encoder.rs:440-1323 (intra, I slice)
    int coding_unit(int x0, int y0, int lg, Tree tree) {
        const int size = 1 << lg;
        for (int y = y0; y < y0 + size && tree != DUAL_TREE_CHROMA; y += 4)
            for (int x = x0; x < x0 + size; x += 4)
                if (leaf_lg(x, y) != lg) return WRENC_BS_EDATA; // the size map is not a quadtree
        if (tree != DUAL_TREE_CHROMA) {
            const int mode = luma_mode(x0, y0);
            if (mode > 66) return WRENC_BS_EDATA;
            // intra_luma_mpm_flag / not_planar / mpm_idx / mpm_remainder (:755-804)
            if (mode == PLANAR) {
                cabac_.encode(CTX_MPM_FLAG, 1);
                cabac_.encode(CTX_NOT_PLANAR + 1, 0); // ctxInc = !intra_subpartitions_mode_flag
            } else {
                int cand[5];
                mpm_list(x0, y0, size, cand);
                int idx = -1;
                for (int i = 0; i < 5 && idx < 0; ++i)
                    if (cand[i] == mode) idx = i;
                if (idx >= 0) {
                    cabac_.encode(CTX_MPM_FLAG, 1);
                    cabac_.encode(CTX_NOT_PLANAR + 1, 1);
                    for (int i = 0; i < idx; ++i) cabac_.bypass(1); // TR cMax 4, all bins bypass
                    if (idx < 4) cabac_.bypass(0);
                } else {
                    cabac_.encode(CTX_MPM_FLAG, 0);
                    int rem = mode - 1; // planar is the first most probable mode
                    for (int i = 0; i < 5; ++i) rem -= cand[i] < mode;
                    // truncated binary, cMax 60: 3 short codewords of 5 bits, the rest 6 bits
                    if (rem < 3)
                        cabac_.bypass_bits((uint32_t)rem, 5);
                    else
                        cabac_.bypass_bits((uint32_t)rem + 3, 6);
                }
            }
        }
        if (tree != DUAL_TREE_LUMA) {
            const int cm = r_.chroma_mode[(size_t)(y0 >> 3) * (W_ >> 3) + (x0 >> 3)];
            // the chroma tree of an 8x8 takes its direct mode from the luma block covering the centre
            // (block_splitter.rs:795-805)
            const int luma_ref = tree == SINGLE_TREE ? luma_mode(x0, y0) : luma_mode(x0 + size / 2, y0 + size / 2);
            if (cm >= LT_CCLM) {
                if (cm > LT_CCLM + 2) return WRENC_BS_EDATA;
                cabac_.encode(CTX_CCLM_FLAG, 1);
                const int idx = cm - LT_CCLM; // TR cMax 2: first bin coded, second bypass
                cabac_.encode(CTX_CCLM_IDX, idx > 0);
                if (idx > 0) cabac_.bypass(idx > 1);
            } else {
                cabac_.encode(CTX_CCLM_FLAG, 0);
                // intra_chroma_pred_mode (Table 20, ctu.rs:1682-1737): the search only ever picks the
                // direct mode (4); the other four values are accepted for completeness
                int v = -1;
                static const int kFixed[4] = {PLANAR, ANG50, ANG18, DC};
                if (cm == luma_ref) {
                    v = 4;
                } else {
                    for (int i = 0; i < 4; ++i)
                        if (cm == kFixed[i] || (cm == 66 && luma_ref == kFixed[i])) v = i;
                }
                if (v < 0) return WRENC_BS_EDATA;
                if (v == 4) {
                    cabac_.encode(CTX_CHROMA_PRED, 0);
                } else {
                    cabac_.encode(CTX_CHROMA_PRED, 1);
                    cabac_.bypass_bits((uint32_t)v, 2);
                }
            }
        }
        mts_dc_only_ = true; // :1211-1214
        mts_zero_out_ = true;
        const int rc = transform_unit(x0, y0, lg, tree);
        if (rc) return rc;
        // mts_idx = 0 (:1299-1318): one context-coded bin
        if (tree != DUAL_TREE_CHROMA && mts_zero_out_ && !mts_dc_only_) cabac_.encode(CTX_MTS_IDX, 0);
        return WRENC_BS_OK;
    }

    const int16_t* plane(int c) const { return c == 0 ? r_.lev_y : (c == 1 ? r_.lev_cb : r_.lev_cr); }
    int stride(int c) const { return c ? W_ >> 1 : W_; }

    // bit (ys * 8 + xs) is set when the 4x4 sub-block (xs, ys) of the TB holds a level (get_sb_coded_flag,
    // ctu.rs:724-737); zero means the coded flag of the TB is zero (ctu.rs:902-950)
    uint64_t sb_mask(int c, int tx, int ty, int lg) const {
        const int st = stride(c), tw = 1 << lg;
        const int16_t* p = plane(c) + (size_t)ty * st + tx;
        uint64_t mask = 0;
        for (int y = 0; y < tw; ++y, p += st)
            for (int x = 0; x < tw; x += 4) {
                uint64_t four;
                memcpy(&four, p + x, sizeof(four));
                if (four) mask |= 1ull << ((y >> 2) * 8 + (x >> 2));
            }
        return mask;
    }

    // ---- the device's token stream: pages of WRENC_BS_TOKEN_PAGE words, the last one the index of the next page ----
    enum { kPage = WRENC_BS_TOKEN_PAGE, kPayload = kPage - 1 };
    bool next_word(uint32_t& w) {
        if (tk_pos_ == kPayload) {
            if (tk_page_ == 0xFFFFFFFFu) return false;
            tk_page_ = tok_->pool[(size_t)tk_page_ * kPage + kPayload];
            tk_pos_ = 0;
        }
        if (tk_page_ == 0xFFFFFFFFu || ((size_t)tk_page_ + 1) * kPage > tok_->pool_words) return false;
        w = tok_->pool[(size_t)tk_page_ * kPage + tk_pos_++];
        return true;
    }
    // `count` tokens straight into the arithmetic coder
    int splice(uint32_t count) {
        while (count) {
            if (tk_pos_ == kPayload) {
                if (tk_page_ == 0xFFFFFFFFu) return WRENC_BS_EDATA;
                tk_page_ = tok_->pool[(size_t)tk_page_ * kPage + kPayload];
                tk_pos_ = 0;
            }
            if (tk_page_ == 0xFFFFFFFFu || ((size_t)tk_page_ + 1) * kPage > tok_->pool_words) return WRENC_BS_EDATA;
            const uint32_t* p = tok_->pool + (size_t)tk_page_ * kPage + tk_pos_;
            uint32_t run = (uint32_t)(kPayload - tk_pos_);
            run = run < count ? run : count;
            for (uint32_t i = 0; i < run; ++i) {
                const uint32_t t = p[i];
                if ((int32_t)t >= 0) {
                    if ((t >> 1) >= (uint32_t)CTX_COUNT) return WRENC_BS_EDATA;
                    cabac_.encode((int)(t >> 1), (int)(t & 1));
                } else {
                    cabac_.bypass_bits(t & 0x1FFFFFFu, (int)((t >> 25) & 63) + 1);
                }
            }
            tk_pos_ += (int)run;
            count -= run;
        }
        return WRENC_BS_OK;
    }
    // transform_unit with the residuals as tokens: header words of the components present, then their tokens
    int transform_unit_tokens(Tree tree) {
        const bool chroma = tree != DUAL_TREE_LUMA, luma = tree != DUAL_TREE_CHROMA;
        uint32_t h[3] = {0, 0, 0};
        if (luma && !next_word(h[0])) return WRENC_BS_EDATA;
        if (chroma && (!next_word(h[1]) || !next_word(h[2]))) return WRENC_BS_EDATA;
        const bool cbf_y = h[0] >> 31, cbf_cb = h[1] >> 31, cbf_cr = h[2] >> 31;
        if (chroma) {
            cabac_.encode(CTX_CB_CBF, cbf_cb);
            cabac_.encode(CTX_CR_CBF + cbf_cb, cbf_cr);
        }
        if (luma) cabac_.encode(CTX_Y_CBF, cbf_y);
        if ((cbf_y || cbf_cb || cbf_cr) && luma && !qp_delta_coded_) {
            cabac_.encode(CTX_QP_DELTA_ABS, 0);
            qp_delta_coded_ = true;
        }
        for (int c = 0; c < 3; ++c) {
            if (!(h[c] >> 31)) continue;
            cabac_.encode(CTX_TS_FLAG + (c != 0), 0);
            if (c == 0) {
                if (h[0] & (1u << 30)) mts_dc_only_ = false;
                if (h[0] & (1u << 29)) mts_zero_out_ = false;
            }
            const int rc = splice(h[c] & 0xFFFFFFu);
            if (rc) return rc;
        }
        return WRENC_BS_OK;
    }

    // ctu_encoder.rs:1463-1784
    int transform_unit(int x0, int y0, int lg, Tree tree) {
        if (tok_) return transform_unit_tokens(tree);
        const bool chroma = tree != DUAL_TREE_LUMA, luma = tree != DUAL_TREE_CHROMA;
        const uint64_t m_y = luma ? sb_mask(0, x0, y0, lg) : 0;
        const uint64_t m_cb = chroma ? sb_mask(1, x0 >> 1, y0 >> 1, lg - 1) : 0;
        const uint64_t m_cr = chroma ? sb_mask(2, x0 >> 1, y0 >> 1, lg - 1) : 0;
        const bool cbf_y = m_y != 0, cbf_cb = m_cb != 0, cbf_cr = m_cr != 0;
        if (chroma) {
            cabac_.encode(CTX_CB_CBF, cbf_cb);          // ctxInc 0 (no BDPCM)
            cabac_.encode(CTX_CR_CBF + cbf_cb, cbf_cr); // ctxInc = tu_cb_coded_flag
        }
        if (luma) cabac_.encode(CTX_Y_CBF, cbf_y); // ctxInc 0 (no ISP, no BDPCM)
        // cu_qp_delta_abs = 0, once per quantisation group (:1603-1637)
        if ((cbf_y || cbf_cb || cbf_cr) && luma && !qp_delta_coded_) {
            cabac_.encode(CTX_QP_DELTA_ABS, 0);
            qp_delta_coded_ = true;
        }
        for (int c = 0; c < 3; ++c) {
            const uint64_t m = c == 0 ? m_y : (c == 1 ? m_cb : m_cr);
            if (!m) continue;
            cabac_.encode(CTX_TS_FLAG + (c != 0), 0); // transform_skip_flag = 0 (:1693-1775)
            const int rc = c == 0 ? residual(0, x0, y0, lg, m) : residual(c, x0 >> 1, y0 >> 1, lg - 1, m);
            if (rc) return rc;
        }
        return WRENC_BS_OK;
    }

    // The two per-TB arrays of 9.3.4.2.8 / 9.3.3.2 are kept with a row stride of kS and two zero columns
    // and rows after the TB, so the five-neighbour templates need no bounds tests (the tests of
    // bool_coder.rs:1133-1174,2152-2244 only ever exclude positions outside the TB, which read as zero
    // here).  tpl_ packs AbsLevelPass1 with "is significant" in bit 8: one sum gives locSumAbsPass1 (low
    // byte, at most 5 * 5) and locNumSig (bits 8..10).
    enum { kS = 36 };

    int template_sum(int xc, int yc) const {
        const int* t = &tpl_[yc * kS + xc];
        return t[1] + t[2] + t[kS] + t[kS + 1] + t[2 * kS];
    }

    // bool_coder.rs:1133-1174 (9.3.3.2)
    int rice_param(int base_level, int xc, int yc) const {
        const int* t = &abs_[yc * kS + xc];
        int s = t[1] + t[2] + t[kS] + t[kS + 1] + t[2 * kS] - base_level * 5;
        return kRiceParams[s < 0 ? 0 : (s > 31 ? 31 : s)];
    }

    // abs_remainder / dec_abs_level (bool_coder.rs:1384-1465): truncated Rice prefix with cMax 6 << k, then
    // limited Exp-Golomb of order k + 1 (maxPreExtLen 11, truncSuffixLen 15, :1305-1331); all bypass
    void code_remainder(int val, int k) {
        const int c_max = 6 << k;
        const int pv = val < c_max ? val : c_max;
        const int pre = pv >> k;
        if (pre < 6) {
            for (int i = 0; i < pre; ++i) cabac_.bypass(1);
            cabac_.bypass(0);
            if (k > 0) cabac_.bypass_bits((uint32_t)(pv - (pre << k)), k);
            return;
        }
        for (int i = 0; i < 6; ++i) cabac_.bypass(1);
        int sym = val - c_max;
        const int kk = k + 1;
        const int code_value = sym >> kk;
        int pre_ext = 0;
        while (pre_ext < 11 && code_value > (2 << pre_ext) - 2) {
            ++pre_ext;
            cabac_.bypass(1);
        }
        int escape;
        if (pre_ext == 11) {
            escape = 15;
        } else {
            cabac_.bypass(0);
            escape = pre_ext + kk;
        }
        sym -= ((1 << pre_ext) - 1) << kk;
        cabac_.bypass_bits((uint32_t)sym, escape);
    }

    // last_sig_coeff_{x,y}_prefix: truncated unary, cMax = 2 * log2 - 1 (bool_coder.rs:713-740), context
    // (binIdx >> ctxShift) + ctxOffset (:2053-2083)
    void code_last_prefix(int base, int c, int lg, int prefix) {
        static const int kOffsetY[6] = {0, 0, 3, 6, 10, 15};
        int off, shift;
        if (c == 0) {
            off = kOffsetY[lg - 1];
            shift = (lg + 1) >> 2;
        } else {
            off = 20;
            shift = (1 << lg) >> 3;
            shift = shift > 2 ? 2 : shift;
        }
        const int c_max = (lg << 1) - 1;
        for (int i = 0; i < prefix; ++i) cabac_.encode(base + (i >> shift) + off, 1);
        if (prefix < c_max) cabac_.encode(base + (prefix >> shift) + off, 0);
    }

    // (prefix, suffix) of a last-significant coordinate (ctu_encoder.rs:1818-1851)
    static void split_last(int v, int& prefix, int& suffix) {
        if (v <= 3) {
            prefix = v;
            suffix = 0;
            return;
        }
        int bits = 1, p;
        for (;;) {
            p = v >> bits;
            suffix = v - (p << bits);
            if (p < 4) break;
            ++bits;
        }
        prefix = ((bits + 1) << 1) + (p & 1);
    }

    // ctu_encoder.rs:1786-2269; tx, ty, lg in samples of component c; mask = sb_mask of the TB
    int residual(int c, int tx, int ty, int lg, uint64_t mask) {
        const int tw = 1 << lg;
        const int16_t* lev = plane(c) + (size_t)ty * stride(c) + tx;
        const int st = stride(c);
        for (int y = 0; y < tw + 2; ++y) {
            memset(&abs_[y * kS], 0, sizeof(int) * (tw + 2));
            memset(&tpl_[y * kS], 0, sizeof(int) * (tw + 2));
        }
        const Scan& sbs = kScans.s[lg - 2]; // sub-blocks of the TB
        const Scan& cs = kScans.s[2];       // coefficients of a 4x4 sub-block
        const int n_sb = 1 << (2 * (lg - 2));
        // last significant coefficient in scan order (ctu.rs:867-899)
        int last_sb = n_sb - 1, last_pos = 15;
        while (last_sb >= 0 && !((mask >> (sbs.y[last_sb] * 8 + sbs.x[last_sb])) & 1)) --last_sb;
        if (last_sb < 0) return WRENC_BS_EDATA;
        while (!lev[(size_t)((sbs.y[last_sb] << 2) + cs.y[last_pos]) * st + (sbs.x[last_sb] << 2) + cs.x[last_pos]]) --last_pos;
        const int last_x = (sbs.x[last_sb] << 2) + cs.x[last_pos], last_y = (sbs.y[last_sb] << 2) + cs.y[last_pos];
        int px, sx, py, sy;
        split_last(last_x, px, sx);
        split_last(last_y, py, sy);
        code_last_prefix(CTX_LAST_X, c, lg, px);
        code_last_prefix(CTX_LAST_Y, c, lg, py);
        if (px > 3) cabac_.bypass_bits((uint32_t)sx, (px >> 1) - 1);
        if (py > 3) cabac_.bypass_bits((uint32_t)sy, (py >> 1) - 1);

        if ((last_sb > 0 || last_pos > 0) && c == 0) mts_dc_only_ = false; // :1945-1947
        int rem_bins = ((1 << (2 * lg)) * 7) >> 2;
        int q_state = 0;
        const int sb_ctx = CTX_SB_CODED + (c ? 2 : 0);
        const int sbw = tw >> 2;
        for (int i = last_sb; i >= 0; --i) {
            const int xs = sbs.x[i], ys = sbs.y[i];
            const int x_off = xs << 2, y_off = ys << 2;
            const bool sb_coded = ((mask >> (ys * 8 + xs)) & 1) || i == 0; // (:1994) the DC sub-block counts as coded
            bool infer_dc = false;
            if (i < last_sb && i > 0) {
                // sb_coded_flag: context from the right and lower sub-blocks (bool_coder.rs:2102-2150)
                int csbf = 0;
                if (xs < sbw - 1) csbf |= (int)((mask >> (ys * 8 + xs + 1)) & 1);
                if (ys < sbw - 1) csbf |= (int)((mask >> ((ys + 1) * 8 + xs)) & 1);
                cabac_.encode(sb_ctx + csbf, sb_coded);
                infer_dc = true;
            }
            // An uncoded sub-block ends here.  Sixteen zero levels leave the quantiser state where it was
            // (0->0, 1->2->1, 2->1->2, 3->3) and take nothing from the bin budget, so the passes of
            // :2014-2253 would have no effect.
            if (!sb_coded) continue;
            // AbsLevel of the sub-block from TransCoeffLevel: (|q| + (state > 1)) / 2 along the state
            // walk (:1968-1985); the parity check is the reference's release assert.  The state this
            // walk ends in is the one the next sub-block starts from (:2254-2267).
            int a[16];
            uint32_t signs = 0;
            int n_signs = 0, next_q = q_state;
            for (int n = 15; n >= 0; --n) {
                const int v = lev[(size_t)(y_off + cs.y[n]) * st + x_off + cs.x[n]];
                const int av = v < 0 ? -v : v;
                if (av) {
                    if ((av & 1) != (next_q > 1)) return WRENC_BS_EDATA;
                    signs = (signs << 1) | (uint32_t)(v < 0);
                    ++n_signs;
                }
                a[n] = (av + (next_q > 1)) >> 1;
                next_q = kQStateTrans[next_q][a[n] & 1];
            }
            if ((xs > 3 || ys > 3) && c == 0) mts_zero_out_ = false; // :2008-2010
            const int first_pos_mode0 = i == last_sb ? last_pos : 15;
            int first_pos_mode1 = first_pos_mode0;
            bool any_gt3 = false;
            // pass 1: sig_coeff_flag, abs_level_gtx_flag[0], par_level_flag, abs_level_gtx_flag[1]
            for (int n = first_pos_mode0; n >= 0; --n) {
                if (rem_bins < 4) break;
                const int xc = x_off + cs.x[n], yc = y_off + cs.y[n];
                const int an = a[n];
                const bool sig = an > 0;
                const bool is_last = xc == last_x && yc == last_y;
                const int d = xc + yc;
                const int tsum = template_sum(xc, yc);
                const int sum_p1 = tsum & 255, num_sig = tsum >> 8;
                if ((n > 0 || !infer_dc) && !is_last) {
                    const int s = (sum_p1 + 1) >> 1;
                    const int qs = q_state > 1 ? q_state - 1 : 0;
                    const int inc = c == 0 ? 12 * qs + (s < 3 ? s : 3) + (d < 2 ? 8 : (d < 5 ? 4 : 0))
                                           : 36 + 8 * qs + (s < 3 ? s : 3) + (d < 2 ? 4 : 0);
                    cabac_.encode(CTX_SIG + inc, sig);
                    --rem_bins;
                    if (sig) infer_dc = false;
                }
                int pass1 = 0;
                if (sig) {
                    int off = sum_p1 - num_sig;
                    off = off > 4 ? 4 : off;
                    int inc;
                    if (is_last)
                        inc = c == 0 ? 0 : 21;
                    else if (c == 0)
                        inc = 1 + off + (d == 0 ? 15 : (d < 3 ? 10 : (d < 10 ? 5 : 0)));
                    else
                        inc = 22 + off + (d == 0 ? 5 : 0);
                    const bool gt1 = an > 1;
                    cabac_.encode(CTX_GTX + inc, gt1);
                    --rem_bins;
                    pass1 = 1;
                    if (gt1) {
                        const bool gt3 = an > 3;
                        cabac_.encode(CTX_PAR + inc, an & 1);
                        cabac_.encode(CTX_GTX + 32 + inc, gt3);
                        rem_bins -= 2;
                        pass1 = 2 + (an & 1) + 2 * (int)gt3;
                        any_gt3 |= gt3;
                    }
                    tpl_[yc * kS + xc] = pass1 | 256;
                }
                q_state = kQStateTrans[q_state][pass1 & 1];
                first_pos_mode1 = n - 1;
            }
            // pass 2: abs_remainder of the coefficients pass 1 covered
            for (int n = first_pos_mode0; n > first_pos_mode1; --n) {
                if (!a[n]) continue;
                const int xc = x_off + cs.x[n], yc = y_off + cs.y[n];
                if (any_gt3 && a[n] > 3)
                    code_remainder((a[n] - (tpl_[yc * kS + xc] & 255)) >> 1, rice_param(4, xc, yc));
                abs_[yc * kS + xc] = a[n];
            }
            // pass 3: dec_abs_level of what the bin budget left out (ctu.rs:739-782)
            for (int n = first_pos_mode1; n >= 0; --n) {
                const int xc = x_off + cs.x[n], yc = y_off + cs.y[n];
                const int k = rice_param(0, xc, yc);
                const int zero_pos = (q_state < 2 ? 1 : 2) << k;
                code_remainder(a[n] == 0 ? zero_pos : (a[n] <= zero_pos ? a[n] - 1 : a[n]), k);
                abs_[yc * kS + xc] = a[n];
                q_state = kQStateTrans[q_state][a[n] & 1];
            }
            // signs of the sub-block in scan order, bypass
            if (n_signs) cabac_.bypass_bits(signs, n_signs);
            q_state = next_q;
        }
        return WRENC_BS_OK;
    }

    const int W_, H_, qp_;
    const wrenc_bs_record& r_;
    CabacEncoder cabac_;
    bool qp_delta_coded_ = false;
    bool mts_dc_only_ = true, mts_zero_out_ = true;
    int abs_[34 * kS]; // AbsLevel of the current TB
    int tpl_[34 * kS]; // AbsLevelPass1 | significant << 8
    const wrenc_bs_tokens* tok_ = nullptr; // residuals as device-made tokens (else: from the level planes of r_)
    uint32_t tk_page_ = 0xFFFFFFFFu;
    int tk_pos_ = 0;
};

int CtuEncoder::encode(Bins& bins, const Ctu& ctu, const SliceHeader& sh) {
    (void)bins; // the coder was constructed on these bins
    (void)sh;   // and with this slice QP
    return coder_.encode_ctu(ctu.x, ctu.y);
}

// slice_encoder.rs:343-427 with one tile and one slice per picture: CTUs in raster order
static int encode_ctus(const Slice& slice, const SliceHeader& sh, Bins& bins) {
    PictureCoder coder(slice.width, slice.height, sh.slice_qp, *slice.record, bins);
    for (int y = 0; y < slice.height; y += 32)
        for (int x = 0; x < slice.width; x += 32) {
            CtuEncoder ctu_encoder(coder); // slice_encoder.rs:378: one CtuEncoder per CTU
            const Ctu ctu = {x, y};
            const int rc = ctu_encoder.encode(bins, ctu, sh);
            if (rc) return rc;
        }
    coder.end_of_slice();
    return WRENC_BS_OK;
}

Bins SliceEncoder::encode(const Slice& slice, const SliceHeader& sh, int* status) {
    Bins bins;
    write_slice_header(bins, sh.slice_qp); // encode_sh, slice_encoder.rs:32-341 (ends byte aligned)
    const size_t header_bits = bins.bit_count();
    const int rc = encode_ctus(slice, sh, bins);
    slice_data_bits_ = (long long)(bins.bit_count() - header_bits);
    if (status) *status = rc;
    bins.align(); // slice_encoder.rs:418
    return bins;
}

// the same picture loop with the residual syntax read from the device's tokens
int write_slice_data_tokens(int width, int height, int qp, const wrenc_bs_tokens& tok, BitWriter& bw) {
    const wrenc_bs_record maps = {tok.cu_log2_size, tok.luma_mode, tok.chroma_mode, nullptr, nullptr, nullptr};
    PictureCoder coder(width, height, qp, maps, tok, bw);
    for (int y = 0; y < height; y += 32)
        for (int x = 0; x < width; x += 32) {
            const int rc = coder.encode_ctu(x, y);
            if (rc) return rc;
        }
    coder.end_of_slice();
    return WRENC_BS_OK;
}

int write_slice_data(int width, int height, int qp, const wrenc_bs_record& rec, BitWriter& bw) {
    const Slice slice = {width, height, &rec};
    const SliceHeader sh = {qp};
    return encode_ctus(slice, sh, bw);
}

} // namespace wrenc_host
