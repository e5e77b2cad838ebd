// cabac.h -- VVC CABAC encoding engine and the context models the all-intra path uses.
//
// Engine: bool_coder.rs:136-296 (9-bit range from 510, two-rate probability states, bypass,
// terminate, outstanding-bit carry propagation, first output bit suppressed).
// Context initialisation: bool_coder.rs:1073-1093.  Initial values / shift indices: the I-slice
// (init_type 0, bool_coder.rs:299-302) rows of cabac_contexts.rs:243-917 for the syntax elements that
// are live under the reference's parameter sets; these are the values of the
// VVC specification's context tables.
#pragma once
#include <cstdint>

#include "bits.h"

namespace wrenc_host {

// first context of each syntax element in the flat model array
enum CtxBase {
    CTX_SPLIT_CU = 0,                      // 9  split_cu_flag
    CTX_MPM_FLAG = CTX_SPLIT_CU + 9,       // 1  intra_luma_mpm_flag
    CTX_NOT_PLANAR = CTX_MPM_FLAG + 1,     // 2  intra_luma_not_planar_flag
    CTX_CCLM_FLAG = CTX_NOT_PLANAR + 2,    // 1  cclm_mode_flag
    CTX_CCLM_IDX = CTX_CCLM_FLAG + 1,      // 1  cclm_mode_idx
    CTX_CHROMA_PRED = CTX_CCLM_IDX + 1,    // 1  intra_chroma_pred_mode
    CTX_MTS_IDX = CTX_CHROMA_PRED + 1,     // 4  mts_idx
    CTX_Y_CBF = CTX_MTS_IDX + 4,           // 4  tu_y_coded_flag
    CTX_CB_CBF = CTX_Y_CBF + 4,            // 2  tu_cb_coded_flag
    CTX_CR_CBF = CTX_CB_CBF + 2,           // 3  tu_cr_coded_flag
    CTX_QP_DELTA_ABS = CTX_CR_CBF + 3,     // 2  cu_qp_delta_abs
    CTX_TS_FLAG = CTX_QP_DELTA_ABS + 2,    // 2  transform_skip_flag
    CTX_LAST_X = CTX_TS_FLAG + 2,          // 23 last_sig_coeff_x_prefix
    CTX_LAST_Y = CTX_LAST_X + 23,          // 23 last_sig_coeff_y_prefix
    CTX_SB_CODED = CTX_LAST_Y + 23,        // 7  sb_coded_flag
    CTX_SIG = CTX_SB_CODED + 7,            // 63 sig_coeff_flag
    CTX_PAR = CTX_SIG + 63,                // 33 par_level_flag
    CTX_GTX = CTX_PAR + 33,                // 72 abs_level_gtx_flag
    CTX_COUNT = CTX_GTX + 72
};

struct CtxInit {
    uint8_t init_value, shift_idx;
};

// {initValue, shiftIdx} per context, in CtxBase order
static const CtxInit kCtxInit[CTX_COUNT] = {
    // split_cu_flag
    {19, 12}, {28, 13}, {38, 8}, {27, 8}, {29, 13}, {38, 12}, {20, 5}, {30, 9}, {31, 9},
    // intra_luma_mpm_flag
    {45, 6},
    // intra_luma_not_planar_flag
    {13, 1}, {28, 5},
    // cclm_mode_flag, cclm_mode_idx, intra_chroma_pred_mode
    {59, 4}, {27, 9}, {34, 5},
    // mts_idx
    {29, 8}, {0, 0}, {28, 9}, {0, 0},
    // tu_y_coded_flag
    {15, 5}, {12, 1}, {5, 8}, {7, 9},
    // tu_cb_coded_flag
    {12, 5}, {21, 0},
    // tu_cr_coded_flag
    {33, 2}, {28, 1}, {36, 0},
    // cu_qp_delta_abs
    {35, 8}, {35, 8},
    // transform_skip_flag
    {25, 1}, {9, 1},
    // last_sig_coeff_x_prefix
    {13, 8}, {5, 5}, {4, 4}, {21, 5}, {14, 4}, {4, 4}, {6, 5}, {14, 4}, {21, 1}, {11, 0}, {14, 4}, {7, 1},
    {14, 0}, {5, 0}, {11, 0}, {21, 0}, {30, 1}, {22, 0}, {13, 0}, {42, 0}, {12, 5}, {4, 4}, {3, 4},
    // last_sig_coeff_y_prefix
    {13, 8}, {5, 5}, {4, 8}, {6, 5}, {13, 5}, {11, 4}, {14, 5}, {6, 5}, {5, 4}, {3, 0}, {14, 5}, {22, 4},
    {6, 1}, {4, 0}, {3, 0}, {6, 1}, {22, 4}, {29, 0}, {20, 0}, {34, 0}, {12, 6}, {4, 5}, {3, 5},
    // sb_coded_flag
    {18, 8}, {31, 5}, {25, 5}, {15, 8}, {18, 5}, {20, 8}, {38, 8},
    // sig_coeff_flag
    {25, 12}, {19, 9}, {28, 9}, {14, 10}, {25, 9}, {20, 9}, {29, 9}, {30, 10}, {19, 8}, {37, 8}, {30, 8},
    {38, 10}, {11, 9}, {38, 13}, {46, 8}, {54, 8}, {27, 8}, {39, 8}, {39, 8}, {39, 5}, {44, 8}, {39, 0},
    {39, 0}, {39, 0}, {18, 8}, {39, 8}, {39, 8}, {39, 8}, {27, 8}, {39, 0}, {39, 4}, {39, 4}, {0, 0},
    {39, 0}, {39, 0}, {39, 0}, {25, 12}, {27, 12}, {28, 9}, {37, 13}, {34, 4}, {53, 5}, {53, 8}, {46, 9},
    {19, 8}, {46, 12}, {38, 12}, {39, 8}, {52, 4}, {39, 0}, {39, 0}, {39, 0}, {11, 8}, {39, 8}, {39, 8},
    {39, 8}, {19, 4}, {39, 0}, {39, 0}, {39, 0}, {25, 13}, {28, 13}, {38, 8},
    // par_level_flag
    {33, 8}, {25, 9}, {18, 12}, {26, 13}, {34, 13}, {27, 13}, {25, 10}, {26, 13}, {19, 13}, {42, 13},
    {35, 13}, {33, 13}, {19, 13}, {27, 13}, {35, 13}, {35, 13}, {34, 10}, {42, 13}, {20, 13}, {43, 13},
    {20, 13}, {33, 8}, {25, 12}, {26, 12}, {42, 12}, {19, 13}, {27, 13}, {26, 13}, {50, 13}, {35, 13},
    {20, 13}, {43, 13}, {11, 6},
    // abs_level_gtx_flag
    {25, 9}, {25, 5}, {11, 10}, {27, 13}, {20, 13}, {21, 10}, {33, 9}, {12, 10}, {28, 13}, {21, 13},
    {22, 13}, {34, 9}, {28, 10}, {29, 10}, {29, 10}, {30, 13}, {36, 8}, {29, 9}, {45, 10}, {30, 10},
    {23, 13}, {40, 8}, {33, 8}, {27, 9}, {28, 12}, {21, 12}, {37, 10}, {36, 5}, {37, 9}, {45, 9}, {38, 9},
    {46, 13}, {25, 1}, {1, 5}, {40, 9}, {25, 9}, {33, 9}, {11, 6}, {17, 5}, {25, 9}, {25, 10}, {18, 10},
    {4, 9}, {17, 9}, {33, 9}, {26, 9}, {19, 9}, {13, 9}, {33, 6}, {19, 8}, {20, 9}, {28, 9}, {22, 10},
    {40, 1}, {9, 5}, {25, 8}, {18, 8}, {26, 9}, {35, 6}, {25, 6}, {26, 9}, {35, 8}, {28, 8}, {37, 9},
    {11, 4}, {5, 2}, {5, 1}, {14, 6}, {10, 1}, {3, 1}, {3, 1}, {3, 1},
};

struct CtxModel {
    uint16_t s0, s1;       // 10-bit fast and 14-bit slow probability estimates
    uint8_t shift0, shift1;
    uint16_t add0, add1;   // 1023 >> shift0, 16383 >> shift1: what a bin equal to one adds to each estimate
};

// bool_coder.rs:1073-1093 (and :142-144 for the two adaptation rates)
inline void init_models(CtxModel* m, int slice_qp) {
    const int qp = slice_qp < 0 ? 0 : (slice_qp > 63 ? 63 : slice_qp);
    for (int i = 0; i < CTX_COUNT; ++i) {
        const int slope = (kCtxInit[i].init_value >> 3) - 4;
        const int offset = (kCtxInit[i].init_value & 7) * 18 + 1;
        int pre = ((slope * (qp - 16)) >> 1) + offset;
        pre = pre < 1 ? 1 : (pre > 127 ? 127 : pre);
        m[i].s0 = (uint16_t)(pre << 3);
        m[i].s1 = (uint16_t)(pre << 7);
        m[i].shift0 = (uint8_t)((kCtxInit[i].shift_idx >> 2) + 2);
        m[i].shift1 = (uint8_t)((kCtxInit[i].shift_idx & 3) + 3 + m[i].shift0);
        m[i].add0 = (uint16_t)(1023 >> m[i].shift0);
        m[i].add1 = (uint16_t)(16383 >> m[i].shift1);
    }
}

#ifdef WRENC_CABAC_SPEC_ENGINE
// The arithmetic encoder exactly as the reference (and clause 9.3.5 of the specification) words it: one
// output bit at a time with an outstanding-bit counter.  Built only for tests/test_bitstream.py, which
// checks that the production engine below writes the same bytes.
class CabacEncoder {
public:
    explicit CabacEncoder(BitWriter& bw) : bw_(bw) {}

    // bool_coder.rs:1106-1111 + ctu_encoder.rs:38-47 (first CTU of the picture)
    void start(int slice_qp) {
        init_models(m_, slice_qp);
        range_ = 510;
        low_ = 0;
        first_ = true;
        outstanding_ = 0;
    }

    // bool_coder.rs:254-296
    void encode(int ctx, int bin) {
        CtxModel& c = m_[ctx];
        const uint32_t q = range_ >> 5;
        const uint32_t p = (uint32_t)c.s1 + 16u * c.s0;
        const uint32_t mps = p >> 14;
        const uint32_t lps = ((q * ((mps ? 32767u - p : p) >> 9)) >> 1) + 4;
        if ((uint32_t)bin == mps) {
            range_ -= lps;
        } else {
            low_ += range_ - lps;
            range_ = lps;
        }
        renorm();
        c.s0 = (uint16_t)(c.s0 - (c.s0 >> c.shift0) + ((1023 * bin) >> c.shift0));
        c.s1 = (uint16_t)(c.s1 - (c.s1 >> c.shift1) + ((16383 * bin) >> c.shift1));
    }

    // bool_coder.rs:202-216
    void bypass(int bin) {
        low_ <<= 1;
        if (bin) low_ += range_;
        if (low_ >= 1024) {
            put(true);
            low_ -= 1024;
        } else if (low_ < 512) {
            put(false);
        } else {
            low_ -= 512;
            ++outstanding_;
        }
    }
    void bypass_bits(uint32_t v, int n) {
        for (int i = n - 1; i >= 0; --i) bypass((v >> i) & 1);
    }

    // end_of_slice_one_bit = 1: bool_coder.rs:218-235 (terminate, flush; the last written bit doubles as
    // rbsp_stop_one_bit)
    void finish() {
        range_ -= 2;
        low_ += range_;
        range_ = 2;
        renorm();
        put((low_ >> 9) & 1);
        const uint32_t two = ((low_ >> 7) & 3) | 1;
        trailing((two >> 1) & 1);
        trailing(two & 1);
        first_ = true;
        outstanding_ = 0;
    }

private:
    // bool_coder.rs:157-171
    void renorm() {
        while (range_ < 256) {
            if (low_ < 256) {
                put(false);
            } else if (low_ >= 512) {
                low_ -= 512;
                put(true);
            } else {
                low_ -= 256;
                ++outstanding_;
            }
            range_ <<= 1;
            low_ <<= 1;
        }
    }
    // bool_coder.rs:174-190
    void put(bool b) {
        if (!first_) bw_.bit(b);
        first_ = false;
        for (; outstanding_; --outstanding_) bw_.bit(!b);
    }
    // bool_coder.rs:193-199
    void trailing(bool b) {
        bw_.bit(b);
        for (; outstanding_; --outstanding_) bw_.bit(!b);
    }

    BitWriter& bw_;
    CtxModel m_[CTX_COUNT];
    uint32_t range_ = 510, low_ = 0;
    bool first_ = true;
    uint32_t outstanding_ = 0;
};
#else
// Production engine: the same interval arithmetic (bool_coder.rs:136-296) with the low end of the interval
// kept in a 32-bit register and whole bytes leaving it, the carry resolved through a count of buffered
// 0xff bytes instead of one outstanding bit at a time.  It writes the same bit sequence as the
// bit-serial form above (tests/test_bitstream.py builds both and compares).  bits_left_ starts at 23, not
// 24: that is the reference's suppressed first output bit (cabac_first_bit_flag, bool_coder.rs:174-180).
class CabacEncoder {
public:
    explicit CabacEncoder(BitWriter& bw) : bw_(bw) {}

    void start(int slice_qp) {
        init_models(m_, slice_qp);
        range_ = 510;
        low_ = 0;
        bits_left_ = 23;
        buffered_byte_ = 0xff;
        num_buffered_ = 0;
    }

    __attribute__((always_inline)) void encode(int ctx, int bin) {
        CtxModel& c = m_[ctx];
        const uint32_t q = range_ >> 5;
        const uint32_t p = (uint32_t)c.s1 + 16u * c.s0;
        const uint32_t mps = p >> 14;
        const uint32_t lps = ((q * ((mps ? 32767u - p : p) >> 9)) >> 1) + 4;
        // Both outcomes are computed and selected without a branch (the bin value is the least predictable
        // thing in the encoder).  LPS: the interval moves up by the MPS part and shrinks to lps, which
        // needs clz(lps) - 23 shifts to come back to 256..511 (lps is 4..236).  MPS: the interval shrinks by
        // lps and never needs more than one shift.
        const uint32_t r_mps = range_ - lps;
        const bool is_lps = (uint32_t)bin != mps;
        const int nb = is_lps ? __builtin_clz(lps) - 23 : (int)(r_mps < 256);
        low_ = (low_ + (is_lps ? r_mps : 0u)) << nb;
        range_ = (is_lps ? lps : r_mps) << nb;
        bits_left_ -= nb;
        if (__builtin_expect(bits_left_ < 12, 0)) write_out();
        c.s0 = (uint16_t)(c.s0 - (c.s0 >> c.shift0) + (bin ? c.add0 : 0));
        c.s1 = (uint16_t)(c.s1 - (c.s1 >> c.shift1) + (bin ? c.add1 : 0));
    }

    void bypass(int bin) {
        low_ <<= 1;
        if (bin) low_ += range_;
        if (--bits_left_ < 12) write_out();
    }
    void bypass_bits(uint32_t v, int n) {
        while (n > 0) { // at most 8 at a time keeps low_ inside its 32 bits
            const int k = n > 8 ? 8 : n;
            n -= k;
            low_ = (low_ << k) + range_ * ((v >> n) & ((1u << k) - 1));
            bits_left_ -= k;
            if (bits_left_ < 12) write_out();
        }
    }

    // end_of_slice_one_bit = 1, flush, and the bit equal to one that ends the slice data
    void finish() {
        range_ -= 2;
        low_ = (low_ + range_) << 7;
        range_ = 2 << 7;
        bits_left_ -= 7;
        if (bits_left_ < 12) write_out();
        if (low_ >> (32 - bits_left_)) { // a carry into the bytes still held back
            bw_.byte((uint8_t)(buffered_byte_ + 1));
            for (; num_buffered_ > 1; --num_buffered_) bw_.byte(0x00);
            low_ -= 1u << (32 - bits_left_);
        } else {
            if (num_buffered_ > 0) bw_.byte((uint8_t)buffered_byte_);
            for (; num_buffered_ > 1; --num_buffered_) bw_.byte(0xff);
        }
        bw_.put(low_ >> 8, 24 - bits_left_);
        bw_.bit(true);
    }

private:
    __attribute__((noinline)) void write_out() {
        const uint32_t lead = low_ >> (24 - bits_left_);
        bits_left_ += 8;
        low_ &= 0xffffffffu >> bits_left_;
        if (lead == 0xff) {
            ++num_buffered_;
        } else if (num_buffered_ > 0) {
            const uint32_t carry = lead >> 8;
            bw_.byte((uint8_t)(buffered_byte_ + carry));
            buffered_byte_ = lead & 0xff;
            const uint8_t fill = (uint8_t)(0xff + carry);
            for (; num_buffered_ > 1; --num_buffered_) bw_.byte(fill);
        } else {
            num_buffered_ = 1;
            buffered_byte_ = lead;
        }
    }

    BitWriter& bw_;
    CtxModel m_[CTX_COUNT];
    uint32_t range_ = 510, low_ = 0;
    int bits_left_ = 23;
    uint32_t buffered_byte_ = 0xff;
    int num_buffered_ = 0;
};
#endif

} // namespace wrenc_host
