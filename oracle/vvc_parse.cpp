// vvc_parse.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Decoder-side parser for the byte streams the host writer (wrenc_amd/csrc/host) produces: NAL
// splitting, removal of emulation prevention, the parameter sets the reference writes (main.rs:223-260),
// picture header, slice header, and CABAC *decoding* of the CTU data (VVC 9.3.4.3 arithmetic decoding,
// 7.3.11 syntax for the all-intra tool subset of SURVEY.md appendix A) back into the record of
// wrenc_oracle.h (size map, modes, TransCoeffLevel planes).  Together with wro_reconstruct_from_record
// this is the in-repo stand-in for the reference's only end-to-end test, "decoder output == encoder
// reconstruction" (scripts/intergration_test.sh), which needs VTM and cannot run here (SURVEY.md 8f
// rank 3).  It is written from the decoding side of the syntax (what a decoder infers), not by
// inverting the writer's code: the two must agree for the round trip to hold.
//
// PARITY UNPINNED like the rest of oracle/: it has not been checked against VTM or the reference binary.
#include "vvc_parse.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

enum {
    E_OK = 0,
    E_NAL = -10,      // no start code / truncated NAL
    E_VPS = -11,
    E_SPS = -12,
    E_PPS = -13,
    E_PH = -14,
    E_SH = -15,
    E_TREE = -16,     // coding tree inconsistent with the picture
    E_RESIDUAL = -17, // residual syntax ran out of range
    E_END = -18,      // end_of_slice_one_bit / trailing bits wrong
    E_INDEX = -19     // picture index out of range
};

struct Nal {
    int layer, type, tid;
    std::vector<uint8_t> rbsp;
};

// Annex B byte stream: start code prefixes 00 00 01, then 7.4.2 emulation prevention removal
int split_nals(const uint8_t* s, size_t n, std::vector<Nal>& out) {
    std::vector<size_t> starts;
    for (size_t i = 0; i + 2 < n; ++i)
        if (s[i] == 0 && s[i + 1] == 0 && s[i + 2] == 1) {
            starts.push_back(i + 3);
            i += 2;
        }
    if (starts.empty()) return E_NAL;
    for (size_t k = 0; k < starts.size(); ++k) {
        size_t b = starts[k], e = k + 1 < starts.size() ? starts[k + 1] - 3 : n;
        while (e > b && k + 1 < starts.size() && s[e - 1] == 0) --e; // zero bytes before the next prefix
        if (e < b + 2) return E_NAL;
        Nal nal;
        nal.layer = s[b] & 63;
        nal.type = s[b + 1] >> 3;
        nal.tid = (s[b + 1] & 7) - 1;
        int zeros = 0;
        for (size_t i = b + 2; i < e; ++i) {
            if (zeros >= 2 && s[i] == 3) {
                zeros = 0;
                continue;
            }
            nal.rbsp.push_back(s[i]);
            zeros = s[i] == 0 ? zeros + 1 : 0;
        }
        out.push_back(std::move(nal));
    }
    return E_OK;
}

struct BitReader {
    const std::vector<uint8_t>& b;
    size_t pos = 0; // in bits
    bool over = false;
    explicit BitReader(const std::vector<uint8_t>& v) : b(v) {}
    int bit() {
        if (pos >= b.size() * 8) {
            over = true;
            return 0;
        }
        const int v = (b[pos >> 3] >> (7 - (pos & 7))) & 1;
        ++pos;
        return v;
    }
    uint32_t u(int n) {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | (uint32_t)bit();
        return v;
    }
    uint32_t ue() {
        int z = 0;
        while (!bit() && z < 32 && !over) ++z;
        return (1u << z) - 1 + u(z);
    }
    int32_t se() {
        const uint32_t k = ue();
        return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
    }
    bool aligned() const { return (pos & 7) == 0; }
    // rbsp_trailing_bits: a one, zeros to the byte boundary, and nothing after
    bool trailing() {
        if (!bit()) return false;
        while (!aligned())
            if (bit()) return false;
        return pos == b.size() * 8 && !over;
    }
    bool align_zero() {
        while (!aligned())
            if (bit()) return false;
        return true;
    }
};

#define WANT(cond, err) \
    do {                \
        if (!(cond)) return (err); \
    } while (0)

// profile_tier_level as the reference writes it (ptl_encoder.rs:31-79): all zero, no GCI
bool parse_ptl(BitReader& r) {
    if (r.u(7) != 0 || r.bit() || r.u(8) != 0 || r.bit() || r.bit()) return false;
    if (r.bit()) return false; // gci_present_flag
    if (!r.align_zero()) return false;
    return r.u(8) == 0; // ptl_num_sub_profiles
}

bool parse_dpb(BitReader& r) { return r.ue() == 8 && r.ue() == 4 && r.ue() == 1; }

int parse_vps(const Nal& n) {
    BitReader r(n.rbsp);
    WANT(n.layer == 1 && n.tid == 0, E_VPS);
    WANT(r.u(4) == 8 && r.u(6) == 0 && r.u(3) == 0 && r.u(6) == 9, E_VPS);
    WANT(r.align_zero() && parse_ptl(r), E_VPS);
    WANT(r.ue() == 0 && parse_dpb(r), E_VPS);
    WANT(!r.bit() && !r.bit() && r.trailing(), E_VPS);
    return E_OK;
}

int parse_sps(const Nal& n, int& width, int& height) {
    BitReader r(n.rbsp);
    WANT(n.layer == 9, E_SPS);
    WANT(r.u(4) == 1 && r.u(4) == 8 && r.u(3) == 0 && r.u(2) == 1 && r.u(2) == 0 && r.bit() == 1, E_SPS);
    WANT(parse_ptl(r), E_SPS);
    WANT(!r.bit() && !r.bit(), E_SPS);
    width = (int)r.ue();
    height = (int)r.ue();
    WANT(!r.bit() && !r.bit() && r.ue() == 0 && !r.bit() && !r.bit() && r.u(4) == 0 && !r.bit() && r.u(2) == 0 &&
             r.u(2) == 0,
         E_SPS);
    WANT(parse_dpb(r), E_SPS);
    WANT(r.ue() == 0 && !r.bit() && r.ue() == 0 && r.ue() == 0 && !r.bit() && r.ue() == 0 && r.ue() == 0, E_SPS);
    WANT(r.bit() == 1 && r.ue() == 5 && !r.bit(), E_SPS);                // transform skip, size, bdpcm
    WANT(r.bit() == 1 && r.bit() == 1 && r.bit() == 1 && !r.bit(), E_SPS); // mts x3, lfnst
    WANT(!r.bit() && r.bit() == 1, E_SPS);                               // joint cbcr, same qp table
    WANT(r.se() == -26 && r.ue() == 62, E_SPS);
    for (int j = 0; j < 63; ++j) WANT(r.ue() == 0 && r.ue() == 1, E_SPS); // identity chroma QP mapping
    for (int i = 0; i < 9; ++i) WANT(!r.bit(), E_SPS); // sao alf lmcs wp wbp ltrp ilp idr_rpl rpl1_same
    for (int lx = 0; lx < 2; ++lx) {
        static const uint32_t kDelta[3] = {0, 2, 3};
        WANT(r.ue() == 1 && r.ue() == 3, E_SPS);
        for (int i = 0; i < 3; ++i) WANT(r.ue() == kDelta[i] && r.bit() == (lx == 0), E_SPS);
    }
    for (int i = 0; i < 7; ++i) WANT(!r.bit(), E_SPS); // wraparound tmvp amvr bdof smvd dmvr mmvd
    WANT(r.ue() == 0, E_SPS);
    for (int i = 0; i < 5; ++i) WANT(!r.bit(), E_SPS); // sbt affine bcw ciip gpm
    WANT(r.ue() == 0, E_SPS);
    WANT(!r.bit() && !r.bit() && !r.bit() && r.bit() == 1 && !r.bit() && !r.bit() && !r.bit(), E_SPS);
    WANT(r.ue() == 0 && !r.bit() && !r.bit() && !r.bit(), E_SPS); // min_qp_prime_ts ibc ladf scaling list
    WANT(r.bit() == 1 && !r.bit() && !r.bit(), E_SPS);             // dep quant, sdh, virtual boundaries
    WANT(!r.bit() && !r.bit() && !r.bit() && !r.bit(), E_SPS);     // hrd field_seq vui extension
    WANT(r.trailing(), E_SPS);
    return E_OK;
}

int parse_pps(const Nal& n, int width, int height, int& init_qp) {
    BitReader r(n.rbsp);
    WANT(n.layer == 9, E_PPS);
    WANT(r.u(6) == 1 && r.u(4) == 1 && !r.bit(), E_PPS);
    WANT((int)r.ue() == width && (int)r.ue() == height, E_PPS);
    WANT(!r.bit() && !r.bit() && !r.bit() && r.bit() == 1 && !r.bit() && !r.bit(), E_PPS);
    WANT(r.ue() == 2 && r.ue() == 2, E_PPS);
    WANT(!r.bit() && !r.bit() && !r.bit() && !r.bit(), E_PPS);
    init_qp = 26 + r.se();
    WANT(r.bit() == 1 && !r.bit(), E_PPS);                // cu_qp_delta_enabled, chroma tool offsets
    WANT(r.bit() == 1 && !r.bit() && r.bit() == 1, E_PPS); // deblocking control: present, no override, disabled
    WANT(!r.bit() && !r.bit() && !r.bit() && r.trailing(), E_PPS);
    return E_OK;
}

int parse_ph(const Nal& n, int& poc_lsb) {
    BitReader r(n.rbsp);
    WANT(n.layer == 9 && n.tid == 0, E_PH);
    WANT(r.bit() == 1 && !r.bit() && !r.bit() && !r.bit() && r.ue() == 1, E_PH);
    poc_lsb = (int)r.u(4);
    WANT(r.ue() == 0 && r.trailing(), E_PH);
    return E_OK;
}

// ---------------------------------------------------------------------------------------------
// CABAC decoding engine (9.3.4.3) and context models (9.3.2.2)
// ---------------------------------------------------------------------------------------------
struct CtxInit {
    uint8_t init_value, shift_idx;
};
const CtxInit kInit[] = {
#include "vvc_ctx_init.inc"
};
enum {
    C_SPLIT = 0,
    C_MPM_FLAG = 9,
    C_NOT_PLANAR = 10,
    C_CCLM_FLAG = 12,
    C_CCLM_IDX = 13,
    C_CHROMA_PRED = 14,
    C_MTS = 15,
    C_CBF_Y = 19,
    C_CBF_CB = 23,
    C_CBF_CR = 25,
    C_QP_DELTA = 28,
    C_TS = 30,
    C_LAST_X = 32,
    C_LAST_Y = 55,
    C_SB_CODED = 78,
    C_SIG = 85,
    C_PAR = 148,
    C_GTX = 181,
    C_COUNT = 253
};
static_assert(sizeof(kInit) / sizeof(kInit[0]) == C_COUNT, "context table size");

struct Model {
    uint16_t s0, s1;
    uint8_t sh0, sh1;
};

struct CabacDecoder {
    BitReader& r;
    Model m[C_COUNT];
    uint32_t range = 510, offset = 0;
    explicit CabacDecoder(BitReader& br) : r(br) {}
    void start(int slice_qp) {
        const int qp = slice_qp < 0 ? 0 : (slice_qp > 63 ? 63 : slice_qp);
        for (int i = 0; i < C_COUNT; ++i) {
            const int slope = (kInit[i].init_value >> 3) - 4, off = (kInit[i].init_value & 7) * 18 + 1;
            int pre = ((slope * (qp - 16)) >> 1) + off;
            pre = pre < 1 ? 1 : (pre > 127 ? 127 : pre);
            m[i].s0 = (uint16_t)(pre << 3);
            m[i].s1 = (uint16_t)(pre << 7);
            m[i].sh0 = (uint8_t)((kInit[i].shift_idx >> 2) + 2);
            m[i].sh1 = (uint8_t)((kInit[i].shift_idx & 3) + 3 + m[i].sh0);
        }
        range = 510;
        offset = r.u(9);
    }
    int decode(int ctx) {
        Model& c = m[ctx];
        const uint32_t q = range >> 5, p = (uint32_t)c.s1 + 16u * c.s0;
        const uint32_t mps = p >> 14;
        const uint32_t lps = ((q * ((mps ? 32767u - p : p) >> 9)) >> 1) + 4;
        int bin;
        range -= lps;
        if (offset >= range) {
            bin = (int)(mps ^ 1);
            offset -= range;
            range = lps;
        } else {
            bin = (int)mps;
        }
        while (range < 256) {
            range <<= 1;
            offset = (offset << 1) | (uint32_t)r.bit();
        }
        c.s0 = (uint16_t)(c.s0 - (c.s0 >> c.sh0) + ((1023 * bin) >> c.sh0));
        c.s1 = (uint16_t)(c.s1 - (c.s1 >> c.sh1) + ((16383 * bin) >> c.sh1));
        return bin;
    }
    int bypass() {
        offset = (offset << 1) | (uint32_t)r.bit();
        if (offset >= range) {
            offset -= range;
            return 1;
        }
        return 0;
    }
    uint32_t bypass_bits(int n) {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | (uint32_t)bypass();
        return v;
    }
    int terminate() {
        range -= 2;
        if (offset >= range) return 1;
        while (range < 256) {
            range <<= 1;
            offset = (offset << 1) | (uint32_t)r.bit();
        }
        return 0;
    }
};

struct Scan {
    uint8_t x[64], y[64];
};
void make_scan(int lg, Scan& s) { // 6.5.2
    const int w = 1 << lg;
    int i = 0, x = 0, y = 0;
    bool stop = false;
    while (!stop) {
        while (y >= 0) {
            if (x < w && y < w) {
                s.x[i] = (uint8_t)x;
                s.y[i] = (uint8_t)y;
                ++i;
            }
            --y;
            ++x;
        }
        y = x;
        x = 0;
        if (i >= w * w) stop = true;
    }
}

const int kTrans[4][2] = {{0, 2}, {2, 0}, {1, 3}, {3, 1}}; // QStateTransTable (7.3.11.11 / Table 133)
const int kRice[32] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3};

struct SliceParser {
    int W, H;
    wro_picture_out* out;
    CabacDecoder& cabac;
    Scan scans[4];
    bool qp_delta_coded = false;
    bool mts_dc_only = true, mts_zero_out = true;
    int abs_[32 * 32], p1_[32 * 32];
    uint8_t sbf_[64];

    SliceParser(int w, int h, wro_picture_out* o, CabacDecoder& c) : W(w), H(h), out(o), cabac(c) {
        for (int l = 0; l < 4; ++l) make_scan(l, scans[l]);
    }

    uint8_t& size_at(int x, int y) { return out->cu_log2_size[(size_t)(y >> 2) * (W >> 2) + (x >> 2)]; }
    uint8_t& lmode_at(int x, int y) { return out->luma_mode[(size_t)(y >> 2) * (W >> 2) + (x >> 2)]; }

    // 7.3.11.4 coding_tree with only the quadtree split allowed; 6.4.1: the split is allowed while the
    // block is larger than MinQtSize (4); an 8x8 that splits opens the local dual tree (modeTypeCondition 1)
    int coding_tree(int x0, int y0, int lg) {
        // 9.3.4.2.2: condL / condA compare the neighbouring coding block's height / width with ours
        const int cond_l = x0 > 0 && (1 << size_at(x0 - 1, y0)) < (1 << lg);
        const int cond_a = y0 > 0 && (1 << size_at(x0, y0 - 1)) < (1 << lg);
        const int split = cabac.decode(C_SPLIT + cond_l + cond_a);
        if (lg == 5) qp_delta_coded = false;
        if (!split) {
            fill_size(x0, y0, lg, lg);
            return coding_unit(x0, y0, lg, 0);
        }
        if (lg > 3) {
            for (int i = 0; i < 4; ++i) {
                const int rc = coding_tree(x0 + ((i & 1) << (lg - 1)), y0 + ((i >> 1) << (lg - 1)), lg - 1);
                if (rc) return rc;
            }
            return E_OK;
        }
        for (int i = 0; i < 4; ++i) {
            const int x = x0 + ((i & 1) << 2), y = y0 + ((i >> 1) << 2);
            fill_size(x, y, 2, 2);
            const int rc = coding_unit(x, y, 2, 1);
            if (rc) return rc;
        }
        return coding_unit(x0, y0, 3, 2);
    }

    void fill_size(int x0, int y0, int lg, int v) {
        for (int y = y0; y < y0 + (1 << lg); y += 4)
            for (int x = x0; x < x0 + (1 << lg); x += 4) size_at(x, y) = (uint8_t)v;
    }

    // 8.4.2 derivation of candModeList
    void mpm(int x0, int y0, int size, int cand[5]) {
        const int a = x0 > 0 ? lmode_at(x0 - 1, y0 + size - 1) : 0;
        const int b = (y0 > 0 && ((y0 - 1) >> 5) == (y0 >> 5)) ? lmode_at(x0 + size - 1, y0 - 1) : 0;
        if (a == b && a > 1) {
            cand[0] = a;
            cand[1] = 2 + ((a + 61) % 64);
            cand[2] = 2 + ((a - 1) % 64);
            cand[3] = 2 + ((a + 60) % 64);
            cand[4] = 2 + (a % 64);
        } else if (a != b && (a > 1 || b > 1)) {
            const int mn = a < b ? a : b, mx = a < b ? b : a;
            if (mn > 1) {
                cand[0] = a;
                cand[1] = b;
                if (mx - mn == 1) {
                    cand[2] = 2 + ((mn + 61) % 64);
                    cand[3] = 2 + ((mx - 1) % 64);
                    cand[4] = 2 + ((mn + 60) % 64);
                } else if (mx - mn >= 62) {
                    cand[2] = 2 + ((mn - 1) % 64);
                    cand[3] = 2 + ((mx + 61) % 64);
                    cand[4] = 2 + (mn % 64);
                } else if (mx - mn == 2) {
                    cand[2] = 2 + ((mn - 1) % 64);
                    cand[3] = 2 + ((mn + 61) % 64);
                    cand[4] = 2 + ((mx - 1) % 64);
                } else {
                    cand[2] = 2 + ((mn + 61) % 64);
                    cand[3] = 2 + ((mn - 1) % 64);
                    cand[4] = 2 + ((mx + 61) % 64);
                }
            } else {
                cand[0] = mx;
                cand[1] = 2 + ((mx + 61) % 64);
                cand[2] = 2 + ((mx - 1) % 64);
                cand[3] = 2 + ((mx + 60) % 64);
                cand[4] = 2 + (mx % 64);
            }
        } else {
            cand[0] = 1;
            cand[1] = 50;
            cand[2] = 18;
            cand[3] = 46;
            cand[4] = 54;
        }
    }

    // 7.3.11.5 coding_unit (intra); tree: 0 single, 1 dual luma, 2 dual chroma
    int coding_unit(int x0, int y0, int lg, int tree) {
        const int size = 1 << lg;
        if (tree != 2) {
            int mode;
            if (cabac.decode(C_MPM_FLAG)) {
                if (!cabac.decode(C_NOT_PLANAR + 1)) {
                    mode = 0;
                } else {
                    int idx = 0;
                    while (idx < 4 && cabac.bypass()) ++idx;
                    int cand[5];
                    mpm(x0, y0, size, cand);
                    mode = cand[idx];
                }
            } else {
                int rem = (int)cabac.bypass_bits(5); // truncated binary, cMax 60
                if (rem >= 3) rem = ((rem << 1) | cabac.bypass()) - 3;
                int cand[5];
                mpm(x0, y0, size, cand);
                for (int i = 0; i < 5; ++i) // ascending
                    for (int j = i + 1; j < 5; ++j)
                        if (cand[j] < cand[i]) {
                            const int t = cand[i];
                            cand[i] = cand[j];
                            cand[j] = t;
                        }
                mode = rem + 1; // 8.4.2: IntraPredModeY = remainder + 1, then + 1 per candidate it reaches
                for (int i = 0; i < 5; ++i) mode += mode >= cand[i];
            }
            if (mode > 66) return E_TREE;
            for (int y = y0; y < y0 + size; y += 4)
                for (int x = x0; x < x0 + size; x += 4) lmode_at(x, y) = (uint8_t)mode;
        }
        if (tree != 1) {
            // 8.4.3: the direct mode is the luma mode at the centre of the block
            const int luma_ref = lmode_at(x0 + size / 2, y0 + size / 2);
            int cm;
            if (cabac.decode(C_CCLM_FLAG)) {
                int idx = cabac.decode(C_CCLM_IDX);
                if (idx) idx += cabac.bypass();
                cm = 81 + idx;
            } else if (!cabac.decode(C_CHROMA_PRED)) {
                cm = luma_ref;
            } else {
                static const int kFixed[4] = {0, 50, 18, 1};
                cm = kFixed[cabac.bypass_bits(2)];
                if (cm == luma_ref) cm = 66;
            }
            for (int y = y0; y < y0 + size; y += 8)
                for (int x = x0; x < x0 + size; x += 8)
                    out->chroma_mode[(size_t)(y >> 3) * (W >> 3) + (x >> 3)] = (uint8_t)cm;
        }
        mts_dc_only = true;
        mts_zero_out = true;
        const int rc = transform_unit(x0, y0, lg, tree);
        if (rc) return rc;
        if (tree != 2 && mts_zero_out && !mts_dc_only) {
            if (cabac.decode(C_MTS)) return E_TREE; // mts_idx is always 0 in these streams
        }
        return E_OK;
    }

    int16_t* plane(int c) { return c == 0 ? out->lev_y : (c == 1 ? out->lev_cb : out->lev_cr); }

    // 7.3.11.10 transform_unit
    int transform_unit(int x0, int y0, int lg, int tree) {
        int cbf_cb = 0, cbf_cr = 0, cbf_y = 0;
        if (tree != 1) {
            cbf_cb = cabac.decode(C_CBF_CB);
            cbf_cr = cabac.decode(C_CBF_CR + cbf_cb);
        }
        if (tree != 2) cbf_y = cabac.decode(C_CBF_Y);
        if ((cbf_y || cbf_cb || cbf_cr) && tree != 2 && !qp_delta_coded) {
            if (cabac.decode(C_QP_DELTA)) return E_TREE; // cu_qp_delta_abs is always 0
            qp_delta_coded = true;
        }
        for (int c = 0; c < 3; ++c) {
            if (!(c == 0 ? cbf_y : (c == 1 ? cbf_cb : cbf_cr))) continue;
            if (cabac.decode(C_TS + (c != 0))) return E_TREE; // transform_skip_flag is always 0
            const int rc = c == 0 ? residual(0, x0, y0, lg) : residual(c, x0 >> 1, y0 >> 1, lg - 1);
            if (rc) return rc;
        }
        return E_OK;
    }

    int read_last_prefix(int base, int c, int lg) {
        static const int kOffsetY[6] = {0, 0, 3, 6, 10, 15};
        const int off = c == 0 ? kOffsetY[lg - 1] : 20;
        int shift = c == 0 ? (lg + 1) >> 2 : (1 << lg) >> 3;
        if (c && shift > 2) shift = 2;
        const int c_max = (lg << 1) - 1;
        int v = 0;
        while (v < c_max && cabac.decode(base + (v >> shift) + off)) ++v;
        return v;
    }

    int read_remainder(int k) {
        int pre = 0;
        while (pre < 6 && cabac.bypass()) ++pre;
        if (pre < 6) return (pre << k) + (int)cabac.bypass_bits(k);
        int ext = 0;
        while (ext < 11 && cabac.bypass()) ++ext;
        const int len = ext == 11 ? 15 : ext + k + 1;
        return (6 << k) + (((1 << ext) - 1) << (k + 1)) + (int)cabac.bypass_bits(len);
    }

    int rice(int base, int xc, int yc, int tw) const {
        int s = 0;
        if (xc < tw - 1) {
            s += abs_[yc * 32 + xc + 1];
            if (xc < tw - 2) s += abs_[yc * 32 + xc + 2];
            if (yc < tw - 1) s += abs_[(yc + 1) * 32 + xc + 1];
        }
        if (yc < tw - 1) {
            s += abs_[(yc + 1) * 32 + xc];
            if (yc < tw - 2) s += abs_[(yc + 2) * 32 + xc];
        }
        s -= base * 5;
        return kRice[s < 0 ? 0 : (s > 31 ? 31 : s)];
    }

    void tmpl(int xc, int yc, int tw, int& num, int& sum) const {
        num = 0;
        sum = 0;
        const int dx[5] = {1, 2, 1, 0, 0}, dy[5] = {0, 0, 1, 1, 2};
        for (int k = 0; k < 5; ++k) {
            const int x = xc + dx[k], y = yc + dy[k];
            if (x >= tw || y >= tw) continue;
            sum += p1_[y * 32 + x];
            num += p1_[y * 32 + x] > 0; // sig_coeff_flag of a coded position <=> AbsLevelPass1 > 0
        }
    }

    // 7.3.11.11 residual_coding (no transform skip, dependent quantisation on)
    int residual(int c, int tx, int ty, int lg) {
        const int tw = 1 << lg, st = c ? W >> 1 : W;
        int16_t* lev = plane(c) + (size_t)ty * st + tx;
        for (int y = 0; y < tw; ++y) {
            memset(&abs_[y * 32], 0, sizeof(int) * tw);
            memset(&p1_[y * 32], 0, sizeof(int) * tw);
        }
        memset(sbf_, 0, sizeof(sbf_));
        const int px = read_last_prefix(C_LAST_X, c, lg), py = read_last_prefix(C_LAST_Y, c, lg);
        int last_x = px, last_y = py;
        if (px > 3) last_x = (1 << ((px >> 1) - 1)) * (2 + (px & 1)) + (int)cabac.bypass_bits((px >> 1) - 1);
        if (py > 3) last_y = (1 << ((py >> 1) - 1)) * (2 + (py & 1)) + (int)cabac.bypass_bits((py >> 1) - 1);
        if (last_x >= tw || last_y >= tw) return E_RESIDUAL;
        const Scan& sbs = scans[lg - 2];
        const Scan& cs = scans[2];
        const int sbw = tw >> 2;
        int last_sb = (1 << (2 * (lg - 2))) - 1, last_pos = 16;
        for (;;) {
            if (last_pos == 0) {
                last_pos = 16;
                if (--last_sb < 0) return E_RESIDUAL;
            }
            --last_pos;
            if ((sbs.x[last_sb] << 2) + cs.x[last_pos] == last_x && (sbs.y[last_sb] << 2) + cs.y[last_pos] == last_y)
                break;
        }
        if ((last_sb > 0 || last_pos > 0) && c == 0) mts_dc_only = false;
        int rem_bins = ((1 << (2 * lg)) * 7) >> 2;
        int q_state = 0;
        for (int i = last_sb; i >= 0; --i) {
            const int start_q = q_state;
            const int xs = sbs.x[i], ys = sbs.y[i], x_off = xs << 2, y_off = ys << 2;
            int infer_dc = 0, coded = 1;
            if (i < last_sb && i > 0) {
                int csbf = 0;
                if (xs < sbw - 1) csbf |= sbf_[ys * 8 + xs + 1];
                if (ys < sbw - 1) csbf |= sbf_[(ys + 1) * 8 + xs];
                coded = cabac.decode(C_SB_CODED + (c ? 2 : 0) + csbf);
                infer_dc = 1;
            }
            sbf_[ys * 8 + xs] = (uint8_t)coded;
            if (coded && (xs > 3 || ys > 3) && c == 0) mts_zero_out = false;
            int a[16], gt3[16], sign[16];
            memset(a, 0, sizeof(a));
            memset(gt3, 0, sizeof(gt3));
            const int first0 = i == last_sb ? last_pos : 15;
            int first1 = first0;
            for (int n = first0; n >= 0 && rem_bins >= 4; --n) {
                const int xc = x_off + cs.x[n], yc = y_off + cs.y[n];
                const bool is_last = xc == last_x && yc == last_y;
                const int d = xc + yc;
                int num = 0, sum = 0;
                tmpl(xc, yc, tw, num, sum);
                int sig;
                if (coded && (n > 0 || !infer_dc) && !is_last) {
                    const int s = (sum + 1) >> 1, qs = q_state > 1 ? q_state - 1 : 0;
                    const int inc = c == 0 ? 12 * qs + (s < 3 ? s : 3) + (d < 2 ? 8 : (d < 5 ? 4 : 0))
                                           : 36 + 8 * qs + (s < 3 ? s : 3) + (d < 2 ? 4 : 0);
                    sig = cabac.decode(C_SIG + inc);
                    --rem_bins;
                    if (sig) infer_dc = 0;
                } else {
                    sig = is_last || (n == 0 && infer_dc && coded); // 7.4.12.11 inference
                }
                int gt1 = 0, par = 0, g3 = 0;
                if (sig) {
                    int off = sum - num;
                    off = off > 4 ? 4 : off;
                    const int inc = is_last ? (c == 0 ? 0 : 21)
                                            : (c == 0 ? 1 + off + (d == 0 ? 15 : (d < 3 ? 10 : (d < 10 ? 5 : 0)))
                                                      : 22 + off + (d == 0 ? 5 : 0));
                    gt1 = cabac.decode(C_GTX + inc);
                    --rem_bins;
                    if (gt1) {
                        par = cabac.decode(C_PAR + inc);
                        --rem_bins;
                        g3 = cabac.decode(C_GTX + 32 + inc);
                        --rem_bins;
                    }
                }
                const int pass1 = sig + par + gt1 + 2 * g3;
                p1_[yc * 32 + xc] = pass1;
                a[n] = pass1;
                gt3[n] = g3;
                q_state = kTrans[q_state][pass1 & 1];
                first1 = n - 1;
            }
            for (int n = first0; n > first1; --n) {
                const int xc = x_off + cs.x[n], yc = y_off + cs.y[n];
                if (gt3[n]) a[n] += 2 * read_remainder(rice(4, xc, yc, tw));
                abs_[yc * 32 + xc] = a[n];
            }
            for (int n = first1; n >= 0; --n) {
                const int xc = x_off + cs.x[n], yc = y_off + cs.y[n];
                if (coded) {
                    const int k = rice(0, xc, yc, tw);
                    const int zero_pos = (q_state < 2 ? 1 : 2) << k;
                    const int dec = read_remainder(k);
                    a[n] = dec == zero_pos ? 0 : (dec < zero_pos ? dec + 1 : dec);
                }
                abs_[yc * 32 + xc] = a[n];
                q_state = kTrans[q_state][a[n] & 1];
            }
            for (int n = 15; n >= 0; --n) sign[n] = a[n] > 0 ? cabac.bypass() : 0;
            q_state = start_q;
            for (int n = 15; n >= 0; --n) {
                if (a[n] > 0) {
                    const int v = 2 * a[n] - (q_state > 1 ? 1 : 0);
                    if (v > 32767) return E_RESIDUAL;
                    lev[(size_t)(y_off + cs.y[n]) * st + x_off + cs.x[n]] = (int16_t)(sign[n] ? -v : v);
                }
                q_state = kTrans[q_state][a[n] & 1];
            }
            if (cabac.r.over) return E_RESIDUAL;
        }
        return E_OK;
    }
};

struct Stream {
    int width = 0, height = 0, init_qp = 0;
    std::vector<Nal> nals;
    std::vector<size_t> ph; // index of each picture header NAL (its slice NAL follows)
};

int open_stream(const uint8_t* s, size_t len, Stream& st) {
    int rc = split_nals(s, len, st.nals);
    if (rc) return rc;
    bool vps = false, sps = false, pps = false;
    for (size_t i = 0; i < st.nals.size(); ++i) {
        const Nal& n = st.nals[i];
        if (n.type == 14) {
            if ((rc = parse_vps(n))) return rc;
            vps = true;
        } else if (n.type == 15) {
            if ((rc = parse_sps(n, st.width, st.height))) return rc;
            sps = true;
        } else if (n.type == 16) {
            WANT(sps, E_PPS);
            if ((rc = parse_pps(n, st.width, st.height, st.init_qp))) return rc;
            pps = true;
        } else if (n.type == 19) {
            WANT(vps && sps && pps, E_PH);
            WANT(i + 1 < st.nals.size() && st.nals[i + 1].type == 7, E_SH);
            st.ph.push_back(i);
        } else if (n.type != 7) {
            return E_NAL;
        }
    }
    WANT(vps && sps && pps, E_NAL);
    WANT(st.width >= 32 && st.height >= 32 && st.width % 32 == 0 && st.height % 32 == 0, E_SPS);
    return E_OK;
}

} // namespace

extern "C" {

void wro_parse_debug_scan(int lg, uint8_t* xy) {
    Scan s;
    make_scan(lg, s);
    for (int i = 0; i < (1 << (2 * lg)); ++i) {
        xy[2 * i] = s.x[i];
        xy[2 * i + 1] = s.y[i];
    }
}

int wro_parse_stream_info(const uint8_t* stream, size_t len, wro_stream_info* info) {
    Stream st;
    const int rc = open_stream(stream, len, st);
    if (rc) return rc;
    info->width = st.width;
    info->height = st.height;
    info->init_qp = st.init_qp;
    info->n_pictures = (int)st.ph.size();
    return 0;
}

int wro_parse_picture(const uint8_t* stream, size_t len, int index, int* poc_lsb, int* slice_qp,
                      wro_picture_out* rec) {
    Stream st;
    int rc = open_stream(stream, len, st);
    if (rc) return rc;
    if (index < 0 || index >= (int)st.ph.size()) return E_INDEX;
    int poc = 0;
    if ((rc = parse_ph(st.nals[st.ph[(size_t)index]], poc))) return rc;
    const Nal& sl = st.nals[st.ph[(size_t)index] + 1];
    WANT(sl.layer == 9 && sl.tid == 0, E_SH);
    BitReader r(sl.rbsp);
    // slice_header (7.3.7) as slice_encoder.rs:32-341 writes it for an IDR picture
    WANT(!r.bit() && !r.bit(), E_SH);
    const int qp = st.init_qp + r.se();
    WANT(r.bit() == 1, E_SH);                   // sh_dep_quant_used_flag
    WANT(r.bit() == 1 && r.align_zero(), E_SH); // byte_alignment()
    WANT(qp >= 0 && qp <= 63, E_SH);
    if (poc_lsb) *poc_lsb = poc;
    if (slice_qp) *slice_qp = qp;
    if (!rec) return 0;
    const int W = st.width, H = st.height;
    memset(rec->lev_y, 0, sizeof(int16_t) * (size_t)W * H);
    memset(rec->lev_cb, 0, sizeof(int16_t) * (size_t)(W / 2) * (H / 2));
    memset(rec->lev_cr, 0, sizeof(int16_t) * (size_t)(W / 2) * (H / 2));
    memset(rec->cu_log2_size, 0, (size_t)(W / 4) * (H / 4));
    memset(rec->luma_mode, 0, (size_t)(W / 4) * (H / 4));
    memset(rec->chroma_mode, 0, (size_t)(W / 8) * (H / 8));
    CabacDecoder cabac(r);
    cabac.start(qp);
    SliceParser sp(W, H, rec, cabac);
    for (int y = 0; y < H; y += 32)
        for (int x = 0; x < W; x += 32)
            if ((rc = sp.coding_tree(x, y, 5))) return rc;
    WANT(cabac.terminate() == 1, E_END); // end_of_slice_one_bit
    // 9.3.4.3.5: the last bit the engine has read is rbsp_stop_one_bit; only alignment zeros may follow
    WANT(!r.over && r.pos >= 1, E_END);
    WANT(((sl.rbsp[(r.pos - 1) >> 3] >> (7 - ((r.pos - 1) & 7))) & 1) == 1, E_END);
    WANT(r.align_zero() && r.pos == sl.rbsp.size() * 8, E_END);
    return 0;
}

} // extern "C"
