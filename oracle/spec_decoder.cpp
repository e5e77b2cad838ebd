// spec_decoder.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A second, independent pixel-domain decoder for the tool subset wrenc's streams use, written from the
// decoding-process clauses of Rec. ITU-T H.266 (V3, 09/2023) and from nothing else:
//   6.4.4      derivation process for neighbouring block availability (IsAvailable[cIdx][x][y])
//   8.4.5.2.8  reference sample availability marking        8.4.5.2.9  substitution
//   8.4.5.2.10 reference sample filtering                   8.4.5.2.11 INTRA_PLANAR
//   8.4.5.2.12 INTRA_DC                                     8.4.5.2.13 INTRA_ANGULAR2..66
//   8.4.5.2.14 INTRA_LT_CCLM / INTRA_L_CCLM / INTRA_T_CCLM  8.4.5.2.15 position-dependent sample filtering
//   8.7.3      scaling process for transform coefficients (sh_dep_quant_used_flag = 1)
//   8.7.4      transformation process (DCT-2, 8.7.4.5)      8.7.5      picture reconstruction
//
// It exists because wrenc's only end-to-end test is "VTM's decode of the stream == the encoder's
// reconstruction" (scripts/intergration_test.sh:1-15 of the reference) and no VTM exists in this image.
// It shares NO function, table or header with wrenc_oracle.cpp (which restates the reference's
// predictor / dequantiser / inverse transform): it is its own translation unit in its own shared library
// (libwrenc_specdec.so) and includes nothing from this directory.  A PDPC weight, a CCLM tap or a scaling
// shift misread from the reference by the oracle (and by the GPU code that is tested against it) shows up
// as "spec decoder != reconstruction".  Structure follows the clauses, not the reference: p[x][y]
// reference arrays indexed from -1, availability from a map of already reconstructed samples (the
// reference uses geometric rules plus tree walks), transforms as the even/odd families of the standard's
// transMatrix.
//
// Input: the record the stream parser (vvc_parse.cpp) CABAC-decodes -- coding-tree sizes, luma modes,
// chroma prediction modes (8.4.2 / 8.4.3 are done by the parser) and TransCoeffLevel planes -- plus the
// slice QP.  The parser checks that the SPS carries the identity chroma QP table (7.4.3.4), so QpC = QpY,
// that sps_chroma_vertical_collocated_flag = 0 and that BitDepth = 8.
//
// Only tests/ and __graft_entry__.smoke() may load this library.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kBitDepth = 8;
constexpr int kCtbLog2 = 5;
constexpr int INTRA_PLANAR = 0, INTRA_DC = 1, INTRA_LT_CCLM = 81, INTRA_L_CCLM = 82, INTRA_T_CCLM = 83;

inline int Clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
inline int Clip1(int v) { return Clip3(0, (1 << kBitDepth) - 1, v); }
inline int FloorLog2(int v) {
    int r = 0;
    while (v > 1) {
        v >>= 1;
        ++r;
    }
    return r;
}
inline int Sign(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// Table 24 -- intraPredAngle for predModeIntra -14 .. 80
const int kIntraPredAngle[95] = {
    512, 341, 256, 171, 128, 102, 86, 73, 64, 57, 51, 45, 39, 35,                               // -14 .. -1
    0, 0,                                                                                        // 0, 1 (unused)
    32, 29, 26, 23, 20, 18, 16, 14, 12, 10, 8, 6, 4, 3, 2, 1, 0,                                 // 2 .. 18
    -1, -2, -3, -4, -6, -8, -10, -12, -14, -16, -18, -20, -23, -26, -29, -32,                    // 19 .. 34
    -29, -26, -23, -20, -18, -16, -14, -12, -10, -8, -6, -4, -3, -2, -1, 0,                      // 35 .. 50
    1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32,                                    // 51 .. 66
    35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512};                               // 67 .. 80
inline int IntraPredAngle(int mode) { return kIntraPredAngle[mode + 14]; }

// Table 25 -- fC interpolation filter coefficients (fG is generated: Table 25's second half is
// {16 - (p >> 1), 32 - (p >> 1), 16 + (p >> 1), p >> 1})
const int kFc[32][4] = {
    {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2},
    {-4, 56, 14, -2}, {-4, 55, 15, -2}, {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3},
    {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4}, {-4, 36, 36, -4}, {-4, 33, 39, -4},
    {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},
    {0, 4, 62, -2},   {0, 2, 63, -1}};

// 8.7.4.5: the DCT-2 transMatrix, given as its even/odd coefficient families (the letters of the
// standard's listing: a; b c; d..g; h..o; p..E; F..k'): family of basis index k = k's lowest set bit.
const int kDc = 64;
const int kFam4[2] = {83, 36};
const int kFam8[4] = {89, 75, 50, 18};
const int kFam16[8] = {90, 87, 80, 70, 57, 43, 25, 9};
const int kFam32[16] = {90, 90, 88, 85, 82, 78, 73, 67, 61, 54, 46, 38, 31, 22, 13, 4};
const int kFam64[32] = {91, 90, 90, 90, 88, 87, 86, 84, 83, 81, 79, 77, 73, 71, 69, 65,
                        62, 59, 56, 52, 48, 44, 41, 37, 33, 28, 24, 20, 15, 11, 7,  2};

int g_trans[64][64]; // transMatrix[m][n]: basis m, sample n
bool g_trans_ready = false;

// magnitude of cos(j * pi / 128) * 64 * sqrt(2) as the standard tabulates it, 0 < j < 64 (j = 32: 64)
int cos_mag(int j) {
    if (j & 1) return kFam64[j >> 1];
    if (j & 2) return kFam32[j >> 2];
    if (j & 4) return kFam16[j >> 3];
    if (j & 8) return kFam8[j >> 4];
    if (j & 16) return kFam4[j >> 5];
    return kDc; // j == 32
}

void build_trans_matrix() {
    if (g_trans_ready) return;
    for (int m = 0; m < 64; ++m)
        for (int n = 0; n < 64; ++n) {
            if (m == 0) {
                g_trans[m][n] = kDc;
                continue;
            }
            int j = ((2 * n + 1) * m) & 255; // angle j * pi / 128, period 256
            int sign = 1;
            if (j > 128) j = 256 - j;        // cos(2 pi - t) = cos t
            if (j > 64) {                    // cos(pi - t) = -cos t
                j = 128 - j;
                sign = -1;
            }
            g_trans[m][n] = j == 64 ? 0 : sign * cos_mag(j);
        }
    g_trans_ready = true;
}

struct Decoder {
    int W = 0, H = 0, qp = 0;
    std::vector<uint8_t> rec[3];
    std::vector<uint8_t> decoded[3]; // IsAvailable[cIdx][xY][yY], one flag per 4x4 luma unit
    const uint8_t* cu_log2 = nullptr;
    const uint8_t* luma_mode = nullptr;
    const uint8_t* chroma_mode = nullptr;
    const int16_t* lev[3] = {nullptr, nullptr, nullptr};

    int cw(int cIdx) const { return cIdx ? W / 2 : W; }
    int ch(int cIdx) const { return cIdx ? H / 2 : H; }
    int sample(int cIdx, int x, int y) const { return rec[cIdx][(size_t)y * cw(cIdx) + x]; }

    // 6.4.4 (checkPredModeY = FALSE; one slice, one tile, no WPP)
    bool available(int cIdx, int xCurr, int yCurr, int xNbY, int yNbY) const {
        if (xNbY < 0 || yNbY < 0 || xNbY >= W || yNbY >= H) return false;
        if ((xNbY >> kCtbLog2) > (xCurr >> kCtbLog2) && (yNbY >> kCtbLog2) >= (yCurr >> kCtbLog2)) return false;
        if ((yNbY >> kCtbLog2) >= (yCurr >> kCtbLog2) + 1) return false;
        return decoded[cIdx][(size_t)(yNbY >> 2) * (W >> 2) + (xNbY >> 2)] != 0;
    }
    void mark_decoded(int cIdx, int xY, int yY, int sizeY) {
        for (int y = yY; y < yY + sizeY; y += 4)
            for (int x = xY; x < xY + sizeY; x += 4) decoded[cIdx][(size_t)(y >> 2) * (W >> 2) + (x >> 2)] = 1;
    }

    // Reference arrays of one transform block: p[x][-1] for x = -1 .. refW - 1 and p[-1][y] for y = -1 .. refH - 1
    struct Refs {
        int refW = 0, refH = 0;
        std::vector<int> top;  // top[x + 1]  = p[x][-1]
        std::vector<int> left; // left[y + 1] = p[-1][y]
        int& T(int x) { return top[(size_t)(x + 1)]; }
        int& L(int y) { return left[(size_t)(y + 1)]; }
        int T(int x) const { return top[(size_t)(x + 1)]; }
        int L(int y) const { return left[(size_t)(y + 1)]; }
    };

    // 8.4.5.2.8 + 8.4.5.2.9 (refIdx = 0)
    Refs reference_samples(int cIdx, int xTbCmp, int yTbCmp, int nTbW, int nTbH) const {
        Refs r;
        r.refW = 2 * nTbW;
        r.refH = 2 * nTbH;
        r.top.assign((size_t)r.refW + 1, -1);
        r.left.assign((size_t)r.refH + 1, -1);
        const int sub = cIdx ? 2 : 1;
        const int xTbY = xTbCmp * sub, yTbY = yTbCmp * sub;
        // marking: unavailable samples stay at -1
        bool any = false;
        for (int y = -1; y <= r.refH - 1; ++y) {
            const int xNbCmp = xTbCmp - 1, yNbCmp = yTbCmp + y;
            if (available(cIdx, xTbY, yTbY, xNbCmp * sub, yNbCmp * sub)) {
                r.L(y) = sample(cIdx, xNbCmp, yNbCmp);
                any = true;
            }
        }
        for (int x = 0; x <= r.refW - 1; ++x) {
            const int xNbCmp = xTbCmp + x, yNbCmp = yTbCmp - 1;
            if (available(cIdx, xTbY, yTbY, xNbCmp * sub, yNbCmp * sub)) {
                r.T(x) = sample(cIdx, xNbCmp, yNbCmp);
                any = true;
            }
        }
        r.T(-1) = r.L(-1); // the corner p[-1][-1] is one sample with two names here
        // substitution
        if (!any) {
            std::fill(r.top.begin(), r.top.end(), 1 << (kBitDepth - 1));
            std::fill(r.left.begin(), r.left.end(), 1 << (kBitDepth - 1));
            return r;
        }
        if (r.L(r.refH - 1) < 0) {
            // search from (x = -1, y = refH - 1) up to y = -1, then x = 0 .. refW - 1 at y = -1
            int found = -1;
            for (int y = r.refH - 2; y >= -1 && found < 0; --y)
                if (r.L(y) >= 0) found = r.L(y);
            for (int x = 0; x <= r.refW - 1 && found < 0; ++x)
                if (r.T(x) >= 0) found = r.T(x);
            r.L(r.refH - 1) = found;
        }
        for (int y = r.refH - 2; y >= -1; --y)
            if (r.L(y) < 0) r.L(y) = r.L(y + 1);
        r.T(-1) = r.L(-1);
        for (int x = 0; x <= r.refW - 1; ++x)
            if (r.T(x) < 0) r.T(x) = r.T(x - 1);
        return r;
    }

    // 8.4.5.2.10
    static Refs filter_refs(const Refs& u) {
        Refs p = u;
        p.L(-1) = (u.L(0) + 2 * u.L(-1) + u.T(0) + 2) >> 2;
        p.T(-1) = p.L(-1);
        for (int y = 0; y <= u.refH - 2; ++y) p.L(y) = (u.L(y + 1) + 2 * u.L(y) + u.L(y - 1) + 2) >> 2;
        p.L(u.refH - 1) = u.L(u.refH - 1);
        for (int x = 0; x <= u.refW - 2; ++x) p.T(x) = (u.T(x - 1) + 2 * u.T(x) + u.T(x + 1) + 2) >> 2;
        p.T(u.refW - 1) = u.T(u.refW - 1);
        return p;
    }

    // 8.4.5.2.6 for predModeIntra 0 .. 66 (no ISP, no MRL, no MIP, no BDPCM, square blocks: no wide-angle mapping)
    void predict_regular(int cIdx, int xTbCmp, int yTbCmp, int nTbW, int nTbH, int predModeIntra,
                         std::vector<int>& pred) const {
        const Refs unfilt = reference_samples(cIdx, xTbCmp, yTbCmp, nTbW, nTbH);
        const bool refFilterFlag = predModeIntra == INTRA_PLANAR || predModeIntra == 2 || predModeIntra == 34 ||
                                   predModeIntra == 66; // (-14, -12, -10, -6, 72, 76, 78, 80 need non-square blocks)
        const bool filterFlag = nTbW * nTbH > 32 && cIdx == 0 && refFilterFlag;
        const Refs p = filterFlag ? filter_refs(unfilt) : unfilt;
        const int log2W = FloorLog2(nTbW), log2H = FloorLog2(nTbH);
        pred.assign((size_t)nTbW * nTbH, 0);
        auto P = [&](int x, int y) -> int& { return pred[(size_t)y * nTbW + x]; };
        int invAngle = 0;
        if (predModeIntra == INTRA_PLANAR) { // 8.4.5.2.11
            for (int y = 0; y < nTbH; ++y)
                for (int x = 0; x < nTbW; ++x) {
                    const int predV = ((nTbH - 1 - y) * p.T(x) + (y + 1) * p.L(nTbH)) << log2W;
                    const int predH = ((nTbW - 1 - x) * p.L(y) + (x + 1) * p.T(nTbW)) << log2H;
                    P(x, y) = (predV + predH + nTbW * nTbH) >> (log2W + log2H + 1);
                }
        } else if (predModeIntra == INTRA_DC) { // 8.4.5.2.12 (nTbW == nTbH)
            int sum = 0;
            for (int x = 0; x < nTbW; ++x) sum += p.T(x);
            for (int y = 0; y < nTbH; ++y) sum += p.L(y);
            const int dcVal = (sum + nTbW) >> (log2W + 1);
            for (int& v : pred) v = dcVal;
        } else { // 8.4.5.2.13
            const int nTbS = (log2W + log2H) >> 1;
            int filterFlagAng = 0;
            if (!refFilterFlag) {
                static const int intraHorVerDistThres[7] = {0, 0, 24, 14, 2, 0, 0};
                const int minDistVerHor = std::min(std::abs(predModeIntra - 50), std::abs(predModeIntra - 18));
                filterFlagAng = minDistVerHor > intraHorVerDistThres[nTbS] ? 1 : 0;
            }
            const int intraPredAngle = IntraPredAngle(predModeIntra);
            if (intraPredAngle != 0) // Round(512 * 32 / intraPredAngle)
                invAngle = intraPredAngle > 0 ? (512 * 32 + intraPredAngle / 2) / intraPredAngle
                                              : -((512 * 32 + (-intraPredAngle) / 2) / (-intraPredAngle));
            const bool ver = predModeIntra >= 34;
            const int nMain = ver ? nTbW : nTbH, nSide = ver ? nTbH : nTbW, refMain = ver ? p.refW : p.refH;
            // ref[x], x = -nSide .. refMain + 2, stored at x + nSide
            std::vector<int> ref((size_t)nSide + refMain + 4, 0);
            auto R = [&](int x) -> int& { return ref[(size_t)(x + nSide)]; };
            auto main_at = [&](int x) { return ver ? p.T(x) : p.L(x); };   // p[-1 + x'][-1] resp. p[-1][-1 + x']
            auto side_at = [&](int y) { return ver ? p.L(y) : p.T(y); };
            for (int x = 0; x <= nMain + 1; ++x) R(x) = main_at(-1 + x);
            if (intraPredAngle < 0) {
                for (int x = -nSide; x <= -1; ++x) R(x) = side_at(-1 + std::min((x * invAngle + 256) >> 9, nSide));
            } else {
                for (int x = nMain + 2; x <= refMain; ++x) R(x) = main_at(-1 + x);
                for (int x = 1; x <= 2; ++x) R(refMain + x) = main_at(-1 + refMain);
            }
            for (int y = 0; y < nTbH; ++y)
                for (int x = 0; x < nTbW; ++x) {
                    const int along = ver ? y : x, across = ver ? x : y;
                    const int iIdx = ((along + 1) * intraPredAngle) >> 5;
                    const int iFact = ((along + 1) * intraPredAngle) & 31;
                    int v;
                    if (cIdx == 0) {
                        int fT[4];
                        for (int j = 0; j < 4; ++j) {
                            const int fG[4] = {16 - (iFact >> 1), 32 - (iFact >> 1), 16 + (iFact >> 1), iFact >> 1};
                            fT[j] = filterFlagAng ? fG[j] : kFc[iFact][j];
                        }
                        int s = 0;
                        for (int i = 0; i < 4; ++i) s += fT[i] * R(across + iIdx + i);
                        v = Clip1((s + 32) >> 6);
                    } else if (iFact != 0) {
                        v = ((32 - iFact) * R(across + iIdx + 1) + iFact * R(across + iIdx + 2) + 16) >> 5;
                    } else {
                        v = R(across + iIdx + 1);
                    }
                    P(x, y) = v;
                }
        }
        // 8.4.5.2.15 (nTbW, nTbH >= 4; refIdx = 0; BdpcmFlag = 0)
        const bool pdpc = predModeIntra == INTRA_PLANAR || predModeIntra == INTRA_DC || predModeIntra <= 18 ||
                          (predModeIntra >= 50 && predModeIntra < INTRA_LT_CCLM);
        if (!pdpc) return;
        int nScale;
        if (predModeIntra > 50)
            nScale = std::min(2, log2H - FloorLog2(3 * invAngle - 2) + 8);
        else if (predModeIntra < 18 && predModeIntra != INTRA_PLANAR && predModeIntra != INTRA_DC)
            nScale = std::min(2, log2W - FloorLog2(3 * invAngle - 2) + 8);
        else
            nScale = (log2W + log2H - 2) >> 2;
        auto weight = [&](int i) { // 32 >> ((i << 1) >> nScale)
            const int s = (i << 1) >> nScale;
            return s > 5 ? 0 : 32 >> s;
        };
        for (int y = 0; y < nTbH; ++y)
            for (int x = 0; x < nTbW; ++x) {
                int refL = 0, refT = 0, wL = 0, wT = 0;
                if (predModeIntra == INTRA_PLANAR || predModeIntra == INTRA_DC) {
                    refL = p.L(y);
                    refT = p.T(x);
                    wT = weight(y);
                    wL = weight(x);
                } else if (predModeIntra == 18 || predModeIntra == 50) {
                    refL = p.L(y) - p.L(-1) + P(x, y);
                    refT = p.T(x) - p.L(-1) + P(x, y);
                    wT = predModeIntra == 18 ? weight(y) : 0;
                    wL = predModeIntra == 50 ? weight(x) : 0;
                } else if (predModeIntra < 18 && nScale >= 0) {
                    const int dXInt = ((y + 1) * invAngle + 256) >> 9;
                    refT = y < (3 << nScale) ? p.T(x + dXInt) : 0;
                    wT = weight(y);
                } else if (predModeIntra > 50 && nScale >= 0) {
                    const int dYInt = ((x + 1) * invAngle + 256) >> 9;
                    refL = x < (3 << nScale) ? p.L(y + dYInt) : 0;
                    wL = weight(x);
                }
                P(x, y) = Clip1((refL * wL + refT * wT + (64 - wL - wT) * P(x, y) + 32) >> 6);
            }
    }

    // 8.4.5.2.14 (SubWidthC = SubHeightC = 2, sps_chroma_vertical_collocated_flag = 0)
    void predict_cclm(int cIdx, int xTbC, int yTbC, int nTbW, int nTbH, int predModeIntra, std::vector<int>& pred) const {
        const int xTbY = xTbC << 1, yTbY = yTbC << 1;
        pred.assign((size_t)nTbW * nTbH, 0);
        const bool availL = available(cIdx, xTbY, yTbY, xTbY - 1, yTbY);
        const bool availT = available(cIdx, xTbY, yTbY, xTbY, yTbY - 1);
        int numTopRight = 0, numLeftBelow = 0;
        if (predModeIntra == INTRA_T_CCLM) {
            bool availTR = true;
            for (int x = nTbW; x <= 2 * nTbW - 1 && availTR; ++x) {
                availTR = available(cIdx, xTbY, yTbY, xTbY + x * 2, yTbY - 1);
                if (availTR) ++numTopRight;
            }
        }
        if (predModeIntra == INTRA_L_CCLM) {
            bool availLB = true;
            for (int y = nTbH; y <= 2 * nTbH - 1 && availLB; ++y) {
                availLB = available(cIdx, xTbY, yTbY, xTbY - 1, yTbY + y * 2);
                if (availLB) ++numLeftBelow;
            }
        }
        int numSampT, numSampL;
        if (predModeIntra == INTRA_LT_CCLM) {
            numSampT = availT ? nTbW : 0;
            numSampL = availL ? nTbH : 0;
        } else {
            numSampT = (availT && predModeIntra == INTRA_T_CCLM) ? nTbW + std::min(numTopRight, nTbH) : 0;
            numSampL = (availL && predModeIntra == INTRA_L_CCLM) ? nTbH + std::min(numLeftBelow, nTbW) : 0;
        }
        const bool bCTUboundary = (yTbY & ((1 << kCtbLog2) - 1)) == 0;
        int cntT = 0, cntL = 0, pickPosT[4] = {0, 0, 0, 0}, pickPosL[4] = {0, 0, 0, 0};
        {
            const int numIs4N = (availT && availL && predModeIntra == INTRA_LT_CCLM) ? 0 : 1;
            if (availT && (predModeIntra == INTRA_LT_CCLM || predModeIntra == INTRA_T_CCLM)) {
                const int startPos = numSampT >> (2 + numIs4N), pickStep = std::max(1, numSampT >> (1 + numIs4N));
                cntT = std::min(numSampT, (1 + numIs4N) << 1);
                for (int pos = 0; pos < cntT; ++pos) pickPosT[pos] = startPos + pos * pickStep;
            }
            if (availL && (predModeIntra == INTRA_LT_CCLM || predModeIntra == INTRA_L_CCLM)) {
                const int startPos = numSampL >> (2 + numIs4N), pickStep = std::max(1, numSampL >> (1 + numIs4N));
                cntL = std::min(numSampL, (1 + numIs4N) << 1);
                for (int pos = 0; pos < cntL; ++pos) pickPosL[pos] = startPos + pos * pickStep;
            }
        }
        if (numSampL == 0 && numSampT == 0) {
            for (int& v : pred) v = 1 << (kBitDepth - 1);
            return;
        }
        // pY[x][y], x = -3 .. 2 * max(nTbW, numSampT) - 1, y = -3 .. 2 * max(nTbH, numSampL) - 1
        const int xMax = 2 * std::max(nTbW, numSampT), yMax = 2 * std::max(nTbH, numSampL);
        const int stride = xMax + 3;
        std::vector<int> win((size_t)stride * (yMax + 3), 0);
        auto pY = [&](int x, int y) -> int& { return win[(size_t)(y + 3) * stride + (x + 3)]; };
        for (int y = 0; y < 2 * nTbH; ++y) // 1. collocated luma
            for (int x = 0; x < 2 * nTbW; ++x) pY(x, y) = sample(0, xTbY + x, yTbY + y);
        if (availL) // 2. neighbouring luma
            for (int y = availT ? -1 : 0; y <= 2 * std::max(nTbH, numSampL) - 1; ++y)
                for (int x = -3; x <= -1; ++x) pY(x, y) = sample(0, xTbY + x, yTbY + y);
        if (!availT)
            for (int y = -2; y <= -1; ++y)
                for (int x = -2; x <= 2 * nTbW - 1; ++x) pY(x, y) = pY(x, 0);
        if (availT)
            for (int y = -3; y <= -1; ++y)
                for (int x = availL ? -1 : 0; x <= 2 * std::max(nTbW, numSampT) - 1; ++x) pY(x, y) = sample(0, xTbY + x, yTbY + y);
        if (!availL)
            for (int y = -2; y <= 2 * nTbH - 1; ++y) pY(-1, y) = pY(0, y);
        // 3. down-sampled collocated luma (F3 / F4 of the clause for the non-collocated case)
        std::vector<int> pDsY((size_t)nTbW * nTbH);
        for (int y = 0; y < nTbH; ++y)
            for (int x = 0; x < nTbW; ++x)
                pDsY[(size_t)y * nTbW + x] = (pY(2 * x - 1, 2 * y) + pY(2 * x - 1, 2 * y + 1) + 2 * pY(2 * x, 2 * y) +
                                             2 * pY(2 * x, 2 * y + 1) + pY(2 * x + 1, 2 * y) + pY(2 * x + 1, 2 * y + 1) + 4) >> 3;
        // 4. / 5. selected neighbouring samples; chroma neighbours p[x][-1], p[-1][y] are reconstructed samples
        int pSelC[4] = {0, 0, 0, 0}, pSelDsY[4] = {0, 0, 0, 0};
        for (int idx = 0; idx < cntT; ++idx) {
            const int x = pickPosT[idx];
            pSelC[idx] = sample(cIdx, xTbC + x, yTbC - 1);
            if (!bCTUboundary)
                pSelDsY[idx] = (pY(2 * x - 1, -1) + pY(2 * x - 1, -2) + 2 * pY(2 * x, -1) + 2 * pY(2 * x, -2) + pY(2 * x + 1, -1) +
                                pY(2 * x + 1, -2) + 4) >> 3;
            else
                pSelDsY[idx] = (pY(2 * x - 1, -1) + 2 * pY(2 * x, -1) + pY(2 * x + 1, -1) + 2) >> 2;
        }
        for (int idx = cntT; idx < cntT + cntL; ++idx) {
            const int y = pickPosL[idx - cntT];
            pSelC[idx] = sample(cIdx, xTbC - 1, yTbC + y);
            pSelDsY[idx] = (pY(-3, 2 * y) + pY(-3, 2 * y + 1) + 2 * pY(-2, 2 * y) + 2 * pY(-2, 2 * y + 1) + pY(-1, 2 * y) +
                            pY(-1, 2 * y + 1) + 4) >> 3;
        }
        // 6. minY, maxY, minC, maxC
        if (cntT + cntL == 2) {
            for (int* a : {pSelC, pSelDsY}) {
                a[3] = a[0];
                a[2] = a[1];
                a[0] = a[1];
                a[1] = a[3];
            }
        }
        int minGrpIdx[2] = {0, 2}, maxGrpIdx[2] = {1, 3};
        if (pSelDsY[minGrpIdx[0]] > pSelDsY[minGrpIdx[1]]) std::swap(minGrpIdx[0], minGrpIdx[1]);
        if (pSelDsY[maxGrpIdx[0]] > pSelDsY[maxGrpIdx[1]]) std::swap(maxGrpIdx[0], maxGrpIdx[1]);
        if (pSelDsY[minGrpIdx[0]] > pSelDsY[maxGrpIdx[1]]) {
            std::swap(minGrpIdx[0], maxGrpIdx[0]);
            std::swap(minGrpIdx[1], maxGrpIdx[1]);
        }
        if (pSelDsY[minGrpIdx[1]] > pSelDsY[maxGrpIdx[0]]) std::swap(minGrpIdx[1], maxGrpIdx[0]);
        const int maxY = (pSelDsY[maxGrpIdx[0]] + pSelDsY[maxGrpIdx[1]] + 1) >> 1;
        const int maxC = (pSelC[maxGrpIdx[0]] + pSelC[maxGrpIdx[1]] + 1) >> 1;
        const int minY = (pSelDsY[minGrpIdx[0]] + pSelDsY[minGrpIdx[1]] + 1) >> 1;
        const int minC = (pSelC[minGrpIdx[0]] + pSelC[minGrpIdx[1]] + 1) >> 1;
        // 7. a, b, k
        int a, b, k;
        const int diff = maxY - minY;
        if (diff != 0) {
            static const int divSigTable[16] = {0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0};
            const int diffC = maxC - minC;
            int x = FloorLog2(diff);
            const int normDiff = ((diff << 4) >> x) & 15;
            x += normDiff != 0 ? 1 : 0;
            const int y = std::abs(diffC) > 0 ? FloorLog2(std::abs(diffC)) + 1 : 0;
            a = y > 0 ? (diffC * (divSigTable[normDiff] | 8) + (1 << (y - 1))) >> y : 0; // diffC == 0 gives a = 0
            k = (3 + x - y) < 1 ? 1 : 3 + x - y;
            a = (3 + x - y) < 1 ? Sign(a) * 15 : a;
            b = minC - ((a * minY) >> k);
        } else {
            k = 0;
            a = 0;
            b = minC;
        }
        // 8.
        for (size_t i = 0; i < pred.size(); ++i) pred[i] = Clip1(((pDsY[i] * a) >> k) + b);
    }

    // 8.7.3 + 8.7.4 + 8.7.5 for one square transform block whose prediction is `pred`
    void reconstruct_tb(int cIdx, int xTb, int yTb, int nTbS, const std::vector<int>& pred) {
        const int log2 = FloorLog2(nTbS);
        const int stride = cw(cIdx);
        // 8.7.3: sh_dep_quant_used_flag = 1, transform_skip_flag = 0, no scaling list (m = 16), qP = slice QP
        static const int levelScale[6] = {40, 45, 51, 57, 64, 72}; // rectNonTsFlag = 0 for square blocks
        const int bdShift = kBitDepth + ((log2 + log2) / 2) - 5 + 1;
        const int bdOffset = (1 << bdShift) >> 1;
        const long long ls = (long long)(16 * levelScale[(qp + 1) % 6]) << ((qp + 1) / 6);
        const int coeffMin = -(1 << 15), coeffMax = (1 << 15) - 1;
        std::vector<int> d((size_t)nTbS * nTbS);
        bool any = false;
        for (int y = 0; y < nTbS; ++y)
            for (int x = 0; x < nTbS; ++x) {
                const int level = lev[cIdx][(size_t)(yTb + y) * stride + xTb + x];
                any = any || level != 0;
                const long long dnc = ((long long)level * ls + bdOffset) >> bdShift;
                d[(size_t)y * nTbS + x] = (int)std::min<long long>(std::max<long long>(dnc, coeffMin), coeffMax);
            }
        std::vector<int> res((size_t)nTbS * nTbS, 0);
        if (any) { // tu_y/cb/cr_coded_flag = 0 leaves the residual at zero
            // 8.7.4.1: columns first (vertical), clip, rows (horizontal); 8.7.4.5 with trType 0
            const int step = 64 / nTbS; // transMatrix[j * 2^(6 - Log2(nTbS))][i]
            std::vector<long long> e((size_t)nTbS * nTbS);
            std::vector<int> g((size_t)nTbS * nTbS);
            for (int x = 0; x < nTbS; ++x)
                for (int i = 0; i < nTbS; ++i) {
                    long long s = 0;
                    for (int j = 0; j < nTbS; ++j) s += (long long)g_trans[j * step][i] * d[(size_t)j * nTbS + x];
                    e[(size_t)i * nTbS + x] = s;
                }
            for (size_t i = 0; i < e.size(); ++i)
                g[i] = (int)std::min<long long>(std::max<long long>((e[i] + 64) >> 7, coeffMin), coeffMax);
            const int trShift = std::max(20 - kBitDepth, 0);
            for (int y = 0; y < nTbS; ++y)
                for (int i = 0; i < nTbS; ++i) {
                    long long s = 0;
                    for (int j = 0; j < nTbS; ++j) s += (long long)g_trans[j * step][i] * g[(size_t)y * nTbS + j];
                    res[(size_t)y * nTbS + i] = (int)((s + (1LL << (trShift - 1))) >> trShift);
                }
        }
        for (int y = 0; y < nTbS; ++y)
            for (int x = 0; x < nTbS; ++x)
                rec[cIdx][(size_t)(yTb + y) * stride + xTb + x] =
                    (uint8_t)Clip1(pred[(size_t)y * nTbS + x] + res[(size_t)y * nTbS + x]);
    }

    void decode_tb(int cIdx, int xY, int yY, int sizeY, int mode) {
        const int sub = cIdx ? 2 : 1;
        const int n = sizeY / sub;
        std::vector<int> pred;
        if (mode >= INTRA_LT_CCLM)
            predict_cclm(cIdx, xY / sub, yY / sub, n, n, mode, pred);
        else
            predict_regular(cIdx, xY / sub, yY / sub, n, n, mode, pred);
        reconstruct_tb(cIdx, xY / sub, yY / sub, n, pred);
    }

    // 7.3.11.4 coding_tree with quad-tree splits only; an 8x8 split opens the local dual tree (four luma
    // coding units, then one chroma coding unit of the 8x8 area)
    int decode_tree(int x0, int y0, int log2) {
        const int sz = 1 << log2;
        const int here = cu_log2[(size_t)(y0 >> 2) * (W >> 2) + (x0 >> 2)];
        if (here > log2 || here < 2) return -1;
        if (here == log2) { // single-tree coding unit: luma, then Cb, Cr
            if (log2 == 2) return -1;
            const int ml = luma_mode[(size_t)(y0 >> 2) * (W >> 2) + (x0 >> 2)];
            const int mc = chroma_mode[(size_t)(y0 >> 3) * (W >> 3) + (x0 >> 3)];
            if (ml > 66 || !(mc <= 66 || (mc >= INTRA_LT_CCLM && mc <= INTRA_T_CCLM))) return -2;
            decode_tb(0, x0, y0, sz, ml);
            mark_decoded(0, x0, y0, sz);
            decode_tb(1, x0, y0, sz, mc);
            decode_tb(2, x0, y0, sz, mc);
            mark_decoded(1, x0, y0, sz);
            mark_decoded(2, x0, y0, sz);
            return 0;
        }
        if (log2 == 3) { // here == 2: DUAL_TREE_LUMA 4x4 units, then the DUAL_TREE_CHROMA unit
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + (i & 1) * 4, y = y0 + (i >> 1) * 4;
                if (cu_log2[(size_t)(y >> 2) * (W >> 2) + (x >> 2)] != 2) return -1;
                const int ml = luma_mode[(size_t)(y >> 2) * (W >> 2) + (x >> 2)];
                if (ml > 66) return -2;
                decode_tb(0, x, y, 4, ml);
                mark_decoded(0, x, y, 4);
            }
            const int mc = chroma_mode[(size_t)(y0 >> 3) * (W >> 3) + (x0 >> 3)];
            if (!(mc <= 66 || (mc >= INTRA_LT_CCLM && mc <= INTRA_T_CCLM))) return -2;
            decode_tb(1, x0, y0, 8, mc);
            decode_tb(2, x0, y0, 8, mc);
            mark_decoded(1, x0, y0, 8);
            mark_decoded(2, x0, y0, 8);
            return 0;
        }
        for (int i = 0; i < 4; ++i) {
            const int rc = decode_tree(x0 + (i & 1) * (sz >> 1), y0 + (i >> 1) * (sz >> 1), log2 - 1);
            if (rc) return rc;
        }
        return 0;
    }
};

} // namespace

extern "C" {

// Reconstructs a picture from its parsed record.  Returns 0, or a negative code for a record that is not a
// quad-tree of 32/16/8/4 coding units with modes in range.
int wsd_decode_record(int width, int height, int slice_qp, const uint8_t* cu_log2_size, const uint8_t* luma_mode,
                      const uint8_t* chroma_mode, const int16_t* lev_y, const int16_t* lev_cb, const int16_t* lev_cr,
                      uint8_t* out_y, uint8_t* out_cb, uint8_t* out_cr) {
    if (width <= 0 || height <= 0 || (width & 31) || (height & 31) || slice_qp < 0 || slice_qp > 63) return -10;
    build_trans_matrix();
    Decoder d;
    d.W = width;
    d.H = height;
    d.qp = slice_qp;
    d.cu_log2 = cu_log2_size;
    d.luma_mode = luma_mode;
    d.chroma_mode = chroma_mode;
    d.lev[0] = lev_y;
    d.lev[1] = lev_cb;
    d.lev[2] = lev_cr;
    for (int c = 0; c < 3; ++c) {
        d.rec[c].assign((size_t)d.cw(c) * d.ch(c), 0);
        d.decoded[c].assign((size_t)(width >> 2) * (height >> 2), 0);
    }
    for (int y = 0; y < height; y += 32)
        for (int x = 0; x < width; x += 32) {
            const int rc = d.decode_tree(x, y, 5);
            if (rc) return rc;
        }
    memcpy(out_y, d.rec[0].data(), d.rec[0].size());
    memcpy(out_cb, d.rec[1].data(), d.rec[1].size());
    memcpy(out_cr, d.rec[2].data(), d.rec[2].size());
    return 0;
}

// transMatrix rows (tests compare it with the reference's table where the reference has one)
void wsd_trans_matrix(int16_t* m64x64) {
    build_trans_matrix();
    for (int m = 0; m < 64; ++m)
        for (int n = 0; n < 64; ++n) m64x64[m * 64 + n] = (int16_t)g_trans[m][n];
}
}
