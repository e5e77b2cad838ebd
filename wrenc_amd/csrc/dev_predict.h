// dev_predict.h -- reference samples, PLANAR / DC / angular / CCLM prediction, SAD lists (intra_predictor.rs)
// Part of the gfx950 device code of the RD-search path; see wrenc_dev.h for the overall model.
#pragma once

namespace wrenc {

// ---------------------------------------------------------------------------
// Intra prediction.  tx, ty: CTU-local luma position of the TU, tlg: log2 luma
// size, comp: component, mode: TU-array prediction mode.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int pdpc_w(int n_scale, int i) {
    const int sh = (i << 1) >> n_scale;
    return sh > 5 ? 0 : (32 >> sh);
}

// Component convention of every stage below: comp 0 = luma block, comp 1 = the chroma PAIR
// (Cb and Cr blocks of the TU processed together: block index blk = 0/1, plane pc = comp + blk).
//
// Reference samples of one block into the per-plane LDS arrays: unfiltered always, plus the
// [1 2 1]-filtered version for luma blocks of more than 32 samples (intra_predictor.rs:146-353).
// The neighbourhood of a block does not change while its candidate modes are evaluated
// (evaluations only write inside the block), so this runs once per block instead of once per mode.
__device__ __forceinline__ void build_refs(Ctx c, int comp, int tx, int ty, int tlg) {
#ifdef WRENC_EXP_SKIP_REFS
    return;
#endif
    c = uni(c);
    comp = uni(comp);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int n = 1 << (tlg - cs);
    const int tn = 1 << tlg;
    const int cx = tx >> cs, cy = ty >> cs;
    const int st = 1 << cs;
    // segment availabilities in substitution-scan order: BL, L, corner, A, AR (bit j = segment j)
    const int avm = block_avail_mask(c, tx, ty, tlg, st);
    const bool any = avm != 0;
    const int total = 4 * n + 1;
    for (int tt = LANE; tt < nb * total; tt += 64) {
        const int blk = tt >= total ? 1 : 0;
        const int t = tt - blk * total;
        const int pc = comp + blk;
        ref_t* refL = SH.refs + (pc == 0 ? R_L0 : (pc == 1 ? R_LC0 : R_LC1));
        ref_t* refA = SH.refs + (pc == 0 ? R_A0 : (pc == 1 ? R_AC0 : R_AC1));
        // unified item: t <= 2n -> left index li = t (li 0 = corner, li k -> y = k-1); else above
        int seg;
        const bool is_left = t <= 2 * n;
        const int li = t, ai = t - (2 * n + 1);
        if (is_left)
            seg = li == 0 ? 2 : (li <= n ? 1 : 0);
        else
            seg = ai < n ? 3 : 4;
        int v;
        if (!any) {
            v = 128;
        } else {
            // source sample: own position if available, else nearest available in scan order
            int sli = li, sai = ai;
            bool src_left = is_left;
            if (!((avm >> seg) & 1)) {
                const int below = avm & ((1 << seg) - 1);
                int j;
                if (below) { // last sample (in scan order) of the nearest earlier available segment
                    j = 31 - __clz(below);
                    if (j == 0) { src_left = true; sli = n + 1; }
                    else if (j == 1) { src_left = true; sli = 1; }
                    else if (j == 2) { src_left = true; sli = 0; }
                    else { src_left = false; sai = n - 1; }
                } else { // first sample of the first available later segment
                    j = __ffs(avm) - 1;
                    if (j == 1) { src_left = true; sli = n; }
                    else if (j == 2) { src_left = true; sli = 0; }
                    else if (j == 3) { src_left = false; sai = 0; }
                    else { src_left = false; sai = n; }
                }
            }
            v = src_left ? rec_get(pc, cx - 1, cy + sli - 1) : rec_get(pc, cx + sai, cy - 1);
        }
        if (is_left)
            refL[li] = (ref_t)v;
        else
            refA[ai] = (ref_t)v;
    }
    WSYNC();
    // [1 2 1] filter, intra_predictor.rs:304-352 (used by modes 0, 2, 34, 66 only)
    if (comp == 0 && n * n > 32) {
        const ref_t* refL = SH.refs + R_L0;
        const ref_t* refA = SH.refs + R_A0;
        for (int t = LANE; t < total; t += 64) {
            if (t <= 2 * n) {
                const int li = t;
                int v;
                if (li == 2 * n)
                    v = refL[li];
                else if (li == 0)
                    v = (refL[1] + 2 * refL[0] + refA[0] + 2) >> 2;
                else
                    v = (refL[li + 1] + 2 * refL[li] + refL[li - 1] + 2) >> 2;
                SH.refs[R_LF + li] = (ref_t)v;
            } else {
                const int ai = t - (2 * n + 1);
                int v;
                if (ai == 2 * n - 1)
                    v = refA[ai];
                else if (ai == 0)
                    v = (refL[0] + 2 * refA[0] + refA[1] + 2) >> 2;
                else
                    v = (refA[ai - 1] + 2 * refA[ai] + refA[ai + 1] + 2) >> 2;
                SH.refs[R_AF + ai] = (ref_t)v;
            }
        }
        WSYNC();
    }
}

// CCLM model parameters (intra_predictor.rs:1604-2031); uniform across the wave
struct CclmParams {
    int a, k, b;
    bool flat128;
    bool avail_l;
};

__device__ __forceinline__ int cclm_w(Ctx c, int tx, int ty, int y, int x, bool avail_l) {
    // padded luma window p_y_xm3_ym3 (:1766-1818): column -1 repeats column 0 when the left
    // neighbour is unavailable; every other read hits reconstructed luma
    if (x < 0 && !avail_l) x = 0;
    return rec_get(0, tx + x, ty + y);
}
__device__ __forceinline__ int cclm_ds6(Ctx c, int tx, int ty, int sy, int sx, bool avail_l) {
    return (cclm_w(c, tx, ty, sy, sx - 1, avail_l) + cclm_w(c, tx, ty, sy + 1, sx - 1, avail_l) +
            cclm_w(c, tx, ty, sy, sx, avail_l) * 2 + cclm_w(c, tx, ty, sy + 1, sx, avail_l) * 2 +
            cclm_w(c, tx, ty, sy, sx + 1, avail_l) + cclm_w(c, tx, ty, sy + 1, sx + 1, avail_l) + 4) >> 3;
}

// the parameters of one mode and both planes as scalars: sets s0 (Cb) and s0 + 1 (Cr) of a cclm_params result
struct CclmPick {
    int a0, a1, k0, k1, b0, b1;
    bool flat128, avail_l;
    bool have; // (false: nothing at hand, predict() derives the parameters itself.  Passed BY VALUE: a pointer to a pick
               // that is only sometimes there kept the struct in scratch memory, 1.6 KB stored and re-read per request)
};
__device__ __forceinline__ CclmPick cclm_pick(const CclmParams& v, int s0) {
    CclmPick p;
    p.a0 = __builtin_amdgcn_readlane(v.a, 8 * s0);
    p.a1 = __builtin_amdgcn_readlane(v.a, 8 * s0 + 8);
    p.k0 = __builtin_amdgcn_readlane(v.k, 8 * s0);
    p.k1 = __builtin_amdgcn_readlane(v.k, 8 * s0 + 8);
    p.b0 = __builtin_amdgcn_readlane(v.b, 8 * s0);
    p.b1 = __builtin_amdgcn_readlane(v.b, 8 * s0 + 8);
    p.flat128 = __builtin_amdgcn_readlane((int)v.flat128, 8 * s0) != 0;
    p.avail_l = __builtin_amdgcn_readlane((int)v.avail_l, 8 * s0) != 0;
    p.have = true;
    return p;
}
// the sets of the three modes in a call that derives them side by side (cclm_params_all): mode index m (0 LT_CCLM,
// 1 T_CCLM, 2 L_CCLM) = sets 2 m (Cb) and 2 m + 1 (Cr)
__device__ __forceinline__ int cclm_mode_index(int mode) { return mode == LT_CCLM ? 0 : (mode == T_CCLM ? 1 : 2); }

// CCLM model parameters (intra_predictor.rs:1604-2031) of up to eight SETS at once: set s = lanes 8 s .. 8 s + 7 derives
// plane 1 + (s & 1) for the mode its lanes pass (constant inside a set); every lane of a set returns the set's result.
// The (at most four) neighbour positions a model is fitted to are taken one per lane -- lanes 0..3 of a set the
// positions above the block, 4..7 the ones to its left, each with its six (or three) luma taps and its chroma sample --
// and gathered in the reference's order (above first) with ds_bpermute.  (Until round 3 a call derived one mode, every
// lane walking the positions one after the other: some fifty dependent LDS reads, and four calls per leaf -- the three
// modes of the SAD list, then the picked one again for its evaluation: 7 % of the kernel's time at max-split-depth 2.)
__device__ __forceinline__ CclmParams cclm_params(Ctx c, int tx, int ty, int tlg, int mode) {
    c = uni(c);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    CclmParams r;
    const int comp = 1 + ((LANE >> 3) & 1);
    const int tn = 1 << tlg;
    const int tw = tn >> 1, th = tw;
    const int cx = tx >> 1, cy = ty >> 1;
    const int gx = c.ctu_x + tx, gy = c.ctu_y + ty;
    // left / above / above-right / below-left of the block: the segment mask of build_refs says all four (bit 1 = the
    // neighbour left of the first row, 3 = above the first column, 4 = above_right_avail, 0 = below_left_avail:
    // test_avail_tab_kernel checks exactly these identities)
    const int avm = block_avail_mask(c, tx, ty, tlg, 1);
    const bool avail_l = (avm & 2) != 0, avail_t = (avm & 8) != 0;
    const bool ar = (avm & 16) != 0, bl = (avm & 1) != 0;
    r.avail_l = avail_l;
    int num_top_right = 0, num_below_left = 0;
    // (wave-wide: the position is the lane, whichever mode the lane itself derives)
    if (__ballot(mode == T_CCLM) != 0ULL) {
        // run of available above-right samples (:1881-1893): one position per lane, then the length
        // of the leading run of set bits
        const bool a = LANE < tw && nb_avail(c, gx, gy, tn, gx + (tw + LANE) * 2, gy - 1, ar, bl);
        num_top_right = min((int)__ffsll(~__ballot(a)) - 1, tw);
    }
    if (__ballot(mode == L_CCLM) != 0ULL) {
        const bool a = LANE < th && nb_avail(c, gx, gy, tn, gx - 1, gy + (th + LANE) * 2, ar, bl);
        num_below_left = min((int)__ffsll(~__ballot(a)) - 1, th);
    }
    const int num_samp_t = mode == LT_CCLM ? (avail_t ? tw : 0) : ((avail_t && mode == T_CCLM) ? tw + min(num_top_right, th) : 0);
    const int num_samp_l = mode == LT_CCLM ? (avail_l ? th : 0) : ((avail_l && mode == L_CCLM) ? th + min(num_below_left, tw) : 0);
    r.flat128 = (num_samp_l == 0 && num_samp_t == 0); // (such a set runs along with counts of 0 and is overruled at the end)
    const bool b_ctu_boundary = ((c.ctu_y + ty) & 31) == 0;
    const int num_is_4 = !(avail_t && avail_l && mode == LT_CCLM) ? 1 : 0;
    const int cnt_t = (avail_t && (mode == LT_CCLM || mode == T_CCLM)) ? min((1 + num_is_4) << 1, num_samp_t) : 0;
    const int cnt_l = (avail_l && (mode == LT_CCLM || mode == L_CCLM)) ? min((1 + num_is_4) << 1, num_samp_l) : 0;
    // this lane's position: i-th above (:1920-1945) or i-th to the left (:1946-1960); no branches: a lane without a
    // position reads some sample nearby and contributes nothing
    const int j = LANE & 7, i = j & 3;
    const bool top = j < 4;
    const int ns = top ? num_samp_t : num_samp_l;
    const int pos = (ns >> (2 + num_is_4)) + i * max(ns >> (1 + num_is_4), 1);
    const bool act = i < (top ? cnt_t : cnt_l);
    const int ra = top ? -1 : 2 * pos, rb = top ? -2 : 2 * pos + 1, cc = top ? 2 * pos : -2; // two luma rows, centre column
    const int row_a = cclm_w(c, tx, ty, ra, cc - 1, avail_l) + cclm_w(c, tx, ty, ra, cc, avail_l) * 2 + cclm_w(c, tx, ty, ra, cc + 1, avail_l);
    const int row_b = cclm_w(c, tx, ty, rb, cc - 1, avail_l) + cclm_w(c, tx, ty, rb, cc, avail_l) * 2 + cclm_w(c, tx, ty, rb, cc + 1, avail_l);
    const int sy = (top && b_ctu_boundary) ? (row_a + 2) >> 2 : (row_a + row_b + 4) >> 3; // (:1893-1918: one row at a CTU's top)
    const int sc = rec_get(comp, top ? cx + pos : cx - 1, top ? cy - 1 : cy + pos);
    const int mine = act ? (sy | (sc << 16)) : 0;
    // p_sel_ds_y / p_sel_c: slots 0 .. cnt_t - 1 from above, then the ones from the left; unused slots stay 0
    const int sbase = (int)LANE & ~7;
    int ysel[4], csel[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int src = sbase + (q < cnt_t ? q : 4 + q - cnt_t);
        const int v = __builtin_amdgcn_ds_bpermute(src << 2, mine);
        ysel[q] = v & 0xFFFF;
        csel[q] = v >> 16;
    }
    // min group {0,2}, max group {1,3} and the four compare-exchanges of :1973-1986,
    // carried out on (luma, chroma) value pairs instead of indices
    int mnAy = ysel[0], mnAc = csel[0], mnBy = ysel[2], mnBc = csel[2], mxAy = ysel[1], mxAc = csel[1], mxBy = ysel[3], mxBc = csel[3], t;
    if (mnAy > mnBy) { t = mnAy; mnAy = mnBy; mnBy = t; t = mnAc; mnAc = mnBc; mnBc = t; }
    if (mxAy > mxBy) { t = mxAy; mxAy = mxBy; mxBy = t; t = mxAc; mxAc = mxBc; mxBc = t; }
    if (mnAy > mxBy) {
        t = mnAy; mnAy = mxAy; mxAy = t; t = mnAc; mnAc = mxAc; mxAc = t;
        t = mnBy; mnBy = mxBy; mxBy = t; t = mnBc; mnBc = mxBc; mxBc = t;
    }
    if (mnBy > mxAy) { t = mnBy; mnBy = mxAy; mxAy = t; t = mnBc; mnBc = mxAc; mxAc = t; }
    const int max_y = (mxAy + mxBy + 1) >> 1;
    const int max_c = (mxAc + mxBc + 1) >> 1;
    const int min_y = (mnAy + mnBy + 1) >> 1;
    const int min_c = (mnAc + mnBc + 1) >> 1;
    const int diff = max_y - min_y;
    if (diff != 0) {
        const int diff_c = max_c - min_c;
        int x = ilog2i(diff);
        const int norm_diff = ((diff << 4) >> x) & 15;
        x += (norm_diff != 0) ? 1 : 0;
        const int adc = diff_c < 0 ? -diff_c : diff_c;
        const int y = adc > 0 ? ilog2i(adc) + 1 : 0;
        const int div_sig = (int)((0x0111122334455670ULL >> (4 * norm_diff)) & 15); // {0,7,6,5,5,4,4,3,3,2,2,1,1,1,1,0}
        int a = diff_c == 0 ? 0 : (diff_c * (div_sig | 8) + (1 << (y - 1))) >> y;
        int k;
        if (3 + x - y < 1) {
            k = 1;
            a = a < 0 ? -15 : (a > 0 ? 15 : 0);
        } else {
            k = 3 + x - y;
        }
        r.a = a;
        r.k = k;
        r.b = min_c - ((a * min_y) >> k);
    } else {
        r.a = 0;
        r.k = 0;
        r.b = min_c;
    }
    if (r.flat128) {
        r.a = 0;
        r.k = 0;
        r.b = 128;
    }
    return r;
}
// the three modes of a block, both planes, in one pass: sets 2 m + pl (cclm_mode_index)
__device__ __forceinline__ CclmParams cclm_params_all(const Ctx& c, int tx, int ty, int tlg) {
    const int m = (LANE >> 4) & 3;
    return cclm_params(c, tx, ty, tlg, m == 1 ? T_CCLM : (m == 2 ? L_CCLM : LT_CCLM));
}

// Original sample for prediction index i (plane pc, component coordinates x, y).  A full
// evaluation reads the picture; SAD lists read the copy of the block's originals that
// stage_org() put into r2 (free while no transform runs), index obase + i.
// byte offset in r2 of the staged originals (luma at +0, Cb | Cr at +1024): the last 1.5 KB of its 2 KB (the
// tables of a SAD list fill r1, nothing else runs meanwhile)
constexpr int kOrgStage = 512;
// Blocks of at most 16x16 luma samples keep their originals in LDS for the WHOLE leaf search (SAD lists and
// full evaluations alike): 256 B luma + 2 x 64 B chroma in the last 384 bytes of r2, which no stage of a
// block that small touches (forward DCT <= 1088 B, quantize3 ends at byte 1536, dequantise <= 512, the angular
// table of a prediction at 1280..1488; SAD-list tables are in r1).  Staged once per leaf (stage_org_leaf); a
// 32x32 block has no such room: its SAD lists stage per list (stage_org), its full evaluations read the picture.
constexpr int kOrgLeaf = 1664;
// byte offset in r2 of the originals of component `comp` (0 luma, 1 chroma pair) of a block of log2 size tlg
__device__ __forceinline__ int org_byte(int comp, int tlg) {
    return tlg <= 4 ? kOrgLeaf + (comp ? 256 : 0) : kOrgStage + (comp ? 1024 : 0);
}
// original of sample i of the block (plane pc, component coordinates x, y); lds: the block's originals are
// staged at byte obyte of r2
template <bool full>
__device__ __forceinline__ int pred_org(const Ctx& c, int pc, int x, int y, int obyte, int i, bool lds) {
    if (full && !lds) return org_get(c, pc, x, y);
    return ((const uint8_t*)SH.r2)[obyte + i];
}
__device__ __forceinline__ void stage_org(const Ctx& c, int comps, int tx, int ty, int tlg, int lbyte = kOrgStage,
                                          int cword = 256) {
    uint32_t* dst = (uint32_t*)((char*)SH.r2 + lbyte);
#ifdef WRENC_EXP_NO_ORG // timing experiment only
    for (int w = LANE; w < 384; w += 64) dst[w] = 0x50607080u + (uint32_t)w;
    WSYNC();
    return;
#endif
    if (comps & 1) {
        const int words = 1 << (2 * tlg - 2);
        for (int w = LANE; w < words; w += 64) {
            const int row = (4 * w) >> tlg, col = (4 * w) & ((1 << tlg) - 1);
            dst[w] = *(const GLOBAL_AS uint32_t*)&c.org[(ty + row) * 32 + tx + col];
        }
    }
    if (comps & 2) {
        const int lg = tlg - 1;
        const int words = 1 << (2 * lg - 2); // per plane
        for (int w = LANE; w < 2 * words; w += 64) {
            const int pl = w >= words ? 1 : 0;
            const int ww = w - pl * words;
            const int row = (4 * ww) >> lg, col = (4 * ww) & ((1 << lg) - 1);
            dst[cword + w] = *(const GLOBAL_AS uint32_t*)&c.org[1024 + pl * 256 + ((ty >> 1) + row) * 16 + (tx >> 1) + col];
        }
    }
    WSYNC();
}

// originals of a block of at most 16x16 luma samples, once per leaf (see kOrgLeaf)
__device__ __forceinline__ void stage_org_leaf(const Ctx& c, int comps, int tx, int ty, int tlg) {
    stage_org(c, comps, tx, ty, tlg, kOrgLeaf, 64);
}

// one predicted sample: accumulate |org - pred|; `full` also stores residual and prediction
// The prediction itself is parked in the block's own area of the reconstruction tile until the
// residual is added to it (nothing reads that area in between: the reference samples are cached, and
// CCLM reads the luma plane while it writes chroma).  Only the final pass, which compares its
// reconstruction with the search's, keeps the tile and parks the prediction in global scratch.
// A PACK of candidates evaluated side by side (dev_search.h, K_LEAF8) parks each candidate's prediction in LDS
// instead: PRED_PARK, byte kParkByte + i of decw (behind the pack's trellis decisions).
enum { PRED_SCRATCH = 0, PRED_TILE = 1, PRED_PARK = 2, PRED_PARK16 = 3 };
constexpr int kParkByte = 160; // 144 bytes of decisions in front, 288 bytes of predictions (3 x (64 + 16 + 16)) behind
static_assert(kParkByte >= 144 && kParkByte % 16 == 0 && kParkByte + 288 <= 448, "Lds::decw: decisions of an 8x8 pack | its park | the server's results");
// PRED_PARK16: a pack of TWO 16x16 candidates (K_LEAF16) has no one free region for its 768 bytes of predictions, but
// three that add up to exactly that: the last 512 bytes of r1 (two candidates' residuals fill the first 1536; chunk
// entries, levels and the inverse transform stay inside them) for the luma predictions, 128 bytes of r2 between the
// scan-order coefficients (1536) and the leaf's originals (kOrgLeaf) for candidate 0's chroma pair, the last 128 bytes
// of decw behind the pack's 384 bytes of decisions for candidate 1's.  park16(i) = the byte of sample index i of the
// pack's layout (luma [cand][256], then chroma [cand][Cb | Cr][64]; nl = 256 x candidates).
__device__ __forceinline__ uint8_t* park16(int i, int nl) {
    if (i < nl) return (uint8_t*)SH.r1 + 1536 + i;
    const int j = i - nl;
    return j < 128 ? (uint8_t*)SH.r2 + 1536 + j : (uint8_t*)SH.decw + 384 + (j - 128);
}
template <bool full>
__device__ __forceinline__ int emit_sample(const Ctx& c, int o, int i, int v, int pc, int x, int y, int to_tile, int nl = 0) {
    const int d = o - v;
    if (full) { // i already includes the block's base in r1 / the prediction scratch
        SH.r1[i] = (int16_t)d;
        if (to_tile == PRED_TILE)
            rec_put(pc, x, y, v);
        else if (to_tile == PRED_PARK)
            ((uint8_t*)SH.decw)[kParkByte + i] = (uint8_t)v;
        else if (to_tile == PRED_PARK16)
            *park16(i, nl) = (uint8_t)v;
        else
            c.pred_scratch[i] = (uint8_t)v;
    }
    return d < 0 ? -d : d;
}

// Prediction of one luma block (comp 0) or of the Cb+Cr pair (comp 1) from the cached reference
// samples (build_refs must have run for this block; CCLM reads the reconstructed luma instead).
// Sample index i runs over nb*n*n: block blk = i / (n*n), then row-major inside the block.
// full: the residual org - pred goes to r1[i] and the prediction byte to this wave's scratch
//       (each lane later re-reads exactly the bytes it wrote).
// Returns the lane's partial sum of |org - pred| (the SAD of block_splitter.rs:96-104).
template <bool full>
__device__ __forceinline__ int predict(Ctx c, int comp, int tx, int ty, int tlg, int mode, int rbase = 0,
                                       int to_tile = PRED_TILE, int nl = 0, CclmPick pick = CclmPick{}) {
#ifdef WRENC_EXP_SKIP_PRED // instruction-count experiment only (profiles/r04_issue_model.md): nothing predicted
    return 0;
#endif
    c = uni(c);
    rbase = uni(rbase);
    to_tile = uni(to_tile);
    nl = uni(nl);
    comp = uni(comp);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    mode = uni(mode);
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int lg = tlg - cs;
    const int n = 1 << lg;
    const int cx = tx >> cs, cy = ty >> cs;
    const int nn = n * n;
    const int obyte = org_byte(comp, tlg);
    const bool olds = tlg <= 4;
    int sad = 0;
    if (mode >= LT_CCLM) {
        // model parameters of both planes in one pass: odd lanes derive Cr, even lanes Cb
        // (pick: the parameters are at hand from the block's CCLM SAD list, sad_list_cclm)
        CclmPick cp = pick;
        if (!pick.have) cp = cclm_pick(cclm_params(c, tx, ty, tlg, mode), 0);
        const int a0 = cp.a0, a1 = cp.a1, k0 = cp.k0, k1 = cp.k1, b0 = cp.b0, b1 = cp.b1;
        const bool flat128 = cp.flat128, avail_l = cp.avail_l;
        for (int i = LANE; i < nb * nn; i += 64) {
            const int blk = i >> (2 * lg);
            const int ii = i & (nn - 1);
            const int x = ii & (n - 1), y = ii >> lg;
            const int o = pred_org<full>(c, comp + blk, cx + x, cy + y, obyte, i, olds); // issued early
            int v;
            if (flat128) {
                v = 128;
            } else {
                const int ds = cclm_ds6(c, tx, ty, 2 * y, 2 * x, avail_l);
                v = (M24(ds, blk ? a1 : a0) >> (blk ? k1 : k0)) + (blk ? b1 : b0);
                v = min(max(v, 0), 255);
            }
            sad += emit_sample<full>(c, o, rbase + i, v, comp + blk, cx + x, cy + y, to_tile, nl);
        }
        WSYNC();
        return sad;
    }
    // luma blocks of more than 32 samples use the filtered references for modes 0, 2, 34, 66
    const bool filt = comp == 0 && nn > 32 && (mode == 0 || mode == 2 || mode == 34 || mode == 66);
    const int oL0 = comp == 0 ? (filt ? R_LF : R_L0) : R_LC0; // index 0 = corner
    const int oA0 = comp == 0 ? (filt ? R_AF : R_A0) : R_AC0;
    const ref_t* L0 = SH.refs + oL0;
    const ref_t* A0 = SH.refs + oA0;
    if (mode == PLANAR || mode == DC) {
        int dcv0 = 0, dcv1 = 0;
        if (mode == DC) {
            int part0 = 0, part1 = 0;
            for (int t = LANE; t < 2 * n; t += 64) {
                part0 += t < n ? A0[t] : L0[t - n + 1];
                if (nb == 2) part1 += t < n ? SH.refs[R_AC1 + t] : SH.refs[R_LC1 + t - n + 1];
            }
            dcv0 = ((wave_sum_i32(part0) + n) >> (lg + 1)) & 0xFF; // `as u8`
            if (nb == 2) dcv1 = ((wave_sum_i32(part1) + n) >> (lg + 1)) & 0xFF;
        }
        const int n_scale = (2 * lg - 2) >> 2;
        for (int i = LANE; i < nb * nn; i += 64) {
            const int blk = i >> (2 * lg);
            const int ii = i & (nn - 1);
            const int x = ii & (n - 1), y = ii >> lg;
            const int o = pred_org<full>(c, comp + blk, cx + x, cy + y, obyte, i, olds); // issued early
            const ref_t* L = SH.refs + (blk ? R_LC1 : oL0);
            const ref_t* A = SH.refs + (blk ? R_AC1 : oA0);
            int v;
            if (mode == PLANAR) {
                const int pv = M24(n - 1 - y, A[x]) + M24(y + 1, L[n + 1]);
                const int ph = M24(n - 1 - x, L[y + 1]) + M24(x + 1, A[n]);
                v = ((pv + ph + n) >> (lg + 1)) & 0xFF;
            } else {
                v = blk ? dcv1 : dcv0;
            }
            const int wl = pdpc_w(n_scale, x), wt = pdpc_w(n_scale, y);
            v = (int16_t)(M24(L[y + 1], wl) + M24(A[x], wt) + M24(64 - wt - wl, v) + 32) >> 6;
            v = min(max(v, 0), 255);
            sad += emit_sample<full>(c, o, rbase + i, v, comp + blk, cx + x, cy + y, to_tile, nl);
        }
        WSYNC();
        return sad;
    }
    // angular 2..66 (intra_predictor.rs:1287-1602), square blocks
    const int at = c.k->ang_tab[mode];
    const int angle = (int)(int16_t)(at & 0xFFFF);
    const int inv_angle = at >> 16;
    bool filter_flag = false;
    if (!(mode == 2 || mode == 34 || mode == 66)) {
        const int md = min(abs(mode - 50), abs(mode - 18));
        const int thr = lg == 2 ? 24 : (lg == 3 ? 14 : (lg == 4 ? 2 : 0));
        filter_flag = md > thr;
    }
    const bool do_pdpc = mode <= 18 || mode >= 50;
    int n_scale = 0;
    if (mode > 50 || (mode > 1 && mode < 18))
        n_scale = min(lg - ilog2i(3 * inv_angle - 2) + 8, 2);
    else
        n_scale = (2 * lg - 2) >> 2;
    // The main reference of the mode, projected once (intra_predictor.rs:1311-1420): entry idx in
    // [-n, 2n + 3] = ref[idx] of the reference's refx / refy arrays: idx >= 0 reads the main side
    // (0 = corner, k = sample k - 1, clamped to 2n), idx < 0 the side array at the inverse-angle
    // projection.  It lives in r2 (no transform runs during a prediction), so
    // a sample's taps are consecutive LDS reads with no selects.
    const bool vertical = mode >= 34;
    // Bytes, each XOR 0x80 (ref - 128 as a signed byte): the filters run as one v_dot4_i32_i8 over four
    // packed taps, see sad_list_angular.
    constexpr int RM0 = 1280, RMS = 104; // byte offset of the table in r2 (below kOrgLeaf), stride per block
    uint8_t* rm = (uint8_t*)SH.r2 + RM0;
    {
        const int ne = 3 * n + 4;
        for (int e = LANE; e < nb * ne; e += 64) {
            const int blk = e >= ne ? 1 : 0;
            const int ee = e - blk * ne;
            const int idx = ee - n;
            const int oL = blk ? R_LC1 : oL0, oA = blk ? R_AC1 : oA0;
            const int k = idx >= 0 ? min(idx, 2 * n) : max(min((M24(idx, inv_angle) + 256) >> 9, n), 0);
            const bool from_above = (idx >= 0) == vertical;
            // k == 0 is the corner (L[0]); above sample k - 1 = A[k - 1], left sample k - 1 = L[k]
            rm[blk * RMS + ee] = (uint8_t)(SH.refs[k == 0 ? oL : (from_above ? oA + k - 1 : oL + k)] ^ 0x80);
        }
        WSYNC();
    }
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg);
        const int ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int o = pred_org<full>(c, comp + blk, cx + x, cy + y, obyte, i, olds); // issued early
        const ref_t* L = SH.refs + (blk ? R_LC1 : oL0);
        const ref_t* A = SH.refs + (blk ? R_AC1 : oA0);
        int v;
        {
            const int along = vertical ? y : x, across = vertical ? x : y;
            const int i_idx = M24(along + 1, angle) >> 5;
            const int i_fact = M24(along + 1, angle) & 31;
            const int ta = blk * RMS + n + across + i_idx; // taps = ref[across + i_idx + 0..3]
            const uint32_t* tp = (const uint32_t*)(rm + (ta & ~3));
            const int taps = (int)__builtin_amdgcn_alignbyte(tp[1], tp[0], ta & 3);
            if (comp == 0) {
                const int w = filter_flag ? 0x00102010 + (i_fact >> 1) * 0x0100FEFF : *(const int*)&SHT.fc[i_fact][0];
                v = min(max(__builtin_amdgcn_sdot4(w, taps, 8192 + 32, false) >> 6, 0), 255);
            } else {
                // i_fact == 0 gives the second tap itself; a convex combination of 8-bit samples needs no clamp
                v = __builtin_amdgcn_sdot4(((32 - i_fact) << 8) | (i_fact << 16), taps, 4096 + 16, false) >> 5;
            }
        }
        if (do_pdpc) {
            // intra_predictor.rs:355-757; left[] = L+1, above[] = A
            int rl = 0, rt = 0, wl = 0, wt = 0;
            if (mode == 18 || mode == 50) {
                const int alrs = L[0];
                rl = (int16_t)(L[y + 1] - alrs + v);
                rt = (int16_t)(A[x] - alrs + v);
                wl = mode == 50 ? pdpc_w(n_scale, x) : 0;
                wt = mode == 18 ? pdpc_w(n_scale, y) : 0;
            } else if (mode < 18 && n_scale >= 0) {
                const int dx_int = (M24(y + 1, inv_angle) + 256) >> 9;
                rt = y < (3 << n_scale) ? A[x + dx_int] : 0;
                wt = pdpc_w(n_scale, y);
            } else if (mode > 50 && n_scale >= 0) {
                const int dy_int = (M24(x + 1, inv_angle) + 256) >> 9;
                rl = x < (3 << n_scale) ? L[1 + y + dy_int] : 0;
                wl = pdpc_w(n_scale, x);
            }
            v = (int16_t)(M24(rl, wl) + M24(rt, wt) + M24(64 - wt - wl, v) + 32) >> 6;
            v = min(max(v, 0), 255);
        }
        sad += emit_sample<full>(c, o, rbase + i, v, comp + blk, cx + x, cy + y, to_tile, nl);
    }
    WSYNC();
    return sad;
}

// Prediction of up to four CANDIDATES of one 4x4 luma block at once (the packed 4x4 leaf search, dev_search.h
// K_LEAF4): candidate s = lanes 16 s .. 16 s + 15, lane = (s, sample); `mode` is the lane's candidate mode
// (PLANAR, DC or 2..66; kNoMode lanes compute nothing).  Same arithmetic as predict() for comp 0, tlg 2 (a block of
// 16 samples never uses the filtered references); the projected main reference of an angular candidate is built
// by the candidate's own 16 lanes into tab4 (16 bytes per candidate, in r2 below kOrgLeaf).  build_refs(c, 0, ..)
// must have run.  Returns the predicted sample.
constexpr int kTab4Byte = 1280; // byte offset in r2: 4 x 16 bytes + 4 of slack for the last dword pair
// pl < 0: luma; pl = 0 / 1 (may differ per lane): the 4x4 Cb / Cr block of an 8x8 CU, predicted from that plane's
// reference samples with the 2-tap chroma interpolation (build_refs(c, 1, ..) must have run).
__device__ __forceinline__ int predict4_lane(const Ctx& c, int mode, int pl = -1) {
#ifdef WRENC_EXP_SKIP_PRED
    return 128;
#endif
    constexpr int n = 4, lg = 2;
    const int lane = lane_fresh();
    const int s = lane >> 4, i = lane & 15;
    const int x = i & 3, y = i >> 2;
    const ref_t* L = SH.refs + (pl < 0 ? R_L0 : (pl ? R_LC1 : R_LC0)); // index 0 = corner
    const ref_t* A = SH.refs + (pl < 0 ? R_A0 : (pl ? R_AC1 : R_AC0));
    uint8_t* tab = (uint8_t*)SH.r2 + kTab4Byte;
    const bool ang = mode >= 2 && mode <= 66;
    int angle = 0, inv_angle = 0;
    bool vertical = false;
    if (ang) {
        const int at = c.k->ang_tab[mode];
        angle = (int)(int16_t)(at & 0xFFFF);
        inv_angle = at >> 16;
        vertical = mode >= 34;
        // entry ee of the candidate's table = ref[ee - n] (intra_predictor.rs:1311-1420), see predict()
        const int idx = i - n;
        const int k = idx >= 0 ? min(idx, 2 * n) : max(min((M24(idx, inv_angle) + 256) >> 9, n), 0);
        const bool from_above = (idx >= 0) == vertical;
        tab[16 * s + i] = (uint8_t)((k == 0 ? L[0] : (from_above ? A[k - 1] : L[k])) ^ 0x80);
    }
    WSYNC();
    int v = 0;
    if (mode == PLANAR || mode == DC) {
        if (mode == PLANAR) {
            const int pv = M24(n - 1 - y, A[x]) + M24(y + 1, L[n + 1]);
            const int ph = M24(n - 1 - x, L[y + 1]) + M24(x + 1, A[n]);
            v = ((pv + ph + n) >> (lg + 1)) & 0xFF;
        } else {
            v = ((A[0] + A[1] + A[2] + A[3] + L[1] + L[2] + L[3] + L[4] + n) >> (lg + 1)) & 0xFF; // `as u8`
        }
        const int wl = pdpc_w(0, x), wt = pdpc_w(0, y); // n_scale = (2 * lg - 2) >> 2 = 0
        v = (int16_t)(M24(L[y + 1], wl) + M24(A[x], wt) + M24(64 - wt - wl, v) + 32) >> 6;
        v = min(max(v, 0), 255);
    } else if (ang) {
        bool filter_flag = false;
        if (!(mode == 2 || mode == 34 || mode == 66)) filter_flag = min(abs(mode - 50), abs(mode - 18)) > 24; // lg == 2
        int n_scale;
        if (mode > 50 || mode < 18)
            n_scale = min(lg - ilog2i(3 * inv_angle - 2) + 8, 2);
        else
            n_scale = (2 * lg - 2) >> 2;
        const int along = vertical ? y : x, across = vertical ? x : y;
        const int i_idx = M24(along + 1, angle) >> 5;
        const int i_fact = M24(along + 1, angle) & 31;
        const int ta = 16 * s + n + across + i_idx; // taps = ref[across + i_idx + 0..3]
        const uint32_t* tp = (const uint32_t*)(tab + (ta & ~3));
        const int taps = (int)__builtin_amdgcn_alignbyte(tp[1], tp[0], ta & 3);
        if (pl < 0) {
            const int w = filter_flag ? 0x00102010 + (i_fact >> 1) * 0x0100FEFF : *(const int*)&SHT.fc[i_fact][0];
            v = min(max(__builtin_amdgcn_sdot4(w, taps, 8192 + 32, false) >> 6, 0), 255);
        } else { // i_fact == 0 gives the second tap itself; a convex combination of 8-bit samples needs no clamp
            v = __builtin_amdgcn_sdot4(((32 - i_fact) << 8) | (i_fact << 16), taps, 4096 + 16, false) >> 5;
        }
        if (mode <= 18 || mode >= 50) { // PDPC, intra_predictor.rs:355-757; left[] = L+1, above[] = A
            int rl = 0, rt = 0, wl = 0, wt = 0;
            bool on = true;
            if (mode == 18 || mode == 50) {
                const int alrs = L[0];
                rl = (int16_t)(L[y + 1] - alrs + v);
                rt = (int16_t)(A[x] - alrs + v);
                wl = mode == 50 ? pdpc_w(n_scale, x) : 0;
                wt = mode == 18 ? pdpc_w(n_scale, y) : 0;
            } else if (mode < 18 && n_scale >= 0) {
                const int dx_int = (M24(y + 1, inv_angle) + 256) >> 9;
                rt = y < (3 << n_scale) ? A[x + dx_int] : 0;
                wt = pdpc_w(n_scale, y);
            } else if (mode > 50 && n_scale >= 0) {
                const int dy_int = (M24(x + 1, inv_angle) + 256) >> 9;
                rl = x < (3 << n_scale) ? L[1 + y + dy_int] : 0;
                wl = pdpc_w(n_scale, x);
            } else {
                on = false;
            }
            if (on) {
                v = (int16_t)(M24(rl, wl) + M24(rt, wt) + M24(64 - wt - wl, v) + 32) >> 6;
                v = min(max(v, 0), 255);
            }
        }
    }
    return v;
}

// SADs of the three CCLM modes of a chroma pair (get_chroma_intra_pred_aux_cost of LT_CCLM, T_CCLM,
// L_CCLM, block_splitter.rs:476-522, 847-854): the down-sampled luma of a sample is the same for
// the three modes, so it is computed once and the three linear models are applied to it.
// Returns the SAD of mode m (0 LT, 1 T, 2 L) in lane m.
// all (optional): the parameters of the three modes as cclm_params_all left them, for the evaluation of the picked one
__device__ __forceinline__ unsigned sad_list_cclm(const Ctx& c, int tx, int ty, int tlg, CclmParams* all = nullptr) {
    const int lg = tlg - 1;
    const int n = 1 << lg;
    const int nn = n * n;
    const int cx = tx >> 1, cy = ty >> 1;
    // model parameters: odd lanes derive Cr, even lanes Cb (cclm_params), then made scalar
    int a[3][2], k[3][2], b[3][2];
    bool flat[3], avail_l = false;
    const CclmParams cp = cclm_params_all(c, tx, ty, tlg);
    if (all) *all = cp;
#ifdef WRENC_EXP_SKIP_SAD
    return (unsigned)LANE;
#endif
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            a[m][pl] = __builtin_amdgcn_readlane(cp.a, 8 * (2 * m + pl));
            k[m][pl] = __builtin_amdgcn_readlane(cp.k, 8 * (2 * m + pl));
            b[m][pl] = __builtin_amdgcn_readlane(cp.b, 8 * (2 * m + pl));
        }
        flat[m] = __builtin_amdgcn_readlane((int)cp.flat128, 16 * m) != 0;
        avail_l = __builtin_amdgcn_readlane((int)cp.avail_l, 0) != 0; // the same for the three modes
    }
    int s0 = 0, s1 = 0, s2 = 0;
    for (int i = LANE; i < 2 * nn; i += 64) {
        const int blk = i >> (2 * lg);
        const int ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int o = ((const uint8_t*)SH.r2)[org_byte(1, tlg) + i];
        const int ds = cclm_ds6(c, tx, ty, 2 * y, 2 * x, avail_l);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            int v = 128;
            if (!flat[m]) v = min(max((M24(ds, blk ? a[m][1] : a[m][0]) >> (blk ? k[m][1] : k[m][0])) + (blk ? b[m][1] : b[m][0]), 0), 255);
            const int d = o - v;
            const int ad = d < 0 ? -d : d;
            if (m == 0)
                s0 += ad;
            else if (m == 1)
                s1 += ad;
            else
                s2 += ad;
        }
    }
    const unsigned t0 = (unsigned)wave_sum_i32(s0), t1 = (unsigned)wave_sum_i32(s1), t2 = (unsigned)wave_sum_i32(s2);
    WSYNC();
    return LANE == 0 ? t0 : (LANE == 1 ? t1 : (LANE == 2 ? t2 : 0u));
}

constexpr int kNoMode = 255; // list entry that is not evaluated (cost f32::MAX)

// SADs of a LIST of angular modes (2..66) of one block: get_intra_pred_aux_cost of each entry
// (block_splitter.rs:64-108), luma block and/or chroma pair.  Same arithmetic as predict<false>,
// organised so that the per-mode fixed work is done once per list:
//   * lane mi derives the parameters of entry mi (angle, inverse angle, filter / PDPC variant);
//     the uniform loop over the entries fetches them with v_readlane;
//   * the projected main references of ALL entries are built in one pass into r1 .. r2 (free
//     during SAD lists), stride 4n per block;
//   * a lane adds the SAD of entry mi into its accumulator when LANE == mi.
// acc (lane mi): summed SAD of entry mi over the components; entries with mode kNoMode stay 0.
#ifndef WRENC_SAD4X4
#define WRENC_SAD4X4 1 // 0: one sample per lane and iteration (rounds 1 .. 3a; kept for A/B runs)
#endif
#ifdef WRENC_EXP_NOINLINE_SAD // code-size experiment (profiles/r04_issue_model.md): one copy of the list, called
#define WRENC_SAD_INLINE __attribute__((noinline))
#else
#define WRENC_SAD_INLINE __forceinline__
#endif
__device__ WRENC_SAD_INLINE unsigned sad_list_angular(const Ctx& c, int comps, int tx, int ty, int tlg, int nmodes,
                                                     unsigned long long modes_lo, unsigned long long modes_hi) {
    unsigned acc = 0;
#ifdef WRENC_EXP_SKIP_SAD // instruction-count experiment only (profiles/r04_issue_model.md)
    return (unsigned)LANE;
#endif
    const int my_mode = LANE < nmodes ? (int)(((LANE < 8 ? modes_lo : modes_hi) >> (8 * (LANE & 7))) & 255u) : kNoMode;
    // [entry][blk][4n] projected references as bytes, each XOR 0x80 (= ref - 128 as a signed byte) so that
    // the interpolation filters run as one v_dot4_i32_i8 over four packed taps: sum(c * (ref - 128)) + 128 *
    // sum(c), and both luma filters sum to 64, the chroma filter to 32
    uint8_t* tab = (uint8_t*)SH.r1;
    uint32_t* ptab = (uint32_t*)SH.decw + 32; // [entry]: inv_angle (low half) | vertical << 16 | valid << 17
    uint32_t* ptab2 = (uint32_t*)SH.decw + 48; // [entry]: angle (low half) | flags << 16 | mode << 24
#pragma unroll 1
    for (int comp = 0; comp < 2; ++comp) {
        if (!((comps >> comp) & 1)) continue;
        const int cs = comp ? 1 : 0;
        const int nb = comp ? 2 : 1;
        const int lg = tlg - cs;
        const int n = 1 << lg;
        const int nn = n * n;
        const int cx = tx >> cs, cy = ty >> cs;
        const int obyte = org_byte(comp, tlg);
        const int lgs = lg + 2; // table stride 4n >= 3n + 4 per block
        // ---- parameters of my entry (intra_predictor.rs:1287-1310, 355-372) ----
        const bool valid = my_mode != kNoMode;
        const int mm = valid ? my_mode : 2;
        const int at = c.k->ang_tab[mm];
        const int my_angle = (int)(int16_t)(at & 0xFFFF);
        const int my_inv = at >> 16;
        int my_flags; // bit 0 filter_flag, bits 1-2 PDPC variant (0 none, 1 mode 18/50, 2 mode < 18, 3 mode > 50), bits 4.. n_scale
        {
            bool filter_flag = false;
            if (!(mm == 2 || mm == 34 || mm == 66)) {
                const int md = min(abs(mm - 50), abs(mm - 18));
                const int thr = lg == 2 ? 24 : (lg == 3 ? 14 : (lg == 4 ? 2 : 0));
                filter_flag = md > thr;
            }
            int n_scale;
            if (mm > 50 || (mm > 1 && mm < 18))
                n_scale = min(lg - ilog2i(3 * my_inv - 2) + 8, 2);
            else
                n_scale = (2 * lg - 2) >> 2;
            int kind = 0;
            if (mm == 18 || mm == 50)
                kind = 1;
            else if (mm < 18 && n_scale >= 0)
                kind = 2;
            else if (mm > 50 && n_scale >= 0)
                kind = 3;
            my_flags = (filter_flag ? 1 : 0) | (kind << 1) | (max(n_scale, 0) << 4);
        }
        PROF_MARK(sl0_);
        if (LANE < nmodes) {
            ptab[LANE] = ((uint32_t)my_inv & 0xFFFFu) | (mm >= 34 ? 0x10000u : 0u) | (valid ? 0x20000u : 0u);
            ptab2[LANE] = ((uint32_t)my_angle & 0xFFFFu) | ((uint32_t)my_flags << 16) | ((uint32_t)mm << 24);
        }
        WSYNC();
        // ---- projected main references of every entry (intra_predictor.rs:1311-1420) ----
        {
            const int oL0 = comp == 0 ? R_L0 : R_LC0, oA0 = comp == 0 ? R_A0 : R_AC0; // (filtered refs: modes 2, 34, 66 below)
            const int total = nmodes << (lgs + cs);
            // four table bytes per lane and iteration: they belong to one entry and lie on one side of index 0 (the
            // table of a block starts n entries below it, n a multiple of 4)
            for (int e = 4 * LANE; e < total; e += 256) {
                const int mi = e >> (lgs + cs);
                const int blk = cs ? ((e >> lgs) & 1) : 0;
                const int ee = e & ((1 << lgs) - 1);
                const uint32_t pw = ptab[mi];
                const int inv_angle = (int)(int16_t)(pw & 0xFFFF);
                const bool vertical = (pw >> 16) & 1;
                const int idx0 = ee - n;
                int oL = blk ? R_LC1 : oL0, oA = blk ? R_AC1 : oA0;
                if (comp == 0 && nn > 32) { // luma blocks of more than 32 samples: modes 2, 34, 66 use the filtered references
                    const int m = (int)(ptab2[mi] >> 24);
                    if (m == 2 || m == 34 || m == 66) {
                        oL = R_LF;
                        oA = R_AF;
                    }
                }
                const bool from_above = (idx0 >= 0) == vertical;
                uint32_t four = 0;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    const int idx = idx0 + b4;
                    const int k = idx0 >= 0 ? min(idx, 2 * n) : max(min((M24(idx, inv_angle) + 256) >> 9, n), 0);
                    four |= (uint32_t)SH.refs[k == 0 ? oL : (from_above ? oA + k - 1 : oL + k)] << (8 * b4);
                }
                *(uint32_t*)&tab[e] = four ^ 0x80808080u;
            }
        }
        WSYNC();
        PROF_MARK(sl1_);
        PROF_ADD2(PH_LEAF + 8, sl0_, sl1_);
#if WRENC_SAD4X4
        // An iteration of the block-per-lane code costs about nine of the sample-per-lane code below (16 samples a lane
        // instead of one, PDPC for every lane as soon as one entry has it) and takes 64 / G entries, whatever the list's
        // length: it pays when the list fills its iterations -- the 13 directional candidates at every size from 8x8, the
        // two probes of a step-search round only for the big blocks.
        const int lgG_ = 2 * (lg - 2) + (nb == 2 ? 1 : 0);
        const int its4_ = (nmodes + (64 >> lgG_) - 1) >> (6 - lgG_), its1_ = nmodes * ((nb * nn + 63) >> 6);
        if (nb * nn > 32 && its4_ * 9 < its1_) {
            // ---- a lane predicts one 4x4 BLOCK of samples of one entry: the G = nb (n / 4)^2 blocks of an entry sit side
            // by side in the wave, 64 / G entries share an iteration (32x32 luma: one entry per iteration, all 1024
            // samples in it).  The four samples of a block that lie next to each other ACROSS the prediction direction (a
            // row of the block for the vertical modes, a column for the horizontal ones) share the projection (i_idx,
            // i_fact), the filter taps and one 7-byte window of the projected references; their four results are
            // packed into a dword and meet the originals in one v_sad_u8 -- against the block's rows, or against its
            // columns (the originals transposed once per lane, in front of the loop over the entries).
            const int lgb = lg - 2;                        // blocks per side, log2
            const int lgG = 2 * lgb + (nb == 2 ? 1 : 0);   // blocks per entry, log2
            const int g = LANE & ((1 << lgG) - 1);
            const int slot = LANE >> lgG;
            const int blk = g >> (2 * lgb);
            const int bq = g & ((1 << (2 * lgb)) - 1);
            const int x0 = 4 * (bq & ((1 << lgb) - 1)), y0 = 4 * (bq >> lgb);
            const uint32_t* orow = (const uint32_t*)((const uint8_t*)SH.r2 + obyte + blk * nn + y0 * n + x0);
            const uint32_t o0 = orow[0], o1 = orow[n >> 2], o2 = orow[2 * (n >> 2)], o3 = orow[3 * (n >> 2)];
            // the 4x4 bytes transposed: t_r = byte r of o0 | o1 | o2 | o3
            const uint32_t a01 = __builtin_amdgcn_perm(o1, o0, 0x05010400u), b01 = __builtin_amdgcn_perm(o1, o0, 0x07030602u);
            const uint32_t a23 = __builtin_amdgcn_perm(o3, o2, 0x05010400u), b23 = __builtin_amdgcn_perm(o3, o2, 0x07030602u);
            const uint32_t t0 = __builtin_amdgcn_perm(a23, a01, 0x05040100u), t1 = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
            const uint32_t t2 = __builtin_amdgcn_perm(b23, b01, 0x05040100u), t3 = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
#ifndef WRENC_SAD_SUMS_AT
#define WRENC_SAD_SUMS_AT 64
#endif
            static_assert(WRENC_SAD_SUMS_AT >= 64 && WRENC_SAD_SUMS_AT + 16 <= 128, "Lds::decw: sums behind ptab2, inside decw");
            uint32_t* sums = (uint32_t*)SH.decw + WRENC_SAD_SUMS_AT; // [entry]: the SAD of this component
#pragma unroll 1
            for (int base = 0; base < nmodes; base += 64 >> lgG) {
                const int mi = base + slot;
                const int mic = min(mi, 15);
                const uint32_t pw = ptab[mic], pw2 = ptab2[mic];
                const bool on = mi < nmodes && ((pw >> 17) & 1);
                const int inv_angle = (int)(int16_t)(pw & 0xFFFF);
                const bool vertical = (pw >> 16) & 1;
                const int angle = on ? (int)(int16_t)(pw2 & 0xFFFF) : 0; // (a lane without an entry stays inside the tables)
                const int flags = (int)((pw2 >> 16) & 0xFF);
                const int mode = (int)(pw2 >> 24);
                const bool filter_flag = flags & 1;
                const int kind = (flags >> 1) & 3;
                const int n_scale = flags >> 4;
                const int a0 = vertical ? y0 : x0, c0 = vertical ? x0 : y0; // along / across the prediction direction
                const bool filt = comp == 0 && nn > 32 && (mode == 2 || mode == 34 || mode == 66);
                const int oL = blk ? R_LC1 : (comp == 0 ? (filt ? R_LF : R_L0) : R_LC0);
                const int oA = blk ? R_AC1 : (comp == 0 ? (filt ? R_AF : R_A0) : R_AC0);
                // PDPC (intra_predictor.rs:355-757) weighs by the position across the direction and is over after
                // 3 << n_scale samples; its reference runs along it: left[] = L + 1 for the vertical modes, above[] = A
                const bool pdpc = on && kind != 0 && c0 < (3 << n_scale);
                const bool any_pdpc = __ballot(pdpc) != 0ULL;
                const ref_t* side = SH.refs + (vertical ? oL + 1 : oA);
                const int alrs = SH.refs[oL];
                const int tb = (((mic << cs) + blk) << lgs) + n + c0;
                // per sample across the direction (k): the PDPC weight, 0 for a lane without PDPC or beyond its reach, and
                // where its reference sits relative to `along`
                int wp[4], dk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    wp[k] = pdpc ? pdpc_w(n_scale, c0 + k) : 0;
                    dk[k] = kind == 1 ? 0 : ((M24(c0 + k + 1, inv_angle) + 256) >> 9);
                }
                int sad = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int along = a0 + r;
                    const int pr = M24(along + 1, angle);
                    const int i_idx = pr >> 5, i_fact = pr & 31;
                    const int ta = tb + i_idx; // the four samples' taps = ref[ta + k + 0..3]
                    const uint32_t* tp = (const uint32_t*)(tab + (ta & ~3));
                    const uint32_t w0 = tp[0], w1 = tp[1], w2 = tp[2];
                    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, ta & 3), hi = __builtin_amdgcn_alignbyte(w2, w1, ta & 3);
                    int v[4];
                    const int wgt = comp == 0 ? (filter_flag ? 0x00102010 + (i_fact >> 1) * 0x0100FEFF : *(const int*)&SHT.fc[i_fact][0])
                                              : (((32 - i_fact) << 8) | (i_fact << 16));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int taps = (int)(k == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, k));
                        if (comp == 0)
                            v[k] = min(max(__builtin_amdgcn_sdot4(wgt, taps, 8192 + 32, false) >> 6, 0), 255);
                        else
                            v[k] = __builtin_amdgcn_sdot4(wgt, taps, 4096 + 16, false) >> 5;
                    }
#ifdef WRENC_EXP_OLD_PDPC // experiment only (profiles/r04_wrong_sads.md): round 3's first version, PDPC under a lane-divergent branch
                    if (any_pdpc && pdpc) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int w_ = pdpc_w(n_scale, c0 + k);
                            int rs_;
                            if (kind == 1)
                                rs_ = (int)(int16_t)(side[along] - alrs + v[k]);
                            else
                                rs_ = side[along + ((M24(c0 + k + 1, inv_angle) + 256) >> 9)];
                            const int pv_ = (int16_t)(M24(rs_, w_) + M24(64 - w_, v[k]) + 32) >> 6;
                            v[k] = min(max(pv_, 0), 255);
                        }
                    }
                    if (false) {
                        for (int k = 0; k < 4; ++k) {
#else
                    if (any_pdpc) { // (wave-uniform; inside it every lane runs the same code: a lane without PDPC weighs by 0,
                                    // as does a sample beyond 3 << n_scale, so what is read for those does not matter)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
#endif
                            const int sv = side[pdpc ? along + dk[k] : 0];
                            const int rs = kind == 1 ? (int)(int16_t)(sv - alrs + v[k]) : sv;
                            const int pv = (int16_t)(M24(rs, wp[k]) + M24(64 - wp[k], v[k]) + 32) >> 6;
                            v[k] = min(max(pv, 0), 255);
                        }
                    }
                    const uint32_t packed = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
                    const uint32_t org = vertical ? (r == 0 ? o0 : (r == 1 ? o1 : (r == 2 ? o2 : o3)))
                                                  : (r == 0 ? t0 : (r == 1 ? t1 : (r == 2 ? t2 : t3)));
                    sad = (int)__builtin_amdgcn_sad_u8(packed, org, (uint32_t)sad);
                }
                if (!on) sad = 0;
                // the entry's total: over the G lanes of its blocks
                int tot = sad;
                if (lgG >= 1) tot += dpp_mov<kDppSwap1>(tot);
                if (lgG >= 2) tot += dpp_mov<kDppSwap2>(tot);
                if (lgG >= 3) tot += dpp_mov<kDppRowHalfMirror>(tot);
                if (lgG >= 4) tot += dpp_mov<kDppRowMirror>(tot);
                if (lgG == 5) {
                    const int ta_ = __builtin_amdgcn_readlane(tot, 0) + __builtin_amdgcn_readlane(tot, 16);
                    const int tb_ = __builtin_amdgcn_readlane(tot, 32) + __builtin_amdgcn_readlane(tot, 48);
                    tot = LANE < 32 ? ta_ : tb_;
                } else if (lgG == 6) {
                    tot = __builtin_amdgcn_readlane(tot, 0) + __builtin_amdgcn_readlane(tot, 16) + __builtin_amdgcn_readlane(tot, 32) +
                          __builtin_amdgcn_readlane(tot, 48);
                }
                if (g == 0 && mi < nmodes) sums[mi] = (uint32_t)tot;
            }
            WSYNC();
            if (LANE < nmodes) acc += sums[LANE];
            WSYNC(); // the next component overwrites the tables
            PROF_MARK(sl2_);
            PROF_ADD2(PH_LEAF + 9, sl1_, sl2_);
            continue;
        }
#endif
        if (nb * nn <= 32) {
            // ---- small blocks (4x4 luma: 16 samples, 4x4 chroma pair: 32): 4 or 2 entries share an
            // iteration, the entry's parameters are per-lane values ----
            const int lgS = nb * nn == 32 ? 5 : 4;
            const int slot = LANE >> lgS;
            const int i = LANE & ((1 << lgS) - 1);
            const int blk = i >> (2 * lg);
            const int ii = i & (nn - 1);
            const int x = ii & (n - 1), y = ii >> lg;
            const int o = ((const uint8_t*)SH.r2)[obyte + i];
            const ref_t* L = SH.refs + (blk ? R_LC1 : (comp == 0 ? R_L0 : R_LC0));
            const ref_t* A = SH.refs + (blk ? R_AC1 : (comp == 0 ? R_A0 : R_AC0));
#pragma unroll 1
            for (int base = 0; base < nmodes; base += 64 >> lgS) {
                const int mi = base + slot;
                const uint32_t pw = ptab[min(mi, 15)], pw2 = ptab2[min(mi, 15)];
                const bool on = mi < nmodes && ((pw >> 17) & 1);
                const int inv_angle = (int)(int16_t)(pw & 0xFFFF);
                const bool vertical = (pw >> 16) & 1;
                const int angle = (int)(int16_t)(pw2 & 0xFFFF);
                const int flags = (int)((pw2 >> 16) & 0xFF);
                const int mode = (int)(pw2 >> 24);
                const bool filter_flag = flags & 1;
                const int kind = (flags >> 1) & 3;
                const int n_scale = flags >> 4;
                const int along = vertical ? y : x, across = vertical ? x : y;
                const int i_idx = M24(along + 1, angle) >> 5;
                const int i_fact = M24(along + 1, angle) & 31;
                const int ta = (((min(mi, 15) << cs) + blk) << lgs) + n + across + i_idx; // taps = ref[across + i_idx + 0..3]
                const uint32_t* tp = (const uint32_t*)(tab + (ta & ~3));
                const int taps = (int)__builtin_amdgcn_alignbyte(tp[1], tp[0], ta & 3);
                int v;
                if (comp == 0) {
                    const int w = filter_flag ? 0x00102010 + (i_fact >> 1) * 0x0100FEFF : *(const int*)&SHT.fc[i_fact][0];
                    v = min(max(__builtin_amdgcn_sdot4(w, taps, 8192 + 32, false) >> 6, 0), 255);
                } else {
                    v = __builtin_amdgcn_sdot4(((32 - i_fact) << 8) | (i_fact << 16), taps, 4096 + 16, false) >> 5;
                }
                if (kind != 0) { // PDPC, intra_predictor.rs:355-757; left[] = L+1, above[] = A
                    int rl = 0, rt = 0, wl = 0, wt = 0;
                    if (kind == 1) {
                        const int alrs = L[0];
                        rl = (int16_t)(L[y + 1] - alrs + v);
                        rt = (int16_t)(A[x] - alrs + v);
                        wl = mode == 50 ? pdpc_w(n_scale, x) : 0;
                        wt = mode == 18 ? pdpc_w(n_scale, y) : 0;
                    } else if (kind == 2) {
                        const int dx_int = (M24(y + 1, inv_angle) + 256) >> 9;
                        rt = y < (3 << n_scale) ? A[x + dx_int] : 0;
                        wt = pdpc_w(n_scale, y);
                    } else {
                        const int dy_int = (M24(x + 1, inv_angle) + 256) >> 9;
                        rl = x < (3 << n_scale) ? L[1 + y + dy_int] : 0;
                        wl = pdpc_w(n_scale, x);
                    }
                    v = (int16_t)(M24(rl, wl) + M24(rt, wt) + M24(64 - wt - wl, v) + 32) >> 6;
                    v = min(max(v, 0), 255);
                }
                const int d = o - v;
                const int rs = row_sum_i32(on ? (d < 0 ? -d : d) : 0); // every lane: total of its row of 16
                // slot totals: 16-sample slots are the rows, 32-sample slots two rows each
                const int t0 = __builtin_amdgcn_readlane(rs, 0), t1 = __builtin_amdgcn_readlane(rs, 16),
                          t2 = __builtin_amdgcn_readlane(rs, 32), t3 = __builtin_amdgcn_readlane(rs, 48);
                if (lgS == 4) {
                    acc += LANE == base ? (unsigned)t0 : (LANE == base + 1 ? (unsigned)t1 : (LANE == base + 2 ? (unsigned)t2 : (LANE == base + 3 ? (unsigned)t3 : 0u)));
                } else {
                    acc += LANE == base ? (unsigned)(t0 + t1) : (LANE == base + 1 ? (unsigned)(t2 + t3) : 0u);
                }
            }
            WSYNC();
            PROF_MARK(sl3_);
            PROF_ADD2(PH_LEAF + 10, sl1_, sl3_);
            continue;
        }
        // ---- entry by entry: one predicted sample per lane and iteration, |org - pred| summed ----
#pragma unroll 1
        for (int mi = 0; mi < nmodes; ++mi) {
            const int mode = __builtin_amdgcn_readlane(my_mode, mi);
            if (mode == kNoMode) continue;
            const int angle = __builtin_amdgcn_readlane(my_angle, mi);
            const int inv_angle = __builtin_amdgcn_readlane(my_inv, mi);
            const int flags = __builtin_amdgcn_readlane(my_flags, mi);
            const bool filter_flag = flags & 1;
            const int kind = (flags >> 1) & 3;
            const int n_scale = flags >> 4;
            const bool vertical = mode >= 34;
            const bool filt = comp == 0 && nn > 32 && (mode == 2 || mode == 34 || mode == 66);
            const int oL0 = comp == 0 ? (filt ? R_LF : R_L0) : R_LC0;
            const int oA0 = comp == 0 ? (filt ? R_AF : R_A0) : R_AC0;
            int sad = 0;
#if WRENC_U_SAD > 0
            WRENC_UNROLL(WRENC_U_SAD)
#endif
            for (int i = LANE; i < nb * nn; i += 64) {
                const int blk = i >> (2 * lg);
                const int ii = i & (nn - 1);
                const int x = ii & (n - 1), y = ii >> lg;
                const int o = ((const uint8_t*)SH.r2)[obyte + i];
                const ref_t* L = SH.refs + (blk ? R_LC1 : oL0);
                const ref_t* A = SH.refs + (blk ? R_AC1 : oA0);
                const int along = vertical ? y : x, across = vertical ? x : y;
                const int i_idx = M24(along + 1, angle) >> 5;
                const int i_fact = M24(along + 1, angle) & 31;
                const int ta = (((mi << cs) + blk) << lgs) + n + across + i_idx; // taps = ref[across + i_idx + 0..3]
                const uint32_t* tp = (const uint32_t*)(tab + (ta & ~3));
                const int taps = (int)__builtin_amdgcn_alignbyte(tp[1], tp[0], ta & 3);
                int v;
                if (comp == 0) {
                    const int w = filter_flag ? 0x00102010 + (i_fact >> 1) * 0x0100FEFF : *(const int*)&SHT.fc[i_fact][0];
                    v = min(max(__builtin_amdgcn_sdot4(w, taps, 8192 + 32, false) >> 6, 0), 255);
                } else {
                    v = __builtin_amdgcn_sdot4(((32 - i_fact) << 8) | (i_fact << 16), taps, 4096 + 16, false) >> 5;
                }
                if (kind != 0) { // PDPC, intra_predictor.rs:355-757; left[] = L+1, above[] = A
                    int rl = 0, rt = 0, wl = 0, wt = 0;
                    if (kind == 1) {
                        const int alrs = L[0];
                        rl = (int16_t)(L[y + 1] - alrs + v);
                        rt = (int16_t)(A[x] - alrs + v);
                        wl = mode == 50 ? pdpc_w(n_scale, x) : 0;
                        wt = mode == 18 ? pdpc_w(n_scale, y) : 0;
                    } else if (kind == 2) {
                        const int dx_int = (M24(y + 1, inv_angle) + 256) >> 9;
                        rt = y < (3 << n_scale) ? A[x + dx_int] : 0;
                        wt = pdpc_w(n_scale, y);
                    } else {
                        const int dy_int = (M24(x + 1, inv_angle) + 256) >> 9;
                        rl = x < (3 << n_scale) ? L[1 + y + dy_int] : 0;
                        wl = pdpc_w(n_scale, x);
                    }
                    v = (int16_t)(M24(rl, wl) + M24(rt, wt) + M24(64 - wt - wl, v) + 32) >> 6;
                    v = min(max(v, 0), 255);
                }
                const int d = o - v;
                sad += d < 0 ? -d : d;
            }
            const int total = wave_sum_i32(sad);
            acc += LANE == mi ? (unsigned)total : 0u;
        }
        WSYNC(); // the next component overwrites the tables
        PROF_MARK(sl4_);
        PROF_ADD2(PH_LEAF + 10, sl1_, sl4_);
    }
    return acc;
}

} // namespace wrenc
