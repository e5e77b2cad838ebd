// dev_search.h -- evaluation requests and the evaluator, the leaf / CTU search state machines, CTU entry (block_splitter.rs, ctu_encoder.rs:1421-1461)
// Part of the gfx950 device code of the RD-search path; see wrenc_dev.h for the overall model.
#pragma once

namespace wrenc {

// ---------------------------------------------------------------------------
// RD search building blocks (block_splitter.rs)
// ---------------------------------------------------------------------------
// Evaluation requests and the evaluator
// ---------------------------------------------------------------------------
enum { K_SADLIST = 0, K_FULL = 1, K_NOP = 2, K_SADSEARCH = 3, K_CCLMSEARCH = 4, K_LEAF4 = 5, K_LEAFC4 = 6, K_LEAF8 = 7, K_LEAF16 = 8, K_SPLIT8 = 9, K_SERVE4 = 10 };
#ifndef WRENC_POOL_MIN_TLG
#define WRENC_POOL_MIN_TLG 6
#endif
enum { COPY_NONE = 0, COPY_SAVE = 1, COPY_RESTORE = 2, COPY_PULL = 3 };
// build knobs (tools/README.md): the level schedule of the team kernel at max-split-depth 3; an 8x8 CU's split as one request
#ifndef WRENC_LEVELS
#define WRENC_LEVELS 1
#endif
#ifndef WRENC_SPLIT8
#define WRENC_SPLIT8 1
#endif
#ifndef WRENC_SERVER // level schedule: member 0, idle once the 32x32 candidate is done, runs pack {planar, DC} of member 2's 8x8 leaves
#define WRENC_SERVER 1
#endif
#ifndef WRENC_LEVELS_ALL_DEPTHS  // 0: the level schedule at max-split-depth 3 only, round 2's team below
#define WRENC_LEVELS_ALL_DEPTHS 1
#endif

struct Req {
    int kind;       // K_SADLIST: predict + SAD of a list of modes (block_splitter.rs:64-108, 476-522);
                    // K_FULL: predict .. reconstruct (:146-185); K_SADSEARCH: the whole SAD part of a leaf search in
                    // one request -- the 13 directional candidates, their first minimum and the two step-search
                    // rounds around it (:899-973): returns the mode (Res::imin) and its SAD (Res::vmin);
                    // K_CCLMSEARCH: the CCLM part of a leaf search in one request -- the SADs of LT / T / L_CCLM, the
                    // pick (:847-854) and the full evaluation of the chroma pair with it: Res::imin = the mode, + parts;
                    // K_LEAF4: the whole search of a 4x4 DUAL_TREE_LUMA leaf (:886-1078) in one request, its full candidates
                    // evaluated side by side in the wave's four 16-lane rows (leaf4_search): Res::imin = the mode, vmin = its cost;
                    // K_LEAFC4: the same for the DUAL_TREE_CHROMA leaf of a split 8x8 CU (:794-885; leafc4_search), mc = the DM mode
                    // K_LEAF8: the whole search of an 8x8 SINGLE_TREE leaf (:886-1078) in one request, its full candidates
                    // evaluated in packs of two and three (leaf8_search): Res::imin / imin2 = luma / chroma mode, vmin = the cost;
                    // n = which parts run here (bit 0 pack {planar, DC}, bit 1 the SAD search + pack {cm, cm - 1, cm + 1},
                    // bit 2 the CCLM part on the winner ml with DM chroma cost fcur)
                    // K_SPLIT8: the split candidate of an 8x8 CU in one request: its four DUAL_TREE_LUMA 4x4 leaf searches, then the
                    // DUAL_TREE_CHROMA one (split8_search); Res::vmin = the split cost (:1116-1123)
                    // K_LEAF16: the five full candidates of a 16x16 SINGLE_TREE leaf and its SAD search in one request, the
                    // candidates in packs of two (leaf16_search): Res::imin = the best luma mode, vmin = its cost, + its parts
    int comps;      // bit 0: luma block, bit 1: Cb+Cr pair
    int tx, ty, tlg;
    int ml, mc;     // K_FULL: luma / chroma mode
    bool shared;    // quantiser: pooled Viterbi of the workgroup (search) or solo (regen, final pass)
    bool active;    // false: walk the schedule only (keeps the workgroup's barriers aligned)
    bool refs0, refs1; // (re)build the luma / chroma reference samples of the block first
    bool final;     // final pass: store the levels, count reconstruction changes
    int stage;      // before anything else: stage the block's originals in LDS (component bits; blocks <= 16x16 only)
    // team schedule only (leaf_step_team): COPY_PULL copies the block from the LDS tile of member copy_from into
    // this member's tile (every member issues it, the source included, and all meet at a barrier behind it);
    // xchg: the team exchanges results after this request
    int copy_from;
    bool xchg;
    int n;          // K_SADLIST: number of entries
    int tree;       // tree type of the leaf that asks (diagnostic trace only)
    // before the evaluation: save the block's reconstruction to a slot / restore it from there
    // (the reference's cache_reconsts / restore_reconsts, block_splitter.rs:807-840, 1085-1145)
    int pre_copy, copy_comps, copy_slot, copy_tx, copy_ty, copy_tlg;
    unsigned long long modes_lo, modes_hi; // K_SADLIST: one byte per entry (8 + 8), the same mode for luma and chroma
    float fcur;     // K_LEAF8, part 4 alone (team schedule): the winner's DM chroma cost (:1040)
};

struct Res {
    // K_FULL: SSD and level cost of the luma block and of the chroma pair
    uint32_t ssd_y, ssd_c;
    long long lvl_y, lvl_c;
    // K_SADLIST: costs of the first three entries, first minimum (strict <) and its index
    float v0, v1, v2, vmin;
    int imin;
    int imin2;      // K_LEAF8: the chroma mode
};

__device__ __forceinline__ float uni_f(float v) { return __int_as_float(uni(__float_as_int(v))); }

// The final pass's self-check ("the final pass reproduces the search" as a property to test): a
// position-weighted sum, mod 2^32, of the samples of a block's reconstruction in the tile (comp 0: luma block, 1: chroma
// pair).  Every weight is odd, so a change of any ONE sample changes the sum; changes of several samples cancel with
// probability 2^-32.  (Until round 3 the final pass kept its prediction in global scratch so that the tile still held
// the search's reconstruction when the new one was made, and compared them sample by sample: 1.5 KB per CTU written to
// memory and read back, per block a store -> load round trip through L2.)
__device__ __forceinline__ unsigned checksum_weight(int i) { return (2u * (unsigned)i + 1u) * 0x9E3779B1u; }
__device__ __forceinline__ unsigned tile_checksum(int comp, int cx, int cy, int lg) {
    const int n = 1 << lg, nn = n * n, nb = comp ? 2 : 1;
    unsigned h = 0;
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg), ii = i & (nn - 1);
        h += (unsigned)rec_get(comp + blk, cx + (ii & (n - 1)), cy + (ii >> lg)) * checksum_weight(i);
    }
    return (unsigned)wave_sum_i32((int)h);
}

// First half of a full evaluation of one component (comp 0: luma block, 1: chroma pair): reference
// samples, prediction, forward transform.  Residual / coefficients at r1[rbase ..], prediction bytes
// in the tile (the final pass first takes the checksum of what the search left there).
__device__ __forceinline__ void full_front(const Ctx& c, const Req& q, int comp, int mode, int rbase,
                                           CclmPick pick = CclmPick{}) {
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int lg = q.tlg - cs;
    PROF_MARK(tr0_);
    if ((comp ? q.refs1 : q.refs0) && mode < LT_CCLM) build_refs(c, comp, q.tx, q.ty, q.tlg);
    if (q.final) {
        const unsigned h = tile_checksum(comp, q.tx >> cs, q.ty >> cs, lg);
        if (LANE == 0) SH.fsum = h;
        WSYNC();
    }
    PROF_MARK(t0_);
    PROF_ADD2(PH_REFS, tr0_, t0_);
    predict<true>(c, comp, q.tx, q.ty, q.tlg, mode, rbase, PRED_TILE, 0, pick);
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, lg, nb, rbase);
    PROF_MARK(t2_);
    PROF_ADD2(PH_FDCT, t1_, t2_);
}

// Second half: levels at r1[rbase ..] -> (final pass: store them) -> dequantise, inverse transform,
// reconstruct into the tile; returns the SSD against the originals (block_splitter.rs:146-185)
__device__ __forceinline__ uint32_t full_back(const Ctx& c, const PicBufs& pb, const Req& q, int comp, int rbase,
                                              bool any_level) {
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int lg = q.tlg - cs;
    const int n = 1 << lg;
    const int nn = n * n;
    const int cx = q.tx >> cs, cy = q.ty >> cs;
    const int obyte = org_byte(comp, q.tlg);
    const bool olds = q.tlg <= 4;
    PROF_MARK(t3_);
    // Final pass: the levels go to the slot's planes -- the 4x4 blocks of them that have any.  A block of zeros is only
    // written where the planes may still hold levels of the slot's previous picture (PicBufs::lev_dirty, loaded into
    // Lds::lev_was when the final pass began): nearly every 4x4 block of nearly all content is zero, and was.
    // A lane takes one row of a 4x4 block (four levels, one 8-byte store), the four rows of a block sit in one quad.
    if (q.final && c.store) {
        const int stride = c.W >> cs;
        const size_t at = (size_t)((c.ctu_y + q.ty) >> cs) * stride + ((c.ctu_x + q.tx) >> cs);
        GLOBAL_AS int16_t* lev0 = AS_GLOBAL(int16_t, pb.lev[0]) + plane_off(c, comp) + at;
        GLOBAL_AS int16_t* lev1 = AS_GLOBAL(int16_t, pb.lev[0]) + plane_off(c, 2) + at;
        const int lgb = lg - 2;                                  // 4x4 blocks per side of the transform block, log2
        const int u0x = (q.tx >> cs) >> 2, u0y = (q.ty >> cs) >> 2; // its place among the CTU's 4x4 blocks of the plane
        const uint32_t w0 = (uint32_t)uni((int)SH.lev_was[0]), w1 = (uint32_t)uni((int)SH.lev_was[1]),
                       w2 = (uint32_t)uni((int)SH.lev_was[2]), w3 = (uint32_t)uni((int)SH.lev_was[3]);
        for (int j = LANE; j < (nb << (2 * lg - 2)); j += 64) {
            const int bq_ = j >> 2, r = j & 3;
            const int pl = bq_ >> (2 * lgb), bq = bq_ & ((1 << (2 * lgb)) - 1);
            const int bby = bq >> lgb, bbx = bq & ((1 << lgb) - 1);
            const int y = 4 * bby + r, x = 4 * bbx;
            const unsigned long long v = *(const unsigned long long*)&SH.r1[rbase + (pl << (2 * lg)) + (y << lg) + x];
            const unsigned long long rows = __ballot(v != 0ULL);
            const bool nzb = ((rows >> (LANE & 60)) & 15ULL) != 0; // the block has a level
            const int ux = u0x + bbx, uy = u0y + bby;
            const int quad = comp ? (uy >> 1) * 2 + (ux >> 1) : (uy >> 2) * 2 + (ux >> 2);
            const int bit = comp ? 16 + (uy & 1) * 2 + (ux & 1) : (uy & 3) * 4 + (ux & 3);
            const uint32_t wq = quad == 0 ? w0 : (quad == 1 ? w1 : (quad == 2 ? w2 : w3));
            if (nzb && r == 0) atomicOr(&SH.lev_now[quad], 1u << bit);
            if (nzb || ((wq >> bit) & 1u)) *(GLOBAL_AS unsigned long long*)&(pl ? lev1 : lev0)[(size_t)y * stride + x] = v;
        }
    }
    // all levels zero: the residual is zero too, and r1 (the levels) already says so
    if (any_level) dequantize_t(c, lg, nb, rbase);
    PROF_MARK(t4_);
    if (any_level) inv_dct_lg(c, lg, nb, rbase);
    PROF_MARK(t5_);
    PROF_ADD2(PH_DEQ, t3_, t4_);
    PROF_ADD2(PH_IDCT, t4_, t5_);
    unsigned int part = 0;
    unsigned hs = 0;
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg), ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int pc = comp + blk;
        const int pred = rec_get(pc, cx + x, cy + y);
        int v = (int16_t)(pred + (int)SH.r1[rbase + i]); // pred as i16 + res, clamp (:178)
        v = min(max(v, 0), 255);
        if (q.final) hs += (unsigned)v * checksum_weight(i);
        rec_put(pc, cx + x, cy + y, v);
        const int d = v - (olds ? (int)((const uint8_t*)SH.r2)[obyte + i] : org_get(c, pc, cx + x, cy + y));
        part += (unsigned)M24(d, d);
    }
    const uint32_t ssd = (uint32_t)wave_sum_i32((int)part); // <= 1024 * 255^2: fits 32 bits
    if (q.final) { // the block's reconstruction must be what the search left in the tile (tile_checksum, full_front)
        const bool changed = (unsigned)wave_sum_i32((int)hs) != (unsigned)uni((int)SH.fsum);
        if (changed && LANE == 0 && c.store) atomicAdd(c.mismatch, 1ULL);
    }
    WSYNC();
    PROF_MARK(t6_);
    PROF_ADD2(PH_RECON, t5_, t6_);
    return ssd;
}

// Save the reconstruction of a block (comps bit 0: luma n x n, bit 1: Cb and Cr (n/2) x (n/2)) from
// the LDS tile to a slot in global scratch, or restore it from there.  Dwords: block corners are
// multiples of 4 samples in every plane that takes part.
__device__ __forceinline__ void copy_block(const Ctx& c, int mode, int comps, int slot, int tx, int ty, int tlg,
                                           int from_member = -1) {
    if (mode == COPY_PULL) { // tile to tile inside the workgroup's LDS (team schedule)
        if (from_member != c.member) {
            const Lds& src = SHW[(WAVE & ~(kTeam - 1)) + from_member];
            if (comps & 1) {
                const int words = 1 << (2 * tlg - 2);
                for (int w = LANE; w < words; w += 64) {
                    const int row = (4 * w) >> tlg, col = (4 * w) & ((1 << tlg) - 1);
                    const int at = (ty + row) * 36 + tx + col + 4;
                    *(uint32_t*)&SH.recY[at] = *(const uint32_t*)&src.recY[at];
                }
            }
            if (comps & 2) {
                const int lg = tlg - 1;
                const int words = 1 << (2 * lg - 2); // per plane
                for (int w = LANE; w < 2 * words; w += 64) {
                    const int pl = w >= words ? 1 : 0;
                    const int ww = w - pl * words;
                    const int row = (4 * ww) >> lg, col = (4 * ww) & ((1 << lg) - 1);
                    const int at = ((ty >> 1) + row) * 20 + (tx >> 1) + col + 4;
                    *(uint32_t*)&SH.recC[pl][at] = *(const uint32_t*)&src.recC[pl][at];
                }
            }
        }
        __syncthreads(); // nobody writes its tile again before every member has pulled
        return;
    }
    GLOBAL_AS uint32_t* g = (GLOBAL_AS uint32_t*)(c.slots + slot_offset(slot));
    const int cbase = 1 << (2 * tlg - 2); // the chroma planes follow the block's luma words
    if (comps & 1) {
        const int words = 1 << (2 * tlg - 2);
        for (int w = LANE; w < words; w += 64) {
            const int row = (4 * w) >> tlg, col = (4 * w) & ((1 << tlg) - 1);
            uint32_t* l = (uint32_t*)&SH.recY[(ty + row) * 36 + tx + col + 4];
            if (mode == COPY_SAVE)
                g[w] = *l;
            else
                *l = g[w];
        }
    }
    if (comps & 2) {
        const int lg = tlg - 1;
        const int words = 1 << (2 * lg - 2); // per plane
        for (int w = LANE; w < 2 * words; w += 64) {
            const int pl = w >= words ? 1 : 0;
            const int ww = w - pl * words;
            const int row = (4 * ww) >> lg, col = (4 * ww) & ((1 << lg) - 1);
            uint32_t* l = (uint32_t*)&SH.recC[pl][((ty >> 1) + row) * 20 + (tx >> 1) + col + 4];
            if (mode == COPY_SAVE)
                g[cbase + w] = *l;
            else
                *l = g[cbase + w];
        }
    }
    WSYNC();
}

// SADs of a list of n angular modes (one byte each in lo | hi; kNoMode = not evaluated, f32::MAX in the reference):
// SADs of the first two entries, the first minimum (smallest (sad, index) pair) and its index; kNoSad where none
struct ListOut {
    unsigned s0, s1, s2, smin;
    int imin;
};
__device__ __forceinline__ ListOut angular_list(const Ctx& c, const Req& q, int n, unsigned long long lo, unsigned long long hi) {
    constexpr unsigned kNoSad = 0xFFFFFFFFu;
    ListOut o;
    o.s0 = o.s1 = o.s2 = o.smin = kNoSad;
    o.imin = 0;
    const unsigned acc = sad_list_angular(c, q.comps, q.tx, q.ty, q.tlg, n, lo, hi);
    const int my_mode = LANE < n ? (int)(((LANE < 8 ? lo : hi) >> (8 * (LANE & 7))) & 255u) : kNoMode;
    if (c.trace && my_mode != kNoMode)
        TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, q.tlg, q.tree, (q.comps & 1) ? 0 : 2, (q.comps & 1) ? my_mode : 0, my_mode,
                  __float_as_int((float)acc));
    // first minimum = smallest (sad, index) pair
    const int key = my_mode != kNoMode ? (int)((acc << 4) | (unsigned)LANE) : 0x7FFFFFFF;
    const int kmin = wave_min_i32(key);
    if (kmin != 0x7FFFFFFF) {
        o.smin = (unsigned)kmin >> 4;
        o.imin = kmin & 15;
    }
    const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)acc, 0), a1 = (unsigned)__builtin_amdgcn_readlane((int)acc, 1),
                   a2 = (unsigned)__builtin_amdgcn_readlane((int)acc, 2);
    if ((int)(lo & 255u) != kNoMode) o.s0 = a0;
    if (n > 1 && (int)((lo >> 8) & 255u) != kNoMode) o.s1 = a1;
    if (n > 2 && (int)((lo >> 16) & 255u) != kNoMode) o.s2 = a2;
    return o;
}

#ifndef WRENC_STEP_ONE_LIST
#define WRENC_STEP_ONE_LIST 1
#endif
// The SAD part of a luma / single-tree leaf search: the 13 directional candidates, their first minimum and the two
// step-search rounds around it (block_splitter.rs:899-973).  cm: the mode found, smin: its SAD.
// ONE copy of the list code (sad_list_angular is a couple of thousand instructions, inlined): the rounds are iterations
// of one loop that differ in the list they ask for and in what they do with its SADs.  (Until round 4 the three kinds of
// round were three inlined copies at each of four call sites: a fifth of the kernel's code, in an instruction cache that
// misses on 2.6 % of its fetches -- profiles/r04_issue_model.md.)
__device__ __forceinline__ void sad_search(const Ctx& c, const Req& q, int& cm_out, unsigned& smin_out) {
    constexpr unsigned kNoSad = 0xFFFFFFFFu;
    // Small blocks (8x8 single-tree and 4x4 luma leaves): a list of two entries costs nearly what a list of six does
    // (parameters, tables, reductions: two thirds of it), so BOTH step-search rounds' probes come from ONE list -- round 1's
    // cm -+ 2 and the four modes round 2 can ask for around cm - 2, cm or cm + 2: cm - 3, cm - 1, cm + 1, cm + 3.  A
    // mode's SAD does not depend on the list it is in; the decisions below are the two rounds' in their order, an
    // entry the reference would not evaluate (Q12) is kNoMode here as there, and only the probes the reference makes
    // are traced.
    const bool six = WRENC_STEP_ONE_LIST && q.tlg <= 3;
    const int rounds = six ? 2 : 3;
    int cm = 0;
    unsigned cur = kNoSad;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        // ---- the round's list ----
        int n;
        unsigned long long lo, hi = 0;
        int e0 = kNoMode, e1 = kNoMode, e2 = kNoMode, e3 = kNoMode, e4 = kNoMode, e5 = kNoMode;
        const int st = r == 1 ? 2 : 1; // (the step of a two-probe round)
        if (r == 0) {
            // the 13 directional candidates {2,7,13,18,23,29,34,39,45,50,55,60,66} (:899-904)
            n = 13;
            lo = 2ULL | (7ULL << 8) | (13ULL << 16) | (18ULL << 24) | (23ULL << 32) | (29ULL << 40) | (34ULL << 48) | (39ULL << 56);
            hi = 45ULL | (50ULL << 8) | (55ULL << 16) | (60ULL << 24) | (66ULL << 32);
        } else if (six) {
            e0 = !(cm < 4) ? cm - 2 : kNoMode;
            e1 = !(cm + 2 > 66) ? cm + 2 : kNoMode;
            e2 = !(cm < 5) ? cm - 3 : kNoMode;
            e3 = !(cm < 3) ? cm - 1 : kNoMode;
            e4 = !(cm + 1 > 66) ? cm + 1 : kNoMode;
            e5 = !(cm + 3 > 66) ? cm + 3 : kNoMode;
            n = 6;
            lo = (unsigned long long)e0 | ((unsigned long long)e1 << 8) | ((unsigned long long)e2 << 16) | ((unsigned long long)e3 << 24) |
                 ((unsigned long long)e4 << 32) | ((unsigned long long)e5 << 40);
        } else {
            // step_search(mode, 2, cost, aux = true) (:905-973): rounds with step 2 and 1; a probe outside 2..66 is f32::MAX
            e0 = !(cm < 2 + st) ? cm - st : kNoMode;
            e1 = !(cm + st > 66) ? cm + st : kNoMode;
            n = 2;
            lo = (unsigned long long)e0 | ((unsigned long long)e1 << 8);
        }
        const unsigned acc = sad_list_angular(c, q.comps, q.tx, q.ty, q.tlg, n, lo, hi);
        const int my_mode = LANE < n ? (int)(((LANE < 8 ? lo : hi) >> (8 * (LANE & 7))) & 255u) : kNoMode;
        const unsigned s0 = (unsigned)__builtin_amdgcn_readlane((int)acc, 0), s1 = (unsigned)__builtin_amdgcn_readlane((int)acc, 1);
        // ---- what the round does with the SADs.  They are integers < 2^20, so comparing them as integers is comparing the
        // reference's f32 values; keep the current mode on ties, then the lower probe (Q12) ----
        bool traced = my_mode != kNoMode;
        if (r == 0) {
            // first minimum = smallest (sad, index) pair
            const int key = my_mode != kNoMode ? (int)((acc << 4) | (unsigned)LANE) : 0x7FFFFFFF;
            const int kmin = wave_min_i32(key);
            const int j = (kmin & 15) + 2; // entry i = candidate i + 2 of the 15, 7 bits each
            cm = j < 8 ? (int)((0x3A5C90D0E08080ULL >> (7 * j)) & 127) : (int)((0x109E3764B53A2ULL >> (7 * (j - 8))) & 127);
            cur = (unsigned)kmin >> 4;
        } else if (six) {
            const unsigned s2 = (unsigned)__builtin_amdgcn_readlane((int)acc, 2), s3 = (unsigned)__builtin_amdgcn_readlane((int)acc, 3);
            const unsigned s4 = (unsigned)__builtin_amdgcn_readlane((int)acc, 4), s5 = (unsigned)__builtin_amdgcn_readlane((int)acc, 5);
            // round 1, step 2
            const unsigned c0 = e0 != kNoMode ? s0 : kNoSad, c1 = e1 != kNoMode ? s1 : kNoSad;
            const unsigned mn = min(min(cur, c0), c1);
            int d = 0; // cm' - cm
            if (cur == mn) {
            } else if (c0 == mn) {
                d = -2;
                cur = c0;
            } else {
                d = 2;
                cur = c1;
            }
            // round 2, step 1, around cm' = cm + d: its probes are entries 2 + (d + 2) / 2 and 3 + (d + 2) / 2
            const int elo = d < 0 ? e2 : (d == 0 ? e3 : e4), ehi = d < 0 ? e3 : (d == 0 ? e4 : e5);
            const unsigned slo = d < 0 ? s2 : (d == 0 ? s3 : s4), shi = d < 0 ? s3 : (d == 0 ? s4 : s5);
            const unsigned p0 = elo != kNoMode ? slo : kNoSad, p1 = ehi != kNoMode ? shi : kNoSad;
            cm += d;
            const unsigned mn2 = min(min(cur, p0), p1);
            if (cur == mn2) {
            } else if (p0 == mn2) {
                cm -= 1;
                cur = p0;
            } else {
                cm += 1;
                cur = p1;
            }
            const int ilo = d < 0 ? 2 : (d == 0 ? 3 : 4);
            traced = traced && (LANE < 2 || LANE == ilo || LANE == ilo + 1);
        } else {
            const unsigned c0 = e0 != kNoMode ? s0 : kNoSad, c1 = e1 != kNoMode ? s1 : kNoSad;
            const unsigned mn = min(min(cur, c0), c1);
            if (cur == mn) {
            } else if (c0 == mn) {
                cm -= st;
                cur = c0;
            } else {
                cm += st;
                cur = c1;
            }
        }
        if (c.trace && traced)
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, q.tlg, q.tree, (q.comps & 1) ? 0 : 2, (q.comps & 1) ? my_mode : 0, my_mode,
                      __float_as_int((float)acc));
    }
    cm_out = cm;
    smin_out = cur;
}

__device__ __forceinline__ Res leaf4_search(const Ctx& c, const Req& q, int* overflow); // below, after the cost functions
__device__ __forceinline__ Res leafc4_search(const Ctx& c, const Req& q, int* overflow);
__device__ __forceinline__ Res leaf8_search(const Ctx& c, const Req& q, int* overflow);
__device__ __forceinline__ Res leaf16_search(const Ctx& c, const Req& q, int* overflow);
__device__ __forceinline__ Res split8_search(const Ctx& c, const Req& q, int* overflow);
__device__ __forceinline__ void serve_pack4(const Ctx& c, const Req& q, int* overflow);

// The evaluator: every block evaluation of the search, of the regeneration and of the final pass
// goes through this one inlined copy (the search logic below is a state machine that hands out
// evaluation requests; no function calls in the hot path).
// D3: the kernel serves max-split-depth 3 (the only depth at which an 8x8 CU splits into 4x4 leaves): the kernels for
// the smaller depths are built without that code.
template <bool D3>
__device__ __forceinline__ Res evaluate(const Ctx& c, const PicBufs& pb, const Req& q, int* overflow) {
    Res r;
    r.ssd_y = 0;
    r.ssd_c = 0;
    r.lvl_y = 0;
    r.lvl_c = 0;
    r.v0 = r.v1 = r.v2 = r.vmin = 3.40282347e+38f;
    r.imin = 0;
    r.imin2 = 0;
    PROF_MARK(tcp0_);
    if (q.pre_copy != COPY_NONE)
        copy_block(c, q.pre_copy, q.copy_comps, q.copy_slot, q.copy_tx, q.copy_ty, q.copy_tlg, q.copy_from);
    PROF_MARK(tcp1_);
    PROF_ADD2(PH_COPY, tcp0_, tcp1_);
    if (q.kind == K_NOP) return r;
    if (q.stage) stage_org_leaf(c, q.stage, q.tx, q.ty, q.tlg);
    // (with K_SPLIT8 and the level schedule on, nothing asks for a 4x4 leaf by itself: one copy of the two searches less)
    if (D3 && !(WRENC_SPLIT8 && WRENC_LEVELS)) {
        if (q.kind == K_LEAF4) return leaf4_search(c, q, overflow);
        if (q.kind == K_LEAFC4) return leafc4_search(c, q, overflow);
    }
    if (q.kind == K_LEAF8) return leaf8_search(c, q, overflow);
    if (q.kind == K_LEAF16) return leaf16_search(c, q, overflow);
    if (D3 && q.kind == K_SPLIT8) return split8_search(c, q, overflow);
    if (D3 && WRENC_SERVER && q.kind == K_SERVE4) { // (team kernel, member 0 only)
        serve_pack4(c, q, overflow);
        return r;
    }
    int mc = q.mc;
    CclmPick cpick_ = CclmPick{};
    if (q.kind == K_CCLMSEARCH) {
        // get_chroma_intra_pred_aux_cost of LT, T, L_CCLM in one sample pass, then the pick of :847-854 (SADs are
        // integers < 2^20: comparing them is comparing the reference's f32 values); the evaluation follows below
        if (q.tlg > 4) stage_org(c, 2, q.tx, q.ty, q.tlg);
        CclmParams call_;
        const unsigned acc = sad_list_cclm(c, q.tx, q.ty, q.tlg, &call_);
        const unsigned lt = (unsigned)__builtin_amdgcn_readlane((int)acc, 0), t = (unsigned)__builtin_amdgcn_readlane((int)acc, 1),
                       l = (unsigned)__builtin_amdgcn_readlane((int)acc, 2);
        if (c.trace && LANE < 3)
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, q.tlg, q.tree, 2, 0, LANE == 0 ? LT_CCLM : (LANE == 1 ? T_CCLM : L_CCLM),
                      __float_as_int((float)acc));
        mc = (lt <= t && lt <= l) ? LT_CCLM : (t <= l ? T_CCLM : L_CCLM);
        r.imin = mc;
        cpick_ = cclm_pick(call_, 2 * cclm_mode_index(mc)); // the evaluation below predicts with these
    }
    if (q.kind == K_FULL || q.kind == K_CCLMSEARCH) {
        // A candidate of the search with an 8x8 or 16x16 luma block quantises its three transform
        // blocks in one pass (quantize3, pooled or not): both components go through the first half, then
        // the pass, then both through the second half.  Everything else runs component by component
        // (the two share r1 / r2).  One copy of each stage either way.
        // Pooling the Viterbi walk of the workgroup's WPB blocks pays for long walks; a block of few positions walks
        // faster alone than the two workgroup barriers per chunk cost (WRENC_POOL_MIN_TLG: smallest CU size, log2,
        // whose transform blocks are pooled; every wave of the workgroup evaluates the same size, so they agree)
        const bool pooled = q.shared && !c.solo && q.tlg >= WRENC_POOL_MIN_TLG;
        #ifdef WRENC_EXP_NO_MERGED
        const bool merged = false;
#else
        const bool merged = !q.final && q.comps == 3 && q.tlg >= 3 && q.tlg <= 4; // (pooled or not)
#endif
        const int p0 = 1 << (2 * q.tlg);
        const int rounds = merged ? 1 : 2;
#pragma unroll 1
        for (int round = 0; round < rounds; ++round) {
            const int cset = merged ? 3 : (q.comps & (1 << round));
            if (!cset) continue;
            if (q.active) {
#pragma unroll 1
                for (int comp = 0; comp < 2; ++comp)
                    if ((cset >> comp) & 1) {
                        PROF_MARK(tf0_);
                        full_front(c, q, comp, comp ? mc : q.ml, (merged && comp) ? p0 : 0, cpick_);
                        PROF_MARK(tf1_);
                        PROF_ADD2(PH_PSZ + ((q.tlg - 2) * 2 + comp), tf0_, tf1_); // stage passes by block size and component
                        PROF_ADD2(PH_PCNT + ((q.tlg - 2) * 2 + comp), 0, 1);
                    }
            }
            PROF_MARK(ts0_);
            bool any_y = false, any_c = false;
            if (merged) {
                quantize3(c, q.tlg, pooled, q.active, overflow, &r.lvl_y, &r.lvl_c, &any_y, &any_c);
            } else {
                bool any = false;
                long long lvl = 0;
                if (pooled)
                    lvl = quantize(c, q.tlg - round, round ? 2 : 1, true, q.active, overflow, &any);
                else if (q.active)
                    lvl = quantize_solo(c, q.tlg - round, round ? 2 : 1, overflow, &any);
                if (round) {
                    r.lvl_c = lvl;
                    any_c = any;
                } else {
                    r.lvl_y = lvl;
                    any_y = any;
                }
            }
            PROF_MARK(ts1_);
            PROF_ADD2(PH_QZ + ((q.tlg - 2) & 3), ts0_, ts1_);
            if (!q.active) { // only kept the shared-Viterbi barriers company
                PROF_ADD2(PH_SKIP, ts0_, ts1_);
                continue;
            }
#pragma unroll 1
            for (int comp = 0; comp < 2; ++comp) {
                if (!((cset >> comp) & 1)) continue;
                PROF_MARK(tb0_);
                const uint32_t ssd = full_back(c, pb, q, comp, (merged && comp) ? p0 : 0, comp ? any_c : any_y);
                PROF_MARK(tb1_);
                PROF_ADD2(PH_PSZ + ((q.tlg - 2) * 2 + comp), tb0_, tb1_);
                if (comp)
                    r.ssd_c = ssd;
                else
                    r.ssd_y = ssd;
            }
        }
        return r;
    }
    // K_SADLIST: get_intra_pred_aux_cost / get_chroma_intra_pred_aux_cost of each listed mode
    PROF_MARK(tr0_);
    if (q.refs0 && (q.comps & 1)) build_refs(c, 0, q.tx, q.ty, q.tlg);
    if (q.refs1 && (q.comps & 2)) build_refs(c, 1, q.tx, q.ty, q.tlg);
    if (q.tlg > 4) stage_org(c, q.comps, q.tx, q.ty, q.tlg); // smaller blocks: staged once per leaf
    PROF_MARK(t0_);
    PROF_ADD2(PH_REFS, tr0_, t0_);
    // SADs stay integers (< 2^20, so the f32 the reference compares is exact and ordered the same
    // way); they become floats once, at the end.  An entry that is not evaluated costs f32::MAX.
    constexpr unsigned kNoSad = 0xFFFFFFFFu;
    unsigned s0 = kNoSad, s1 = kNoSad, s2 = kNoSad, smin = kNoSad;
    const int m_first = (int)(q.modes_lo & 255u);
    const int m_second = (int)((q.modes_lo >> 8) & 255u);
    if (q.kind == K_SADSEARCH) {
        int cm;
        sad_search(c, q, cm, smin);
        r.imin = uni(cm);
    } else if ((m_first >= 2 && m_first <= 66) || (m_first == kNoMode && m_second <= 66)) {
        // a list of angular modes (a step-search pair)
        const ListOut l = angular_list(c, q, q.n, q.modes_lo, q.modes_hi);
        s0 = l.s0;
        s1 = l.s1;
        s2 = l.s2;
        smin = l.smin;
        r.imin = l.imin;
    } else if (q.comps == 2 && q.n == 3 && (unsigned)q.modes_lo == ((unsigned)LT_CCLM | ((unsigned)T_CCLM << 8) | ((unsigned)L_CCLM << 16))) {
        // the three CCLM modes of a chroma pair, one sample pass
        const unsigned acc = sad_list_cclm(c, q.tx, q.ty, q.tlg);
        s0 = (unsigned)__builtin_amdgcn_readlane((int)acc, 0);
        s1 = (unsigned)__builtin_amdgcn_readlane((int)acc, 1);
        s2 = (unsigned)__builtin_amdgcn_readlane((int)acc, 2);
        if (c.trace && LANE < 3)
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, q.tlg, q.tree, 2, 0, LANE == 0 ? LT_CCLM : (LANE == 1 ? T_CCLM : L_CCLM),
                      __float_as_int((float)acc));
        smin = s0;
        r.imin = 0;
        if (s1 < smin) {
            smin = s1;
            r.imin = 1;
        }
        if (s2 < smin) {
            smin = s2;
            r.imin = 2;
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < q.n; ++i) {
            const int m = (int)(((i < 8 ? q.modes_lo : q.modes_hi) >> (8 * (i & 7))) & 255u);
            unsigned sad = kNoSad;
            if (m != kNoMode) {
                sad = 0;
#pragma unroll 1
                for (int comp = 0; comp < 2; ++comp) {
                    if (!((q.comps >> comp) & 1)) continue;
                    PROF_MARK(tp0_);
                    sad += (unsigned)wave_sum_i32(predict<false>(c, comp, q.tx, q.ty, q.tlg, m));
                    PROF_MARK(tp1_);
                    PROF_ADD2(PH_PSZ + ((q.tlg - 2) * 2 + comp), tp0_, tp1_);
                    PROF_ADD2(PH_PCNT + ((q.tlg - 2) * 2 + comp), 0, 1);
                }
            }
            if (c.trace && LANE == 0 && m != kNoMode)
                TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, q.tlg, q.tree, (q.comps & 1) ? 0 : 2, (q.comps & 1) ? m : 0, m,
                          __float_as_int((float)sad));
            if (i == 0) s0 = sad;
            if (i == 1) s1 = sad;
            if (i == 2) s2 = sad;
            if (sad < smin) { // first minimum
                smin = sad;
                r.imin = i;
            }
        }
    }
    r.v0 = s0 == kNoSad ? 3.40282347e+38f : uni_f((float)s0);
    r.v1 = s1 == kNoSad ? 3.40282347e+38f : uni_f((float)s1);
    r.v2 = s2 == kNoSad ? 3.40282347e+38f : uni_f((float)s2);
    r.vmin = smin == kNoSad ? 3.40282347e+38f : uni_f((float)smin);
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    return r;
}

// luma mode of the CU covering picture position (CTU-local x, y), as the search sees it
// (SURVEY.md Q7): inside the CTU -> root CU's mode; left CTU -> its final map; else none.
__device__ __forceinline__ int nb_luma_mode(Ctx c, int x, int y, bool* exists) {
    if (x >= 0 && y >= 0) {
        *exists = true;
        return c.cu32_mode;
    }
    if (y >= 0 && x < 0 && c.ctu_x > 0) {
        *exists = true;
        return uni((int)SH.left_mode[y >> 2]);
    }
    *exists = false;
    return PLANAR;
}

// mode class index for the header-bit table: 0 planar, 1..5 mpm_idx, 6..66 remainder
// (ctu.rs:1498-1635)
// The MPM list of a block (ctu.rs:1498-1635) depends on the block alone, not on the candidate mode: the packed leaf
// searches derive it once per leaf (mpm_list) and classify their five candidates against it (mpm_class_of).
struct MpmList {
    int k0, k1, k2, k3, k4;
};
__device__ __forceinline__ MpmList mpm_list(const Ctx& c, int bx, int by, int lg) {
    const int n = 1 << lg;
    bool le, ae;
    int left = nb_luma_mode(c, bx - 1, by + n - 1, &le);
    if (!le) left = PLANAR;
    int above;
    if (by - 1 < 0) {
        // above the CTU: either no CU (picture edge) or forced PLANAR across the CTU row (:1518-1523)
        above = PLANAR;
    } else {
        above = nb_luma_mode(c, bx + n - 1, by - 1, &ae);
        if (!ae) above = PLANAR;
    }
    int k0, k1, k2, k3, k4;
    if (left == above && left > DC) {
        const int m = left;
        k0 = m;
        k1 = 2 + (m + 61) % 64;
        k2 = 2 + (m - 1) % 64;
        k3 = 2 + (m + 60) % 64;
        k4 = 2 + m % 64;
    } else if (left != above && (left > DC || above > DC)) {
        const int mn = min(left, above), mx = max(left, above);
        if (mn > DC) {
            const int d = mx - mn;
            k0 = left;
            k1 = above;
            if (d == 1) {
                k2 = 2 + (mn + 61) % 64;
                k3 = 2 + (mx - 1) % 64;
                k4 = 2 + (mn + 60) % 64;
            } else if (d >= 62) {
                k2 = 2 + (mn - 1) % 64;
                k3 = 2 + (mx + 61) % 64;
                k4 = 2 + mn % 64;
            } else if (d == 2) {
                k2 = 2 + (mn - 1) % 64;
                k3 = 2 + (mn + 61) % 64;
                k4 = 2 + (mx - 1) % 64;
            } else {
                k2 = 2 + (mn + 61) % 64;
                k3 = 2 + (mn - 1) % 64;
                k4 = 2 + (mx + 61) % 64;
            }
        } else {
            k0 = mx;
            k1 = 2 + (mx + 61) % 64;
            k2 = 2 + (mx - 1) % 64;
            k3 = 2 + (mx + 60) % 64;
            k4 = 2 + mx % 64;
        }
    } else {
        k0 = DC;
        k1 = 50;
        k2 = 18;
        k3 = 46;
        k4 = 54;
    }
    MpmList l;
    l.k0 = k0;
    l.k1 = k1;
    l.k2 = k2;
    l.k3 = k3;
    l.k4 = k4;
    return l;
}
// mode class index for the header-bit table: 0 planar, 1..5 mpm_idx, 6..66 remainder
__device__ __forceinline__ int mpm_class_of(const MpmList& l, int mode) {
    if (mode == PLANAR) return 0;
    if (l.k0 == mode) return 1;
    if (l.k1 == mode) return 2;
    if (l.k2 == mode) return 3;
    if (l.k3 == mode) return 4;
    if (l.k4 == mode) return 5;
    // remainder = mode - 1 - #(candidates below mode) after sorting (:1613-1628)
    const int smaller = (l.k0 < mode) + (l.k1 < mode) + (l.k2 < mode) + (l.k3 < mode) + (l.k4 < mode);
    return 6 + (mode - 1 - smaller);
}
__device__ __forceinline__ int mpm_class(const Ctx& c, int bx, int by, int lg, int mode) {
    if (mode == PLANAR) return 0;
    return mpm_class_of(mpm_list(c, bx, by, lg), mode);
}

__device__ __forceinline__ float rd_cost(unsigned long long ssd, long long level, float lambda) {
    // block_splitter.rs:472-473: ssd as f32 + lambda * (level as f32 / 16384.0).  Rust never
    // contracts a*b+c into an FMA; HIP's default -ffp-contract=fast would, so contraction is
    // switched off here (and with -ffp-contract=off on the command line).
#pragma clang fp contract(off)
    const float lv = (float)level * (1.0f / 16384.0f);
    const float prod = lambda * lv;
    return (float)ssd + prod;
}

// SSD and level cost of the luma and of the chroma pair of one evaluated candidate.  Evaluations
// are deterministic functions of (block, mode, neighbourhood[, luma recon for CCLM]), so where the
// reference re-runs an evaluation it has already done (block_splitter.rs:1040,1068-1075) the
// parts are re-used and only the cost is re-assembled.
// get_intra_pred_cost (block_splitter.rs:110-474) from already evaluated parts, modes [ml, mc, mc]
__device__ __forceinline__ float assemble_cost(const Ctx& c, int tree, int cls, int mc, const EvalParts& e) {
    const bool single = tree == TREE_SINGLE;
    const int cc = (single && mc >= LT_CCLM) ? 1 + (mc - LT_CCLM) : 0;
    const unsigned long long ssd = (unsigned long long)e.ssd_y + (single ? (unsigned long long)e.ssd_c : 0ULL);
    const long long level = e.lvl_y + (single ? e.lvl_c : 0LL) + c.k->hb_luma[single ? 0 : 1][cc][cls];
    return rd_cost(ssd, level, c.k->lambda_rd);
}

// get_chroma_intra_pred_cost (block_splitter.rs:524-780) from already evaluated parts
__device__ __forceinline__ float assemble_chroma_cost(const Ctx& c, int mc, const EvalParts& e) {
    const long long level = e.lvl_c + c.k->hb_chroma[mc >= LT_CCLM ? 1 + (mc - LT_CCLM) : 0];
    return rd_cost((unsigned long long)e.ssd_c, level, c.k->lambda_rd_chroma);
}

__device__ __forceinline__ int pick_cclm(float lt, float t, float l) {
    // block_splitter.rs:847-854
    if (lt <= t && lt <= l) return LT_CCLM;
    if (t <= l) return T_CCLM;
    return L_CCLM;
}

// ---------------------------------------------------------------------------
// K_LEAF4: the whole search of a 4x4 DUAL_TREE_LUMA leaf in one request (block_splitter.rs:886-1078 with 16 samples
// per candidate).  A full evaluation of a 4x4 block is a chain of fixed-latency stages that keeps 16 of the wave's
// 64 lanes busy, and at max-split-depth 3 these leaves are 61 % of a CTU.  The full candidates of a leaf do not
// depend on each other (they read only neighbours outside the block, :887-898, :974), so they are evaluated SIDE BY
// SIDE, one candidate per 16-lane row: pack A = {planar, DC}, then the SAD search (sad_search), then pack B =
// {cm, cm - 1, cm + 1}.  Every stage is the one the single evaluation uses, run over nb blocks (forward / inverse
// transform, dequantisation) or written for rows (predict4_lane, quantize_p16).  The decisions are the reference's, in
// its order: first minimum of [planar, DC, cm, cm - 1, cm + 1] as a running strict-less update; a candidate outside
// 2..66 is not evaluated (f32::MAX there).  The best candidate's reconstruction goes to the tile when its pack is
// done (nothing reads the block's own area meanwhile: the reference samples are cached); no save / restore at all.
// ---------------------------------------------------------------------------
// the LDS working set of member m of this wave's team
__device__ __forceinline__ const Lds& team_lds(const Ctx& c, int m) { return SHW[(WAVE & ~(kTeam - 1)) + m]; }

// Level schedule, the pack-A server (see lv_decide for the schedule): polled LDS words, bounded like lv_meet
__device__ __forceinline__ unsigned lv_word(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lv_word_add(uint32_t* p) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (LANE == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lv_word_wait(const uint32_t* p, unsigned target) {
    if (LANE == 0) {
        int polls = 0;
        while (lv_word(p) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1 << 23)) {
                SHT.lvb.pad_ = 1;
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    WSYNC();
}

struct Pack4Out {
    uint32_t ssd[3];
    long long lvl[3];
    int rec; // per lane: the reconstructed sample of (candidate LANE / 16, sample LANE % 16)
};
__device__ __forceinline__ Pack4Out pack4_eval(const Ctx& c, const Req& q, int nb, int m0, int m1, int m2, int* overflow) {
    Pack4Out o;
    const int lane = lane_fresh();
    const int s = lane >> 4, i = lane & 15;
    const int mode = s == 0 ? m0 : (s == 1 ? m1 : (s == 2 ? m2 : kNoMode));
    const bool on = s < nb && mode != kNoMode;
    PROF_MARK(t0_);
    const int v = predict4_lane(c, on ? mode : kNoMode);
    const int org = ((const uint8_t*)SH.r2)[kOrgLeaf + i];
    if (s < nb) SH.r1[lane] = (int16_t)(on ? org - v : 0); // (a candidate that is not evaluated rides along as a zero block)
    WSYNC();
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, 2, nb, 0);
    PROF_MARK(t2_);
    PROF_ADD2(PH_FDCT, t1_, t2_);
    long long lvl[4];
    int any_mask = 0;
    quantize_p16(c, nb, overflow, lvl, &any_mask);
    PROF_MARK(t3_);
    if (any_mask) { // (all levels zero: the residuals are zero too, and r1 already says so)
        dequantize_t(c, 2, nb, 0);
        inv_dct_lg(c, 2, nb, 0);
    }
    PROF_MARK(t4_);
    PROF_ADD2(PH_IDCT, t3_, t4_);
    int rec = (int16_t)(v + (int)SH.r1[s < nb ? lane : 0]); // pred as i16 + res, clamp (:178)
    rec = min(max(rec, 0), 255);
    const int d = rec - org;
    const int row = row_sum_i32(on ? M24(d, d) : 0);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        o.ssd[b] = (uint32_t)__builtin_amdgcn_readlane(row, 16 * b);
        o.lvl[b] = lvl[b];
    }
    o.rec = rec;
    WSYNC();
    PROF_MARK(t5_);
    PROF_ADD2(PH_RECON, t4_, t5_);
    return o;
}

__device__ __forceinline__ Res leaf4_search(const Ctx& c, const Req& q, int* overflow) {
    Res r;
    r.ssd_y = 0;
    r.ssd_c = 0;
    r.lvl_y = 0;
    r.lvl_c = 0;
    r.v0 = r.v1 = r.v2 = 3.40282347e+38f;
    const int lane = lane_fresh();
    if (q.refs0) build_refs(c, 0, q.tx, q.ty, 2);
    const int x = lane & 3, y = (lane >> 2) & 3, row = lane >> 4;
    float best = 3.40282347e+38f;
    int best_mode = PLANAR;
    // one candidate of a pack that has come back: its cost, the trace record, the running first minimum; returns
    // whether it is the new best
    const MpmList mpl_ = mpm_list(c, q.tx, q.ty, 2);
#define LEAF4_CANDIDATE(P, B, M)                                                                                       \
    do {                                                                                                               \
        EvalParts e_;                                                                                                  \
        e_.ssd_y = (P).ssd[B];                                                                                         \
        e_.ssd_c = 0;                                                                                                  \
        e_.lvl_y = (P).lvl[B];                                                                                         \
        e_.lvl_c = 0;                                                                                                  \
        const int cls_ = mpm_class_of(mpl_, (M));                                                                      \
        const float val_ = uni_f(assemble_cost(c, TREE_DUAL_LUMA, cls_, (M), e_));                                     \
        if (c.trace && lane == 0) /* (team schedule: each half is traced by the member that runs it) */                \
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 2, TREE_DUAL_LUMA, 1, (M), (M), __float_as_int(val_));           \
        if (first_ || val_ < best) {                                                                                   \
            best = val_;                                                                                               \
            best_mode = (M);                                                                                           \
            win_ = (B);                                                                                                \
        }                                                                                                              \
        first_ = false;                                                                                                \
    } while (0)
    bool first_ = true;
    // q.n: which half runs here (team schedule: member 0 takes pack A, member 1 the SAD search and pack B; 3 = both)
    // level schedule (team kernel): member 3's 4x4 luma leaves get their pack A from member 0, the server (serve_pack4;
    // the same protocol as for member 2's 8x8 leaves, leaf8_search)
    constexpr int kSrv4Byte = 448; // in the server's decw: 2 x 16 reconstructed samples, then ssd[2] (u32), lvl[2] (i64)
    const bool served = WRENC_SERVER && c.solo && c.member == 3 && q.n == 3 && lv_word(&SHT.lvb.srv_ready) != 0;
    unsigned my_job = 0;
    if (served) {
        if (lane == 0) {
            SHT.lvb.job4_bx = (uint8_t)q.tx;
            SHT.lvb.job4_by = (uint8_t)q.ty;
        }
        my_job = uni((int)lv_word(&SHT.lvb.job4_posted)) + 1u;
        lv_word_add(&SHT.lvb.job4_posted);
    } else if (q.n & 1) {
        // pack A: planar and DC (:887-898)
        PROF_MARK(l4a0_);
        const Pack4Out a = pack4_eval(c, q, 2, PLANAR, DC, kNoMode, overflow);
        int win_ = -1;
        LEAF4_CANDIDATE(a, 0, PLANAR);
        LEAF4_CANDIDATE(a, 1, DC);
        if (row == win_) rec_put(0, q.tx + x, q.ty + y, a.rec);
        WSYNC();
        PROF_MARK(l4a1_);
        PROF_ADD2(PH_L4 + 1, l4a0_, l4a1_);
    }
    if (q.n & 2) {
        int cm;
        unsigned smin;
        PROF_MARK(l4s0_);
        sad_search(c, q, cm, smin);
        PROF_MARK(l4s1_);
        PROF_ADD2(PH_L4 + 2, l4s0_, l4s1_);
        cm = uni(cm);
        // pack B: step_search(mode, 1, _, aux = false) on {cm, cm - 1, cm + 1} (:974)
        const int lo = !(cm < 3) ? cm - 1 : kNoMode, hi = !(cm + 1 > 66) ? cm + 1 : kNoMode;
        const Pack4Out b = pack4_eval(c, q, 3, cm, lo, hi, overflow);
        int win_ = -1;
        LEAF4_CANDIDATE(b, 0, cm);
        if (lo != kNoMode) LEAF4_CANDIDATE(b, 1, lo);
        if (hi != kNoMode) LEAF4_CANDIDATE(b, 2, hi);
        if (row == win_) rec_put(0, q.tx + x, q.ty + y, b.rec);
        WSYNC();
        PROF_MARK(l4b1_);
        PROF_ADD2(PH_L4 + 3, l4s1_, l4b1_);
    }
    if (served) {
        // [planar, DC] from the server against pack B's first minimum, which wins only if strictly cheaper
        PROF_MARK(sw0_);
        lv_word_wait(&SHT.lvb.job4_done, my_job);
        PROF_MARK(sw1_);
        PROF_ADDM(3, sw0_, sw1_); // (profile build: mem_nop_m3 = what member 3 waits for the server)
        const uint8_t* sv = (const uint8_t*)team_lds(c, 0).decw + kSrv4Byte;
        const float bestB = best;
        const int modeB = best_mode;
        Pack4Out a;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            a.ssd[k2] = (uint32_t)uni((int)((const uint32_t*)(sv + 32))[k2]);
            const unsigned long long lv = ((const unsigned long long*)(sv + 40))[k2];
            a.lvl[k2] = (long long)(((unsigned long long)(unsigned)uni((int)(lv >> 32)) << 32) | (unsigned)uni((int)lv));
        }
        first_ = true;
        int win_ = -1;
        LEAF4_CANDIDATE(a, 0, PLANAR);
        LEAF4_CANDIDATE(a, 1, DC);
        if (bestB < best) {
            best = bestB;
            best_mode = modeB;
        } else {
            if (lane < 16) rec_put(0, q.tx + x, q.ty + y, sv[16 * win_ + lane]);
            WSYNC();
        }
        lv_word_add(&SHT.lvb.job4_ack);
    }
#undef LEAF4_CANDIDATE
    r.vmin = best;
    r.imin = best_mode;
    return r;
}

// The server's side of a 4x4 leaf (level schedule, q.kind K_SERVE4): reference samples and originals of the block from
// member 3's LDS, pack {planar, DC}, the two candidates' reconstructions and parts to decw[448 ..] of this wave.
__device__ __forceinline__ void serve_pack4(const Ctx& c, const Req& q, int* overflow) {
    const int lane = lane_fresh();
    const Lds& own = team_lds(c, 3);
    for (int w = lane; w < 98; w += 64) ((uint32_t*)SH.refs)[w] = ((const uint32_t*)own.refs)[w];
    if (lane < 4) ((uint32_t*)SH.r2)[kOrgLeaf / 4 + lane] = ((const uint32_t*)own.r2)[kOrgLeaf / 4 + lane];
    WSYNC();
    const Pack4Out a = pack4_eval(c, q, 2, PLANAR, DC, kNoMode, overflow);
    uint8_t* sv = (uint8_t*)SH.decw + 448;
    if (lane < 32) sv[lane] = (uint8_t)a.rec;
    if (lane < 2) {
        ((uint32_t*)(sv + 32))[lane] = lane ? a.ssd[1] : a.ssd[0];
        ((long long*)(sv + 40))[lane] = lane ? a.lvl[1] : a.lvl[0];
    }
    WSYNC();
    lv_word_add(&SHT.lvb.job4_done);
}

// K_LEAFC4: the DUAL_TREE_CHROMA leaf of a split 8x8 CU (block_splitter.rs:794-885) in one request.  The three CCLM
// SADs and the pick (:847-854) as in K_CCLMSEARCH, then the picked CCLM mode and the DM mode (q.mc) are evaluated SIDE
// BY SIDE: rows 0 / 1 = Cb / Cr of the CCLM candidate, rows 2 / 3 = Cb / Cr of the DM candidate, four 4x4 blocks
// through the stages of pack4_eval.  The winner's rows (DM on a tie, :857-873) write their reconstruction to the tile.
__device__ __forceinline__ Res leafc4_search(const Ctx& c, const Req& q, int* overflow) {
    Res r;
    r.ssd_y = 0;
    r.ssd_c = 0;
    r.lvl_y = 0;
    r.lvl_c = 0;
    r.v0 = r.v1 = r.v2 = 3.40282347e+38f;
    const int lane = lane_fresh();
    const int dm = q.mc;
    // get_chroma_intra_pred_aux_cost of LT, T, L_CCLM in one sample pass, then the pick (SADs are integers < 2^20)
    CclmParams call_;
    const unsigned acc = sad_list_cclm(c, q.tx, q.ty, 3, &call_);
    const unsigned lt = (unsigned)__builtin_amdgcn_readlane((int)acc, 0), t = (unsigned)__builtin_amdgcn_readlane((int)acc, 1),
                   l = (unsigned)__builtin_amdgcn_readlane((int)acc, 2);
    if (c.trace && lane < 3)
        TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, q.tree, 2, 0, lane == 0 ? LT_CCLM : (lane == 1 ? T_CCLM : L_CCLM),
                  __float_as_int((float)acc));
    const int cm = (lt <= t && lt <= l) ? LT_CCLM : (t <= l ? T_CCLM : L_CCLM);
    if (q.refs1) build_refs(c, 1, q.tx, q.ty, 3);
    PROF_MARK(t0_);
    // model parameters of both planes of the picked mode: derived with the SAD list's
    const CclmPick cpk_ = cclm_pick(call_, 2 * cclm_mode_index(cm));
    const int a0 = cpk_.a0, a1 = cpk_.a1, k0 = cpk_.k0, k1 = cpk_.k1, b0 = cpk_.b0, b1 = cpk_.b1;
    const bool flat128 = cpk_.flat128, avail_l = cpk_.avail_l;
    const int row = lane >> 4, i = lane & 15, x = i & 3, y = i >> 2;
    const int pl = row & 1;
    int v = predict4_lane(c, row >= 2 ? dm : kNoMode, pl); // rows 2, 3: the DM candidate (every lane passes the WSYNC inside)
    if (row < 2) {
        v = 128;
        if (!flat128) {
            const int ds = cclm_ds6(c, q.tx, q.ty, 2 * y, 2 * x, avail_l);
            v = (M24(ds, pl ? a1 : a0) >> (pl ? k1 : k0)) + (pl ? b1 : b0);
            v = min(max(v, 0), 255);
        }
    }
    const int org = ((const uint8_t*)SH.r2)[kOrgLeaf + 256 + 16 * pl + i];
    SH.r1[lane] = (int16_t)(org - v);
    WSYNC();
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, 2, 4, 0);
    long long lvl[4];
    int any_mask = 0;
    quantize_p16(c, 4, overflow, lvl, &any_mask);
    if (any_mask) {
        dequantize_t(c, 2, 4, 0);
        inv_dct_lg(c, 2, 4, 0);
    }
    int rec = (int16_t)(v + (int)SH.r1[lane]);
    rec = min(max(rec, 0), 255);
    const int d = rec - org;
    const int rs = row_sum_i32(M24(d, d));
    EvalParts ec, ed;
    ec.ssd_y = ed.ssd_y = 0;
    ec.lvl_y = ed.lvl_y = 0;
    ec.ssd_c = (uint32_t)(__builtin_amdgcn_readlane(rs, 0) + __builtin_amdgcn_readlane(rs, 16));
    ed.ssd_c = (uint32_t)(__builtin_amdgcn_readlane(rs, 32) + __builtin_amdgcn_readlane(rs, 48));
    ec.lvl_c = lvl[0] + lvl[1];
    ed.lvl_c = lvl[2] + lvl[3];
    const float c0 = uni_f(assemble_chroma_cost(c, cm, ec));
    const float dm_cost = uni_f(assemble_chroma_cost(c, dm, ed));
    if (c.trace && lane == 0) {
        TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, q.tree, 3, 0, cm, __float_as_int(c0));
        TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, q.tree, 3, 0, dm, __float_as_int(dm_cost));
    }
    const float cost = fminf(c0, fminf(dm_cost, 3.40282347e+38f));
    const bool dm_wins = dm_cost == cost;
    if ((row >= 2) == dm_wins) rec_put(1 + pl, (q.tx >> 1) + x, (q.ty >> 1) + y, rec);
    WSYNC();
    r.vmin = cost;
    r.imin = dm_wins ? dm : cm;
    return r;
}

// ---------------------------------------------------------------------------
// K_LEAF8: the whole search of an 8x8 SINGLE_TREE leaf in one request (block_splitter.rs:886-1078 with a luma 8x8 block
// and a Cb + Cr 4x4 pair per candidate).  At max-split-depth 2 these leaves are two thirds of a CTU's evaluations and
// each one was a chain of short stages between control steps: 64 + 32 samples per candidate, a pooled 64-step trellis
// walk per candidate between two workgroup barriers.  The full candidates of a leaf read only neighbours outside the
// block (:887-898, :974), so they are evaluated in PACKS: pack A = {planar, DC}, then the SAD search, then pack B =
// {cm, cm - 1, cm + 1}; a pack's candidates go through every stage together -- luma blocks one pass each, the 4x4
// chroma blocks of all candidates four to a pass (predict4_lane), the transforms over nb blocks, and ONE trellis pass
// for the pack's six or nine chains walked side by side by this wave alone (quantize_pk: no workgroup barrier).
// Predictions and reconstructions of the pack are parked in LDS (PRED_PARK); the running best candidate's
// reconstruction goes to the tile when its pack is done, so there is no save / restore through global scratch.
// Then the CCLM part (:1040-1072) as in K_CCLMSEARCH, the CCLM candidate's prediction and reconstruction in
// registers: the tile keeps the DM chroma unless CCLM wins.  The decisions are the reference's, in its order: first
// minimum of [planar, DC, cm, cm - 1, cm + 1] as a running strict-less update; DM on a tie with CCLM.
// ---------------------------------------------------------------------------
struct Pack8Out {
    uint32_t ssd_y[3], ssd_c[3];
    long long lvl_y[3], lvl_c[3];
};
__device__ __forceinline__ Pack8Out pack8_eval(const Ctx& c, const Req& q, int nc, int m0, int m1, int m2, int* overflow) {
    Pack8Out o;
    const int lane = lane_fresh();
    const int row = lane >> 4, i16 = lane & 15;
    const int nL = 64 * nc;
    uint8_t* park = (uint8_t*)SH.decw + kParkByte;
    const uint8_t* org = (const uint8_t*)SH.r2 + kOrgLeaf;
    PROF_MARK(t0_);
    // luma: one pass per candidate (a candidate outside 2..66 rides along as a zero block)
#pragma unroll 1
    for (int cd = 0; cd < nc; ++cd) {
        const int mode = cd == 0 ? m0 : (cd == 1 ? m1 : m2);
        if (mode != kNoMode) {
            predict<true>(c, 0, q.tx, q.ty, 3, mode, 64 * cd, PRED_PARK);
        } else {
            SH.r1[64 * cd + lane] = 0;
            park[64 * cd + lane] = 0;
        }
    }
    // chroma: the 4x4 blocks of all candidates, four to a pass (row = block 2 cand + plane)
#pragma unroll 1
    for (int ps = 0; ps < 2; ++ps) {
        if (4 * ps >= 2 * nc) continue;
        const int blk = 4 * ps + row;
        const int cd = blk >> 1, pl = blk & 1;
        const int mode = cd == 0 ? m0 : (cd == 1 ? m1 : m2);
        const bool in = blk < 2 * nc;
        const bool on = in && mode != kNoMode;
        const int v = predict4_lane(c, on ? mode : kNoMode, pl);
        if (in) {
            SH.r1[nL + 16 * blk + i16] = (int16_t)(on ? (int)org[256 + 16 * pl + i16] - v : 0);
            park[nL + 16 * blk + i16] = (uint8_t)v;
        }
        WSYNC();
    }
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, 3, nc, 0);
    fwd_dct_lg(c, 2, 2 * nc, nL);
    PROF_MARK(t2_);
    PROF_ADD2(PH_FDCT, t1_, t2_);
    bool any_y = false, any_c = false;
    quantize_pk<3>(c, nc, overflow, o.lvl_y, o.lvl_c, &any_y, &any_c);
    PROF_MARK(t3_);
    if (any_y) { // (all levels zero: the residuals are zero too, and r1 already says so)
        dequantize_t(c, 3, nc, 0);
        inv_dct_lg(c, 3, nc, 0);
    }
    if (any_c) {
        dequantize_t(c, 2, 2 * nc, nL);
        inv_dct_lg(c, 2, 2 * nc, nL);
    }
    PROF_MARK(t4_);
    PROF_ADD2(PH_IDCT, t3_, t4_);
    // reconstruction (pred as i16 + res, clamp: :178) and SSD; the reconstruction takes the prediction's place
#pragma unroll
    for (int cd = 0; cd < 3; ++cd) {
        o.ssd_y[cd] = 0;
        o.ssd_c[cd] = 0;
        if (cd < nc) {
            int rec = (int16_t)((int)park[64 * cd + lane] + (int)SH.r1[64 * cd + lane]);
            rec = min(max(rec, 0), 255);
            park[64 * cd + lane] = (uint8_t)rec;
            const int d = rec - (int)org[lane];
            o.ssd_y[cd] = (uint32_t)wave_sum_i32(M24(d, d));
        }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (4 * ps >= 2 * nc) continue;
        const int blk = 4 * ps + row;
        const bool in = blk < 2 * nc;
        int dd = 0;
        if (in) {
            int rec = (int16_t)((int)park[nL + 16 * blk + i16] + (int)SH.r1[nL + 16 * blk + i16]);
            rec = min(max(rec, 0), 255);
            park[nL + 16 * blk + i16] = (uint8_t)rec;
            const int d = rec - (int)org[256 + 16 * (blk & 1) + i16];
            dd = M24(d, d);
        }
        const int rs = row_sum_i32(dd);
        const uint32_t s01 = (uint32_t)(__builtin_amdgcn_readlane(rs, 0) + __builtin_amdgcn_readlane(rs, 16));
        const uint32_t s23 = (uint32_t)(__builtin_amdgcn_readlane(rs, 32) + __builtin_amdgcn_readlane(rs, 48));
        if (ps == 0) {
            o.ssd_c[0] = s01;
            o.ssd_c[1] = s23;
        } else {
            o.ssd_c[2] = s01;
        }
    }
    WSYNC();
    PROF_MARK(t5_);
    PROF_ADD2(PH_RECON, t4_, t5_);
    return o;
}

// a pack candidate's reconstruction from the park into the tile (luma 8x8, Cb and Cr 4x4)
__device__ __forceinline__ void pack8_to_tile(const Req& q, int nc, int cd) {
    const int lane = lane_fresh();
    const uint8_t* park = (const uint8_t*)SH.decw + kParkByte;
    rec_put(0, q.tx + (lane & 7), q.ty + (lane >> 3), park[64 * cd + lane]);
    if (lane < 32) {
        const int pl = lane >> 4, i = lane & 15;
        rec_put(1 + pl, (q.tx >> 1) + (i & 3), (q.ty >> 1) + (i >> 2), park[64 * nc + 32 * cd + lane]);
    }
    WSYNC();
}

__device__ __forceinline__ Res leaf8_search(const Ctx& c, const Req& q, int* overflow) {
    Res r;
    r.ssd_y = 0;
    r.ssd_c = 0;
    r.lvl_y = 0;
    r.lvl_c = 0;
    r.v0 = r.v1 = r.v2 = 3.40282347e+38f;
    r.imin = PLANAR;
    r.imin2 = PLANAR;
    const int lane = lane_fresh();
    if (q.refs0) build_refs(c, 0, q.tx, q.ty, 3);
    if (q.refs1) build_refs(c, 1, q.tx, q.ty, 3);
    float best = 3.40282347e+38f;
    int best_mode = q.ml, best_cls = 0;
    EvalParts eb;
    eb.ssd_y = eb.ssd_c = 0;
    eb.lvl_y = eb.lvl_c = 0;
    bool first_ = true;
    const MpmList mpl_ = mpm_list(c, q.tx, q.ty, 3);
#define LEAF8_CANDIDATE(P, B, M)                                                                                       \
    do {                                                                                                               \
        EvalParts e_;                                                                                                  \
        e_.ssd_y = (P).ssd_y[B];                                                                                       \
        e_.ssd_c = (P).ssd_c[B];                                                                                       \
        e_.lvl_y = (P).lvl_y[B];                                                                                       \
        e_.lvl_c = (P).lvl_c[B];                                                                                       \
        const int cls_ = mpm_class_of(mpl_, (M));                                                                      \
        const float val_ = uni_f(assemble_cost(c, TREE_SINGLE, cls_, (M), e_));                                        \
        if (c.trace && lane == 0) /* (team schedule: each part is traced by the member that runs it) */                \
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, TREE_SINGLE, 1, (M), (M), __float_as_int(val_));              \
        if (first_ || val_ < best) {                                                                                   \
            best = val_;                                                                                               \
            best_mode = (M);                                                                                           \
            best_cls = cls_;                                                                                           \
            eb = e_;                                                                                                   \
            win_ = (B);                                                                                                \
        }                                                                                                              \
        first_ = false;                                                                                                \
    } while (0)
    // Level schedule (team kernel): member 0, idle once the CTU's 32x32 candidate is done, SERVES pack A of member 2's 8x8
    // leaves.  q.n & 16: this is the server's request -- reference samples and originals of the block from member 2's
    // LDS, pack {planar, DC}, the candidates' parts to xr[0 .. 1], their reconstructions stay in this wave's park.
    if (q.n & 16) {
        const Lds& own = team_lds(c, 2);
        for (int w = lane; w < 98; w += 64) ((uint32_t*)SH.refs)[w] = ((const uint32_t*)own.refs)[w];
        for (int w = lane; w < 96; w += 64) ((uint32_t*)SH.r2)[kOrgLeaf / 4 + w] = ((const uint32_t*)own.r2)[kOrgLeaf / 4 + w];
        WSYNC();
        const Pack8Out a = pack8_eval(c, q, 2, PLANAR, DC, kNoMode, overflow);
        if (lane < 2) {
            XRes x;
            x.ssd_y = lane ? a.ssd_y[1] : a.ssd_y[0];
            x.ssd_c = lane ? a.ssd_c[1] : a.ssd_c[0];
            x.lvl_y = lane ? a.lvl_y[1] : a.lvl_y[0];
            x.lvl_c = lane ? a.lvl_c[1] : a.lvl_c[0];
            SH.xr[lane] = x;
        }
        WSYNC();
        lv_word_add(&SHT.lvb.job_done);
        return r;
    }
    // member 2 of a team in the level schedule, and the server is polling: post the job (the block's reference samples
    // and originals are ready and stay untouched until the leaf is decided), skip pack A here, merge its results below
    const bool served = WRENC_SERVER && c.solo && c.member == 2 && (q.n & 7) == 7 && lv_word(&SHT.lvb.srv_ready) != 0;
    unsigned my_job = 0;
    if (served) {
        if (lane == 0) {
            SHT.lvb.job_bx = (uint8_t)q.tx;
            SHT.lvb.job_by = (uint8_t)q.ty;
        }
        my_job = uni((int)lv_word(&SHT.lvb.job_posted)) + 1u;
        lv_word_add(&SHT.lvb.job_posted);
    } else if (q.n & 1) {
        // pack A: planar and DC (:887-898)
        PROF_MARK(la0_);
        const Pack8Out a = pack8_eval(c, q, 2, PLANAR, DC, kNoMode, overflow);
        int win_ = -1;
        LEAF8_CANDIDATE(a, 0, PLANAR);
        LEAF8_CANDIDATE(a, 1, DC);
        pack8_to_tile(q, 2, win_);
        PROF_MARK(la1_);
        PROF_ADD2(PH_LEAF + 0, la0_, la1_);
    }
    if (q.n & 2) {
        int cm;
        unsigned smin;
        PROF_MARK(ls0_);
        sad_search(c, q, cm, smin);
        PROF_MARK(ls1_);
        PROF_ADD2(PH_LEAF + 1, ls0_, ls1_);
        cm = uni(cm);
        // pack B: step_search(mode, 1, _, aux = false) on {cm, cm - 1, cm + 1} (:974)
        const int lo = !(cm < 3) ? cm - 1 : kNoMode, hi = !(cm + 1 > 66) ? cm + 1 : kNoMode;
        const Pack8Out b = pack8_eval(c, q, 3, cm, lo, hi, overflow);
        int win_ = -1;
        LEAF8_CANDIDATE(b, 0, cm);
        if (lo != kNoMode) LEAF8_CANDIDATE(b, 1, lo);
        if (hi != kNoMode) LEAF8_CANDIDATE(b, 2, hi);
        if (win_ >= 0) pack8_to_tile(q, 3, win_);
        PROF_MARK(lb1_);
        PROF_ADD2(PH_LEAF + 2, ls1_, lb1_);
    }
    if (served) {
        // the server's pack A: first minimum of [planar, DC] against the first minimum of pack B found above; the
        // reference's order is [planar, DC, cm, cm - 1, cm + 1], so pack B's best wins only if strictly cheaper
        PROF_MARK(sw0_);
        lv_word_wait(&SHT.lvb.job_done, my_job);
        PROF_MARK(sw1_);
        PROF_ADDM(3, sw0_, sw1_); // (mem_nop_m2)
        const Lds& srv = team_lds(c, 0);
        const float bestB = best;
        const int modeB = best_mode, clsB = best_cls;
        const EvalParts ebB = eb;
        first_ = true;
        int win_ = -1;
        Pack8Out a;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const XRes& x = srv.xr[k2];
            a.ssd_y[k2] = (uint32_t)uni((int)x.ssd_y);
            a.ssd_c[k2] = (uint32_t)uni((int)x.ssd_c);
            const unsigned long long ly = (unsigned long long)x.lvl_y, lc = (unsigned long long)x.lvl_c;
            a.lvl_y[k2] = (long long)(((unsigned long long)(unsigned)uni((int)(ly >> 32)) << 32) | (unsigned)uni((int)ly));
            a.lvl_c[k2] = (long long)(((unsigned long long)(unsigned)uni((int)(lc >> 32)) << 32) | (unsigned)uni((int)lc));
        }
        LEAF8_CANDIDATE(a, 0, PLANAR);
        LEAF8_CANDIDATE(a, 1, DC);
        if (bestB < best) { // pack B's best stays (its reconstruction is in the tile already)
            best = bestB;
            best_mode = modeB;
            best_cls = clsB;
            eb = ebB;
        } else {            // a candidate of pack A: its reconstruction from the server's park
            const uint8_t* park = (const uint8_t*)srv.decw + kParkByte;
            rec_put(0, q.tx + (lane & 7), q.ty + (lane >> 3), park[64 * win_ + lane]);
            if (lane < 32) {
                const int pl2 = lane >> 4, i2 = lane & 15;
                rec_put(1 + pl2, (q.tx >> 1) + (i2 & 3), (q.ty >> 1) + (i2 >> 2), park[128 + 32 * win_ + lane]);
            }
            WSYNC();
        }
        lv_word_add(&SHT.lvb.job_ack);
    }
#undef LEAF8_CANDIDATE
    r.vmin = best;
    r.imin = best_mode;
    r.imin2 = best_mode;
    r.ssd_y = eb.ssd_y;
    r.ssd_c = eb.ssd_c;
    r.lvl_y = eb.lvl_y;
    r.lvl_c = eb.lvl_c;
    if (!(q.n & 4)) return r;
    // ---- the CCLM part on the winner, whose reconstruction is in the tile (:1040-1072) ----
    float cur;
    if (q.n & 3) {
        // :1040 get_chroma_intra_pred_cost(mode) repeats the winner's chroma evaluation: its parts are at hand
        cur = uni_f(assemble_chroma_cost(c, best_mode, eb));
        if (c.trace && lane == 0)
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, TREE_SINGLE, 3, 0, best_mode, __float_as_int(cur));
    } else {
        cur = q.fcur; // (team schedule: the team decided the winner, every member assembled this cost)
    }
    PROF_MARK(lc0_);
    CclmParams call_;
    const unsigned acc = sad_list_cclm(c, q.tx, q.ty, 3, &call_);
    const unsigned lt = (unsigned)__builtin_amdgcn_readlane((int)acc, 0), t = (unsigned)__builtin_amdgcn_readlane((int)acc, 1),
                   l = (unsigned)__builtin_amdgcn_readlane((int)acc, 2);
    if (c.trace && lane < 3)
        TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, TREE_SINGLE, 2, 0, lane == 0 ? LT_CCLM : (lane == 1 ? T_CCLM : L_CCLM),
                  __float_as_int((float)acc));
    const int cm = (lt <= t && lt <= l) ? LT_CCLM : (t <= l ? T_CCLM : L_CCLM);
    PROF_MARK(t0_);
    PROF_ADD2(PH_LEAF + 11, lc0_, t0_);
    const CclmPick cpk_ = cclm_pick(call_, 2 * cclm_mode_index(cm)); // (derived with the SAD list's)
    const int a0 = cpk_.a0, a1 = cpk_.a1, k0 = cpk_.k0, k1 = cpk_.k1, b0 = cpk_.b0, b1 = cpk_.b1;
    const bool flat128 = cpk_.flat128, avail_l = cpk_.avail_l;
    const int row = lane >> 4, i = lane & 15, x = i & 3, y = i >> 2;
    const int pl = row & 1;
    const bool mine = row < 2; // rows 0 / 1 = Cb / Cr
    int v = 128;
    if (mine && !flat128) {
        const int ds = cclm_ds6(c, q.tx, q.ty, 2 * y, 2 * x, avail_l);
        v = (M24(ds, pl ? a1 : a0) >> (pl ? k1 : k0)) + (pl ? b1 : b0);
        v = min(max(v, 0), 255);
    }
    const int org = ((const uint8_t*)SH.r2)[kOrgLeaf + 256 + 16 * pl + i];
    if (mine) SH.r1[lane] = (int16_t)(org - v);
    WSYNC();
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, 2, 2, 0);
    long long lvl[4];
    int any_mask = 0;
    quantize_p16(c, 2, overflow, lvl, &any_mask);
    if (any_mask) {
        dequantize_t(c, 2, 2, 0);
        inv_dct_lg(c, 2, 2, 0);
    }
    int rec = (int16_t)(v + (int)SH.r1[mine ? lane : 0]);
    rec = min(max(rec, 0), 255);
    const int d = rec - org;
    const int rs = row_sum_i32(mine ? M24(d, d) : 0);
    EvalParts e = eb;
    e.ssd_c = (uint32_t)(__builtin_amdgcn_readlane(rs, 0) + __builtin_amdgcn_readlane(rs, 16));
    e.lvl_c = lvl[0] + lvl[1];
    const float cclm_cost = uni_f(assemble_chroma_cost(c, cm, e));
    if (c.trace && lane == 0) TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 3, TREE_SINGLE, 3, 0, cm, __float_as_int(cclm_cost));
    const bool dm_wins = cur == fminf(cclm_cost, fminf(cur, 3.40282347e+38f));
    // :1062-1072 final get_intra_pred_cost: the winner's luma with the DM chroma (still in the tile) or the CCLM chroma
    if (!dm_wins && mine) rec_put(1 + pl, (q.tx >> 1) + x, (q.ty >> 1) + y, rec);
    WSYNC();
    PROF_MARK(lc9_);
    PROF_ADD2(PH_LEAF + 3, lc0_, lc9_);
    if (q.n & 3) {
        r.vmin = dm_wins ? uni_f(assemble_cost(c, TREE_SINGLE, best_cls, best_mode, eb))
                         : uni_f(assemble_cost(c, TREE_SINGLE, best_cls, cm, e));
        r.imin2 = dm_wins ? best_mode : cm;
    } else { // part 4 alone: the CCLM candidate's mode and chroma parts, the team assembles the rest
        r.imin = cm;
        r.ssd_c = e.ssd_c;
        r.lvl_c = e.lvl_c;
    }
    return r;
}

// ---------------------------------------------------------------------------
// K_LEAF16: the full candidates of a 16x16 SINGLE_TREE leaf in one request (block_splitter.rs:886-1037), in packs of
// TWO: {planar, DC}, the SAD search, {cm, cm - 1}, {cm + 1}.  Two candidates' residuals (2 x (256 + 2 x 64) i16) are
// what r1 holds; a pack goes through prediction, transforms and reconstruction block after block as the single
// evaluation does, and through ONE trellis pass for its six chains (quantize_pk<4>: the two luma chains side by side,
// the four chroma chains riding along, this wave alone, no workgroup barrier) -- three walks of 256 steps per leaf
// instead of five pooled ones, and one control step instead of six.  The two candidates' predictions (768 B) are parked in
// the three corners of LDS the pack leaves free (PRED_PARK16, dev_predict.h); the running best candidate's reconstruction
// goes from there to the tile when its pack is done.  The CCLM part follows as its own request (leaf_step, C_WINNER): its DM-chroma
// restore takes the winner from slot 0, where the request after this one saves it.
// ---------------------------------------------------------------------------
struct Pack16Out {
    uint32_t ssd_y[2], ssd_c[2];
    long long lvl_y[3], lvl_c[3];
};
__device__ __forceinline__ Pack16Out pack16_eval(const Ctx& c, const Req& q, int nc, int m0, int m1, int* overflow) {
    Pack16Out o;
    const int lane = lane_fresh();
    const int nL = 256 * nc;
    const uint8_t* org = (const uint8_t*)SH.r2 + kOrgLeaf; // luma 256 | Cb 64 | Cr 64
    PROF_MARK(t0_);
#pragma unroll 1
    for (int cd = 0; cd < nc; ++cd) {
        const int mode = cd == 0 ? m0 : m1;
        if (mode != kNoMode) {
            predict<true>(c, 0, q.tx, q.ty, 4, mode, 256 * cd, PRED_PARK16, nL);
            predict<true>(c, 1, q.tx, q.ty, 4, mode, nL + 128 * cd, PRED_PARK16, nL);
        } else { // a candidate outside 2..66 rides along as a zero block
            for (int i = lane; i < 256; i += 64) {
                SH.r1[256 * cd + i] = 0;
                *park16(256 * cd + i, nL) = 0;
            }
            for (int i = lane; i < 128; i += 64) {
                SH.r1[nL + 128 * cd + i] = 0;
                *park16(nL + 128 * cd + i, nL) = 0;
            }
            WSYNC();
        }
    }
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, 4, nc, 0);
    fwd_dct_lg(c, 3, 2 * nc, nL);
    PROF_MARK(t2_);
    PROF_ADD2(PH_FDCT, t1_, t2_);
    bool any_y = false, any_c = false;
    quantize_pk<4>(c, nc, overflow, o.lvl_y, o.lvl_c, &any_y, &any_c);
    PROF_MARK(t3_);
    if (any_y) {
        dequantize_t(c, 4, nc, 0);
        inv_dct_lg(c, 4, nc, 0);
    }
    if (any_c) {
        dequantize_t(c, 3, 2 * nc, nL);
        inv_dct_lg(c, 3, 2 * nc, nL);
    }
    PROF_MARK(t4_);
    PROF_ADD2(PH_IDCT, t3_, t4_);
#pragma unroll
    for (int cd = 0; cd < 2; ++cd) {
        o.ssd_y[cd] = 0;
        o.ssd_c[cd] = 0;
        if (cd < nc) {
            int py = 0, pc = 0;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int i = lane + 64 * kq;
                uint8_t* pk = park16(256 * cd + i, nL);
                int rec = (int16_t)((int)*pk + (int)SH.r1[256 * cd + i]); // pred as i16 + res, clamp (:178)
                rec = min(max(rec, 0), 255);
                *pk = (uint8_t)rec;
                const int d = rec - (int)org[i];
                py += M24(d, d);
            }
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                const int i = lane + 64 * kq;
                uint8_t* pk = park16(nL + 128 * cd + i, nL);
                int rec = (int16_t)((int)*pk + (int)SH.r1[nL + 128 * cd + i]);
                rec = min(max(rec, 0), 255);
                *pk = (uint8_t)rec;
                const int d = rec - (int)org[256 + i];
                pc += M24(d, d);
            }
            o.ssd_y[cd] = (uint32_t)wave_sum_i32(py);
            o.ssd_c[cd] = (uint32_t)wave_sum_i32(pc);
        }
    }
    WSYNC();
    PROF_MARK(t5_);
    PROF_ADD2(PH_RECON, t4_, t5_);
    return o;
}

// a pack candidate's reconstruction from its park (park16) into the tile (luma 16x16, Cb and Cr 8x8)
__device__ __forceinline__ void pack16_to_tile(const Ctx& c, const Req& q, int nc, int cd) {
    const int lane = lane_fresh();
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int i = lane + 64 * kq;
        rec_put(0, q.tx + (i & 15), q.ty + (i >> 4), *park16(256 * cd + i, 256 * nc));
    }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
        const int i = lane + 64 * kq;
        rec_put(1 + (i >> 6), (q.tx >> 1) + (i & 7), (q.ty >> 1) + ((i & 63) >> 3), *park16(256 * nc + 128 * cd + i, 256 * nc));
    }
    WSYNC();
}

__device__ __forceinline__ Res leaf16_search(const Ctx& c, const Req& q, int* overflow) {
    Res r;
    r.v0 = r.v1 = r.v2 = 3.40282347e+38f;
    const int lane = lane_fresh();
    if (q.refs0) build_refs(c, 0, q.tx, q.ty, 4);
    if (q.refs1) build_refs(c, 1, q.tx, q.ty, 4);
    float best = 3.40282347e+38f;
    int best_mode = PLANAR;
    EvalParts eb;
    eb.ssd_y = eb.ssd_c = 0;
    eb.lvl_y = eb.lvl_c = 0;
    bool first_ = true;
    const MpmList mpl_ = mpm_list(c, q.tx, q.ty, 4);
#define LEAF16_CANDIDATE(P, B, M)                                                                                      \
    do {                                                                                                               \
        EvalParts e_;                                                                                                  \
        e_.ssd_y = (P).ssd_y[B];                                                                                       \
        e_.ssd_c = (P).ssd_c[B];                                                                                       \
        e_.lvl_y = (P).lvl_y[B];                                                                                       \
        e_.lvl_c = (P).lvl_c[B];                                                                                       \
        const int cls_ = mpm_class_of(mpl_, (M));                                                                      \
        const float val_ = uni_f(assemble_cost(c, TREE_SINGLE, cls_, (M), e_));                                        \
        if (c.trace && lane == 0)                                                                                      \
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, 4, TREE_SINGLE, 1, (M), (M), __float_as_int(val_));              \
        if (first_ || val_ < best) {                                                                                   \
            best = val_;                                                                                               \
            best_mode = (M);                                                                                           \
            eb = e_;                                                                                                   \
            win_ = (B);                                                                                                \
        }                                                                                                              \
        first_ = false;                                                                                                \
    } while (0)
    {
        // planar and DC (:887-898)
        PROF_MARK(la0_);
        const Pack16Out a = pack16_eval(c, q, 2, PLANAR, DC, overflow);
        int win_ = -1;
        LEAF16_CANDIDATE(a, 0, PLANAR);
        LEAF16_CANDIDATE(a, 1, DC);
        pack16_to_tile(c, q, 2, win_);
        PROF_MARK(la1_);
        PROF_ADD2(PH_LEAF + 4, la0_, la1_);
    }
    int cm;
    unsigned smin;
    PROF_MARK(ls0_);
    sad_search(c, q, cm, smin);
    PROF_MARK(ls1_);
    PROF_ADD2(PH_LEAF + 5, ls0_, ls1_);
    cm = uni(cm);
    // step_search(mode, 1, _, aux = false) on {cm, cm - 1, cm + 1} (:974): {cm, cm - 1}, then cm + 1
    const int lo = !(cm < 3) ? cm - 1 : kNoMode, hi = !(cm + 1 > 66) ? cm + 1 : kNoMode;
    {
        const Pack16Out b = pack16_eval(c, q, 2, cm, lo, overflow);
        int win_ = -1;
        LEAF16_CANDIDATE(b, 0, cm);
        if (lo != kNoMode) LEAF16_CANDIDATE(b, 1, lo);
        if (win_ >= 0) pack16_to_tile(c, q, 2, win_);
    }
    PROF_MARK(lb1_);
    PROF_ADD2(PH_LEAF + 6, ls1_, lb1_);
    if (hi != kNoMode) {
        const Pack16Out b = pack16_eval(c, q, 1, hi, kNoMode, overflow);
        int win_ = -1;
        LEAF16_CANDIDATE(b, 0, hi);
        if (win_ >= 0) pack16_to_tile(c, q, 1, win_);
    }
    PROF_MARK(lc1_);
    PROF_ADD2(PH_LEAF + 7, lb1_, lc1_);
#undef LEAF16_CANDIDATE
    r.vmin = best;
    r.imin = best_mode;
    r.imin2 = best_mode;
    r.ssd_y = eb.ssd_y;
    r.ssd_c = eb.ssd_c;
    r.lvl_y = eb.lvl_y;
    r.lvl_c = eb.lvl_c;
    return r;
}

// the decision maps of a block: at most 8 x 8 units of 4x4 (one lane each), sizes are powers of two
__device__ __forceinline__ void fill_maps(int bx, int by, int lg, int luma_mode, int chroma_mode, bool luma,
                                          bool chroma) {
    const int l4 = lg - 2; // log2 of the block's side in 4x4 units
    if (luma && LANE < (1 << (2 * l4))) {
        const int idx = ((by >> 2) + (LANE >> l4)) * 8 + (bx >> 2) + (LANE & ((1 << l4) - 1));
        SH.cu_log2[idx] = (uint8_t)lg;
        SH.luma_mode[idx] = (uint8_t)luma_mode;
    }
    if (chroma) {
        const int l8 = max(l4 - 1, 0);
        if (LANE < (1 << (2 * l8)))
            SH.chroma_mode[((by >> 3) + (LANE >> l8)) * 4 + (bx >> 3) + (LANE & ((1 << l8) - 1))] = (uint8_t)chroma_mode;
    }
    WSYNC();
}


// K_SPLIT8: the split candidate of an 8x8 CU (ctu.rs:1990-2063: four DUAL_TREE_LUMA 4x4 CUs, then the DUAL_TREE_CHROMA
// CU) in ONE request: leaf4_search four times, leafc4_search once, with what the tree walk did between them -- the
// leaf's originals staged, the decision maps filled (the chroma leaf's DM mode is the luma mode of the 4x4 covering
// the parent's centre, block_splitter.rs:795-805), the costs summed in z-order in f32 from 0.0 (:1116-1123).  Five control
// steps of 3-6 k cycles each fewer per 8x8 node, 80 per CTU at max-split-depth 3.
// (Only the kernels built for max-split-depth 3 contain it, D3 below: inlined into the one evaluator of a kernel that also
// serves depth 2, which never splits an 8x8, it cost that depth 2.5 %; as an out-of-line function it cost both depths more.)
__device__ __forceinline__ Res split8_search(const Ctx& c, const Req& q, int* overflow) {
    Res r;
    r.ssd_y = 0;
    r.ssd_c = 0;
    r.lvl_y = 0;
    r.lvl_c = 0;
    r.v0 = r.v1 = r.v2 = 3.40282347e+38f;
    r.imin = 0;
    r.imin2 = 0;
    float split8 = 0.0f;
#pragma unroll 1
    for (int i8 = 0; i8 < 4; ++i8) {
        Req ql = {}; // (only what leaf4_search reads: a copy of q would keep thirty scalars alive across the four searches)
        ql.kind = K_LEAF4;
        ql.comps = 1;
        ql.tx = q.tx + (i8 & 1) * 4;
        ql.ty = q.ty + (i8 >> 1) * 4;
        ql.tlg = 2;
        ql.refs0 = true;
        ql.refs1 = false;
        ql.n = 3;
        ql.tree = TREE_DUAL_LUMA;
        PROF_MARK(l4g0_);
        stage_org_leaf(c, 1, ql.tx, ql.ty, 2);
        PROF_MARK(l4g1_);
        PROF_ADD2(PH_L4 + 0, l4g0_, l4g1_);
        const Res rl = leaf4_search(c, ql, overflow);
        fill_maps(ql.tx, ql.ty, 2, rl.imin, 0, true, false);
        split8 = uni_f(split8 + rl.vmin);
    }
    Req qc = {};
    qc.kind = K_LEAFC4;
    qc.tx = q.tx;
    qc.ty = q.ty;
    qc.comps = 2;
    qc.tlg = 3;
    qc.ml = 0;
    qc.mc = uni((int)SH.luma_mode[((q.ty + 4) >> 2) * 8 + ((q.tx + 4) >> 2)]); // DM = the luma mode at the parent's centre
    qc.refs0 = false;
    qc.refs1 = true;
    qc.tree = TREE_DUAL_CHROMA;
    PROF_MARK(l4c0_);
    stage_org_leaf(c, 2, q.tx, q.ty, 3);
    const Res rc = leafc4_search(c, qc, overflow);
    fill_maps(q.tx, q.ty, 3, 0, rc.imin, false, true);
    PROF_MARK(l4c1_);
    PROF_ADD2(PH_L4 + 4, l4c0_, l4c1_);
    r.vmin = uni_f(split8 + rc.vmin);
    return r;
}

// ---------------------------------------------------------------------------
// Search control as state machines: a step function runs until it needs a block evaluated, stores
// the request and where to continue, and returns true; the driver evaluates the block and calls
// it again with the result.  All state lives in LDS (CtuSt / LeafSt, wave-uniform); the control
// flow is a plain loop around a switch (reducible, all scalar branches).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void req_full(Req& q, int comps, int tx, int ty, int tlg, int ml, int mc, bool shared,
                                         bool active, bool refs0, bool refs1, bool final) {
    q.kind = K_FULL;
    q.comps = comps;
    q.tx = tx;
    q.ty = ty;
    q.tlg = tlg;
    q.ml = ml;
    q.mc = mc;
    q.shared = shared;
    q.active = active;
    q.refs0 = refs0;
    q.refs1 = refs1;
    q.final = final;
    q.stage = (final && tlg <= 4) ? comps : 0; // a final-pass block is evaluated once: stage its originals now
    q.pre_copy = COPY_NONE;
    q.xchg = false;
}

__device__ __forceinline__ void req_copy(Req& q, int mode, int comps, int slot, int tx, int ty, int tlg) {
    q.pre_copy = mode;
    q.copy_comps = comps;
    q.copy_slot = slot;
    q.copy_tx = tx;
    q.copy_ty = ty;
    q.copy_tlg = tlg;
    q.copy_from = -1;
}

enum {
    C_START = 0, C_PLANAR, C_DCM, C_LIST, C_PAIR_EMIT, C_PAIR, C_F0, C_F1, C_F2, C_WIN, C_CX, C_CCLM, C_DM,
    C_DC_START, C_DC2, C_DC3, C_DC4, C_DC5, C_L4, C_LC4, C_L8, C_L16, C_WINNER
};

// slot: where the search keeps its best candidate's reconstruction (0 = the leaf's own slot; 1 + level when the wave
// schedule goes on to test the block's split: the slot then already holds the unsplit candidate, see ctu_step)
__device__ __forceinline__ void leaf_init(LeafSt& s, int tree, int bx, int by, int lg, int dm_mode, int slot = 0) {
    s.cont = (uint8_t)(tree == TREE_DUAL_CHROMA ? C_DC_START : C_START);
    s.tree = (uint8_t)tree;
    s.bx = (uint8_t)bx;
    s.by = (uint8_t)by;
    s.lg = (uint8_t)lg;
    s.dm_mode = (uint8_t)dm_mode;
    s.need_refs0 = 1;
    s.need_refs1 = 1;
    s.need_org = lg <= 4 ? 1 : 0;
    s.need_save = 0;
    s.tile_best = 0;
    s.slot = (uint8_t)slot;
}

// a new best candidate's reconstruction is saved to slot 0 by the request that follows it (before
// anything overwrites the tile)
__device__ __forceinline__ void leaf_attach_save(LeafSF& s, Req& q) {
    q.pre_copy = COPY_NONE;
    if (s.need_save) {
        req_copy(q, COPY_SAVE, s.tree == TREE_SINGLE ? 3 : 1, s.slot, s.bx, s.by, s.lg);
        s.need_save = 0;
    }
}
// the first evaluation of a leaf stages the originals of all the leaf's components (blocks <= 16x16)
__device__ __forceinline__ void leaf_attach_org(LeafSF& s, Req& q) {
    q.stage = 0;
    if (s.need_org) {
        q.stage = s.tree == TREE_SINGLE ? 3 : (s.tree == TREE_DUAL_LUMA ? 1 : 2);
        s.need_org = 0;
    }
}
// a request that only saves / restores a reconstruction
__device__ __forceinline__ void leaf_copy_only(LeafSF& s, Req& q, int mode, int comps, int cont) {
    q.kind = K_NOP;
    q.xchg = false;
    req_copy(q, mode, comps, s.slot, s.bx, s.by, s.lg);
    s.cont = (uint8_t)cont;
}

// full evaluation (get_intra_pred_cost, block_splitter.rs:110-474) of comps with modes [ml, mc, mc];
// the first request of a leaf for a component also (re)builds its reference samples
__device__ __forceinline__ void leaf_full(LeafSF& s, Req& q, int comps, int ml, int mc, bool act, int cont,
                                          bool solo = false) {
    const bool r0 = (comps & 1) && s.need_refs0 != 0;
    const bool r1 = (comps & 2) && mc < LT_CCLM && s.need_refs1 != 0;
    req_full(q, comps, s.bx, s.by, s.lg, ml, mc, !solo, act, r0, r1, false);
    q.tree = s.tree;
    leaf_attach_org(s, q);
    leaf_attach_save(s, q);
    if (act) {
        if (r0) s.need_refs0 = 0;
        if (r1) s.need_refs1 = 0;
    }
    s.op_ml = (uint8_t)ml;
    s.op_mc = (uint8_t)mc;
    s.op_act = act ? 1 : 0;
    s.cont = (uint8_t)cont;
}

// SAD list (get_intra_pred_aux_cost / get_chroma_intra_pred_aux_cost) of n modes, one byte each
__device__ __forceinline__ void leaf_sadlist(LeafSF& s, Req& q, int comps, int n, uint32_t m0, uint32_t m1, uint32_t m2,
                                             uint32_t m3, bool chroma_refs, int cont) {
    q.kind = K_SADLIST;
    q.xchg = false;
    q.tree = s.tree;
    q.comps = comps;
    q.tx = s.bx;
    q.ty = s.by;
    q.tlg = s.lg;
    q.n = n;
    q.modes_lo = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
    q.modes_hi = (unsigned long long)m2 | ((unsigned long long)m3 << 32);
    q.refs0 = (comps & 1) && s.need_refs0 != 0;
    q.refs1 = (comps & 2) && chroma_refs && s.need_refs1 != 0;
    if (q.refs0) s.need_refs0 = 0;
    if (q.refs1) s.need_refs1 = 0;
    leaf_attach_org(s, q);
    leaf_attach_save(s, q);
    s.cont = (uint8_t)cont;
}

// the whole CCLM part of a leaf search as one request (K_CCLMSEARCH): chroma pair only
__device__ __forceinline__ void leaf_cclmsearch(LeafSF& s, Req& q, int cont, bool solo = false) {
    leaf_full(s, q, 2, 0, LT_CCLM, true, cont, solo); // (the mode is a placeholder >= LT_CCLM: no reference samples needed)
    q.kind = K_CCLMSEARCH;
}

// the whole SAD part of a luma / single-tree leaf search as one request (K_SADSEARCH)
__device__ __forceinline__ void leaf_sadsearch(LeafSF& s, Req& q, int comps, int cont) {
    leaf_sadlist(s, q, comps, 13, 0, 0, 0, 0, true, cont);
    q.kind = K_SADSEARCH;
}

// the whole search of a 4x4 DUAL_TREE_LUMA leaf as one request (K_LEAF4, leaf4_search)
#ifndef WRENC_LEAF4
#define WRENC_LEAF4 1
#endif
__device__ __forceinline__ bool leaf_is_leaf4(const LeafSF& s) { return WRENC_LEAF4 && s.tree == TREE_DUAL_LUMA && s.lg == 2; }
__device__ __forceinline__ void leaf_leaf4(LeafSF& s, Req& q, int cont) {
    req_full(q, 1, s.bx, s.by, s.lg, 0, 0, false, true, s.need_refs0 != 0, false, false);
    q.kind = K_LEAF4;
    q.n = 3;
    q.tree = s.tree;
    leaf_attach_org(s, q);
    leaf_attach_save(s, q);
    s.need_refs0 = 0;
    s.cont = (uint8_t)cont;
}

// the whole search of an 8x8 SINGLE_TREE leaf as one request (K_LEAF8, leaf8_search); parts: bit 0 pack {planar, DC},
// bit 1 SAD search + pack {cm, cm - 1, cm + 1}, bit 2 the CCLM part
#ifndef WRENC_LEAF8
#define WRENC_LEAF8 1
#endif
__device__ __forceinline__ bool leaf_is_leaf8(const LeafSF& s) { return WRENC_LEAF8 && s.tree == TREE_SINGLE && s.lg == 3; }
__device__ __forceinline__ void leaf_leaf8(LeafSF& s, Req& q, int parts, int cont) {
    req_full(q, 3, s.bx, s.by, s.lg, 0, 0, false, true, s.need_refs0 != 0, s.need_refs1 != 0, false);
    q.kind = K_LEAF8;
    q.n = parts;
    q.tree = s.tree;
    q.fcur = 0.0f;
    leaf_attach_org(s, q);
    leaf_attach_save(s, q);
    s.need_refs0 = 0;
    s.need_refs1 = 0;
    s.cont = (uint8_t)cont;
}

// the full candidates of a 16x16 SINGLE_TREE leaf as one request (K_LEAF16, leaf16_search); wave schedule only
#ifndef WRENC_LEAF16
#define WRENC_LEAF16 1
#endif
__device__ __forceinline__ bool leaf_is_leaf16(const LeafSF& s) { return WRENC_LEAF16 && s.tree == TREE_SINGLE && s.lg == 4; }
__device__ __forceinline__ void leaf_leaf16(LeafSF& s, Req& q, int cont) {
    leaf_leaf8(s, q, 3, cont);
    q.kind = K_LEAF16;
}

__device__ __forceinline__ EvalParts res_parts(const Res& r) {
    EvalParts e;
    e.ssd_y = r.ssd_y;
    e.ssd_c = r.ssd_c;
    e.lvl_y = r.lvl_y;
    e.lvl_c = r.lvl_c;
    return e;
}
__device__ __forceinline__ void put_parts(EvalPartsSF& d, const EvalParts& e) {
    d.ssd_y = e.ssd_y;
    d.ssd_c = e.ssd_c;
    d.lvl_y = e.lvl_y;
    d.lvl_c = e.lvl_c;
}

// result of a full candidate with luma mode M: running first minimum over the candidates in the
// reference's order; a new best is saved by the next request, any other active candidate has
// overwritten the tile
#define LEAF_CANDIDATE(M)                 \
    do {                                  \
        if (val < s.best_cost) {          \
            s.best_cost = val;            \
            put_parts(s.e_best, rp);      \
            s.mode = (uint8_t)(M);        \
            s.best_cls = (uint8_t)cls;    \
            s.need_save = 1;              \
            s.tile_best = 1;              \
        } else if (s.op_act) {            \
            s.tile_best = 0;              \
        }                                 \
    } while (0)

// One step of a leaf search: SINGLE_TREE / DUAL_TREE_LUMA blocks (block_splitter.rs:886-1078) and
// DUAL_TREE_CHROMA blocks (:794-885; lg = luma log2 = 3).  r is the result of the request the
// previous step made (unused at the first step).  Returns false when the leaf is decided
// (s.cost, s.luma_mode, s.chroma_mode).  The reference's "first minimum wins" selections are kept
// as strict-less running updates in the reference's candidate order; a candidate = one request
// (luma block and chroma pair together, SAD candidates as one list).
template <bool D3>
__device__ __forceinline__ bool leaf_step(const Ctx& c, LeafSF& s, const Res& r, Req& q) {
    const int tree = s.tree;
    const int both = tree == TREE_SINGLE ? 3 : 1;
    int cont = s.cont;
    // RD cost of the full evaluation that just came back (candidates of C_PLANAR .. C_F2)
    float val = 0.0f;
    int cls = 0;
    const EvalParts rp = res_parts(r);
    if (cont == C_PLANAR || cont == C_DCM || cont == C_F0 || cont == C_F1 || cont == C_F2) {
        if (s.op_act) {
            cls = mpm_class(c, s.bx, s.by, s.lg, s.op_ml);
            val = uni_f(assemble_cost(c, tree, cls, s.op_mc, rp));
            if (c.trace && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 1, s.op_ml, s.op_mc, __float_as_int(val));
        } else {
            val = 3.40282347e+38f; // a skipped evaluation is f32::MAX in the reference
        }
    }
    for (;;) {
        switch (cont) {
        case C_START: // candidates {0,1,2,7,13,18,23,29,34,39,45,50,55,60,66} (:887)
            if (D3 && leaf_is_leaf4(s)) { // a 4x4 luma leaf: the whole search in one request
                leaf_leaf4(s, q, C_L4);
                return true;
            }
            if (leaf_is_leaf8(s)) { // an 8x8 single-tree leaf: the whole search in one request, candidates in packs
                leaf_leaf8(s, q, 7, C_L8);
                return true;
            }
            if (leaf_is_leaf16(s)) { // a 16x16 leaf: its five full candidates and the SAD search in one request
                leaf_leaf16(s, q, C_L16);
                return true;
            }
            leaf_full(s, q, both, PLANAR, PLANAR, true, C_PLANAR);
            return true;
        case C_L16: // the winner of [planar, DC, cm, cm - 1, cm + 1] is in the tile; it is saved by the next request
            s.best_cost = r.vmin;
            put_parts(s.e_best, rp);
            s.mode = (uint8_t)r.imin;
            s.best_cls = (uint8_t)mpm_class(c, s.bx, s.by, s.lg, r.imin);
            s.need_save = 1;
            s.tile_best = 1;
            cont = C_WINNER;
            break;
        case C_L8:
            s.cost = r.vmin;
            s.luma_mode = (uint8_t)r.imin;
            s.chroma_mode = (uint8_t)r.imin2;
            return false;
        case C_L4:
            s.cost = r.vmin;
            s.luma_mode = (uint8_t)r.imin;
            s.chroma_mode = (uint8_t)r.imin;
            return false;
        case C_PLANAR:
            s.best_cost = val;
            put_parts(s.e_best, rp);
            s.mode = PLANAR;
            s.best_cls = (uint8_t)cls;
            s.need_save = 1;
            s.tile_best = 1;
            leaf_full(s, q, both, DC, DC, true, C_DCM);
            return true;
        case C_DCM:
            LEAF_CANDIDATE(DC);
            // the 13 directional candidates, their first minimum (:899-904) and step_search(mode, 2, cost, aux = true)
            // (:905-973) in ONE request: the evaluator runs the three lists back to back
            leaf_sadsearch(s, q, both, C_LIST);
            return true;
        case C_LIST: {
            // step_search(mode, 1, _, aux=false) (:974) on {cur, cur - 1, cur + 1}, then the minimum of
            // {planar, DC, dir} (:975-978): first minimum of [planar, DC, cur, cur - 1, cur + 1], kept as
            // one running best.  Out-of-range neighbours are "evaluated" inactive: the wave still
            // walks the schedule so that the workgroup's shared Viterbi barriers stay aligned
            const int cm = r.imin;
            s.cur_mode = (uint8_t)cm;
            s.cur_cost = r.vmin;
            leaf_full(s, q, both, cm, cm, true, C_F0);
            return true;
        }
        case C_F0: {
            const int cm = s.cur_mode;
            LEAF_CANDIDATE(cm);
            leaf_full(s, q, both, cm - 1, cm - 1, !(cm < 3), C_F1);
            return true;
        }
        case C_F1: {
            const int cm = s.cur_mode;
            LEAF_CANDIDATE(cm - 1);
            leaf_full(s, q, both, cm + 1, cm + 1, !(cm + 1 > 66), C_F2);
            return true;
        }
        case C_F2:
            LEAF_CANDIDATE(s.cur_mode + 1);
            cont = C_WINNER;
            break;
        case C_WINNER: {
            s.cost = s.best_cost;
            const int m = s.mode;
            s.luma_mode = (uint8_t)m;
            s.chroma_mode = (uint8_t)m;
            // :989-1037 re-runs the winner's luma to have its reconstruction in the tile; here the
            // winner's reconstruction comes back from slot 0 unless it is still in the tile
            const bool in_tile = s.tile_best != 0;
            if (tree == TREE_DUAL_LUMA) {
                // :1073-1076 repeats the luma evaluation for planar / DC: same parts, same header bits,
                // so the cost it assigns is the candidate's cost already in s.cost
                if (in_tile) return false;
                leaf_copy_only(s, q, COPY_RESTORE, 1, C_WIN);
                return true;
            }
            // :1040 get_chroma_intra_pred_cost(mode) repeats the winner's chroma evaluation: re-use it
            s.cur_cost = uni_f(assemble_chroma_cost(c, m, s.e_best.get()));
            if (c.trace && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, m, __float_as_int((float)s.cur_cost));
            // the three CCLM probes, the pick and the evaluation of the picked mode in one request (K_CCLMSEARCH)
            leaf_cclmsearch(s, q, C_CCLM);
            if (!in_tile) req_copy(q, COPY_RESTORE, 1, s.slot, s.bx, s.by, s.lg); // (its save went out earlier)
            return true;
        }
        case C_WIN:
            return false;
        case C_CCLM: {
            // the CCLM candidate = the winner's luma parts + the chroma parts just evaluated
            s.cclm_mode = (uint8_t)r.imin;
            EvalParts e = s.e_best.get();
            e.ssd_c = rp.ssd_c;
            e.lvl_c = rp.lvl_c;
            const float cclm_cost = uni_f(assemble_chroma_cost(c, s.cclm_mode, e));
            if (c.trace && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.cclm_mode, __float_as_int(cclm_cost));
            const float cur = s.cur_cost;
            const bool dm_wins = cur == fminf(cclm_cost, fminf(cur, 3.40282347e+38f));
            // :1062-1072 final get_intra_pred_cost: luma = the winner; the chroma pair is the DM
            // evaluation (its reconstruction comes back from slot 0) or the CCLM evaluation just made
            const int m = s.mode;
            const int bcls = s.best_cls; // mpm_class of the winner, from its candidate evaluation
            if (dm_wins) {
                s.cost = uni_f(assemble_cost(c, tree, bcls, m, s.e_best.get()));
                leaf_copy_only(s, q, COPY_RESTORE, 2, C_DM);
                return true;
            }
            s.chroma_mode = s.cclm_mode;
            s.cost = uni_f(assemble_cost(c, tree, bcls, s.cclm_mode, e));
            return false;
        }
        case C_DM:
            return false;
        // ---- DUAL_TREE_CHROMA leaf (:794-885) ----
        case C_DC_START: // the three CCLM probes, the pick and the evaluation of the picked mode (K_CCLMSEARCH)
            if (WRENC_LEAF4) { // the whole chroma leaf in one request, CCLM and DM candidates side by side (K_LEAFC4)
                leaf_full(s, q, 2, 0, s.dm_mode, true, C_LC4, true);
                q.kind = K_LEAFC4;
                return true;
            }
            leaf_cclmsearch(s, q, C_DC3);
            return true;
        case C_LC4:
            s.luma_mode = 0;
            s.cost = r.vmin;
            s.chroma_mode = (uint8_t)r.imin;
            return false;
        case C_DC3:
            s.cclm_mode = (uint8_t)r.imin;
            s.c0 = uni_f(assemble_chroma_cost(c, s.cclm_mode, rp));
            if (c.trace && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.cclm_mode, __float_as_int((float)s.c0));
            leaf_full(s, q, 2, 0, s.dm_mode, true, C_DC4);
            req_copy(q, COPY_SAVE, 2, s.slot, s.bx, s.by, s.lg); // keep the CCLM reconstruction (:807-840)
            return true;
        case C_DC4: {
            const float dm_cost = uni_f(assemble_chroma_cost(c, s.dm_mode, rp));
            if (c.trace && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.dm_mode, __float_as_int(dm_cost));
            const float cost = fminf(s.c0, fminf(dm_cost, 3.40282347e+38f));
            s.luma_mode = 0;
            s.cost = cost;
            if (dm_cost == cost) {
                s.chroma_mode = s.dm_mode;
                return false;
            }
            s.chroma_mode = s.cclm_mode;
            leaf_copy_only(s, q, COPY_RESTORE, 2, C_DC5); // :869-873 restore_reconsts
            return true;
        }
        default: // C_DC5
            return false;
        }
    }
}


// ---------------------------------------------------------------------------
// Team schedule: kTeam = 4 waves search ONE CTU.  Every member runs the same control flow on its own copy
// of the state and of the reconstruction tile; where the reference's leaf search evaluates candidates that
// do not depend on each other (block_splitter.rs:887-898: the 15 first-round candidates read only
// neighbours outside the block; :905-973 the two probes of a step-search round; :974 the three full
// evaluations of the last round; the three CCLM probes), each member evaluates ONE of them, the members
// publish their results in LDS (XRes) and meet at a workgroup barrier, and every member then makes the
// reference's decisions from all results in the reference's order.  A full candidate's reconstruction is
// saved to its evaluator's slot 0 in global scratch; once the leaf is decided every member restores the
// winner from the slot of the member that holds it, so all tiles agree again before the next block.
// Quantisation is solo (every member walks its own trellis): between exchanges the members run freely.
// ---------------------------------------------------------------------------

// published result of member m in the exchange that just completed (parity par)
__device__ __forceinline__ EvalParts xparts(const Ctx& c, int par, int m) {
    const XRes& x = team_lds(c, m).xr[par];
    EvalParts e;
    e.ssd_y = (uint32_t)uni((int)x.ssd_y);
    e.ssd_c = (uint32_t)uni((int)x.ssd_c);
    const unsigned long long a = (unsigned long long)x.lvl_y, b = (unsigned long long)x.lvl_c;
    e.lvl_y = (long long)(((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)a));
    e.lvl_c = (long long)(((unsigned long long)(unsigned)uni((int)(b >> 32)) << 32) | (unsigned)uni((int)b));
    return e;
}
// SAD list results: first minimum (f32 bits in ssd_y), its index (ssd_c), the first entry's cost (lvl_y)
__device__ __forceinline__ float xvmin(const Ctx& c, int par, int m) { return __int_as_float(uni((int)team_lds(c, m).xr[par].ssd_y)); }
__device__ __forceinline__ int ximin(const Ctx& c, int par, int m) { return uni((int)team_lds(c, m).xr[par].ssd_c); }
__device__ __forceinline__ float xv0(const Ctx& c, int par, int m) { return __int_as_float(uni((int)team_lds(c, m).xr[par].lvl_y)); }

__device__ __forceinline__ void team_publish(const Req& q, const Res& r, int par) {
    XRes x;
    if (q.kind == K_LEAF8) {
        // a half of the packed search: its best candidate's parts, the mode in the top byte of ssd_y; the CCLM part:
        // the picked mode in ssd_y, the CCLM candidate's chroma parts
        x.ssd_y = (q.n & 4) ? (uint32_t)r.imin : (r.ssd_y | ((uint32_t)r.imin << 24));
        x.ssd_c = r.ssd_c;
        x.lvl_y = r.lvl_y;
        x.lvl_c = r.lvl_c;
    } else if (q.kind == K_FULL || q.kind == K_CCLMSEARCH) {
        x.ssd_y = q.kind == K_CCLMSEARCH ? (uint32_t)r.imin : r.ssd_y; // a chroma-only request: the picked mode rides here
        x.ssd_c = r.ssd_c;
        x.lvl_y = r.lvl_y;
        x.lvl_c = r.lvl_c;
    } else {
        x.ssd_y = (uint32_t)__float_as_int(r.vmin);
        x.ssd_c = (uint32_t)r.imin;
        x.lvl_y = (long long)(unsigned)__float_as_int(r.v0);
        x.lvl_c = 0;
    }
    if (LANE == 0) SH.xr[par] = x;
}

enum { TC_START = 0, TC_A, TC_D, TC_E, TC_F, TC_DONE, TC_DC_START, TC_DC_A, TC_DC_B, TC_L4, TC_L8A, TC_L8B };
// round 2's team stages for the nodes below the CTU (8x8 and 4x4 leaves, the chroma leaf): only built when the level
// schedule does not take those nodes
#define WRENC_OLD_TEAM_SMALL_LEAVES (!(WRENC_LEVELS && WRENC_LEVELS_ALL_DEPTHS))

// a member with nothing to evaluate in a stage
__device__ __forceinline__ void team_idle(Req& q) {
    q.kind = K_NOP;
    q.pre_copy = COPY_NONE;
}
// every member copies comps of the leaf's block from the tile of member `holder` (who holds the winner's
// reconstruction) into its own tile: attached to q as a COPY_PULL, which every member must issue
__device__ __forceinline__ void team_restore(const Ctx& c, const LeafSF& s, Req& q, int comps, int holder) {
    req_copy(q, COPY_PULL, comps, 0, s.bx, s.by, s.lg);
    q.copy_from = holder;
}

// the leaf is decided: the winner's comps come from member `holder`'s tile -- deferred to the driver loop, which
// pulls before the next evaluation (one control step and one empty request less per leaf)
__device__ __forceinline__ void team_defer_pull(CtuSt& t, const LeafSF& s, int comps, int holder) {
    t.dp1 = (uint8_t)((s.bx >> 2) | ((s.by >> 2) << 3));
    t.dp0 = (uint8_t)(1 | (comps << 1) | (holder << 3) | ((s.lg - 2) << 5));
}

// One step of a leaf search in the team schedule; par = parity of the exchange that delivered the results
// of the previous step's requests.  Same decisions, in the same order, as leaf_step.
__device__ __forceinline__ bool leaf_step_team(const Ctx& c, CtuSt& t, LeafSF& s, Req& q, int par, const Res& r) {
    const int tree = s.tree;
    const int both = tree == TREE_SINGLE ? 3 : 1;
    const int me = c.member;
    int cont = s.cont;
    for (;;) {
        switch (cont) {
        case TC_START: // stage A: planar | DC | the directional SAD search (:887-973)
#if WRENC_OLD_TEAM_SMALL_LEAVES // (the level schedule takes every node below the CTU: these stages of round 2's team are unreachable)
            if (leaf_is_leaf4(s)) {
                // a 4x4 luma leaf: the packed search in two halves side by side -- member 0 planar and DC, member 1
                // the SAD search and {cm, cm - 1, cm + 1} -- one exchange, then everybody pulls the winner
                if (me < 2) {
                    leaf_leaf4(s, q, TC_L4);
                    q.n = 1 + me;
                } else {
                    team_idle(q);
                }
                s.cont = TC_L4;
                q.xchg = true;
                return true;
            }
            if (leaf_is_leaf8(s)) {
                // an 8x8 single-tree leaf: the packed search in two halves side by side -- member 0 pack {planar, DC},
                // member 1 the SAD search and pack {cm, cm - 1, cm + 1} -- then the CCLM part on the member that holds
                // the winner (its tile has the winner's reconstruction already), then everybody pulls from it
                if (me < 2)
                    leaf_leaf8(s, q, 1 + me, TC_L8A);
                else
                    team_idle(q);
                s.cont = TC_L8A;
                q.xchg = true;
                return true;
            }
#endif
            if (me == 0) {
                leaf_full(s, q, both, PLANAR, PLANAR, true, TC_A, true);
            } else if (me == 1) {
                leaf_full(s, q, both, DC, DC, true, TC_A, true);
            } else if (me == 2) {
                leaf_sadsearch(s, q, both, TC_A); // the 13 directional SADs + both step-search rounds (K_SADSEARCH)
            } else {
                team_idle(q);
            }
            s.cont = TC_A;
            q.xchg = true;
            return true;
#if WRENC_OLD_TEAM_SMALL_LEAVES // (the level schedule takes every node below the CTU: these stages of round 2's team are unreachable)
        case TC_L4: {
            // first minimum of [planar, DC | cm, cm - 1, cm + 1]: the second half wins only if strictly cheaper
            const float va = xvmin(c, par, 0), vb = xvmin(c, par, 1);
            const int holder = vb < va ? 1 : 0;
            const int m = ximin(c, par, holder);
            s.cost = holder ? vb : va;
            s.luma_mode = (uint8_t)m;
            s.chroma_mode = (uint8_t)m;
            team_defer_pull(t, s, 1, holder);
            return false;
        }
        case TC_L8A: {
            // first minimum of [planar, DC | cm, cm - 1, cm + 1]: the second half wins only if strictly cheaper; each
            // half published its best candidate's parts with the mode in the top byte of ssd_y (an 8x8 SSD is < 2^23)
            EvalParts e0 = xparts(c, par, 0), e1 = xparts(c, par, 1);
            const int m0 = (int)(e0.ssd_y >> 24), m1 = (int)(e1.ssd_y >> 24);
            e0.ssd_y &= 0xFFFFFFu;
            e1.ssd_y &= 0xFFFFFFu;
            const int cls0 = mpm_class(c, s.bx, s.by, s.lg, m0), cls1 = mpm_class(c, s.bx, s.by, s.lg, m1);
            const float va = uni_f(assemble_cost(c, tree, cls0, m0, e0)), vb = uni_f(assemble_cost(c, tree, cls1, m1, e1));
            const int holder = vb < va ? 1 : 0;
            const int m = holder ? m1 : m0;
            s.holder = (uint8_t)holder;
            s.best_cost = holder ? vb : va;
            put_parts(s.e_best, holder ? e1 : e0);
            s.mode = (uint8_t)m;
            s.best_cls = (uint8_t)(holder ? cls1 : cls0);
            // :1040 the winner's chroma cost; the CCLM part (three probes, the pick, the evaluation, DM against CCLM) runs
            // on the holder alone
            s.cur_cost = uni_f(assemble_chroma_cost(c, m, s.e_best.get()));
            if (me == holder) {
                if (c.trace && LANE == 0)
                    TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, m, __float_as_int((float)s.cur_cost));
                leaf_leaf8(s, q, 4, TC_L8B);
                q.ml = m;
                q.fcur = s.cur_cost;
            } else {
                team_idle(q);
            }
            s.cont = TC_L8B;
            q.xchg = true;
            return true;
        }
        case TC_L8B: {
            const int holder = s.holder;
            const EvalParts rp = xparts(c, par, holder); // ssd_y: the CCLM mode picked; ssd_c / lvl_c: its chroma parts
            const int cm = (int)rp.ssd_y;
            EvalParts e = s.e_best.get();
            e.ssd_c = rp.ssd_c;
            e.lvl_c = rp.lvl_c;
            const float cclm_cost = uni_f(assemble_chroma_cost(c, cm, e));
            const float cur = s.cur_cost;
            const bool dm_wins = cur == fminf(cclm_cost, fminf(cur, 3.40282347e+38f));
            const int m = s.mode, bcls = s.best_cls;
            s.luma_mode = (uint8_t)m;
            s.chroma_mode = (uint8_t)(dm_wins ? m : cm);
            s.cost = dm_wins ? uni_f(assemble_cost(c, tree, bcls, m, s.e_best.get())) : uni_f(assemble_cost(c, tree, bcls, cm, e));
            team_defer_pull(t, s, 3, holder);
            return false;
        }
#endif
        case TC_A: {
            const EvalParts e0 = xparts(c, par, 0), e1 = xparts(c, par, 1);
            const float v0 = uni_f(assemble_cost(c, tree, 0, PLANAR, e0));
            const int cls1 = mpm_class(c, s.bx, s.by, s.lg, DC);
            const float v1 = uni_f(assemble_cost(c, tree, cls1, DC, e1));
            if (c.write && LANE == 0) {
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 1, PLANAR, PLANAR, __float_as_int(v0));
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 1, DC, DC, __float_as_int(v1));
            }
            if (v1 < v0) { // first minimum of [planar, DC, ...] as a running strict-less update
                s.best_cost = v1;
                put_parts(s.e_best, e1);
                s.mode = DC;
                s.best_cls = (uint8_t)cls1;
                s.holder = 1;
            } else {
                s.best_cost = v0;
                put_parts(s.e_best, e0);
                s.mode = PLANAR;
                s.best_cls = 0;
                s.holder = 0;
            }
            // member 2 searched the directional modes (K_SADSEARCH): its winner and its SAD
            s.cur_mode = (uint8_t)ximin(c, par, 2);
            s.cur_cost = xvmin(c, par, 2);
            // stage D: cm, cm - 1, cm + 1 side by side (:974) on the three members that do not hold the
            // best of {planar, DC}; candidate k goes to the k-th of them in ascending order
            const int cm = s.cur_mode, holder = s.holder;
            const int k = me - (me > holder ? 1 : 0); // my candidate, if I am not the holder
            const int mode = k == 0 ? cm : (k == 1 ? cm - 1 : cm + 1);
            const bool act = k == 0 || (k == 1 ? !(cm < 3) : !(cm + 1 > 66));
            if (me != holder && act) {
                leaf_full(s, q, both, mode, mode, true, TC_D, true);
            } else {
                team_idle(q);
            }
            s.cont = TC_D;
            q.xchg = true;
            return true;
        }
        case TC_D: {
            const int cm = s.cur_mode, holder0 = s.holder;
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                const int mode = k == 0 ? cm : (k == 1 ? cm - 1 : cm + 1);
                const bool act = k == 0 || (k == 1 ? !(cm < 3) : !(cm + 1 > 66));
                if (!act) continue; // a skipped evaluation costs f32::MAX in the reference: never a new minimum
                const int mem = k + (k >= holder0 ? 1 : 0);
                const EvalParts e = xparts(c, par, mem);
                const int cls = mpm_class(c, s.bx, s.by, s.lg, mode);
                const float val = uni_f(assemble_cost(c, tree, cls, mode, e));
                if (c.write && LANE == 0)
                    TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 1, mode, mode, __float_as_int(val));
                if (val < s.best_cost) {
                    s.best_cost = val;
                    put_parts(s.e_best, e);
                    s.mode = (uint8_t)mode;
                    s.best_cls = (uint8_t)cls;
                    s.holder = (uint8_t)mem;
                }
            }
            s.cost = s.best_cost;
            const int m = s.mode, holder = s.holder;
            s.luma_mode = (uint8_t)m;
            s.chroma_mode = (uint8_t)m;
            if (tree == TREE_DUAL_LUMA) { // every tile gets the winner's luma; nothing else to decide
                team_defer_pull(t, s, 1, holder);
                return false;
            }
            if (WRENC_LEVELS && (WRENC_LEVELS_ALL_DEPTHS || c.k->max_depth == 3) && s.lg == 5 && c.k->max_depth >= 1) {
                // The CTU's 32x32 candidate, and the level schedule follows: its LUMA mode is all the other levels need
                // (their MPM classes), and it is decided now.  Everybody pulls the winner (member 0 keeps
                // the candidate for the CTU's decision); then members 1 .. 3 are done with this leaf and start their
                // levels, while member 0 finishes the candidate alone -- its CCLM part as the one-wave search does it
                // (leaf_step from C_WINNER: the winner saved to slot 0 by the CCLM request, restored from there if DM wins).
                team_defer_pull(t, s, 3, holder);
                if (me != 0) return false;
                s.need_save = 1;
                s.tile_best = 1;
                s.cont = C_WINNER;
                t.lvmode = 1; // from here on ctu_step runs this member's leaf through leaf_step; an empty request first
                team_idle(q); // (the driver loop pulls the winner in front of it, as the others do at the switch)
                q.xchg = false;
                return true;
            }
            // :1040 the winner's chroma cost, then the three CCLM probes side by side on the winner's luma
            s.cur_cost = uni_f(assemble_chroma_cost(c, m, s.e_best.get()));
            if (c.write && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, m, __float_as_int((float)s.cur_cost));
            // the three CCLM probes side by side on the winner's luma, which everybody pulls first (one member doing
            // the three-mode pass and the evaluation in one request, as the wave schedule does, measured 3 % slower here)
            if (me < 3)
                leaf_sadlist(s, q, 2, 1, (uint32_t)(me == 0 ? LT_CCLM : (me == 1 ? T_CCLM : L_CCLM)), 0, 0, 0, false, TC_E);
            else
                team_idle(q);
            team_restore(c, s, q, 3, holder);
            s.cont = TC_E;
            q.xchg = true;
            return true;
        }
        case TC_E: {
            const int cm = pick_cclm(xv0(c, par, 0), xv0(c, par, 1), xv0(c, par, 2));
            s.cclm_mode = (uint8_t)cm;
            const int ev = (s.holder + 1) & (kTeam - 1); // never the holder: its tile keeps the DM chroma
            s.evalr = (uint8_t)ev;
            if (me == ev) {
                leaf_full(s, q, 2, 0, cm, true, TC_F, true);
            } else {
                team_idle(q);
            }
            s.cont = TC_F;
            q.xchg = true;
            return true;
        }
        case TC_F: {
            const int ev = s.evalr, holder = s.holder;
            const EvalParts rp = xparts(c, par, ev);
            EvalParts e = s.e_best.get();
            e.ssd_c = rp.ssd_c;
            e.lvl_c = rp.lvl_c;
            const float cclm_cost = uni_f(assemble_chroma_cost(c, s.cclm_mode, e));
            if (c.write && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.cclm_mode, __float_as_int(cclm_cost));
            const float cur = s.cur_cost;
            const bool dm_wins = cur == fminf(cclm_cost, fminf(cur, 3.40282347e+38f));
            const int m = s.mode, bcls = s.best_cls;
            if (dm_wins) { // only the evaluator's tile holds the CCLM chroma: it takes the DM chroma back
                s.cost = uni_f(assemble_cost(c, tree, bcls, m, s.e_best.get()));
                team_defer_pull(t, s, 2, holder);
                return false;
            }
            s.chroma_mode = s.cclm_mode;
            s.cost = uni_f(assemble_cost(c, tree, bcls, s.cclm_mode, e));
            team_defer_pull(t, s, 2, ev);
            return false;
        }
        case TC_DONE:
            return false;
        // ---- DUAL_TREE_CHROMA leaf (:794-885): the three CCLM probes and the DM evaluation side by side ----
#if WRENC_OLD_TEAM_SMALL_LEAVES // (the level schedule takes every node below the CTU: these stages of round 2's team are unreachable)
        case TC_DC_START: // the three CCLM probes and the DM evaluation side by side
            if (me < 3) {
                leaf_sadlist(s, q, 2, 1, (uint32_t)(me == 0 ? LT_CCLM : (me == 1 ? T_CCLM : L_CCLM)), 0, 0, 0, false, TC_DC_A);
            } else {
                leaf_full(s, q, 2, 0, s.dm_mode, true, TC_DC_A, true);
            }
            s.cont = TC_DC_A;
            q.xchg = true;
            return true;
        case TC_DC_A: {
            const int cm = pick_cclm(xv0(c, par, 0), xv0(c, par, 1), xv0(c, par, 2));
            s.cclm_mode = (uint8_t)cm;
            const float dm_cost = uni_f(assemble_chroma_cost(c, s.dm_mode, xparts(c, par, 3)));
            s.cur_cost = dm_cost;
            if (me == 0) {
                leaf_full(s, q, 2, 0, cm, true, TC_DC_B, true);
            } else {
                team_idle(q);
            }
            s.cont = TC_DC_B;
            q.xchg = true;
            return true;
        }
#endif
        default: // (TC_DC_B)
#if WRENC_OLD_TEAM_SMALL_LEAVES // (the level schedule takes every node below the CTU: these stages of round 2's team are unreachable)
        {
            const float c0 = uni_f(assemble_chroma_cost(c, s.cclm_mode, xparts(c, par, 0)));
            const float dm_cost = s.cur_cost;
            if (c.write && LANE == 0) {
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.cclm_mode, __float_as_int(c0));
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.dm_mode, __float_as_int(dm_cost));
            }
            const float cost = fminf(c0, fminf(dm_cost, 3.40282347e+38f));
            s.luma_mode = 0;
            s.cost = cost;
            const bool dm = dm_cost == cost;
            s.chroma_mode = dm ? s.dm_mode : s.cclm_mode;
            team_defer_pull(t, s, 2, dm ? 3 : 0); // member 3 evaluated DM, member 0 the CCLM mode
            return false;
        }
#else
            return false;
#endif
        }
    }
}

// ---------------------------------------------------------------------------
// Level schedule (team kernel at max-split-depth 3): after the team has searched the CTU's 32x32 candidate together
// (every in-CTU cost needs that candidate's luma mode: SURVEY.md Q7), the members take one TREE LEVEL each and run
// ahead side by side: member 1 the four 16x16 leaf searches, member 2 the sixteen 8x8 ones, member 3 the sixteen
// 8x8 SPLITS (four 4x4 luma leaves + the chroma leaf each); member 0 keeps the 32x32 candidate in its tile.  A
// node's unsplit search and the search of everything below it read only neighbours outside the node
// (block_splitter.rs:1081-1123), so the levels depend on each other only through DECISIONS: node (L, i) is decided --
// unsplit against the sum of its decided children, :1116-1145 -- by the members L .. 3 together once member L has
// searched it and its children are decided; whoever lost copies the winner's reconstruction and decision maps from
// the tile of the member that holds them (LDS to LDS), so every member's tile shows later blocks exactly the
// neighbourhood the reference's depth-first search shows them.  Each member runs the ordinary one-wave leaf search
// (leaf_step with the packed requests K_LEAF16 / K_LEAF8 / K_LEAF4 / K_LEAFC4), solo trellis walks throughout.
// A decision has two meeting points (everybody arrived: the costs are posted; everybody has copied: tiles may be
// written again); they are counters in LDS that the members of the decision poll -- a workgroup barrier would also
// stop the members of other levels, who are in the middle of their own searches.  Same decisions, same f32 sums in
// the same order as ctu_step's depth-first walk; the final pass is shared by quadrant (every member ends with the
// final tile and maps).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lv_meet(int L, int which, int target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (LANE == 0) {
        __hip_atomic_fetch_add(&SHT.lvb.cnt[L][which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (bounded: a member that never arrives must not hang the device; the cap is seconds, a CTU takes milliseconds,
        // and the host reports it: SHT.lvb.pad_ -> WRENC_GPU_EHIP through the overflow word, bit 1)
        int polls = 0;
        while (__hip_atomic_load(&SHT.lvb.cnt[L][which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)target) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1 << 23)) {
                SHT.lvb.pad_ = 1;
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    WSYNC();
}
// position of node i of tree level L in z-order (L = 0: the CTU, 1: 16x16, 2: 8x8)
__device__ __forceinline__ void lv_node(int L, int i, int& bx, int& by) {
    if (L == 0) {
        bx = by = 0;
    } else if (L == 1) {
        bx = (i & 1) * 16;
        by = (i >> 1) * 16;
    } else {
        const int qd = i >> 2, sb = i & 3;
        bx = (qd & 1) * 16 + (sb & 1) * 8;
        by = (qd >> 1) * 16 + (sb >> 1) * 8;
    }
}
// the decision maps of a block from member `from`'s LDS (the split candidate's: they differ per 4x4 unit)
__device__ __forceinline__ void lv_pull_maps(const Ctx& c, int from, int bx, int by, int lg) {
    const Lds& src = team_lds(c, from);
    const int l4 = lg - 2;
    if (LANE < (1 << (2 * l4))) {
        const int idx = ((by >> 2) + (LANE >> l4)) * 8 + (bx >> 2) + (LANE & ((1 << l4) - 1));
        SH.cu_log2[idx] = src.cu_log2[idx];
        SH.luma_mode[idx] = src.luma_mode[idx];
    }
    const int l8 = lg - 3;
    if (LANE < (1 << (2 * l8))) {
        const int idx = ((by >> 3) + (LANE >> l8)) * 4 + (bx >> 3) + (LANE & ((1 << l8) - 1));
        SH.chroma_mode[idx] = src.chroma_mode[idx];
    }
    WSYNC();
}
// block from member `from`'s tile into this member's (copy_block's COPY_PULL without its workgroup barrier)
__device__ __forceinline__ void lv_pull(const Ctx& c, int from, int tx, int ty, int tlg) {
    const Lds& src = team_lds(c, from);
    const int words = 1 << (2 * tlg - 2);
    for (int w = LANE; w < words; w += 64) {
        const int row = (4 * w) >> tlg, col = (4 * w) & ((1 << tlg) - 1);
        const int at = (ty + row) * 36 + tx + col + 4;
        *(uint32_t*)&SH.recY[at] = *(const uint32_t*)&src.recY[at];
    }
    const int lg = tlg - 1;
    const int cwords = 1 << (2 * lg - 2); // per plane
    for (int w = LANE; w < 2 * cwords; w += 64) {
        const int pl = w >= cwords ? 1 : 0;
        const int ww = w - pl * cwords;
        const int row = (4 * ww) >> lg, col = (4 * ww) & ((1 << lg) - 1);
        const int at = ((ty >> 1) + row) * 20 + (tx >> 1) + col + 4;
        *(uint32_t*)&SH.recC[pl][at] = *(const uint32_t*)&src.recC[pl][at];
    }
    WSYNC();
}
// Decision of node (L, idx), by members L .. 3: returns the decided cost (block_splitter.rs:1125-1151)
__device__ __forceinline__ float lv_decide(const Ctx& c, int L, int idx) {
    const int me = c.member;
    const int dl = c.k->max_depth;                      // levels 1 .. dl are searched by members 1 .. dl
    const int nmem = L == 0 ? kTeam : dl - L + 1;       // (the CTU's decision: everybody, idle members included)
    PROF_MARK(tx0_);
    lv_meet(L, 0, (idx + 1) * nmem);
    PROF_MARK(tx1_);
    PROF_ADD2(PH_XCHG, tx0_, tx1_);
    PROF_ADDM(2, tx0_, tx1_);
    const float ns = uni_f(SHT.lvb.ns_cost[L]), sp = uni_f(SHT.lvb.split[L]);
    const int ml = uni((int)SHT.lvb.ml[L]), mc = uni((int)SHT.lvb.mc[L]);
    int bx, by;
    lv_node(L, idx, bx, by);
    const int lg = 5 - L;
    const bool unsplit = sp > ns; // :1125-1145 (ties => split)
    if (unsplit) {
        if (me > L) lv_pull(c, L, bx, by, lg);
        fill_maps(bx, by, lg, ml, mc, true, true);
    } else if (me == L || me > dl) { // (members without a level hold nothing of the split candidate either)
        lv_pull(c, L + 1, bx, by, lg);
        lv_pull_maps(c, L + 1, bx, by, lg);
    }
    PROF_MARK(tx2_);
    lv_meet(L, 1, (idx + 1) * nmem);
    PROF_MARK(tx3_);
    PROF_ADD2(PH_COPY, tx1_, tx2_);
    PROF_ADD2(PH_XCHG, tx2_, tx3_);
    PROF_ADDM(2, tx2_, tx3_);
    return unsplit ? ns : sp;
}
__device__ __forceinline__ void lv_post_unsplit(int L, float ns, int ml, int mc) {
    if (LANE == 0) {
        SHT.lvb.ns_cost[L] = ns;
        SHT.lvb.ml[L] = (uint8_t)ml;
        SHT.lvb.mc[L] = (uint8_t)mc;
    }
}
__device__ __forceinline__ void lv_post_split(int L, float sp) {
    if (LANE == 0) SHT.lvb.split[L] = sp;
}


// split_ct (block_splitter.rs:782-1154) for one CTU + the final pass (ctu_encoder.rs:1421-1461):
// exhaustive quad-tree search as an explicit depth-first walk (level 0 = 32x32 ... level 2 = 8x8;
// an 8x8 node's split is four DUAL_TREE_LUMA 4x4 leaves + one DUAL_TREE_CHROMA 4x4 leaf,
// ctu.rs:1990-2063), per-level state in LDS.
//
// Decided blocks whose reconstruction was overwritten by later candidates come back from the
// slots in global scratch they were saved to (copy_block): the reference's cache_reconsts /
// restore_reconsts (block_splitter.rs:807-840, 1085-1145), with the saved planes kept in L2/HBM
// instead of LDS.
enum { T_START = 0, T_ENTER, T_NODE_LEAF, T_LEAF4_EMIT, T_LEAF4, T_LEAFC, T_REGEN_DONE, T_RETURN, T_FINAL_Z, T_FZ_TAIL, T_FZ_NEXT,
       T_LV_UNIT, T_LV_LEAFDONE, T_LV_UP, T_SPLIT8, T_LV_SERVE };

template <bool TEAM, bool D3>
__device__ __forceinline__ bool ctu_step(Ctx& c, const Res& r, Req& q) {
    CtuSt& t = SH.st;
    bool in_leaf = t.in_leaf != 0;
    int cont = t.cont;
    for (;;) {
        if (in_leaf) {
            // team schedule: the results of the previous requests are in the members' XRes of parity xpar ^ 1
            LeafSF ls = snap_leaf(t.leaf);
            if ((TEAM && !t.lvmode) ? leaf_step_team(c, t, ls, q, t.xpar ^ 1, r) : leaf_step<D3>(c, ls, r, q)) {
                if (t.pend) { // the first request of a node's first child saves the unsplit candidate
                    req_copy(q, COPY_SAVE, t.pend, t.pslot, t.pbx, t.pby, t.plg);
                    t.pend = 0;
                }
                t.cont = (uint8_t)cont;
                t.in_leaf = 1;
                return true;
            }
            t.in_leaf = 0;
            in_leaf = false;
        }
        switch (cont) {
        case T_START:
            t.level = 0;
            t.bx = 0;
            t.by = 0;
            cont = T_ENTER;
            break;
        case T_ENTER: { // enter node (bx, by) at `level`: the unsplit candidate
            const int lg = 5 - t.level;
            t.lg = (uint8_t)lg;
            // wave schedule, a node whose split is tested next: the search saves its best candidate straight to the
            // node's slot 1 + level, which then holds the unsplit candidate without a save of its own (T_NODE_LEAF)
            leaf_init(t.leaf, TREE_SINGLE, t.bx, t.by, lg, 0, (!TEAM && t.level < t.max_depth) ? 1 + t.level : 0);
            if (TEAM && !t.lvmode) t.leaf.cont = TC_START;
            in_leaf = true;
            cont = T_NODE_LEAF;
            break;
        }
        case T_NODE_LEAF: {
            const float ns = t.leaf.cost;
            const int ml = t.leaf.luma_mode, mc = t.leaf.chroma_mode;
            const int level = t.level, lg = t.lg;
            t.ns_cost_cur = ns;
            t.ns_luma_cur = (uint8_t)ml;
            t.ns_chroma_cur = (uint8_t)mc;
            fill_maps(t.bx, t.by, lg, ml, mc, true, true);
            if (level == 0) c.cu32_mode = ml;
            if (t.max_depth - level == 0) {
                t.ret = ns;
                cont = T_RETURN;
                break;
            }
            if (TEAM && WRENC_LEVELS && (WRENC_LEVELS_ALL_DEPTHS || D3)) { // (max-split-depth >= 1 here)
                // ---- level schedule from here on (see lv_decide): every member has searched the 32x32 candidate ----
                if (t.dp0) { // its winner into every member's tile first (team_defer_pull)
                    const int dp = t.dp0, dq = t.dp1;
                    t.dp0 = 0;
                    copy_block(c, COPY_PULL, (dp >> 1) & 3, 0, (dq & 7) << 2, (dq >> 3) << 2, (dp >> 5) + 2, (dp >> 3) & 3);
                }
                t.lvmode = 1;
                t.lv_i = 0;
                t.lv_acc0 = 0.0f;
                t.lv_acc1 = 0.0f;
                if (c.member == 0) lv_post_unsplit(0, ns, ml, mc);
                // (members 1 .. max-split-depth take a level; member 0 and the others wait for the CTU's decision)
                cont = (c.member == 0 || c.member > t.max_depth) ? T_LV_UP : T_LV_UNIT;
                break;
            }
            // the unsplit candidate's reconstruction goes to slot 1 + level (cache_reconsts, :1085-1100).  Wave schedule:
            // the search saved its winner there already (luma and DM chroma, leaf_init above), so all that is left to
            // save is the chroma pair when the CCLM candidate won; a packed 8x8 search (K_LEAF8) saved nothing.
            t.pend = (uint8_t)((TEAM || (WRENC_LEAF8 && lg == 3)) ? 3 : (mc >= LT_CCLM ? 2 : 0));
            t.pbx = t.bx;
            t.pby = t.by;
            t.plg = (uint8_t)lg;
            t.pslot = (uint8_t)(1 + level);
            if (LANE == 0) {
                SH.ns_cost[level] = ns;
                SH.ns_luma[level] = (uint8_t)ml;
                SH.ns_chroma[level] = (uint8_t)mc;
                SH.split_cost[level] = 0.0f;
                SH.child[level] = 0;
            }
            WSYNC();
            if (lg > 3 || !D3) { // (an 8x8 node splits at max-split-depth 3 only)
                t.level = (uint8_t)(level + 1); // descend into child 0 (same top-left corner)
                cont = T_ENTER;
                break;
            }
            // 8x8: four DUAL_TREE_LUMA 4x4 leaves, then the DUAL_TREE_CHROMA leaf
            if (WRENC_SPLIT8) { // ... as one request (K_SPLIT8)
                req_full(q, 3, t.bx, t.by, 3, 0, 0, false, true, false, false, false);
                q.kind = K_SPLIT8;
                q.stage = 0;
                q.tree = TREE_DUAL_LUMA;
                if (t.pend) { // the unsplit 8x8 candidate is saved first
                    req_copy(q, COPY_SAVE, t.pend, t.pslot, t.pbx, t.pby, t.plg);
                    t.pend = 0;
                }
                t.cont = T_SPLIT8;
                return true;
            }
            t.split8 = 0.0f;
            t.i8 = 0;
            cont = T_LEAF4_EMIT;
            break;
        }
        case T_SPLIT8: {
            if constexpr (D3) { // the split candidate of the 8x8 node came back: against the unsplit one, as T_LEAFC does
                const float split8 = r.vmin;
                if (TEAM && t.lvmode) { // level schedule: member 3's unit is done
                    lv_post_split(2, split8);
                    cont = T_LV_UP;
                    break;
                }
                if (split8 > t.ns_cost_cur) { // :1125-1145: the unsplit 8x8 wins, put it back
                    t.rbx = t.bx;
                    t.rby = t.by;
                    t.rlg = t.lg;
                    t.rl = t.ns_luma_cur;
                    t.rc = t.ns_chroma_cur;
                    q.kind = K_NOP;
                    q.xchg = false;
                    req_copy(q, COPY_RESTORE, 3, 1 + t.level, t.rbx, t.rby, t.rlg);
                    t.cont = T_REGEN_DONE;
                    return true;
                }
                t.ret = split8;
                cont = T_RETURN;
                break;
            }
            }
            break;
        case T_LEAF4_EMIT: {
            if constexpr (D3) {
                const int i8 = t.i8;
                leaf_init(t.leaf, TREE_DUAL_LUMA, t.bx + (i8 & 1) * 4, t.by + (i8 >> 1) * 4, 2, 0);
                if (TEAM && !t.lvmode) t.leaf.cont = TC_START;
                in_leaf = true;
                cont = T_LEAF4;
                break;
            }
            }
            break;
        case T_LEAF4: {
            if constexpr (D3) {
                fill_maps(t.leaf.bx, t.leaf.by, 2, t.leaf.luma_mode, 0, true, false);
                t.split8 = t.split8 + t.leaf.cost;
                const int i8 = t.i8 + 1;
                t.i8 = (uint8_t)i8;
                if (i8 < 4) {
                    cont = T_LEAF4_EMIT;
                    break;
                }
                // DM = luma mode of the CU covering the parent's centre (block_splitter.rs:795-805)
                const int bx = t.bx, by = t.by;
                leaf_init(t.leaf, TREE_DUAL_CHROMA, bx, by, 3, uni((int)SH.luma_mode[((by + 4) >> 2) * 8 + ((bx + 4) >> 2)]));
                if (TEAM && !t.lvmode) t.leaf.cont = TC_DC_START;
                in_leaf = true;
                cont = T_LEAFC;
                break;
            }
            }
            break;
        case T_LEAFC: {
            if constexpr (D3) {
                fill_maps(t.bx, t.by, 3, 0, t.leaf.chroma_mode, false, true);
                const float split8 = t.split8 + t.leaf.cost;
                if (TEAM && t.lvmode) { // member 3's unit is done: the split candidate of 8x8 node lv_i
                    lv_post_split(2, split8);
                    cont = T_LV_UP;
                    break;
                }
                if (split8 > t.ns_cost_cur) { // :1125-1145: the unsplit 8x8 wins, put it back
                    t.rbx = t.bx;
                    t.rby = t.by;
                    t.rlg = t.lg;
                    t.rl = t.ns_luma_cur;
                    t.rc = t.ns_chroma_cur;
                    q.kind = K_NOP;
                    q.xchg = false;
                    req_copy(q, COPY_RESTORE, 3, 1 + t.level, t.rbx, t.rby, t.rlg);
                    t.cont = T_REGEN_DONE;
                    return true;
                }
                t.ret = split8;
                cont = T_RETURN;
                break;
            }
            }
            break;
        case T_REGEN_DONE:
            fill_maps(t.rbx, t.rby, t.rlg, t.rl, t.rc, true, true);
            t.ret = uni_f(SH.ns_cost[t.level]);
            cont = T_RETURN;
            break;
        case T_RETURN: { // return `ret` from the finished node at `level` to its parent
            const int level = t.level;
            if (level == 0) {
                t.ctu_cost = t.ret;
                t.z = 0;
                if (TEAM && t.dp0) { // the last leaf's winner is still to be pulled: one empty request for every member
                    q.kind = K_NOP;
                    q.pre_copy = COPY_NONE;
                    q.xchg = false;
                    t.cont = T_RETURN;
                    return true;
                }
                // final pass: every block reads only final neighbours, and every member's tile and maps are final, so the
                // team shares it by 16x16 quadrant (a CU is emitted by the member that owns its top-left unit)
                t.z = (uint8_t)(TEAM ? 16 * c.member : 0);
                t.zend = (uint8_t)(TEAM ? 16 * c.member + 16 : 64);
                cont = T_FINAL_Z;
                break;
            }
            const int pl = level - 1;
            const int psz = 1 << (5 - pl);
            const int pbx = t.bx & ~(psz - 1), pby = t.by & ~(psz - 1);
            // children in z-order, f32 (:1116-1123)
            const float acc = uni_f(uni_f(SH.split_cost[pl]) + t.ret);
            const int ch = uni((int)SH.child[pl]) + 1;
            WSYNC();
            if (LANE == 0) {
                SH.split_cost[pl] = acc;
                SH.child[pl] = (uint8_t)ch;
            }
            WSYNC();
            if (ch < 4) { // next sibling
                t.bx = (uint8_t)(pbx + (ch & 1) * (psz >> 1));
                t.by = (uint8_t)(pby + (ch >> 1) * (psz >> 1));
                cont = T_ENTER;
                break;
            }
            // parent complete: split vs unsplit (:1125-1145)
            t.bx = (uint8_t)pbx;
            t.by = (uint8_t)pby;
            t.level = (uint8_t)pl;
            if (acc > uni_f(SH.ns_cost[pl])) {
                t.rbx = (uint8_t)pbx;
                t.rby = (uint8_t)pby;
                t.rlg = (uint8_t)(5 - pl);
                t.rl = (uint8_t)uni((int)SH.ns_luma[pl]);
                t.rc = (uint8_t)uni((int)SH.ns_chroma[pl]);
                q.kind = K_NOP;
                q.xchg = false;
                req_copy(q, COPY_RESTORE, 3, 1 + pl, t.rbx, t.rby, t.rlg);
                t.cont = T_REGEN_DONE;
                return true;
            }
            t.ret = acc;
            break; // cont stays T_RETURN
        }
        // ---- level schedule: this member's next unit (members 1, 2: a leaf search; member 3: an 8x8 split) ----
        case T_LV_UNIT: {
            if constexpr (TEAM && (WRENC_LEVELS_ALL_DEPTHS || D3)) {
                const int me = c.member, i = t.lv_i;
                int bx, by;
                lv_node(me == 1 ? 1 : 2, i, bx, by);
                t.bx = (uint8_t)bx;
                t.by = (uint8_t)by;
                if constexpr (D3) {
                    if (me == 3) { // four DUAL_TREE_LUMA 4x4 leaves, then the DUAL_TREE_CHROMA leaf
                        t.lg = 3;
                        if (WRENC_SPLIT8) { // ... as one request (K_SPLIT8)
                            req_full(q, 3, bx, by, 3, 0, 0, false, true, false, false, false);
                            q.kind = K_SPLIT8;
                            q.stage = 0;
                            q.tree = TREE_DUAL_LUMA;
                            t.cont = T_SPLIT8;
                            return true;
                        }
                        t.split8 = 0.0f;
                        t.i8 = 0;
                        cont = T_LEAF4_EMIT;
                        break;
                    }
                }
                const int lg = me == 1 ? 4 : 3;
                t.lg = (uint8_t)lg;
                leaf_init(t.leaf, TREE_SINGLE, bx, by, lg, 0);
                in_leaf = true;
                cont = T_LV_LEAFDONE;
                break;
            }
            }
            break;
        case T_LV_LEAFDONE: {
            if constexpr (TEAM && (WRENC_LEVELS_ALL_DEPTHS || D3)) {
                const int me = c.member;
                if (me < t.max_depth) { // the unsplit candidate of node (member, lv_i), to be decided against its children
                    lv_post_unsplit(me, t.leaf.cost, t.leaf.luma_mode, t.leaf.chroma_mode);
                } else {                // the deepest level: the leaf is the node's decision
                    fill_maps(t.bx, t.by, t.lg, t.leaf.luma_mode, t.leaf.chroma_mode, true, true);
                }
                cont = T_LV_UP;
                break;
            }
            }
            break;
        case T_LV_UP: {
            if constexpr (TEAM && (WRENC_LEVELS_ALL_DEPTHS || D3)) { // the decisions this member takes part in now, deepest first
                const int me = c.member, i = t.lv_i, dl = t.max_depth;
                if (WRENC_SERVER && me == 0 && dl >= 2 && !t.lv_i) { // (lv_i of member 0: 1 = the server is through)
                    cont = T_LV_SERVE;
                    break;
                }
                if (WRENC_SERVER && me == 2 && me <= dl && i == 15) lv_word_add(&SHT.lvb.job_fin); // no more 8x8 leaves
                if (WRENC_SERVER && D3 && me == 3 && i == 15) lv_word_add(&SHT.lvb.job4_fin);              // ... 4x4 leaves
                if (me >= 2 && me <= dl) {
                    // an 8x8 node: decided against its split at max-split-depth 3, final at depth 2
                    const float d2 = dl == 3 ? lv_decide(c, 2, i) : (float)t.leaf.cost;
                    const float a1 = uni_f(t.lv_acc1 + d2); // children in z-order, f32, from 0.0 (:1116-1123)
                    t.lv_acc1 = a1;
                    if ((i & 3) != 3) {
                        t.lv_i = (uint8_t)(i + 1);
                        cont = T_LV_UNIT;
                        break;
                    }
                    if (me == 2) lv_post_split(1, a1);
                    t.lv_acc1 = 0.0f;
                }
                if (me >= 1 && me <= dl) {
                    const int i1 = me == 1 ? i : (i >> 2);
                    const float d1 = dl >= 2 ? lv_decide(c, 1, i1) : (float)t.leaf.cost; // (depth 1: the 16x16 leaf is final)
                    const float a0 = uni_f(t.lv_acc0 + d1);
                    t.lv_acc0 = a0;
                    if (i1 != 3) {
                        t.lv_i = (uint8_t)(i + 1);
                        cont = T_LV_UNIT;
                        break;
                    }
                    if (me == 1) lv_post_split(0, a0);
                }
                t.ret = lv_decide(c, 0, 0);
                t.level = 0;
                cont = T_RETURN;
                break;
            }
            }
            break;
        case T_LV_SERVE: {
            if constexpr (TEAM && (WRENC_LEVELS_ALL_DEPTHS || D3)) {
                // Member 0 between its 32x32 candidate and the CTU's decision: pack {planar, DC} of member 2's 8x8 leaves
                // (leaf8_search, q.n & 16), one job at a time: wait for a job or for "no more jobs", wait until the
                // previous job's results have been taken, run it.
                if (LANE == 0) SHT.lvb.srv_ready = 1;
                const bool two = D3 && t.max_depth == 3; // member 3 posts jobs too
                const unsigned served8 = uni((int)lv_word(&SHT.lvb.job_done)), served4 = uni((int)lv_word(&SHT.lvb.job4_done));
                int what = 0; // 1: a job of member 2, 3: of member 3, 2: both are through
                if (LANE == 0) {
                    int polls = 0;
                    for (;;) {
                        const unsigned p8 = lv_word(&SHT.lvb.job_posted), p4 = lv_word(&SHT.lvb.job4_posted);
                        if (two && p4 > served4) { // (the shorter job first: member 3's level is the longer one)
                            what = 3;
                            break;
                        }
                        if (p8 > served8) {
                            what = 1;
                            break;
                        }
                        if (lv_word(&SHT.lvb.job_fin) != 0 && (!two || lv_word(&SHT.lvb.job4_fin) != 0) &&
                            lv_word(&SHT.lvb.job_posted) == served8 && lv_word(&SHT.lvb.job4_posted) == served4) {
                            what = 2;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        if (++polls > (1 << 23)) {
                            SHT.lvb.pad_ = 1;
                            what = 2;
                            break;
                        }
                    }
                }
                what = uni(what);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (what == 2) {
                    t.lv_i = 1;
                    cont = T_LV_UP;
                    break;
                }
                if (what == 3) {
                    lv_word_wait(&SHT.lvb.job4_ack, served4); // the previous results have been taken
                    req_full(q, 1, uni((int)SHT.lvb.job4_bx), uni((int)SHT.lvb.job4_by), 2, 0, 0, false, true, false, false, false);
                    q.kind = K_SERVE4;
                    q.n = 1;
                    q.stage = 0;
                    q.tree = TREE_DUAL_LUMA;
                    t.cont = T_LV_SERVE;
                    return true;
                }
                lv_word_wait(&SHT.lvb.job_ack, served8); // the previous job's reconstructions are still in this wave's park
                req_full(q, 3, uni((int)SHT.lvb.job_bx), uni((int)SHT.lvb.job_by), 3, 0, 0, false, true, false, false, false);
                q.kind = K_LEAF8;
                q.n = 16;
                q.stage = 0;
                q.tree = TREE_SINGLE;
                q.fcur = 0.0f;
                t.cont = T_LV_SERVE;
                return true;
            }
            }
            break;
        // ---- final pass (ctu_encoder.rs:1421-1461): coding order = z-order over the 4x4 units; a
        // CU is emitted at its top-left unit (luma TB, then the chroma TBs) ----
        case T_FINAL_Z: {
            const int z = t.z;
            if (z >= t.zend) {
                t.cont = T_START;
                return false;
            }
            const int bx = 4 * ((z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4));
            const int by = 4 * (((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4));
            const int lg = uni((int)SH.cu_log2[(by >> 2) * 8 + (bx >> 2)]);
            t.bx = (uint8_t)bx;
            t.by = (uint8_t)by;
            t.lg = (uint8_t)lg;
            if ((bx & ((1 << lg) - 1)) == 0 && (by & ((1 << lg) - 1)) == 0) {
                const int ml = uni((int)SH.luma_mode[(by >> 2) * 8 + (bx >> 2)]);
                const int mc = uni((int)SH.chroma_mode[(by >> 3) * 4 + (bx >> 3)]);
                req_full(q, lg >= 3 ? 3 : 1, bx, by, lg, ml, mc, false, true, true, true, true);
                t.cont = T_FZ_TAIL;
                return true;
            }
            cont = T_FZ_NEXT;
            break;
        }
        case T_FZ_TAIL: {
            const int bx = t.bx, by = t.by;
            if (t.lg == 2 && (t.z & 3) == 3) { // after the fourth 4x4 luma CU: the 8x8's chroma CU
                req_full(q, 2, bx & ~7, by & ~7, 3, 0, uni((int)SH.chroma_mode[(by >> 3) * 4 + (bx >> 3)]), false, true,
                         true, true, true);
                t.cont = T_FZ_NEXT;
                return true;
            }
            cont = T_FZ_NEXT;
            break;
        }
        default: { // T_FZ_NEXT: on to the next CU (a CU of 2^lg samples a side covers 4^(lg - 2) units of the z-order)
            const int lgz = t.lg;
            t.z = (uint8_t)(t.z + (lgz <= 2 ? 1 : (1 << (2 * (lgz - 2)))));
            cont = T_FINAL_Z;
            break;
        }
        }
    }
}

// ---------------------------------------------------------------------------
// CTU entry: load, search, final pass, store
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_tables(Ctx c) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        SHT.ldq[i] = (int32_t)c.k->ldq[i];
        SHT.lv[i] = (int32_t)c.k->lv[i];
    }
    for (int i = threadIdx.x; i < 128; i += blockDim.x) ((int8_t*)SHT.fc)[i] = ((const CONST_AS int8_t*)c.k->fc)[i];
    if (threadIdx.x < 6) (&SHT.lvb.cnt[0][0])[threadIdx.x] = 0; // level schedule: the meeting points of this CTU's decisions
    if (threadIdx.x == 0) SHT.lvb.pad_ = 0;
    if (threadIdx.x < 9) (&SHT.lvb.srv_ready)[threadIdx.x] = 0;
    __syncthreads();
}


template <bool TEAM, bool D3>
__device__ __forceinline__ void encode_ctu(Ctx& c, const PicBufs& pb, int ctu_col, int ctu_row, int* overflow) {
    const CONST_AS DevConst* k = c.k;
    const int W = k->W;
    const int Wc = W >> 1;
    c.ctu_x = ctu_col * 32;
    c.ctu_y = ctu_row * 32;
    c.cu32_mode = PLANAR;
#ifdef WRENC_PROFILE
    for (int i = threadIdx.x; i < PH_COUNT; i += blockDim.x) s_prof[i] = 0;
    __syncthreads();
#endif
    PROF_MARK(tt0_);
    load_tables(c);
    GLOBAL_AS uint8_t* const rec = AS_GLOBAL(uint8_t, pb.rec[0]);
    // neighbour border of the reconstruction: row -1 (x = -4..67) and columns -4..-1
    for (int i = LANE; i < 72; i += 64) {
        const int gx = c.ctu_x - 4 + i, gy = c.ctu_y - 1;
        SH.recYtop[i] = (gx >= 0 && gx < W && gy >= 0) ? rec[(size_t)gy * W + gx] : 0;
    }
    // columns -4..-1 and the left CTU's modes: two cache lines of the left CTU's border record instead of 64
    // row pieces of the planes (lane = dword: 32 luma rows, 16 + 16 chroma rows, 2 dwords of modes)
    GLOBAL_AS uint8_t* const my_border =
        AS_GLOBAL(uint8_t, pb.border) + (size_t)(ctu_row * k->ctu_cols + ctu_col) * kBorderBytes;
    {
        const uint32_t v = (c.ctu_x > 0 && LANE < 66) ? ((const GLOBAL_AS uint32_t*)(my_border - kBorderBytes))[LANE] : 0u;
        const uint32_t v2 = (c.ctu_x > 0 && LANE < 2) ? ((const GLOBAL_AS uint32_t*)(my_border - kBorderBytes))[64 + LANE] : 0u;
        if (LANE < 32)
            *(uint32_t*)&SH.recY[LANE * 36] = v;
        else if (LANE < 48)
            *(uint32_t*)&SH.recC[0][(LANE - 32) * 20] = v;
        else
            *(uint32_t*)&SH.recC[1][(LANE - 48) * 20] = v;
        if (LANE < 2) *(uint32_t*)&SH.left_mode[4 * LANE] = v2;
    }
    for (int comp = 1; comp < 3; ++comp)
        for (int i = LANE; i < 40; i += 64) {
            const int gx = (c.ctu_x >> 1) - 4 + i, gy = (c.ctu_y >> 1) - 1;
            SH.recCtop[comp - 1][i] = (gx >= 0 && gx < Wc && gy >= 0) ? rec[plane_off(c, comp) + (size_t)gy * Wc + gx] : 0;
        }
    // tile.rs:49-58: planes start at zero
    for (int i = LANE; i < 32 * 32; i += 64) SH.recY[(i >> 5) * 36 + (i & 31) + 4] = 0;
    for (int comp = 1; comp < 3; ++comp)
        for (int i = LANE; i < 256; i += 64) SH.recC[comp - 1][(i >> 4) * 20 + (i & 15) + 4] = 0;
    WSYNC();
    // ---- the search + final pass: one evaluator, driven by the state machine ----
    #ifndef WRENC_EXP_LDS_PAD
    // LDS is handed out in granules of 1280 bytes (measured: a workgroup of 32,480 bytes -- 672 more than today, for scan
    // tables -- fits the CU four times, not five: 520 -> 460 frames/s, gpurun_out/r3w/ab.log)
    static_assert((sizeof(Lds) * WPB + sizeof(LdsTab) + 1279) / 1280 * 1280 * kWorkgroupsPerCU <= 160 * 1024,
                  "kWorkgroupsPerCU workgroups must fit the CU's 160 KB of LDS");
#endif
    SH.st.cont = T_START;
    SH.st.in_leaf = 0;
    SH.st.pend = 0;
    SH.st.dp0 = 0;
    SH.st.xpar = 0;
    SH.st.lvmode = 0;
    SH.st.fz_on = 0;
    if (LANE == 0) SH.q_pm[0][0][3] = 0; // parity of the pooled quantisation calls (dev_quant.h, zero_flag_cell)
    SH.st.max_depth = (uint8_t)k->max_depth;
    Res r = {};
    Req q = {};
    for (;;) {
#ifdef WRENC_PROFILE
        // control time by where the step starts: leaf continuation (0..19) or 20 + tree continuation
        const int cb_ = SH.st.in_leaf ? (int)SH.st.leaf.cont : 20 + (int)SH.st.cont;
#endif
        PROF_MARK(tc0_);
        const bool more = ctu_step<TEAM, D3>(c, r, q);
        PROF_MARK(tc1_);
        PROF_ADD2(PH_CTRL, tc0_, tc1_);
        PROF_ADDM(0, tc0_, tc1_);
        PROF_ADD2(PH_CB + (cb_ & 31), tc0_, tc1_);
        PROF_ADD2(PH_CBN + (cb_ & 31), 0, 1);
        PROF_ADD2(PH_NSTEP, 0, 1);
        PROF_ADD2(PH_NFULL, 0, (q.kind == K_FULL ? 1 : 0));
        if (!more) break;
        if (TEAM) { // a decided leaf's winner into every member's tile (team_defer_pull), before anything else
            const int dp = SH.st.dp0;
            if (dp) {
                const int dq = SH.st.dp1;
                SH.st.dp0 = 0;
                copy_block(c, COPY_PULL, (dp >> 1) & 3, 0, (dq & 7) << 2, (dq >> 3) << 2, (dp >> 5) + 2, (dp >> 3) & 3);
            }
        }
        if (q.final && !SH.st.fz_on) { // the first block of the final pass (the search no longer needs ns_cost / split_cost)
            SH.st.fz_on = 1;
            if (LANE < 4) {
                SH.lev_was[LANE] = AS_GLOBAL(uint32_t, pb.lev_dirty)[(size_t)(ctu_row * k->ctu_cols + ctu_col) * 4 + LANE];
                SH.lev_now[LANE] = 0;
            }
            WSYNC();
        }
#ifdef WRENC_EXP_CTRL_ONLY // timing / counting experiment only (wrong results): the control flow without evaluations
        r = Res{};
#else
        PROF_MARK(te0_);
        r = evaluate<D3>(c, pb, q, overflow);
        PROF_MARK(te1_);
        PROF_ADDM(q.kind == K_NOP ? 3 : 1, te0_, te1_);
        PROF_ADD2(PH_EV + (((q.kind & 15) * 4 + ((q.kind == K_NOP ? q.copy_tlg : q.tlg) - 2)) & 63), te0_, te1_); // by request kind and block size
        PROF_ADD2(PH_EVN + (((q.kind & 15) * 4 + ((q.kind == K_NOP ? q.copy_tlg : q.tlg) - 2)) & 63), 0, 1);
#ifdef WRENC_PROFILE
        if (LANE == 0 && WAVE < 4 && cb_ < 12) s_prof[PH_ST + 4 * cb_ + WAVE] += te1_ - te0_; // eval time by step origin, member
        if (threadIdx.x == 0 && cb_ < 12) s_prof[PH_STN + cb_] += 1;
#endif
#endif
        if (TEAM && q.xchg) {
            // publish, meet the team (the workgroup's teams walk the same schedule: the same barriers), flip the
            // parity: a fast member's next result goes to the other buffer while slow members still read this one
            PROF_MARK(tx0_);
            const int par = SH.st.xpar;
            team_publish(q, r, par);
            SH.st.xpar = (uint8_t)(par ^ 1);
            __syncthreads();
            PROF_MARK(tx1_);
            PROF_ADD2(PH_XCHG, tx0_, tx1_);
            PROF_ADDM(2, tx0_, tx1_);
        }
    }
    const float cost = SH.st.ctu_cost;
    // which blocks of levels this encode left non-zero: the words of the quadrants whose CUs this wave emitted (wave
    // schedule: all four; a team member: its own, member 0 all four when the CTU is one 32x32 CU)
    // a team whose meeting point timed out (lv_meet) has made its decisions from half-posted costs: it stores nothing more
    // (the host reports WRENC_GPU_EHIP at the next sync / download and the context is to be destroyed, wrenc_gpu.h)
    const bool dead = TEAM && uni((int)SHT.lvb.pad_) != 0;
    if (c.store && SH.st.fz_on && !dead) {
        WSYNC();
        const bool all4 = !TEAM || uni((int)SH.cu_log2[0]) == 5;
        if (LANE < 4 && (all4 || LANE == c.member))
            AS_GLOBAL(uint32_t, pb.lev_dirty)[(size_t)(ctu_row * k->ctu_cols + ctu_col) * 4 + LANE] = SH.lev_now[LANE];
    }
    // store recon + decisions
    if (c.write && !dead) {
        for (int i = LANE; i < 1024 / 4; i += 64) {
            const int y = i >> 3, x4 = (i & 7) * 4;
            *(GLOBAL_AS uint32_t*)&rec[(size_t)(c.ctu_y + y) * W + c.ctu_x + x4] = *(const uint32_t*)&SH.recY[y * 36 + x4 + 4];
        }
        for (int comp = 1; comp < 3; ++comp)
            for (int i = LANE; i < 256 / 4; i += 64) {
                const int y = i >> 2, x4 = (i & 3) * 4;
                *(GLOBAL_AS uint32_t*)&rec[plane_off(c, comp) + (size_t)((c.ctu_y >> 1) + y) * Wc + (c.ctu_x >> 1) + x4] =
                    *(const uint32_t*)&SH.recC[comp - 1][y * 20 + x4 + 4];
            }
        const int i = LANE; // 64 4x4 units
        const size_t o = (size_t)((c.ctu_y >> 2) + (i >> 3)) * (W >> 2) + (c.ctu_x >> 2) + (i & 7);
        AS_GLOBAL(uint8_t, pb.cu_log2)[o] = SH.cu_log2[i];
        AS_GLOBAL(uint8_t, pb.luma_mode)[o] = SH.luma_mode[i];
        if (i < 16) {
            const size_t oc = (size_t)((c.ctu_y >> 3) + (i >> 2)) * (W >> 3) + (c.ctu_x >> 3) + (i & 3);
            AS_GLOBAL(uint8_t, pb.chroma_mode)[oc] = SH.chroma_mode[i];
        }
        if (i == 0) AS_GLOBAL(float, pb.ctu_cost)[ctu_row * k->ctu_cols + ctu_col] = cost;
        // the border record for the CTU to the right (see kBorderBytes)
        {
            uint32_t v;
            if (LANE < 32)
                v = *(const uint32_t*)&SH.recY[LANE * 36 + 32];
            else if (LANE < 48)
                v = *(const uint32_t*)&SH.recC[0][(LANE - 32) * 20 + 16];
            else
                v = *(const uint32_t*)&SH.recC[1][(LANE - 48) * 20 + 16];
            ((GLOBAL_AS uint32_t*)my_border)[LANE] = v;
            if (LANE < 2) {
                const int r0 = 4 * LANE; // modes of the 4x4 units (7, r0 .. r0 + 3)
                ((GLOBAL_AS uint32_t*)my_border)[64 + LANE] =
                    (uint32_t)SH.luma_mode[r0 * 8 + 7] | ((uint32_t)SH.luma_mode[(r0 + 1) * 8 + 7] << 8) |
                    ((uint32_t)SH.luma_mode[(r0 + 2) * 8 + 7] << 16) | ((uint32_t)SH.luma_mode[(r0 + 3) * 8 + 7] << 24);
            }
        }
    }
#ifdef WRENC_PROFILE
    PROF_MARK(tt1_);
    PROF_ADD2(PH_TOTAL, tt0_, tt1_);
    if (threadIdx.x == 0) s_prof[PH_HIST + (int)min((unsigned long long)63, (tt1_ - tt0_) >> 17)] += 1; // how long this CTU took (the launch waits for the slowest)
    __syncthreads();
    for (int i = threadIdx.x; i < PH_COUNT; i += blockDim.x) atomicAdd(&g_prof[i], s_prof[i]);
#endif
}

} // namespace wrenc
