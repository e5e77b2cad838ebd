// wrenc_dev.h -- CDNA4 (gfx950) device code of the all-intra RD-search path.
//
// Execution model: ONE 64-lane wavefront owns one 32x32 CTU.  Every block is a
// single wave (blockDim.x == 64), so control flow is wave-uniform and
// __syncthreads() is only a compiler/LDS ordering point.  The CTU's original
// samples, its reconstruction (with the neighbour border needed for intra
// reference samples), all transform buffers and the decision maps live in LDS;
// HBM is touched once to load the CTU + border and once to store recon, levels
// and decisions.
//
// What is computed follows the reference function by function (paths relative to
// the reference's src/); how it is computed is wave-parallel:
//   predict        intra_predictor.rs:56-2055
//   fwd/inv DCT-2  transformer.rs:2040-2737
//   dep-quant      quantizer.rs:338-759 as a backward 4-state Viterbi, one lane
//                  per state, exchanging path costs with DPP quad permutes
//   leaf search    block_splitter.rs:782-1154
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wrenc {

enum { PLANAR = 0, DC = 1, LT_CCLM = 81, L_CCLM = 82, T_CCLM = 83 };
enum { TREE_SINGLE = 0, TREE_DUAL_LUMA = 1, TREE_DUAL_CHROMA = 2 };

// Constants resolved on the host (see wrenc_gpu_config in include/wrenc_gpu.h).
struct DevConst {
    int32_t W, H, qp, max_depth, ctu_cols, ctu_rows;
    int32_t lsc;              // quantizer.rs:617-622 (16*LEVEL_SCALE[0][(qp+1)%6]) << ((qp+1)/6)
    uint64_t div_magic;       // floor(2^47 / lsc) + 1: exact n / lsc for n < 2^26
    int64_t lambda_q;
    float lambda_rd;
    float lambda_rd_chroma;
    int64_t ldq[1024];        // lambda_q * dq_table[bits]  (quantizer.rs:29-31)
    int64_t lv[1024];
    int64_t hb_luma[2][4][67];
    int64_t hb_chroma[4];
    int16_t dct[4][32][32];   // T_N[u][k] = dct64[u * 64/N][k], N = 4 << idx (transformer.rs:1212-1221)
    int16_t dct_t[4][32][32]; // transposed: dct_t[idx][y][i] = T_N[i][y]
    uint8_t diag4[16][2];     // 4x4 up-right diagonal scan (x, y)   (ctu.rs:14-81)
    uint8_t diag_sb[4][64][2]; // sub-block scan for 1, 4, 16, 64 sub-blocks
    uint16_t scan_idx[4][1024]; // raster index y*n+x of reverse-scan position p (p = 0: last in scan)
    int16_t intra_angle[95];  // common.rs:145
    int8_t fc[32][4];         // common.rs:153
};

// One picture's device buffers.
struct PicBufs {
    const uint8_t* org[3];
    uint8_t* rec[3];
    int16_t* lev[3];
    uint8_t* cu_log2;
    uint8_t* luma_mode;
    uint8_t* chroma_mode;
    float* ctu_cost;
};

// LDS working set of one wave / one CTU.
struct __attribute__((aligned(16))) Lds {
    // Transform working set, time-multiplexed (see code_component):
    //   r1: residual -> coefficients -> Viterbi chunk costs / levels -> reconstructed residual
    //   r2: stage-1 DCT output (i32) -> scan-order coefficients + quotients -> dequantised^T + V
    int16_t r1[1024];
    int32_t r2[33 * 32];
    // reference samples of the current block, built once per (block, component) and reused by
    // every candidate mode: luma unfiltered + [1 2 1]-filtered, chroma unfiltered
    int16_t refL0[66], refA0[64], refLf0[66], refAf0[64];
    int16_t refLc[2][34], refAc[2][32];
    uint8_t recYtop[72];       // y = -1, x = -4..67 (index x+4)
    uint8_t recY[32 * 36];     // x = -4..31 (index x+4), stride 36
    uint8_t recCtop[2][40];    // y = -1, x = -4..35
    uint8_t recC[2][16 * 20];  // x = -4..15, stride 20
    uint32_t decw[128];        // trellis decisions: 4 bits per position, 8 positions per word
    int32_t q_istar[2];        // shared-Viterbi hand-off, per block: first position with a non-zero state-0 level
    int32_t q_active;          // this wave's TB takes part in the shared Viterbi
    uint8_t cu_log2[64];       // per 4x4 luma unit
    uint8_t luma_mode[64];
    uint8_t chroma_mode[16];   // per 8x8 luma unit
    uint8_t left_mode[8];      // luma mode of the CU left of the CTU, per 4 rows
    float ns_cost[4];          // per tree level: no-split cost, running split cost
    float split_cost[4];
    uint8_t ns_luma[4], ns_chroma[4], child[4];
};

// Per-wave uniform context.
// Per-wave uniform context, passed BY VALUE (a few registers) so that the out-of-line
// stage functions never reload it from memory.
struct Ctx {
    const DevConst* __restrict__ k;
    const uint8_t* org[3];              // original planes of this wave's picture (read-only, L2-resident)
    uint8_t* pred_scratch;              // 1 KB per wave in HBM: prediction bytes between predict and recon
    unsigned long long* mismatch;
    int ctu_x, ctu_y; // luma, picture coordinates
    int cu32_mode;    // in-CTU neighbour lookups during search resolve to the root CU
    int write;        // 0 for a padding wave (batch not a multiple of WPB): compute, never store
};

// LDS: one working set per wave (= per CTU), WPB waves per workgroup, plus tables shared
// by the workgroup.  File scope so that every access is a DS instruction (no FLAT ops).
// The waves of a workgroup process the SAME CTU position of WPB different pictures, so
// they execute the same schedule; the 4-lane Viterbi of all WPB transform blocks is run by
// wave 0 in WPB quads at once (see quantize()).
#ifndef WRENC_WPB
#define WRENC_WPB 8
#endif
constexpr int WPB = WRENC_WPB;
struct LdsTab {
    int32_t ldq[256];
    int32_t lv[256];
    int8_t fc[32][4]; // common.rs:153 (copied from the constant block)
};
__shared__ Lds SHW[WPB];
__shared__ LdsTab SHT;
#define LANE ((int)(threadIdx.x & 63))
#define WAVE (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))
#define SH (SHW[WAVE])

// Everything in Ctx and every block-geometry argument is wave-uniform.  Out-of-line
// functions receive arguments in VGPRs; re-deriving them through readfirstlane lets the
// compiler keep them in SGPRs (scalar ALU, scalar branches, s_load from the constant block).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ Ctx uni(Ctx c) {
    static_assert(sizeof(Ctx) % 4 == 0, "Ctx must be a whole number of dwords");
    int w[sizeof(Ctx) / 4];
    __builtin_memcpy(w, &c, sizeof(Ctx));
#pragma unroll
    for (unsigned i = 0; i < sizeof(Ctx) / 4; ++i) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
    Ctx r;
    __builtin_memcpy(&r, w, sizeof(Ctx));
    return r;
}

// One wave per block: LDS operations of a wave are issued and serviced in program order,
// so "synchronising" only has to stop the compiler from reordering LDS accesses.
#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

// Diagnostic build only (-DWRENC_PROFILE): per-phase cycle counters, summed per wave and
// added to a global table at CTU end.  Never compiled into the product library.
#ifdef WRENC_PROFILE
enum { PH_PREDICT, PH_FDCT, PH_QPRE, PH_QBACK, PH_QTRACE, PH_DEQ, PH_IDCT, PH_RECON, PH_TOTAL, PH_COUNT };
__device__ unsigned long long g_prof[PH_COUNT];
__shared__ unsigned long long s_prof[PH_COUNT];
#define PROF_T0() const unsigned long long prof_t0_ = __builtin_readcyclecounter()
#define PROF_ADD(ph) do { if (threadIdx.x == 0) s_prof[ph] += __builtin_readcyclecounter() - prof_t0_; } while (0)
#define PROF_MARK(var) const unsigned long long var = __builtin_readcyclecounter()
#define PROF_ADD2(ph, a, b) do { if (threadIdx.x == 0) s_prof[ph] += (b) - (a); } while (0)
#else
#define PROF_T0()
#define PROF_ADD(ph)
#define PROF_MARK(var)
#define PROF_ADD2(ph, a, b)
#endif

// ---------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// minimum over aligned groups of `width` lanes (width = 64 or 32)
__device__ __forceinline__ int group_min_i32(int v, int width) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        if (m < width) v = min(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int ilog2i(int v) { return 31 - __clz(v); }

// ---------------------------------------------------------------------------
// recon tile access (CTU-local component coordinates)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rec_get(int c, int x, int y) {
    if (c == 0) return y < 0 ? SH.recYtop[x + 4] : SH.recY[y * 36 + x + 4];
    return y < 0 ? SH.recCtop[c - 1][x + 4] : SH.recC[c - 1][y * 20 + x + 4];
}
__device__ __forceinline__ void rec_put(int c, int x, int y, int v) {
    if (c == 0)
        SH.recY[y * 36 + x + 4] = (uint8_t)v;
    else
        SH.recC[c - 1][y * 20 + x + 4] = (uint8_t)v;
}
// original sample at CTU-local component coordinates (global load; the planes are read-only
// for the whole launch, so the loads are cacheable and need no ordering)
__device__ __forceinline__ int org_get(const Ctx& c, int pc, int x, int y) {
    const int cs = pc ? 1 : 0;
    const int stride = c.k->W >> cs;
    const uint8_t* base = pc == 0 ? c.org[0] : (pc == 1 ? c.org[1] : c.org[2]); // pc may differ per lane
    return base[(size_t)((c.ctu_y >> cs) + y) * stride + (c.ctu_x >> cs) + x];
}

// ---------------------------------------------------------------------------
// availability (ctu.rs:2083-2188, encoder_context.rs:918-956)
// bx, by: CTU-local luma position, lg: log2 luma size
// ---------------------------------------------------------------------------
__device__ inline bool above_right_avail(Ctx c, int bx, int by, int lg) {
    for (;;) {
        const int n = 1 << lg;
        if (c.ctu_x + bx + n >= c.k->W) return false;
        if (lg == 5) return c.ctu_y > 0 && c.ctu_x + 32 < c.k->W;
        const int px = bx & ~(2 * n - 1), py = by & ~(2 * n - 1);
        if (bx == px && by == py) return c.ctu_y + by > 0;
        if (by == py) { // top-right child: parent's
            bx = px;
            by = py;
            lg += 1;
            continue;
        }
        if (bx == px) return true;
        return false;
    }
}
__device__ inline bool below_left_avail(Ctx c, int bx, int by, int lg) {
    for (;;) {
        const int n = 1 << lg;
        if (c.ctu_y + by + n >= c.k->H) return false;
        if (lg == 5) return false;
        const int px = bx & ~(2 * n - 1), py = by & ~(2 * n - 1);
        if (px < bx) return false;
        if (by + n < py + 2 * n) return c.ctu_x + bx > 0;
        bx = px;
        by = py;
        lg += 1;
    }
}
__device__ __forceinline__ bool nb_avail(Ctx c, int gx, int gy, int tn, int xn, int yn,
                                         bool ar, bool bl) {
    return xn >= 0 && yn >= 0 && xn < c.k->W && yn < c.k->H &&
           ((xn >> 5) <= (gx >> 5) || (yn >> 5) < (gy >> 5)) && (yn >> 5) < (gy >> 5) + 1 &&
           (xn < gx + tn || ar) && (yn < gy + tn || bl);
}

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
__device__ __noinline__ void build_refs(Ctx c, int comp, int tx, int ty, int tlg) {
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
    const int gx = c.ctu_x + tx, gy = c.ctu_y + ty;
    const bool ar = above_right_avail(c, tx, ty, tlg);
    const bool bl = below_left_avail(c, tx, ty, tlg);
    const int st = 1 << cs;
    // segment availabilities in substitution-scan order: BL, L, corner, A, AR (bit j = segment j)
    int avm = 0;
    avm |= nb_avail(c, gx, gy, tn, gx - st, gy + tn, ar, bl) ? 1 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx - st, gy, ar, bl) ? 2 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx - st, gy - st, ar, bl) ? 4 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx, gy - st, ar, bl) ? 8 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx + tn, gy - st, ar, bl) ? 16 : 0;
    const bool any = avm != 0;
    const int total = 4 * n + 1;
    for (int tt = LANE; tt < nb * total; tt += 64) {
        const int blk = tt >= total ? 1 : 0;
        const int t = tt - blk * total;
        const int pc = comp + blk;
        int16_t* refL = pc == 0 ? SH.refL0 : SH.refLc[pc - 1];
        int16_t* refA = pc == 0 ? SH.refA0 : SH.refAc[pc - 1];
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
            refL[li] = (int16_t)v;
        else
            refA[ai] = (int16_t)v;
    }
    WSYNC();
    // [1 2 1] filter, intra_predictor.rs:304-352 (used by modes 0, 2, 34, 66 only)
    if (comp == 0 && n * n > 32) {
        const int16_t* refL = SH.refL0;
        const int16_t* refA = SH.refA0;
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
                SH.refLf0[li] = (int16_t)v;
            } else {
                const int ai = t - (2 * n + 1);
                int v;
                if (ai == 2 * n - 1)
                    v = refA[ai];
                else if (ai == 0)
                    v = (refL[0] + 2 * refA[0] + refA[1] + 2) >> 2;
                else
                    v = (refA[ai - 1] + 2 * refA[ai] + refA[ai + 1] + 2) >> 2;
                SH.refAf0[ai] = (int16_t)v;
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

// `comp` (plane 1 or 2) may differ per lane: everything that depends on it is per-lane data
__device__ __forceinline__ CclmParams cclm_params(Ctx c, int comp, int tx, int ty, int tlg, int mode) {
    c = uni(c);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    mode = uni(mode);
    CclmParams r;
    const int tn = 1 << tlg;
    const int tw = tn >> 1, th = tw;
    const int cx = tx >> 1, cy = ty >> 1;
    const int gx = c.ctu_x + tx, gy = c.ctu_y + ty;
    const bool avail_l = nb_avail(c, gx, gy, tn, gx - 1, gy, false, false);
    const bool avail_t = nb_avail(c, gx, gy, tn, gx, gy - 1, false, false);
    r.avail_l = avail_l;
    int num_top_right = 0, num_below_left = 0;
    if (mode == T_CCLM) {
        const bool ar = above_right_avail(c, tx, ty, tlg), bl = below_left_avail(c, tx, ty, tlg);
        bool a = true;
        for (int x = tw; x < 2 * tw && a; ++x) {
            a = nb_avail(c, gx, gy, tn, gx + x * 2, gy - 1, ar, bl);
            if (a) ++num_top_right;
        }
    }
    if (mode == L_CCLM) {
        const bool ar = above_right_avail(c, tx, ty, tlg), bl = below_left_avail(c, tx, ty, tlg);
        bool a = true;
        for (int y = th; y < 2 * th && a; ++y) {
            a = nb_avail(c, gx, gy, tn, gx - 1, gy + y * 2, ar, bl);
            if (a) ++num_below_left;
        }
    }
    int num_samp_t, num_samp_l;
    if (mode == LT_CCLM) {
        num_samp_t = avail_t ? tw : 0;
        num_samp_l = avail_l ? th : 0;
    } else {
        num_samp_t = (avail_t && mode == T_CCLM) ? tw + min(num_top_right, th) : 0;
        num_samp_l = (avail_l && mode == L_CCLM) ? th + min(num_below_left, tw) : 0;
    }
    r.flat128 = (num_samp_l == 0 && num_samp_t == 0);
    r.a = 0;
    r.k = 0;
    r.b = 128;
    if (r.flat128) return r;
    const bool b_ctu_boundary = ((c.ctu_y + ty) & 31) == 0;
    const int num_is_4 = !(avail_t && avail_l && mode == LT_CCLM) ? 1 : 0;
    int cnt_t = 0, cnt_l = 0;
    int y0 = 0, y1 = 0, y2 = 0, y3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0; // p_sel_ds_y / p_sel_c
    auto put = [&](int i, int yy, int cc_) {
        if (i == 0) { y0 = yy; c0 = cc_; }
        else if (i == 1) { y1 = yy; c1 = cc_; }
        else if (i == 2) { y2 = yy; c2 = cc_; }
        else { y3 = yy; c3 = cc_; }
    };
    if (avail_t && (mode == LT_CCLM || mode == T_CCLM)) {
        const int start = num_samp_t >> (2 + num_is_4);
        const int step = max(num_samp_t >> (1 + num_is_4), 1);
        cnt_t = min((1 + num_is_4) << 1, num_samp_t);
        for (int i = 0; i < cnt_t; ++i) {
            const int pos = start + i * step;
            const int sc = rec_get(comp, cx + pos, cy - 1);
            const int sx = 2 * pos;
            int sy;
            if (!b_ctu_boundary)
                sy = (cclm_w(c, tx, ty, -1, sx - 1, avail_l) + cclm_w(c, tx, ty, -2, sx - 1, avail_l) +
                      cclm_w(c, tx, ty, -1, sx, avail_l) * 2 + cclm_w(c, tx, ty, -2, sx, avail_l) * 2 +
                      cclm_w(c, tx, ty, -1, sx + 1, avail_l) + cclm_w(c, tx, ty, -2, sx + 1, avail_l) + 4) >> 3;
            else
                sy = (cclm_w(c, tx, ty, -1, sx - 1, avail_l) + cclm_w(c, tx, ty, -1, sx, avail_l) * 2 +
                      cclm_w(c, tx, ty, -1, sx + 1, avail_l) + 2) >> 2;
            put(i, sy, sc);
        }
    }
    if (avail_l && (mode == LT_CCLM || mode == L_CCLM)) {
        const int start = num_samp_l >> (2 + num_is_4);
        const int step = max(num_samp_l >> (1 + num_is_4), 1);
        cnt_l = min((1 + num_is_4) << 1, num_samp_l);
        for (int i = 0; i < cnt_l; ++i) {
            const int pos = start + i * step;
            put(cnt_t + i, cclm_ds6(c, tx, ty, 2 * pos, -2, avail_l), rec_get(comp, cx - 1, cy + pos));
        }
    }
    // min group {0,2}, max group {1,3} and the four compare-exchanges of :1973-1986,
    // carried out on (luma, chroma) value pairs instead of indices
    int mnAy = y0, mnAc = c0, mnBy = y2, mnBc = c2, mxAy = y1, mxAc = c1, mxBy = y3, mxBc = c3, t;
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
    return r;
}

// one predicted sample of plane pc: accumulate |org - pred|; FULL also stores residual and prediction
template <bool FULL>
__device__ __forceinline__ int emit_sample(const Ctx& c, int pc, int x, int y, int i, int v) {
    const int d = org_get(c, pc, x, y) - v;
    if (FULL) {
        SH.r1[i] = (int16_t)d;
        c.pred_scratch[i] = (uint8_t)v;
    }
    return d < 0 ? -d : d;
}

// Prediction of one luma block (comp 0) or of the Cb+Cr pair (comp 1) from the cached reference
// samples (build_refs must have run for this block; CCLM reads the reconstructed luma instead).
// Sample index i runs over nb*n*n: block blk = i / (n*n), then row-major inside the block.
// FULL: the residual org - pred goes to r1[i] and the prediction byte to this wave's scratch
//       (each lane later re-reads exactly the bytes it wrote).
// Returns the lane's partial sum of |org - pred| (the SAD of block_splitter.rs:96-104).
template <bool FULL>
__device__ __forceinline__ int predict(Ctx c, int comp, int tx, int ty, int tlg, int mode) {
    c = uni(c);
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
    int sad = 0;
    if (mode >= LT_CCLM) {
        // model parameters of both planes in one pass: odd lanes derive Cr, even lanes Cb
        const CclmParams cpv = cclm_params(c, 1 + (LANE & 1), tx, ty, tlg, mode);
        const int a0 = __builtin_amdgcn_readlane(cpv.a, 0), a1 = __builtin_amdgcn_readlane(cpv.a, 1);
        const int k0 = __builtin_amdgcn_readlane(cpv.k, 0), k1 = __builtin_amdgcn_readlane(cpv.k, 1);
        const int b0 = __builtin_amdgcn_readlane(cpv.b, 0), b1 = __builtin_amdgcn_readlane(cpv.b, 1);
        const bool flat128 = __builtin_amdgcn_readlane((int)cpv.flat128, 0) != 0;
        const bool avail_l = __builtin_amdgcn_readlane((int)cpv.avail_l, 0) != 0;
        for (int i = LANE; i < nb * nn; i += 64) {
            const int blk = i >> (2 * lg);
            const int ii = i & (nn - 1);
            const int x = ii & (n - 1), y = ii >> lg;
            int v;
            if (flat128) {
                v = 128;
            } else {
                const int ds = cclm_ds6(c, tx, ty, 2 * y, 2 * x, avail_l);
                v = ((ds * (blk ? a1 : a0)) >> (blk ? k1 : k0)) + (blk ? b1 : b0);
                v = min(max(v, 0), 255);
            }
            sad += emit_sample<FULL>(c, comp + blk, cx + x, cy + y, i, v);
        }
        WSYNC();
        return sad;
    }
    // luma blocks of more than 32 samples use the filtered references for modes 0, 2, 34, 66
    const bool filt = comp == 0 && nn > 32 && (mode == 0 || mode == 2 || mode == 34 || mode == 66);
    const int16_t* L0 = comp == 0 ? (filt ? SH.refLf0 : SH.refL0) : SH.refLc[0]; // index 0 = corner
    const int16_t* A0 = comp == 0 ? (filt ? SH.refAf0 : SH.refA0) : SH.refAc[0];
    if (mode == PLANAR || mode == DC) {
        int dcv0 = 0, dcv1 = 0;
        if (mode == DC) {
            int part0 = 0, part1 = 0;
            for (int t = LANE; t < 2 * n; t += 64) {
                part0 += t < n ? A0[t] : L0[t - n + 1];
                if (nb == 2) part1 += t < n ? SH.refAc[1][t] : SH.refLc[1][t - n + 1];
            }
            dcv0 = ((wave_sum_i32(part0) + n) >> (lg + 1)) & 0xFF; // `as u8`
            if (nb == 2) dcv1 = ((wave_sum_i32(part1) + n) >> (lg + 1)) & 0xFF;
        }
        const int n_scale = (2 * lg - 2) >> 2;
        for (int i = LANE; i < nb * nn; i += 64) {
            const int blk = i >> (2 * lg);
            const int ii = i & (nn - 1);
            const int x = ii & (n - 1), y = ii >> lg;
            const int16_t* L = blk ? SH.refLc[1] : L0;
            const int16_t* A = blk ? SH.refAc[1] : A0;
            int v;
            if (mode == PLANAR) {
                const int pv = (n - 1 - y) * A[x] + (y + 1) * L[n + 1];
                const int ph = (n - 1 - x) * L[y + 1] + (x + 1) * A[n];
                v = ((pv + ph + n) >> (lg + 1)) & 0xFF;
            } else {
                v = blk ? dcv1 : dcv0;
            }
            const int wl = pdpc_w(n_scale, x), wt = pdpc_w(n_scale, y);
            v = (int16_t)(L[y + 1] * wl + A[x] * wt + (64 - wt - wl) * v + 32) >> 6;
            v = min(max(v, 0), 255);
            sad += emit_sample<FULL>(c, comp + blk, cx + x, cy + y, i, v);
        }
        WSYNC();
        return sad;
    }
    // angular 2..66 (intra_predictor.rs:1287-1602), square blocks
    const int angle = c.k->intra_angle[14 + mode];
    int inv_angle = 0;
    if (angle > 0)
        inv_angle = (512 * 32 + angle / 2) / angle;
    else if (angle < 0)
        inv_angle = -((512 * 32 + (-angle) / 2) / -angle);
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
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg);
        const int ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int16_t* L = blk ? SH.refLc[1] : L0;
        const int16_t* A = blk ? SH.refAc[1] : A0;
        const int alrs = L[0];
        int v;
        if (mode >= 34) {
            const int i_idx = ((y + 1) * angle) >> 5;
            const int i_fact = ((y + 1) * angle) & 31;
            // refx[idx]: 0 corner, 1.. above; negative idx -> projected left samples
            auto ref = [&](int idx) -> int {
                if (idx < 0) return L[min((idx * inv_angle + 256) >> 9, n)];
                if (idx == 0) return alrs;
                return A[min(idx - 1, 2 * n - 1)];
            };
            if (comp == 0) {
                int acc = 0;
                for (int t = 0; t < 4; ++t) {
                    const int f = filter_flag ? (t == 0 ? 16 - (i_fact >> 1)
                                                        : t == 1 ? 32 - (i_fact >> 1)
                                                                 : t == 2 ? 16 + (i_fact >> 1) : (i_fact >> 1))
                                              : (int)SHT.fc[i_fact][t];
                    acc += f * ref(x + i_idx + t);
                }
                v = min(max((acc + 32) >> 6, 0), 255);
            } else if (i_fact != 0) {
                v = (((32 - i_fact) * ref(x + i_idx + 1) + i_fact * ref(x + i_idx + 2) + 16) >> 5) & 0xFF;
            } else {
                v = ref(x + i_idx + 1) & 0xFF;
            }
        } else {
            const int i_idx = ((x + 1) * angle) >> 5;
            const int i_fact = ((x + 1) * angle) & 31;
            auto ref = [&](int idx) -> int {
                if (idx < 0) {
                    const int t = min((idx * inv_angle + 256) >> 9, n);
                    return t == 0 ? alrs : A[t - 1];
                }
                return L[min(idx, 2 * n)];
            };
            if (comp == 0) {
                int acc = 0;
                for (int t = 0; t < 4; ++t) {
                    const int f = filter_flag ? (t == 0 ? 16 - (i_fact >> 1)
                                                        : t == 1 ? 32 - (i_fact >> 1)
                                                                 : t == 2 ? 16 + (i_fact >> 1) : (i_fact >> 1))
                                              : (int)SHT.fc[i_fact][t];
                    acc += f * ref(y + i_idx + t);
                }
                v = min(max((acc + 32) >> 6, 0), 255);
            } else if (i_fact != 0) {
                v = (((32 - i_fact) * ref(y + i_idx + 1) + i_fact * ref(y + i_idx + 2) + 16) >> 5) & 0xFF;
            } else {
                v = ref(y + i_idx + 1) & 0xFF;
            }
        }
        if (do_pdpc) {
            // intra_predictor.rs:355-757; left[] = L+1, above[] = A
            int rl = 0, rt = 0, wl = 0, wt = 0;
            if (mode == 18 || mode == 50) {
                rl = (int16_t)(L[y + 1] - alrs + v);
                rt = (int16_t)(A[x] - alrs + v);
                wl = mode == 50 ? pdpc_w(n_scale, x) : 0;
                wt = mode == 18 ? pdpc_w(n_scale, y) : 0;
            } else if (mode < 18 && n_scale >= 0) {
                const int dx_int = ((y + 1) * inv_angle + 256) >> 9;
                rt = y < (3 << n_scale) ? A[x + dx_int] : 0;
                wt = pdpc_w(n_scale, y);
            } else if (mode > 50 && n_scale >= 0) {
                const int dy_int = ((x + 1) * inv_angle + 256) >> 9;
                rl = x < (3 << n_scale) ? L[1 + y + dy_int] : 0;
                wl = pdpc_w(n_scale, x);
            }
            v = (int16_t)(rl * wl + rt * wt + (64 - wt - wl) * v + 32) >> 6;
            v = min(max(v, 0), 255);
        }
        sad += emit_sample<FULL>(c, comp + blk, cx + x, cy + y, i, v);
    }
    WSYNC();
    return sad;
}

// ---------------------------------------------------------------------------
// DCT-2 (transformer.rs).  Lane u = lane % N owns basis row T_N[u][.] in
// registers; G = 64/N lane groups walk the rows/columns; the other operand is
// read from LDS as a wave-broadcast.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc) {
#if __has_builtin(__builtin_amdgcn_sdot2)
    typedef short s2 __attribute__((ext_vector_type(2)));
    s2 va, vb;
    va.x = (short)(a & 0xFFFF);
    va.y = (short)(a >> 16);
    vb.x = (short)(b & 0xFFFF);
    vb.y = (short)(b >> 16);
    return __builtin_amdgcn_sdot2(va, vb, acc, false);
#else
    return acc + (int)(short)(a & 0xFFFF) * (int)(short)(b & 0xFFFF) + ((int)a >> 16) * ((int)b >> 16);
#endif
}

// forward: nb residual blocks in r1 ([blk][y][x] i16) -> coefficients in place, via r2;
// transformer.rs:2040-2378
template <int LG>
__device__ void fwd_dct(Ctx c, int nb) {
    constexpr int N = 1 << LG;
    constexpr int G = 64 / N;
    constexpr int HS = N + 1; // r2 row stride
    const int u = LANE & (N - 1);
    const int g = LANE >> LG;
    uint32_t t[N / 2];
    {
        const uint32_t* src = (const uint32_t*)&c.k->dct[LG - 2][u][0];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) t[k] = src[k];
    }
    // stage 1: H[u][y] = (sum_x T[u][x] r[y][x] + d) >> (LG-1)   (:2139-2209); rows of all blocks
#pragma unroll 1
    for (int yy = g; yy < nb * N; yy += G) {
        const uint32_t* row = (const uint32_t*)&SH.r1[yy * N];
        int acc = 0;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) acc = dot2(row[k], t[k], acc);
        const int blk = yy >> LG, y = yy & (N - 1);
        SH.r2[blk * (N * HS) + u * HS + y] = (acc + (1 << (LG - 2))) >> (LG - 1);
    }
    WSYNC();
    // stage 2: C[v][x] = (sum_y T[v][y] H[x][y] + d) >> (LG+6)  (:2246-2316); lane v = u
#pragma unroll 1
    for (int xx = g; xx < nb * N; xx += G) {
        const int blk = xx >> LG, x = xx & (N - 1);
        const int32_t* col = &SH.r2[blk * (N * HS) + x * HS];
        int acc = 0;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            // |T| <= 90 and |H| <= 46410: 24-bit multiplies are exact (v_mad_i32_i24)
            acc += __mul24((int)(short)(t[k] & 0xFFFF), col[2 * k]);
            acc += __mul24((int)t[k] >> 16, col[2 * k + 1]);
        }
        SH.r1[blk * (N * N) + u * N + x] = (int16_t)((acc + (1 << (LG + 5))) >> (LG + 6));
    }
    WSYNC();
}

// inverse: nb transposed dequantised blocks in the lower half of r2 ([blk][x][i], i16) ->
// residuals r1 ([blk][y][x]); the intermediate lives in the upper half of r2.  transformer.rs:2380-2737
template <int LG>
__device__ void inv_dct(Ctx c, int nb) {
    constexpr int N = 1 << LG;
    constexpr int G = 64 / N;
    const int u = LANE & (N - 1);
    const int g = LANE >> LG;
    const int16_t* dqt = (const int16_t*)SH.r2;
    int16_t* vbuf = (int16_t*)SH.r2 + 1024;
    uint32_t t[N / 2]; // Tt[u][i] = T_N[i][u]
    {
        const uint32_t* src = (const uint32_t*)&c.k->dct_t[LG - 2][u][0];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) t[k] = src[k];
    }
    // stage 1 (vertical): V[y][x] = clamp16((sum_i T[i][y] d[i][x] + 64) >> 7); lane y = u
#pragma unroll 1
    for (int xx = g; xx < nb * N; xx += G) {
        const uint32_t* col = (const uint32_t*)&dqt[xx * N]; // dT[blk][x][.]
        int acc = 0;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) acc = dot2(col[k], t[k], acc);
        int v = (acc + 64) >> 7;
        v = min(max(v, -32768), 32767);
        const int blk = xx >> LG, x = xx & (N - 1);
        vbuf[blk * (N * N) + u * N + x] = (int16_t)v;
    }
    WSYNC();
    // stage 2 (horizontal): r[y][x] = (sum_i T[i][x] V[y][i] + 2048) >> 12; lane x = u
#pragma unroll 1
    for (int yy = g; yy < nb * N; yy += G) {
        const uint32_t* row = (const uint32_t*)&vbuf[yy * N];
        int acc = 0;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) acc = dot2(row[k], t[k], acc);
        SH.r1[yy * N + u] = (int16_t)((acc + 2048) >> 12);
    }
    WSYNC();
}

__device__ __forceinline__ void fwd_dct_lg(Ctx c, int lg, int nb) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    switch (lg) {
    case 2: fwd_dct<2>(c, nb); break;
    case 3: fwd_dct<3>(c, nb); break;
    case 4: fwd_dct<4>(c, nb); break;
    default: fwd_dct<5>(c, nb); break;
    }
}
__device__ __forceinline__ void inv_dct_lg(Ctx c, int lg, int nb) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    switch (lg) {
    case 2: inv_dct<2>(c, nb); break;
    case 3: inv_dct<3>(c, nb); break;
    case 4: inv_dct<4>(c, nb); break;
    default: inv_dct<5>(c, nb); break;
    }
}

// ---------------------------------------------------------------------------
// Dependent quantisation (quantizer.rs:338-759) + level cost (block_splitter.rs:415-460)
// ---------------------------------------------------------------------------
__device__ __forceinline__ long long ldq_at(Ctx c, int bits) {
    return bits < 256 ? (long long)SHT.ldq[bits] : c.k->ldq[bits];
}
__device__ __forceinline__ long long lv_at(Ctx c, int a) {
    return a < 256 ? (long long)SHT.lv[a] : c.k->lv[a];
}
template <int CTRL>
__device__ __forceinline__ int dpp_quad(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ int dec_nib(const uint16_t* dec16, int p) {
    const uint16_t* g = dec16 + (p >> 4) * 4;
    const int k = p & 15;
    return ((g[0] >> k) & 1) | (((g[1] >> k) & 1) << 1) | (((g[2] >> k) & 1) << 2) | (((g[3] >> k) & 1) << 3);
}

__device__ __forceinline__ int compose_map(int g2, int g1) {
    // (g2 o g1)(s) = g2[g1[s]]; maps {0..3}->{0..3} packed 2 bits per entry
    int r = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) r |= ((g2 >> (2 * ((g1 >> (2 * s)) & 3))) & 3) << (2 * s);
    return r;
}

// Path costs are kept in 32 bits.  Only cost DIFFERENCES between the four states decide the
// path, and they are bounded: any state reaches any other state's continuation within two
// steps (q_state_trans_table is 2-step complete), and one step costs at most
// 128*65535 + lambda_q*dq_table[1023] < 2^25 (QP 63), so |C_s - C_s'| < 2^26.2.  Subtracting the
// quad minimum every 16 positions therefore keeps every value below 2^26.2 + 16*2^25 < 2^30.
// A zero coefficient has no second branch; it is given the cost 2^29, which can never win
// against branch 0 (K0 <= n0 + 2^25 <= n1 + 2^26.2 + 2^25 < n1 + 2^29) and cannot overflow.
constexpr int kNoBranch = 1 << 29;

// Dependent quantisation of nb transform blocks of side n (nb = 1 luma, 2 = Cb+Cr pair):
// coefficients r1 ([blk][y][x]) -> levels in place; returns the summed level cost
// (block_splitter.rs:436-458).  Scratch: r2, decw.  `*overflow` is set when a level needs a table
// entry >= 1024 (the reference panics there).
//
// Backward pass = 4-state Viterbi equivalent of the reference's memoised DFS (SURVEY.md Q3,
// proven equal to the literal DFS in tests/test_oracle.py).  Per chunk of positions all lanes
// precompute the two branch costs for both values of delta = (state > 1); then ONE lane per
// state and block walks the chunk, exchanging path costs with two DPP quad permutes.
//   shared == true : every wave of the workgroup is in this call with blocks of the same size
//                    (same schedule, see SHW above); wave 0 walks all WPB*nb blocks at once, one
//                    quad of lanes per block, between two workgroup barriers per chunk.
//                    `active == false` = this wave only keeps the barriers company.
//   shared == false: the wave walks its own blocks in quads 0..nb-1 (final pass, tests).
// Forward trace = composition of per-position state maps (prefix scan over lanes), then every
// lane emits its own positions and their level costs.
__device__ __forceinline__ long long quantize(Ctx c, int lg, int nb, bool shared, bool active, int* overflow) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    const DevConst* k = c.k;
    const int n = 1 << lg;
    const int P = n * n;
    const int lgP = 2 * lg;
    const int sh = 8 + lg - 5 + 1; // quantizer.rs:558-569
    const int off = (1 << sh) >> 1;
    const int lsc = k->lsc;
    const uint16_t* scan = k->scan_idx[lg - 2];
    int16_t* tcs = (int16_t*)SH.r2;          // [blk][p]: coefficient in reverse-scan order
    int16_t* qds = (int16_t*)SH.r2 + 1024;   // [blk][p]: |(tc << sh) - off| / lsc
    int32_t* cc = (int32_t*)SH.r1;           // chunk: [blk][CH][6] ints (coefficients are dead after the gather)
    const uint16_t* dec16 = (const uint16_t*)SH.decw; // decisions: [blk][sub-block][state] 16-bit masks
    PROF_MARK(q0_);
    int istar0 = P, istar1 = P;
    if (active) {
        int first0 = P, first1 = P;
        for (int idx = LANE; idx < nb * P; idx += 64) {
            const int blk = idx >> lgP, p = idx & (P - 1);
            const int tc = SH.r1[blk * P + scan[p]];
            int S = (int)((unsigned)tc << sh) - off;
            if (tc < 0) S = -S;
            const int qd = tc == 0 ? 0 : (int)(((unsigned long long)(unsigned)S * k->div_magic) >> 47);
            tcs[idx] = (int16_t)tc;
            qds[idx] = (int16_t)qd;
            if (tc != 0 && (qd >> 1) > 0) {
                if (blk)
                    first1 = min(first1, p);
                else
                    first0 = min(first0, p);
            }
        }
        istar0 = wave_min_i32(first0);
        if (nb == 2) istar1 = wave_min_i32(first1);
    }
    if (LANE == 0) {
        SH.q_istar[0] = istar0;
        SH.q_istar[1] = istar1;
        SH.q_active = active ? 1 : 0;
    }
    PROF_MARK(q1_);
    PROF_ADD2(PH_QPRE, q0_, q1_);
    const int ldq1 = (int)ldq_at(c, 1);
    const int st = LANE & 3;
    const int delta = st > 1 ? 1 : 0;
    const int CH = min(P, nb == 2 ? 32 : 64); // chunk positions per block
    // which block this lane's quad walks: (wave, blk) = (quad / nb, quad % nb) in shared mode
    const int quad = LANE >> 2;
    const int wblk = nb == 2 ? (quad & 1) : 0;
    const int wwave = nb == 2 ? (quad >> 1) : quad;
    const bool walker = shared ? (WAVE == 0 && wwave < WPB) : (quad < nb);
    const Lds* tb = shared ? &SHW[wwave < WPB ? wwave : 0] : &SH;
    const int32_t* wcc = (const int32_t*)tb->r1 + wblk * CH * 6;
    int C = 0;
    int ovf = 0;
    for (int base = P - CH; base >= 0; base -= CH) {
        WSYNC();
        if (active && LANE < nb * CH) {
            // per position: [c0 d0, c1 d0, c0 d1, c1 d1, c0 d0 inside the trailing run, flags]
            const int blk = LANE >= CH ? 1 : 0;
            const int p = base + LANE - blk * CH;
            const int tc = tcs[blk * P + p];
            const int qd = qds[blk * P + p];
            const bool dcn = p == P - 1;
            int flags = 0;
            int c0tz = 0;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                int c0, c1;
                if (tc != 0) {
                    const int a0 = dcn ? (qd >> 1) : ((qd + d) >> 1); // quantizer.rs:378 / :441
                    int q0 = dcn ? (int)(int16_t)(2 * a0 - d) : (a0 > 0 ? 2 * a0 - d : 0);
                    const int a1 = a0 + 1;
                    int q1 = dcn ? (int)(int16_t)(2 * a1 - d) : 2 * a1 - d;
                    if (tc < 0) {
                        q0 = dcn ? (int)(int16_t)(-q0) : -q0;
                        q1 = dcn ? (int)(int16_t)(-q1) : -q1;
                    }
                    const int d0 = abs(tc - ((q0 * lsc + off) >> sh));
                    const int d1 = abs(tc - ((q1 * lsc + off) >> sh));
                    if (a1 + 1 >= 1024) ovf = 1;
                    c0 = (int)(128LL * d0 + ldq_at(c, min(a0 + 1, 1023)));
                    c1 = (int)(128LL * d1 + ldq_at(c, min(a1 + 1, 1023)));
                    flags |= (a0 & 1) << d;              // bit d: parity of a0 -> which successor state
                    if (d == 0) {
                        flags |= (a0 == 0 ? 1 : 0) << 2; // bit 2: a0 == 0 in state class delta 0
                        c0tz = a0 == 0 ? c0 - ldq1 : c0; // bits 0 instead of 1 in the trailing run (:449-453)
                    }
                } else {
                    c0 = ldq1; // zero coefficient outside the trailing run: dq_table[1] (:433)
                    c1 = kNoBranch;
                    if (d == 0) {
                        flags |= 1 << 2;
                        c0tz = 0;
                    }
                }
                cc[LANE * 6 + 2 * d] = c0;
                cc[LANE * 6 + 2 * d + 1] = c1;
            }
            cc[LANE * 6 + 4] = c0tz;
            cc[LANE * 6 + 5] = flags;
        }
        if (shared)
            __syncthreads();
        else
            WSYNC();
        if (walker && (!shared || tb->q_active)) {
            const int wistar = shared ? tb->q_istar[wblk] : (wblk ? istar1 : istar0);
            uint16_t* wdec = (uint16_t*)const_cast<uint32_t*>(tb->decw) + wblk * (P >> 2);
            for (int g16 = CH - 16; g16 >= 0; g16 -= 16) { // one 4x4 sub-block per iteration
                unsigned bits = 0;
#pragma unroll
                for (int kk = 15; kk >= 0; --kk) {
                    const int i = g16 + kk;
                    const int p = base + i;
                    const int2 cv = *(const int2*)&wcc[i * 6 + 2 * delta];
                    const int2 ex = *(const int2*)&wcc[i * 6 + 4]; // (c0 in trailing run, flags)
                    const bool tz = st == 0 && p <= wistar;
                    const bool par = (ex.y >> delta) & 1;
                    const int CA = dpp_quad<0xD8>(C); // C[trans[s][0]]: quad_perm [0,2,1,3]
                    const int CB = dpp_quad<0x72>(C); // C[trans[s][1]]: quad_perm [2,0,3,1]
                    const int K0 = (tz ? ex.x : cv.x) + (par ? CB : CA); // keep a0
                    const int K1 = cv.y + (par ? CA : CB);               // take a0 + 1
                    const bool pick1 = K1 < K0;                            // tie -> a0 (:505)
                    C = pick1 ? K1 : K0;
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        if (!pick1 && tz && ((ex.y >> 2) & 1)) C -= ldq1;
                    }
                    bits |= (pick1 ? 1u : 0u) << kk;
                }
                // renormalise: subtract the quad minimum (decisions depend on differences only)
                int m = min(C, dpp_quad<0xB1>(C));  // quad_perm [1,0,3,2]
                m = min(m, dpp_quad<0x4E>(m));      // quad_perm [2,3,0,1]
                C -= m;
                wdec[((base + g16) >> 4) * 4 + st] = (uint16_t)bits;
            }
        }
        if (shared) __syncthreads();
    }
    WSYNC();
    PROF_MARK(q2_);
    PROF_ADD2(PH_QBACK, q1_, q2_);
    if (!active) return 0;
    // ---- forward trace from state 0 (quantizer.rs:686-721) + level-cost walk ----
    // lanes are split evenly between the blocks; each lane owns `per` consecutive positions
    const int half = nb == 2 ? 32 : 64;
    const int blk = nb == 2 ? (LANE >> 5) : 0;
    const int lane_in = LANE & (half - 1);
    const int per = P >= half ? P / half : 1;
    const int p0 = lane_in * per;
    const bool act = p0 < P;
    const int16_t* btcs = tcs + blk * P;
    const int16_t* bqds = qds + blk * P;
    const uint16_t* bdec = dec16 + blk * (P >> 2);
    int fmap = 0xE4; // identity map
    if (act) {
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p], qd = bqds[p], nib = dec_nib(bdec, p);
            int g = 0;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int dl = sidx > 1 ? 1 : 0;
                int a = 0;
                if (tc != 0) a = ((p == P - 1) ? (qd >> 1) : ((qd + dl) >> 1)) + ((nib >> sidx) & 1);
                g |= ((0x7D28 >> (2 * (2 * sidx + (a & 1)))) & 3) << (2 * sidx);
            }
            fmap = compose_map(g, fmap);
        }
    }
    // inclusive prefix composition across the lanes of a block
    int pre = fmap;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int other = __shfl_up(pre, d, 64);
        if (lane_in >= d) pre = compose_map(pre, other);
    }
    int entry = __shfl_up(pre, 1, 64) & 3; // state after all previous lanes of the block, starting from 0
    if (lane_in == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = P;
    if (act) {
        int state = entry;
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p], qd = bqds[p], nib = dec_nib(bdec, p);
            const int dl = state > 1 ? 1 : 0;
            int q = 0, a = 0;
            if (tc != 0) {
                a = ((p == P - 1) ? (qd >> 1) : ((qd + dl) >> 1)) + ((nib >> state) & 1);
                if (p == P - 1)
                    q = (int)(int16_t)(2 * a - dl); // usize wrap + `as i16` (quantizer.rs:379,391)
                else
                    q = a > 0 ? 2 * a - dl : 0;
                if (tc < 0) q = -q;
            }
            SH.r1[blk * P + scan[p]] = (int16_t)q;
            const int qc = abs(q);
            if (qc == 0) {
                zmask |= 1u << j;
            } else {
                const int aw = (qc + dl) >> 1;
                if (aw >= 1024) ovf = 1;
                sum_nz += lv_at(c, min(aw, 1023));
                fnz = min(fnz, p);
            }
            state = (0x7D28 >> (2 * (2 * state + (a & 1)))) & 3;
        }
    }
    const int pf = group_min_i32(fnz, half); // zeros before a block's first non-zero level cost nothing
    if (act) {
        int nz_after = 0;
        for (int j = 0; j < per; ++j)
            if (((zmask >> j) & 1) && p0 + j > pf) ++nz_after;
        sum_nz += (long long)nz_after * lv_at(c, 0);
    }
    const long long sum = (long long)wave_sum_u64((unsigned long long)sum_nz);
    if (__ballot(ovf != 0) != 0ULL) *overflow = 1;
    WSYNC();
    PROF_MARK(q3_);
    PROF_ADD2(PH_QTRACE, q2_, q3_);
    return sum;
}

// levels r1 (row-major) -> transposed dequantised coefficients in r2 (dT[x][i] = d[i][x]);
// quantizer.rs:761-1079
__device__ __forceinline__ void dequantize_t(Ctx c, int lg, int nb) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    const int n = 1 << lg;
    const int nn = n * n;
    const int sh = 8 + lg - 5 + 1;
    const int off = (1 << sh) >> 1;
    const int lsc = c.k->lsc;
    int16_t* out = (int16_t*)SH.r2;
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg), ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        int v = ((int)SH.r1[i] * lsc + off) >> sh;
        v = min(max(v, -32768), 32767);
        out[blk * nn + x * n + y] = (int16_t)v;
    }
    WSYNC();
}

// ---------------------------------------------------------------------------
// RD search building blocks (block_splitter.rs)
// ---------------------------------------------------------------------------
struct CompCost {
    unsigned long long ssd;
    long long level;
};

// predict -> T -> Q -> DQ -> IT -> recon (+SSD) of the luma block (comp 0) or of the Cb+Cr pair
// (comp 1) of a TU (block_splitter.rs:146-185); returns SSD and level cost summed over the blocks.
// The TU's reference samples must be current (build_refs).  lev0 != nullptr: the final pass --
// the levels go to plane position lev0 (and lev1 for Cr), row stride lev_stride, and samples whose
// reconstruction differs from what the search left in the tile are counted in *changed.
__device__ __noinline__ CompCost code_component(Ctx c, int comp, int tx, int ty, int tlg, int mode, bool shared,
                                                bool active, int16_t* lev0, int16_t* lev1, int lev_stride,
                                                int* changed, int* overflow) {
    c = uni(c);
    comp = uni(comp);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    mode = uni(mode);
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int lg = tlg - cs;
    const int n = 1 << lg;
    const int nn = n * n;
    const int cx = tx >> cs, cy = ty >> cs;
    CompCost r;
    if (!active) { // keep the shared-Viterbi barriers company (all waves run the same schedule)
        r.level = quantize(c, lg, nb, shared, false, overflow);
        r.ssd = 0;
        return r;
    }
    PROF_MARK(t0_);
    predict<true>(c, comp, tx, ty, tlg, mode);
    PROF_MARK(t1_);
    fwd_dct_lg(c, lg, nb);
    PROF_MARK(t2_);
    r.level = quantize(c, lg, nb, shared, true, overflow);
    PROF_MARK(t3_);
    if (lev0 != nullptr && c.write)
        for (int i = LANE; i < nb * nn; i += 64) {
            const int blk = i >> (2 * lg), ii = i & (nn - 1);
            (blk ? lev1 : lev0)[(size_t)(ii >> lg) * lev_stride + (ii & (n - 1))] = SH.r1[i];
        }
    dequantize_t(c, lg, nb);
    PROF_MARK(t4_);
    inv_dct_lg(c, lg, nb);
    PROF_MARK(t5_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    PROF_ADD2(PH_FDCT, t1_, t2_);
    PROF_ADD2(PH_DEQ, t3_, t4_);
    PROF_ADD2(PH_IDCT, t4_, t5_);
    unsigned int part = 0;
    int diff = 0;
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg), ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int pc = comp + blk;
        int v = (int16_t)((int)c.pred_scratch[i] + (int)SH.r1[i]); // pred as i16 + res, clamp (:178)
        v = min(max(v, 0), 255);
        if (changed != nullptr && v != rec_get(pc, cx + x, cy + y)) ++diff;
        rec_put(pc, cx + x, cy + y, v);
        const int d = v - org_get(c, pc, cx + x, cy + y);
        part += (unsigned)(d * d);
    }
    r.ssd = wave_sum_u64((unsigned long long)part);
    if (changed != nullptr) *changed += wave_sum_i32(diff);
    WSYNC();
    PROF_MARK(t6_);
    PROF_ADD2(PH_RECON, t5_, t6_);
    return r;
}

// predict + SAD of the luma block or the Cb+Cr pair (block_splitter.rs:64-108); nothing is stored
__device__ __noinline__ unsigned int sad_component(Ctx c, int comp, int tx, int ty, int tlg, int mode) {
    c = uni(c);
    comp = uni(comp);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    mode = uni(mode);
    PROF_MARK(t0_);
    const int part = predict<false>(c, comp, tx, ty, tlg, mode);
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    return (unsigned)wave_sum_i32(part);
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
        return SH.left_mode[y >> 2];
    }
    *exists = false;
    return PLANAR;
}

// mode class index for the header-bit table: 0 planar, 1..5 mpm_idx, 6..66 remainder
// (ctu.rs:1498-1635)
__device__ __noinline__ int mpm_class(Ctx c, int bx, int by, int lg, int mode) {
    c = uni(c);
    bx = uni(bx);
    by = uni(by);
    lg = uni(lg);
    mode = uni(mode);
    if (mode == PLANAR) return 0;
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
    if (k0 == mode) return 1;
    if (k1 == mode) return 2;
    if (k2 == mode) return 3;
    if (k3 == mode) return 4;
    if (k4 == mode) return 5;
    // remainder = mode - 1 - #(candidates below mode) after sorting (:1613-1628)
    const int smaller = (k0 < mode) + (k1 < mode) + (k2 < mode) + (k3 < mode) + (k4 < mode);
    return 6 + (mode - 1 - smaller);
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

struct LeafResult {
    float cost;
    int luma_mode;
    int chroma_mode; // TU-array chroma prediction mode
};

// SSD and level cost of the luma and of the chroma pair of one evaluated candidate.  Evaluations
// are deterministic functions of (block, mode, neighbourhood[, luma recon for CCLM]), so where the
// reference re-runs an evaluation it has already done (block_splitter.rs:1040,1068-1075) the
// parts are re-used and only the cost is re-assembled.
struct EvalParts {
    unsigned long long ssd_y, ssd_c;
    long long lvl_y, lvl_c;
};

// get_intra_pred_cost (block_splitter.rs:110-474) from already evaluated parts, modes [ml, mc, mc]
__device__ __forceinline__ float assemble_cost(const Ctx& c, int tree, int cls, int mc, const EvalParts& e) {
    const bool single = tree == TREE_SINGLE;
    const int cc = (single && mc >= LT_CCLM) ? 1 + (mc - LT_CCLM) : 0;
    const unsigned long long ssd = e.ssd_y + (single ? e.ssd_c : 0ULL);
    const long long level = e.lvl_y + (single ? e.lvl_c : 0LL) + c.k->hb_luma[single ? 0 : 1][cc][cls];
    return rd_cost(ssd, level, c.k->lambda_rd);
}

// get_chroma_intra_pred_cost (block_splitter.rs:524-780) from already evaluated parts
__device__ __forceinline__ float assemble_chroma_cost(const Ctx& c, int mc, const EvalParts& e) {
    const long long level = e.lvl_c + c.k->hb_chroma[mc >= LT_CCLM ? 1 + (mc - LT_CCLM) : 0];
    return rd_cost(e.ssd_c, level, c.k->lambda_rd_chroma);
}

// evaluate the chroma pair with mode mc into e.ssd_c / e.lvl_c
__device__ __forceinline__ CompCost eval_chroma(Ctx c, int bx, int by, int lg, int mc, bool active, int* overflow) {
    return code_component(c, 1, bx, by, lg, mc, true, active, nullptr, nullptr, 0, nullptr, overflow);
}

struct FullRes {
    float cost;
    EvalParts e;
};

// get_intra_pred_cost (block_splitter.rs:110-474) for modes [ml, mc, mc], with its parts
__device__ __noinline__ FullRes full_cost(Ctx c, int tree, int bx, int by, int lg, int ml, int mc, bool active,
                                         int* overflow) {
    c = uni(c);
    tree = uni(tree);
    bx = uni(bx);
    by = uni(by);
    lg = uni(lg);
    ml = uni(ml);
    mc = uni(mc);
    EvalParts p = {0, 0, 0, 0};
    {
        const CompCost r = code_component(c, 0, bx, by, lg, ml, true, active, nullptr, nullptr, 0, nullptr, overflow);
        p.ssd_y = r.ssd;
        p.lvl_y = r.level;
    }
    if (tree == TREE_SINGLE) {
        const CompCost cc = eval_chroma(c, bx, by, lg, mc, active, overflow);
        p.ssd_c = cc.ssd;
        p.lvl_c = cc.level;
    }
    FullRes out;
    out.e = p;
    // a skipped evaluation is f32::MAX in the reference
    out.cost = active ? assemble_cost(c, tree, mpm_class(c, bx, by, lg, ml), mc, p) : 3.40282347e+38f;
    return out;
}

// get_intra_pred_aux_cost (block_splitter.rs:64-108) for modes [m; 3]
__device__ __forceinline__ float aux_cost(Ctx c, int tree, int bx, int by, int lg, int m) {
    unsigned long long sad = sad_component(c, 0, bx, by, lg, m);
    if (tree == TREE_SINGLE) sad += sad_component(c, 1, bx, by, lg, m);
    return (float)sad;
}

// get_chroma_intra_pred_cost (block_splitter.rs:524-780)
__device__ __forceinline__ float chroma_full_cost(Ctx c, int bx, int by, int lg, int mc, EvalParts& e, int* overflow) {
    const CompCost cc = eval_chroma(c, bx, by, lg, mc, true, overflow);
    e.ssd_c = cc.ssd;
    e.lvl_c = cc.level;
    return assemble_chroma_cost(c, mc, e);
}

// get_chroma_intra_pred_aux_cost (block_splitter.rs:476-522)
__device__ __forceinline__ float chroma_aux_cost(Ctx c, int bx, int by, int lg, int mc) {
    return (float)(unsigned long long)sad_component(c, 1, bx, by, lg, mc);
}

__device__ __forceinline__ int pick_cclm(float lt, float t, float l) {
    // block_splitter.rs:847-854
    if (lt <= t && lt <= l) return LT_CCLM;
    if (t <= l) return T_CCLM;
    return L_CCLM;
}

// Re-create the reconstruction of a decided block by running its evaluation again with the
// solo Viterbi (no workgroup barriers: which blocks need this differs from wave to wave).  The
// neighbourhood is unchanged, so the result equals what the evaluation produced the first time;
// this replaces the reference's cache_reconsts / restore_reconsts copies (block_splitter.rs:
// 807-840, 1085-1145) without keeping saved planes in LDS.
__device__ __noinline__ void regen_block(Ctx c, int bx, int by, int lg, int luma_mode, int chroma_mode, bool luma,
                                         bool chroma, int* overflow) {
    c = uni(c);
    bx = uni(bx);
    by = uni(by);
    lg = uni(lg);
    luma_mode = uni(luma_mode);
    chroma_mode = uni(chroma_mode);
    if (luma) {
        build_refs(c, 0, bx, by, lg);
        code_component(c, 0, bx, by, lg, luma_mode, false, true, nullptr, nullptr, 0, nullptr, overflow);
    }
    if (chroma) {
        if (chroma_mode < LT_CCLM) build_refs(c, 1, bx, by, lg);
        code_component(c, 1, bx, by, lg, chroma_mode, false, true, nullptr, nullptr, 0, nullptr, overflow);
    }
}

// leaf search of a DUAL_TREE_CHROMA block (block_splitter.rs:794-885); lg = luma log2 (3)
__device__ __noinline__ LeafResult leaf_chroma(Ctx c, int bx, int by, int lg, int dm_mode, int* overflow) {
    c = uni(c);
    bx = uni(bx);
    by = uni(by);
    lg = uni(lg);
    dm_mode = uni(dm_mode);
    build_refs(c, 1, bx, by, lg);
    const float lt = chroma_aux_cost(c, bx, by, lg, LT_CCLM);
    const float t = chroma_aux_cost(c, bx, by, lg, T_CCLM);
    const float l = chroma_aux_cost(c, bx, by, lg, L_CCLM);
    const int cclm_mode = pick_cclm(lt, t, l);
    EvalParts e_cclm = {0, 0, 0, 0}, e_dm = {0, 0, 0, 0};
    const float cclm_cost = chroma_full_cost(c, bx, by, lg, cclm_mode, e_cclm, overflow);
    const float cur = chroma_full_cost(c, bx, by, lg, dm_mode, e_dm, overflow);
    LeafResult r;
    r.luma_mode = 0;
    const float mn = fminf(cclm_cost, fminf(cur, 3.40282347e+38f));
    r.cost = mn;
    if (cur == mn) {
        r.chroma_mode = dm_mode;
    } else {
        r.chroma_mode = cclm_mode;
        regen_block(c, bx, by, lg, 0, cclm_mode, false, true, overflow); // :869-873 restore_reconsts
    }
    return r;
}

// leaf search of a SINGLE_TREE / DUAL_TREE_LUMA block (block_splitter.rs:886-1078)
__device__ __noinline__ LeafResult leaf_luma(Ctx c, int tree, int bx, int by, int lg, int* overflow) {
    c = uni(c);
    tree = uni(tree);
    bx = uni(bx);
    by = uni(by);
    lg = uni(lg);
    build_refs(c, 0, bx, by, lg);
    if (tree == TREE_SINGLE) build_refs(c, 1, bx, by, lg);
    float cost_planar = 0.f, cost_dc = 0.f;
    EvalParts e_planar = {0, 0, 0, 0}, e_dc = {0, 0, 0, 0};
    float min_dir_cost = 3.40282347e+38f;
    int min_dir_mode = 2;
    for (int i = 0; i < 15; ++i) {
        // {0,1,2,7,13,18,23,29,34,39,45,50,55,60,66} (:887), 7 bits each
        const int m = i < 8 ? (int)((0x3A5C90D0E08080ULL >> (7 * i)) & 127) : (int)((0x109E3764B53A2ULL >> (7 * (i - 8))) & 127);
        if (m <= 1) {
            const FullRes fr = full_cost(c, tree, bx, by, lg, m, m, true, overflow);
            if (m == 0) {
                cost_planar = fr.cost;
                e_planar = fr.e;
            } else {
                cost_dc = fr.cost;
                e_dc = fr.e;
            }
        } else {
            const float v = aux_cost(c, tree, bx, by, lg, m);
            if (v < min_dir_cost) { // first minimum (:899-904)
                min_dir_cost = v;
                min_dir_mode = m;
            }
        }
    }
    // step_search(mode, 2, cost, aux=true) (:905-973)
    int cur_mode = min_dir_mode;
    float cur_cost = min_dir_cost;
    for (int step = 2; step > 0; step >>= 1) {
        const float c0 = cur_mode < 2 + step ? 3.40282347e+38f : aux_cost(c, tree, bx, by, lg, cur_mode - step);
        const float c1 = cur_mode + step > 66 ? 3.40282347e+38f : aux_cost(c, tree, bx, by, lg, cur_mode + step);
        const float mn = fminf(fminf(cur_cost, c0), c1);
        if (cur_cost == mn) {
        } else if (c0 == mn) {
            cur_mode -= step;
            cur_cost = c0;
        } else {
            cur_mode += step;
            cur_cost = c1;
        }
    }
    // step_search(mode, 1, _, aux=false) (:974)
    EvalParts e_dir = {0, 0, 0, 0};
    {
        // out-of-range neighbours are "evaluated" inactive: the wave still walks the schedule so
        // that the workgroup's shared Viterbi barriers stay aligned; the result is f32::MAX
        const FullRes f = full_cost(c, tree, bx, by, lg, cur_mode, cur_mode, true, overflow);
        const FullRes f0 = full_cost(c, tree, bx, by, lg, cur_mode - 1, cur_mode - 1, !(cur_mode < 3), overflow);
        const FullRes f1 = full_cost(c, tree, bx, by, lg, cur_mode + 1, cur_mode + 1, !(cur_mode + 1 > 66), overflow);
        cur_cost = f.cost;
        e_dir = f.e;
        const float c0 = f0.cost, c1 = f1.cost;
        const float mn = fminf(fminf(cur_cost, c0), c1);
        if (cur_cost == mn) {
        } else if (c0 == mn) {
            cur_mode -= 1;
            cur_cost = c0;
            e_dir = f0.e;
        } else {
            cur_mode += 1;
            cur_cost = c1;
            e_dir = f1.e;
        }
    }
    // min of {planar, DC, dir}, first index wins (:975-978)
    float min_cost = fminf(cur_cost, fminf(cost_dc, fminf(cost_planar, 3.40282347e+38f)));
    int mode;
    EvalParts e_win;
    if (cost_planar == min_cost) {
        mode = 0;
        e_win = e_planar;
    } else if (cost_dc == min_cost) {
        mode = 1;
        e_win = e_dc;
    } else {
        mode = cur_mode;
        e_win = e_dir;
    }
    // luma re-run with the winner (:989-1037): puts the winner's luma reconstruction into the tile
    code_component(c, 0, bx, by, lg, mode, true, true, nullptr, nullptr, 0, nullptr, overflow);
    LeafResult r;
    r.luma_mode = mode;
    r.chroma_mode = mode;
    if (tree != TREE_DUAL_LUMA) {
        // :1040 get_chroma_intra_pred_cost(mode) repeats the winner's chroma evaluation: re-use it
        const float cur = assemble_chroma_cost(c, mode, e_win);
        const float lt = chroma_aux_cost(c, bx, by, lg, LT_CCLM);
        const float t = chroma_aux_cost(c, bx, by, lg, T_CCLM);
        const float l = chroma_aux_cost(c, bx, by, lg, L_CCLM);
        const int cclm_mode = pick_cclm(lt, t, l);
        EvalParts e_cclm = e_win;
        const float cclm_cost = chroma_full_cost(c, bx, by, lg, cclm_mode, e_cclm, overflow);
        const float mn = fminf(cclm_cost, fminf(cur, 3.40282347e+38f));
        const bool dm_wins = cur == mn;
        // :1062-1072 final get_intra_pred_cost: luma = the re-run above; the chroma pair is the DM
        // evaluation (re-done only to put its reconstruction back when DM wins; an inactive walk of
        // the schedule otherwise) or the CCLM evaluation just made
        eval_chroma(c, bx, by, lg, mode, dm_wins, overflow);
        const int cls = mpm_class(c, bx, by, lg, mode);
        if (dm_wins) {
            min_cost = assemble_cost(c, tree, cls, mode, e_win);
        } else {
            r.chroma_mode = cclm_mode;
            min_cost = assemble_cost(c, tree, cls, cclm_mode, e_cclm);
        }
    } else if (mode <= 1) {
        // :1073-1076 repeats the luma evaluation just re-run: same parts, same cost
        min_cost = assemble_cost(c, tree, mpm_class(c, bx, by, lg, mode), mode, e_win);
    }
    r.cost = min_cost;
    return r;
}

// ---------------------------------------------------------------------------
// Decision maps and recon save/restore
// ---------------------------------------------------------------------------
__device__ __noinline__ void fill_maps(Ctx c, int bx, int by, int lg, int luma_mode, int chroma_mode,
                          bool luma, bool chroma) {
    c = uni(c);
    bx = uni(bx);
    by = uni(by);
    lg = uni(lg);
    luma_mode = uni(luma_mode);
    chroma_mode = uni(chroma_mode);
    const int n4 = (1 << lg) >> 2;
    if (luma)
        for (int i = LANE; i < n4 * n4; i += 64) {
            const int idx = ((by >> 2) + i / n4) * 8 + (bx >> 2) + i % n4;
            SH.cu_log2[idx] = (uint8_t)lg;
            SH.luma_mode[idx] = (uint8_t)luma_mode;
        }
    if (chroma) {
        const int n8 = max(n4 >> 1, 1);
        for (int i = LANE; i < n8 * n8; i += 64)
            SH.chroma_mode[((by >> 3) + i / n8) * 4 + (bx >> 3) + i % n8] = (uint8_t)chroma_mode;
    }
    WSYNC();
}

// ---------------------------------------------------------------------------
// split_ct (block_splitter.rs:782-1154): exhaustive quad-tree search of one CTU as an
// explicit depth-first walk (level 0 = 32x32 ... level 2 = 8x8; an 8x8 node's split is
// four DUAL_TREE_LUMA 4x4 leaves + one DUAL_TREE_CHROMA 4x4 leaf, ctu.rs:1990-2063).
// Per-level state lives in LDS (wave-uniform).
// ---------------------------------------------------------------------------
__device__ __noinline__ float split_node8(Ctx c, int bx, int by, int* overflow) {
    c = uni(c);
    bx = uni(bx);
    by = uni(by);
    float split_cost = 0.0f;
    for (int i = 0; i < 4; ++i) {
        const int cxx = bx + (i & 1) * 4, cyy = by + (i >> 1) * 4;
        const LeafResult r = leaf_luma(c, TREE_DUAL_LUMA, cxx, cyy, 2, overflow);
        fill_maps(c, cxx, cyy, 2, r.luma_mode, 0, true, false);
        split_cost = split_cost + r.cost;
    }
    // DM = luma mode of the CU covering the parent's centre (block_splitter.rs:795-805)
    const int dm = SH.luma_mode[((by + 4) >> 2) * 8 + ((bx + 4) >> 2)];
    const LeafResult r = leaf_chroma(c, bx, by, 3, dm, overflow);
    fill_maps(c, bx, by, 3, 0, r.chroma_mode, false, true);
    return split_cost + r.cost;
}

__device__ float split_ct_ctu(Ctx& c, int max_depth, int* overflow) {
    int level = 0;
    int bx = 0, by = 0;
    float ret = 0.0f;
    for (;;) {
        // ---- enter node (bx, by) at `level` ----
        const int lg = 5 - level;
        const LeafResult ns = leaf_luma(c, TREE_SINGLE, bx, by, lg, overflow);
        fill_maps(c, bx, by, lg, ns.luma_mode, ns.chroma_mode, true, true);
        if (level == 0) c.cu32_mode = ns.luma_mode;
        bool done;
        if (max_depth - level == 0) {
            ret = ns.cost;
            done = true;
        } else {
            if (LANE == 0) {
                SH.ns_cost[level] = ns.cost;
                SH.ns_luma[level] = (uint8_t)ns.luma_mode;
                SH.ns_chroma[level] = (uint8_t)ns.chroma_mode;
                SH.split_cost[level] = 0.0f;
                SH.child[level] = 0;
            }
            WSYNC();
            if (lg > 3) {
                level += 1; // descend into child 0 (same top-left corner)
                continue;
            }
            const float sc = split_node8(c, bx, by, overflow);
            if (sc > ns.cost) { // :1125-1145
                regen_block(c, bx, by, lg, ns.luma_mode, ns.chroma_mode, true, true, overflow);
                fill_maps(c, bx, by, lg, ns.luma_mode, ns.chroma_mode, true, true);
                ret = ns.cost;
            } else {
                ret = sc;
            }
            done = true;
        }
        // ---- return `ret` from a finished node to its ancestors ----
        while (done) {
            if (level == 0) return ret;
            const int pl = level - 1;
            const int psz = 1 << (5 - pl);
            const int pbx = bx & ~(psz - 1), pby = by & ~(psz - 1);
            const float acc = SH.split_cost[pl] + ret; // children in z-order, f32 (:1116-1123)
            const int ch = SH.child[pl] + 1;
            WSYNC();
            if (LANE == 0) {
                SH.split_cost[pl] = acc;
                SH.child[pl] = (uint8_t)ch;
            }
            WSYNC();
            if (ch < 4) { // next sibling
                bx = pbx + (ch & 1) * (psz >> 1);
                by = pby + (ch >> 1) * (psz >> 1);
                done = false;
            } else { // parent complete
                const float nsc = SH.ns_cost[pl];
                bx = pbx;
                by = pby;
                level = pl;
                if (acc > nsc) {
                    regen_block(c, bx, by, 5 - pl, SH.ns_luma[pl], SH.ns_chroma[pl], true, true, overflow);
                    fill_maps(c, bx, by, 5 - pl, SH.ns_luma[pl], SH.ns_chroma[pl], true, true);
                    ret = nsc;
                } else {
                    ret = acc;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Final pass (ctu_encoder.rs:1421-1461) in coding order; writes levels to HBM
// ---------------------------------------------------------------------------
// comp 0 = luma block, comp 1 = the Cb+Cr pair of the TU at (tx, ty)
__device__ __noinline__ void final_component(Ctx c, const PicBufs& pb, int comp, int tx, int ty, int tlg, int mode,
                                             int* overflow) {
    c = uni(c);
    comp = uni(comp);
    tx = uni(tx);
    ty = uni(ty);
    tlg = uni(tlg);
    mode = uni(mode);
    const int cs = comp ? 1 : 0;
    const int stride = c.k->W >> cs;
    const size_t at = (size_t)((c.ctu_y + ty) >> cs) * stride + ((c.ctu_x + tx) >> cs);
    if (mode < LT_CCLM) build_refs(c, comp, tx, ty, tlg);
    int changed = 0;
    code_component(c, comp, tx, ty, tlg, mode, false, true, pb.lev[comp] + at, comp ? pb.lev[2] + at : nullptr, stride,
                   &changed, overflow);
    if (changed && LANE == 0 && c.write) atomicAdd(c.mismatch, (unsigned long long)changed);
}

// coding order = z-order over the 4x4 units; a CU is emitted at its top-left unit
__device__ void final_pass_ctu(Ctx c, const PicBufs& pb, int* overflow) {
    for (int z = 0; z < 64; ++z) {
        const int x4 = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
        const int y4 = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
        const int lg = SH.cu_log2[y4 * 8 + x4];
        const int bx = x4 * 4, by = y4 * 4;
        const int sz = 1 << lg;
        if ((bx & (sz - 1)) == 0 && (by & (sz - 1)) == 0) {
            const int ml = SH.luma_mode[y4 * 8 + x4];
            final_component(c, pb, 0, bx, by, lg, ml, overflow);
            if (lg >= 3) {
                const int mc = SH.chroma_mode[(by >> 3) * 4 + (bx >> 3)];
                final_component(c, pb, 1, bx, by, lg, mc, overflow);
            }
        }
        if (lg == 2 && (z & 3) == 3) { // after the fourth 4x4 luma CU: the 8x8's chroma CU
            const int pbx = bx & ~7, pby = by & ~7;
            const int mc = SH.chroma_mode[(pby >> 3) * 4 + (pbx >> 3)];
            final_component(c, pb, 1, pbx, pby, 3, mc, overflow);
        }
    }
}

// ---------------------------------------------------------------------------
// CTU entry: load, search, final pass, store
// ---------------------------------------------------------------------------
__device__ void load_tables(Ctx c) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        SHT.ldq[i] = (int32_t)c.k->ldq[i];
        SHT.lv[i] = (int32_t)c.k->lv[i];
    }
    for (int i = threadIdx.x; i < 128; i += blockDim.x) ((int8_t*)SHT.fc)[i] = ((const int8_t*)c.k->fc)[i];
    __syncthreads();
}

__device__ void encode_ctu(Ctx& c, const PicBufs& pb, int ctu_col, int ctu_row, int* overflow) {
    const DevConst* k = c.k;
    const int W = k->W, H = k->H;
    const int Wc = W >> 1;
    c.ctu_x = ctu_col * 32;
    c.ctu_y = ctu_row * 32;
    c.cu32_mode = PLANAR;
#ifdef WRENC_PROFILE
    if (threadIdx.x < PH_COUNT) s_prof[threadIdx.x] = 0;
    __syncthreads();
#endif
    PROF_MARK(tt0_);
    load_tables(c);
    // neighbour border of the reconstruction: row -1 (x = -4..67) and columns -4..-1
    for (int i = LANE; i < 72; i += 64) {
        const int gx = c.ctu_x - 4 + i, gy = c.ctu_y - 1;
        SH.recYtop[i] = (gx >= 0 && gx < W && gy >= 0) ? pb.rec[0][(size_t)gy * W + gx] : 0;
    }
    for (int i = LANE; i < 32 * 4; i += 64) {
        const int y = i >> 2, x = (i & 3) - 4;
        const int gx = c.ctu_x + x, gy = c.ctu_y + y;
        SH.recY[y * 36 + x + 4] = gx >= 0 ? pb.rec[0][(size_t)gy * W + gx] : 0;
    }
    for (int comp = 1; comp < 3; ++comp) {
        for (int i = LANE; i < 40; i += 64) {
            const int gx = (c.ctu_x >> 1) - 4 + i, gy = (c.ctu_y >> 1) - 1;
            SH.recCtop[comp - 1][i] = (gx >= 0 && gx < Wc && gy >= 0) ? pb.rec[comp][(size_t)gy * Wc + gx] : 0;
        }
        for (int i = LANE; i < 16 * 4; i += 64) {
            const int y = i >> 2, x = (i & 3) - 4;
            const int gx = (c.ctu_x >> 1) + x, gy = (c.ctu_y >> 1) + y;
            SH.recC[comp - 1][y * 20 + x + 4] = gx >= 0 ? pb.rec[comp][(size_t)gy * Wc + gx] : 0;
        }
    }
    // tile.rs:49-58: planes start at zero
    for (int i = LANE; i < 32 * 32; i += 64) SH.recY[(i >> 5) * 36 + (i & 31) + 4] = 0;
    for (int comp = 1; comp < 3; ++comp)
        for (int i = LANE; i < 256; i += 64) SH.recC[comp - 1][(i >> 4) * 20 + (i & 15) + 4] = 0;
    if (LANE < 8)
        SH.left_mode[LANE] =
            c.ctu_x > 0 ? pb.luma_mode[(size_t)((c.ctu_y >> 2) + LANE) * (W >> 2) + (c.ctu_x >> 2) - 1] : 0;
    WSYNC();
    (void)H;
    const float cost = split_ct_ctu(c, k->max_depth, overflow);
    final_pass_ctu(c, pb, overflow);
    // store recon + decisions
    if (c.write) {
    for (int i = LANE; i < 1024 / 4; i += 64) {
        const int y = i >> 3, x4 = (i & 7) * 4;
        *(uint32_t*)&pb.rec[0][(size_t)(c.ctu_y + y) * W + c.ctu_x + x4] = *(const uint32_t*)&SH.recY[y * 36 + x4 + 4];
    }
    for (int comp = 1; comp < 3; ++comp)
        for (int i = LANE; i < 256 / 4; i += 64) {
            const int y = i >> 2, x4 = (i & 3) * 4;
            *(uint32_t*)&pb.rec[comp][(size_t)((c.ctu_y >> 1) + y) * Wc + (c.ctu_x >> 1) + x4] =
                *(const uint32_t*)&SH.recC[comp - 1][y * 20 + x4 + 4];
        }
    {
        const int i = LANE; // 64 4x4 units
        const size_t o = (size_t)((c.ctu_y >> 2) + (i >> 3)) * (W >> 2) + (c.ctu_x >> 2) + (i & 7);
        pb.cu_log2[o] = SH.cu_log2[i];
        pb.luma_mode[o] = SH.luma_mode[i];
        if (i < 16) {
            const size_t oc = (size_t)((c.ctu_y >> 3) + (i >> 2)) * (W >> 3) + (c.ctu_x >> 3) + (i & 3);
            pb.chroma_mode[oc] = SH.chroma_mode[i];
        }
        if (i == 0) pb.ctu_cost[ctu_row * k->ctu_cols + ctu_col] = cost;
    }
    }
#ifdef WRENC_PROFILE
    PROF_MARK(tt1_);
    PROF_ADD2(PH_TOTAL, tt0_, tt1_);
    __syncthreads();
    if (threadIdx.x < PH_COUNT) atomicAdd(&g_prof[threadIdx.x], s_prof[threadIdx.x]);
#endif
}

} // namespace wrenc
