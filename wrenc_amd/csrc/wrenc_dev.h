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
    int32_t ang_tab[67];      // per mode: intraPredAngle (low half) | invAngle (high half), read with one scalar load
    int8_t fc[32][4];         // common.rs:153
};

// Pointers that are loaded from memory (PicBufs) lose their address space; these casts tell the
// compiler they are global memory, so that it emits global_* instead of flat_* accesses.
#define GLOBAL_AS __attribute__((address_space(1)))
// The constant block is written by the host before the launch and never during it.
#define CONST_AS __attribute__((address_space(4)))
#define AS_GLOBAL(T, p) ((GLOBAL_AS T*)(p))

// Per-wave global scratch: 1 KB of prediction bytes, then kReconSlots saved reconstructions
// (slot 0: best candidate of the running leaf; 1 + level: unsplit candidate of the open node at
// that tree level), each 1024 B luma + 2 x 256 B chroma.
constexpr int kReconSlots = 4;
constexpr int kSlotBytes = 1536;
constexpr int kWaveScratch = 1024 + kReconSlots * kSlotBytes;

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
#define LANE ((int)(threadIdx.x & 63))
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// SSD and level cost of the luma block and of the chroma pair of one evaluated candidate
struct EvalParts {
    uint32_t ssd_y, ssd_c; // <= 1024 * 255^2: fits 32 bits
    long long lvl_y, lvl_c;
};


// A wave-uniform field of the coroutine state, resident in LDS: reads come back through
// readfirstlane (scalar registers, scalar branches).
template <class T>
struct UF {
    T v;
    __device__ __forceinline__ T get() const {
        if constexpr (sizeof(T) == 8) {
            const unsigned long long x = (unsigned long long)v;
            return (T)(((unsigned long long)(unsigned)uni((int)(x >> 32)) << 32) | (unsigned)uni((int)x));
        } else if constexpr (sizeof(T) == 4 && !(T(0.5f) == T(0))) {
            return __int_as_float(uni(__float_as_int((float)v)));
        } else {
            return (T)uni((int)v);
        }
    }
    // every lane stores the same value to the same address (one LDS pass; keeps the control flow
    // free of lane predicates so that all of it stays scalar)
    __device__ __forceinline__ void set(T x) { v = x; }
    __device__ __forceinline__ operator T() const { return get(); }
    __device__ __forceinline__ UF& operator=(T x) {
        set(x);
        return *this;
    }
    __device__ __forceinline__ UF& operator=(const UF& o) {
        set(o.get());
        return *this;
    }
    __device__ __forceinline__ UF& operator+=(int x) {
        set((T)(get() + x));
        return *this;
    }
    __device__ __forceinline__ UF& operator-=(int x) {
        set((T)(get() - x));
        return *this;
    }
    __device__ __forceinline__ UF& operator>>=(int x) {
        set((T)(get() >> x));
        return *this;
    }
    __device__ __forceinline__ UF& operator++() {
        set((T)(get() + 1));
        return *this;
    }
};

// EvalParts as kept in the coroutine state
struct EvalPartsU {
    UF<uint32_t> ssd_y, ssd_c;
    UF<long long> lvl_y, lvl_c;
    __device__ __forceinline__ EvalParts get() const {
        EvalParts e;
        e.ssd_y = ssd_y;
        e.ssd_c = ssd_c;
        e.lvl_y = lvl_y;
        e.lvl_c = lvl_c;
        return e;
    }
};

// Leaf search state (block_splitter.rs:794-1078 as a state machine, see leaf_step).
struct LeafSt {
    UF<uint8_t> cont;                   // where to continue with the result of the pending request
    UF<uint8_t> op_ml, op_mc, op_act;   // modes / activity of the pending full evaluation
    UF<uint8_t> tree, bx, by, lg;
    UF<uint8_t> need_refs0, need_refs1; // reference samples of the block not built yet (luma / chroma pair)
    UF<uint8_t> step;
    UF<uint8_t> cur_mode, best_mode, mode, cclm_mode, dm_mode, dm_wins;
    UF<uint8_t> luma_mode, chroma_mode; // result
    UF<uint8_t> best_cls;               // header-bit class (mpm_class) of the best luma mode
    UF<uint8_t> need_save, tile_best;   // best candidate's reconstruction: not saved yet / still in the tile
    UF<float> best_cost;                // best of {planar, DC} so far / of {planar, DC, dir}
    UF<float> cur_cost, c0;
    UF<float> cost;                     // result
    EvalPartsU e_best;
};

// CTU search + final pass state (see ctu_step)
struct CtuSt {
    UF<uint8_t> cont, in_leaf;
    UF<uint8_t> level, bx, by, lg, max_depth;
    UF<uint8_t> i8, z, rl, rc;       // 4x4 child index, final-pass z-order index, regen modes
    UF<uint8_t> rbx, rby, rlg;       // regen block
    UF<uint8_t> ns_luma_cur, ns_chroma_cur;
    UF<uint8_t> pend, pbx, pby, plg, pslot; // reconstruction save to attach to the next request
    UF<float> ret, ns_cost_cur, split8, ctu_cost;
    LeafSt leaf;
};

// element offsets into Lds::refs: left (index 0 = corner) / above references of luma unfiltered,
// luma filtered, Cb, Cr
constexpr int R_L0 = 0, R_A0 = 66, R_LF = 130, R_AF = 196, R_LC0 = 260, R_LC1 = 294, R_AC0 = 328, R_AC1 = 360;

struct __attribute__((aligned(16))) Lds {
    // Transform working set, time-multiplexed (see code_component):
    //   r1: residual -> coefficients -> Viterbi chunk costs / levels -> reconstructed residual
    //   r2: stage-1 DCT output (i32) -> scan-order coefficients + quotients -> dequantised^T + V
    int16_t r1[1024];
    int32_t r2[33 * 32];
    // reference samples of the current block, built once per (block, component) and reused by
    // every candidate mode: luma unfiltered + [1 2 1]-filtered, chroma unfiltered
    // one array addressed by element offsets (R_*), so that choosing among the sets is integer
    // arithmetic on a DS address, never a pointer select
    int16_t refs[392];
    uint8_t recYtop[72];       // y = -1, x = -4..67 (index x+4)
    uint8_t recY[32 * 36];     // x = -4..31 (index x+4), stride 36
    uint8_t recCtop[2][40];    // y = -1, x = -4..35
    uint8_t recC[2][16 * 20];  // x = -4..15, stride 20
    uint32_t decw[128];        // trellis decisions: 4 bits per position, 8 positions per word
    int32_t q_istar[2];        // shared-Viterbi hand-off, per block: first position with a non-zero state-0 level
    int32_t q_active;          // this wave's TB takes part in the shared Viterbi
    uint16_t q_pm[3][4][4];    // per block and sub-block of the chunk: parity masks (delta 0, 1), state-0 flag
    uint8_t cu_log2[64];       // per 4x4 luma unit
    uint8_t luma_mode[64];
    uint8_t chroma_mode[16];   // per 8x8 luma unit
    uint8_t left_mode[8];      // luma mode of the CU left of the CTU, per 4 rows
    float ns_cost[4];          // per tree level: no-split cost, running split cost
    float split_cost[4];
    uint8_t ns_luma[4], ns_chroma[4], child[4];
    CtuSt st;                  // search coroutine state
};

// Per-wave uniform context.
// Per-wave uniform context, passed BY VALUE (a few registers) so that the out-of-line
// stage functions never reload it from memory.
struct Ctx {
    const CONST_AS DevConst* k;         // constant address space: uniform reads become scalar loads
    const GLOBAL_AS uint8_t* org;       // original planes of this wave's picture: Y, Cb, Cr back to back (read-only)
    int W, WH;                          // luma width, luma plane size
    uint8_t* pred_scratch;              // 1 KB per wave in HBM: prediction bytes between predict and recon
    GLOBAL_AS uint8_t* slots;           // kReconSlots saved reconstructions of this wave (see copy_block)
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
#define WAVE (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))
#define SH (SHW[WAVE])

// Everything in Ctx and every block-geometry argument is wave-uniform.  Out-of-line
// functions receive arguments in VGPRs; re-deriving them through readfirstlane lets the
// compiler keep them in SGPRs (scalar ALU, scalar branches, s_load from the constant block).
__device__ __forceinline__ Ctx uni(Ctx c) {
    // the pointers come from kernel arguments / scalar loads and keep their (global) address
    // space only if they are not laundered through integers: make just the integers scalar
    c.ctu_x = uni(c.ctu_x);
    c.ctu_y = uni(c.ctu_y);
    c.cu32_mode = uni(c.cu32_mode);
    c.write = uni(c.write);
    return c;
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
enum { PH_PREDICT, PH_FDCT, PH_QPRE, PH_QBACK, PH_QTRACE, PH_DEQ, PH_IDCT, PH_RECON, PH_TOTAL, PH_CTRL, PH_REFS, PH_SKIP, PH_NSTEP, PH_NFULL, PH_PSZ, PH_PSZ_END = PH_PSZ + 8, PH_PCNT, PH_PCNT_END = PH_PCNT + 8, PH_QB_PRE, PH_QB_WAIT1, PH_QB_WALK, PH_QB_WAIT2, PH_COUNT };
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

// Diagnostic build (-DWRENC_TRACE, never the product library): every candidate evaluation of the
// search is appended to a device buffer, 8 ints per record, in the layout of the oracle's trace
// (oracle/wrenc_oracle.h: x, y, log2 size, tree, kind, luma mode, chroma mode, f32 bits).
#ifdef WRENC_TRACE
constexpr unsigned kTraceMax = 1u << 19;
__device__ unsigned int g_trace_n;
__device__ int g_trace[kTraceMax * 8];
__device__ __forceinline__ void trace_rec(int x, int y, int lg, int tree, int kind, int ml, int mc, int bits) {
    const unsigned idx = atomicAdd(&g_trace_n, 1u);
    if (idx < kTraceMax) {
        int* r = g_trace + (size_t)idx * 8;
        r[0] = x;
        r[1] = y;
        r[2] = lg;
        r[3] = tree;
        r[4] = kind;
        r[5] = ml;
        r[6] = mc;
        r[7] = bits;
    }
}
#define TRACE_REC(...) trace_rec(__VA_ARGS__)
#else
#define TRACE_REC(...) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------
// Cross-lane reductions with DPP inside the 16-lane rows and v_readlane across the four rows:
// no LDS-crossbar round trips (ds_bpermute), and the result is a scalar.  All lanes must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppSwap1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppSwap2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141; // lane i <-> 7 - i inside each 8 lanes
constexpr int kDppRowMirror = 0x140;     // lane i <-> 15 - i inside each row
__device__ __forceinline__ int row_sum_i32(int v) { // every lane: sum over its row of 16
    v += dpp_mov<kDppSwap1>(v);
    v += dpp_mov<kDppSwap2>(v);
    v += dpp_mov<kDppRowHalfMirror>(v);
    v += dpp_mov<kDppRowMirror>(v);
    return v;
}
__device__ __forceinline__ int row_min_i32(int v) {
    v = min(v, dpp_mov<kDppSwap1>(v));
    v = min(v, dpp_mov<kDppSwap2>(v));
    v = min(v, dpp_mov<kDppRowHalfMirror>(v));
    v = min(v, dpp_mov<kDppRowMirror>(v));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v = row_sum_i32(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = row_min_i32(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// signed 64-bit sum in three limbs: v = lo + 2^24 * (mid + 2^24 * top), lo and mid 24 bits unsigned,
// top the signed rest (each limb's 64-lane sum fits 32 bits for |v| < 2^57)
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    const long long hi = v >> 24;
    const long long a = (long long)(unsigned)wave_sum_i32((int)(v & 0xFFFFFF));
    const long long b = (long long)(unsigned)wave_sum_i32((int)(hi & 0xFFFFFF));
    const long long c = (long long)wave_sum_i32((int)(hi >> 24));
    return a + ((b + (c << 24)) << 24);
}
// minimum over aligned groups of `width` lanes (width = 64 or 32)
__device__ __forceinline__ int group_min_i32(int v, int width) {
    v = row_min_i32(v);
    const int lo = min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
    const int hi = min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
    if (width == 64) return min(lo, hi);
    return LANE < 32 ? lo : hi;
}
__device__ __forceinline__ int ilog2i(int v) { return 31 - __clz(v); }
// Full-rate multiply (v_mul_i32_i24): both factors fit 24 bits everywhere it is used (sample values,
// filter taps, weights, block coordinates, angles, levels, quantiser scales); a plain `*` on ints
// compiles to the quarter-rate v_mul_lo_u32.
#define M24(a, b) __mul24((int)(a), (int)(b))

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
// element offset of plane pc inside a picture's Y | Cb | Cr slab (integer arithmetic only: the
// three planes are one allocation, so no pointer is ever selected per lane)
__device__ __forceinline__ unsigned plane_off(const Ctx& c, int pc) {
    return pc == 0 ? 0u : (pc == 1 ? (unsigned)c.WH : (unsigned)(c.WH + (c.WH >> 2)));
}
__device__ __forceinline__ int org_get(const Ctx& c, int pc, int x, int y) {
    const int cs = pc ? 1 : 0;
    const int stride = c.W >> cs;
    return c.org[plane_off(c, pc) + (unsigned)(((c.ctu_y >> cs) + y) * stride + (c.ctu_x >> cs) + x)];
}

// ---------------------------------------------------------------------------
// availability (ctu.rs:2083-2188, encoder_context.rs:918-956)
// bx, by: CTU-local luma position, lg: log2 luma size
// ---------------------------------------------------------------------------
__device__ inline bool above_right_avail(Ctx c, int bx, int by, int lg) {
    for (;;) {
        const int n = 1 << lg;
        if (c.ctu_x + bx + n >= c.W) return false;
        if (lg == 5) return c.ctu_y > 0 && c.ctu_x + 32 < c.W;
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
    return xn >= 0 && yn >= 0 && xn < c.W && yn < c.k->H &&
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
__device__ __forceinline__ void build_refs(Ctx c, int comp, int tx, int ty, int tlg) {
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
        int16_t* refL = SH.refs + (pc == 0 ? R_L0 : (pc == 1 ? R_LC0 : R_LC1));
        int16_t* refA = SH.refs + (pc == 0 ? R_A0 : (pc == 1 ? R_AC0 : R_AC1));
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
        const int16_t* refL = SH.refs + R_L0;
        const int16_t* refA = SH.refs + R_A0;
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
                SH.refs[R_LF + li] = (int16_t)v;
            } else {
                const int ai = t - (2 * n + 1);
                int v;
                if (ai == 2 * n - 1)
                    v = refA[ai];
                else if (ai == 0)
                    v = (refL[0] + 2 * refA[0] + refA[1] + 2) >> 2;
                else
                    v = (refA[ai - 1] + 2 * refA[ai] + refA[ai + 1] + 2) >> 2;
                SH.refs[R_AF + ai] = (int16_t)v;
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
        // run of available above-right samples (:1881-1893): one position per lane, then the length
        // of the leading run of set bits
        const bool a = LANE < tw && nb_avail(c, gx, gy, tn, gx + (tw + LANE) * 2, gy - 1, ar, bl);
        num_top_right = min((int)__ffsll(~__ballot(a)) - 1, tw);
    }
    if (mode == L_CCLM) {
        const bool ar = above_right_avail(c, tx, ty, tlg), bl = below_left_avail(c, tx, ty, tlg);
        const bool a = LANE < th && nb_avail(c, gx, gy, tn, gx - 1, gy + (th + LANE) * 2, ar, bl);
        num_below_left = min((int)__ffsll(~__ballot(a)) - 1, th);
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
    // selects, not an indexed array: the four slots stay in registers
#define CCLM_PUT(I, YY, CC)            \
    do {                               \
        const int i_ = (I);            \
        const int yv_ = (YY), cv_ = (CC); \
        y0 = i_ == 0 ? yv_ : y0;       \
        c0 = i_ == 0 ? cv_ : c0;       \
        y1 = i_ == 1 ? yv_ : y1;       \
        c1 = i_ == 1 ? cv_ : c1;       \
        y2 = i_ == 2 ? yv_ : y2;       \
        c2 = i_ == 2 ? cv_ : c2;       \
        y3 = i_ == 3 ? yv_ : y3;       \
        c3 = i_ == 3 ? cv_ : c3;       \
    } while (0)
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
            CCLM_PUT(i, sy, sc);
        }
    }
    if (avail_l && (mode == LT_CCLM || mode == L_CCLM)) {
        const int start = num_samp_l >> (2 + num_is_4);
        const int step = max(num_samp_l >> (1 + num_is_4), 1);
        cnt_l = min((1 + num_is_4) << 1, num_samp_l);
        for (int i = 0; i < cnt_l; ++i) {
            const int pos = start + i * step;
            CCLM_PUT(cnt_t + i, cclm_ds6(c, tx, ty, 2 * pos, -2, avail_l), rec_get(comp, cx - 1, cy + pos));
        }
    }
#undef CCLM_PUT
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

// Original sample for prediction index i (plane pc, component coordinates x, y).  A full
// evaluation reads the picture; SAD lists read the copy of the block's originals that
// stage_org() put into r2 (free while no transform runs), index obase + i.
// byte offset in r2 of the staged originals (luma at +0, Cb | Cr at +1024): the last 1.5 KB, so that
// r1 and the first 2688 bytes of r2 are one free region during SAD lists
constexpr int kOrgStage = 2688;
template <bool full>
__device__ __forceinline__ int pred_org(const Ctx& c, int pc, int x, int y, int obase, int i) {
    if (full) return org_get(c, pc, x, y);
    return ((const uint8_t*)SH.r2)[kOrgStage + obase + i];
}
__device__ __forceinline__ void stage_org(const Ctx& c, int comps, int tx, int ty, int tlg) {
    uint32_t* dst = (uint32_t*)((char*)SH.r2 + kOrgStage);
    if (comps & 1) {
        const int words = 1 << (2 * tlg - 2);
        for (int w = LANE; w < words; w += 64) {
            const int row = (4 * w) >> tlg, col = (4 * w) & ((1 << tlg) - 1);
            dst[w] = *(const GLOBAL_AS uint32_t*)&c.org[(unsigned)((c.ctu_y + ty + row) * c.W + c.ctu_x + tx + col)];
        }
    }
    if (comps & 2) {
        const int lg = tlg - 1;
        const int words = 1 << (2 * lg - 2); // per plane
        for (int w = LANE; w < 2 * words; w += 64) {
            const int pl = w >= words ? 1 : 0;
            const int ww = w - pl * words;
            const int row = (4 * ww) >> lg, col = (4 * ww) & ((1 << lg) - 1);
            dst[256 + w] = *(const GLOBAL_AS uint32_t*)&c.org[plane_off(c, 1 + pl) +
                                                             (unsigned)((((c.ctu_y + ty) >> 1) + row) * (c.W >> 1) +
                                                                        ((c.ctu_x + tx) >> 1) + col)];
        }
    }
    WSYNC();
}

// one predicted sample: accumulate |org - pred|; `full` also stores residual and prediction
// The prediction itself is parked in the block's own area of the reconstruction tile until the
// residual is added to it (nothing reads that area in between: the reference samples are cached, and
// CCLM reads the luma plane while it writes chroma).  Only the final pass, which compares its
// reconstruction with the search's, keeps the tile and parks the prediction in global scratch.
template <bool full>
__device__ __forceinline__ int emit_sample(const Ctx& c, int o, int i, int v, int pc, int x, int y, bool to_tile) {
    const int d = o - v;
    if (full) { // i already includes the block's base in r1 / the prediction scratch
        SH.r1[i] = (int16_t)d;
        if (to_tile)
            rec_put(pc, x, y, v);
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
                                       bool to_tile = true) {
    c = uni(c);
    rbase = uni(rbase);
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
    const int obase = comp ? 1024 : 0;
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
            const int o = pred_org<full>(c, comp + blk, cx + x, cy + y, obase, i); // issued early
            int v;
            if (flat128) {
                v = 128;
            } else {
                const int ds = cclm_ds6(c, tx, ty, 2 * y, 2 * x, avail_l);
                v = (M24(ds, blk ? a1 : a0) >> (blk ? k1 : k0)) + (blk ? b1 : b0);
                v = min(max(v, 0), 255);
            }
            sad += emit_sample<full>(c, o, rbase + i, v, comp + blk, cx + x, cy + y, to_tile);
        }
        WSYNC();
        return sad;
    }
    // luma blocks of more than 32 samples use the filtered references for modes 0, 2, 34, 66
    const bool filt = comp == 0 && nn > 32 && (mode == 0 || mode == 2 || mode == 34 || mode == 66);
    const int oL0 = comp == 0 ? (filt ? R_LF : R_L0) : R_LC0; // index 0 = corner
    const int oA0 = comp == 0 ? (filt ? R_AF : R_A0) : R_AC0;
    const int16_t* L0 = SH.refs + oL0;
    const int16_t* A0 = SH.refs + oA0;
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
            const int o = pred_org<full>(c, comp + blk, cx + x, cy + y, obase, i); // issued early
            const int16_t* L = SH.refs + (blk ? R_LC1 : oL0);
            const int16_t* A = SH.refs + (blk ? R_AC1 : oA0);
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
            sad += emit_sample<full>(c, o, rbase + i, v, comp + blk, cx + x, cy + y, to_tile);
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
    // projection.  It lives in the upper half of r2 (no transform runs during a prediction), so
    // a sample's taps are consecutive LDS reads with no selects.
    const bool vertical = mode >= 34;
    constexpr int RM0 = 1024, RMS = 104; // int16 index of the table in r2, stride per block
    int16_t* rm = (int16_t*)SH.r2 + RM0;
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
            rm[blk * RMS + ee] = SH.refs[k == 0 ? oL : (from_above ? oA + k - 1 : oL + k)];
        }
        WSYNC();
    }
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg);
        const int ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int o = pred_org<full>(c, comp + blk, cx + x, cy + y, obase, i); // issued early
        const int16_t* L = SH.refs + (blk ? R_LC1 : oL0);
        const int16_t* A = SH.refs + (blk ? R_AC1 : oA0);
        int v;
        {
            const int along = vertical ? y : x, across = vertical ? x : y;
            const int i_idx = M24(along + 1, angle) >> 5;
            const int i_fact = M24(along + 1, angle) & 31;
            const int16_t* tap = rm + blk * RMS + n + across + i_idx; // tap[t] = ref[across + i_idx + t]
            if (comp == 0) {
                int f0, f1, f2, f3;
                if (filter_flag) {
                    f0 = 16 - (i_fact >> 1);
                    f1 = 32 - (i_fact >> 1);
                    f2 = 16 + (i_fact >> 1);
                    f3 = i_fact >> 1;
                } else {
                    const int w = *(const int*)&SHT.fc[i_fact][0];
                    f0 = (int)(int8_t)w;
                    f1 = (int)(int8_t)(w >> 8);
                    f2 = (int)(int8_t)(w >> 16);
                    f3 = w >> 24;
                }
                const int acc = M24(f0, tap[0]) + M24(f1, tap[1]) + M24(f2, tap[2]) + M24(f3, tap[3]);
                v = min(max((acc + 32) >> 6, 0), 255);
            } else {
                // i_fact == 0 gives tap[1] itself; a convex combination of 8-bit samples needs no `& 0xFF`
                v = (M24(32 - i_fact, tap[1]) + M24(i_fact, tap[2]) + 16) >> 5;
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
        sad += emit_sample<full>(c, o, rbase + i, v, comp + blk, cx + x, cy + y, to_tile);
    }
    WSYNC();
    return sad;
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
__device__ __forceinline__ unsigned sad_list_angular(const Ctx& c, int comps, int tx, int ty, int tlg, int nmodes,
                                                     unsigned long long modes_lo, unsigned long long modes_hi) {
    unsigned acc = 0;
    const int my_mode = LANE < nmodes ? (int)(((LANE < 8 ? modes_lo : modes_hi) >> (8 * (LANE & 7))) & 255u) : kNoMode;
    int16_t* tab = SH.r1;                   // [entry][blk][4n] projected references
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
        const int obase = comp ? 1024 : 0;
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
        if (LANE < nmodes) {
            ptab[LANE] = ((uint32_t)my_inv & 0xFFFFu) | (mm >= 34 ? 0x10000u : 0u) | (valid ? 0x20000u : 0u);
            ptab2[LANE] = ((uint32_t)my_angle & 0xFFFFu) | ((uint32_t)my_flags << 16) | ((uint32_t)mm << 24);
        }
        WSYNC();
        // ---- projected main references of every entry (intra_predictor.rs:1311-1420) ----
        {
            const int oL0 = comp == 0 ? R_L0 : R_LC0, oA0 = comp == 0 ? R_A0 : R_AC0; // (filtered refs: modes 2, 34, 66 below)
            const int total = nmodes << (lgs + cs);
            for (int e = LANE; e < total; e += 64) {
                const int mi = e >> (lgs + cs);
                const int blk = cs ? ((e >> lgs) & 1) : 0;
                const int ee = e & ((1 << lgs) - 1);
                const uint32_t pw = ptab[mi];
                const int inv_angle = (int)(int16_t)(pw & 0xFFFF);
                const bool vertical = (pw >> 16) & 1;
                const int idx = ee - n;
                int oL = blk ? R_LC1 : oL0, oA = blk ? R_AC1 : oA0;
                if (comp == 0 && nn > 32) { // luma blocks of more than 32 samples: modes 2, 34, 66 use the filtered references
                    const int m = (int)(((mi < 8 ? modes_lo : modes_hi) >> (8 * (mi & 7))) & 255u);
                    if (m == 2 || m == 34 || m == 66) {
                        oL = R_LF;
                        oA = R_AF;
                    }
                }
                const int k = idx >= 0 ? min(idx, 2 * n) : max(min((M24(idx, inv_angle) + 256) >> 9, n), 0);
                const bool from_above = (idx >= 0) == vertical;
                tab[e] = SH.refs[k == 0 ? oL : (from_above ? oA + k - 1 : oL + k)];
            }
        }
        WSYNC();
        if (nb * nn <= 32) {
            // ---- small blocks (4x4 luma: 16 samples, 4x4 chroma pair: 32): 4 or 2 entries share an
            // iteration, the entry's parameters are per-lane values ----
            const int lgS = nb * nn == 32 ? 5 : 4;
            const int slot = LANE >> lgS;
            const int i = LANE & ((1 << lgS) - 1);
            const int blk = i >> (2 * lg);
            const int ii = i & (nn - 1);
            const int x = ii & (n - 1), y = ii >> lg;
            const int o = ((const uint8_t*)SH.r2)[kOrgStage + obase + i];
            const int16_t* L = SH.refs + (blk ? R_LC1 : (comp == 0 ? R_L0 : R_LC0));
            const int16_t* A = SH.refs + (blk ? R_AC1 : (comp == 0 ? R_A0 : R_AC0));
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
                const int16_t* tap = tab + (((min(mi, 15) << cs) + blk) << lgs) + n + across + i_idx;
                int v;
                if (comp == 0) {
                    const int w = *(const int*)&SHT.fc[i_fact][0];
                    const int h = i_fact >> 1;
                    const int f0 = filter_flag ? 16 - h : (int)(int8_t)w;
                    const int f1 = filter_flag ? 32 - h : (int)(int8_t)(w >> 8);
                    const int f2 = filter_flag ? 16 + h : (int)(int8_t)(w >> 16);
                    const int f3 = filter_flag ? h : (w >> 24);
                    const int a4 = M24(f0, tap[0]) + M24(f1, tap[1]) + M24(f2, tap[2]) + M24(f3, tap[3]);
                    v = min(max((a4 + 32) >> 6, 0), 255);
                } else {
                    v = (M24(32 - i_fact, tap[1]) + M24(i_fact, tap[2]) + 16) >> 5;
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
            for (int i = LANE; i < nb * nn; i += 64) {
                const int blk = i >> (2 * lg);
                const int ii = i & (nn - 1);
                const int x = ii & (n - 1), y = ii >> lg;
                const int o = ((const uint8_t*)SH.r2)[kOrgStage + obase + i];
                const int16_t* L = SH.refs + (blk ? R_LC1 : oL0);
                const int16_t* A = SH.refs + (blk ? R_AC1 : oA0);
                const int along = vertical ? y : x, across = vertical ? x : y;
                const int i_idx = M24(along + 1, angle) >> 5;
                const int i_fact = M24(along + 1, angle) & 31;
                const int16_t* tap = tab + (((mi << cs) + blk) << lgs) + n + across + i_idx; // tap[t] = ref[across + i_idx + t]
                int v;
                if (comp == 0) {
                    int f0, f1, f2, f3;
                    if (filter_flag) {
                        f0 = 16 - (i_fact >> 1);
                        f1 = 32 - (i_fact >> 1);
                        f2 = 16 + (i_fact >> 1);
                        f3 = i_fact >> 1;
                    } else {
                        const int w = *(const int*)&SHT.fc[i_fact][0];
                        f0 = (int)(int8_t)w;
                        f1 = (int)(int8_t)(w >> 8);
                        f2 = (int)(int8_t)(w >> 16);
                        f3 = w >> 24;
                    }
                    const int a4 = M24(f0, tap[0]) + M24(f1, tap[1]) + M24(f2, tap[2]) + M24(f3, tap[3]);
                    v = min(max((a4 + 32) >> 6, 0), 255);
                } else {
                    v = (M24(32 - i_fact, tap[1]) + M24(i_fact, tap[2]) + 16) >> 5;
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
    }
    return acc;
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
__device__ void fwd_dct(Ctx c, int nb, int o1) {
    constexpr int N = 1 << LG;
    constexpr int G = 64 / N;
    constexpr int HS = N + 1; // r2 row stride
    const int u = LANE & (N - 1);
    const int g = LANE >> LG;
    uint32_t t[N / 2];
    {
        const CONST_AS uint32_t* src = (const CONST_AS uint32_t*)&c.k->dct[LG - 2][u][0];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) t[k] = src[k];
    }
    // stage 1: H[u][y] = (sum_x T[u][x] r[y][x] + d) >> (LG-1)   (:2139-2209); rows of all blocks
#pragma unroll 1
    for (int yy = g; yy < nb * N; yy += G) {
        const uint32_t* row = (const uint32_t*)&SH.r1[o1 + yy * N];
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
        SH.r1[o1 + blk * (N * N) + u * N + x] = (int16_t)((acc + (1 << (LG + 5))) >> (LG + 6));
    }
    WSYNC();
}

// inverse: nb transposed dequantised blocks in the lower half of r2 ([blk][x][i], i16) ->
// residuals r1 ([blk][y][x]); the intermediate lives in the upper half of r2.  transformer.rs:2380-2737
template <int LG>
__device__ void inv_dct(Ctx c, int nb, int o1) {
    constexpr int N = 1 << LG;
    constexpr int G = 64 / N;
    const int u = LANE & (N - 1);
    const int g = LANE >> LG;
    const int16_t* dqt = (const int16_t*)SH.r2;
    int16_t* vbuf = (int16_t*)SH.r2 + 1024;
    uint32_t t[N / 2]; // Tt[u][i] = T_N[i][u]
    {
        const CONST_AS uint32_t* src = (const CONST_AS uint32_t*)&c.k->dct_t[LG - 2][u][0];
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
        SH.r1[o1 + yy * N + u] = (int16_t)((acc + 2048) >> 12);
    }
    WSYNC();
}

// o1: where the blocks start in r1 (i16 units, a multiple of 2)
__device__ __forceinline__ void fwd_dct_lg(Ctx c, int lg, int nb, int o1 = 0) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    o1 = uni(o1);
    switch (lg) {
    case 2: fwd_dct<2>(c, nb, o1); break;
    case 3: fwd_dct<3>(c, nb, o1); break;
    case 4: fwd_dct<4>(c, nb, o1); break;
    default: fwd_dct<5>(c, nb, o1); break;
    }
}
__device__ __forceinline__ void inv_dct_lg(Ctx c, int lg, int nb, int o1 = 0) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    o1 = uni(o1);
    switch (lg) {
    case 2: inv_dct<2>(c, nb, o1); break;
    case 3: inv_dct<3>(c, nb, o1); break;
    case 4: inv_dct<4>(c, nb, o1); break;
    default: inv_dct<5>(c, nb, o1); break;
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

// Decisions of the 16 positions of one sub-block: one 16-bit mask per state (bit k = position k
// takes a0 + 1 when it is reached in that state), two dwords.
struct DecMasks {
    uint32_t m01, m23;
};
__device__ __forceinline__ DecMasks dec_masks(const uint16_t* dec16, int p) {
    const uint2 v = *(const uint2*)(dec16 + (p >> 4) * 4);
    DecMasks m;
    m.m01 = v.x;
    m.m23 = v.y;
    return m;
}
__device__ __forceinline__ int dec_nib(DecMasks m, int p) { // bit s = decision of position p in state s
    const int k = p & 15;
    const uint32_t t01 = m.m01 >> k, t23 = m.m23 >> k;
    return (int)((t01 & 1u) | ((t01 >> 15) & 2u) | ((t23 & 1u) << 2) | ((t23 >> 13) & 8u));
}

// State maps {0..3} -> {0..3} are kept as one byte per state, so that composing two maps is one
// byte permute (v_perm_b32): (g2 o g1)(s) = g2[g1[s]].
constexpr int kMapId = 0x03020100;
__device__ __forceinline__ int compose_map(int g2, int g1) {
    return (int)__builtin_amdgcn_perm(0u, (uint32_t)g2, (uint32_t)g1);
}
// map of one position: state s goes to q_state_trans_table[s][parity of a_s] (encoder_context.rs:339),
// a_s = a0 of the state's delta class + the position's decision in state s; the table entry is
// (s >> 1) + 2 * (parity ^ (s & 1))
__device__ __forceinline__ int position_map(int tc, int qd, bool dcn, int nib) {
    int pv = 0; // bit s = parity of a_s
    if (tc != 0) {
        const int b0 = (qd >> 1) & 1;
        const int b1 = dcn ? b0 : (((qd + 1) >> 1) & 1);
        pv = nib ^ (b0 ? 3 : 0) ^ (b1 ? 12 : 0);
    }
    const unsigned x = (unsigned)(pv ^ 10);
    return (int)(0x01010000u + (((x * 0x00204081u) & 0x01010101u) << 1));
}

// Which wave of the workgroup walks the pooled Viterbi.  Waves w and w + 4 share a SIMD with the same
// two waves of the CU's other workgroup; if every workgroup walked in wave 0, one SIMD of each CU would
// carry all the serial walks and its waves would reach every barrier last.  Spread by workgroup index.
__device__ __forceinline__ int walker_wave() { return (int)((blockIdx.x * 2654435761u) >> 30); }

// Path costs are kept in 32 bits, DOUBLED, with the tie-break of quantizer.rs:505 in the low bit.
// Only cost DIFFERENCES between the four states decide the path, and they are bounded: any
// state reaches any other state's continuation within two steps (q_state_trans_table is 2-step
// complete), and one step costs at most 128*65535 + lambda_q*dq_table[1023] < 2^25 (QP 63), so
// |C_s - C_s'| < 2^26.2.  Subtracting the quad minimum every 16 positions therefore keeps every
// cost below 2^26.2 + 16*2^25 < 2^29.1, its double below 2^30.1.
// A zero coefficient has no second branch; it is given the cost 2^27, which can never win
// against branch 0 (K0 <= n0 + 2^25 <= n1 + 2^26.2 + 2^25 < n1 + 2^27) and cannot overflow.
//
// Walk step of state s: the two candidates are K0 = c0 + C[trans[s][par]] ("keep a0") and
// K1 = c1 + C[trans[s][par ^ 1]] ("take a0 + 1"), par = parity of a0; K1 wins only if K1 < K0.
// The chunk precompute stores, per position and state class, u = cost that goes with
// C[trans[s][0]] and w = cost that goes with C[trans[s][1]], as 2*cost + tie bit such that the
// single comparison KB < KA (KA = u + CA, KB = w + CB) is exact: choseB == pick1 ^ par.
constexpr int kNoBranch = 1 << 27;

// lambda_q * dq_table[idx] (quantizer.rs:29-31): the first 256 entries are in LDS; larger levels are
// rare, and a wave without any takes no branch
__device__ __forceinline__ int ldq_fast(const Ctx& c, int idx) {
    int v = SHT.ldq[min(idx, 255)];
    if (__ballot(idx > 255) != 0ULL) {
        if (idx > 255) v = (int)c.k->ldq[idx];
    }
    return v;
}

// Chunk entry of one position (see the comment above kNoBranch): writes (u, w) of the three state
// classes, returns the parities of a0 in the two delta classes and the state-0 "kept zero inside the
// trailing run" flag.  Branch-free apart from the rare large-level table reads.
//   tc, qd: coefficient and quotient of the position; dcn: the DC position (p == P - 1), whose
//   levels wrap through i16 (quantizer.rs:378-391); tzp: p <= istar; sh / off / lsc: quantiser scale
__device__ __forceinline__ void chunk_entry(const Ctx& c, int* en, int tc, int qd, bool dcn, bool tzp, int sh, int off,
                                            int lsc, int ldq1, int* par0_out, int* par1_out, int* adj_out,
                                            int* ovf) {
    const bool nz = tc != 0;
    int c0d[2], c1d[2], par[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int a0 = (qd + (dcn ? 0 : d)) >> 1; // quantizer.rs:378 / :441
        const int a1 = a0 + 1;
        // 2*a - d fits i16, so the reference's `as i16` only matters for a0 = 0, d = 1 at the DC position (-1)
        int q0 = (a0 > 0 || dcn) ? 2 * a0 - d : 0;
        int q1 = 2 * a1 - d;
        if (tc < 0) {
            q0 = -q0;
            q1 = -q1;
        }
        const int d0 = abs(tc - ((M24(q0, lsc) + off) >> sh)); // |q| <= 2047, lsc < 2^21
        const int d1 = abs(tc - ((M24(q1, lsc) + off) >> sh));
        if (nz && a1 + 1 >= 1024) *ovf = 1;
        const int l0 = ldq_fast(c, min(a0 + 1, 1023)), l1 = ldq_fast(c, min(a1 + 1, 1023));
        c0d[d] = nz ? 128 * d0 + l0 : ldq1;       // zero coefficient outside the trailing run: dq_table[1] (:433)
        c1d[d] = nz ? 128 * d1 + l1 : kNoBranch;
        par[d] = nz ? (a0 & 1) : 0;               // parity of a0 -> which successor state
    }
    const int a00 = qd >> 1;                      // a0 of delta class 0
    const bool zero0 = !nz || a00 == 0;
    // bits 0 instead of 1 for a zero kept inside the trailing run (:449-453)
    const int c0tz = nz ? (a00 == 0 ? c0d[0] - ldq1 : c0d[0]) : 0;
    const int c0s0 = tzp ? c0tz : c0d[0];
    const int p0 = par[0], p1 = par[1];
    en[0] = 2 * (p0 ? c1d[0] : c0s0) + p0;
    en[1] = 2 * (p0 ? c0s0 : c1d[0]) + 1 - p0;
    en[2] = 2 * (p0 ? c1d[0] : c0d[0]) + p0;
    en[3] = 2 * (p0 ? c0d[0] : c1d[0]) + 1 - p0;
    en[4] = 2 * (p1 ? c1d[1] : c0d[1]) + p1;
    en[5] = 2 * (p1 ? c0d[1] : c1d[1]) + 1 - p1;
    *par0_out = p0;
    *par1_out = p1;
    *adj_out = (tzp && zero0) ? 1 : 0;
}

// level-cost table (block_splitter.rs:436-458), same access pattern as ldq_fast
__device__ __forceinline__ int lv_fast(const Ctx& c, int a) {
    int v = SHT.lv[min(a, 255)];
    if (__ballot(a > 255) != 0ULL) {
        if (a > 255) v = (int)c.k->lv[a];
    }
    return v;
}

// One position of the forward trace (quantizer.rs:686-721) in `state`: returns the level, advances the
// state, and accumulates the level-cost terms of the position (block_splitter.rs:436-458): the
// table cost of a non-zero level, a bit in zmask for a zero, the first non-zero position.
__device__ __forceinline__ int emit_level(const Ctx& c, int tc, int qd, bool dcn, int nib, int p, int j, int& state,
                                          unsigned& zmask, long long& sum_nz, int& fnz, int& ovf) {
    const int dl = state > 1 ? 1 : 0;
    const bool nz = tc != 0;
    const int a = nz ? ((qd + (dcn ? 0 : dl)) >> 1) + ((nib >> state) & 1) : 0;
    // 2*a - dl fits i16: the reference's usize wrap + `as i16` (quantizer.rs:379,391) only shows for
    // a = 0, dl = 1 at the DC position (-1)
    int q = (nz && (a > 0 || dcn)) ? 2 * a - dl : 0;
    if (tc < 0) q = -q;
    const int qc = abs(q);
    const bool zero = qc == 0;
    zmask |= (zero ? 1u : 0u) << j;
    const int aw = (qc + dl) >> 1;
    if (!zero && aw >= 1024) ovf = 1;
    const int lv = lv_fast(c, min(aw, 1023));
    sum_nz += zero ? 0 : lv;
    fnz = zero ? fnz : min(fnz, p);
    state = (0x7D28 >> (2 * (2 * state + (a & 1)))) & 3;
    return q;
}

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
    const CONST_AS DevConst* k = c.k;
    const int n = 1 << lg;
    const int P = n * n;
    const int lgP = 2 * lg;
    const int sh = 8 + lg - 5 + 1; // quantizer.rs:558-569
    const int off = (1 << sh) >> 1;
    const int lsc = k->lsc;
    const CONST_AS uint16_t* scan = k->scan_idx[lg - 2];
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
    const bool walker = shared ? (WAVE == walker_wave() && wwave < WPB) : (quad < nb);
    const Lds* tb = shared ? &SHW[wwave < WPB ? wwave : 0] : &SH;
    const int32_t* wcc = (const int32_t*)tb->r1 + wblk * CH * 6;
    int C = 0;
    int ovf = 0;
    for (int base = P - CH; base >= 0; base -= CH) {
        PROF_MARK(qb0_);
        WSYNC();
        if (active) {
            // per position and state class (0: state 0, 1: state 1, 2: states 2 and 3): (u, w) doubled,
            // see above; per sub-block: parity masks of the two delta classes and, for state 0, whether
            // its first position in coding order (kk == 15) keeps a zero inside the trailing run
            const bool mine = LANE < nb * CH;
            const int blk = LANE >= CH ? 1 : 0;
            const int i = LANE - blk * CH;
            const int p = base + i;
            int par0 = 0, par1 = 0, adj = 0;
            if (mine)
                chunk_entry(c, cc + LANE * 6, tcs[blk * P + p], qds[blk * P + p], p == P - 1, p <= (blk ? istar1 : istar0),
                            sh, off, lsc, ldq1, &par0, &par1, &adj, &ovf);
            const unsigned long long b0 = __ballot(mine && par0), b1 = __ballot(mine && par1), ba = __ballot(mine && adj);
            if (mine && (LANE & 15) == 0) {
                uint16_t* pm = SH.q_pm[blk][i >> 4];
                pm[0] = (uint16_t)(b0 >> LANE);
                pm[1] = (uint16_t)(b1 >> LANE);
                pm[2] = (uint16_t)((ba >> (LANE + 15)) & 1);
            }
        }
        PROF_MARK(qb1_);
        if (shared)
            __syncthreads();
        else
            WSYNC();
        PROF_MARK(qb2_);
        if (walker && (!shared || tb->q_active)) {
            const int cls = st == 0 ? 0 : (st == 1 ? 1 : 2);
            uint16_t* wdec = (uint16_t*)const_cast<uint32_t*>(tb->decw) + wblk * (P >> 2);
            for (int g16 = CH - 16; g16 >= 0; g16 -= 16) { // one 4x4 sub-block per iteration
                const uint16_t* pm = tb->q_pm[wblk][g16 >> 4];
                const unsigned parmask = pm[st > 1 ? 1 : 0];
                const bool adj = st == 0 && pm[2] != 0;
                // all 16 entries of the sub-block are fetched before its walk (a serial dependency
                // chain that should not wait for LDS position by position)
                int2 cur[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) cur[kk] = *(const int2*)&wcc[(g16 + kk) * 6 + 2 * cls];
                unsigned bits = 0;
#pragma unroll
                for (int kk = 15; kk >= 0; --kk) {
                    const int2 e = cur[kk];
                    const int KA = e.x + dpp_quad<0xD8>(C); // C[trans[s][0]]: quad_perm [0,2,1,3]
                    const int KB = e.y + dpp_quad<0x72>(C); // C[trans[s][1]]: quad_perm [2,0,3,1]
                    const bool choseB = KB < KA;
                    C = (choseB ? KB : KA) & ~1;
                    bits = (bits << 1) | (choseB ? 1u : 0u);
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        const bool pick1 = choseB != (((parmask >> 15) & 1) != 0);
                        if (!pick1 && adj) C -= 2 * ldq1;
                    }
                }
                bits ^= parmask; // choseB -> pick1
                // renormalise: subtract the quad minimum (decisions depend on differences only)
                int m = min(C, dpp_quad<0xB1>(C));  // quad_perm [1,0,3,2]
                m = min(m, dpp_quad<0x4E>(m));      // quad_perm [2,3,0,1]
                C -= m;
                wdec[((base + g16) >> 4) * 4 + st] = (uint16_t)bits;
            }
        }
        PROF_MARK(qb3_);
        if (shared) __syncthreads();
        PROF_MARK(qb4_);
        PROF_ADD2(PH_QB_PRE, qb0_, qb1_);
        PROF_ADD2(PH_QB_WAIT1, qb1_, qb2_);
        PROF_ADD2(PH_QB_WALK, qb2_, qb3_);
        PROF_ADD2(PH_QB_WAIT2, qb3_, qb4_);
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
    int fmap = kMapId;
    const DecMasks dm = dec_masks(bdec, act ? p0 : 0); // a lane's positions lie in one sub-block (per divides 16)
    if (act) {
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            fmap = compose_map(position_map(btcs[p], bqds[p], p == P - 1, dec_nib(dm, p)), fmap);
        }
    }
    // inclusive prefix composition across the lanes of a block: Hillis-Steele inside the 16-lane rows
    // with row_shr DPP moves (lanes without a source get the identity map), then the row totals
    // travel with row_bcast:15 / row_bcast:31 (the two blocks of a chroma pair are lanes 0..31 and
    // 32..63, so they simply skip the last step).  No LDS-crossbar shuffles.
    int pre = fmap;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
    if (nb == 1) pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x143, 0xC, 0xF, false)); // row_bcast:31 -> rows 2, 3
    // state after all previous lanes of the block, starting from 0
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x138, 0xF, 0xF, false) & 3; // wave_shr:1
    if (lane_in == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = P;
    if (act) {
        int state = entry;
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            SH.r1[blk * P + scan[p]] =
                (int16_t)emit_level(c, btcs[p], bqds[p], p == P - 1, dec_nib(dm, p), p, j, state, zmask, sum_nz, fnz, ovf);
        }
    }
    const int pf = group_min_i32(fnz, half); // zeros before a block's first non-zero level cost nothing
    if (act) // zeros after the first non-zero position: positions j > pf - p0 of this lane
        sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
    const long long sum = wave_sum_i64(sum_nz);
    if (__ballot(ovf != 0) != 0ULL) *overflow = 1;
    WSYNC();
    PROF_MARK(q3_);
    PROF_ADD2(PH_QTRACE, q2_, q3_);
    return sum;
}

// Dependent quantisation of the three transform blocks of one candidate in ONE pooled pass: luma
// n0 x n0 at r1[0, P0), Cb and Cr (n0/2)^2 at r1[P0, P0 + Pc) and r1[P0 + Pc, P0 + 2 Pc), n0 = 8 or
// 16 (search only: every wave of the workgroup is in this call with the same block size).  Same
// algorithm as quantize(); the chroma chains are a quarter as long as the luma chain, so a chunk is
// 64 luma + 16 + 16 chroma positions and the chroma blocks ride along for free: wave 0 walks the
// 8 luma blocks (lanes 0..31) and the 8 Cb blocks (lanes 32..63), wave 1 the 8 Cr blocks.
// Scratch: r2 = [scan-order coefficients | quotients | chunk entries], decw.
__device__ __forceinline__ void quantize3(Ctx c, int lg0, bool active, int* overflow, long long* lvl_y,
                                          long long* lvl_c) {
    static_assert(WPB == 8, "the merged pass maps 8 waves x 3 blocks onto two walker waves");
    c = uni(c);
    lg0 = uni(lg0);
    const CONST_AS DevConst* k = c.k;
    const int lgc = lg0 - 1;
    const int P0 = 1 << (2 * lg0), Pc = P0 >> 2, T = P0 + 2 * Pc;
    const int sh0 = lg0 + 4, shc = lgc + 4; // 8 + lg - 5 + 1 (quantizer.rs:558-569)
    const int lsc = k->lsc;
    const CONST_AS uint16_t* scan0 = k->scan_idx[lg0 - 2];
    const CONST_AS uint16_t* scanc = k->scan_idx[lgc - 2];
    int16_t* tcs = (int16_t*)SH.r2;               // [T]: coefficient in reverse-scan order, block after block
    int16_t* qds = (int16_t*)SH.r2 + T;           // [T]: |(tc << sh) - off| / lsc
    constexpr int kCcByte = 1536;                 // 2 * 2 * T <= 1536 for T <= 384
    int32_t* cc = (int32_t*)((char*)SH.r2 + kCcByte); // chunk: [96][6] ints
    *lvl_y = 0;
    *lvl_c = 0;
    PROF_MARK(q0_);
    int istar0 = P0, istar1 = Pc, istar2 = Pc;
    if (active) {
        int first0 = P0, first1 = Pc, first2 = Pc;
        for (int idx = LANE; idx < T; idx += 64) {
            const int b = idx < P0 ? 0 : (idx < P0 + Pc ? 1 : 2);
            const int boff = b == 0 ? 0 : (b == 1 ? P0 : P0 + Pc);
            const int p = idx - boff;
            const int sh = b == 0 ? sh0 : shc;
            const int off = (1 << sh) >> 1;
            const int tc = SH.r1[boff + (b == 0 ? scan0[p] : scanc[p])];
            int S = (int)((unsigned)tc << sh) - off;
            if (tc < 0) S = -S;
            const int qd = tc == 0 ? 0 : (int)(((unsigned long long)(unsigned)S * k->div_magic) >> 47);
            tcs[idx] = (int16_t)tc;
            qds[idx] = (int16_t)qd;
            if (tc != 0 && (qd >> 1) > 0) {
                if (b == 0)
                    first0 = min(first0, p);
                else if (b == 1)
                    first1 = min(first1, p);
                else
                    first2 = min(first2, p);
            }
        }
        istar0 = wave_min_i32(first0);
        istar1 = wave_min_i32(first1);
        istar2 = wave_min_i32(first2);
    }
    if (LANE == 0) SH.q_active = active ? 1 : 0;
    PROF_MARK(q1_);
    PROF_ADD2(PH_QPRE, q0_, q1_);
    const int ldq1 = (int)ldq_at(c, 1);
    const int st = LANE & 3;
    const int cls = st == 0 ? 0 : (st == 1 ? 1 : 2);
    // walker lanes: wave 0 lanes 0..31 luma of wave LANE/4, lanes 32..63 Cb; wave 1 lanes 0..31 Cr
    const int wv = (WAVE - walker_wave()) & (WPB - 1); // 0 and 1: the two walker waves
    const int wb = wv == 0 ? (LANE < 32 ? 0 : 1) : 2;
    const bool walker = wv == 0 || (wv == 1 && LANE < 32);
    const Lds* tb = &SHW[(LANE & 31) >> 2];
    const int32_t* wcc = (const int32_t*)((const char*)tb->r2 + kCcByte) + (wb == 0 ? 0 : (wb == 1 ? 64 : 80)) * 6;
    uint16_t* wdec = (uint16_t*)const_cast<uint32_t*>(tb->decw) + (wb == 0 ? 0 : (wb == 1 ? (P0 >> 2) : (P0 >> 2) + (Pc >> 2)));
    const int wnsb = wb == 0 ? 4 : 1; // sub-blocks of the walker's block per chunk
    int C = 0;
    int ovf = 0;
    const int nch = P0 >> 6;
    for (int ch = 0; ch < nch; ++ch) {
        const int base0 = P0 - 64 * (ch + 1), basec = Pc - 16 * (ch + 1);
        PROF_MARK(qb0_);
        WSYNC();
        if (active) {
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                const int e = LANE + 64 * pass;
                const bool mine = e < 96;
                const int b = e < 64 ? 0 : (e < 80 ? 1 : 2);
                const int i = b == 0 ? e : ((e - 64) & 15);
                const int p = (b == 0 ? base0 : basec) + i;
                const int Pb = b == 0 ? P0 : Pc;
                const int gidx = (b == 0 ? 0 : (b == 1 ? P0 : P0 + Pc)) + p;
                int par0 = 0, par1 = 0, adj = 0;
                if (mine) {
                    const int sh = b == 0 ? sh0 : shc;
                    chunk_entry(c, cc + e * 6, tcs[gidx], qds[gidx], p == Pb - 1,
                                p <= (b == 0 ? istar0 : (b == 1 ? istar1 : istar2)), sh, (1 << sh) >> 1, lsc, ldq1, &par0,
                                &par1, &adj, &ovf);
                }
                const unsigned long long b0 = __ballot(mine && par0), b1 = __ballot(mine && par1),
                                         ba = __ballot(mine && adj);
                if (mine && (LANE & 15) == 0) {
                    // pass 0: luma sub-block LANE / 16; pass 1: lanes 0..15 Cb, 16..31 Cr (one sub-block each)
                    uint16_t* pm = SH.q_pm[b][b == 0 ? (LANE >> 4) : 0];
                    pm[0] = (uint16_t)(b0 >> LANE);
                    pm[1] = (uint16_t)(b1 >> LANE);
                    pm[2] = (uint16_t)((ba >> (LANE + 15)) & 1);
                }
            }
        }
        PROF_MARK(qb1_);
        __syncthreads();
        PROF_MARK(qb2_);
        if (walker && tb->q_active) {
            for (int sbi = wnsb - 1; sbi >= 0; --sbi) { // one 4x4 sub-block per iteration
                const int g16 = sbi * 16;
                const uint16_t* pm = tb->q_pm[wb][sbi];
                const unsigned parmask = pm[st > 1 ? 1 : 0];
                const bool adj = st == 0 && pm[2] != 0;
                int2 cur[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) cur[kk] = *(const int2*)&wcc[(g16 + kk) * 6 + 2 * cls];
                unsigned bits = 0;
#pragma unroll
                for (int kk = 15; kk >= 0; --kk) {
                    const int2 en = cur[kk];
                    const int KA = en.x + dpp_quad<0xD8>(C); // C[trans[s][0]]: quad_perm [0,2,1,3]
                    const int KB = en.y + dpp_quad<0x72>(C); // C[trans[s][1]]: quad_perm [2,0,3,1]
                    const bool choseB = KB < KA;
                    C = (choseB ? KB : KA) & ~1;
                    bits = (bits << 1) | (choseB ? 1u : 0u);
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        const bool pick1 = choseB != (((parmask >> 15) & 1) != 0);
                        if (!pick1 && adj) C -= 2 * ldq1;
                    }
                }
                bits ^= parmask; // choseB -> pick1
                int m = min(C, dpp_quad<0xB1>(C));
                m = min(m, dpp_quad<0x4E>(m));
                C -= m;
                wdec[(((wb == 0 ? base0 : basec) + g16) >> 4) * 4 + st] = (uint16_t)bits;
            }
        }
        PROF_MARK(qb3_);
        __syncthreads();
        PROF_MARK(qb4_);
        PROF_ADD2(PH_QB_PRE, qb0_, qb1_);
        PROF_ADD2(PH_QB_WAIT1, qb1_, qb2_);
        PROF_ADD2(PH_QB_WALK, qb2_, qb3_);
        PROF_ADD2(PH_QB_WAIT2, qb3_, qb4_);
    }
    WSYNC();
    PROF_MARK(q2_);
    PROF_ADD2(PH_QBACK, q1_, q2_);
    if (!active) return;
    // ---- forward trace + level cost: lanes 0..31 luma, 32..47 Cb, 48..63 Cr ----
    const int b = LANE < 32 ? 0 : (LANE < 48 ? 1 : 2);
    const int lane_in = b == 0 ? LANE : (LANE & 15);
    const int Pb = b == 0 ? P0 : Pc;
    const int per = b == 0 ? (P0 >> 5) : (Pc >> 4); // P0 / 32 = Pc / 16 * 2
    const int boff = b == 0 ? 0 : (b == 1 ? P0 : P0 + Pc);
    const int p0 = lane_in * per;
    const int16_t* btcs = tcs + boff;
    const int16_t* bqds = qds + boff;
    const uint16_t* bdec = (const uint16_t*)SH.decw + (b == 0 ? 0 : (b == 1 ? (P0 >> 2) : (P0 >> 2) + (Pc >> 2)));
    int fmap = kMapId;
    const DecMasks dm = dec_masks(bdec, p0); // a lane's positions lie in one sub-block (per divides 16)
    for (int j = 0; j < per; ++j) {
        const int p = p0 + j;
        fmap = compose_map(position_map(btcs[p], bqds[p], p == Pb - 1, dec_nib(dm, p)), fmap);
    }
    int pre = fmap;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x142, 0x2, 0xF, false)); // row_bcast:15 -> row 1 (luma)
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x138, 0xF, 0xF, false) & 3; // wave_shr:1
    if (lane_in == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = Pb;
    {
        int state = entry;
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            SH.r1[boff + (b == 0 ? scan0[p] : scanc[p])] =
                (int16_t)emit_level(c, btcs[p], bqds[p], p == Pb - 1, dec_nib(dm, p), p, j, state, zmask, sum_nz, fnz, ovf);
        }
    }
    // zeros before a block's first non-zero level cost nothing: minimum per block (rows 0-1 | 2 | 3)
    {
        const int rm = row_min_i32(fnz);
        const int m0 = min(__builtin_amdgcn_readlane(rm, 0), __builtin_amdgcn_readlane(rm, 16));
        const int m1 = __builtin_amdgcn_readlane(rm, 32), m2 = __builtin_amdgcn_readlane(rm, 48);
        const int pf = b == 0 ? m0 : (b == 1 ? m1 : m2);
        sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
    }
    // level cost of the luma block (rows 0-1) and of the chroma pair (rows 2-3), three limbs each
    {
        const long long hi = sum_nz >> 24;
        const int ra = row_sum_i32((int)(sum_nz & 0xFFFFFF)), rb = row_sum_i32((int)(hi & 0xFFFFFF)),
                  rc = row_sum_i32((int)(hi >> 24));
        const long long ya = (long long)(unsigned)(__builtin_amdgcn_readlane(ra, 0) + __builtin_amdgcn_readlane(ra, 16));
        const long long yb = (long long)(unsigned)(__builtin_amdgcn_readlane(rb, 0) + __builtin_amdgcn_readlane(rb, 16));
        const long long yc = (long long)(__builtin_amdgcn_readlane(rc, 0) + __builtin_amdgcn_readlane(rc, 16));
        const long long ca = (long long)(unsigned)(__builtin_amdgcn_readlane(ra, 32) + __builtin_amdgcn_readlane(ra, 48));
        const long long cb = (long long)(unsigned)(__builtin_amdgcn_readlane(rb, 32) + __builtin_amdgcn_readlane(rb, 48));
        const long long cc2 = (long long)(__builtin_amdgcn_readlane(rc, 32) + __builtin_amdgcn_readlane(rc, 48));
        *lvl_y = ya + ((yb + (yc << 24)) << 24);
        *lvl_c = ca + ((cb + (cc2 << 24)) << 24);
    }
    if (__ballot(ovf != 0) != 0ULL) *overflow = 1;
    WSYNC();
    PROF_MARK(q3_);
    PROF_ADD2(PH_QTRACE, q2_, q3_);
}

// levels r1 (row-major) -> transposed dequantised coefficients in r2 (dT[x][i] = d[i][x]);
// quantizer.rs:761-1079
__device__ __forceinline__ void dequantize_t(Ctx c, int lg, int nb, int o1 = 0) {
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    o1 = uni(o1);
    const int n = 1 << lg;
    const int nn = n * n;
    const int sh = 8 + lg - 5 + 1;
    const int off = (1 << sh) >> 1;
    const int lsc = c.k->lsc;
    int16_t* out = (int16_t*)SH.r2;
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg), ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        int v = (M24(SH.r1[o1 + i], lsc) + off) >> sh;
        v = min(max(v, -32768), 32767);
        out[blk * nn + x * n + y] = (int16_t)v;
    }
    WSYNC();
}

// ---------------------------------------------------------------------------
// RD search building blocks (block_splitter.rs)
// ---------------------------------------------------------------------------
// Evaluation requests and the evaluator
// ---------------------------------------------------------------------------
enum { K_SADLIST = 0, K_FULL = 1, K_NOP = 2 };
enum { COPY_NONE = 0, COPY_SAVE = 1, COPY_RESTORE = 2 };

struct Req {
    int kind;       // K_SADLIST: predict + SAD of a list of modes (block_splitter.rs:64-108, 476-522);
                    // K_FULL: predict .. reconstruct (:146-185)
    int comps;      // bit 0: luma block, bit 1: Cb+Cr pair
    int tx, ty, tlg;
    int ml, mc;     // K_FULL: luma / chroma mode
    bool shared;    // quantiser: pooled Viterbi of the workgroup (search) or solo (regen, final pass)
    bool active;    // false: walk the schedule only (keeps the workgroup's barriers aligned)
    bool refs0, refs1; // (re)build the luma / chroma reference samples of the block first
    bool final;     // final pass: store the levels, count reconstruction changes
    int n;          // K_SADLIST: number of entries
    int tree;       // tree type of the leaf that asks (diagnostic trace only)
    // before the evaluation: save the block's reconstruction to a slot / restore it from there
    // (the reference's cache_reconsts / restore_reconsts, block_splitter.rs:807-840, 1085-1145)
    int pre_copy, copy_comps, copy_slot, copy_tx, copy_ty, copy_tlg;
    unsigned long long modes_lo, modes_hi; // K_SADLIST: one byte per entry (8 + 8), the same mode for luma and chroma
};

struct Res {
    // K_FULL: SSD and level cost of the luma block and of the chroma pair
    uint32_t ssd_y, ssd_c;
    long long lvl_y, lvl_c;
    // K_SADLIST: costs of the first three entries, first minimum (strict <) and its index
    float v0, v1, v2, vmin;
    int imin;
};

__device__ __forceinline__ float uni_f(float v) { return __int_as_float(uni(__float_as_int(v))); }

// First half of a full evaluation of one component (comp 0: luma block, 1: chroma pair): reference
// samples, prediction, forward transform.  Residual / coefficients at r1[rbase ..], prediction bytes
// in the tile (final pass: at pred_scratch[rbase ..]).
__device__ __forceinline__ void full_front(const Ctx& c, const Req& q, int comp, int mode, int rbase) {
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int lg = q.tlg - cs;
    PROF_MARK(tr0_);
    if ((comp ? q.refs1 : q.refs0) && mode < LT_CCLM) build_refs(c, comp, q.tx, q.ty, q.tlg);
    PROF_MARK(t0_);
    PROF_ADD2(PH_REFS, tr0_, t0_);
    predict<true>(c, comp, q.tx, q.ty, q.tlg, mode, rbase, !q.final);
    PROF_MARK(t1_);
    PROF_ADD2(PH_PREDICT, t0_, t1_);
    fwd_dct_lg(c, lg, nb, rbase);
    PROF_MARK(t2_);
    PROF_ADD2(PH_FDCT, t1_, t2_);
}

// Second half: levels at r1[rbase ..] -> (final pass: store them) -> dequantise, inverse transform,
// reconstruct into the tile; returns the SSD against the originals (block_splitter.rs:146-185)
__device__ __forceinline__ uint32_t full_back(const Ctx& c, const PicBufs& pb, const Req& q, int comp, int rbase) {
    const int cs = comp ? 1 : 0;
    const int nb = comp ? 2 : 1;
    const int lg = q.tlg - cs;
    const int n = 1 << lg;
    const int nn = n * n;
    const int cx = q.tx >> cs, cy = q.ty >> cs;
    PROF_MARK(t3_);
    if (q.final && c.write) {
        const int stride = c.W >> cs;
        const size_t at = (size_t)((c.ctu_y + q.ty) >> cs) * stride + ((c.ctu_x + q.tx) >> cs);
        GLOBAL_AS int16_t* lev0 = AS_GLOBAL(int16_t, pb.lev[0]) + plane_off(c, comp) + at;
        GLOBAL_AS int16_t* lev1 = AS_GLOBAL(int16_t, pb.lev[0]) + plane_off(c, 2) + at;
        for (int i = LANE; i < nb * nn; i += 64) {
            const int blk = i >> (2 * lg), ii = i & (nn - 1);
            (blk ? lev1 : lev0)[(size_t)(ii >> lg) * stride + (ii & (n - 1))] = SH.r1[rbase + i];
        }
    }
    dequantize_t(c, lg, nb, rbase);
    PROF_MARK(t4_);
    inv_dct_lg(c, lg, nb, rbase);
    PROF_MARK(t5_);
    PROF_ADD2(PH_DEQ, t3_, t4_);
    PROF_ADD2(PH_IDCT, t4_, t5_);
    unsigned int part = 0;
    int diff = 0;
    for (int i = LANE; i < nb * nn; i += 64) {
        const int blk = i >> (2 * lg), ii = i & (nn - 1);
        const int x = ii & (n - 1), y = ii >> lg;
        const int pc = comp + blk;
        const int pred = q.final ? (int)c.pred_scratch[rbase + i] : rec_get(pc, cx + x, cy + y);
        int v = (int16_t)(pred + (int)SH.r1[rbase + i]); // pred as i16 + res, clamp (:178)
        v = min(max(v, 0), 255);
        if (q.final && v != rec_get(pc, cx + x, cy + y)) ++diff;
        rec_put(pc, cx + x, cy + y, v);
        const int d = v - org_get(c, pc, cx + x, cy + y);
        part += (unsigned)M24(d, d);
    }
    const uint32_t ssd = (uint32_t)wave_sum_i32((int)part); // <= 1024 * 255^2: fits 32 bits
    if (q.final) {
        const int changed = wave_sum_i32(diff);
        if (changed && LANE == 0 && c.write) atomicAdd(c.mismatch, (unsigned long long)changed);
    }
    WSYNC();
    PROF_MARK(t6_);
    PROF_ADD2(PH_RECON, t5_, t6_);
    return ssd;
}

// Save the reconstruction of a block (comps bit 0: luma n x n, bit 1: Cb and Cr (n/2) x (n/2)) from
// the LDS tile to a slot in global scratch, or restore it from there.  Dwords: block corners are
// multiples of 4 samples in every plane that takes part.
__device__ __forceinline__ void copy_block(const Ctx& c, int mode, int comps, int slot, int tx, int ty, int tlg) {
    GLOBAL_AS uint32_t* g = (GLOBAL_AS uint32_t*)(c.slots + slot * kSlotBytes);
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
                g[256 + pl * 64 + ww] = *l;
            else
                *l = g[256 + pl * 64 + ww];
        }
    }
    WSYNC();
}

// The evaluator: every block evaluation of the search, of the regeneration and of the final pass
// goes through this one inlined copy (the search logic below is a state machine that hands out
// evaluation requests; no function calls in the hot path).
__device__ __forceinline__ Res evaluate(const Ctx& c, const PicBufs& pb, const Req& q, int* overflow) {
    Res r;
    r.ssd_y = 0;
    r.ssd_c = 0;
    r.lvl_y = 0;
    r.lvl_c = 0;
    r.v0 = r.v1 = r.v2 = r.vmin = 3.40282347e+38f;
    r.imin = 0;
    if (q.pre_copy != COPY_NONE) copy_block(c, q.pre_copy, q.copy_comps, q.copy_slot, q.copy_tx, q.copy_ty, q.copy_tlg);
    if (q.kind == K_NOP) return r;
    if (q.kind == K_FULL) {
        // A candidate of the search with an 8x8 or 16x16 luma block quantises its three transform
        // blocks in one pooled pass (quantize3): both components go through the first half, then
        // the pass, then both through the second half.  Everything else runs component by component
        // (the two share r1 / r2).  One copy of each stage either way.
        const bool merged = WPB == 8 && q.shared && q.comps == 3 && q.tlg <= 4;
        const int p0 = 1 << (2 * q.tlg);
        const int rounds = merged ? 1 : 2;
#pragma unroll 1
        for (int round = 0; round < rounds; ++round) {
            const int cset = merged ? 3 : (q.comps & (1 << round));
            if (!cset) continue;
            if (q.active) {
#pragma unroll 1
                for (int comp = 0; comp < 2; ++comp)
                    if ((cset >> comp) & 1) full_front(c, q, comp, comp ? q.mc : q.ml, (merged && comp) ? p0 : 0);
            }
            PROF_MARK(ts0_);
            if (merged) {
                quantize3(c, q.tlg, q.active, overflow, &r.lvl_y, &r.lvl_c);
            } else {
                const long long lvl = quantize(c, q.tlg - round, round ? 2 : 1, q.shared, q.active, overflow);
                if (round)
                    r.lvl_c = lvl;
                else
                    r.lvl_y = lvl;
            }
            PROF_MARK(ts1_);
            if (!q.active) { // only kept the shared-Viterbi barriers company
                PROF_ADD2(PH_SKIP, ts0_, ts1_);
                continue;
            }
#pragma unroll 1
            for (int comp = 0; comp < 2; ++comp) {
                if (!((cset >> comp) & 1)) continue;
                const uint32_t ssd = full_back(c, pb, q, comp, (merged && comp) ? p0 : 0);
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
    stage_org(c, q.comps, q.tx, q.ty, q.tlg);
    PROF_MARK(t0_);
    PROF_ADD2(PH_REFS, tr0_, t0_);
    // SADs stay integers (< 2^20, so the f32 the reference compares is exact and ordered the same
    // way); they become floats once, at the end.  An entry that is not evaluated costs f32::MAX.
    constexpr unsigned kNoSad = 0xFFFFFFFFu;
    unsigned s0 = kNoSad, s1 = kNoSad, s2 = kNoSad, smin = kNoSad;
    const int m_first = (int)(q.modes_lo & 255u);
    const int m_second = (int)((q.modes_lo >> 8) & 255u);
    if ((m_first >= 2 && m_first <= 66) || (m_first == kNoMode && m_second <= 66)) {
        // a list of angular modes (the 13 directional candidates, a step-search pair)
        const unsigned acc = sad_list_angular(c, q.comps, q.tx, q.ty, q.tlg, q.n, q.modes_lo, q.modes_hi);
        const int my_mode = LANE < q.n ? (int)(((LANE < 8 ? q.modes_lo : q.modes_hi) >> (8 * (LANE & 7))) & 255u) : kNoMode;
        if (c.write && my_mode != kNoMode)
            TRACE_REC(c.ctu_x + q.tx, c.ctu_y + q.ty, q.tlg, q.tree, (q.comps & 1) ? 0 : 2, (q.comps & 1) ? my_mode : 0, my_mode,
                      __float_as_int((float)acc));
        // first minimum = smallest (sad, index) pair
        const int key = my_mode != kNoMode ? (int)((acc << 4) | (unsigned)LANE) : 0x7FFFFFFF;
        const int kmin = wave_min_i32(key);
        if (kmin != 0x7FFFFFFF) {
            smin = (unsigned)kmin >> 4;
            r.imin = kmin & 15;
        }
        const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)acc, 0), a1 = (unsigned)__builtin_amdgcn_readlane((int)acc, 1),
                       a2 = (unsigned)__builtin_amdgcn_readlane((int)acc, 2);
        if (m_first != kNoMode) s0 = a0;
        if (q.n > 1 && m_second != kNoMode) s1 = a1;
        if (q.n > 2 && (int)((q.modes_lo >> 16) & 255u) != kNoMode) s2 = a2;
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
            if (c.write && LANE == 0 && m != kNoMode)
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
__device__ __forceinline__ int mpm_class(const Ctx& c, int bx, int by, int lg, int mode) {
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

__device__ __noinline__ void fill_maps(int bx, int by, int lg, int luma_mode, int chroma_mode, bool luma,
                                       bool chroma) {
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
    q.pre_copy = COPY_NONE;
}

__device__ __forceinline__ void req_copy(Req& q, int mode, int comps, int slot, int tx, int ty, int tlg) {
    q.pre_copy = mode;
    q.copy_comps = comps;
    q.copy_slot = slot;
    q.copy_tx = tx;
    q.copy_ty = ty;
    q.copy_tlg = tlg;
}

enum {
    C_START = 0, C_PLANAR, C_DCM, C_LIST, C_PAIR_EMIT, C_PAIR, C_F0, C_F1, C_F2, C_WIN, C_CX, C_CCLM, C_DM,
    C_DC_START, C_DC2, C_DC3, C_DC4, C_DC5
};

__device__ __forceinline__ void leaf_init(LeafSt& s, int tree, int bx, int by, int lg, int dm_mode) {
    s.cont = (uint8_t)(tree == TREE_DUAL_CHROMA ? C_DC_START : C_START);
    s.tree = (uint8_t)tree;
    s.bx = (uint8_t)bx;
    s.by = (uint8_t)by;
    s.lg = (uint8_t)lg;
    s.dm_mode = (uint8_t)dm_mode;
    s.need_refs0 = 1;
    s.need_refs1 = 1;
    s.need_save = 0;
    s.tile_best = 0;
}

// a new best candidate's reconstruction is saved to slot 0 by the request that follows it (before
// anything overwrites the tile)
__device__ __forceinline__ void leaf_attach_save(LeafSt& s, Req& q) {
    q.pre_copy = COPY_NONE;
    if (s.need_save) {
        req_copy(q, COPY_SAVE, s.tree == TREE_SINGLE ? 3 : 1, 0, s.bx, s.by, s.lg);
        s.need_save = 0;
    }
}
// a request that only saves / restores a reconstruction
__device__ __forceinline__ void leaf_copy_only(LeafSt& s, Req& q, int mode, int comps, int cont) {
    q.kind = K_NOP;
    req_copy(q, mode, comps, 0, s.bx, s.by, s.lg);
    s.cont = (uint8_t)cont;
}

// full evaluation (get_intra_pred_cost, block_splitter.rs:110-474) of comps with modes [ml, mc, mc];
// the first request of a leaf for a component also (re)builds its reference samples
__device__ __forceinline__ void leaf_full(LeafSt& s, Req& q, int comps, int ml, int mc, bool act, int cont,
                                          bool solo = false) {
    const bool r0 = (comps & 1) && s.need_refs0 != 0;
    const bool r1 = (comps & 2) && mc < LT_CCLM && s.need_refs1 != 0;
    req_full(q, comps, s.bx, s.by, s.lg, ml, mc, !solo, act, r0, r1, false);
    q.tree = s.tree;
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
__device__ __forceinline__ void leaf_sadlist(LeafSt& s, Req& q, int comps, int n, uint32_t m0, uint32_t m1, uint32_t m2,
                                             uint32_t m3, bool chroma_refs, int cont) {
    q.kind = K_SADLIST;
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
    leaf_attach_save(s, q);
    s.cont = (uint8_t)cont;
}

__device__ __forceinline__ EvalParts res_parts(const Res& r) {
    EvalParts e;
    e.ssd_y = r.ssd_y;
    e.ssd_c = r.ssd_c;
    e.lvl_y = r.lvl_y;
    e.lvl_c = r.lvl_c;
    return e;
}
__device__ __forceinline__ void put_parts(EvalPartsU& d, const EvalParts& e) {
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
__device__ __forceinline__ bool leaf_step(const Ctx& c, LeafSt& s, const Res& r, Req& q) {
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
            if (c.write && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 1, s.op_ml, s.op_mc, __float_as_int(val));
        } else {
            val = 3.40282347e+38f; // a skipped evaluation is f32::MAX in the reference
        }
    }
    for (;;) {
        switch (cont) {
        case C_START: // candidates {0,1,2,7,13,18,23,29,34,39,45,50,55,60,66} (:887)
            leaf_full(s, q, both, PLANAR, PLANAR, true, C_PLANAR);
            return true;
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
            // the 13 directional candidates: SAD, first minimum (:899-904)
            leaf_sadlist(s, q, both, 13, 2u | (7u << 8) | (13u << 16) | (18u << 24),
                         23u | (29u << 8) | (34u << 16) | (39u << 24), 45u | (50u << 8) | (55u << 16) | (60u << 24), 66u,
                         true, C_LIST);
            return true;
        case C_LIST: {
            // entry i of the list = candidate i + 2 of {0,1,2,7,13,18,23,29,34,39,45,50,55,60,66}, 7 bits each
            const int j = r.imin + 2;
            const int m = j < 8 ? (int)((0x3A5C90D0E08080ULL >> (7 * j)) & 127)
                                : (int)((0x109E3764B53A2ULL >> (7 * (j - 8))) & 127);
            // step_search(mode, 2, cost, aux=true) (:905-973)
            s.cur_mode = (uint8_t)m;
            s.cur_cost = r.vmin;
            s.step = 2;
            cont = C_PAIR_EMIT;
            break;
        }
        case C_PAIR_EMIT: {
            const int cm = s.cur_mode, st = s.step;
            const int lo = !(cm < 2 + st) ? cm - st : kNoMode;
            const int hi = !(cm + st > 66) ? cm + st : kNoMode;
            leaf_sadlist(s, q, both, 2, (uint32_t)lo | ((uint32_t)hi << 8), 0, 0, 0, true, C_PAIR);
            return true;
        }
        case C_PAIR: {
            const float cur = s.cur_cost, c0 = r.v0, c1 = r.v1;
            const int st = s.step;
            const float mn = fminf(fminf(cur, c0), c1);
            if (cur == mn) {
            } else if (c0 == mn) {
                s.cur_mode -= st;
                s.cur_cost = c0;
            } else {
                s.cur_mode += st;
                s.cur_cost = c1;
            }
            if ((st >> 1) > 0) {
                s.step = (uint8_t)(st >> 1);
                cont = C_PAIR_EMIT;
                break;
            }
            // step_search(mode, 1, _, aux=false) (:974) on {cur, cur - 1, cur + 1}, then the minimum of
            // {planar, DC, dir} (:975-978): first minimum of [planar, DC, cur, cur - 1, cur + 1], kept as
            // one running best.  Out-of-range neighbours are "evaluated" inactive: the wave still
            // walks the schedule so that the workgroup's shared Viterbi barriers stay aligned
            const int cm = s.cur_mode;
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
        case C_F2: {
            LEAF_CANDIDATE(s.cur_mode + 1);
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
            if (c.write && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, m, __float_as_int((float)s.cur_cost));
            leaf_sadlist(s, q, 2, 3, (uint32_t)LT_CCLM | ((uint32_t)T_CCLM << 8) | ((uint32_t)L_CCLM << 16), 0, 0, 0, false,
                         C_CX);
            if (!in_tile) req_copy(q, COPY_RESTORE, 1, 0, s.bx, s.by, s.lg); // (its save went out earlier)
            return true;
        }
        case C_WIN:
            return false;
        case C_CX: {
            const int cm = pick_cclm(r.v0, r.v1, r.v2);
            s.cclm_mode = (uint8_t)cm;
            leaf_full(s, q, 2, 0, cm, true, C_CCLM);
            return true;
        }
        case C_CCLM: {
            // the CCLM candidate = the winner's luma parts + the chroma parts just evaluated
            EvalParts e = s.e_best.get();
            e.ssd_c = rp.ssd_c;
            e.lvl_c = rp.lvl_c;
            const float cclm_cost = uni_f(assemble_chroma_cost(c, s.cclm_mode, e));
            if (c.write && LANE == 0)
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
        case C_DC_START:
            leaf_sadlist(s, q, 2, 3, (uint32_t)LT_CCLM | ((uint32_t)T_CCLM << 8) | ((uint32_t)L_CCLM << 16), 0, 0, 0, false,
                         C_DC2);
            return true;
        case C_DC2: {
            const int cm = pick_cclm(r.v0, r.v1, r.v2);
            s.cclm_mode = (uint8_t)cm;
            leaf_full(s, q, 2, 0, cm, true, C_DC3);
            return true;
        }
        case C_DC3:
            s.c0 = uni_f(assemble_chroma_cost(c, s.cclm_mode, rp));
            if (c.write && LANE == 0)
                TRACE_REC(c.ctu_x + s.bx, c.ctu_y + s.by, s.lg, tree, 3, 0, s.cclm_mode, __float_as_int((float)s.c0));
            leaf_full(s, q, 2, 0, s.dm_mode, true, C_DC4);
            req_copy(q, COPY_SAVE, 2, 0, s.bx, s.by, s.lg); // keep the CCLM reconstruction (:807-840)
            return true;
        case C_DC4: {
            const float dm_cost = uni_f(assemble_chroma_cost(c, s.dm_mode, rp));
            if (c.write && LANE == 0)
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

// split_ct (block_splitter.rs:782-1154) for one CTU + the final pass (ctu_encoder.rs:1421-1461):
// exhaustive quad-tree search as an explicit depth-first walk (level 0 = 32x32 ... level 2 = 8x8;
// an 8x8 node's split is four DUAL_TREE_LUMA 4x4 leaves + one DUAL_TREE_CHROMA 4x4 leaf,
// ctu.rs:1990-2063), per-level state in LDS.
//
// Decided blocks whose reconstruction was overwritten by later candidates come back from the
// slots in global scratch they were saved to (copy_block): the reference's cache_reconsts /
// restore_reconsts (block_splitter.rs:807-840, 1085-1145), with the saved planes kept in L2/HBM
// instead of LDS.
enum { T_START = 0, T_ENTER, T_NODE_LEAF, T_LEAF4_EMIT, T_LEAF4, T_LEAFC, T_REGEN_DONE, T_RETURN, T_FINAL_Z, T_FZ_TAIL, T_FZ_NEXT };

__device__ __forceinline__ bool ctu_step(Ctx& c, const Res& r, Req& q) {
    CtuSt& t = SH.st;
    bool in_leaf = t.in_leaf != 0;
    int cont = t.cont;
    for (;;) {
        if (in_leaf) {
            if (leaf_step(c, t.leaf, r, q)) {
                if (t.pend) { // the first request of a node's first child saves the unsplit candidate
                    req_copy(q, COPY_SAVE, 3, t.pslot, t.pbx, t.pby, t.plg);
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
            leaf_init(t.leaf, TREE_SINGLE, t.bx, t.by, lg, 0);
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
            // the unsplit candidate's reconstruction goes to slot 1 + level (cache_reconsts, :1085-1100)
            t.pend = 1;
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
            if (lg > 3) {
                t.level = (uint8_t)(level + 1); // descend into child 0 (same top-left corner)
                cont = T_ENTER;
                break;
            }
            // 8x8: four DUAL_TREE_LUMA 4x4 leaves, then the DUAL_TREE_CHROMA leaf
            t.split8 = 0.0f;
            t.i8 = 0;
            cont = T_LEAF4_EMIT;
            break;
        }
        case T_LEAF4_EMIT: {
            const int i8 = t.i8;
            leaf_init(t.leaf, TREE_DUAL_LUMA, t.bx + (i8 & 1) * 4, t.by + (i8 >> 1) * 4, 2, 0);
            in_leaf = true;
            cont = T_LEAF4;
            break;
        }
        case T_LEAF4: {
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
            in_leaf = true;
            cont = T_LEAFC;
            break;
        }
        case T_LEAFC: {
            fill_maps(t.bx, t.by, 3, 0, t.leaf.chroma_mode, false, true);
            const float split8 = t.split8 + t.leaf.cost;
            if (split8 > t.ns_cost_cur) { // :1125-1145: the unsplit 8x8 wins, put it back
                t.rbx = t.bx;
                t.rby = t.by;
                t.rlg = t.lg;
                t.rl = t.ns_luma_cur;
                t.rc = t.ns_chroma_cur;
                q.kind = K_NOP;
                req_copy(q, COPY_RESTORE, 3, 1 + t.level, t.rbx, t.rby, t.rlg);
                t.cont = T_REGEN_DONE;
                return true;
            }
            t.ret = split8;
            cont = T_RETURN;
            break;
        }
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
                req_copy(q, COPY_RESTORE, 3, 1 + pl, t.rbx, t.rby, t.rlg);
                t.cont = T_REGEN_DONE;
                return true;
            }
            t.ret = acc;
            break; // cont stays T_RETURN
        }
        // ---- final pass (ctu_encoder.rs:1421-1461): coding order = z-order over the 4x4 units; a
        // CU is emitted at its top-left unit (luma TB, then the chroma TBs) ----
        case T_FINAL_Z: {
            const int z = t.z;
            if (z == 64) {
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
        default: // T_FZ_NEXT
            t.z = (uint8_t)(t.z + 1);
            cont = T_FINAL_Z;
            break;
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
    __syncthreads();
}


__device__ __forceinline__ void encode_ctu(Ctx& c, const PicBufs& pb, int ctu_col, int ctu_row, int* overflow) {
    const CONST_AS DevConst* k = c.k;
    const int W = k->W;
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
    GLOBAL_AS uint8_t* const rec = AS_GLOBAL(uint8_t, pb.rec[0]);
    // neighbour border of the reconstruction: row -1 (x = -4..67) and columns -4..-1
    for (int i = LANE; i < 72; i += 64) {
        const int gx = c.ctu_x - 4 + i, gy = c.ctu_y - 1;
        SH.recYtop[i] = (gx >= 0 && gx < W && gy >= 0) ? rec[(size_t)gy * W + gx] : 0;
    }
    for (int i = LANE; i < 32 * 4; i += 64) {
        const int y = i >> 2, x = (i & 3) - 4;
        const int gx = c.ctu_x + x, gy = c.ctu_y + y;
        SH.recY[y * 36 + x + 4] = gx >= 0 ? rec[(size_t)gy * W + gx] : 0;
    }
    for (int comp = 1; comp < 3; ++comp) {
        for (int i = LANE; i < 40; i += 64) {
            const int gx = (c.ctu_x >> 1) - 4 + i, gy = (c.ctu_y >> 1) - 1;
            SH.recCtop[comp - 1][i] = (gx >= 0 && gx < Wc && gy >= 0) ? rec[plane_off(c, comp) + (size_t)gy * Wc + gx] : 0;
        }
        for (int i = LANE; i < 16 * 4; i += 64) {
            const int y = i >> 2, x = (i & 3) - 4;
            const int gx = (c.ctu_x >> 1) + x, gy = (c.ctu_y >> 1) + y;
            SH.recC[comp - 1][y * 20 + x + 4] = gx >= 0 ? rec[plane_off(c, comp) + (size_t)gy * Wc + gx] : 0;
        }
    }
    // tile.rs:49-58: planes start at zero
    for (int i = LANE; i < 32 * 32; i += 64) SH.recY[(i >> 5) * 36 + (i & 31) + 4] = 0;
    for (int comp = 1; comp < 3; ++comp)
        for (int i = LANE; i < 256; i += 64) SH.recC[comp - 1][(i >> 4) * 20 + (i & 15) + 4] = 0;
    if (LANE < 8)
        SH.left_mode[LANE] =
            c.ctu_x > 0 ? AS_GLOBAL(uint8_t, pb.luma_mode)[(size_t)((c.ctu_y >> 2) + LANE) * (W >> 2) + (c.ctu_x >> 2) - 1] : 0;
    WSYNC();
    // ---- the search + final pass: one evaluator, driven by the coroutines ----
    static_assert(sizeof(Lds) * WPB + sizeof(LdsTab) <= 81920, "two workgroups per CU need <= 80 KB each");
    SH.st.cont = T_START;
    SH.st.in_leaf = 0;
    SH.st.pend = 0;
    SH.st.max_depth = (uint8_t)k->max_depth;
    Res r = {};
    Req q = {};
    for (;;) {
        PROF_MARK(tc0_);
        const bool more = ctu_step(c, r, q);
        PROF_MARK(tc1_);
        PROF_ADD2(PH_CTRL, tc0_, tc1_);
        PROF_ADD2(PH_NSTEP, 0, 1);
        PROF_ADD2(PH_NFULL, 0, (q.kind == K_FULL ? 1 : 0));
        if (!more) break;
        r = evaluate(c, pb, q, overflow);
    }
    const float cost = SH.st.ctu_cost;
    // store recon + decisions
    if (c.write) {
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
    }
#ifdef WRENC_PROFILE
    PROF_MARK(tt1_);
    PROF_ADD2(PH_TOTAL, tt0_, tt1_);
    __syncthreads();
    if (threadIdx.x < PH_COUNT) atomicAdd(&g_prof[threadIdx.x], s_prof[threadIdx.x]);
#endif
}

} // namespace wrenc
