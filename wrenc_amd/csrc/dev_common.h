// dev_common.h -- constants, per-launch / per-picture structures, the per-wave LDS working set, search state, wave helpers, tile access, availability
// Part of the gfx950 device code of the RD-search path; see wrenc_dev.h for the overall model.
#pragma once
#include <cstddef>

namespace wrenc {

enum { PLANAR = 0, DC = 1, LT_CCLM = 81, L_CCLM = 82, T_CCLM = 83 };
enum { TREE_SINGLE = 0, TREE_DUAL_LUMA = 1, TREE_DUAL_CHROMA = 2 };

// Constants resolved on the host (see wrenc_gpu_config in include/wrenc_gpu.h).
struct DevConst {
    int32_t W, H, qp, max_depth, ctu_cols, ctu_rows;
    int32_t lsc;              // quantizer.rs:617-622 (16*LEVEL_SCALE[0][(qp+1)%6]) << ((qp+1)/6)
    uint32_t div_magic;       // floor(2^k / lsc) + 1, k = 26 + ceil(log2 lsc): exact n / lsc = mul_hi(n, magic) >> (k - 32) for n < 2^26
    uint32_t div_shift;       // k - 32
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
    // MFMA experiment (dev_transform.h, fwd_dct32_mfma): the 32-point basis as signed bytes (|T| <= 90),
    // [u][x] for the stage that contracts over x, and with the k order of an MFMA accumulator
    // ([v][h][j] = T[v][8 (j / 4) + 4 h + j % 4]) for the stage that contracts over the first stage's output
    int8_t dct32_a[32][32];
    int8_t dct32_p[32][2][16];
    // the same for the inverse transform (inv_dct32_mfma): [y][i] = T[i][y] for its first stage, [x][h][j] =
    // T[8 (j / 4) + 4 h + j % 4][x] for its second; both stages feed 16-bit operands as a signed high byte and a low
    // byte minus 128, so 128 * sum_i T[i][n] comes back with the rounding constant: idct32_k1[y] = 128 S[y] + 64,
    // idct32_k2[h][w] = 128 S[x] + 2048 for the x of accumulator w in lane half h
    int8_t idct32_b[32][32];
    int8_t idct32_p[32][2][16];
    int32_t idct32_k1[32];
    int32_t idct32_k2[2][16];
    // The head proof's per-coefficient conditions (dev_quant.h, head_alpha) as range tests, per block size lg - 2: a coefficient tc
    // (a) does not end the region  <=>  (unsigned)(tc + r[0]) < (unsigned)r[1];  (b) the same at the DC position with r[2], r[3];
    // (c) has a quotient below 2  <=>  (unsigned)(tc + r[4]) < (unsigned)r[5].  Filled by the host from the same formulas
    // (fill_head_ranges; test_head_ranges_kernel compares them with the device functions over every 16-bit coefficient).
    int32_t head_rng[4][6];
    // Which of a block's five reference segments (bit 0 below-left, 1 left, 2 corner, 3 above, 4 above-right) the coding
    // order makes available, for a block at (bx, by) of log2 size lg inside a CTU away from every picture edge:
    // [lg - 2][(by / 4) * 8 + bx / 4] (block_avail_mask adds the edges).  From the formulas themselves on the host
    // (fill_avail_tab); test_avail_tab_kernel holds the two against each other at every position of small pictures.
    uint8_t avail_tab[4][64];
};

// Pointers that are loaded from memory (PicBufs) lose their address space; these casts tell the
// compiler they are global memory, so that it emits global_* instead of flat_* accesses.
#define GLOBAL_AS __attribute__((address_space(1)))
// The constant block is written by the host before the launch and never during it.
#define CONST_AS __attribute__((address_space(4)))
#define AS_GLOBAL(T, p) ((GLOBAL_AS T*)(p))
#define LDS_AS __attribute__((address_space(3)))

// Per-wave global scratch: kReconSlots saved reconstructions (slot 0: best candidate of the running leaf, up to 32x32;
// 1 + level: unsplit candidate of the open node at that tree level, i.e. 32x32, 16x16, 8x8), each the block's luma
// samples followed by its Cb and Cr samples, back to back: 28 cache lines per wave.
constexpr int kReconSlots = 4;
constexpr int kSlotBytes = 1536;
__host__ __device__ constexpr int slot_offset(int slot) { return slot < 2 ? slot * kSlotBytes : (slot == 2 ? 2 * kSlotBytes : 2 * kSlotBytes + 384); }
constexpr int kWaveScratch = 2 * kSlotBytes + 384 + 128;

// One picture's device buffers.
// Layout of the two CTU-granular buffers of a picture (what one CTU needs sits in whole cache lines of its own;
// rows of the planar picture are shared by four CTUs of four different anti-diagonals):
//   org_t:  per CTU kOrgTile bytes: Y 32x32 row-major | Cb 16x16 | Cr 16x16     (written by retile_kernel at upload)
//   border: per CTU kBorderBytes: the CTU's four rightmost reconstructed columns, Y 32 rows x 4 | Cb 16 x 4 |
//           Cr 16 x 4, then the luma modes of its rightmost 4x4 column (8): what the CTU to its right loads
constexpr int kOrgTile = 1536;
constexpr int kBorderBytes = 272;
struct PicBufs {
    const uint8_t* org[3];
    const uint8_t* org_t;
    uint8_t* border;
    uint8_t* rec[3];
    int16_t* lev[3];
    // per CTU four words (one per 16x16 quadrant): which 4x4 blocks of lev[] may hold a non-zero level.  The planes start
    // zeroed and every encode keeps the rule "a block whose bit is clear is zero", so the final pass stores the levels of
    // the transform blocks that have any, zeroes the ones that had some the last time, and leaves the rest alone.
    uint32_t* lev_dirty;
    uint8_t* cu_log2;
    uint8_t* luma_mode;
    uint8_t* chroma_mode;
    float* ctu_cost;
};

// LDS working set of one wave / one CTU.
// The lane index, behind an optimisation barrier.  The CTU search is one long loop; whatever the compiler can derive
// from a plain `threadIdx.x & 63` alone (row and quad numbers, per-lane table addresses of every stage) it computes
// once in front of that loop and keeps for the whole kernel -- which at 96 VGPRs means a dozen spilled registers,
// written to scratch by every CTU and reloaded in front of the stages' inner loops.  Behind the barrier the index is
// an ordinary value: what depends on it is recomputed where it is used (a v_and or two), 1 spilled VGPR instead of
// 14, +1 % frames/s (gpurun_out/r3s/lane.log).
__device__ __forceinline__ int lane_fresh() {
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}
#define LANE lane_fresh()
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// SSD and level cost of the luma block and of the chroma pair of one evaluated candidate
struct EvalParts {
    uint32_t ssd_y, ssd_c; // <= 1024 * 255^2: fits 32 bits
    long long lvl_y, lvl_c;
};


// A wave-uniform field of the search state, resident in LDS: reads come back through
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

// EvalParts as kept in the search state
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
    UF<uint8_t> need_org;               // originals of the block not staged in LDS yet (blocks <= 16x16)
    UF<uint8_t> step;
    UF<uint8_t> cur_mode, mode, cclm_mode, dm_mode;
    UF<uint8_t> holder, evalr;          // team schedule: the member whose slot 0 holds the best candidate / the CCLM evaluator
    UF<uint8_t> luma_mode, chroma_mode; // result
    UF<uint8_t> best_cls;               // header-bit class (mpm_class) of the best luma mode
    UF<uint8_t> need_save, tile_best;   // best candidate's reconstruction: not saved yet / still in the tile
    UF<uint8_t> slot;                   // the scratch slot the best candidate's reconstruction is saved to
    UF<float> best_cost;                // best of {planar, DC} so far / of {planar, DC, dir}
    UF<float> cur_cost, c0;
    UF<float> cost;                     // result
    EvalPartsU e_best;
};

// A step of the leaf search works on a REGISTER copy of LeafSt: one LDS read fetches the whole struct
// (lane i reads dword i), every field then comes out of that vector register with v_readlane instead of an LDS
// round trip of its own (a step used to make ~20 dependent LDS reads: profiles/r02_issue_model.md).  Writes go
// to the register copy and through to LDS, so the next step's snapshot and the tree-level code see them.
template <class T>
struct SF {
    T v;
    UF<T>* p;
    __device__ __forceinline__ T get() const { return v; }
    __device__ __forceinline__ operator T() const { return v; }
    __device__ __forceinline__ SF& operator=(T x) {
        v = x;
        p->set(x);
        return *this;
    }
    __device__ __forceinline__ SF& operator=(const SF& o) { return *this = o.v; }
    __device__ __forceinline__ SF& operator+=(int x) { return *this = (T)(v + x); }
    __device__ __forceinline__ SF& operator-=(int x) { return *this = (T)(v - x); }
};
struct EvalPartsSF {
    SF<uint32_t> ssd_y, ssd_c;
    SF<long long> lvl_y, lvl_c;
    __device__ __forceinline__ EvalParts get() const {
        EvalParts e;
        e.ssd_y = ssd_y;
        e.ssd_c = ssd_c;
        e.lvl_y = lvl_y;
        e.lvl_c = lvl_c;
        return e;
    }
};
struct LeafSF {
    SF<uint8_t> cont, op_ml, op_mc, op_act, tree, bx, by, lg, need_refs0, need_refs1, need_org, step, cur_mode, mode,
        cclm_mode, dm_mode, holder, evalr, luma_mode, chroma_mode, best_cls, need_save, tile_best, slot;
    SF<float> best_cost, cur_cost, c0, cost;
    EvalPartsSF e_best;
};
static_assert(sizeof(LeafSt) == 64, "snap_leaf reads the struct as 16 dwords");
__device__ __forceinline__ LeafSF snap_leaf(LeafSt& l) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // after this step's own stores to the struct (leaf_init)
    // (a plain load: the fence above orders it after the stores; `volatile` would turn it into a FLAT access)
    const int w = ((const int*)&l)[LANE & 15];
    LeafSF s;
#define SNAP_U8(f)                                                                                          \
    s.f.v = (uint8_t)((unsigned)__builtin_amdgcn_readlane(w, (int)(offsetof(LeafSt, f) >> 2)) >> (8 * (offsetof(LeafSt, f) & 3))); \
    s.f.p = &l.f
#define SNAP_F32(f)                                                                       \
    s.f.v = __int_as_float(__builtin_amdgcn_readlane(w, (int)(offsetof(LeafSt, f) >> 2))); \
    s.f.p = &l.f
    SNAP_U8(cont); SNAP_U8(op_ml); SNAP_U8(op_mc); SNAP_U8(op_act); SNAP_U8(tree); SNAP_U8(bx); SNAP_U8(by); SNAP_U8(lg);
    SNAP_U8(need_refs0); SNAP_U8(need_refs1); SNAP_U8(need_org); SNAP_U8(step); SNAP_U8(cur_mode); SNAP_U8(mode);
    SNAP_U8(cclm_mode); SNAP_U8(dm_mode); SNAP_U8(holder); SNAP_U8(evalr); SNAP_U8(luma_mode); SNAP_U8(chroma_mode);
    SNAP_U8(best_cls); SNAP_U8(need_save); SNAP_U8(tile_best); SNAP_U8(slot);
    SNAP_F32(best_cost); SNAP_F32(cur_cost); SNAP_F32(c0); SNAP_F32(cost);
#undef SNAP_U8
#undef SNAP_F32
    constexpr int eb = (int)(offsetof(LeafSt, e_best) >> 2);
    s.e_best.ssd_y.v = (uint32_t)__builtin_amdgcn_readlane(w, eb + (int)(offsetof(EvalPartsU, ssd_y) >> 2));
    s.e_best.ssd_y.p = &l.e_best.ssd_y;
    s.e_best.ssd_c.v = (uint32_t)__builtin_amdgcn_readlane(w, eb + (int)(offsetof(EvalPartsU, ssd_c) >> 2));
    s.e_best.ssd_c.p = &l.e_best.ssd_c;
    {
        constexpr int o = eb + (int)(offsetof(EvalPartsU, lvl_y) >> 2);
        s.e_best.lvl_y.v = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(w, o + 1) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane(w, o));
        s.e_best.lvl_y.p = &l.e_best.lvl_y;
    }
    {
        constexpr int o = eb + (int)(offsetof(EvalPartsU, lvl_c) >> 2);
        s.e_best.lvl_c.v = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(w, o + 1) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane(w, o));
        s.e_best.lvl_c.p = &l.e_best.lvl_c;
    }
    return s;
}

// CTU search + final pass state (see ctu_step)
struct CtuSt {
    UF<uint8_t> cont, in_leaf, xpar; // xpar: parity of the team's next exchange
    UF<uint8_t> level, bx, by, lg, max_depth;
    UF<uint8_t> i8, z, rl, rc;       // 4x4 child index, final-pass z-order index, regen modes
    UF<uint8_t> rbx, rby, rlg;       // regen block
    UF<uint8_t> ns_luma_cur, ns_chroma_cur;
    UF<uint8_t> pend, pbx, pby, plg, pslot; // reconstruction save to attach to the next request
    // team schedule: a decided leaf's winner still to be pulled into every member's tile, done by the driver loop
    // before the next evaluation.  dp0 = on | comps << 1 | from << 3 | (lg - 2) << 5, dp1 = bx / 4 | (by / 4) << 3
    UF<uint8_t> dp0, dp1;
    // level schedule (team kernel at max-split-depth 3, dev_search.h): on | this member's unit | end of its final-pass range
    UF<uint8_t> lvmode, lv_i, zend;
    UF<uint8_t> fz_on;               // the final pass has begun (Lds::lev_was / lev_now are loaded)
    UF<float> ret, ns_cost_cur, split8, ctu_cost;
    UF<float> lv_acc0, lv_acc1; // level schedule: running split cost of the open 32x32 / 16x16 node
    LeafSt leaf;
};

// Team schedule (ctu_search_team_kernel): what a member publishes to its team after a stage (dev_search.h,
// leaf_step_team).  A full evaluation publishes its parts, a SAD list its first minimum and its first entry.
struct XRes {
    uint32_t ssd_y, ssd_c;
    long long lvl_y, lvl_c;
};

// element offsets into Lds::refs: left (index 0 = corner) / above references of luma unfiltered,
// luma filtered, Cb, Cr
constexpr int R_L0 = 0, R_A0 = 66, R_LF = 130, R_AF = 196, R_LC0 = 260, R_LC1 = 294, R_AC0 = 328, R_AC1 = 360;

typedef uint8_t ref_t; // a reference sample (reconstructed or [1 2 1]-filtered: 0..255)
struct __attribute__((aligned(16))) Lds {
    // Transform working set, time-multiplexed through a full evaluation (dev_search.h):
    //   r1: residual -> coefficients -> Viterbi chunk costs / levels -> inverse-transform intermediate -> residual
    //   r2: stage-1 DCT output (i32, blocks up to 16x16; the 32x32 transform keeps it in MFMA accumulators) ->
    //       scan-order coefficients (+ chroma chunk costs of the merged pass) -> dequantised^T
    // Bytes [kOrgLeaf, 2048) of r2 hold the originals of a block of at most 16x16 for its whole leaf search
    // (dev_predict.h); no stage of a block that small reaches them.
    // 2 KB each: five workgroups of four waves per CU (LDS is what caps the waves in flight, and the kernel's
    // throughput is proportional to them: profiles/r02_issue_model.md).
    int16_t r1[1024];
    int32_t r2[512];
    // reference samples of the current block, built once per (block, component) and reused by
    // every candidate mode: luma unfiltered + [1 2 1]-filtered, chroma unfiltered
    // one array addressed by element offsets (R_*), so that choosing among the sets is integer
    // arithmetic on a DS address, never a pointer select
    ref_t refs[392];
    uint8_t recYtop[72];       // y = -1, x = -4..67 (index x+4)
    uint8_t recY[32 * 36];     // x = -4..31 (index x+4), stride 36
    uint8_t recCtop[2][40];    // y = -1, x = -4..35
    uint8_t recC[2][16 * 20];  // x = -4..15, stride 20
    // 512 bytes with SIX users; who holds which bytes, and for how long (profiles/r04_wrong_sads.md: an overlap here is what
    // round 3's wrong-SAD builds had):
    //   [0, nb P / 2)   trellis decisions of a quantiser call, one u16 mask per sub-block and state: 512 B for a 32x32 block,
    //                   144 B for a pack of three 8x8 candidates, 384 B for a pack of two 16x16 ones, 32 B for four 4x4
    //                   blocks; dead when the call returns
    //   [128, 256)      SAD lists: ptab, ptab2 (16 entries each); [4 WRENC_SAD_SUMS_AT, + 64): sums.  Live inside one
    //                   sad_list_angular call
    //   [160, 448)      PRED_PARK (kParkByte): predictions, then reconstructions, of an 8x8 pack; live from pack8_eval's
    //                   predictions to pack8_to_tile -- a pack's winner is in the tile BEFORE the leaf's SAD search runs
    //                   (the server's pack A: until the owner has taken the winner, job_ack)
    //   [384, 512)      PRED_PARK16: the last 128 bytes of a 16x16 pack's park, behind its 384 bytes of decisions; same
    //                   lifetime as PRED_PARK
    //   [448, 504)      team kernel, member 0 as pack-A server of 4x4 leaves (kSrv4Byte): results until the owner's job4_ack
    uint32_t decw[128];
    int32_t q_istar[2];        // shared-Viterbi hand-off, per block: first position with a non-zero state-0 level
    int32_t q_active;          // this wave's TB takes part in the shared Viterbi
    uint32_t fsum;             // final pass: checksum of the search's reconstruction of the block being re-made
    uint16_t q_pm[3][4][4];    // per block and sub-block of the chunk: parity masks (delta 0, 1), state-0 flag
    uint8_t cu_log2[64];       // per 4x4 luma unit
    uint8_t luma_mode[64];
    uint8_t chroma_mode[16];   // per 8x8 luma unit
    uint8_t left_mode[8];      // luma mode of the CU left of the CTU, per 4 rows
    // per tree level: no-split cost, running split cost (the search of the wave schedule); in the final pass the same
    // cells hold the CTU's "these 4x4 blocks of levels may be non-zero in the slot's planes" bits, one word per 16x16
    // quadrant (bits 0..15: luma units, 16..19: chroma blocks): as the previous encode of the slot left them / as
    // this one leaves them (dev_search.h full_back)
    union {
        float ns_cost[4];
        uint32_t lev_was[4];
    };
    union {
        float split_cost[4];
        uint32_t lev_now[4];
    };
    uint8_t ns_luma[4], ns_chroma[4], child[4];
    CtuSt st;                  // state of the search (dev_search.h)
    XRes xr[2];                // team schedule: this member's published result, double-buffered by stage parity
};

// Per-wave uniform context.
// Per-wave uniform context, passed BY VALUE (a few registers) so that the (inlined)
// stage functions never reload it from memory.
struct Ctx {
    const CONST_AS DevConst* k;         // constant address space: uniform reads become scalar loads
    const GLOBAL_AS uint8_t* org;       // originals of THIS CTU: the kOrgTile bytes of its tile in PicBufs::org_t (read-only)
    int W, WH;                          // luma width, luma plane size
    uint8_t* pred_scratch;              // building-block test kernel only (PRED_SCRATCH): where predict() puts its bytes
    GLOBAL_AS uint8_t* slots;           // kReconSlots saved reconstructions of this wave (see copy_block)
    unsigned long long* mismatch;
    int ctu_x, ctu_y; // luma, picture coordinates
    int cu32_mode;    // in-CTU neighbour lookups during search resolve to the root CU
    int write;        // 0 for a padding wave (batch not a multiple of WPB): compute, never store
    int member;       // team schedule: this wave's place in its team of kTeam waves (0 otherwise)
    int store;        // team schedule: this wave's final-pass blocks are of a real picture (it stores their levels); = write otherwise
    int solo;         // team kernel: no pooled (workgroup-wide) trellis walk, whatever a request says
    int trace;        // diagnostic trace: this wave's evaluations are of a real picture
};

// LDS: one working set per wave (= per CTU), WPB waves per workgroup, plus tables shared
// by the workgroup.  File scope so that every access is a DS instruction (no FLAT ops).
// The waves of a workgroup process the SAME CTU position of WPB different pictures, so
// they execute the same schedule; the 4-lane Viterbi of all WPB transform blocks is run by
// one wave in WPB quads at once (see quantize()).
#ifndef WRENC_WPB
#define WRENC_WPB 4
#endif
constexpr int WPB = WRENC_WPB;
// Workgroups resident per CU, which LDS decides: 5 x 4 waves = 5 waves per SIMD (96 VGPRs each).
constexpr int kWorkgroupsPerCU = 20 / WPB;
// Team schedule: kTeam waves share ONE CTU (independent candidates of a leaf search run side by side), a
// workgroup holds WPB / kTeam teams = the same CTU of that many pictures.
constexpr int kTeam = 4;
// Level schedule: what the members of a team tell each other about the node being decided at tree level L (0: 32x32,
// 1: 16x16, 2: 8x8) -- the unsplit candidate's cost and modes from member L, the split candidate's cost from member
// L + 1 -- and the arrival counters of the two meeting points of a decision (monotone over a CTU).
struct LvBox {
    float ns_cost[3], split[3];
    uint8_t ml[3], mc[3];
    uint16_t pad_;
    uint32_t cnt[3][2];
    // the pack-A server (member 0 runs pack {planar, DC} of member 2's 8x8 leaves, dev_search.h lv_serve): server is
    // polling | jobs posted / done / consumed | member 2 has no more jobs; the posted job's block
    uint32_t srv_ready, job_posted, job_done, job_ack, job_fin;
    uint32_t job4_posted, job4_done, job4_ack, job4_fin; // the same for member 3's 4x4 luma leaves
    uint8_t job_bx, job_by, job4_bx, job4_by;
};
struct LdsTab {
    int32_t ldq[256];
    int32_t lv[256];
    int8_t fc[32][4]; // common.rs:153 (copied from the constant block)
    LvBox lvb;
#ifdef WRENC_EXP_LDS_PAD
    char pad[WRENC_EXP_LDS_PAD]; // occupancy experiments only (profiles/r02_issue_model.md)
#endif
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
    c.member = uni(c.member);
    c.store = uni(c.store);
    c.solo = uni(c.solo);
    c.trace = uni(c.trace);
    return c;
}

// Unroll knobs of the stage loops (experiments: tools/README.md); 0 = the compiler's choice
#define WRENC_PRAGMA_(x) _Pragma(#x)
#define WRENC_UNROLL(n) WRENC_PRAGMA_(unroll n)
#ifndef WRENC_U_DCT
#define WRENC_U_DCT 1
#endif
#ifndef WRENC_U_SAD
#define WRENC_U_SAD 0
#endif

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
enum { PH_PREDICT, PH_FDCT, PH_QPRE, PH_QBACK, PH_QTRACE, PH_DEQ, PH_IDCT, PH_RECON, PH_TOTAL, PH_CTRL, PH_REFS, PH_SKIP, PH_NSTEP, PH_NFULL, PH_PSZ, PH_PSZ_END = PH_PSZ + 8, PH_PCNT, PH_PCNT_END = PH_PCNT + 8, PH_QB_PRE, PH_QB_WAIT1, PH_QB_WALK, PH_QB_WAIT2, PH_XCHG, PH_COPY, PH_CB, PH_CB_END = PH_CB + 32, PH_CBN, PH_CBN_END = PH_CBN + 32, PH_MEM, PH_MEM_END = PH_MEM + 16, PH_ST, PH_ST_END = PH_ST + 48, PH_STN, PH_STN_END = PH_STN + 12, PH_EV, PH_EV_END = PH_EV + 64, PH_EVN, PH_EVN_END = PH_EVN + 64, PH_QZ, PH_QZ_END = PH_QZ + 4, PH_LEAF, PH_LEAF_END = PH_LEAF + 12, PH_L4, PH_L4_END = PH_L4 + 6, PH_HIST, PH_HIST_END = PH_HIST + 64, PH_COUNT }; // PH_HIST: CTU durations, buckets of 2^17 ticks
__device__ unsigned long long g_prof[PH_COUNT];
__shared__ unsigned long long s_prof[PH_COUNT];
#define PROF_T0() const unsigned long long prof_t0_ = __builtin_readcyclecounter()
#define PROF_ADD(ph) do { if (threadIdx.x == 0) s_prof[ph] += __builtin_readcyclecounter() - prof_t0_; } while (0)
#define PROF_MARK(var) const unsigned long long var = __builtin_readcyclecounter()
#define PROF_ADD2(ph, a, b) do { if (threadIdx.x == 0) s_prof[ph] += (b) - (a); } while (0)
// per member of team 0 (waves 0..3): bucket k of PH_MEM + 4 * k + member
#define PROF_ADDM(k, a, b) do { if (LANE == 0 && WAVE < 4) s_prof[PH_MEM + 4 * (k) + WAVE] += (b) - (a); } while (0)
#else
#define PROF_ADDM(k, a, b)
#define PROF_T0()
#define PROF_ADD(ph)
#define PROF_MARK(var)
#define PROF_ADD2(ph, a, b)
#endif

// Diagnostic build (-DWRENC_TRACE, never the product library): every candidate evaluation of the
// search is appended to a device buffer, 8 ints per record, in the layout of the oracle's trace
// (x, y, log2 size, tree, kind, luma mode, chroma mode, f32 bits; the CPU checker records the same layout).
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
#ifdef WRENC_EXP_NO_ORG // timing experiment only (wrong results): what the global loads of originals cost
    return (x * 7 + y * 13 + pc * 31) & 255;
#endif
    return pc == 0 ? c.org[y * 32 + x] : c.org[768 + pc * 256 + y * 16 + x];
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
// The five segment availabilities of build_refs (dev_predict.h) as one mask, from the formulas above (the reference's:
// above_right_avail, below_left_avail, five nb_avail -- some 150 scalar instructions per block, a tenth of the kernel's
// scalar stream at max-split-depth 3, profiles/r04_issue_model.md).
__device__ __forceinline__ int block_avail_formula(Ctx c, int tx, int ty, int tlg, int st) {
    const int tn = 1 << tlg;
    const int gx = c.ctu_x + tx, gy = c.ctu_y + ty;
    const bool ar = above_right_avail(c, tx, ty, tlg);
    const bool bl = below_left_avail(c, tx, ty, tlg);
    int avm = 0;
    avm |= nb_avail(c, gx, gy, tn, gx - st, gy + tn, ar, bl) ? 1 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx - st, gy, ar, bl) ? 2 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx - st, gy - st, ar, bl) ? 4 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx, gy - st, ar, bl) ? 8 : 0;
    avm |= nb_avail(c, gx, gy, tn, gx + tn, gy - st, ar, bl) ? 16 : 0;
    return avm;
}
// The same from the table: what the coding order allows inside the CTU (and across its borders) is a function of the
// block's place in the CTU alone; the picture's edges take segments away: everything left of the block needs a column
// left of it, everything above a row above it, above-right a column right of it, below-left a row below it.
#ifndef WRENC_AVAIL_TAB
#define WRENC_AVAIL_TAB 1 // 0: the formulas at every call (until round 4; for A/B runs)
#endif
__device__ __forceinline__ int block_avail_mask(Ctx c, int tx, int ty, int tlg, int st) {
    if (!WRENC_AVAIL_TAB) return block_avail_formula(c, tx, ty, tlg, st);
    const int tn = 1 << tlg;
    const int gx = c.ctu_x + tx, gy = c.ctu_y + ty;
    const int in_ctu = c.k->avail_tab[tlg - 2][(ty >> 2) * 8 + (tx >> 2)];
    const int left = gx > 0 ? 7 : 0, top = gy > 0 ? 28 : 0;       // BL, L, corner | corner, A, AR
    const int right = gx + tn < c.W ? 31 : 15, bottom = gy + tn < c.k->H ? 31 : 30;
    return in_ctu & ((left | 24) & (top | 3)) & right & bottom;
}

} // namespace wrenc
