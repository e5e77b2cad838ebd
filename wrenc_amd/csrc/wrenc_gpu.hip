// wrenc_gpu.hip -- kernels + C ABI (include/wrenc_gpu.h) of the MI355X all-intra
// RD-search path.  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -o libwrenc_gpu.so wrenc_gpu.hip
//
// Scheduling: a picture's CTUs depend on their left, above-left, above and
// above-right neighbours (encoder_context.rs:934-948), so CTU (r, c) can run once
// every CTU on anti-diagonal c + 2r - 1 is done.  One launch processes one
// anti-diagonal of every picture of the batch (one wave per CTU); stream order
// between launches is the only synchronisation between CTUs.  (The one in-kernel
// wait is a workgroup's acquire of a scratch region from a bitmap that always has
// more regions than workgroups can be resident, see ctu_search_kernel.)
#include "../../include/wrenc_gpu.h"
#include "wrenc_dev.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace wrenc;

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// Scratch regions for resident workgroups (WPB x kWaveScratch bytes each: the saved reconstructions of dev_search.h
// copy_block), handed out through a bitmap.  The pool covers the workgroups that can be resident at once, not the
// ones of a launch, and it is PARTITIONED BY XCD: a workgroup takes the lowest free region of the partition of the XCD
// it runs on (s_getreg XCC_ID).  The regions in use on an XCD are then the same 160 or so (32 CUs x kWorkgroupsPerCU)
// for the whole run, written and re-read through that XCD's own L2 only: about 2 MB that can stay in the 4 MB L2
// instead of going out to the fabric with every save, and no other L2 ever holds a line of them, so giving a region
// back needs no L2 write-back.  (Round 2 picked a region anywhere in a pool of 4096 from a hash of the workgroup
// index and ran __threadfence() = buffer_wbl2 sc1 + buffer_inv sc1 at every workgroup's end: every saved
// reconstruction went out to memory.)  A region is only ever used by one workgroup at a time and nothing is read
// that the same workgroup did not write, so its contents need no hand-over.
// Should a partition ever be full (another device, another occupancy) a workgroup takes a region of the OVERFLOW
// partition, which any XCD may use and whose users write their L2 lines back before they release it, as round 2 did;
// the word behind the bitmap counts how often that happened (wrenc_gpu_test_scratch_overflows: 0 on MI355X).
constexpr int kXcdMax = 8;
constexpr int kRegionsPerXcd = 256;    // whole bitmap words; >= CUs per XCD x kWorkgroupsPerCU (32 x 5 = 160)
constexpr int kAffineRegions = kXcdMax * kRegionsPerXcd;
constexpr int kOverflowRegions = 2048; // >= every workgroup that can be resident, should XCC_ID not tell them apart
constexpr int kScratchSlots = kAffineRegions + kOverflowRegions;
constexpr int kSlotMapWords = kScratchSlots / 64 + 1; // + the overflow counter

__device__ __forceinline__ int xcc_id() {
    // hwreg(HW_REG_XCC_ID = 20, offset 0, size 4): the XCD this wave runs on
    return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (kXcdMax - 1));
}
// the lowest free region of bitmap words [w0, w0 + nw), or -1 if they are all taken
__device__ __forceinline__ int take_region(unsigned long long* slot_map, unsigned w0, unsigned nw) {
    for (unsigned w = w0; w < w0 + nw; ++w) {
        unsigned long long cur = __hip_atomic_load(&slot_map[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (~cur) {
            const int b = __ffsll((unsigned long long)~cur) - 1;
            const unsigned long long bit = 1ULL << b;
            const unsigned long long old = atomicOr(&slot_map[w], bit);
            if (!(old & bit)) return (int)(w * 64 + b);
            cur = old; // somebody else got it: look again in what the word held then
        }
    }
    return -1;
}
// (The region number travels through a cell of wave 0's LDS that the search only uses later.)
__device__ __forceinline__ int acquire_scratch(unsigned long long* slot_map) {
    if (threadIdx.x == 0) {
        int slot = take_region(slot_map, (unsigned)xcc_id() * (kRegionsPerXcd / 64), kRegionsPerXcd / 64);
        if (slot < 0) {
            atomicAdd(&slot_map[kScratchSlots / 64], 1ULL);
            // the overflow partition has a region for every workgroup that can be resident: a free one turns up
            while (slot < 0) slot = take_region(slot_map, kAffineRegions / 64, kOverflowRegions / 64);
        }
        SHW[0].q_istar[0] = slot;
    }
    __syncthreads();
    const int scratch_slot = uni(SHW[0].q_istar[0]);
    __syncthreads(); // everybody has read the cell before the search may overwrite it
    return scratch_slot;
}
// give the scratch region back once every wave is through with it
__device__ __forceinline__ void release_scratch(unsigned long long* slot_map, int scratch_slot) {
    if (scratch_slot >= kAffineRegions) __threadfence(); // overflow partition: the next user may sit behind another L2
    __syncthreads();
    if (threadIdx.x == 0) atomicAnd(&slot_map[scratch_slot >> 6], ~(1ULL << (scratch_slot & 63)));
}

// Originals of a freshly uploaded picture, planar -> one contiguous tile per CTU (PicBufs::org_t): the search then
// fetches a CTU's 1.5 KB as 12 whole cache lines instead of 64 row pieces of lines it shares with three other CTUs.
// One wave per CTU, lane = dword.
__global__ __launch_bounds__(64) void retile_kernel(const uint8_t* __restrict__ planar, uint8_t* __restrict__ tiled, int W, int H,
                                                   int ctu_cols) {
    const int ctu = blockIdx.x, col = ctu % ctu_cols, row = ctu / ctu_cols;
    const size_t wh = (size_t)W * H;
    uint32_t* dst = (uint32_t*)(tiled + (size_t)ctu * kOrgTile);
    for (int w = threadIdx.x; w < kOrgTile / 4; w += 64) {
        size_t src;
        if (w < 256) { // luma: 8 dwords per row
            src = (size_t)(row * 32 + (w >> 3)) * W + col * 32 + (w & 7) * 4;
        } else {       // chroma: 4 dwords per row, Cb then Cr
            const int pl = (w - 256) >> 6, ww = (w - 256) & 63;
            src = wh + (size_t)pl * (wh >> 2) + (size_t)(row * 16 + (ww >> 2)) * (W >> 1) + col * 16 + (ww & 3) * 4;
        }
        dst[w] = *(const uint32_t*)(planar + src);
    }
}

// Compact read-back (include/wrenc_gpu.h): one workgroup per picture walks the picture's 4x4 blocks of levels in mask
// order, 1024 at a time: a thread loads one block (four 8-byte rows), the waves' ballots are the mask words, and the
// blocks with a non-zero level go to the payload at the running count + their rank (wave ballots, one LDS scan over
// the 16 waves).  The planes are read once at HBM rate; what crosses PCIe afterwards is the mask and the coded blocks.
__global__ __launch_bounds__(1024) void compact_levels_kernel(const PicBufs* __restrict__ slots, int first_slot, int W, int H,
                                                               uint32_t* masks, size_t mask_words, int16_t* payloads,
                                                               size_t payload_stride_blocks, unsigned* counts) {
    __shared__ unsigned s_wave[16];
    __shared__ unsigned s_base;
    const PicBufs pb = slots[first_slot + blockIdx.x];
    const int16_t* lev = pb.lev[0];
    uint32_t* mask = masks + (size_t)blockIdx.x * mask_words;
    int4* pay = (int4*)(payloads + (size_t)blockIdx.x * payload_stride_blocks * 16);
    const int bw = W >> 2, bh = H >> 2;                 // luma plane in 4x4 blocks
    const int nl = bw * bh, ncb = nl >> 2, total = nl + 2 * ncb;
    const size_t wh = (size_t)W * H;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < total; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        int2 r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0}, r3 = {0, 0};
        if (b < total) {
            // plane, block row / column, stride
            int pb_ = b, pw = bw;
            size_t off = 0;
            int stride = W;
            if (b >= nl) {
                const int c = b - nl;
                const int pl = c >= ncb ? 1 : 0;
                pb_ = c - pl * ncb;
                pw = bw >> 1;
                stride = W >> 1;
                off = wh + (size_t)pl * (wh >> 2);
            }
            const int by = pb_ / pw, bx = pb_ - by * pw;
            const int16_t* p = lev + off + (size_t)(4 * by) * stride + 4 * bx;
            r0 = *(const int2*)p;
            r1 = *(const int2*)(p + stride);
            r2 = *(const int2*)(p + 2 * stride);
            r3 = *(const int2*)(p + 3 * stride);
        }
        const bool nz = (r0.x | r0.y | r1.x | r1.y | r2.x | r2.y | r3.x | r3.y) != 0;
        const unsigned long long bal = __ballot(nz);
        if (lane == 0) {
            if (b < total) mask[b >> 5] = (uint32_t)bal;
            if (b + 32 < total) mask[(b >> 5) + 1] = (uint32_t)(bal >> 32);
            s_wave[wave] = (unsigned)__popcll(bal);
        }
        __syncthreads();
        unsigned before = s_base;
        for (int wv = 0; wv < wave; ++wv) before += s_wave[wv];
        if (nz) {
            const size_t at = (size_t)before + (unsigned)__popcll(bal & ((1ULL << lane) - 1ULL));
            if (at < payload_stride_blocks) {
                pay[2 * at] = make_int4(r0.x, r0.y, r1.x, r1.y);
                pay[2 * at + 1] = make_int4(r2.x, r2.y, r3.x, r3.y);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned sum = 0;
            for (int wv = 0; wv < 16; ++wv) sum += s_wave[wv];
            s_base += sum;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = s_base;
}

// Wave schedule: one workgroup = the same CTU of WPB consecutive pictures, one wave each.
// D3: built for max-split-depth 3 (with the 4x4 leaves of split 8x8 CUs) or for the smaller depths (without)
template <bool D3>
__global__ __launch_bounds__(64 * WPB, 5) void ctu_search_kernel(const DevConst* __restrict__ k,
                                                              const PicBufs* __restrict__ slots, int first_slot,
                                                              int n_pictures, int diag, int r_min, int count,
                                                              uint8_t* pred_scratch, unsigned long long* slot_map,
                                                              unsigned long long* mismatch, int* overflow) {
    const int scratch_slot = acquire_scratch(slot_map);
    const int group = blockIdx.x / count;
    const int j = blockIdx.x - group * count;
    const int row = r_min + j;
    const int col = diag - 2 * row;
    int pic = group * WPB + WAVE;
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    c.mismatch = mismatch;
    c.write = pic < n_pictures ? 1 : 0;
    c.trace = c.write;
    c.store = c.write;
    c.solo = 0;
    c.member = 0;
    if (pic >= n_pictures) pic = n_pictures - 1; // padding wave: same work, no stores
    const PicBufs pb = slots[first_slot + pic];
    c.org = (const GLOBAL_AS uint8_t*)pb.org_t + (size_t)(row * k->ctu_cols + col) * kOrgTile;
    c.W = k->W;
    c.WH = k->W * k->H;
    c.pred_scratch = nullptr;
    c.slots = (GLOBAL_AS uint8_t*)(pred_scratch + ((size_t)scratch_slot * WPB + WAVE) * kWaveScratch);
    int ovf = 0;
    encode_ctu<false, D3>(c, pb, col, row, &ovf);
    if (ovf && LANE == 0) atomicOr(overflow, 1);
    release_scratch(slot_map, scratch_slot);
}

// Team schedule: one workgroup = the same CTU of WPB / kTeam consecutive pictures, kTeam waves each
// (dev_search.h, leaf_step_team).  For encode calls with too few pictures to fill the GPU one wave per CTU.
template <bool D3>
__global__ __launch_bounds__(64 * WPB, 5) void ctu_search_team_kernel(const DevConst* __restrict__ k,
                                                                   const PicBufs* __restrict__ slots, int first_slot,
                                                                   int n_pictures, int diag, int r_min, int count,
                                                                   uint8_t* pred_scratch, unsigned long long* slot_map,
                                                                   unsigned long long* mismatch, int* overflow) {
    const int scratch_slot = acquire_scratch(slot_map);
    constexpr int kTeams = WPB >= kTeam ? WPB / kTeam : 1; // (WPB < kTeam: occupancy experiments, wave schedule only)
    const int group = blockIdx.x / count;
    const int j = blockIdx.x - group * count;
    const int row = r_min + j;
    const int col = diag - 2 * row;
    int pic = group * kTeams + WAVE / kTeam;
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    c.mismatch = mismatch;
    c.member = WAVE & (kTeam - 1);
    c.trace = pic < n_pictures ? 1 : 0;
    c.write = (c.trace && c.member == 0) ? 1 : 0; // one member stores the picture's results
    c.store = c.trace;                            // (every member stores the levels of its share of the final pass)
    c.solo = 1;
    if (pic >= n_pictures) pic = n_pictures - 1;  // padding team: same work, no stores
    const PicBufs pb = slots[first_slot + pic];
    c.org = (const GLOBAL_AS uint8_t*)pb.org_t + (size_t)(row * k->ctu_cols + col) * kOrgTile;
    c.W = k->W;
    c.WH = k->W * k->H;
    c.pred_scratch = nullptr;
    c.slots = (GLOBAL_AS uint8_t*)(pred_scratch + ((size_t)scratch_slot * WPB + WAVE) * kWaveScratch);
    int ovf = 0;
    encode_ctu<true, D3>(c, pb, col, row, &ovf);
    if (ovf && LANE == 0) atomicOr(overflow, 1);
    if (SHT.lvb.pad_ && LANE == 0) atomicOr(overflow, 2); // a meeting point of the level schedule timed out
    release_scratch(slot_map, scratch_slot);
}

// building-block kernels: one wave per block of side 1 << lg
__global__ __launch_bounds__(64) void test_fwd_dct_kernel(const DevConst* __restrict__ k,
                                                          const int16_t* in, int lg, int16_t* out) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    const int nn = 1 << (2 * lg);
    for (int i = threadIdx.x; i < nn; i += 64) SH.r1[i] = in[(size_t)blockIdx.x * nn + i];
    WSYNC();
    fwd_dct_lg(c, lg, 1);
    for (int i = threadIdx.x; i < nn; i += 64) out[(size_t)blockIdx.x * nn + i] = SH.r1[i];
}

// MFMA experiment: the 32x32 forward transform as i8 MFMAs (dev_transform.h); `reps` repeats the transform
// of the same block in place for the micro-benchmark (the result of the last repetition is stored)
__shared__ int32_t s_h32[33 * 32]; // the v_dot2 arm's i32 intermediate (the search kernel has no such room)
__global__ __launch_bounds__(64) void test_fwd_dct32_kernel(const DevConst* __restrict__ k, const int16_t* in, int16_t* out,
                                                           int mfma, int reps) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    for (int rep = 0; rep < reps; ++rep) {
        for (int i = threadIdx.x; i < 1024; i += 64) SH.r1[i] = in[(size_t)blockIdx.x * 1024 + i];
        WSYNC();
        if (mfma)
            fwd_dct32_mfma(c, 0);
        else
            fwd_dct<5, (int)sizeof(s_h32)>(c, 1, 0, (LDS_AS int32_t*)s_h32);
    }
    for (int i = threadIdx.x; i < 1024; i += 64) out[(size_t)blockIdx.x * 1024 + i] = SH.r1[i];
}

// the same for the inverse 32x32 transform: dequantised blocks in (row-major), residuals out
__global__ __launch_bounds__(64) void test_inv_dct32_kernel(const DevConst* __restrict__ k, const int16_t* in, int16_t* out,
                                                           int mfma, int reps) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    for (int rep = 0; rep < reps; ++rep) {
        for (int i = threadIdx.x; i < 1024; i += 64) // transposed load: dT[x][i] = d[i][x]
            ((int16_t*)SH.r2)[(i & 31) * 32 + (i >> 5)] = in[(size_t)blockIdx.x * 1024 + i];
        WSYNC();
        if (mfma)
            inv_dct32_mfma(c, 0);
        else
            inv_dct<5>(c, 1, 0);
    }
    for (int i = threadIdx.x; i < 1024; i += 64) out[(size_t)blockIdx.x * 1024 + i] = SH.r1[i];
}

__global__ __launch_bounds__(64) void test_inv_dct_kernel(const DevConst* __restrict__ k,
                                                          const int16_t* in, int lg, int16_t* out) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    const int n = 1 << lg, nn = n * n;
    for (int i = threadIdx.x; i < nn; i += 64) // transposed load: dT[x][i] = d[i][x]
        ((int16_t*)SH.r2)[(i & (n - 1)) * n + (i >> lg)] = in[(size_t)blockIdx.x * nn + i];
    WSYNC();
    inv_dct_lg(c, lg, 1);
    for (int i = threadIdx.x; i < nn; i += 64) out[(size_t)blockIdx.x * nn + i] = SH.r1[i];
}

__global__ __launch_bounds__(64) void test_quantize_kernel(const DevConst* __restrict__ k,
                                                           const int16_t* in, int lg, int16_t* out,
                                                           long long* cost, int* overflow) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    load_tables(c);
    const int nn = 1 << (2 * lg);
    for (int i = threadIdx.x; i < nn; i += 64) SH.r1[i] = in[(size_t)blockIdx.x * nn + i];
    WSYNC();
    int ovf = 0;
    bool any = false;
    const long long lc = quantize_solo(c, lg, 1, &ovf, &any);
    for (int i = threadIdx.x; i < nn; i += 64) out[(size_t)blockIdx.x * nn + i] = SH.r1[i];
    if (threadIdx.x == 0) {
        cost[blockIdx.x] = lc;
        if (ovf) atomicOr(overflow, 1);
    }
}

// DevConst::head_rng against the functions it stands for, over every 16-bit coefficient (grid: 1024 blocks of 64 lanes;
// block size and DC flag in blockIdx.y): out[0] += coefficients the range lets into the region that head_alpha ends it
// at (never allowed), out[1] += coefficients the range ends the region at that head_alpha would let in (allowed, not
// expected), out[2] += coefficients whose "quotient >= 2" differs from the range's, out[3] += alpha differences
__global__ __launch_bounds__(64) void test_head_ranges_kernel(const DevConst* __restrict__ kk, int* out) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)kk;
    load_tables(c);
    const CONST_AS DevConst* k = c.k;
    const int lg = 2 + (int)(blockIdx.y >> 1);
    const bool dcn = blockIdx.y & 1;
    const int tc = (int)(blockIdx.x * 64 + threadIdx.x) - 32768;
    const int sh = lg + 4, off = (1 << sh) >> 1;
    const HeadK hk = head_consts(sh, off, k->lsc);
    const HeadT ht = head_ranges(k, lg);
    const int qd = quotient(k, tc, sh, off);
    bool bad;
    const int alpha = head_alpha(tc, qd, dcn, hk, &bad);
    const bool rbad = head_bad(tc, dcn, ht);
    if (bad && !rbad) atomicAdd(out + 0, 1);
    if (!bad && rbad) atomicAdd(out + 1, 1);
    if ((qd >= 2) != head_sig(tc, ht)) atomicAdd(out + 2, 1);
    if (alpha != head_alpha1(tc, hk)) atomicAdd(out + 3, 1);
}

// DevConst::avail_tab + the picture's edges (block_avail_mask) against the formulas (block_avail_formula): one block per
// thread -- blockIdx.x = CTU, threadIdx.x = 4x4 position, blockIdx.y = size -- for luma and chroma spacing; *out +=
// differences
__global__ __launch_bounds__(64) void test_avail_tab_kernel(const DevConst* __restrict__ kk, int* out) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)kk;
    c.W = kk->W;
    c.ctu_x = 32 * (int)(blockIdx.x % (unsigned)kk->ctu_cols);
    c.ctu_y = 32 * (int)(blockIdx.x / (unsigned)kk->ctu_cols);
    const int lg = 2 + (int)blockIdx.y;
    const int bx = 4 * (int)(threadIdx.x & 7), by = 4 * (int)(threadIdx.x >> 3);
    if ((bx | by) & ((1 << lg) - 1)) return;
    for (int st = 1; st <= 2; ++st)
        if (block_avail_formula(c, bx, by, lg, st) != block_avail_mask(c, bx, by, lg, st)) atomicAdd(out, 1);
    // what cclm_params reads off the mask
    const int m = block_avail_mask(c, bx, by, lg, 1), tn = 1 << lg, gx = c.ctu_x + bx, gy = c.ctu_y + by;
    if (above_right_avail(c, bx, by, lg) != (((m >> 4) & 1) != 0)) atomicAdd(out, 1);
    if (below_left_avail(c, bx, by, lg) != ((m & 1) != 0)) atomicAdd(out, 1);
    if (nb_avail(c, gx, gy, tn, gx - 1, gy, false, false) != (((m >> 1) & 1) != 0)) atomicAdd(out, 1);
    if (nb_avail(c, gx, gy, tn, gx, gy - 1, false, false) != (((m >> 3) & 1) != 0)) atomicAdd(out, 1);
}

// quantize_p16 (the packed 4x4 leaf search's quantiser): each wave takes up to four consecutive 4x4 blocks
__global__ __launch_bounds__(64) void test_quantize_p16_kernel(const DevConst* __restrict__ k, const int16_t* in, int count,
                                                               int16_t* out, long long* cost, int* overflow) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    load_tables(c);
    const int first = 4 * blockIdx.x;
    const int nb = min(4, count - first);
    if (threadIdx.x < 16 * nb) SH.r1[threadIdx.x] = in[(size_t)first * 16 + threadIdx.x];
    WSYNC();
    int ovf = 0, any = 0;
    long long lvl[4];
    quantize_p16(c, nb, &ovf, lvl, &any);
    if (threadIdx.x < 16 * nb) out[(size_t)first * 16 + threadIdx.x] = SH.r1[threadIdx.x];
    if (threadIdx.x == 0) {
        for (int b = 0; b < nb; ++b) cost[first + b] = lvl[b];
        if (ovf) atomicOr(overflow, 1);
    }
}

// quantize_pk (the packed leaf searches' quantiser): one wave per pack of nc candidates
__global__ __launch_bounds__(64) void test_quantize_pk_kernel(const DevConst* __restrict__ k, const int16_t* in, int lgl, int nc,
                                                              int16_t* out, long long* cost, int* overflow) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    load_tables(c);
    const int total = nc * 3 * (1 << (2 * lgl)) / 2;
    for (int i = threadIdx.x; i < total; i += 64) SH.r1[i] = in[(size_t)blockIdx.x * total + i];
    WSYNC();
    int ovf = 0;
    bool any_y = false, any_c = false;
    long long ly[3], lc[3];
    if (lgl == 3)
        quantize_pk<3>(c, nc, &ovf, ly, lc, &any_y, &any_c);
    else
        quantize_pk<4>(c, nc, &ovf, ly, lc, &any_y, &any_c);
    for (int i = threadIdx.x; i < total; i += 64) out[(size_t)blockIdx.x * total + i] = SH.r1[i];
    if (threadIdx.x == 0) {
        for (int b = 0; b < nc; ++b) {
            cost[((size_t)blockIdx.x * nc + b) * 2] = ly[b];
            cost[((size_t)blockIdx.x * nc + b) * 2 + 1] = lc[b];
        }
        if (ovf) atomicOr(overflow, 1);
    }
}

__global__ __launch_bounds__(64) void test_dequantize_kernel(const DevConst* __restrict__ k,
                                                             const int16_t* in, int lg, int16_t* out) {
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    const int n = 1 << lg, nn = n * n;
    for (int i = threadIdx.x; i < nn; i += 64) SH.r1[i] = in[(size_t)blockIdx.x * nn + i];
    WSYNC();
    dequantize_t(c, lg, 1);
    for (int i = threadIdx.x; i < nn; i += 64) // undo the transpose
        out[(size_t)blockIdx.x * nn + i] = ((const int16_t*)SH.r2)[(i & (n - 1)) * n + (i >> lg)];
}

// Prediction of single blocks in the environment of given reconstruction planes (one wave per item):
// the CTU's reconstruction tile and border are loaded from the planes, then reference samples +
// prediction run exactly as in a full evaluation of the final pass (prediction bytes to scratch).
// item = {x, y (luma, picture), log2 luma size, comp (0 luma block, 1 Cb+Cr pair), mode, output offset}
// comp 3 + comps (comps: bit 0 luma, bit 1 the chroma pair): a SAD LIST (sad_list_angular) of the block against its own samples
// in the planes as "originals"; mode = m0 | entries << 8 | stride << 16, entry j = m0 + j * stride (kNoMode beyond 66); the
// output is the 16 accumulators of lanes 0..15 (u32 each); comp 7: the CCLM SAD list of the chroma pair (mode 0), same output
__global__ __launch_bounds__(64) void test_predict_kernel(const DevConst* __restrict__ k, const uint8_t* planes,
                                                          const int* items, uint8_t* scratch, uint8_t* out) {
    const int* it = items + 6 * blockIdx.x;
    const int x = it[0], y = it[1], tlg = it[2], comp = it[3], mode = it[4];
    Ctx c = {};
    c.k = (const CONST_AS DevConst*)k;
    c.org = (const GLOBAL_AS uint8_t*)planes; // residuals are computed against these and ignored
    c.W = k->W;
    c.WH = k->W * k->H;
    c.pred_scratch = scratch + (size_t)blockIdx.x * 1024;
    c.ctu_x = x & ~31;
    c.ctu_y = y & ~31;
    c.write = 1;
    load_tables(c);
    const int W = c.W, Wc = W >> 1;
    const GLOBAL_AS uint8_t* rec = (const GLOBAL_AS uint8_t*)planes;
    for (int i = LANE; i < 72; i += 64) {
        const int gx = c.ctu_x - 4 + i, gy = c.ctu_y - 1;
        SH.recYtop[i] = (gx >= 0 && gx < W && gy >= 0) ? rec[(size_t)gy * W + gx] : 0;
    }
    for (int i = LANE; i < 32 * 36; i += 64) {
        const int yy = i / 36, xx = i % 36 - 4;
        const int gx = c.ctu_x + xx, gy = c.ctu_y + yy;
        SH.recY[yy * 36 + xx + 4] = gx >= 0 ? rec[(size_t)gy * W + gx] : 0;
    }
    for (int pc = 1; pc < 3; ++pc) {
        for (int i = LANE; i < 40; i += 64) {
            const int gx = (c.ctu_x >> 1) - 4 + i, gy = (c.ctu_y >> 1) - 1;
            SH.recCtop[pc - 1][i] = (gx >= 0 && gx < Wc && gy >= 0) ? rec[plane_off(c, pc) + (size_t)gy * Wc + gx] : 0;
        }
        for (int i = LANE; i < 16 * 20; i += 64) {
            const int yy = i / 20, xx = i % 20 - 4;
            const int gx = (c.ctu_x >> 1) + xx, gy = (c.ctu_y >> 1) + yy;
            SH.recC[pc - 1][yy * 20 + xx + 4] = gx >= 0 ? rec[plane_off(c, pc) + (size_t)gy * Wc + gx] : 0;
        }
    }
    WSYNC();
    const int tx = x & 31, ty = y & 31;
    if (comp == 2) {
        // the packed predictor of the 4x4 leaf search: the item's mode in row 0, other modes in the rows next to it
        build_refs(c, 0, tx, ty, 2);
        const int row = LANE >> 4;
        const int v = predict4_lane(c, row == 0 ? mode : (mode + 1 + 22 * row) % 67);
        if (LANE < 16) out[it[5] + LANE] = (uint8_t)v;
        return;
    }
    if (comp == 7) { // the CCLM SAD list of the chroma pair (sad_list_cclm): lanes 0..2 = LT_CCLM, T_CCLM, L_CCLM
        uint8_t* oc = (uint8_t*)SH.r2 + org_byte(1, tlg);
        const int nc = 1 << (tlg - 1);
        for (int i = LANE; i < 2 * nc * nc; i += 64) {
            const int pl = i >= nc * nc ? 1 : 0, ii = i - pl * nc * nc;
            oc[i] = rec[plane_off(c, 1 + pl) + (size_t)((y >> 1) + ii / nc) * Wc + (x >> 1) + ii % nc];
        }
        WSYNC();
        const unsigned acc = sad_list_cclm(c, tx, ty, tlg);
        if (LANE < 16) ((uint32_t*)(out + it[5]))[LANE] = acc;
        return;
    }
    if (comp >= 4) {
        const int comps = comp - 3, m0 = mode & 255, nm = (mode >> 8) & 255, stride = mode >> 16;
        // the originals where a SAD list reads them (stage_org / stage_org_leaf's layout): luma, then Cb | Cr
        uint8_t* ol = (uint8_t*)SH.r2 + org_byte(0, tlg);
        uint8_t* oc = (uint8_t*)SH.r2 + org_byte(1, tlg);
        const int n = 1 << tlg, nc = n >> 1;
        for (int i = LANE; i < n * n; i += 64) ol[i] = rec[(size_t)(y + (i >> tlg)) * W + x + (i & (n - 1))];
        if (comps & 2)
            for (int i = LANE; i < 2 * nc * nc; i += 64) {
                const int pl = i >= nc * nc ? 1 : 0, ii = i - pl * nc * nc;
                oc[i] = rec[plane_off(c, 1 + pl) + (size_t)((y >> 1) + ii / nc) * Wc + (x >> 1) + ii % nc];
            }
        WSYNC();
        if (comps & 1) build_refs(c, 0, tx, ty, tlg);
        if (comps & 2) build_refs(c, 1, tx, ty, tlg);
        unsigned long long lo = 0, hi = 0;
        for (int j = 0; j < nm; ++j) {
            const int m = m0 + j * stride;
            const unsigned long long e = (unsigned long long)(m <= 66 ? m : kNoMode) << (8 * (j & 7));
            if (j < 8) lo |= e; else hi |= e;
        }
        const unsigned acc = sad_list_angular(c, comps, tx, ty, tlg, nm, lo, hi);
        if (LANE < 16) ((uint32_t*)(out + it[5]))[LANE] = acc;
        return;
    }
    if (mode < LT_CCLM) build_refs(c, comp, tx, ty, tlg);
    predict<true>(c, comp, tx, ty, tlg, mode, 0, false);
    const int n = 1 << (tlg - (comp ? 1 : 0));
    const int total = (comp ? 2 : 1) * n * n;
    __threadfence_block();
    for (int i = LANE; i < total; i += 64) out[it[5] + i] = c.pred_scratch[i];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

thread_local std::string g_create_error;

// An encode call splits its pictures over this many HIP streams.  Pictures are independent, so the
// streams' anti-diagonal launches overlap and one lane's tail (partially filled GPU) is filled by
// the other lanes' work.
#ifndef WRENC_ENCODE_LANES
#define WRENC_ENCODE_LANES 4
#endif
constexpr int kEncodeLanes = WRENC_ENCODE_LANES;


// AUTO picks the team schedule for an anti-diagonal while one wave per CTU would leave more than half of the
// GPU's wave slots empty (slots: CUs x kWorkgroupsPerCU x WPB = 5120 on an MI355X).  Measured (DESIGN.md section 5,
// gpurun_out/r3b): threshold at 25 / 50 / 75 / 100 % of the slots gives 282 / 306 / 270 / 269 frames/s at 1080p depth 2
// with 128 pictures, 54.4 / 55.9 / 53.8 / 48.9 at 3840x2176 depth 3 with 128; 50 % is best at every batch size tried.
// End of round 3 (level schedule at every depth, leaf searches a third faster): at max-split-depth 3, where all four
// members of a team have a tree level of their own, 65 % is better at every batch size between 30 and 240 pictures of
// 3840x2176 (60 pictures: 89.0 against 86.1 frames/s; 90: 102.6 / 100.9; 128: 115.8 / 114.5; 7680x4320, 32 pictures:
// 23.2 / 22.5; gpurun_out/s69, s70); at depth 2 it is 50 % still (128 pictures of 1080p: 459 against 454 at 65 %).
constexpr int kTeamBelowSlotsPct = 50, kTeamBelowSlotsPctDepth3 = 65;

// DCT-2 integer cosines c[j] ~ 64*sqrt(2)*cos(j*pi/128), H.266 8.7.4.5
// (the reference's 64-point matrix, transformer.rs:934-1191, is row k = c[(2n+1)k])
const int kCos[65] = {64, 91, 90, 90, 90, 90, 90, 90, 89, 88, 88, 87, 87, 86, 85, 84, 83, 83, 82, 81, 80, 79,
                      78, 77, 75, 73, 73, 71, 70, 69, 67, 65, 64, 62, 61, 59, 57, 56, 54, 52, 50, 48, 46, 44,
                      43, 41, 38, 37, 36, 33, 31, 28, 25, 24, 22, 20, 18, 15, 13, 11, 9,  7,  4,  2,  0};

int dct64(int k, int n) {
    int t = ((2 * n + 1) * k) % 256;
    if (t > 128) t = 256 - t;
    return t > 64 ? -kCos[128 - t] : kCos[t];
}

// H.266 Table 24 (common.rs:145) and Table 25 fC (common.rs:153)
const int16_t kIntraAngle[95] = {
    512, 341, 256, 171, 128, 102, 86,  73,  64,  57,  51,  45,  39,  35,  0,   0,   32,  29,  26,
    23,  20,  18,  16,  14,  12,  10,  8,   6,   4,   3,   2,   1,   0,   -1,  -2,  -3,  -4,  -6,
    -8,  -10, -12, -14, -16, -18, -20, -23, -26, -29, -32, -29, -26, -23, -20, -18, -16, -14, -12,
    -10, -8,  -6,  -4,  -3,  -2,  -1,  0,   1,   2,   3,   4,   6,   8,   10,  12,  14,  16,  18,
    20,  23,  26,  29,  32,  35,  39,  45,  51,  57,  64,  73,  86,  102, 128, 171, 256, 341, 512};
const int8_t kFC[32][4] = {
    {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2},
    {-4, 56, 14, -2}, {-4, 55, 15, -2}, {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3},
    {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4}, {-4, 36, 36, -4}, {-4, 33, 39, -4},
    {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},
    {0, 4, 62, -2},   {0, 2, 63, -1}};

void diag_scan(int lw, int lh, uint8_t (*out)[2]) { // ctu.rs:54-77
    const int bw = 1 << lw, bh = 1 << lh;
    int i = 0, x = 0, y = 0;
    bool stop = false;
    while (!stop) {
        while (y >= 0) {
            if (x < bw && y < bh) {
                out[i][0] = (uint8_t)x;
                out[i][1] = (uint8_t)y;
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

} // namespace

struct wrenc_gpu_ctx {
    wrenc_gpu_config cfg;
    long long wave_slots = 0;                  // waves of the search kernel the device holds at once
    long long device_wave_slots = 0;           // (wave_slots can be overridden by a test: wrenc_gpu_test_set_wave_slots)
    hipStream_t stream = nullptr;              // lane 0 of the encode; timing events
    hipStream_t copy_stream = nullptr;         // uploads and downloads: they overlap the search of other slots
    hipEvent_t ev_uploaded = nullptr;          // end of the uploads an encode call has to wait for
    bool uploads_pending = false;
    std::vector<hipEvent_t> enc_events;        // ring: one "this encode call is done" event per call in flight
    size_t enc_event_next = 0;
    std::vector<hipEvent_t> slot_event;        // per slot: the event of the encode call that last searched it
    std::vector<hipStream_t> lanes;            // extra encode lanes (pictures are independent)
    std::vector<hipEvent_t> lane_done;
    hipEvent_t ev_fork = nullptr;
    hipEvent_t last_done = nullptr;            // completion event of the most recent encode call
    DevConst* d_const = nullptr;
    PicBufs* d_slots = nullptr;
    std::vector<PicBufs> slots;
    std::vector<int> state; // 0 empty, 1 uploaded, 2 encoded
    unsigned long long* d_mismatch = nullptr;
    int* d_overflow = nullptr;
    uint8_t* d_pred_scratch = nullptr; // kScratchSlots x WPB x kWaveScratch: saved reconstructions (dev_search.h copy_block)
    unsigned long long* d_slot_map = nullptr; // kScratchSlots bits: scratch regions in use
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;
    int last_launches = 0;
    std::vector<char> launch_team;             // per launch of the last call: the team kernel?
    std::vector<long long> launch_ctus;        // ... and the CTU-pictures it searched
    bool stats_enabled = false; // per-launch timing events: bench / profiling only (wrenc_gpu_stats_enable)
    // compact read-back: device scratch for one call (masks, payloads with room for every block, counts), grown on demand
    uint32_t* d_cmask = nullptr;
    int16_t* d_cpayload = nullptr;
    unsigned* d_ccount = nullptr;
    int compact_cap = 0; // pictures the scratch holds
    // token read-back: the page pool of one call, its allocation counter, the CTUs' first pages
    uint32_t* d_tok_pool = nullptr;
    size_t tok_pool_words = 0;
    unsigned* d_tok_counter = nullptr;
    uint32_t* d_tok_first = nullptr;
    int tok_first_cap = 0;
    int schedule = WRENC_GPU_SCHEDULE_AUTO;
    int last_schedule = WRENC_GPU_SCHEDULE_WAVE; // what the most recent encode call ran
    bool stats_valid = false;
    std::string err;
    int ctu_cols = 0, ctu_rows = 0;
};

namespace {

int fail(wrenc_gpu_ctx* ctx, int code, const std::string& msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, WRENC_GPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

size_t plane_bytes(const wrenc_gpu_config& c, int comp, size_t elem) {
    const size_t w = comp ? c.width / 2 : c.width, h = comp ? c.height / 2 : c.height;
    return w * h * elem;
}

// DevConst::head_rng: the coefficients that do not end the head proof's region, and those with a quotient below 2, as one
// interval around zero each (dev_quant.h: head_alpha, quotient -- the same integer arithmetic here on the host, over
// every 16-bit coefficient; test_head_ranges_kernel holds the result against the device functions).  An interval that
// stops at the first coefficient that fails can only be too narrow, which costs a longer walk, never a wrong level.
void fill_head_ranges(DevConst& k) {
    const int ldq1 = (int32_t)k.ldq[1], ldq2 = (int32_t)k.ldq[2], ldq3 = (int32_t)k.ldq[3]; // (the 32-bit table of the kernels' LDS)
    for (int idx = 0; idx < 4; ++idx) {
        const int sh = idx + 2 + 4, off = (1 << sh) >> 1, lsc = k.lsc;
        const int dp[4] = {0, (lsc + off) >> sh, (2 * lsc + off) >> sh, (3 * lsc + off) >> sh};
        const int dn[4] = {0, (-lsc + off) >> sh, (-2 * lsc + off) >> sh, (-3 * lsc + off) >> sh};
        const auto quot = [&](int tc) {
            int S = (int)((unsigned)tc << sh) - off;
            if (tc < 0) S = -S;
            return tc == 0 ? 0 : (int)((uint32_t)(((uint64_t)(uint32_t)S * k.div_magic) >> 32) >> k.div_shift);
        };
        const auto bad = [&](int tc, bool dcn) {
            const int qd = quot(tc);
            const bool neg = tc < 0;
            const int d1 = abs(tc - (neg ? dn[1] : dp[1])), d2 = abs(tc - (neg ? dn[2] : dp[2])), d3 = abs(tc - (neg ? dn[3] : dp[3]));
            const int d1o = abs(tc - (neg ? dp[1] : dn[1]));
            const int c0tz = 128 * abs(tc);
            const int c0d0 = c0tz + ldq1;
            const int c1d0 = 128 * d2 + ldq2;
            const int c0d1 = dcn ? 128 * d1o + ldq1 : (qd ? 128 * d1 + ldq2 : c0d0);
            const int c1d1 = (qd && !dcn) ? 128 * d3 + ldq3 : 128 * d1 + ldq2;
            int alpha = c1d0 - c0tz;
            int beta = std::min(std::min(c0d0, c0d1), c1d1) - c0tz;
            if (tc == 0) {
                alpha = 1 << 28;
                beta = ldq1;
            }
            return qd >= 2 || alpha < 0 || beta < 0 || ldq1 < 0;
        };
        const auto interval = [&](int32_t* r, auto&& fails) { // r[0] = -lowest, r[1] = how many: the run of passing coefficients around 0
            r[0] = r[1] = 0;
            if (fails(0)) return;
            int tp = 0, tn = 0;
            while (tp < 32767 && !fails(tp + 1)) ++tp;
            while (tn < 32768 && !fails(-(tn + 1))) ++tn;
            r[0] = tn;
            r[1] = tp + tn + 1;
        };
        interval(&k.head_rng[idx][0], [&](int tc) { return bad(tc, false); });
        interval(&k.head_rng[idx][2], [&](int tc) { return bad(tc, true); });
        interval(&k.head_rng[idx][4], [&](int tc) { return quot(tc) >= 2; });
    }
}

// DevConst::avail_tab: the reference's availability rules (ctu.rs:2083-2188, encoder_context.rs:918-956 -- the same
// arithmetic as above_right_avail / below_left_avail / nb_avail of dev_common.h) evaluated on the host for a CTU in the
// middle of a 3 x 3-CTU picture, every block position and size.
void fill_avail_tab(DevConst& k) {
    const int W = 96, H = 96, ctu_x = 32, ctu_y = 32;
    const auto above_right = [&](int bx, int by, int lg) {
        for (;;) {
            const int n = 1 << lg;
            if (ctu_x + bx + n >= W) return false;
            if (lg == 5) return ctu_y > 0 && ctu_x + 32 < W;
            const int px = bx & ~(2 * n - 1), py = by & ~(2 * n - 1);
            if (bx == px && by == py) return ctu_y + by > 0;
            if (by == py) {
                bx = px;
                by = py;
                lg += 1;
                continue;
            }
            return bx == px;
        }
    };
    const auto below_left = [&](int bx, int by, int lg) {
        for (;;) {
            const int n = 1 << lg;
            if (ctu_y + by + n >= H) return false;
            if (lg == 5) return false;
            const int px = bx & ~(2 * n - 1), py = by & ~(2 * n - 1);
            if (px < bx) return false;
            if (by + n < py + 2 * n) return ctu_x + bx > 0;
            bx = px;
            by = py;
            lg += 1;
        }
    };
    for (int lg = 2; lg <= 5; ++lg)
        for (int by = 0; by < 32; by += 4)
            for (int bx = 0; bx < 32; bx += 4) {
                int avm = 0;
                if (!(bx & ((1 << lg) - 1)) && !(by & ((1 << lg) - 1))) {
                    const int tn = 1 << lg, gx = ctu_x + bx, gy = ctu_y + by;
                    const bool ar = above_right(bx, by, lg), bl = below_left(bx, by, lg);
                    const auto nb = [&](int xn, int yn) {
                        return xn >= 0 && yn >= 0 && xn < W && yn < H && ((xn >> 5) <= (gx >> 5) || (yn >> 5) < (gy >> 5)) &&
                               (yn >> 5) < (gy >> 5) + 1 && (xn < gx + tn || ar) && (yn < gy + tn || bl);
                    };
                    avm = (nb(gx - 1, gy + tn) ? 1 : 0) | (nb(gx - 1, gy) ? 2 : 0) | (nb(gx - 1, gy - 1) ? 4 : 0) | (nb(gx, gy - 1) ? 8 : 0) |
                          (nb(gx + tn, gy - 1) ? 16 : 0);
                }
                k.avail_tab[lg - 2][(by >> 2) * 8 + (bx >> 2)] = (uint8_t)avm;
            }
}

void fill_dev_const(const wrenc_gpu_config& cfg, DevConst& k) {
    memset(&k, 0, sizeof(k));
    k.W = cfg.width;
    k.H = cfg.height;
    k.qp = cfg.qp;
    k.max_depth = cfg.max_split_depth;
    k.ctu_cols = cfg.width / 32;
    k.ctu_rows = cfg.height / 32;
    static const int level_scale[6] = {40, 45, 51, 57, 64, 72}; // quantizer.rs:8
    k.lsc = (16 * level_scale[(cfg.qp + 1) % 6]) << ((cfg.qp + 1) / 6);
    {
        // n / lsc for n < 2^26 (|tc << sh| of a 16-bit coefficient) as mul_hi(n, m) >> s: with k = 26 + ceil(log2 lsc) and
        // m = floor(2^k / lsc) + 1 the error n (m lsc - 2^k) / (lsc 2^k) stays below 1 / lsc, and m < 2^28 fits 32 bits
        // (n itself does not depend on the QP: a 16-bit coefficient, the largest shift 8 + 5 - 5 + 1 = 9 of a 32x32 block, the rounding offset)
        static_assert((32768LL << 9) + (1 << 8) < (1LL << 26), "quotient(): |(tc << sh) - off| stays below 2^26");
        int lg = 0;
        while ((1 << lg) < k.lsc) ++lg;
        const int kk = 26 + lg;
        k.div_magic = (uint32_t)(((1ULL << kk) / (uint64_t)k.lsc) + 1);
        k.div_shift = (uint32_t)(kk - 32);
    }
    k.lambda_q = cfg.lambda_q;
    k.lambda_rd = cfg.lambda_rd;
    k.lambda_rd_chroma = cfg.lambda_rd_chroma;
    for (int i = 0; i < 1024; ++i) {
        k.ldq[i] = cfg.lambda_q * cfg.dq_table[i];
        k.lv[i] = cfg.lv_table[i];
    }
    memcpy(k.hb_luma, cfg.header_bits_luma, sizeof(k.hb_luma));
    memcpy(k.hb_chroma, cfg.header_bits_chroma, sizeof(k.hb_chroma));
    for (int idx = 0; idx < 4; ++idx) {
        const int n = 4 << idx, step = 64 / n;
        for (int u = 0; u < n; ++u)
            for (int x = 0; x < n; ++x) {
                k.dct[idx][u][x] = (int16_t)dct64(u * step, x);
                k.dct_t[idx][x][u] = (int16_t)dct64(u * step, x);
            }
    }
    diag_scan(2, 2, k.diag4);
    for (int idx = 0; idx < 4; ++idx) diag_scan(idx, idx, k.diag_sb[idx]);
    for (int idx = 0; idx < 4; ++idx) { // reverse-scan position -> raster index (ctu.rs:827-845: 4x4 sub-blocks)
        const int lg = idx + 2, n = 1 << lg, nsb = 1 << (2 * lg - 4);
        for (int p = 0; p < n * n; ++p) {
            const int sb = nsb - 1 - (p >> 4), sp = 15 - (p & 15);
            const int x = (k.diag_sb[idx][sb][0] << 2) + k.diag4[sp][0];
            const int y = (k.diag_sb[idx][sb][1] << 2) + k.diag4[sp][1];
            k.scan_idx[idx][p] = (uint16_t)(y * n + x);
        }
    }
    memcpy(k.intra_angle, kIntraAngle, sizeof(kIntraAngle));
    for (int mode = 0; mode < 67; ++mode) {
        // intraPredAngle and invAngle (intra_predictor.rs:1287-1310) of mode 2..66, one dword per mode
        const int angle = mode >= 2 ? kIntraAngle[14 + mode] : 0;
        int inv = 0;
        if (angle > 0)
            inv = (512 * 32 + angle / 2) / angle;
        else if (angle < 0)
            inv = -((512 * 32 + (-angle) / 2) / -angle);
        k.ang_tab[mode] = (int32_t)(((uint32_t)(uint16_t)(int16_t)angle) | ((uint32_t)(uint16_t)(int16_t)inv << 16));
    }
    memcpy(k.fc, kFC, sizeof(kFC));
    for (int u = 0; u < 32; ++u) // MFMA experiment: the 32-point basis as signed bytes
        for (int x = 0; x < 32; ++x) k.dct32_a[u][x] = (int8_t)dct64(u * 2, x);
    for (int v = 0; v < 32; ++v)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 16; ++j) k.dct32_p[v][h][j] = (int8_t)dct64(v * 2, 8 * (j / 4) + 4 * h + j % 4);
    int colsum[32];
    for (int n = 0; n < 32; ++n) {
        colsum[n] = 0;
        for (int i = 0; i < 32; ++i) colsum[n] += dct64(i * 2, n);
    }
    for (int y = 0; y < 32; ++y) {
        for (int i = 0; i < 32; ++i) k.idct32_b[y][i] = (int8_t)dct64(i * 2, y);
        k.idct32_k1[y] = 128 * colsum[y] + 64;
    }
    for (int x = 0; x < 32; ++x)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 16; ++j) k.idct32_p[x][h][j] = (int8_t)dct64((8 * (j / 4) + 4 * h + j % 4) * 2, x);
    for (int h = 0; h < 2; ++h)
        for (int w = 0; w < 16; ++w) k.idct32_k2[h][w] = 128 * colsum[8 * (w / 4) + 4 * h + w % 4] + 2048;
    fill_head_ranges(k);
    fill_avail_tab(k);
}

} // namespace

namespace {
template <class Launch>
int run_block_test(wrenc_gpu_ctx* ctx, const int16_t* in, int log2n, int count, int16_t* out, Launch launch) {
    if (!ctx || !in || !out) return WRENC_GPU_EINVAL;
    if (log2n < 2 || log2n > 5 || count < 1) return fail(ctx, WRENC_GPU_EINVAL, "log2n must be 2..5 and count >= 1");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t bytes = (size_t)count * sizeof(int16_t) << (2 * log2n);
    int16_t *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_in, bytes));
    hipError_t e = hipMalloc((void**)&d_out, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        launch(d_in, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, WRENC_GPU_EHIP, hipGetErrorString(e));
    return WRENC_GPU_OK;
}
} // namespace

extern "C" {

// The RD-model constants the reference reads from --extra-params, with its defaults for the live
// (dependent quantisation + trellis) variants: block_splitter.rs:29-44,187-375,594-693,775 and
// quantizer.rs:16-19,650-683.  Types follow the reference's parse::<f64> / <f32> / <i64>.
struct RdParams {
    double lv_pow = 0.48592678233563835, lv_offset = 0.15150746310196822;               // lv_dq_trellis_table
    double quant_lv_pow = 0.5004010166085378;                                            // dq_table
    double quant_qp_div = 5.218413785332902, quant_lambda_mul = 1.2709404305806742;      // lambda of the trellis
    long long quant_lambda_offset = 11;
    float qp_div = 4.4043665f, lambda_mul = 1.1282581f;
    float non_planar_offset = 2.2153597f, mpm_idx_offset = 1.3660221f, mpm_remainder_mult = 0.5007182f,
          mpm_remainder_offset = 2.2973304f, planar_offset = 0.9626864f, header_bits = 1.1772872f,
          chroma_header_bits = 1.309252f, cclm_pow = 0.4587651f, mpm_idx_pow = 0.40271285f,
          mpm_remainder_pow = 0.34385094f, cclm_mode_idx_offset = 2.1f, non_cclm_offset = 0.89f, cclm_offset = 0.53f;
    bool has_a = false; // "a" replaces lambda_mul in the chroma cost function (block_splitter.rs:775-778)
    float a = 0.0f;
};

static void resolve_config(wrenc_gpu_config* cfg, const RdParams& p) {
    const int qp = cfg->qp;
    // block_splitter.rs:29-53 (lv_dq_trellis), quantizer.rs:16-25
    for (int i = 0; i < 1024; ++i) {
        cfg->lv_table[i] = (int64_t)(std::pow((double)i + p.lv_offset, p.lv_pow) * 16384.0);
        cfg->dq_table[i] = (int64_t)std::pow((double)(i * 16384), p.quant_lv_pow);
    }
    // quantizer.rs:650-683
    cfg->lambda_q = (int64_t)(std::pow(2.0, (double)qp / p.quant_qp_div) * p.quant_lambda_mul) + p.quant_lambda_offset;
    // block_splitter.rs:289-309,472
    cfg->lambda_rd = std::pow(2.0f, (float)qp / p.qp_div) * p.lambda_mul;
    cfg->lambda_rd_chroma = std::pow(2.0f, (float)qp / p.qp_div) * (p.has_a ? p.a : p.lambda_mul); // :775-778
    // block_splitter.rs:187-406 (dep-quant + trellis variants)
    for (int tree = 0; tree < 2; ++tree)
        for (int cc = 0; cc < 4; ++cc)
            for (int cls = 0; cls < 67; ++cls) {
                float cclm_bits;
                if (cc > 0)
                    cclm_bits = p.cclm_offset + std::pow((float)(cc - 1) + p.cclm_mode_idx_offset, p.cclm_pow);
                else if (tree == 1)
                    cclm_bits = 0.0f;
                else
                    cclm_bits = p.non_cclm_offset;
                float mode_bits;
                if (cls == 0) {
                    mode_bits = p.planar_offset;
                } else {
                    float t;
                    if (cls <= 5)
                        t = std::pow((float)(cls - 1) + p.mpm_idx_offset, p.mpm_idx_pow);
                    else
                        t = p.mpm_remainder_mult * std::pow((float)(cls - 6) + p.mpm_remainder_offset, p.mpm_remainder_pow);
                    mode_bits = p.non_planar_offset + t;
                }
                mode_bits = mode_bits + cclm_bits;
                const float hb = tree == 0 ? p.header_bits + mode_bits : p.header_bits / 3.0f + mode_bits;
                cfg->header_bits_luma[tree][cc][cls] = (int64_t)(hb * 16384.0f);
            }
    for (int cc = 0; cc < 4; ++cc) {
        const float mode_bits =
            cc > 0 ? p.cclm_offset + std::pow((float)(cc - 1) + p.cclm_mode_idx_offset, p.cclm_pow) : p.non_cclm_offset;
        cfg->header_bits_chroma[cc] = (int64_t)((p.chroma_header_bits + mode_bits) * 16384.0f);
    }
}

int wrenc_gpu_default_config(wrenc_gpu_config* cfg, int width, int height, int qp, int max_split_depth) {
    if (!cfg) return WRENC_GPU_EINVAL;
    memset(cfg, 0, sizeof(*cfg));
    cfg->width = width;
    cfg->height = height;
    cfg->qp = qp;
    cfg->max_split_depth = max_split_depth;
    cfg->device = 0;
    cfg->n_slots = 1;
    resolve_config(cfg, RdParams());
    return WRENC_GPU_OK;
}

int wrenc_gpu_config_extra_params(wrenc_gpu_config* cfg, const char* extra_params) {
    if (!cfg) return WRENC_GPU_EINVAL;
    RdParams p;
    const struct { const char* key; double* f64; float* f32; long long* i64; } live[] = {
        {"lv_pow_dq_trellis", &p.lv_pow, nullptr, nullptr},
        {"lv_offset_dq_trellis", &p.lv_offset, nullptr, nullptr},
        {"quant_lv_pow", &p.quant_lv_pow, nullptr, nullptr},
        {"quant_qp_div_trellis", &p.quant_qp_div, nullptr, nullptr},
        {"quant_lambda_mul_trellis", &p.quant_lambda_mul, nullptr, nullptr},
        {"quant_lambda_offset_trellis", nullptr, nullptr, &p.quant_lambda_offset},
        {"qp_div_dq_trellis", nullptr, &p.qp_div, nullptr},
        {"lambda_mul_dq_trellis", nullptr, &p.lambda_mul, nullptr},
        {"non_planar_offset_dq_trellis", nullptr, &p.non_planar_offset, nullptr},
        {"mpm_idx_offset_dq_trellis", nullptr, &p.mpm_idx_offset, nullptr},
        {"mpm_remainder_mult_dq_trellis", nullptr, &p.mpm_remainder_mult, nullptr},
        {"mpm_remainder_offset_dq_trellis", nullptr, &p.mpm_remainder_offset, nullptr},
        {"planer_offset_dq_trellis", nullptr, &p.planar_offset, nullptr}, // sic
        {"header_bits_dq_trellis", nullptr, &p.header_bits, nullptr},
        {"chroma_header_bits_dq_trellis", nullptr, &p.chroma_header_bits, nullptr},
        {"cclm_pow", nullptr, &p.cclm_pow, nullptr},
        {"mpm_idx_pow", nullptr, &p.mpm_idx_pow, nullptr},
        {"mpm_remainder_pow", nullptr, &p.mpm_remainder_pow, nullptr},
        {"cclm_mode_idx_offset_dq_trellis", nullptr, &p.cclm_mode_idx_offset, nullptr},
        {"non_cclm_offset_dq_trellis", nullptr, &p.non_cclm_offset, nullptr},
        {"cclm_offset_dq_trellis", nullptr, &p.cclm_offset, nullptr},
        {"a", nullptr, &p.a, nullptr},
    };
    const std::string text = extra_params ? extra_params : "";
    size_t pos = 0;
    while (!text.empty() && pos <= text.size()) {
        size_t end = text.find(',', pos);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq == std::string::npos || item.find('=', eq + 1) != std::string::npos)
            return fail(nullptr, WRENC_GPU_EINVAL, "Invalid extra-params: " + text); // main.rs:205-215
        const std::string key = item.substr(0, eq), val = item.substr(eq + 1);
        for (const auto& l : live) {
            if (key != l.key) continue;
            char* rest = nullptr;
            if (l.f64) *l.f64 = strtod(val.c_str(), &rest);
            if (l.f32) *l.f32 = strtof(val.c_str(), &rest);
            if (l.i64) *l.i64 = strtoll(val.c_str(), &rest, 10);
            if (val.empty() || (rest && *rest)) // the reference's parse().unwrap() panics here
                return fail(nullptr, WRENC_GPU_EINVAL, "extra-params: " + key + " has no numeric value: " + val);
            if (key == "a") p.has_a = true;
        }
        // any other key is stored and never read by the live code path, as in the reference
        pos = end + 1;
    }
    resolve_config(cfg, p);
    return WRENC_GPU_OK;
}

const char* wrenc_gpu_last_error(const wrenc_gpu_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void wrenc_gpu_destroy(wrenc_gpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t st : ctx->lanes) (void)hipStreamSynchronize(st);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (PicBufs& b : ctx->slots) {
        // planes 1 and 2 point into plane 0's slab
        if (b.org[0]) (void)hipFree((void*)b.org[0]);
        if (b.org_t) (void)hipFree((void*)b.org_t);
        if (b.border) (void)hipFree(b.border);
        if (b.rec[0]) (void)hipFree(b.rec[0]);
        if (b.lev[0]) (void)hipFree(b.lev[0]);
        if (b.lev_dirty) (void)hipFree(b.lev_dirty);
        if (b.cu_log2) (void)hipFree(b.cu_log2);
        if (b.luma_mode) (void)hipFree(b.luma_mode);
        if (b.chroma_mode) (void)hipFree(b.chroma_mode);
        if (b.ctu_cost) (void)hipFree(b.ctu_cost);
    }
    if (ctx->d_const) (void)hipFree(ctx->d_const);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->d_mismatch) (void)hipFree(ctx->d_mismatch);
    if (ctx->d_overflow) (void)hipFree(ctx->d_overflow);
    if (ctx->d_tok_pool) (void)hipFree(ctx->d_tok_pool);
    if (ctx->d_tok_counter) (void)hipFree(ctx->d_tok_counter);
    if (ctx->d_tok_first) (void)hipFree(ctx->d_tok_first);
    if (ctx->d_pred_scratch) (void)hipFree(ctx->d_pred_scratch);
    if (ctx->d_slot_map) (void)hipFree(ctx->d_slot_map);
    if (ctx->d_cmask) (void)hipFree(ctx->d_cmask);
    if (ctx->d_cpayload) (void)hipFree(ctx->d_cpayload);
    if (ctx->d_ccount) (void)hipFree(ctx->d_ccount);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (hipStream_t st : ctx->lanes) (void)hipStreamDestroy(st);
    for (hipEvent_t e : ctx->lane_done) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_uploaded) (void)hipEventDestroy(ctx->ev_uploaded);
    for (hipEvent_t e : ctx->enc_events) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int wrenc_gpu_create(const wrenc_gpu_config* cfg, wrenc_gpu_ctx** out) {
    if (!cfg || !out) return fail(nullptr, WRENC_GPU_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 31) || (cfg->height & 31))
        return fail(nullptr, WRENC_GPU_EINVAL, "width and height must be positive multiples of 32");
    if (cfg->qp < 0 || cfg->qp > 63) return fail(nullptr, WRENC_GPU_EINVAL, "qp out of range 0..63");
    if (cfg->max_split_depth < 0 || cfg->max_split_depth > 3)
        return fail(nullptr, WRENC_GPU_EINVAL, "max_split_depth out of range 0..3");
    if (cfg->n_slots < 1) return fail(nullptr, WRENC_GPU_EINVAL, "n_slots must be >= 1");
    // The trellis keeps path costs in 32 bits (dev_quant.h, above kNoBranch): exact as long as one step costs less than 2^25,
    // i.e. 128 * 65535 + lambda_q * dq_table[bits] < 2^25 for every table entry -- true for the reference's defaults at every
    // QP (22.6 M at QP 63) and for rate models anywhere near them, not for e.g. quant_lv_pow = 2.5 or quant_qp_div_trellis =
    // 1.5 (--extra-params), where the reference's own i64 products reach 10^11 .. 10^19.  Refused rather than searched
    // with other results than the reference's.
    {
        constexpr long long kStepRoom = (1LL << 25) - 128LL * 65535LL;
        bool fits = cfg->lambda_q >= 0 && cfg->lambda_q < kStepRoom;
        for (int i = 0; fits && i < 1024; ++i)
            fits = cfg->dq_table[i] >= 0 && cfg->dq_table[i] < kStepRoom && cfg->lambda_q * cfg->dq_table[i] < kStepRoom;
        if (!fits)
            return fail(nullptr, WRENC_GPU_EINVAL,
                        "the quantiser's rate model (lambda_q x dq_table) is outside the range the device's 32-bit trellis costs cover");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, WRENC_GPU_ENODEV, "no HIP device available");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, WRENC_GPU_ENODEV, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, WRENC_GPU_ENODEV, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, WRENC_GPU_ENODEV, std::string("built for gfx950, device is ") + prop.gcnArchName);
    wrenc_gpu_ctx* ctx = new (std::nothrow) wrenc_gpu_ctx();
    if (!ctx) return fail(nullptr, WRENC_GPU_ENOMEM, "out of host memory");
    ctx->cfg = *cfg;
    ctx->wave_slots = ctx->device_wave_slots = (long long)prop.multiProcessorCount * kWorkgroupsPerCU * WPB;
    ctx->ctu_cols = cfg->width / 32;
    ctx->ctu_rows = cfg->height / 32;
    auto bail = [&](int code, const std::string& msg) {
        g_create_error = msg;
        wrenc_gpu_destroy(ctx);
        return code;
    };
#define CREATE_TRY(expr)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return bail(e_ == hipErrorOutOfMemory ? WRENC_GPU_ENOMEM : WRENC_GPU_EHIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)
    CREATE_TRY(hipSetDevice(cfg->device));
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_uploaded, hipEventDisableTiming));
    for (int i = 0; i < 8; ++i) {
        hipEvent_t e = nullptr;
        CREATE_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->enc_events.push_back(e);
    }
    ctx->slot_event.assign((size_t)cfg->n_slots, nullptr);
    CREATE_TRY(hipEventCreate(&ctx->ev_fork));
    for (int i = 1; i < kEncodeLanes; ++i) {
        hipStream_t st = nullptr;
        CREATE_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->lanes.push_back(st);
        hipEvent_t e = nullptr;
        CREATE_TRY(hipEventCreate(&e));
        ctx->lane_done.push_back(e);
    }
    // Created after the encode lanes on purpose: the runtime deals streams onto its hardware queues in
    // creation order (4 per process unless GPU_MAX_HW_QUEUES says otherwise), and two lanes on one queue
    // would run one after the other.  With 4 queues the copy stream shares lane 0's.
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&ctx->ev_begin));
    CREATE_TRY(hipEventCreate(&ctx->ev_end));
    {
        DevConst* hk = new (std::nothrow) DevConst();
        if (!hk) return bail(WRENC_GPU_ENOMEM, "out of host memory");
        fill_dev_const(*cfg, *hk);
        hipError_t e = hipMalloc((void**)&ctx->d_const, sizeof(DevConst));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_const, hk, sizeof(DevConst), hipMemcpyHostToDevice);
        delete hk;
        CREATE_TRY(e);
    }
    CREATE_TRY(hipMalloc((void**)&ctx->d_mismatch, sizeof(unsigned long long)));
    CREATE_TRY(hipMemset(ctx->d_mismatch, 0, sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_overflow, 2 * sizeof(int))); // [0]: encode calls (sticky), [1]: the building-block test entries
    CREATE_TRY(hipMemset(ctx->d_overflow, 0, 2 * sizeof(int)));
    ctx->slots.assign(cfg->n_slots, PicBufs{});
    ctx->state.assign(cfg->n_slots, 0);
    for (int s = 0; s < cfg->n_slots; ++s) {
        PicBufs& b = ctx->slots[s];
        // Y | Cb | Cr of a picture are one slab each for originals, reconstruction and levels: the
        // kernel addresses a plane by an element offset, never by selecting a pointer per lane
        const size_t wh = (size_t)cfg->width * cfg->height;
        uint8_t* org_slab = nullptr;
        CREATE_TRY(hipMalloc((void**)&org_slab, wh + wh / 2));
        b.org[0] = org_slab;
        b.org[1] = org_slab + wh;
        b.org[2] = org_slab + wh + wh / 4;
        const size_t n_ctus = (size_t)ctx->ctu_cols * ctx->ctu_rows;
        CREATE_TRY(hipMalloc((void**)&b.org_t, n_ctus * kOrgTile));
        CREATE_TRY(hipMalloc((void**)&b.border, n_ctus * kBorderBytes));
        CREATE_TRY(hipMalloc((void**)&b.rec[0], wh + wh / 2));
        b.rec[1] = b.rec[0] + wh;
        b.rec[2] = b.rec[0] + wh + wh / 4;
        CREATE_TRY(hipMalloc((void**)&b.lev[0], (wh + wh / 2) * sizeof(int16_t)));
        b.lev[1] = b.lev[0] + wh;
        b.lev[2] = b.lev[0] + wh + wh / 4;
        // the planes start zeroed, with no block marked as holding levels (PicBufs::lev_dirty)
        CREATE_TRY(hipMemsetAsync(b.lev[0], 0, (wh + wh / 2) * sizeof(int16_t), ctx->stream));
        CREATE_TRY(hipMalloc((void**)&b.lev_dirty, n_ctus * 4 * sizeof(uint32_t)));
        CREATE_TRY(hipMemsetAsync(b.lev_dirty, 0, n_ctus * 4 * sizeof(uint32_t), ctx->stream));
        CREATE_TRY(hipMalloc((void**)&b.cu_log2, (size_t)(cfg->width / 4) * (cfg->height / 4)));
        CREATE_TRY(hipMalloc((void**)&b.luma_mode, (size_t)(cfg->width / 4) * (cfg->height / 4)));
        CREATE_TRY(hipMalloc((void**)&b.chroma_mode, (size_t)(cfg->width / 8) * (cfg->height / 8)));
        CREATE_TRY(hipMalloc((void**)&b.ctu_cost, sizeof(float) * ctx->ctu_cols * ctx->ctu_rows));
    }
    CREATE_TRY(hipMalloc((void**)&ctx->d_slots, sizeof(PicBufs) * cfg->n_slots));
    CREATE_TRY(hipMemcpy(ctx->d_slots, ctx->slots.data(), sizeof(PicBufs) * cfg->n_slots, hipMemcpyHostToDevice));
    CREATE_TRY(hipStreamSynchronize(ctx->stream)); // the level planes are zero before anything can reach them
#undef CREATE_TRY
    *out = ctx;
    return WRENC_GPU_OK;
}

int wrenc_gpu_upload(wrenc_gpu_ctx* ctx, int slot, const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                     size_t stride_y, size_t stride_c) {
    if (!ctx) return WRENC_GPU_EINVAL;
    if (slot < 0 || slot >= ctx->cfg.n_slots || !y || !cb || !cr) return fail(ctx, WRENC_GPU_EINVAL, "bad slot or null plane");
    const size_t w = ctx->cfg.width, h = ctx->cfg.height;
    if (stride_y < w || stride_c < w / 2) return fail(ctx, WRENC_GPU_EINVAL, "stride smaller than row");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    PicBufs& b = ctx->slots[slot];
    // the planes may still be read by a search in flight on this slot
    if (ctx->slot_event[slot]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->slot_event[slot], 0));
    hipStream_t cs = ctx->copy_stream;
    HIP_TRY(ctx, hipMemcpy2DAsync((void*)b.org[0], w, y, stride_y, w, h, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipMemcpy2DAsync((void*)b.org[1], w / 2, cb, stride_c, w / 2, h / 2, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipMemcpy2DAsync((void*)b.org[2], w / 2, cr, stride_c, w / 2, h / 2, hipMemcpyHostToDevice, cs));
    // planar staging -> CTU tiles (what the search reads), behind the copies on the same stream
    hipLaunchKernelGGL(retile_kernel, dim3(ctx->ctu_cols * ctx->ctu_rows), dim3(64), 0, cs, b.org[0], (uint8_t*)b.org_t, (int)w,
                       (int)h, ctx->ctu_cols);
    HIP_TRY(ctx, hipGetLastError());
    ctx->uploads_pending = true;
    ctx->state[slot] = 1;
    return WRENC_GPU_OK;
}

int wrenc_gpu_encode(wrenc_gpu_ctx* ctx, int first_slot, int n_pictures) {
    if (!ctx) return WRENC_GPU_EINVAL;
    if (first_slot < 0 || n_pictures < 1 || first_slot + n_pictures > ctx->cfg.n_slots)
        return fail(ctx, WRENC_GPU_EINVAL, "slot range out of bounds");
    for (int s = first_slot; s < first_slot + n_pictures; ++s)
        if (ctx->state[s] == 0) return fail(ctx, WRENC_GPU_ESTATE, "slot has no uploaded picture");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const int cols = ctx->ctu_cols, rows = ctx->ctu_rows;
    const int ndiag = cols + 2 * (rows - 1);
    // Which schedule, decided PER ANTI-DIAGONAL: one wave per CTU fills the GPU only with thousands of CTUs
    // runnable side by side (pictures x CTUs of the diagonal); below that a team of kTeam waves per CTU
    // shortens the CTU's chain of dependent evaluations instead (kTeamBelowSlotsPct above).  The thin first and last diagonals of a big batch run as teams, its wide ones as waves.
    // Pictures are dealt to lanes in units of WPB (the wave schedule's workgroup) whatever the schedule, so a
    // picture stays on one stream; a team launch covers its lane's pictures in groups of WPB / kTeam.
    const int per_group = WPB;
    constexpr int kTeamsPerGroup = WPB >= kTeam ? WPB / kTeam : 1;
    const int total_groups = (n_pictures + per_group - 1) / per_group;
    int n_team_diags = 0, n_wave_diags = 0;
    const int n_lanes = total_groups < kEncodeLanes ? total_groups : kEncodeLanes;
    if (!ctx->d_pred_scratch) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_pred_scratch, (size_t)kScratchSlots * WPB * kWaveScratch));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_slot_map, kSlotMapWords * 8));
        HIP_TRY(ctx, hipMemset(ctx->d_slot_map, 0, kSlotMapWords * 8));
    }
    const bool timed = ctx->stats_enabled;
    if (timed) {
        // the timing events are re-recorded by every call: the previous call's must have been reached first
        if (ctx->stats_valid) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
        while ((int)ctx->ev_pool.size() < 2 * ndiag * n_lanes) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev_pool.push_back(e);
        }
    }
    // The scratch-region bitmap is only ever changed by running workgroups (acquire at start, release at
    // end).  A kernel that was aborted would leave its bits set for good, so the map is cleared whenever no
    // encode call is in flight (the last call's completion event has been reached).
    if (ctx->last_done == nullptr || hipEventQuery(ctx->last_done) == hipSuccess)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_slot_map, 0, kScratchSlots / 8, ctx->stream)); // (not the overflow counter)
    if (ctx->uploads_pending) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_uploaded, ctx->copy_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_uploaded, 0));
        ctx->uploads_pending = false;
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int l = 1; l < n_lanes; ++l) HIP_TRY(ctx, hipStreamWaitEvent(ctx->lanes[l - 1], ctx->ev_fork, 0));
    int launches = 0;
    const bool d3 = ctx->cfg.max_split_depth == 3; // the kernels built with / without the 4x4 leaves of split 8x8 CUs
    ctx->launch_team.clear();
    ctx->launch_ctus.clear();
    for (int d = 0; d < ndiag; ++d) {
        // rows r with 0 <= d - 2r < cols
        int r_min = d - (cols - 1);
        r_min = r_min <= 0 ? 0 : (r_min + 1) / 2;
        int r_max = d / 2;
        if (r_max > rows - 1) r_max = rows - 1;
        const int count = r_max - r_min + 1;
        if (count <= 0) continue;
        const bool team = ctx->schedule == WRENC_GPU_SCHEDULE_TEAM ||
                          (ctx->schedule == WRENC_GPU_SCHEDULE_AUTO &&
                           (long long)n_pictures * count * 100 <= ctx->wave_slots * (d3 ? kTeamBelowSlotsPctDepth3 : kTeamBelowSlotsPct));
        ++(team ? n_team_diags : n_wave_diags);
        for (int l = 0; l < n_lanes; ++l) {
            const int g0 = (int)((long long)total_groups * l / n_lanes), g1 = (int)((long long)total_groups * (l + 1) / n_lanes);
            const int lane_first = first_slot + g0 * per_group;
            int lane_pics = (g1 - g0) * per_group;
            if (g0 * per_group + lane_pics > n_pictures) lane_pics = n_pictures - g0 * per_group;
            if (lane_pics <= 0) continue;
            hipStream_t st = l == 0 ? ctx->stream : ctx->lanes[l - 1];
            if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[2 * launches], st));
            const dim3 grid(team ? count * ((lane_pics + kTeamsPerGroup - 1) / kTeamsPerGroup) : count * (g1 - g0));
#define WRENC_LAUNCH(KERNEL)                                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL), grid, dim3(64 * WPB), 0, st, ctx->d_const, ctx->d_slots, lane_first, lane_pics, d, \
                       r_min, count, ctx->d_pred_scratch, ctx->d_slot_map, ctx->d_mismatch, ctx->d_overflow)
            if (team) {
                if (d3)
                    WRENC_LAUNCH(ctu_search_team_kernel<true>);
                else
                    WRENC_LAUNCH(ctu_search_team_kernel<false>);
            } else {
                if (d3)
                    WRENC_LAUNCH(ctu_search_kernel<true>);
                else
                    WRENC_LAUNCH(ctu_search_kernel<false>);
            }
#undef WRENC_LAUNCH
            HIP_TRY(ctx, hipGetLastError());
            if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[2 * launches + 1], st));
            ctx->launch_team.push_back(team ? 1 : 0);
            ctx->launch_ctus.push_back((long long)count * lane_pics);
            ++launches;
        }
    }
    for (int l = 1; l < n_lanes; ++l) {
        HIP_TRY(ctx, hipEventRecord(ctx->lane_done[l - 1], ctx->lanes[l - 1]));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_done[l - 1], 0));
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    // downloads of these slots wait for this call only, not for searches queued after it
    hipEvent_t done = ctx->enc_events[ctx->enc_event_next++ % ctx->enc_events.size()];
    HIP_TRY(ctx, hipEventRecord(done, ctx->stream));
    ctx->last_done = done;
    ctx->last_schedule = n_wave_diags == 0 ? WRENC_GPU_SCHEDULE_TEAM : (n_team_diags == 0 ? WRENC_GPU_SCHEDULE_WAVE : WRENC_GPU_SCHEDULE_AUTO);
    ctx->last_launches = launches;
    ctx->stats_valid = timed;
    for (int s = first_slot; s < first_slot + n_pictures; ++s) {
        ctx->state[s] = 2;
        ctx->slot_event[s] = done;
    }
    return WRENC_GPU_OK;
}

int wrenc_gpu_sync(wrenc_gpu_ctx* ctx) {
    if (!ctx) return WRENC_GPU_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t st : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(st));
    int ovf = 0;
    HIP_TRY(ctx, hipMemcpy(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost));
    if (ovf & 2) return fail(ctx, WRENC_GPU_EHIP, "internal: a team member never reached a meeting point of the level schedule");
    if (ovf) return fail(ctx, WRENC_GPU_ELEVEL, "a quantised level reached 1024 (reference panics: block_splitter.rs:453)");
    return WRENC_GPU_OK;
}

int wrenc_gpu_download(wrenc_gpu_ctx* ctx, int slot, wrenc_gpu_picture* out) {
    if (!ctx || !out) return WRENC_GPU_EINVAL;
    if (slot < 0 || slot >= ctx->cfg.n_slots) return fail(ctx, WRENC_GPU_EINVAL, "bad slot");
    if (ctx->state[slot] != 2) return fail(ctx, WRENC_GPU_ESTATE, "slot has not been encoded");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const PicBufs& b = ctx->slots[slot];
    const wrenc_gpu_config& c = ctx->cfg;
    uint8_t* rec[3] = {out->rec_y, out->rec_cb, out->rec_cr};
    int16_t* lev[3] = {out->lev_y, out->lev_cb, out->lev_cr};
    // on the copy stream, behind the encode call that searched this slot and nothing later
    hipStream_t cs = ctx->copy_stream;
    if (ctx->slot_event[slot]) HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->slot_event[slot], 0));
    for (int k = 0; k < 3; ++k) {
        if (rec[k]) HIP_TRY(ctx, hipMemcpyAsync(rec[k], b.rec[k], plane_bytes(c, k, 1), hipMemcpyDeviceToHost, cs));
        if (lev[k]) HIP_TRY(ctx, hipMemcpyAsync(lev[k], b.lev[k], plane_bytes(c, k, 2), hipMemcpyDeviceToHost, cs));
    }
    const size_t n4 = (size_t)(c.width / 4) * (c.height / 4), n8 = (size_t)(c.width / 8) * (c.height / 8);
    if (out->cu_log2_size) HIP_TRY(ctx, hipMemcpyAsync(out->cu_log2_size, b.cu_log2, n4, hipMemcpyDeviceToHost, cs));
    if (out->luma_mode) HIP_TRY(ctx, hipMemcpyAsync(out->luma_mode, b.luma_mode, n4, hipMemcpyDeviceToHost, cs));
    if (out->chroma_mode) HIP_TRY(ctx, hipMemcpyAsync(out->chroma_mode, b.chroma_mode, n8, hipMemcpyDeviceToHost, cs));
    if (out->ctu_cost)
        HIP_TRY(ctx, hipMemcpyAsync(out->ctu_cost, b.ctu_cost, sizeof(float) * ctx->ctu_cols * ctx->ctu_rows,
                                    hipMemcpyDeviceToHost, cs));
    int ovf = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipStreamSynchronize(cs));
    if (ovf & 2) return fail(ctx, WRENC_GPU_EHIP, "internal: a team member never reached a meeting point of the level schedule");
    if (ovf) return fail(ctx, WRENC_GPU_ELEVEL, "a quantised level reached 1024 (reference panics: block_splitter.rs:453)");
    return WRENC_GPU_OK;
}

size_t wrenc_gpu_compact_mask_words(int width, int height) {
    const size_t blocks = (size_t)(width / 4) * (height / 4) * 3 / 2;
    return (blocks + 31) / 32;
}

int wrenc_gpu_download_compact(wrenc_gpu_ctx* ctx, int first_slot, int n, wrenc_gpu_compact* out) {
    if (!ctx || !out) return WRENC_GPU_EINVAL;
    if (first_slot < 0 || n < 1 || first_slot + n > ctx->cfg.n_slots) return fail(ctx, WRENC_GPU_EINVAL, "slot range out of bounds");
    for (int s = first_slot; s < first_slot + n; ++s)
        if (ctx->state[s] != 2) return fail(ctx, WRENC_GPU_ESTATE, "slot has not been encoded");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const wrenc_gpu_config& c = ctx->cfg;
    const size_t mask_words = wrenc_gpu_compact_mask_words(c.width, c.height);
    const size_t blocks = (size_t)(c.width / 4) * (c.height / 4) * 3 / 2;
    if (ctx->compact_cap < n) {
        if (ctx->d_cmask) (void)hipFree(ctx->d_cmask);
        if (ctx->d_cpayload) (void)hipFree(ctx->d_cpayload);
        if (ctx->d_ccount) (void)hipFree(ctx->d_ccount);
        ctx->d_cmask = nullptr;
        ctx->d_cpayload = nullptr;
        ctx->d_ccount = nullptr;
        ctx->compact_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cmask, (size_t)n * mask_words * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cpayload, (size_t)n * blocks * 16 * sizeof(int16_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ccount, (size_t)n * sizeof(unsigned)));
        ctx->compact_cap = n;
    }
    hipStream_t cs = ctx->copy_stream;
    hipEvent_t last = nullptr; // the slots of one call usually share one encode call's event
    for (int s = first_slot; s < first_slot + n; ++s)
        if (ctx->slot_event[s] && ctx->slot_event[s] != last) {
            last = ctx->slot_event[s];
            HIP_TRY(ctx, hipStreamWaitEvent(cs, last, 0));
        }
    hipLaunchKernelGGL(compact_levels_kernel, dim3(n), dim3(1024), 0, cs, ctx->d_slots, first_slot, c.width, c.height, ctx->d_cmask,
                       mask_words, ctx->d_cpayload, blocks, ctx->d_ccount);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<unsigned> counts((size_t)n);
    int ovf = 0;
    HIP_TRY(ctx, hipMemcpyAsync(counts.data(), ctx->d_ccount, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipMemcpyAsync(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipStreamSynchronize(cs));
    if (ovf & 2) return fail(ctx, WRENC_GPU_EHIP, "internal: a team member never reached a meeting point of the level schedule");
    if (ovf) return fail(ctx, WRENC_GPU_ELEVEL, "a quantised level reached 1024 (reference panics: block_splitter.rs:453)");
    bool short_buf = false;
    const size_t n4 = (size_t)(c.width / 4) * (c.height / 4), n8 = (size_t)(c.width / 8) * (c.height / 8);
    for (int k = 0; k < n; ++k) {
        wrenc_gpu_compact& o = out[k];
        const PicBufs& b = ctx->slots[first_slot + k];
        o.n_blocks = counts[(size_t)k];
        if (o.mask) HIP_TRY(ctx, hipMemcpyAsync(o.mask, ctx->d_cmask + (size_t)k * mask_words, mask_words * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
        if (o.n_blocks > o.payload_cap || (o.n_blocks && !o.payload)) {
            short_buf = true;
        } else if (o.n_blocks) {
            HIP_TRY(ctx, hipMemcpyAsync(o.payload, ctx->d_cpayload + (size_t)k * blocks * 16, o.n_blocks * 16 * sizeof(int16_t),
                                        hipMemcpyDeviceToHost, cs));
        }
        if (o.cu_log2_size) HIP_TRY(ctx, hipMemcpyAsync(o.cu_log2_size, b.cu_log2, n4, hipMemcpyDeviceToHost, cs));
        if (o.luma_mode) HIP_TRY(ctx, hipMemcpyAsync(o.luma_mode, b.luma_mode, n4, hipMemcpyDeviceToHost, cs));
        if (o.chroma_mode) HIP_TRY(ctx, hipMemcpyAsync(o.chroma_mode, b.chroma_mode, n8, hipMemcpyDeviceToHost, cs));
        uint8_t* rec[3] = {o.rec_y, o.rec_cb, o.rec_cr};
        for (int p = 0; p < 3; ++p)
            if (rec[p]) HIP_TRY(ctx, hipMemcpyAsync(rec[p], b.rec[p], plane_bytes(c, p, 1), hipMemcpyDeviceToHost, cs));
    }
    HIP_TRY(ctx, hipStreamSynchronize(cs));
    if (short_buf) return fail(ctx, WRENC_GPU_ENOMEM, "wrenc_gpu_download_compact: payload_cap is smaller than n_blocks of a picture");
    return WRENC_GPU_OK;
}

int wrenc_gpu_download_tokens(wrenc_gpu_ctx* ctx, int first_slot, int n, wrenc_gpu_tokens* out, uint32_t* pool, size_t pool_cap_words,
                              size_t* pool_words_used) {
    if (!ctx || !out || !pool || !pool_words_used || n < 1) return WRENC_GPU_EINVAL;
    if (first_slot < 0 || first_slot + n > ctx->cfg.n_slots) return fail(ctx, WRENC_GPU_EINVAL, "bad slot range");
    for (int s = first_slot; s < first_slot + n; ++s)
        if (ctx->state[s] != 2) return fail(ctx, WRENC_GPU_ESTATE, "slot has not been encoded");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const wrenc_gpu_config& c = ctx->cfg;
    const int ctus = ctx->ctu_cols * ctx->ctu_rows;
    const size_t pages = pool_cap_words / kTokPage;
    if (pages < 1 || pages > 0xFFFFFFF0u) return fail(ctx, WRENC_GPU_EINVAL, "wrenc_gpu_download_tokens: pool_cap_words");
    if (ctx->tok_pool_words < pages * kTokPage) {
        if (ctx->d_tok_pool) (void)hipFree(ctx->d_tok_pool);
        ctx->d_tok_pool = nullptr;
        ctx->tok_pool_words = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tok_pool, pages * kTokPage * sizeof(uint32_t)));
        ctx->tok_pool_words = pages * kTokPage;
    }
    constexpr int kCounterWords = kTokPools * kTokCounterStride + 1; // the sub-pools' page counters, the "a sub-pool ran out" flag
    if (!ctx->d_tok_counter) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tok_counter, kCounterWords * sizeof(unsigned)));
    if (ctx->tok_first_cap < n) {
        if (ctx->d_tok_first) (void)hipFree(ctx->d_tok_first);
        ctx->d_tok_first = nullptr;
        ctx->tok_first_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tok_first, (size_t)n * ctus * sizeof(uint32_t)));
        ctx->tok_first_cap = n;
    }
    hipStream_t cs = ctx->copy_stream;
    hipEvent_t last = nullptr;
    for (int s = first_slot; s < first_slot + n; ++s)
        if (ctx->slot_event[s] && ctx->slot_event[s] != last) {
            last = ctx->slot_event[s];
            HIP_TRY(ctx, hipStreamWaitEvent(cs, last, 0));
        }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_tok_counter, 0, kCounterWords * sizeof(unsigned), cs));
    const int waves = n * ctus;
    // sub-pools: enough CTUs in each (16 or more) that they fill evenly; one for a handful of CTUs
    int n_pools = 1;
    while (n_pools < kTokPools && n_pools * 32 <= waves) n_pools *= 2;
    const size_t sub = pages / (size_t)n_pools; // pages per sub-pool
    hipLaunchKernelGGL(residual_tokens_kernel, dim3((waves + 3) / 4), dim3(256), 0, cs, ctx->d_const, ctx->d_slots, first_slot, n,
                       ctx->d_tok_pool, (unsigned)sub, n_pools - 1, ctx->d_tok_counter, ctx->d_tok_first, (int*)(ctx->d_tok_counter + kCounterWords - 1));
    HIP_TRY(ctx, hipGetLastError());
    std::vector<unsigned> cnt((size_t)kCounterWords);
    int ovf = 0;
    HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), ctx->d_tok_counter, kCounterWords * sizeof(unsigned), hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipMemcpyAsync(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipStreamSynchronize(cs));
    if (ovf & 2) return fail(ctx, WRENC_GPU_EHIP, "internal: a team member never reached a meeting point of the level schedule");
    if (ovf) return fail(ctx, WRENC_GPU_ELEVEL, "a quantised level reached 1024 (reference panics: block_splitter.rs:453)");
    size_t asked = 0;
    bool short_of = cnt[(size_t)kCounterWords - 1] != 0;
    for (int p = 0; p < n_pools; ++p) {
        asked += cnt[(size_t)p * kTokCounterStride];
        short_of = short_of || cnt[(size_t)p * kTokCounterStride] > sub;
    }
    *pool_words_used = asked * kTokPage;
    if (short_of) return fail(ctx, WRENC_GPU_ENOMEM, "wrenc_gpu_download_tokens: the token pool is too small for these pictures");
    for (int p = 0; p < n_pools; ++p) // every sub-pool's pages in use, to the same place in the caller's pool
        if (cnt[(size_t)p * kTokCounterStride])
            HIP_TRY(ctx, hipMemcpyAsync(pool + (size_t)p * sub * kTokPage, ctx->d_tok_pool + (size_t)p * sub * kTokPage,
                                        (size_t)cnt[(size_t)p * kTokCounterStride] * kTokPage * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    const size_t n4 = (size_t)(c.width / 4) * (c.height / 4), n8 = (size_t)(c.width / 8) * (c.height / 8);
    for (int k = 0; k < n; ++k) {
        wrenc_gpu_tokens& o = out[k];
        const PicBufs& b = ctx->slots[first_slot + k];
        if (o.first_page) HIP_TRY(ctx, hipMemcpyAsync(o.first_page, ctx->d_tok_first + (size_t)k * ctus, (size_t)ctus * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
        if (o.cu_log2_size) HIP_TRY(ctx, hipMemcpyAsync(o.cu_log2_size, b.cu_log2, n4, hipMemcpyDeviceToHost, cs));
        if (o.luma_mode) HIP_TRY(ctx, hipMemcpyAsync(o.luma_mode, b.luma_mode, n4, hipMemcpyDeviceToHost, cs));
        if (o.chroma_mode) HIP_TRY(ctx, hipMemcpyAsync(o.chroma_mode, b.chroma_mode, n8, hipMemcpyDeviceToHost, cs));
        uint8_t* rec[3] = {o.rec_y, o.rec_cb, o.rec_cr};
        for (int p = 0; p < 3; ++p)
            if (rec[p]) HIP_TRY(ctx, hipMemcpyAsync(rec[p], b.rec[p], plane_bytes(c, p, 1), hipMemcpyDeviceToHost, cs));
    }
    HIP_TRY(ctx, hipStreamSynchronize(cs));
    return WRENC_GPU_OK;
}

int wrenc_gpu_test_load_record(wrenc_gpu_ctx* ctx, int slot, const wrenc_gpu_picture* rec) {
    if (!ctx || !rec || !rec->cu_log2_size || !rec->luma_mode || !rec->chroma_mode || !rec->lev_y || !rec->lev_cb || !rec->lev_cr)
        return WRENC_GPU_EINVAL;
    if (slot < 0 || slot >= ctx->cfg.n_slots) return fail(ctx, WRENC_GPU_EINVAL, "bad slot");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    const wrenc_gpu_config& c = ctx->cfg;
    const PicBufs& b = ctx->slots[slot];
    const size_t n4 = (size_t)(c.width / 4) * (c.height / 4), n8 = (size_t)(c.width / 8) * (c.height / 8);
    const int16_t* lev[3] = {rec->lev_y, rec->lev_cb, rec->lev_cr};
    for (int k = 0; k < 3; ++k) HIP_TRY(ctx, hipMemcpy(b.lev[k], lev[k], plane_bytes(c, k, 2), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(b.cu_log2, rec->cu_log2_size, n4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(b.luma_mode, rec->luma_mode, n4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(b.chroma_mode, rec->chroma_mode, n8, hipMemcpyHostToDevice));
    // (the slot's "these 4x4 blocks may be non-zero" bookkeeping no longer matches its planes: every block may be)
    HIP_TRY(ctx, hipMemset(b.lev_dirty, 0xFF, (size_t)ctx->ctu_cols * ctx->ctu_rows * 4 * sizeof(uint32_t)));
    ctx->state[slot] = 2;
    return WRENC_GPU_OK;
}

void wrenc_gpu_expand_levels(int width, int height, const uint32_t* mask, const int16_t* payload, int16_t* lev_y, int16_t* lev_cb,
                             int16_t* lev_cr) {
    const size_t wh = (size_t)width * height;
    memset(lev_y, 0, wh * sizeof(int16_t));
    memset(lev_cb, 0, wh / 4 * sizeof(int16_t));
    memset(lev_cr, 0, wh / 4 * sizeof(int16_t));
    const int bw = width / 4, bh = height / 4;
    const size_t nl = (size_t)bw * bh, ncb = nl / 4, total = nl + 2 * ncb;
    size_t at = 0;
    for (size_t w0 = 0; w0 < total; w0 += 32) {
        uint32_t m = mask[w0 >> 5];
        while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1;
            const size_t b = w0 + (size_t)bit;
            if (b >= total) break;
            int16_t* plane = lev_y;
            size_t pb_ = b;
            int pw = bw, stride = width;
            if (b >= nl) {
                const size_t cidx = b - nl;
                const int pl = cidx >= ncb ? 1 : 0;
                pb_ = cidx - (size_t)pl * ncb;
                pw = bw / 2;
                stride = width / 2;
                plane = pl ? lev_cr : lev_cb;
            }
            const size_t by = pb_ / (size_t)pw, bx = pb_ - by * (size_t)pw;
            int16_t* d = plane + (4 * by) * (size_t)stride + 4 * bx;
            const int16_t* s = payload + 16 * at++;
            for (int r = 0; r < 4; ++r) memcpy(d + (size_t)r * stride, s + 4 * r, 4 * sizeof(int16_t));
        }
    }
}

void* wrenc_gpu_alloc_host(wrenc_gpu_ctx* ctx, size_t bytes) {
    if (!ctx || bytes == 0) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)fail(ctx, WRENC_GPU_ENOMEM, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void wrenc_gpu_free_host(wrenc_gpu_ctx* ctx, void* p) {
    if (ctx && p) (void)hipHostFree(p);
}

int wrenc_gpu_encode_picture(wrenc_gpu_ctx* ctx, const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                             wrenc_gpu_picture* out) {
    if (!ctx) return WRENC_GPU_EINVAL;
    int rc = wrenc_gpu_upload(ctx, 0, y, cb, cr, ctx->cfg.width, ctx->cfg.width / 2);
    if (rc) return rc;
    rc = wrenc_gpu_encode(ctx, 0, 1);
    if (rc) return rc;
    return wrenc_gpu_download(ctx, 0, out);
}

int wrenc_gpu_last_encode_stats(wrenc_gpu_ctx* ctx, float* total_ms, float* kernel_ms_sum, int* n_launches) {
    if (!ctx) return WRENC_GPU_EINVAL;
    if (!ctx->stats_enabled) return fail(ctx, WRENC_GPU_ESTATE, "per-launch timing is off (wrenc_gpu_stats_enable)");
    if (!ctx->stats_valid) return fail(ctx, WRENC_GPU_ESTATE, "no encode has been queued since timing was switched on");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
    float t = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev_begin, ctx->ev_end));
    float sum = 0.f;
    for (int i = 0; i < ctx->last_launches; ++i) {
        float k = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&k, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]));
        sum += k;
    }
    if (total_ms) *total_ms = t;
    if (kernel_ms_sum) *kernel_ms_sum = sum;
    if (n_launches) *n_launches = ctx->last_launches;
    return WRENC_GPU_OK;
}

int wrenc_gpu_last_encode_kernel_stats(wrenc_gpu_ctx* ctx, wrenc_gpu_kernel_stats out[2]) {
    if (!ctx || !out) return WRENC_GPU_EINVAL;
    if (!ctx->stats_enabled) return fail(ctx, WRENC_GPU_ESTATE, "per-launch timing is off (wrenc_gpu_stats_enable)");
    if (!ctx->stats_valid) return fail(ctx, WRENC_GPU_ESTATE, "no encode has been queued since timing was switched on");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
    for (int k = 0; k < 2; ++k) out[k] = wrenc_gpu_kernel_stats{0.f, 0, 0};
    for (int i = 0; i < ctx->last_launches; ++i) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]));
        wrenc_gpu_kernel_stats& o = out[ctx->launch_team[(size_t)i] ? 1 : 0];
        o.ms_sum += ms;
        o.launches += 1;
        o.ctu_pictures += ctx->launch_ctus[(size_t)i];
    }
    return WRENC_GPU_OK;
}

int wrenc_gpu_set_schedule(wrenc_gpu_ctx* ctx, int schedule) {
    if (!ctx) return WRENC_GPU_EINVAL;
    if (schedule != WRENC_GPU_SCHEDULE_AUTO && schedule != WRENC_GPU_SCHEDULE_WAVE && schedule != WRENC_GPU_SCHEDULE_TEAM)
        return fail(ctx, WRENC_GPU_EINVAL, "schedule must be WRENC_GPU_SCHEDULE_AUTO, _WAVE or _TEAM");
    ctx->schedule = schedule;
    return WRENC_GPU_OK;
}

int wrenc_gpu_last_schedule(const wrenc_gpu_ctx* ctx) { return ctx ? ctx->last_schedule : WRENC_GPU_EINVAL; }

int wrenc_gpu_device_info(const wrenc_gpu_ctx* ctx, long long* wave_slots, int* encode_lanes) {
    if (!ctx) return WRENC_GPU_EINVAL;
    if (wave_slots) *wave_slots = ctx->device_wave_slots;
    if (encode_lanes) *encode_lanes = kEncodeLanes;
    return WRENC_GPU_OK;
}

int wrenc_gpu_test_set_wave_slots(wrenc_gpu_ctx* ctx, long long slots) {
    if (!ctx) return WRENC_GPU_EINVAL;
    ctx->wave_slots = slots > 0 ? slots : ctx->device_wave_slots;
    return WRENC_GPU_OK;
}

int wrenc_gpu_test_head_ranges(wrenc_gpu_ctx* ctx, int counts[4], int ranges[24]) {
    if (!ctx || !counts) return WRENC_GPU_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, 4 * sizeof(int)));
    hipError_t e = hipMemset(d, 0, 4 * sizeof(int));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(test_head_ranges_kernel, dim3(1024, 8), dim3(64), 0, 0, ctx->d_const, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(counts, d, 4 * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(ctx, e);
    if (ranges) {
        DevConst* hk = new DevConst;
        fill_dev_const(ctx->cfg, *hk);
        memcpy(ranges, hk->head_rng, sizeof(hk->head_rng));
        delete hk;
    }
    return WRENC_GPU_OK;
}

int wrenc_gpu_test_avail_tab(wrenc_gpu_ctx* ctx, int* differences) {
    if (!ctx || !differences) return WRENC_GPU_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    int* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(int)));
    hipError_t e = hipMemset(d, 0, sizeof(int));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(test_avail_tab_kernel, dim3(ctx->ctu_cols * ctx->ctu_rows, 4), dim3(64), 0, 0, ctx->d_const, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(differences, d, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(ctx, e);
    return WRENC_GPU_OK;
}

int wrenc_gpu_test_scratch_overflows(wrenc_gpu_ctx* ctx, long long* count) {
    if (!ctx || !count) return WRENC_GPU_EINVAL;
    *count = 0;
    if (!ctx->d_slot_map) return WRENC_GPU_OK; // no encode call yet
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->d_slot_map + kScratchSlots / 64, sizeof(v), hipMemcpyDeviceToHost));
    *count = (long long)v;
    return WRENC_GPU_OK;
}

int wrenc_gpu_stats_enable(wrenc_gpu_ctx* ctx, int on) {
    if (!ctx) return WRENC_GPU_EINVAL;
    ctx->stats_enabled = on != 0;
    ctx->stats_valid = false;
    return WRENC_GPU_OK;
}

int wrenc_gpu_final_pass_mismatches(wrenc_gpu_ctx* ctx, long long* count) {
    if (!ctx || !count) return WRENC_GPU_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->d_mismatch, sizeof(v), hipMemcpyDeviceToHost));
    *count = (long long)v;
    return WRENC_GPU_OK;
}

#ifdef WRENC_TRACE
// diagnostic build only: copies up to max_records records (8 ints each) of the last encode call's trace,
// returns the number of records made; resets the trace
extern "C" long wrenc_gpu_trace_read(wrenc_gpu_ctx* ctx, int* out, long max_records) {
    if (!ctx) return -1;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), sizeof(n)) != hipSuccess) return -1;
    long take = n < kTraceMax ? (long)n : (long)kTraceMax;
    if (take > max_records) take = max_records;
    if (take > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), (size_t)take * 8 * sizeof(int)) != hipSuccess) return -1;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (long)n;
}
#endif

#ifdef WRENC_PROFILE
// diagnostic build only: read and clear the per-phase cycle counters
int wrenc_gpu_prof_read(wrenc_gpu_ctx* ctx, unsigned long long* out, int n) {
    if (!ctx || !out) return WRENC_GPU_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long host[PH_COUNT];
    HIP_TRY(ctx, hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(host)));
    for (int i = 0; i < n && i < PH_COUNT; ++i) out[i] = host[i];
    memset(host, 0, sizeof(host));
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_prof), host, sizeof(host)));
    return WRENC_GPU_OK;
}
#endif

// ---- building-block entry points ----

// the 32x32 forward transform runs as i8 MFMAs on two base-256 digits of the residual (fwd_dct32_mfma): exact for
// |residual| <= 255, which is all the search can produce; the test entries refuse anything else
static bool residuals_fit_9_bits(const int16_t* res, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (res[i] < -255 || res[i] > 255) return false;
    return true;
}

int wrenc_gpu_test_fwd_dct(wrenc_gpu_ctx* ctx, const int16_t* res, int log2n, int count, int16_t* coef) {
    if (ctx && res && log2n == 5 && count >= 1 && !residuals_fit_9_bits(res, (size_t)count * 1024))
        return fail(ctx, WRENC_GPU_EINVAL, "wrenc_gpu_test_fwd_dct: a 32x32 residual lies outside +-255");
    return run_block_test(ctx, res, log2n, count, coef, [&](int16_t* i, int16_t* o) {
        hipLaunchKernelGGL(test_fwd_dct_kernel, dim3(count), dim3(64), 0, ctx->stream, ctx->d_const, i, log2n, o);
    });
}
// a 32x32 transform micro-benchmark: `count` blocks, `reps` repetitions each, HIP-event duration of the one kernel
static int run_dct32_bench(wrenc_gpu_ctx* ctx, const int16_t* in, int count, int16_t* out, int use_mfma, int reps,
                           float* kernel_ms, bool inverse) {
    if (!ctx || !in || !out || count < 1 || reps < 1) return WRENC_GPU_EINVAL;
    if (!inverse && use_mfma && !residuals_fit_9_bits(in, (size_t)count * 1024))
        return fail(ctx, WRENC_GPU_EINVAL, "wrenc_gpu_test_fwd_dct32: a residual lies outside +-255 (MFMA path)");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t bytes = (size_t)count * 1024 * sizeof(int16_t);
    int16_t *d_in = nullptr, *d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_in, bytes));
    hipError_t e = hipMalloc((void**)&d_out, bytes);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) {
        if (inverse)
            hipLaunchKernelGGL(test_inv_dct32_kernel, dim3(count), dim3(64), 0, ctx->stream, ctx->d_const, d_in, d_out, use_mfma, reps);
        else
            hipLaunchKernelGGL(test_fwd_dct32_kernel, dim3(count), dim3(64), 0, ctx->stream, ctx->d_const, d_in, d_out, use_mfma, reps);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
    if (kernel_ms) *kernel_ms = ms;
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(ctx, WRENC_GPU_EHIP, hipGetErrorString(e));
    return WRENC_GPU_OK;
}
int wrenc_gpu_test_fwd_dct32(wrenc_gpu_ctx* ctx, const int16_t* res, int count, int16_t* coef, int use_mfma, int reps,
                             float* kernel_ms) {
    return run_dct32_bench(ctx, res, count, coef, use_mfma, reps, kernel_ms, false);
}
int wrenc_gpu_test_inv_dct32(wrenc_gpu_ctx* ctx, const int16_t* deq, int count, int16_t* res, int use_mfma, int reps,
                             float* kernel_ms) {
    return run_dct32_bench(ctx, deq, count, res, use_mfma, reps, kernel_ms, true);
}
int wrenc_gpu_test_inv_dct(wrenc_gpu_ctx* ctx, const int16_t* deq, int log2n, int count, int16_t* res) {
    return run_block_test(ctx, deq, log2n, count, res, [&](int16_t* i, int16_t* o) {
        hipLaunchKernelGGL(test_inv_dct_kernel, dim3(count), dim3(64), 0, ctx->stream, ctx->d_const, i, log2n, o);
    });
}
int wrenc_gpu_test_dequantize(wrenc_gpu_ctx* ctx, const int16_t* levels, int log2n, int count, int16_t* deq) {
    return run_block_test(ctx, levels, log2n, count, deq, [&](int16_t* i, int16_t* o) {
        hipLaunchKernelGGL(test_dequantize_kernel, dim3(count), dim3(64), 0, ctx->stream, ctx->d_const, i, log2n, o);
    });
}
int wrenc_gpu_test_quantize(wrenc_gpu_ctx* ctx, const int16_t* coef, int log2n, int count, int16_t* levels,
                            int64_t* level_cost) {
    if (!ctx || !level_cost) return WRENC_GPU_EINVAL;
    if (count < 1) return fail(ctx, WRENC_GPU_EINVAL, "count must be >= 1");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    long long* d_cost = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_cost, sizeof(long long) * count));
    int rc = run_block_test(ctx, coef, log2n, count, levels, [&](int16_t* i, int16_t* o) {
        hipLaunchKernelGGL(test_quantize_kernel, dim3(count), dim3(64), 0, ctx->stream, ctx->d_const, i, log2n, o,
                           d_cost, ctx->d_overflow + 1);
    });
    if (rc == WRENC_GPU_OK) {
        hipError_t e = hipMemcpy(level_cost, d_cost, sizeof(long long) * count, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, WRENC_GPU_EHIP, hipGetErrorString(e));
    }
    (void)hipFree(d_cost);
    return rc;
}

int wrenc_gpu_test_quantize_p16(wrenc_gpu_ctx* ctx, const int16_t* coef, int count, int16_t* levels, int64_t* level_cost) {
    if (!ctx || !level_cost) return WRENC_GPU_EINVAL;
    if (count < 1) return fail(ctx, WRENC_GPU_EINVAL, "count must be >= 1");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    long long* d_cost = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_cost, sizeof(long long) * count));
    int rc = run_block_test(ctx, coef, 2, count, levels, [&](int16_t* i, int16_t* o) {
        hipLaunchKernelGGL(test_quantize_p16_kernel, dim3((count + 3) / 4), dim3(64), 0, ctx->stream, ctx->d_const, i, count, o,
                           d_cost, ctx->d_overflow + 1);
    });
    if (rc == WRENC_GPU_OK) {
        hipError_t e = hipMemcpy(level_cost, d_cost, sizeof(long long) * count, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, WRENC_GPU_EHIP, hipGetErrorString(e));
    }
    (void)hipFree(d_cost);
    return rc;
}

int wrenc_gpu_test_quantize_pk(wrenc_gpu_ctx* ctx, const int16_t* coef, int log2n, int nc, int n_packs, int16_t* levels,
                               int64_t* level_cost) {
    if (!ctx || !coef || !levels || !level_cost) return WRENC_GPU_EINVAL;
    if (!((log2n == 3 && nc >= 1 && nc <= 3) || (log2n == 4 && nc >= 1 && nc <= 2)) || n_packs < 1)
        return fail(ctx, WRENC_GPU_EINVAL, "wrenc_gpu_test_quantize_pk: log2n 3 with nc 1..3 or log2n 4 with nc 1..2");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t total = (size_t)n_packs * nc * 3 * ((size_t)1 << (2 * log2n)) / 2;
    int16_t *d_in = nullptr, *d_out = nullptr;
    long long* d_cost = nullptr;
    hipError_t e = hipMalloc((void**)&d_in, total * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, total * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_cost, sizeof(long long) * 2 * nc * n_packs);
    if (e == hipSuccess) e = hipMemcpy(d_in, coef, total * sizeof(int16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(test_quantize_pk_kernel, dim3(n_packs), dim3(64), 0, ctx->stream, ctx->d_const, d_in, log2n, nc, d_out,
                           d_cost, ctx->d_overflow + 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(levels, d_out, total * sizeof(int16_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(level_cost, d_cost, sizeof(long long) * 2 * nc * n_packs, hipMemcpyDeviceToHost);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_cost) (void)hipFree(d_cost);
    if (e != hipSuccess) return fail(ctx, WRENC_GPU_EHIP, hipGetErrorString(e));
    return WRENC_GPU_OK;
}

int wrenc_gpu_test_predict(wrenc_gpu_ctx* ctx, const uint8_t* rec_y, const uint8_t* rec_cb, const uint8_t* rec_cr,
                           int n_items, const int32_t* items, uint8_t* out, size_t out_bytes) {
    if (!ctx || !rec_y || !rec_cb || !rec_cr || !items || !out || n_items < 1) return WRENC_GPU_EINVAL;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    // operand shapes are checked on the host: a bad item must never reach the kernel
    std::vector<int> dev_items((size_t)n_items * 6);
    size_t total = 0;
    for (int i = 0; i < n_items; ++i) {
        const int32_t* q = items + 5 * i;
        const int x = q[0], y = q[1], lg = q[2], comp = q[3], mode = q[4];
        const int n = 1 << lg;
        const bool list = comp >= 4 && comp <= 6; // a SAD list: mode = m0 | entries << 8 | stride << 16
        const int m0 = mode & 255, nm = (mode >> 8) & 255, stride = mode >> 16;
        const bool ok = lg >= 2 && lg <= 5 && x >= 0 && y >= 0 && x + n <= W && y + n <= H && !(x & (n - 1)) && !(y & (n - 1)) &&
                        (comp == 0 || (comp == 1 && lg >= 3) || (comp == 2 && lg == 2) || (comp == 4) || (list && lg >= 3) ||
                         (comp == 7 && lg >= 3)) &&
                        (comp == 7 ? mode == 0 : list ? (mode >= 0 && m0 >= 2 && m0 <= 66 && nm >= 1 && nm <= 13 && stride >= 1 && stride <= 64)
                              : ((mode >= 0 && mode <= 66) || (comp == 1 && mode >= LT_CCLM && mode <= T_CCLM)));
        if (!ok) return fail(ctx, WRENC_GPU_EINVAL, "wrenc_gpu_test_predict: bad item");
        int* d = &dev_items[(size_t)i * 6];
        d[0] = x; d[1] = y; d[2] = lg; d[3] = comp; d[4] = mode; d[5] = (int)total;
        total += (list || comp == 7) ? 64 : (comp == 1 ? (size_t)n * n / 2 : (size_t)n * n);
    }
    if (total != out_bytes) return fail(ctx, WRENC_GPU_EINVAL, "wrenc_gpu_test_predict: output size does not match the items");
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    const size_t wh = (size_t)W * H;
    uint8_t *d_planes = nullptr, *d_scratch = nullptr, *d_out = nullptr;
    int* d_items = nullptr;
    hipError_t e = hipMalloc((void**)&d_planes, wh + wh / 2);
    if (e == hipSuccess) e = hipMalloc((void**)&d_scratch, (size_t)n_items * 1024);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, total);
    if (e == hipSuccess) e = hipMalloc((void**)&d_items, dev_items.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_planes, rec_y, wh, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_planes + wh, rec_cb, wh / 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_planes + wh + wh / 4, rec_cr, wh / 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_items, dev_items.data(), dev_items.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(test_predict_kernel, dim3(n_items), dim3(64), 0, ctx->stream, ctx->d_const, d_planes, d_items,
                           d_scratch, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, total, hipMemcpyDeviceToHost);
    if (d_planes) (void)hipFree(d_planes);
    if (d_scratch) (void)hipFree(d_scratch);
    if (d_out) (void)hipFree(d_out);
    if (d_items) (void)hipFree(d_items);
    if (e != hipSuccess) return fail(ctx, WRENC_GPU_EHIP, hipGetErrorString(e));
    return WRENC_GPU_OK;
}

} // extern "C"
