// dev_bins.h -- residual_coding of every transform block as a stream of CABAC tokens, made on the device (round 4).
// Part of the gfx950 device code; see wrenc_dev.h for the overall model.
//
// north_star keeps the serial arithmetic coder on the host, "fed by GPU-produced modes and coefficients".  Half of the
// host writer's time was not the coder but residual_coding's syntax walk (ctu_encoder.rs:1786-2269: context selection
// bool_coder.rs:2053-2400, binarisations :1133-1465) -- work that is serial in the reference only because it is written
// as a loop: within a transform block every bin's context depends on (i) the dependent-quantisation state, a chain of
// parities = a prefix composition of 4-state maps, (ii) the bin budget, a prefix sum, and (iii) levels of neighbours that
// are coded EARLIER (right / below), which are data, not coder state.  So one wavefront takes a CTU, walks its transform
// blocks in coding order and emits, 64 coefficients per step, the tokens the host's coder then consumes blindly:
//     context-coded bin   (ctx << 1) | bin                      ctx = index into the host's flat model array (cabac.h)
//     bypass group        1 << 31 | (nbits - 1) << 25 | value   nbits <= 25 (longer codes are split)
// Per transform unit (the host's transform_unit call): one header word per component present -- token count | flags --
// then the components' tokens, luma, Cb, Cr.  The CU-level syntax (split flags, modes, coded-block flags, cu_qp_delta,
// transform_skip_flag, mts_idx) stays on the host: a few bins per CU, all from the maps.
//
// Token storage: pages of kTokPage words, taken from a pool with an atomic counter as a CTU needs them; the last word of
// a page links to the next one, a table holds every CTU's first page.  (A CTU of the bench content needs one page; a
// CTU of noise at QP 22 about a hundred.)  The pool is kTokPools sub-pools with a counter (in a cache line of its own)
// each, CTU g taking its pages from sub-pool g mod their number: with ONE counter the pass ran at the rate of atomics on
// one address -- 11 ns per page, 83 ms for the 7.4 M pages of 256 textured 1080p pictures, whatever the code around them.
//
// Same bytes as the host-only writer: tests/test_gpu_tokens.py compares the two streams on searched pictures and on
// random records no search would emit (escape codes, exhausted bin budgets, every block size).
#pragma once

namespace wrenc {

constexpr int kTokPage = 64;               // words per page: kTokPage - 1 tokens + the link to the next page
constexpr int kTokPayload = kTokPage - 1;
constexpr int kTokMaxPages = 160;          // pages one CTU can need: < (1.75 + 2) * 1536 + 96 * 24 tokens
constexpr uint32_t kTokNone = 0xFFFFFFFFu;
constexpr int kTokPools = 64;              // sub-pools at most (a power of two; a call with few CTUs uses fewer, see wrenc_gpu_download_tokens)
constexpr int kTokCounterStride = 32;      // words from one sub-pool's counter to the next

// first context of each syntax element of residual_coding in the host's flat model array (host/cabac.h CtxBase)
constexpr int CTXD_LAST_X = 32, CTXD_LAST_Y = CTXD_LAST_X + 23, CTXD_SB_CODED = CTXD_LAST_Y + 23, CTXD_SIG = CTXD_SB_CODED + 7,
              CTXD_PAR = CTXD_SIG + 63, CTXD_GTX = CTXD_PAR + 33;

#ifndef WRENC_TOKENS_4X4_ROWS
#define WRENC_TOKENS_4X4_ROWS 1 // 0: every 4x4 block a step of its own (as first built; for A/B runs)
#endif
struct TokLds {
    int16_t lv[1024];                 // the transform block's levels, raster
    uint16_t tpl[34 * 34 + 2];        // AbsLevelPass1 | significant << 8, two zero columns / rows behind the block
    uint8_t ab[34 * 34 + 2];          // min(AbsLevel, 255)
    uint32_t pt[kTokMaxPages];        // the CTU's pages in order
};
#ifdef WRENC_TOKENS_KERNEL_TU
__shared__ TokLds TOKW[4];
#define TK (TOKW[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))])

struct TokOut {
    GLOBAL_AS uint32_t* pool;
    unsigned* page_counter;   // of this CTU's sub-pool
    unsigned pool_first;      // the sub-pool's first page
    unsigned pool_pages;      // pages per sub-pool
    int n_tok;        // tokens of this CTU so far (virtual index of the next one)
    int n_pages;      // pages it holds
    int* overflow;
    bool dead;        // the pool ran out: nothing more is written (the host falls back to the level planes)
};

__device__ __forceinline__ GLOBAL_AS uint32_t* tok_addr(const TokOut& o, int v) {
    const int pg = v / kTokPayload;
    return o.pool + (size_t)TK.pt[pg] * kTokPage + (v - pg * kTokPayload);
}
// make room for `count` more tokens (uniform)
__device__ __forceinline__ void tok_reserve(TokOut& o, int count) {
    while (!o.dead && o.n_pages * kTokPayload < o.n_tok + count) {
        unsigned pg = 0;
        if (LANE == 0) pg = atomicAdd(o.page_counter, 1u);
        pg = (unsigned)uni((int)pg);
        if (pg >= o.pool_pages || o.n_pages >= kTokMaxPages) {
            o.dead = true;
            if (LANE == 0) atomicOr(o.overflow, 1);
            break;
        }
        pg += o.pool_first;
        if (LANE == 0) {
            o.pool[(size_t)pg * kTokPage + kTokPayload] = kTokNone;
            if (o.n_pages > 0) o.pool[(size_t)TK.pt[o.n_pages - 1] * kTokPage + kTokPayload] = pg;
            TK.pt[o.n_pages] = pg;
        }
        ++o.n_pages;
        WSYNC();
    }
}
__device__ __forceinline__ void tok_put(const TokOut& o, int v, uint32_t t) {
    if (!o.dead) *tok_addr(o, v) = t;
}
__device__ __forceinline__ uint32_t tok_ctx(int ctx, int bin) { return ((uint32_t)ctx << 1) | (uint32_t)(bin & 1); }
__device__ __forceinline__ uint32_t tok_bypass(uint32_t value, int nbits) {
    return 0x80000000u | ((uint32_t)(nbits - 1) << 25) | (value & 0x1FFFFFFu);
}

// exclusive prefix sum over the wave in lane order; *total = the wave's sum
__device__ __forceinline__ int wave_excl_sum(int v, int* total) {
    int s = v;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xF, 0xF, false); // row_shr:1
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xF, 0xF, false); // row_shr:2
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xF, 0xF, false); // row_shr:4
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xF, 0xF, false); // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(s, 15), r1 = __builtin_amdgcn_readlane(s, 31), r2 = __builtin_amdgcn_readlane(s, 47),
              r3 = __builtin_amdgcn_readlane(s, 63);
    const int row = LANE >> 4;
    const int before = row == 0 ? 0 : (row == 1 ? r0 : (row == 2 ? r0 + r1 : r0 + r1 + r2));
    *total = r0 + r1 + r2 + r3;
    return before + s - v;
}
// the same inside each row of 16 lanes; *row_total = the row's sum (in every lane of the row)
__device__ __forceinline__ int row_excl_sum(int v, int* row_total) {
    int s = v;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xF, 0xF, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xF, 0xF, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xF, 0xF, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xF, 0xF, false);
    *row_total = row_sum_i32(v);
    return s - v;
}

// (prefix, suffix) of a last-significant coordinate (ctu_encoder.rs:1818-1851)
__device__ __forceinline__ void split_last_d(int v, int& prefix, int& suffix) {
    if (v <= 3) {
        prefix = v;
        suffix = 0;
        return;
    }
    const int bits = 30 - __clz(v);      // the smallest shift with v >> bits < 4
    const int p = v >> bits;
    suffix = v - (p << bits);
    prefix = ((bits + 1) << 1) + (p & 1);
}

// abs_remainder / dec_abs_level (bool_coder.rs:1384-1465, :1305-1331): Rice prefix with cMax 6 << k, limited Exp-Golomb of
// order k + 1 behind it; one bypass group, or two when the escape code is taken.  Returns the number of tokens.
__device__ __forceinline__ int remainder_tokens(int val, int k, uint32_t* t0, uint32_t* t1) {
    const int c_max = 6 << k;
    const int pv = min(val, c_max);
    const int pre = pv >> k;
    if (pre < 6) {
        const uint32_t ones = (1u << pre) - 1u;
        *t0 = tok_bypass(((ones << 1) << k) | (uint32_t)(pv - (pre << k)), pre + 1 + k);
        *t1 = 0;
        return 1;
    }
    int sym = val - c_max;
    const int kk = k + 1;
    const int cv = sym >> kk;
    const int pre_ext = min(31 - __clz(cv + 1), 11);
    const int z = pre_ext < 11 ? 1 : 0;
    const int escape = pre_ext == 11 ? 15 : pre_ext + kk;
    *t0 = tok_bypass(((1u << (6 + pre_ext)) - 1u) << z, 6 + pre_ext + z);
    sym -= ((1 << pre_ext) - 1) << kk;
    *t1 = tok_bypass((uint32_t)sym, escape);
    return 2;
}

__device__ __forceinline__ int rice_of(int s) { // cabac_contexts.rs:919
    s = min(max(s, 0), 31);
    return s < 7 ? 0 : (s < 14 ? 1 : (s < 28 ? 2 : 3));
}

// residual_coding of one transform block (ctu_encoder.rs:1786-2269): component c (0 luma, 1 Cb, 2 Cr), log2 size lg, its
// levels at lev[0 .. n) x stride.  Writes the tokens behind o.n_tok and returns the header word (0: no level, the block's
// coded flag is zero).
__device__ __forceinline__ uint32_t tb_tokens(const CONST_AS DevConst* k, TokOut& o, int c, int lg, const GLOBAL_AS int16_t* lev, int stride) {
    const int lane = lane_fresh();
    const int n = 1 << lg, P = n * n;
    const CONST_AS uint16_t* scan = k->scan_idx[lg - 2];
    // ---- stage the levels (rows of the plane) and clear the neighbourhood arrays ----
    int nzl = 0;
    for (int i = lane; i < P / 4; i += 64) { // four levels of one row per lane
        const int y = (4 * i) >> lg, x = (4 * i) & (n - 1);
        const unsigned long long v4 = *(const GLOBAL_AS unsigned long long*)(lev + (size_t)y * stride + x);
        *(unsigned long long*)&TK.lv[4 * i] = v4;
        nzl |= v4 != 0ULL;
    }
    if (__ballot(nzl != 0) == 0ULL) return 0u;
    for (int i = lane; i < (n + 2) * 34 / 2 + 1; i += 64) ((uint32_t*)TK.tpl)[i] = 0u;
    for (int i = lane; i < (n + 2) * 34 / 4 + 1; i += 64) ((uint32_t*)TK.ab)[i] = 0u;
    WSYNC();
    // ---- the last significant position: the first p (reverse-scan order: p = 0 is the last scan position) with a level ----
    int p_last = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        const unsigned long long b = __ballot(p < P && TK.lv[scan[p < P ? p : 0]] != 0);
        if (b != 0ULL) {
            p_last = p0 + (int)__builtin_ctzll(b);
            break;
        }
    }
    const int start = o.n_tok;
    const int n_sb = P >> 4;
    const int sbp_last = p_last >> 4;
    uint32_t flags = 0;
    {
        // last_sig_coeff_{x,y}_prefix (truncated unary, cMax 2 lg - 1, contexts bool_coder.rs:2053-2083) and suffixes
        const int r = scan[p_last];
        const int last_x = r & (n - 1), last_y = r >> lg;
        int px, sx, py, sy;
        split_last_d(last_x, px, sx);
        split_last_d(last_y, py, sy);
        int off, shift;
        if (c == 0) {
            off = lg == 2 ? 0 : (lg == 3 ? 3 : (lg == 4 ? 6 : 10)); // kOffsetY[lg - 1]
            shift = (lg + 1) >> 2;
        } else {
            off = 20;
            shift = min(n >> 3, 2);
        }
        const int c_max = 2 * lg - 1;
        const int nx = px + (px < c_max ? 1 : 0), ny = py + (py < c_max ? 1 : 0);
        const int sfx = px > 3 ? 1 : 0, sfy = py > 3 ? 1 : 0;
        const int total = nx + ny + sfx + sfy;
        tok_reserve(o, total);
        uint32_t t = 0;
        if (lane < nx)
            t = tok_ctx(CTXD_LAST_X + (lane >> shift) + off, lane < px);
        else if (lane < nx + ny)
            t = tok_ctx(CTXD_LAST_Y + ((lane - nx) >> shift) + off, (lane - nx) < py);
        else if (lane == nx + ny && sfx)
            t = tok_bypass((uint32_t)sx, (px >> 1) - 1);
        else
            t = tok_bypass((uint32_t)sy, (py >> 1) - 1);
        if (lane < total) tok_put(o, o.n_tok + lane, t);
        o.n_tok += total;
        if (c == 0 && p_last != P - 1) flags |= 1u << 30;       // (last_sb > 0 || last_pos > 0): MtsDcOnly = 0 (:1945-1947)
    }
    int rem = (P * 7) >> 2;           // the context-coded bin budget of the block
    int q_carry = 0;                  // dependent-quantisation state in front of the chunk
    unsigned long long coded = 0ULL;  // sub-blocks with a level so far, bit ys * 8 + xs
    const int sb_ctx = CTXD_SB_CODED + (c ? 2 : 0);
    const int sbw = n >> 2;
    for (int p0 = (p_last >> 6) << 6; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        const bool inb = p < P;
        const bool valid = inb && p >= p_last;
        const int r = scan[inb ? p : 0];
        const int xc = r & (n - 1), yc = r >> lg;
        const int v = valid ? (int)TK.lv[r] : 0;
        const int av = abs(v);
        const int row = lane >> 4;
        const int sbp = p >> 4;                        // sub-block in coding order
        const int nn = 15 - (lane & 15);               // position inside it, forward scan
        const int xs = xc >> 2, ys = yc >> 2;
        // which sub-blocks of the chunk hold a level
        const unsigned long long nzb = __ballot(v != 0);
        const bool real = ((nzb >> (lane & 48)) & 0xFFFFULL) != 0ULL;
        const bool in_rng = inb && sbp >= sbp_last;    // the sub-block is reached by the coding loop
        {
            // (one lane per row ORs its sub-block's bit in; the chunk's four bits are needed by its own rows)
            unsigned long long add = 0ULL;
#pragma unroll
            for (int rw = 0; rw < 4; ++rw) {
                const int sx_ = __builtin_amdgcn_readlane(xs, 16 * rw), sy_ = __builtin_amdgcn_readlane(ys, 16 * rw);
                const bool rl = ((nzb >> (16 * rw)) & 0xFFFFULL) != 0ULL;
                if (rl) add |= 1ULL << (sy_ * 8 + sx_);
            }
            coded |= add;
        }
        const bool coded_eff = in_rng && (real || sbp == n_sb - 1); // (:1994) the DC sub-block counts as coded
        const bool hasflag = inb && sbp > sbp_last && sbp < n_sb - 1;
        // dependent-quantisation state in front of every coefficient: prefix composition of the positions' state maps
        int pre = valid ? position_map(v, av, false, 0) : kMapId;
        pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
        pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
        pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
        pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
        pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
        pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x143, 0xC, 0xF, false)); // row_bcast:31 -> rows 2, 3
        int before = __builtin_amdgcn_update_dpp(kMapId, pre, 0x138, 0xF, 0xF, false);            // wave_shr:1
        if (lane == 0) before = kMapId;
        const int q_in = (before >> (8 * q_carry)) & 3;
        const int q_next = (__builtin_amdgcn_readlane(pre, 63) >> (8 * q_carry)) & 3;
        const int a = v ? (av + (q_in > 1 ? 1 : 0)) >> 1 : 0;   // AbsLevel (:1968-1985)
        const bool sig = a > 0, gt1 = a > 1, gt3 = a > 3;
        const bool is_last = p == p_last;
        // the DC position's sig_coeff_flag is inferred when the sub-block's coded flag was sent and nothing else in it is
        // significant
        const unsigned long long sgb = __ballot(sig);
        const bool others0 = ((sgb >> (lane & 48)) & 0x7FFFULL) == 0ULL; // lanes 0..14 of the row = positions 15..1
        const bool infer = nn == 0 && hasflag && others0;
        const bool act = coded_eff && valid;
        const bool sigcoded = act && !is_last && !infer;
        const int nb1 = act ? (sigcoded ? 1 : 0) + (sig ? (gt1 ? 3 : 1) : 0) : 0;
        int used;
        const int ex = wave_excl_sum(nb1, &used);
        const bool covered = act && rem - ex >= 4;               // pass 1 reaches the coefficient (:2014-2016)
        {
            int spent;
            (void)wave_excl_sum(covered ? nb1 : 0, &spent);
            rem -= spent;
        }
        const int p1 = covered && sig ? (gt1 ? 2 + (a & 1) + (gt3 ? 2 : 0) : 1) : 0;
        if (act) {
            TK.tpl[yc * 34 + xc] = (uint16_t)(p1 ? (256 | p1) : 0);
            TK.ab[yc * 34 + xc] = (uint8_t)min(a, 255);
        }
        WSYNC();
        int c1 = 0, c2 = 0;
        uint32_t t1[4] = {0, 0, 0, 0}, t2[2] = {0, 0};
        if (act) {
            const uint16_t* tp = &TK.tpl[yc * 34 + xc];
            const uint8_t* ap = &TK.ab[yc * 34 + xc];
            const int tsum = tp[1] + tp[2] + tp[34] + tp[35] + tp[68];
            const int asum = ap[1] + ap[2] + ap[34] + ap[35] + ap[68];
            const int d = xc + yc;
            if (covered) {
                const int sum_p1 = tsum & 255, num_sig = tsum >> 8;
                if (sigcoded) {
                    const int s = (sum_p1 + 1) >> 1;
                    const int qs = q_in > 1 ? q_in - 1 : 0;
                    const int inc = c == 0 ? 12 * qs + min(s, 3) + (d < 2 ? 8 : (d < 5 ? 4 : 0)) : 36 + 8 * qs + min(s, 3) + (d < 2 ? 4 : 0);
                    t1[c1++] = tok_ctx(CTXD_SIG + inc, sig);
                }
                if (sig) {
                    const int off = min(sum_p1 - num_sig, 4);
                    int inc;
                    if (is_last)
                        inc = c == 0 ? 0 : 21;
                    else if (c == 0)
                        inc = 1 + off + (d == 0 ? 15 : (d < 3 ? 10 : (d < 10 ? 5 : 0)));
                    else
                        inc = 22 + off + (d == 0 ? 5 : 0);
                    t1[c1++] = tok_ctx(CTXD_GTX + inc, gt1);
                    if (gt1) {
                        t1[c1++] = tok_ctx(CTXD_PAR + inc, a & 1);
                        t1[c1++] = tok_ctx(CTXD_GTX + 32 + inc, gt3);
                    }
                }
                if (gt3) c2 = remainder_tokens((a - 4) >> 1, rice_of(asum - 20), &t2[0], &t2[1]);   // abs_remainder (pass 2)
            } else {
                const int kr = rice_of(asum);                                                        // dec_abs_level (pass 3)
                const int zero_pos = (q_in < 2 ? 1 : 2) << kr;
                c2 = remainder_tokens(a == 0 ? zero_pos : (a <= zero_pos ? a - 1 : a), kr, &t2[0], &t2[1]);
            }
        }
        // signs of the sub-block in coding order, first one in the most significant bit
        const unsigned long long nzm = __ballot(act && v != 0);
        const unsigned rowm = (unsigned)((nzm >> (lane & 48)) & 0xFFFFULL);
        const int n_signs = __popc(rowm);
        const int my_bit = __popc(rowm >> ((lane & 15) + 1));
        const int sign_val = row_sum_i32((act && v < 0) ? (1 << my_bit) : 0);
        // where everything goes: per sub-block [sb_coded_flag] [pass-1 bins] [remainders] [signs]
        int C1, C2;
        const int e1 = row_excl_sum(c1, &C1), e2 = row_excl_sum(c2, &C2);
        const int hf = hasflag ? 1 : 0;
        const int row_size = hf + C1 + C2 + (n_signs ? 1 : 0);
        const int s0 = __builtin_amdgcn_readlane(row_size, 0), s1 = __builtin_amdgcn_readlane(row_size, 16),
                  s2 = __builtin_amdgcn_readlane(row_size, 32), s3 = __builtin_amdgcn_readlane(row_size, 48);
        const int row_base = row == 0 ? 0 : (row == 1 ? s0 : (row == 2 ? s0 + s1 : s0 + s1 + s2));
        const int total = s0 + s1 + s2 + s3;
        tok_reserve(o, total);
        const int base = o.n_tok + row_base;
        if (hasflag && (lane & 15) == 0) {
            // sb_coded_flag: context from the right and lower sub-blocks (bool_coder.rs:2102-2150)
            int csbf = 0;
            if (xs < sbw - 1) csbf |= (int)((coded >> (ys * 8 + xs + 1)) & 1ULL);
            if (ys < sbw - 1) csbf |= (int)((coded >> ((ys + 1) * 8 + xs)) & 1ULL);
            tok_put(o, base, tok_ctx(sb_ctx + csbf, real));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < c1) tok_put(o, base + hf + e1 + j, t1[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j < c2) tok_put(o, base + hf + C1 + e2 + j, t2[j]);
        if (n_signs && (lane & 15) == 15) tok_put(o, base + hf + C1 + C2, tok_bypass((uint32_t)sign_val, n_signs));
        o.n_tok += total;
        q_carry = q_next;
        if (c == 0 && __ballot(act && (xs > 3 || ys > 3)) != 0ULL) flags |= 1u << 29; // MtsZeroOutSigCoeffFlag = 0 (:2008-2010)
        WSYNC();
    }
    return 0x80000000u | flags | (uint32_t)(o.n_tok - start);
}

// residual_coding of up to FOUR 4x4 transform blocks in one step: block r = lanes 16 r .. 16 r + 15, one lane per coefficient
// (a 4x4 block is one sub-block, and tb_tokens above spends a whole 64-lane step on its 16 coefficients: the four luma
// blocks of a split 8x8 CU and its chroma pair were six steps, an unsplit 8x8 CU's chroma pair two).  Same tokens in the
// same order as tb_tokens block after block: everything that runs along the scan -- the dependent-quantisation state, the
// bin budget (28 per block), the first coded position, the token offsets -- is a prefix INSIDE a row of 16 lanes here.
// comp / lev / stride of block r: cs[r], l[r], st[r].  own_headers: every block's header word goes right in front of its
// tokens (the four luma transform units of a split CU); otherwise the caller has made room for the headers and gets
// them back in hdr[] (the chroma pair of one transform unit).
__device__ __forceinline__ void tb4_tokens(const CONST_AS DevConst* k, TokOut& o, int ntu, const int cs[4], const GLOBAL_AS int16_t* const l[4],
                                           const int st[4], bool own_headers, uint32_t hdr[4]) {
    const int lane = lane_fresh();
    const int row = lane >> 4, i16 = lane & 15;
    const bool mine = row < ntu;
    const CONST_AS uint16_t* scan = k->scan_idx[0];
    const int r = scan[i16];
    const int xc = r & 3, yc = r >> 2;
    const int c = row == 0 ? cs[0] : (row == 1 ? cs[1] : (row == 2 ? cs[2] : cs[3]));
    const GLOBAL_AS int16_t* lp = row == 0 ? l[0] : (row == 1 ? l[1] : (row == 2 ? l[2] : l[3]));
    const int stride = row == 0 ? st[0] : (row == 1 ? st[1] : (row == 2 ? st[2] : st[3]));
    const int lvl = mine ? (int)lp[(size_t)yc * stride + xc] : 0;
    hdr[0] = hdr[1] = hdr[2] = hdr[3] = 0u;
    const unsigned long long nzb = __ballot(lvl != 0);
    if (nzb == 0ULL && !own_headers) return;
    // neighbourhood arrays: block r at 40 r, 6 entries a row (two zero columns / rows behind the block)
    for (int i = lane; i < 80; i += 64) ((uint32_t*)TK.tpl)[i] = 0u;
    if (lane < 40) ((uint32_t*)TK.ab)[lane] = 0u;
    WSYNC();
    const unsigned rowm = (unsigned)((nzb >> (lane & 48)) & 0xFFFFULL);
    const bool has = rowm != 0u;                       // the block has a level
    const int p_last = has ? (int)__builtin_ctz(rowm) : 16;
    const bool valid = mine && i16 >= p_last;
    const int v = valid ? lvl : 0;
    const int av = abs(v);
    // last_sig_coeff_{x,y}_prefix of the block (log2 size 2: no suffixes, one context per bin)
    const int rl = scan[has ? p_last : 0];
    const int px = rl & 3, py = rl >> 2;
    const int nx = px + (px < 3 ? 1 : 0), ny = py + (py < 3 ? 1 : 0);
    const int last_n = nx + ny;
    const int loff = c == 0 ? 0 : 20;
    // dependent-quantisation state in front of every coefficient: prefix composition inside the row, from state 0
    int pre = valid ? position_map(v, av, false, 0) : kMapId;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    const int before = __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false);      // row_shr:1 (lane 0 of a row: identity)
    const int q_in = before & 3;
    const int a = v ? (av + (q_in > 1 ? 1 : 0)) >> 1 : 0;   // AbsLevel (:1968-1985)
    const bool sig = a > 0, gt1 = a > 1, gt3 = a > 3;
    const bool is_last = i16 == p_last;
    const bool act = valid;
    const bool sigcoded = act && !is_last;
    const int nb1 = act ? (sigcoded ? 1 : 0) + (sig ? (gt1 ? 3 : 1) : 0) : 0;
    int used;
    const int ex = row_excl_sum(nb1, &used);
    const bool covered = act && 28 - ex >= 4;               // pass 1 reaches the coefficient (:2014-2016)
    const int p1 = covered && sig ? (gt1 ? 2 + (a & 1) + (gt3 ? 2 : 0) : 1) : 0;
    const int cell = 40 * row + 6 * yc + xc;
    if (act) {
        TK.tpl[cell] = (uint16_t)(p1 ? (256 | p1) : 0);
        TK.ab[cell] = (uint8_t)min(a, 255);
    }
    WSYNC();
    int c1 = 0, c2 = 0;
    uint32_t t1[4] = {0, 0, 0, 0}, t2[2] = {0, 0};
    if (act) {
        const uint16_t* tp = &TK.tpl[cell];
        const uint8_t* ap = &TK.ab[cell];
        const int tsum = tp[1] + tp[2] + tp[6] + tp[7] + tp[12];
        const int asum = ap[1] + ap[2] + ap[6] + ap[7] + ap[12];
        const int d = xc + yc;
        if (covered) {
            const int sum_p1 = tsum & 255, num_sig = tsum >> 8;
            if (sigcoded) {
                const int s_ = (sum_p1 + 1) >> 1;
                const int qs = q_in > 1 ? q_in - 1 : 0;
                const int inc = c == 0 ? 12 * qs + min(s_, 3) + (d < 2 ? 8 : (d < 5 ? 4 : 0)) : 36 + 8 * qs + min(s_, 3) + (d < 2 ? 4 : 0);
                t1[c1++] = tok_ctx(CTXD_SIG + inc, sig);
            }
            if (sig) {
                const int off = min(sum_p1 - num_sig, 4);
                int inc;
                if (is_last)
                    inc = c == 0 ? 0 : 21;
                else if (c == 0)
                    inc = 1 + off + (d == 0 ? 15 : (d < 3 ? 10 : (d < 10 ? 5 : 0)));
                else
                    inc = 22 + off + (d == 0 ? 5 : 0);
                t1[c1++] = tok_ctx(CTXD_GTX + inc, gt1);
                if (gt1) {
                    t1[c1++] = tok_ctx(CTXD_PAR + inc, a & 1);
                    t1[c1++] = tok_ctx(CTXD_GTX + 32 + inc, gt3);
                }
            }
            if (gt3) c2 = remainder_tokens((a - 4) >> 1, rice_of(asum - 20), &t2[0], &t2[1]);   // abs_remainder (pass 2)
        } else {
            const int kr = rice_of(asum);                                                        // dec_abs_level (pass 3)
            const int zero_pos = (q_in < 2 ? 1 : 2) << kr;
            c2 = remainder_tokens(a == 0 ? zero_pos : (a <= zero_pos ? a - 1 : a), kr, &t2[0], &t2[1]);
        }
    }
    // signs of the block in coding order, first one in the most significant bit
    const unsigned long long nzm = __ballot(act && v != 0);
    const unsigned srow = (unsigned)((nzm >> (lane & 48)) & 0xFFFFULL);
    const int n_signs = __popc(srow);
    const int my_bit = __popc(srow >> (i16 + 1));
    const int sign_val = row_sum_i32((act && v < 0) ? (1 << my_bit) : 0);
    // where everything goes: per block [header, if its own] [last position] [pass-1 bins] [remainders] [signs]
    int C1, C2;
    const int e1 = row_excl_sum(c1, &C1), e2 = row_excl_sum(c2, &C2);
    const int n_tok = has ? last_n + C1 + C2 + (n_signs ? 1 : 0) : 0;
    const int hd = own_headers ? 1 : 0;
    const int row_size = mine ? hd + n_tok : 0;
    const int s0 = __builtin_amdgcn_readlane(row_size, 0), s1 = __builtin_amdgcn_readlane(row_size, 16),
              s2 = __builtin_amdgcn_readlane(row_size, 32), s3 = __builtin_amdgcn_readlane(row_size, 48);
    const int row_base = row == 0 ? 0 : (row == 1 ? s0 : (row == 2 ? s0 + s1 : s0 + s1 + s2));
    const int total = s0 + s1 + s2 + s3;
    tok_reserve(o, total);
    const uint32_t h = has ? (0x80000000u | ((c == 0 && p_last != 15) ? 1u << 30 : 0u) | (uint32_t)n_tok) : 0u; // (bit 30: MtsDcOnly = 0, :1945-1947)
    const int base = o.n_tok + row_base + hd;
    if (own_headers && mine && i16 == 0) tok_put(o, base - 1, h);
    if (has) {
        if (i16 < nx)
            tok_put(o, base + i16, tok_ctx(CTXD_LAST_X + i16 + loff, i16 < px));
        else if (i16 < last_n)
            tok_put(o, base + i16, tok_ctx(CTXD_LAST_Y + (i16 - nx) + loff, (i16 - nx) < py));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < c1) tok_put(o, base + last_n + e1 + j, t1[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j < c2) tok_put(o, base + last_n + C1 + e2 + j, t2[j]);
        if (n_signs && i16 == 15) tok_put(o, base + last_n + C1 + C2, tok_bypass((uint32_t)sign_val, n_signs));
    }
    o.n_tok += total;
    hdr[0] = (uint32_t)__builtin_amdgcn_readlane((int)h, 0);
    hdr[1] = (uint32_t)__builtin_amdgcn_readlane((int)h, 16);
    hdr[2] = (uint32_t)__builtin_amdgcn_readlane((int)h, 32);
    hdr[3] = (uint32_t)__builtin_amdgcn_readlane((int)h, 48);
    WSYNC();
}

// One wavefront per CTU: its transform units in coding order (coding_tree, ctu_encoder.rs:227-438), per unit the header
// words of its components, then their tokens.
__global__ __launch_bounds__(256) void residual_tokens_kernel(const DevConst* __restrict__ kc, const PicBufs* __restrict__ slots,
                                                              int first_slot, int n_pictures, uint32_t* pool, unsigned pool_pages, int pool_mask,
                                                              unsigned* page_counter, uint32_t* first_page, int* overflow) {
    const CONST_AS DevConst* k = (const CONST_AS DevConst*)kc;
    const int ctus = k->ctu_cols * k->ctu_rows;
    const int g = (int)blockIdx.x * 4 + uni((int)(threadIdx.x >> 6));
    if (g >= n_pictures * ctus) return;
    const int pic = g / ctus, ctu = g - pic * ctus;
    const int cy = ctu / k->ctu_cols, cx = ctu - cy * k->ctu_cols;
    const PicBufs pb = slots[first_slot + pic];
    const int W = k->W, Wc = W >> 1;
    TokOut o;
    o.pool = (GLOBAL_AS uint32_t*)pool;
    o.page_counter = page_counter + (g & pool_mask) * kTokCounterStride;
    o.pool_first = (unsigned)(g & pool_mask) * pool_pages;
    o.pool_pages = pool_pages;
    o.n_tok = 0;
    o.n_pages = 0;
    o.overflow = overflow;
    o.dead = false;
    const GLOBAL_AS uint8_t* cul = AS_GLOBAL(const uint8_t, pb.cu_log2);
    const GLOBAL_AS int16_t* ly = AS_GLOBAL(const int16_t, pb.lev[0]);
    const GLOBAL_AS int16_t* lcb = AS_GLOBAL(const int16_t, pb.lev[1]);
    const GLOBAL_AS int16_t* lcr = AS_GLOBAL(const int16_t, pb.lev[2]);
    int z = 0;
    while (z < 64) { // 4x4 units of the CTU in z-order
        const int ux = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), uy = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
        const int x0 = cx * 32 + 4 * ux, y0 = cy * 32 + 4 * uy;
        const int lg = uni((int)cul[(size_t)(y0 >> 2) * (W >> 2) + (x0 >> 2)]);
        if (lg >= 3) {
            // a single-tree CU = one transform unit: luma, Cb, Cr
            tok_reserve(o, 3);
            const int h = o.n_tok;
            o.n_tok += 3;
            const uint32_t hy = tb_tokens(k, o, 0, lg, ly + (size_t)y0 * W + x0, W);
            uint32_t hb, hr;
            if (WRENC_TOKENS_4X4_ROWS && lg == 3) { // the 4x4 chroma pair in one step
                const int cs[4] = {1, 2, 0, 0}, st[4] = {Wc, Wc, 0, 0};
                const GLOBAL_AS int16_t* const l[4] = {lcb + (size_t)(y0 >> 1) * Wc + (x0 >> 1), lcr + (size_t)(y0 >> 1) * Wc + (x0 >> 1), nullptr, nullptr};
                uint32_t hd4[4];
                tb4_tokens(k, o, 2, cs, l, st, false, hd4);
                hb = hd4[0];
                hr = hd4[1];
            } else {
                hb = tb_tokens(k, o, 1, lg - 1, lcb + (size_t)(y0 >> 1) * Wc + (x0 >> 1), Wc);
                hr = tb_tokens(k, o, 2, lg - 1, lcr + (size_t)(y0 >> 1) * Wc + (x0 >> 1), Wc);
            }
            if (LANE == 0) {
                tok_put(o, h, hy);
                tok_put(o, h + 1, hb);
                tok_put(o, h + 2, hr);
            }
            z += 1 << (2 * (lg - 2));
        } else {
            // an 8x8 CU split into four 4x4 luma CUs (one transform unit each), then the chroma CU of the 8x8
            if (WRENC_TOKENS_4X4_ROWS) {
                uint32_t hd4[4];
                {
                    const int cs[4] = {0, 0, 0, 0}, st[4] = {W, W, W, W};
                    const GLOBAL_AS int16_t* b = ly + (size_t)y0 * W + x0;
                    const GLOBAL_AS int16_t* const l[4] = {b, b + 4, b + (size_t)4 * W, b + (size_t)4 * W + 4};
                    tb4_tokens(k, o, 4, cs, l, st, true, hd4);
                }
                tok_reserve(o, 2);
                const int h = o.n_tok;
                o.n_tok += 2;
                const int cs[4] = {1, 2, 0, 0}, st[4] = {Wc, Wc, 0, 0};
                const GLOBAL_AS int16_t* const l[4] = {lcb + (size_t)(y0 >> 1) * Wc + (x0 >> 1), lcr + (size_t)(y0 >> 1) * Wc + (x0 >> 1), nullptr, nullptr};
                tb4_tokens(k, o, 2, cs, l, st, false, hd4);
                if (LANE == 0) {
                    tok_put(o, h, hd4[0]);
                    tok_put(o, h + 1, hd4[1]);
                }
                z += 4;
                continue;
            }
            for (int i = 0; i < 4; ++i) {
                tok_reserve(o, 1);
                const int h = o.n_tok;
                o.n_tok += 1;
                const uint32_t hy = tb_tokens(k, o, 0, 2, ly + (size_t)(y0 + 4 * (i >> 1)) * W + x0 + 4 * (i & 1), W);
                if (LANE == 0) tok_put(o, h, hy);
            }
            tok_reserve(o, 2);
            const int h = o.n_tok;
            o.n_tok += 2;
            const uint32_t hb = tb_tokens(k, o, 1, 2, lcb + (size_t)(y0 >> 1) * Wc + (x0 >> 1), Wc);
            const uint32_t hr = tb_tokens(k, o, 2, 2, lcr + (size_t)(y0 >> 1) * Wc + (x0 >> 1), Wc);
            if (LANE == 0) {
                tok_put(o, h, hb);
                tok_put(o, h + 1, hr);
            }
            z += 4;
        }
    }
    if (LANE == 0) first_page[(size_t)pic * ctus + ctu] = (o.n_pages > 0 && !o.dead) ? TK.pt[0] : kTokNone;
}
#endif // WRENC_TOKENS_KERNEL_TU

} // namespace wrenc
