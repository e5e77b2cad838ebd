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
// CTU of noise at QP 22 about a hundred.)
//
// Same bytes as the host-only writer: tests/test_gpu_tokens.py compares the two streams on searched pictures and on
// random records no search would emit (escape codes, exhausted bin budgets, every block size).
#pragma once

namespace wrenc {

constexpr int kTokPage = 64;               // words per page: kTokPage - 1 tokens + the link to the next page
constexpr int kTokPayload = kTokPage - 1;
constexpr int kTokMaxPages = 160;          // pages one CTU can need: < (1.75 + 2) * 1536 + 96 * 24 tokens
constexpr uint32_t kTokNone = 0xFFFFFFFFu;

// first context of each syntax element of residual_coding in the host's flat model array (host/cabac.h CtxBase)
constexpr int CTXD_LAST_X = 32, CTXD_LAST_Y = CTXD_LAST_X + 23, CTXD_SB_CODED = CTXD_LAST_Y + 23, CTXD_SIG = CTXD_SB_CODED + 7,
              CTXD_PAR = CTXD_SIG + 63, CTXD_GTX = CTXD_PAR + 33;

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
    unsigned* page_counter;
    unsigned pool_pages;
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

// One wavefront per CTU: its transform units in coding order (coding_tree, ctu_encoder.rs:227-438), per unit the header
// words of its components, then their tokens.
__global__ __launch_bounds__(256) void residual_tokens_kernel(const DevConst* __restrict__ kc, const PicBufs* __restrict__ slots,
                                                              int first_slot, int n_pictures, uint32_t* pool, unsigned pool_pages,
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
    o.page_counter = page_counter;
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
            const uint32_t hb = tb_tokens(k, o, 1, lg - 1, lcb + (size_t)(y0 >> 1) * Wc + (x0 >> 1), Wc);
            const uint32_t hr = tb_tokens(k, o, 2, lg - 1, lcr + (size_t)(y0 >> 1) * Wc + (x0 >> 1), Wc);
            if (LANE == 0) {
                tok_put(o, h, hy);
                tok_put(o, h + 1, hb);
                tok_put(o, h + 2, hr);
            }
            z += 1 << (2 * (lg - 2));
        } else {
            // an 8x8 CU split into four 4x4 luma CUs (one transform unit each), then the chroma CU of the 8x8
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
