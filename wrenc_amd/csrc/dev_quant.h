// dev_quant.h -- dependent quantisation as a pooled 4-state Viterbi, forward trace + level cost, dequantisation (quantizer.rs)
// Part of the gfx950 device code of the RD-search path; see wrenc_dev.h for the overall model.
#pragma once

namespace wrenc {

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
__device__ __forceinline__ int walker_wave() { return (int)((blockIdx.x * 2654435761u) >> 30) & (WPB < 4 ? WPB - 1 : 3); }

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

// |(tc << sh) - off| / lsc, the quotient every decision of a position starts from (quantizer.rs:378, :441); 0 for a
// zero coefficient.  Recomputed where it is needed (a shift, one v_mul_hi_u32 by a 32-bit reciprocal, a shift) rather
// than kept per position: the 2 KB that array took are what lets a fifth workgroup fit the CU's LDS.
__device__ __forceinline__ int quotient(const CONST_AS DevConst* k, int tc, int sh, int off) {
    int S = (int)((unsigned)tc << sh) - off;
    if (tc < 0) S = -S;
    return tc == 0 ? 0 : (int)(__umulhi((unsigned)S, k->div_magic) >> k->div_shift);
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

// bits = 2 * bits + (kb < ka).  Path costs stay below 2^30 (above), so kb < ka is the sign of kb - ka, and
// v_alignbit_b32(bits, kb - ka, 31) shifts it in: two plain VALU instructions that the scheduler can put behind the
// step's v_and, where the DPP read of the next step needs two wait states anyway (the inline-asm v_cmp + v_addc pair
// this replaces was opaque to the hazard recogniser, which added an s_nop 1 to every step).
__device__ __forceinline__ unsigned shift_in_less(unsigned bits, int kb, int ka) {
    return __builtin_amdgcn_alignbit(bits, (unsigned)(kb - ka), 31);
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
// parity of this wave's pooled quantisation calls, kept in an unused cell of q_pm (reset in encode_ctu)
__device__ __forceinline__ int zero_flag_cell() {
    const int par = uni((int)SH.q_pm[0][0][3]) & 1;
    if (LANE == 0) SH.q_pm[0][0][3] = (uint16_t)(par ^ 1);
    return par;
}

// Zero blocks: when every coefficient of the call is zero the levels are zero and cost nothing (zeros behind the
// last non-zero level are free, block_splitter.rs:436-458), and r1 already holds them.  A wave in that case skips
// the trace; when NO wave of the workgroup has a non-zero coefficient (the usual case in flat areas) the pooled
// walk ends after the first barrier of its first chunk, where the waves see each other's flags.  *any_level tells
// the caller whether any level is non-zero (else dequantisation and the inverse transform are skipped too).
__device__ __forceinline__ long long quantize(Ctx c, int lg, int nb, bool shared, bool active, int* overflow, bool* any_level) {
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
    int16_t* tcs = (int16_t*)SH.r2;          // [blk][p]: coefficient in reverse-scan order (all of r2 for a 32x32 block)
    int32_t* cc = (int32_t*)SH.r1;           // chunk: [blk][CH][6] ints (coefficients are dead after the gather)
    const uint16_t* dec16 = (const uint16_t*)SH.decw; // decisions: [blk][sub-block][state] 16-bit masks
    PROF_MARK(q0_);
    int istar0 = P, istar1 = P;
    bool any_nz = false;
    if (active) {
        int first0 = P, first1 = P, nzl = 0;
        for (int idx = LANE; idx < nb * P; idx += 64) {
            const int blk = idx >> lgP, p = idx & (P - 1);
            const int tc = SH.r1[blk * P + scan[p]];
            nzl |= tc;
            const int qd = quotient(k, tc, sh, off);
            tcs[idx] = (int16_t)tc;
            if (tc != 0 && (qd >> 1) > 0) {
                if (blk)
                    first1 = min(first1, p);
                else
                    first0 = min(first0, p);
            }
        }
        istar0 = wave_min_i32(first0);
        if (nb == 2) istar1 = wave_min_i32(first1);
        any_nz = __ballot(nzl != 0) != 0ULL;
    }
    *any_level = false;
    if (!shared && !any_nz) return 0; // solo call on a zero block (or an inactive one): nothing to walk
    // "this wave has a non-zero coefficient", read by the whole workgroup below; two cells in turn (every wave makes
    // the same sequence of pooled calls), so that a wave already in its next call cannot overwrite a flag that a
    // slow wave has yet to read
    const int fcell = shared ? zero_flag_cell() : 0;
    if (LANE == 0) {
        SH.q_istar[fcell] = any_nz ? 1 : 0;
        SH.q_active = (active && any_nz) ? 1 : 0;
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
        if (active && any_nz) { // (a zero block writes no entries: r1 stays its all-zero levels)
            // per position and state class (0: state 0, 1: state 1, 2: states 2 and 3): (u, w) doubled,
            // see above; per sub-block: parity masks of the two delta classes and, for state 0, whether
            // its first position in coding order (kk == 15) keeps a zero inside the trailing run
            const bool mine = LANE < nb * CH;
            const int blk = LANE >= CH ? 1 : 0;
            const int i = LANE - blk * CH;
            const int p = base + i;
            int par0 = 0, par1 = 0, adj = 0;
            if (mine) {
                const int tc = tcs[blk * P + p];
                chunk_entry(c, cc + LANE * 6, tc, quotient(k, tc, sh, off), p == P - 1, p <= (blk ? istar1 : istar0),
                            sh, off, lsc, ldq1, &par0, &par1, &adj, &ovf);
            }
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
        if (shared && base == P - CH) {
            // first chunk: does ANY wave of the workgroup have a non-zero coefficient?  (lane w reads wave w's flag)
            const bool wg_nz = __ballot(LANE < WPB && SHW[LANE < WPB ? LANE : 0].q_istar[fcell] != 0) != 0ULL;
            if (!wg_nz) break; // every wave takes this exit: no barrier is left behind
        }
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
                    C = min(KA, KB) & ~1;
                    bits = shift_in_less(bits, KB, KA);
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        const bool choseB = KB < KA;
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
    if (!active || !any_nz) return 0;
    // ---- forward trace from state 0 (quantizer.rs:686-721) + level-cost walk ----
    // lanes are split evenly between the blocks; each lane owns `per` consecutive positions
    const int half = nb == 2 ? 32 : 64;
    const int blk = nb == 2 ? (LANE >> 5) : 0;
    const int lane_in = LANE & (half - 1);
    const int per = P >= half ? P / half : 1;
    const int p0 = lane_in * per;
    const bool act = p0 < P;
    const int16_t* btcs = tcs + blk * P;
    const uint16_t* bdec = dec16 + blk * (P >> 2);
    int fmap = kMapId;
    const DecMasks dm = dec_masks(bdec, act ? p0 : 0); // a lane's positions lie in one sub-block (per divides 16)
    if (act) {
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p];
            fmap = compose_map(position_map(tc, quotient(k, tc, sh, off), p == P - 1, dec_nib(dm, p)), fmap);
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
            const int tc = btcs[p];
            SH.r1[blk * P + scan[p]] = (int16_t)emit_level(c, tc, quotient(k, tc, sh, off), p == P - 1, dec_nib(dm, p), p, j,
                                                           state, zmask, sum_nz, fnz, ovf);
        }
    }
    const int pf = group_min_i32(fnz, half); // zeros before a block's first non-zero level cost nothing
    if (act) // zeros after the first non-zero position: positions j > pf - p0 of this lane
        sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
    *any_level = __ballot(pf < P) != 0ULL;
    const long long sum = wave_sum_i64(sum_nz);
    if (__ballot(ovf != 0) != 0ULL) *overflow = 1;
    WSYNC();
    PROF_MARK(q3_);
    PROF_ADD2(PH_QTRACE, q2_, q3_);
    return sum;
}

// ---------------------------------------------------------------------------
// The head of a chain, proven zero without walking it (round 4).
//
// Proof and CPU model: quantize_viterbi_sc in the CPU checker under oracle/ (tests/test_oracle.py runs it against the literal DFS,
// quantizer.rs:338-517, at QP 18..51).  In short: the forward trace starts in state 0 at p = 0 (the last scan position)
// and stays there while state 0 decides "zero", so above the first significant coefficient only state 0's decisions
// matter.  Per position p with a0(state 0) = 0 (quotient 0 or 1), in undoubled costs:
//     alpha_p = c1(delta 0) - c0tz        what state 0 -- and state 1, whose odd level is the only other way INTO state 0 --
//                                          pays more for level 1 than state 0 pays for its free zero
//     beta_p  = min over the branches of states 1..3 that do not lead to state 0 of (cost - c0tz)
// With G_p = min_{s != 0} V_p[s] - V_p[0] (V = exact cost-to-go):  state 0 decides zero at p  <=  G_{p+1} >= -alpha_p,  and
// G_p >= min(alpha_p, beta_p + G_{p+1}) + rebate_p.  So if alpha, beta >= 0 on the REGION [0, 16 sb) and the exact path
// costs at 16 sb (the walker has them after sub-block sb) satisfy G >= -min alpha, every level of the region is zero
// and the trace reaches 16 sb in state 0: the walk ends there.  The region is cut at the first position that cannot be
// part of it (quotient >= 2, alpha < 0, beta < 0, or the DC position with its own formula); on the bench content it
// covers 93 % of the sub-blocks of a 32x32 or 16x16 block and the test behind it does not fail once
// (tools/dq_shortcut_stats.py).  A failed test only means the walk goes on as before.
// ---------------------------------------------------------------------------
constexpr int kAlphaInf = 1 << 28;
struct HeadK {
    int ldq1, ldq2, ldq3;          // lambda_q * dq_table[1..3]
    int dp1, dp2, dp3, dn1, dn2, dn3; // dequantised levels +-1, +-2, +-3 at this block size: (q * lsc + off) >> sh
};
__device__ __forceinline__ HeadK head_consts(int sh, int off, int lsc) {
    HeadK h;
    h.ldq1 = uni(SHT.ldq[1]);
    h.ldq2 = uni(SHT.ldq[2]);
    h.ldq3 = uni(SHT.ldq[3]);
    h.dp1 = (lsc + off) >> sh;
    h.dp2 = (2 * lsc + off) >> sh;
    h.dp3 = (3 * lsc + off) >> sh;
    h.dn1 = (-lsc + off) >> sh;
    h.dn2 = (-2 * lsc + off) >> sh;
    h.dn3 = (-3 * lsc + off) >> sh;
    return h;
}
// alpha of one position and whether it ends the region (same arithmetic as chunk_entry for quotients 0 and 1:
// delta 0: a0 = 0, a1 = 1 (q1 = 2); delta 1: quotient 0: a0 = 0, a1 = 1 (q1 = 1), quotient 1: a0 = 1 (q0 = 1), a1 = 2 (q1 = 3);
// at the DC position (dcn; quantizer.rs:378-391) delta is not added to the quotient and delta 1's "zero" is the level -1
// of the usize wrap: a0 = 0 with q0 of the OPPOSITE sign, a1 = 1 (q1 = 1)).  The DC position can be part of the region like
// any other -- with nothing behind it G is 0 there, which is all the induction needs -- and then the region is the whole
// block: every level zero, nothing to walk (round 4, (W) in the CPU model).
__device__ __forceinline__ int head_alpha(int tc, int qd, bool dcn, const HeadK& h, bool* bad) {
    const bool neg = tc < 0;
    const int d1 = abs(tc - (neg ? h.dn1 : h.dp1));
    const int d2 = abs(tc - (neg ? h.dn2 : h.dp2));
    const int d3 = abs(tc - (neg ? h.dn3 : h.dp3));
    const int d1o = abs(tc - (neg ? h.dp1 : h.dn1)); // level 1 of the opposite sign (DC position, delta 1)
    const int c0tz = 128 * abs(tc); // state 0's zero inside the trailing run costs no bits (:449-453)
    const int c0d0 = c0tz + h.ldq1;
    const int c1d0 = 128 * d2 + h.ldq2;
    const int c0d1 = dcn ? 128 * d1o + h.ldq1 : (qd ? 128 * d1 + h.ldq2 : c0d0);
    const int c1d1 = (qd && !dcn) ? 128 * d3 + h.ldq3 : 128 * d1 + h.ldq2;
    int alpha = c1d0 - c0tz;
    int beta = min(min(c0d0, c0d1), c1d1) - c0tz;
    if (tc == 0) { // no second branch; the zero costs dq_table[1] outside the trailing run, nothing inside
        alpha = kAlphaInf;
        beta = h.ldq1;
    }
    *bad = qd >= 2 || alpha < 0 || beta < 0 || h.ldq1 < 0;
    return min(alpha, kAlphaInf);
}
// The same conditions as range tests (round 4, after (W)): `bad` and "quotient >= 2" depend on nothing but the coefficient and
// the block size, and each is false exactly on one interval around zero (the costs are piecewise linear in |tc|), which the
// host finds by running the formulas above over every 16-bit coefficient (DevConst::head_rng).  A coefficient outside the
// interval is taken as ending the region: if the true set were not an interval, that would only shorten regions, never
// admit a position the proof does not cover.  alpha needs no quotient either: c1(delta 0) is level 2's cost at both quotients.
struct HeadT {
    int tn, cnt, tnd, cntd, tnq, cntq;
};
__device__ __forceinline__ HeadT head_ranges(const CONST_AS DevConst* k, int lg) {
    const CONST_AS int32_t* r = k->head_rng[lg - 2];
    HeadT t;
    t.tn = r[0];
    t.cnt = r[1];
    t.tnd = r[2];
    t.cntd = r[3];
    t.tnq = r[4];
    t.cntq = r[5];
    return t;
}
__device__ __forceinline__ bool head_bad(int tc, bool dcn, const HeadT& t) {
    return (unsigned)(tc + (dcn ? t.tnd : t.tn)) >= (unsigned)(dcn ? t.cntd : t.cnt);
}
__device__ __forceinline__ bool head_sig(int tc, const HeadT& t) { return (unsigned)(tc + t.tnq) >= (unsigned)t.cntq; } // quotient >= 2
__device__ __forceinline__ int head_alpha1(int tc, const HeadK& h) {
    const int d2 = abs(tc - (tc < 0 ? h.dn2 : h.dp2));
    return tc == 0 ? kAlphaInf : min(128 * d2 + h.ldq2 - 128 * abs(tc), kAlphaInf);
}
// One batch of 64 positions (lane = position p0 + LANE of a chain of P) of the region search: `open` while no position has
// ended the region (still open behind the last batch: the whole block is zero, head_sb gives P / 16); arun = this lane's
// minimum of alpha over the region so far; sb = the sub-block the walk must reach
#ifndef WRENC_HEAD_EXIT
#define WRENC_HEAD_EXIT 1 // 0: every chain is walked to its end (round 3's behaviour; for A/B runs)
#endif
__device__ __forceinline__ void head_batch(int tc, int p, int P, bool valid, const HeadK& h, const HeadT& t, bool& open, int& arun, int& sb) {
    if (!WRENC_HEAD_EXIT) {
        sb = 0;
        open = false;
    }
    if (!open) return;
    const int alpha = head_alpha1(tc, h);
    const unsigned long long B = __ballot(valid && head_bad(tc, p == P - 1, t));
    if (B == 0ULL) {
        if (valid) arun = min(arun, alpha);
    } else {
        const int fb = (int)__builtin_ctzll(B);
        sb = (p - LANE + fb) >> 4; // (p - LANE = the batch's first position)
        if (valid && LANE < (fb & ~15)) arun = min(arun, alpha);
        open = false;
    }
}
// the sub-block the walk must reach, once every batch of the chain has been seen: nsb (= nothing to walk) if none ended the region
__device__ __forceinline__ int head_sb(bool open, int sb, int nsb) { return (open && WRENC_HEAD_EXIT) ? nsb : sb; }
// After the walk of sub-block sb (path costs C of the quad's four states, doubled): is G >= -alpha_min ?
__device__ __forceinline__ bool head_test(int C, int st, int amin) {
    const int c0 = dpp_quad<0x00>(C);                  // state 0's cost in every lane of the quad
    int m = st == 0 ? 0x7FFFFFFF : C;
    m = min(m, dpp_quad<0xB1>(m));
    m = min(m, dpp_quad<0x4E>(m));
    return (m - c0) + 2 * amin >= 0;                   // |m - c0| < 2^30 (see kNoBranch), amin <= 2^28
}

// quantize() for ONE wave on its own blocks (no pooling), with the head exit: same results.
__device__ __forceinline__ long long quantize_solo(Ctx c, int lg, int nb, int* overflow, bool* any_level) {
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
    int16_t* tcs = (int16_t*)SH.r2;          // [blk][p]: coefficient in reverse-scan order (all of r2 for a 32x32 block)
    int32_t* cc = (int32_t*)SH.r1;           // chunk: [blk][CH][6] ints (coefficients are dead after the gather)
    uint16_t* dec16 = (uint16_t*)SH.decw;    // decisions: [blk][sub-block][state] 16-bit masks
    PROF_MARK(q0_);
    *any_level = false;
#ifdef WRENC_EXP_SKIP_QUANT // instruction-count experiment only (profiles/r04_issue_model.md): every level zero, nothing computed
    for (int i = LANE; i < (nb * P) / 8; i += 64) *(uint4*)&SH.r1[8 * i] = make_uint4(0u, 0u, 0u, 0u);
    WSYNC();
    return 0;
#endif
    const HeadK hk = head_consts(sh, off, lsc);
    const HeadT ht = head_ranges(k, lg);
    const int nsb = P >> 4;
    int istar0 = P, istar1 = P, sbs0 = nsb - 1, sbs1 = nsb - 1, amin0 = kAlphaInf, amin1 = kAlphaInf;
    int nzl = 0;
#pragma unroll 1
    for (int blk = 0; blk < nb; ++blk) {
        int first = P, arun = kAlphaInf, sb = nsb - 1; // (first: uniform)
        bool open = true;
#pragma unroll 1
        for (int p0 = 0; p0 < P; p0 += 64) {
            const int p = p0 + LANE;
            const bool valid = p < P;
            const int tc = valid ? (int)SH.r1[blk * P + scan[valid ? p : 0]] : 0;
            nzl |= tc;
            if (valid) tcs[blk * P + p] = (int16_t)tc;
            if (first == P) {
                const unsigned long long sig = __ballot(head_sig(tc, ht));
                if (sig != 0ULL) first = p0 + (int)__builtin_ctzll(sig);
            }
            head_batch(tc, p, P, valid, hk, ht, open, arun, sb);
        }
        sb = head_sb(open, sb, nsb);
        if (sb > 0 && sb < nsb) arun = wave_min_i32(arun);
        if (blk) {
            istar1 = first;
            sbs1 = sb;
            amin1 = arun;
        } else {
            istar0 = first;
            sbs0 = sb;
            amin0 = arun;
        }
    }
    if (__ballot(nzl != 0) == 0ULL) return 0; // zero blocks: nothing to walk, the levels are the zeros already in r1
    WSYNC();
    if (sbs0 == nsb && (nb == 1 || sbs1 == nsb)) {
        // every level of the call is proven zero: nothing to walk, nothing to trace (the coefficients in r1 make way for them)
        for (int i = LANE; i < (nb * P) / 8; i += 64) *(uint4*)&SH.r1[8 * i] = make_uint4(0u, 0u, 0u, 0u);
        WSYNC();
        PROF_MARK(qz_);
        PROF_ADD2(PH_QPRE, q0_, qz_);
        return 0;
    }
    // decisions of the sub-blocks the walk may never reach: zero in every state (the trace then stays in state 0 there)
    if (LANE < sbs0) *(uint2*)(dec16 + LANE * 4) = make_uint2(0u, 0u);
    if (nb == 2 && LANE < sbs1) *(uint2*)(dec16 + (P >> 2) + LANE * 4) = make_uint2(0u, 0u);
    PROF_MARK(q1_);
    PROF_ADD2(PH_QPRE, q0_, q1_);
    const int ldq1 = hk.ldq1;
    const int st = LANE & 3;
    const int CH = min(P, nb == 2 ? 32 : 64); // chunk positions per block
    const int quad = LANE >> 2;
    const bool walker = quad < nb;
    const int wblk = walker ? quad : 0;
    const int32_t* wcc = (const int32_t*)SH.r1 + wblk * CH * 6;
    const int cls = st == 0 ? 0 : (st == 1 ? 1 : 2);
    uint16_t* wdec = dec16 + wblk * (P >> 2);
    int wsb = wblk ? sbs1 : sbs0;            // the walker's block: walk down to this sub-block, then test
    const int wamin = wblk ? amin1 : amin0;
    int lowest = nb == 2 ? min(sbs0, sbs1) : sbs0;
    int C = 0;
    int ovf = 0;
    for (int base = P - CH; base >= 0 && base + CH > 16 * lowest; base -= CH) {
        PROF_MARK(qb0_);
        WSYNC();
        {
            const bool mine = LANE < nb * CH;
            const int blk = LANE >= CH ? 1 : 0;
            const int i = LANE - blk * CH;
            const int p = base + i;
            int par0 = 0, par1 = 0, adj = 0;
            if (mine) {
                const int tc = tcs[blk * P + p];
                chunk_entry(c, cc + LANE * 6, tc, quotient(k, tc, sh, off), p == P - 1, p <= (blk ? istar1 : istar0),
                            sh, off, lsc, ldq1, &par0, &par1, &adj, &ovf);
            }
            const unsigned long long b0 = __ballot(mine && par0), b1 = __ballot(mine && par1), ba = __ballot(mine && adj);
            if (mine && (LANE & 15) == 0) {
                uint16_t* pm = SH.q_pm[blk][i >> 4];
                pm[0] = (uint16_t)(b0 >> LANE);
                pm[1] = (uint16_t)(b1 >> LANE);
                pm[2] = (uint16_t)((ba >> (LANE + 15)) & 1);
            }
        }
        WSYNC();
        PROF_MARK(qb1_);
        if (walker) {
            for (int g16 = CH - 16; g16 >= 0; g16 -= 16) { // one 4x4 sub-block per iteration
                const int sb = (base + g16) >> 4;
                if (sb < lowest) break;                     // (uniform) every block's head is proven zero from here on
                const uint16_t* pm = SH.q_pm[wblk][g16 >> 4];
                const unsigned parmask = pm[st > 1 ? 1 : 0];
                const bool adj = st == 0 && pm[2] != 0;
                int2 cur[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) cur[kk] = *(const int2*)&wcc[(g16 + kk) * 6 + 2 * cls];
                unsigned bits = 0;
#pragma unroll
                for (int kk = 15; kk >= 0; --kk) {
                    const int2 e = cur[kk];
                    const int KA = e.x + dpp_quad<0xD8>(C); // C[trans[s][0]]: quad_perm [0,2,1,3]
                    const int KB = e.y + dpp_quad<0x72>(C); // C[trans[s][1]]: quad_perm [2,0,3,1]
                    C = min(KA, KB) & ~1;
                    bits = shift_in_less(bits, KB, KA);
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        const bool choseB = KB < KA;
                        const bool pick1 = choseB != (((parmask >> 15) & 1) != 0);
                        if (!pick1 && adj) C -= 2 * ldq1;
                    }
                }
                bits ^= parmask; // choseB -> pick1
                int m = min(C, dpp_quad<0xB1>(C));  // renormalise (see quantize())
                m = min(m, dpp_quad<0x4E>(m));
                C -= m;
                wdec[sb * 4 + st] = (uint16_t)bits;
                // the end of this block's region: proven?  (A quad whose block is done walks on while the other block
                // needs it: what it writes are the true decisions of those sub-blocks, the trace is the same.)
                const bool proven = head_test(C, st, wamin);
                if (sb == wsb && sb > 0 && !proven) wsb = 0; // no: walk the whole chain
                lowest = __builtin_amdgcn_readlane(wsb, 0);
                if (nb == 2) lowest = min(lowest, __builtin_amdgcn_readlane(wsb, 4));
            }
        }
        lowest = __builtin_amdgcn_readlane(wsb, 0);
        if (nb == 2) lowest = min(lowest, __builtin_amdgcn_readlane(wsb, 4));
        PROF_MARK(qb3_);
        PROF_ADD2(PH_QB_PRE, qb0_, qb1_);
        PROF_ADD2(PH_QB_WALK, qb1_, qb3_);
    }
    WSYNC();
    PROF_MARK(q2_);
    PROF_ADD2(PH_QBACK, q1_, q2_);
    // ---- forward trace from state 0 (quantizer.rs:686-721) + level-cost walk, as in quantize(), from where each block's
    //      walk ended: the levels before that are zero (r1's chunk entries are dead: zero all levels first), the trace
    //      reaches that position in state 0, and the block's lanes share what is left ----
    const int start0 = 16 * __builtin_amdgcn_readlane(wsb, 0), start1 = nb == 2 ? 16 * __builtin_amdgcn_readlane(wsb, 4) : 0;
    for (int i = LANE; i < (nb * P) / 8; i += 64) *(uint4*)&SH.r1[8 * i] = make_uint4(0u, 0u, 0u, 0u);
    WSYNC();
    const int half = nb == 2 ? 32 : 64;
    const int blk = nb == 2 ? (LANE >> 5) : 0;
    const int lane_in = LANE & (half - 1);
    const int start = blk ? start1 : start0;
    int per = 1;
    while (per * half < P - start) per *= 2;   // (uniform per block; per divides 16: P - start <= 1024 = 16 * 64)
    const int p0 = start + lane_in * per;
    const bool act = p0 < P;
    const int16_t* btcs = tcs + blk * P;
    const uint16_t* bdec = dec16 + blk * (P >> 2);
    int fmap = kMapId;
    const DecMasks dm = dec_masks(bdec, act ? p0 : 0); // a lane's positions lie in one sub-block (per divides 16)
    if (act) {
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p];
            fmap = compose_map(position_map(tc, quotient(k, tc, sh, off), p == P - 1, dec_nib(dm, p)), fmap);
        }
    }
    int pre = fmap;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
    if (nb == 1) pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x143, 0xC, 0xF, false)); // row_bcast:31 -> rows 2, 3
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x138, 0xF, 0xF, false) & 3; // wave_shr:1
    if (lane_in == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = P;
    if (act) {
        int state = entry;
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p];
            SH.r1[blk * P + scan[p]] = (int16_t)emit_level(c, tc, quotient(k, tc, sh, off), p == P - 1, dec_nib(dm, p), p, j,
                                                           state, zmask, sum_nz, fnz, ovf);
        }
    }
    const int pf = group_min_i32(fnz, half); // zeros before a block's first non-zero level cost nothing
    if (act) // zeros after the first non-zero position: positions j > pf - p0 of this lane
        sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
    *any_level = __ballot(pf < P) != 0ULL;
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
// 64 luma + 16 + 16 chroma positions and the chroma blocks ride along for free.  The 3 * WPB walks take one quad
// of lanes each, all luma blocks first, then Cb, then Cr, over as many walker waves as that needs (WPB = 4: twelve
// quads of one wave).
// Scratch: r2 = [scan-order coefficients | chroma chunk entries], r1 = luma chunk entries (the coefficients are
// dead after the gather and every level is written at the end, as in quantize()), decw.
// shared == false: the wave walks its own three blocks in quads 0..2, no workgroup barrier (as in quantize()).
__device__ __forceinline__ void quantize3(Ctx c, int lg0, bool shared, bool active, int* overflow, long long* lvl_y,
                                          long long* lvl_c, bool* any_y, bool* any_c) {
    static_assert(WPB <= 8, "a walker wave has 16 quads; walker_wave() + 1 must stay below WPB");
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
    constexpr int kCcByte = 768;                  // 2 * T <= 768 for T <= 384
    int32_t* cc0 = (int32_t*)SH.r1;                     // chunk, luma: [64][6] ints
    int32_t* cc1 = (int32_t*)((char*)SH.r2 + kCcByte);  // chunk, Cb | Cr: [32][6] ints (ends at byte 1536 < kOrgLeaf)
    *lvl_y = 0;
    *lvl_c = 0;
    PROF_MARK(q0_);
    int istar0 = P0, istar1 = Pc, istar2 = Pc;
    bool any_nz = false;
    *any_y = false;
    *any_c = false;
    if (active) {
        int first0 = P0, first1 = Pc, first2 = Pc, nzl = 0;
        for (int idx = LANE; idx < T; idx += 64) {
            const int b = idx < P0 ? 0 : (idx < P0 + Pc ? 1 : 2);
            const int boff = b == 0 ? 0 : (b == 1 ? P0 : P0 + Pc);
            const int p = idx - boff;
            const int sh = b == 0 ? sh0 : shc;
            const int off = (1 << sh) >> 1;
            const int tc = SH.r1[boff + (b == 0 ? scan0[p] : scanc[p])];
            nzl |= tc;
            const int qd = quotient(k, tc, sh, off);
            tcs[idx] = (int16_t)tc;
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
        any_nz = __ballot(nzl != 0) != 0ULL;
    }
    if (!shared && !any_nz) return; // solo call on three zero blocks (or an inactive wave): nothing to walk
    const int fcell = shared ? zero_flag_cell() : 0;
    if (LANE == 0) {
        SH.q_istar[fcell] = any_nz ? 1 : 0;
        SH.q_active = (active && any_nz) ? 1 : 0;
    }
    PROF_MARK(q1_);
    PROF_ADD2(PH_QPRE, q0_, q1_);
    const int ldq1 = (int)ldq_at(c, 1);
    const int st = LANE & 3;
    const int cls = st == 0 ? 0 : (st == 1 ? 1 : 2);
    // walker quads: global quad gq = 16 * (walker wave) + LANE / 4 walks block kind gq / WPB of wave gq % WPB
    // (solo: quad gq walks block kind gq of this wave)
    constexpr int kWalkers = (3 * WPB + 15) / 16;
    const int wv = shared ? ((WAVE - walker_wave()) & (WPB - 1)) : 0; // 0 .. kWalkers - 1: the walker waves
    const int gq = 16 * wv + (LANE >> 2);
    const int wb = min(shared ? gq / WPB : gq, 2);
    const bool walker = shared ? (wv < kWalkers && gq < 3 * WPB) : (gq < 3);
    const Lds* tb = shared ? &SHW[gq % WPB] : &SH;
    const int32_t* wcc = wb == 0 ? (const int32_t*)tb->r1
                                 : (const int32_t*)((const char*)tb->r2 + kCcByte) + (wb == 1 ? 0 : 16) * 6;
    uint16_t* wdec = (uint16_t*)const_cast<uint32_t*>(tb->decw) + (wb == 0 ? 0 : (wb == 1 ? (P0 >> 2) : (P0 >> 2) + (Pc >> 2)));
    const int wnsb = wb == 0 ? 4 : 1; // sub-blocks of the walker's block per chunk
    int C = 0;
    int ovf = 0;
    const int nch = P0 >> 6;
    for (int ch = 0; ch < nch; ++ch) {
        const int base0 = P0 - 64 * (ch + 1), basec = Pc - 16 * (ch + 1);
        PROF_MARK(qb0_);
        WSYNC();
        if (active && any_nz) {
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
                    const int tc = tcs[gidx];
                    chunk_entry(c, b == 0 ? cc0 + e * 6 : cc1 + (e - 64) * 6, tc, quotient(k, tc, sh, (1 << sh) >> 1), p == Pb - 1,
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
        if (shared)
            __syncthreads();
        else
            WSYNC();
        PROF_MARK(qb2_);
        if (shared && ch == 0) { // zero blocks in every wave of the workgroup: see quantize()
            const bool wg_nz = __ballot(LANE < WPB && SHW[LANE < WPB ? LANE : 0].q_istar[fcell] != 0) != 0ULL;
            if (!wg_nz) break;
        }
        if (walker && (!shared || tb->q_active)) {
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
                    C = min(KA, KB) & ~1;
                    bits = shift_in_less(bits, KB, KA);
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        const bool choseB = KB < KA;
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
    if (!active || !any_nz) return;
    // ---- forward trace + level cost: lanes 0..31 luma, 32..47 Cb, 48..63 Cr ----
    const int b = LANE < 32 ? 0 : (LANE < 48 ? 1 : 2);
    const int lane_in = b == 0 ? LANE : (LANE & 15);
    const int Pb = b == 0 ? P0 : Pc;
    const int per = b == 0 ? (P0 >> 5) : (Pc >> 4); // P0 / 32 = Pc / 16 * 2
    const int boff = b == 0 ? 0 : (b == 1 ? P0 : P0 + Pc);
    const int p0 = lane_in * per;
    const int16_t* btcs = tcs + boff;
    const int shb = b == 0 ? sh0 : shc, offb = (1 << shb) >> 1;
    const uint16_t* bdec = (const uint16_t*)SH.decw + (b == 0 ? 0 : (b == 1 ? (P0 >> 2) : (P0 >> 2) + (Pc >> 2)));
    int fmap = kMapId;
    const DecMasks dm = dec_masks(bdec, p0); // a lane's positions lie in one sub-block (per divides 16)
    for (int j = 0; j < per; ++j) {
        const int p = p0 + j;
        const int tc = btcs[p];
        fmap = compose_map(position_map(tc, quotient(k, tc, shb, offb), p == Pb - 1, dec_nib(dm, p)), fmap);
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
            const int tc = btcs[p];
            SH.r1[boff + (b == 0 ? scan0[p] : scanc[p])] = (int16_t)emit_level(
                c, tc, quotient(k, tc, shb, offb), p == Pb - 1, dec_nib(dm, p), p, j, state, zmask, sum_nz, fnz, ovf);
        }
    }
    // zeros before a block's first non-zero level cost nothing: minimum per block (rows 0-1 | 2 | 3)
    {
        const int rm = row_min_i32(fnz);
        const int m0 = min(__builtin_amdgcn_readlane(rm, 0), __builtin_amdgcn_readlane(rm, 16));
        const int m1 = __builtin_amdgcn_readlane(rm, 32), m2 = __builtin_amdgcn_readlane(rm, 48);
        const int pf = b == 0 ? m0 : (b == 1 ? m1 : m2);
        sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
        *any_y = m0 < P0;
        *any_c = m1 < Pc || m2 < Pc;
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

// Dependent quantisation of nb <= 4 luma blocks of 4x4 at once (the candidates of a packed 4x4 leaf search,
// dev_search.h K_LEAF4): block b = lanes 16 b .. 16 b + 15, one lane per position, one quad walks each block's
// 16 positions (a single chunk: no barrier of any kind).  Same algorithm and arithmetic as quantize(); block b's
// level cost (block_splitter.rs:436-458) comes back in lvl[b], "has a non-zero level" in bit b of *any_mask.
// Coefficients r1[16 b ..] -> levels in place (a lane keeps its coefficient in a register).  Scratch: r1 chunk
// entries, decw.
__device__ __forceinline__ void quantize_p16(Ctx c, int nb, int* overflow, long long lvl[4], int* any_mask) {
    c = uni(c);
    nb = uni(nb);
    const int lane = lane_fresh();
    const CONST_AS DevConst* k = c.k;
    constexpr int P = 16, lg = 2;
    constexpr int sh = 8 + lg - 5 + 1; // quantizer.rs:558-569
    constexpr int off = (1 << sh) >> 1;
    const int lsc = k->lsc;
    const CONST_AS uint16_t* scan = k->scan_idx[0];
    int32_t* cc = (int32_t*)SH.r1;
    PROF_MARK(q0_);
    const int blk = lane >> 4, p = lane & 15;
    const bool mine = blk < nb;
#ifdef WRENC_EXP_SKIP_QUANT
    lvl[0] = lvl[1] = lvl[2] = lvl[3] = 0;
    *any_mask = 0;
    if (lane < 16 * nb) SH.r1[lane] = 0;
    WSYNC();
    return;
#endif
    const HeadT ht = head_ranges(k, 2);
    const int tc = mine ? (int)SH.r1[blk * P + scan[p]] : 0;
    const int istar = row_min_i32(head_sig(tc, ht) ? p : P); // of this lane's block
    const unsigned long long nzb = __ballot(tc != 0);
    lvl[0] = lvl[1] = lvl[2] = lvl[3] = 0;
    *any_mask = 0;
    if (nzb == 0ULL) return; // every block is zero: the levels are the zero coefficients already in r1
    if (WRENC_HEAD_EXIT) {
        // every block's levels proven zero (the head proof over the whole block, head_alpha): nothing to walk or trace
        if (__ballot(mine && head_bad(tc, p == P - 1, ht)) == 0ULL) {
            WSYNC();
            if (mine) SH.r1[blk * P + scan[p]] = 0;
            WSYNC();
            return;
        }
    }
    PROF_MARK(q1_);
    PROF_ADD2(PH_QPRE, q0_, q1_);
    WSYNC(); // every lane has its coefficient before the chunk entries overwrite r1
    const int qd = quotient(k, tc, sh, off); // (0 for a lane without a block)
    const int ldq1 = (int)ldq_at(c, 1);
    int ovf = 0;
    {
        int par0 = 0, par1 = 0, adj = 0;
        if (mine) chunk_entry(c, cc + lane * 6, tc, qd, p == P - 1, p <= istar, sh, off, lsc, ldq1, &par0, &par1, &adj, &ovf);
        const unsigned long long b0 = __ballot(mine && par0), b1 = __ballot(mine && par1), ba = __ballot(mine && adj);
        if (mine && p == 0) {
            uint16_t* pm = SH.q_pm[0][blk];
            pm[0] = (uint16_t)(b0 >> lane);
            pm[1] = (uint16_t)(b1 >> lane);
            pm[2] = (uint16_t)((ba >> (lane + 15)) & 1);
        }
    }
    WSYNC();
    PROF_MARK(qb1_);
    uint16_t* dec16 = (uint16_t*)SH.decw;
    {
        const int st = lane & 3, quad = lane >> 2;
        if (quad < nb) {
            const int cls = st == 0 ? 0 : (st == 1 ? 1 : 2);
            const int32_t* wcc = cc + quad * P * 6;
            const uint16_t* pm = SH.q_pm[0][quad];
            const unsigned parmask = pm[st > 1 ? 1 : 0];
            const bool adj = st == 0 && pm[2] != 0;
            int2 cur[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) cur[kk] = *(const int2*)&wcc[kk * 6 + 2 * cls];
            unsigned bits = 0;
            int C = 0;
#pragma unroll
            for (int kk = 15; kk >= 0; --kk) {
                const int2 e = cur[kk];
                const int KA = e.x + dpp_quad<0xD8>(C); // C[trans[s][0]]: quad_perm [0,2,1,3]
                const int KB = e.y + dpp_quad<0x72>(C); // C[trans[s][1]]: quad_perm [2,0,3,1]
                C = min(KA, KB) & ~1;
                bits = shift_in_less(bits, KB, KA);
                if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                    const bool choseB = KB < KA;
                    const bool pick1 = choseB != (((parmask >> 15) & 1) != 0);
                    if (!pick1 && adj) C -= 2 * ldq1;
                }
            }
            bits ^= parmask; // choseB -> pick1
            dec16[quad * 4 + st] = (uint16_t)bits;
        }
    }
    WSYNC();
    PROF_MARK(q2_);
    PROF_ADD2(PH_QBACK, q1_, q2_);
    PROF_ADD2(PH_QB_WALK, qb1_, q2_);
    // ---- forward trace from state 0 (quantizer.rs:686-721) + level cost, one position per lane ----
    const DecMasks dm = dec_masks(dec16 + blk * 4, 0);
    const int nib = dec_nib(dm, p);
    int pre = mine ? position_map(tc, qd, p == P - 1, nib) : kMapId;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xF, 0xF, false) & 3; // state after the previous lane of the row
    if (p == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = P;
    if (mine) {
        int state = entry;
        SH.r1[blk * P + scan[p]] = (int16_t)emit_level(c, tc, qd, p == P - 1, nib, p, 0, state, zmask, sum_nz, fnz, ovf);
    }
    const int pf = row_min_i32(fnz); // zeros before a block's first non-zero level cost nothing
    if (mine && (zmask & 1u) && p > pf) sum_nz += SHT.lv[0];
    {
        const long long hi = sum_nz >> 24;
        const int ra = row_sum_i32((int)(sum_nz & 0xFFFFFF)), rb = row_sum_i32((int)(hi & 0xFFFFFF)), rc = row_sum_i32((int)(hi >> 24));
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long long a0 = (long long)(unsigned)__builtin_amdgcn_readlane(ra, 16 * b);
            const long long a1 = (long long)(unsigned)__builtin_amdgcn_readlane(rb, 16 * b);
            const long long a2 = (long long)__builtin_amdgcn_readlane(rc, 16 * b);
            lvl[b] = a0 + ((a1 + (a2 << 24)) << 24);
            if (__builtin_amdgcn_readlane(pf, 16 * b) < P) *any_mask |= 1 << b;
        }
    }
    if (__ballot(ovf != 0) != 0ULL) *overflow = 1;
    WSYNC();
    PROF_MARK(q3_);
    PROF_ADD2(PH_QTRACE, q2_, q3_);
}

// ---------------------------------------------------------------------------
// Packed quantisation: the transform blocks of nc CANDIDATES of one single-tree CU at once (the packed leaf searches,
// dev_search.h K_LEAF8 / K_LEAF16).  LGL = log2 of the luma block: 3 (8x8 CU, nc <= 3) or 4 (16x16 CU, nc <= 2).
// Layout in r1: the candidates' luma blocks [c][PL], then their chroma blocks [c][Cb | Cr][PC], PC = PL / 4.
// Every block is a chain of its own and the chains do not depend on each other, so they are walked SIDE BY SIDE by
// this wave alone -- no pooling over the workgroup, no barrier.  A ROUND gives each of the wave's four 16-lane rows the
// next 16 positions (one 4x4 sub-block) of one chain: the 64 lanes compute the chunk entries of those 64 positions,
// then one quad per row walks them.  Rows 0 .. nc - 1 take the luma chains (DC end first); the other rows, and all
// rows once the luma chains are done, take the chroma chains one after the other, a chain staying in its row until
// it is finished.  The serial part of a pack of three 8x8 candidates is 5 x 16 steps for nine chains, of two 16x16
// candidates 16 x 16 steps for six chains -- where the one-candidate-per-request search walked the luma chain of each
// candidate between two workgroup barriers per 64 positions.  Same arithmetic as quantize(): chunk_entry, the
// one-compare walk, the forward trace by composed state maps, emit_level.  Levels in place; per candidate the level
// cost of the luma block and of the chroma pair (block_splitter.rs:436-458) and whether any level of the pack's luma /
// chroma blocks is non-zero.
// Scratch: r2[0, 3 PL nc) scan-order coefficients, r1 chunk entries, decw decisions (72 / 192 u16), q_pm.
// ---------------------------------------------------------------------------

// forward trace + level cost of up to four blocks of 64 positions, block b = row b (16 lanes, four consecutive
// positions per lane: they lie in one sub-block); levels to r1[rbase + 64 b + scan[p]]; per block the level cost and
// "has a non-zero level"
__device__ __forceinline__ void trace_rows64(const Ctx& c, int nblk, const int16_t* tcs, const uint16_t* dec16, int rbase,
                                             const CONST_AS uint16_t* scan, int sh, int off, long long lvl[4], bool any[4], int& ovf,
                                             int start = 0) {
    // start (per row: a multiple of 16): the positions before it are proven zero (their levels are zero already, the
    // trace reaches `start` in state 0); the row's lanes share what is left, 1, 2 or 4 consecutive positions each
    const int lane = lane_fresh();
    const int row = lane >> 4, i16 = lane & 15;
    const CONST_AS DevConst* k = c.k;
    const int per = start >= 48 ? 1 : (start >= 32 ? 2 : 4);
    const int p0 = start + i16 * per;
    const bool act = row < nblk && p0 < 64;
    const int b = row < nblk ? row : 0;
    const int16_t* btcs = tcs + b * 64;
    const DecMasks dm = dec_masks(dec16 + b * 16, act ? p0 : 0);
    int fmap = kMapId;
    if (act) {
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p];
            fmap = compose_map(position_map(tc, quotient(k, tc, sh, off), p == 63, dec_nib(dm, p)), fmap);
        }
    }
    int pre = fmap;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xF, 0xF, false) & 3; // state after the previous lane of the row
    if (i16 == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = 64;
    if (act) {
        int state = entry;
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = btcs[p];
            SH.r1[rbase + b * 64 + scan[p]] = (int16_t)emit_level(c, tc, quotient(k, tc, sh, off), p == 63, dec_nib(dm, p), p, j, state,
                                                                  zmask, sum_nz, fnz, ovf);
        }
    }
    const int pf = row_min_i32(fnz); // zeros before a block's first non-zero level cost nothing
    if (act) sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
    const long long hi = sum_nz >> 24;
    const int ra = row_sum_i32((int)(sum_nz & 0xFFFFFF)), rb = row_sum_i32((int)(hi & 0xFFFFFF)), rc = row_sum_i32((int)(hi >> 24));
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
        const long long a0 = (long long)(unsigned)__builtin_amdgcn_readlane(ra, 16 * bb);
        const long long a1 = (long long)(unsigned)__builtin_amdgcn_readlane(rb, 16 * bb);
        const long long a2 = (long long)__builtin_amdgcn_readlane(rc, 16 * bb);
        lvl[bb] = a0 + ((a1 + (a2 << 24)) << 24);
        any[bb] = __builtin_amdgcn_readlane(pf, 16 * bb) < 64;
    }
}

// the same for up to four blocks of 16 positions, one position per lane (as quantize_p16)
__device__ __forceinline__ void trace_rows16(const Ctx& c, int nblk, const int16_t* tcs, const uint16_t* dec16, int rbase,
                                             const CONST_AS uint16_t* scan, int sh, int off, long long lvl[4], bool any[4], int& ovf) {
    const int lane = lane_fresh();
    const int row = lane >> 4, i16 = lane & 15;
    const CONST_AS DevConst* k = c.k;
    const bool mine = row < nblk;
    const int bb = mine ? row : 0;
    const int tc = mine ? (int)tcs[bb * 16 + i16] : 0;
    const int qd = quotient(k, tc, sh, off);
    const DecMasks dm = dec_masks(dec16 + bb * 4, 0);
    const int nib = dec_nib(dm, i16);
    int pre = mine ? position_map(tc, qd, i16 == 15, nib) : kMapId;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xF, 0xF, false) & 3;
    if (i16 == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = 16;
    if (mine) {
        int state = entry;
        SH.r1[rbase + bb * 16 + scan[i16]] = (int16_t)emit_level(c, tc, qd, i16 == 15, nib, i16, 0, state, zmask, sum_nz, fnz, ovf);
    }
    const int pf = row_min_i32(fnz);
    if (mine && (zmask & 1u) && i16 > pf) sum_nz += SHT.lv[0];
    const long long hi = sum_nz >> 24;
    const int ra = row_sum_i32((int)(sum_nz & 0xFFFFFF)), rb = row_sum_i32((int)(hi & 0xFFFFFF)), rc = row_sum_i32((int)(hi >> 24));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const long long a0 = (long long)(unsigned)__builtin_amdgcn_readlane(ra, 16 * b);
        const long long a1 = (long long)(unsigned)__builtin_amdgcn_readlane(rb, 16 * b);
        const long long a2 = (long long)__builtin_amdgcn_readlane(rc, 16 * b);
        lvl[b] = a0 + ((a1 + (a2 << 24)) << 24);
        any[b] = __builtin_amdgcn_readlane(pf, 16 * b) < 16;
    }
}

// the same for ONE block of 256 positions over the whole wave (four consecutive positions per lane), as quantize()
__device__ __forceinline__ void trace_wave256(const Ctx& c, const int16_t* tcs, const uint16_t* dec16, int rbase,
                                              const CONST_AS uint16_t* scan, int sh, int off, long long* lvl, bool* any, int& ovf,
                                              int start = 0) {
    // start (uniform, a multiple of 16): see trace_rows64; the wave's lanes share the positions from there on
    const int lane = lane_fresh();
    const CONST_AS DevConst* k = c.k;
    const int per = start >= 192 ? 1 : (start >= 128 ? 2 : 4);
    const int p0 = start + lane * per;
    const bool act = p0 < 256;
    const DecMasks dm = dec_masks(dec16, act ? p0 : 0);
    int fmap = kMapId;
    if (act) {
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = tcs[p];
            fmap = compose_map(position_map(tc, quotient(k, tc, sh, off), p == 255, dec_nib(dm, p)), fmap);
        }
    }
    int pre = fmap;
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x111, 0xF, 0xF, false)); // row_shr:1
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x112, 0xF, 0xF, false)); // row_shr:2
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x114, 0xF, 0xF, false)); // row_shr:4
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x118, 0xF, 0xF, false)); // row_shr:8
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
    pre = compose_map(pre, __builtin_amdgcn_update_dpp(kMapId, pre, 0x143, 0xC, 0xF, false)); // row_bcast:31 -> rows 2, 3
    int entry = __builtin_amdgcn_update_dpp(0, pre, 0x138, 0xF, 0xF, false) & 3; // wave_shr:1
    if (lane == 0) entry = 0;
    long long sum_nz = 0;
    unsigned zmask = 0;
    int fnz = 256;
    if (act) {
        int state = entry;
        for (int j = 0; j < per; ++j) {
            const int p = p0 + j;
            const int tc = tcs[p];
            SH.r1[rbase + scan[p]] = (int16_t)emit_level(c, tc, quotient(k, tc, sh, off), p == 255, dec_nib(dm, p), p, j, state, zmask,
                                                         sum_nz, fnz, ovf);
        }
    }
    const int pf = wave_min_i32(fnz);
    if (act) sum_nz += (long long)__popc(zmask >> min(max(pf - p0 + 1, 0), 16)) * SHT.lv[0];
    *any = pf < 256;
    *lvl = wave_sum_i64(sum_nz);
}

template <int LGL>
__device__ __forceinline__ void quantize_pk(Ctx c, int nc, int* overflow, long long lvl_y[3], long long lvl_c[3], bool* any_y,
                                            bool* any_c) {
    static_assert(LGL == 3 || LGL == 4, "8x8 and 16x16 CUs");
    c = uni(c);
    nc = uni(nc);
    const int lane = lane_fresh();
    const CONST_AS DevConst* k = c.k;
    constexpr int PL = 1 << (2 * LGL), PC = PL / 4;       // positions of a luma / chroma block
    constexpr int SBL = PL / 16, SBC = PC / 16;           // their 4x4 sub-blocks
    constexpr int shl = 8 + LGL - 5 + 1, offl = (1 << shl) >> 1; // quantizer.rs:558-569
    constexpr int shc = shl - 1, offc = (1 << shc) >> 1;
    const int lsc = k->lsc;
    const CONST_AS uint16_t* scanl = k->scan_idx[LGL - 2];
    const CONST_AS uint16_t* scanc = k->scan_idx[LGL - 3];
    int16_t* tcs = (int16_t*)SH.r2;
    int32_t* cc = (int32_t*)SH.r1;
    uint16_t* dec16 = (uint16_t*)SH.decw;  // luma candidate c: [4 SBL c + 4 sb + state]; chroma chain b: [4 SBL nc + 4 SBC b + 4 sb + state]
    uint16_t* ist = &SH.q_pm[1][0][0];     // first significant position: [c] luma candidate c, [4 + b] chroma chain b
    const int row = lane >> 4, i16 = lane & 15;
    const int nL = PL * nc;
    const int nch = 2 * nc;                // chroma chains: 2 c + plane
    PROF_MARK(q0_);
    lvl_y[0] = lvl_y[1] = lvl_y[2] = 0;
    lvl_c[0] = lvl_c[1] = lvl_c[2] = 0;
    *any_y = false;
    *any_c = false;
#ifdef WRENC_EXP_SKIP_QUANT
    for (int i = lane; i < (nL + nch * PC) / 8; i += 64) *(uint4*)&SH.r1[8 * i] = make_uint4(0u, 0u, 0u, 0u);
    WSYNC();
    return;
#endif
    // ---- coefficients into scan order, the first significant position of every chain, and how far down its head is
    //      provably zero (head_batch): lane i of v_low / v_amin = chain i (luma candidate i; 4 + b: chroma chain b) ----
    int nzl = 0;
    int v_low = 0, v_amin = kAlphaInf;
    const HeadK hkl = head_consts(shl, offl, lsc);
    const HeadT htl = head_ranges(k, LGL), htc = head_ranges(k, LGL - 1);
#pragma unroll 1
    for (int cd = 0; cd < nc; ++cd) {
        int first = PL, arun = kAlphaInf, sb = SBL - 1; // (first: uniform)
        bool open = true;
#pragma unroll
        for (int p = lane; p < PL; p += 64) {
            const int tc = SH.r1[cd * PL + scanl[p]];
            nzl |= tc;
            tcs[cd * PL + p] = (int16_t)tc;
            if (first == PL) {
                const unsigned long long sig = __ballot(head_sig(tc, htl));
                if (sig != 0ULL) first = (p - lane) + (int)__builtin_ctzll(sig);
            }
            head_batch(tc, p, PL, true, hkl, htl, open, arun, sb);
        }
        sb = head_sb(open, sb, SBL);
        if (sb > 0 && sb < SBL) arun = wave_min_i32(arun);
        if (lane == 0) ist[cd] = (uint16_t)first;
        if (lane == cd) {
            v_low = sb;
            v_amin = arun;
        }
    }
    if constexpr (LGL == 3) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) { // (4x4 chroma blocks: one sub-block; all of it proven zero, or all of it walked)
            const int blk = 4 * ps + row;
            const bool mine = blk < nch;
            int tc = 0;
            if (mine) {
                tc = SH.r1[nL + blk * 16 + scanc[i16]];
                nzl |= tc;
                tcs[nL + blk * 16 + i16] = (int16_t)tc;
            }
            const int first = row_min_i32(head_sig(tc, htc) ? i16 : 16);
            if (mine && i16 == 0) ist[4 + blk] = (uint16_t)first;
            const unsigned long long B = __ballot(mine && head_bad(tc, i16 == 15, htc));
#pragma unroll
            for (int rw = 0; rw < 4; ++rw) // (lane 4 + block of v_low: 1 = nothing to walk)
                if (lane == 4 + 4 * ps + rw) v_low = (WRENC_HEAD_EXIT && ((B >> (16 * rw)) & 0xFFFFULL) == 0ULL) ? 1 : 0;
        }
    } else {
        const HeadK hkc = head_consts(shc, offc, lsc);
#pragma unroll 1
        for (int b = 0; b < nch; ++b) {
            const int tc = SH.r1[nL + b * PC + scanc[lane]];
            nzl |= tc;
            tcs[nL + b * PC + lane] = (int16_t)tc;
            const unsigned long long sig = __ballot(head_sig(tc, htc));
            const int first = sig != 0ULL ? (int)__builtin_ctzll(sig) : PC;
            int arun = kAlphaInf, sb = SBC - 1;
            bool open = true;
            head_batch(tc, lane, PC, true, hkc, htc, open, arun, sb);
            sb = head_sb(open, sb, SBC);
            if (sb > 0 && sb < SBC) arun = wave_min_i32(arun);
            if (lane == 0) ist[4 + b] = (uint16_t)first;
            if (lane == 4 + b) {
                v_low = sb;
                v_amin = arun;
            }
        }
    }
    if (__ballot(nzl != 0) == 0ULL) return; // every block of the pack is zero: the levels are the zeros already in r1
    {
        // every chain proven zero from end to end: no walk, no trace; the coefficients in r1 make way for the zero levels
        const bool todo = lane < 12 && (lane < nc || (lane >= 4 && lane < 4 + nch)) && v_low < (lane < 4 ? SBL : SBC);
        if (__ballot(todo) == 0ULL) {
            WSYNC();
            for (int i = lane; i < (nL + nch * PC) / 8; i += 64) *(uint4*)&SH.r1[8 * i] = make_uint4(0u, 0u, 0u, 0u);
            WSYNC();
            PROF_MARK(qz_);
            PROF_ADD2(PH_QPRE, q0_, qz_);
            return;
        }
    }
    // the decisions of sub-blocks the walk never reaches are zero in every state (the trace stays in state 0 there)
    {
        constexpr int kMaxNc = LGL == 3 ? 3 : 2;
        constexpr int kDecWords = (4 * SBL * kMaxNc + 4 * SBC * 2 * kMaxNc + 3) / 4; // 8-byte words of the largest pack
        static_assert(kDecWords <= 64 && 8 * kDecWords <= 512, "decw");
        if (lane < (4 * SBL * nc + 4 * SBC * nch + 3) / 4) *(uint2*)(dec16 + 4 * lane) = make_uint2(0u, 0u);
    }
    WSYNC();
    PROF_MARK(q1_);
    PROF_ADD2(PH_QPRE, q0_, q1_);
    const int ldq1 = (int)ldq_at(c, 1);
    const int st = lane & 3, quad = lane >> 2;
    const int cls = st == 0 ? 0 : (st == 1 ? 1 : 2);
    int C = 0;
    int ovf = 0;
    // The rows' chains (uniform).  A ROUND gives each row the next sub-block of its chain; a row whose chain is done
    // takes the next chain of the pack: the luma chains first, then the chroma chains.  rleft = sub-blocks still to
    // walk, rlow = the sub-block its walk stops at if the head test passes there (0: walk to the end, no test).
    int rid[4] = {-1, -1, -1, -1}, rleft[4] = {0, 0, 0, 0}, rlow[4] = {0, 0, 0, 0}, ramin[4] = {0, 0, 0, 0};
    int next_chain = 0;
    const int nct = nc + nch;
#pragma unroll 1
    for (;;) {
        bool fresh[4];
        bool any_row = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fresh[j] = false;
            if (rleft[j] == 0) {
                rid[j] = -1;
                while (next_chain < nct) {
                    const int id = next_chain < nc ? next_chain : 4 + (next_chain - nc); // lane of v_low / v_amin, index of ist
                    ++next_chain;
                    const int low = __builtin_amdgcn_readlane(v_low, id);
                    const int len = (id < 4 ? SBL : SBC) - low;
                    if (len <= 0) continue; // proven zero from end to end: nothing to walk
                    rid[j] = id;
                    rlow[j] = low;
                    ramin[j] = __builtin_amdgcn_readlane(v_amin, id);
                    rleft[j] = len;
                    fresh[j] = true;
                    break;
                }
            }
            any_row = any_row || rid[j] >= 0;
        }
        if (!any_row) break;
        // this lane's row: chain, sub-block of the round
        const int myid = row == 0 ? rid[0] : (row == 1 ? rid[1] : (row == 2 ? rid[2] : rid[3]));
        const int mysb = (row == 0 ? rlow[0] + rleft[0] : (row == 1 ? rlow[1] + rleft[1] : (row == 2 ? rlow[2] + rleft[2] : rlow[3] + rleft[3]))) - 1;
        {
            const bool mine = myid >= 0;
            const bool is_l = myid < 4;
            const int p = 16 * mysb + i16;
            const int sh = is_l ? shl : shc, off = is_l ? offl : offc;
            int par0 = 0, par1 = 0, adj = 0;
            if (mine) {
                const int tc = tcs[is_l ? myid * PL + p : nL + (myid - 4) * PC + p];
                const int first = ist[myid];
                chunk_entry(c, cc + lane * 6, tc, quotient(k, tc, sh, off), p == (is_l ? PL - 1 : PC - 1), p <= first, sh, off, lsc, ldq1,
                            &par0, &par1, &adj, &ovf);
            }
            const unsigned long long b0 = __ballot(mine && par0), b1 = __ballot(mine && par1), ba = __ballot(mine && adj);
            if (mine && i16 == 0) {
                uint16_t* pm = SH.q_pm[0][row];
                pm[0] = (uint16_t)(b0 >> lane);
                pm[1] = (uint16_t)(b1 >> lane);
                pm[2] = (uint16_t)((ba >> (lane + 15)) & 1);
            }
        }
        WSYNC();
        PROF_MARK(qb1_);
        bool unproven = false;
        if (quad < 4) {
            // (quad q walks row q: the same selects with the quad number)
            const int wid = quad == 0 ? rid[0] : (quad == 1 ? rid[1] : (quad == 2 ? rid[2] : rid[3]));
            const int wleft = quad == 0 ? rleft[0] : (quad == 1 ? rleft[1] : (quad == 2 ? rleft[2] : rleft[3]));
            const int wlow = quad == 0 ? rlow[0] : (quad == 1 ? rlow[1] : (quad == 2 ? rlow[2] : rlow[3]));
            const int wamin = quad == 0 ? ramin[0] : (quad == 1 ? ramin[1] : (quad == 2 ? ramin[2] : ramin[3]));
            const bool wfresh = quad == 0 ? fresh[0] : (quad == 1 ? fresh[1] : (quad == 2 ? fresh[2] : fresh[3]));
            if (wid >= 0) {
                if (wfresh) C = 0; // a new chain
                const int wsb = wlow + wleft - 1;
                const int32_t* wcc = cc + quad * 16 * 6;
                const uint16_t* pm = SH.q_pm[0][quad];
                const unsigned parmask = pm[st > 1 ? 1 : 0];
                const bool adj = st == 0 && pm[2] != 0;
                int2 cur[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) cur[kk] = *(const int2*)&wcc[kk * 6 + 2 * cls];
                unsigned bits = 0;
#pragma unroll
                for (int kk = 15; kk >= 0; --kk) {
                    const int2 e = cur[kk];
                    const int KA = e.x + dpp_quad<0xD8>(C); // C[trans[s][0]]: quad_perm [0,2,1,3]
                    const int KB = e.y + dpp_quad<0x72>(C); // C[trans[s][1]]: quad_perm [2,0,3,1]
                    C = min(KA, KB) & ~1;
                    bits = shift_in_less(bits, KB, KA);
                    if (kk == 15) { // first position of a sub-block in coding order (:512-514)
                        const bool choseB = KB < KA;
                        const bool pick1 = choseB != (((parmask >> 15) & 1) != 0);
                        if (!pick1 && adj) C -= 2 * ldq1;
                    }
                }
                bits ^= parmask; // choseB -> pick1
                int m = min(C, dpp_quad<0xB1>(C)); // renormalise (see quantize())
                m = min(m, dpp_quad<0x4E>(m));
                C -= m;
                dec16[(wid < 4 ? 4 * SBL * wid : 4 * SBL * nc + 4 * SBC * (wid - 4)) + 4 * wsb + st] = (uint16_t)bits;
                // the last sub-block of the chain's planned walk: is the head above it proven zero?
                const bool proven = head_test(C, st, wamin);
                unproven = wleft == 1 && wlow > 0 && !proven;
            }
        }
        const unsigned long long ub = __ballot(unproven);
        WSYNC();
        PROF_MARK(qb2_);
        PROF_ADD2(PH_QB_WALK, qb1_, qb2_);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rid[j] >= 0) {
                --rleft[j];
                if ((ub >> (4 * j)) & 1ULL) { // not proven: the chain walks on to its end
                    rleft[j] = rlow[j];
                    rlow[j] = 0;
                    if (lane == rid[j]) v_low = 0;
                }
            }
    }
    PROF_MARK(q2_);
    PROF_ADD2(PH_QBACK, q1_, q2_);
    // ---- forward trace from state 0 (quantizer.rs:686-721) + level cost, from where each chain's walk ended (v_low);
    //      the levels before that are zero: the chunk entries in r1 are dead, zero all of the pack's levels first ----
    for (int i = lane; i < (nL + nch * PC) / 8; i += 64) *(uint4*)&SH.r1[8 * i] = make_uint4(0u, 0u, 0u, 0u);
    WSYNC();
    if constexpr (LGL == 3) {
        long long l4[4];
        bool a4[4];
        const int s0_ = 16 * __builtin_amdgcn_readlane(v_low, 0), s1_ = 16 * __builtin_amdgcn_readlane(v_low, 1),
                  s2_ = 16 * __builtin_amdgcn_readlane(v_low, 2);
        trace_rows64(c, nc, tcs, dec16, 0, scanl, shl, offl, l4, a4, ovf, row == 0 ? s0_ : (row == 1 ? s1_ : (row == 2 ? s2_ : 0)));
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (b < nc) {
                lvl_y[b] = l4[b];
                if (a4[b]) *any_y = true;
            }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            if (4 * ps >= nch) continue; // (uniform)
            trace_rows16(c, min(4, nch - 4 * ps), tcs + nL + 64 * ps, dec16 + 4 * SBL * nc + 16 * ps, nL + 64 * ps, scanc, shc, offc, l4, a4, ovf);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (4 * ps + b < nch) {
                    lvl_c[(4 * ps + b) >> 1] += l4[b];
                    if (a4[b]) *any_c = true;
                }
        }
    } else {
#pragma unroll 1
        for (int cd = 0; cd < nc; ++cd) {
            long long l1;
            bool a1;
            trace_wave256(c, tcs + cd * PL, dec16 + 4 * SBL * cd, cd * PL, scanl, shl, offl, &l1, &a1, ovf,
                          16 * __builtin_amdgcn_readlane(v_low, cd));
            if (cd == 0)
                lvl_y[0] = l1;
            else
                lvl_y[1] = l1;
            if (a1) *any_y = true;
        }
        long long l4[4];
        bool a4[4];
        const int s4_ = 16 * __builtin_amdgcn_readlane(v_low, 4), s5_ = 16 * __builtin_amdgcn_readlane(v_low, 5),
                  s6_ = 16 * __builtin_amdgcn_readlane(v_low, 6), s7_ = 16 * __builtin_amdgcn_readlane(v_low, 7);
        trace_rows64(c, nch, tcs + nL, dec16 + 4 * SBL * nc, nL, scanc, shc, offc, l4, a4, ovf,
                     row == 0 ? s4_ : (row == 1 ? s5_ : (row == 2 ? s6_ : s7_))); // four 8x8 chroma blocks at most
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (b < nch) {
                lvl_c[b >> 1] += l4[b];
                if (a4[b]) *any_c = true;
            }
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

} // namespace wrenc
