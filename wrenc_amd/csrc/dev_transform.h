// dev_transform.h -- forward and inverse DCT-2 4..32 (transformer.rs)
// Part of the gfx950 device code of the RD-search path; see wrenc_dev.h for the overall model.
#pragma once

namespace wrenc {

// ---------------------------------------------------------------------------
// DCT-2 (transformer.rs).  Lane u = lane % N owns basis row T_N[u][.] in
// registers; G = 64/N lane groups walk the rows/columns; the other operand is
// read from LDS as a wave-broadcast.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc) { // v_dot2_i32_i16
    typedef short s2 __attribute__((ext_vector_type(2)));
    s2 va, vb;
    va.x = (short)(a & 0xFFFF);
    va.y = (short)(a >> 16);
    vb.x = (short)(b & 0xFFFF);
    vb.y = (short)(b >> 16);
    return __builtin_amdgcn_sdot2(va, vb, acc, false);
}

// forward: nb residual blocks in r1 ([blk][y][x] i16) -> coefficients in place, via r2;
// transformer.rs:2040-2378.  The i32 intermediate of all nb blocks goes through r2 at once when it fits its
// 2 KB, block after block otherwise (the two 16x16 blocks of a 32x32 CU's chroma pair).
// h: the intermediate's buffer of HBYTES bytes in LDS (r2 in the search; the micro-benchmark of the 32x32 v_dot2
// version brings its own, wrenc_gpu.hip).
template <int LG, int HBYTES>
__device__ void fwd_dct(Ctx c, int nb, int o1, LDS_AS int32_t* h) {
    constexpr int N = 1 << LG;
    constexpr int G = 64 / N;
    constexpr int HS = N + 1; // row stride of the intermediate
    const int u = LANE & (N - 1);
    const int g = LANE >> LG;
    uint32_t t[N / 2];
    {
        const CONST_AS uint32_t* src = (const CONST_AS uint32_t*)&c.k->dct[LG - 2][u][0];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) t[k] = src[k];
    }
    constexpr int kFit = HBYTES / (N * HS * (int)sizeof(int32_t)); // blocks whose intermediate fits the buffer
    static_assert(kFit >= 1, "one block's stage-1 output must fit the buffer");
    const int per = nb <= kFit ? nb : 1; // blocks per round
#pragma unroll 1
    for (int b0 = 0; b0 < nb; b0 += per) {
        // stage 1: H[u][y] = (sum_x T[u][x] r[y][x] + d) >> (LG-1)   (:2139-2209); rows of the round's blocks
WRENC_UNROLL(WRENC_U_DCT)
        for (int yy = g; yy < per * N; yy += G) {
            const uint32_t* row = (const uint32_t*)&SH.r1[o1 + (b0 * N + yy) * N];
            int acc = 0;
#pragma unroll
            for (int k = 0; k < N / 2; ++k) acc = dot2(row[k], t[k], acc);
            const int blk = yy >> LG, y = yy & (N - 1);
            h[blk * (N * HS) + u * HS + y] = (acc + (1 << (LG - 2))) >> (LG - 1);
        }
        WSYNC();
        // stage 2: C[v][x] = (sum_y T[v][y] H[x][y] + d) >> (LG+6)  (:2246-2316); lane v = u
WRENC_UNROLL(WRENC_U_DCT)
        for (int xx = g; xx < per * N; xx += G) {
            const int blk = xx >> LG, x = xx & (N - 1);
            const LDS_AS int32_t* col = &h[blk * (N * HS) + x * HS];
            int acc = 0;
#pragma unroll
            for (int k = 0; k < N / 2; ++k) {
                // |T| <= 90 and |H| <= 46410: 24-bit multiplies are exact (v_mad_i32_i24)
                acc += __mul24((int)(short)(t[k] & 0xFFFF), col[2 * k]);
                acc += __mul24((int)t[k] >> 16, col[2 * k + 1]);
            }
            SH.r1[o1 + (b0 + blk) * (N * N) + u * N + x] = (int16_t)((acc + (1 << (LG + 5))) >> (LG + 6));
        }
        WSYNC();
    }
}

// inverse: nb transposed dequantised blocks in r2 ([blk][x][i], i16) -> residuals r1 ([blk][y][x]).  The
// intermediate V takes the place of the levels in r1 (dead once dequantised) and the second stage runs in
// place: a row of V is read whole by the 2^LG lanes that then write that row, and no other lane touches it.
// transformer.rs:2380-2737
template <int LG>
__device__ void inv_dct(Ctx c, int nb, int o1) {
    constexpr int N = 1 << LG;
    constexpr int G = 64 / N;
    const int u = LANE & (N - 1);
    const int g = LANE >> LG;
    const int16_t* dqt = (const int16_t*)SH.r2;
    int16_t* vbuf = SH.r1 + o1;
    uint32_t t[N / 2]; // Tt[u][i] = T_N[i][u]
    {
        const CONST_AS uint32_t* src = (const CONST_AS uint32_t*)&c.k->dct_t[LG - 2][u][0];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) t[k] = src[k];
    }
    // stage 1 (vertical): V[y][x] = clamp16((sum_i T[i][y] d[i][x] + 64) >> 7); lane y = u
WRENC_UNROLL(WRENC_U_DCT)
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
WRENC_UNROLL(WRENC_U_DCT)
    for (int yy = g; yy < nb * N; yy += G) {
        const uint32_t* row = (const uint32_t*)&vbuf[yy * N];
        uint32_t rv[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) rv[k] = row[k];
        WSYNC(); // the whole row is in registers before any lane overwrites an element of it
        int acc = 0;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) acc = dot2(rv[k], t[k], acc);
        vbuf[yy * N + u] = (int16_t)((acc + 2048) >> 12);
    }
    WSYNC();
}

// ---------------------------------------------------------------------------
// MFMA EXPERIMENT (north_star: "MFMA tried only for the 32x32 separable int16 transform and kept only if
// rocprof shows a real win"): the forward 32x32 DCT-2 (transformer.rs:2040-2378) as five
// v_mfma_i32_32x32x32_i8.  Exact integer arithmetic: the basis fits signed bytes, the other operand is split
// into balanced base-256 digits (v = d0 + 256 d1 + 65536 d2, d0, d1 in [-128, 127]), one MFMA per digit,
// i32 accumulators recombined with shifts.
//   stage 1: Ht[y][u] = sum_x R[y][x] T[u][x]      A = digits of R (9 bits: 2 digits), B = T
//   stage 2: C[v][u]  = sum_y T[v][y] H[u][y]      A = T in accumulator k order, B = digits of Ht (17 bits: 3)
// The first stage's accumulators ARE the second stage's B operand (same lane, same k set), so the intermediate
// never goes through LDS.  wrenc_gpu_test_fwd_dct32 is its parity gate and micro-benchmark (tools/dct_bench.py)
// against the v_dot2 version (fwd_dct<5>); measurements and the decision to keep it in DESIGN.md.
// ---------------------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef short s2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_digit1(uint32_t p) { // per i16 half: (v + 128) >> 8, arithmetic
    s2_t v;
    v.x = (short)(p & 0xFFFF);
    v.y = (short)(p >> 16);
    v = (v + (short)128) >> 8;
    return (uint32_t)(unsigned short)v.x | ((uint32_t)(unsigned short)v.y << 16);
}
// bytes b of four dwords -> one dword (byte k from dword k)
__device__ __forceinline__ uint32_t gather_byte(uint32_t a, uint32_t b, uint32_t c, uint32_t d, int byte) {
    const uint32_t sel = 0x0C0C0400u + 0x0101u * (uint32_t)byte; // [b(lo src), b(hi src), 0, 0]
    const uint32_t ab = __builtin_amdgcn_perm(b, a, sel);
    const uint32_t cd = __builtin_amdgcn_perm(d, c, sel);
    return __builtin_amdgcn_perm(cd, ab, 0x05040100u);           // [ab.b0, ab.b1, cd.b0, cd.b1]
}

__device__ __forceinline__ void fwd_dct32_mfma(Ctx c, int o1) {
    const int r = LANE & 31, h = LANE >> 5;
    const v4i_t tA = *(const CONST_AS v4i_t*)&c.k->dct32_a[r][16 * h];
    const v4i_t tP = *(const CONST_AS v4i_t*)&c.k->dct32_p[r][h][0];
    // ---- stage 1 ----
    const uint4 q0 = *(const uint4*)&SH.r1[o1 + r * 32 + 16 * h];
    const uint4 q1 = *(const uint4*)&SH.r1[o1 + r * 32 + 16 * h + 8];
    const uint32_t p[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    v4i_t lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = (int)__builtin_amdgcn_perm(p[2 * i + 1], p[2 * i], 0x06040200u); // byte 0 of the four i16
        hi[i] = (int)__builtin_amdgcn_perm(pk_digit1(p[2 * i + 1]), pk_digit1(p[2 * i]), 0x06040200u);
    }
    // one accumulator set, digit after digit from the top: acc = (acc << 8) + digit product (16 live registers
    // instead of one set per digit: the search kernel has no registers to spare)
    v16i_t z = {};
    v16i_t acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, tA, z, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < 16; ++w) acc[w] = (acc[w] << 8) + (1 << 3); // (h + (1 << (LG - 2))) >> (LG - 1), LG = 5 (:2201-2209)
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, tA, acc, 0, 0, 0);
    int H[16]; // Ht[y(w, h)][u = r], y(w, h) = 8 (w / 4) + 4 h + w % 4
#pragma unroll
    for (int w = 0; w < 16; ++w) H[w] = acc[w] >> 4;
    // ---- stage 2: digits of H straight from the registers ----
    v4i_t d0, d1, d2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = (uint32_t)H[4 * i], b = (uint32_t)H[4 * i + 1], cc = (uint32_t)H[4 * i + 2], d = (uint32_t)H[4 * i + 3];
        d0[i] = (int)gather_byte(a, b, cc, d, 0);                                     // sext8(v)
        d1[i] = (int)gather_byte(a + 128u, b + 128u, cc + 128u, d + 128u, 1);         // byte 1 of v + 128
        d2[i] = (int)gather_byte(a + 32896u, b + 32896u, cc + 32896u, d + 32896u, 2); // byte 2 of v + 128 + 32768
    }
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tP, d2, z, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < 16; ++w) acc[w] <<= 8;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tP, d1, acc, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < 16; ++w) acc[w] = (acc[w] << 8) + (1 << 10); // (+ (1 << (LG + 5))) >> (LG + 6) (:2309-2316)
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tP, d0, acc, 0, 0, 0);
    WSYNC(); // every lane has read its residuals before the coefficients overwrite them
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = 8 * (w >> 2) + 4 * h + (w & 3);
        SH.r1[o1 + v * 32 + r] = (int16_t)(acc[w] >> 11);
    }
    WSYNC();
}

// The inverse 32x32 transform (transformer.rs:2380-2737) the same way, four MFMAs.  16-bit operands (the dequantised
// coefficients, the clipped intermediate) go in as two bytes each: the signed high byte and the low byte minus 128
// (v = 256 hi + (lo - 128) + 128), so every product sum is short by 128 * sum_i T[i][n], a per-output constant that
// comes back together with the rounding offset (DevConst::idct32_k1 / k2).
//   stage 1: Vt[x][y] = sum_i dT[x][i] T[i][y]      A = bytes of the transposed dequantised block (r2), B = T
//   stage 2: Rt[x][y] = sum_i T[i][x] V[y][i]       A = T in accumulator k order, B = bytes of V straight from the
//                                                   first stage's accumulators (same lane, same k set)
// Residuals to r1[o1 ..] ([y][x]); nothing goes through LDS in between.
__device__ __forceinline__ void inv_dct32_mfma(Ctx c, int o1) {
    const int r = LANE & 31, h = LANE >> 5;
    const v4i_t tB = *(const CONST_AS v4i_t*)&c.k->idct32_b[r][16 * h];
    const v4i_t tP = *(const CONST_AS v4i_t*)&c.k->idct32_p[r][h][0];
    const int k1 = c.k->idct32_k1[r];
    // ---- stage 1 ----
    const int16_t* dqt = (const int16_t*)SH.r2;
    const uint4 q0 = *(const uint4*)&dqt[r * 32 + 16 * h];
    const uint4 q1 = *(const uint4*)&dqt[r * 32 + 16 * h + 8];
    const uint32_t p[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    v4i_t lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = (int)(__builtin_amdgcn_perm(p[2 * i + 1], p[2 * i], 0x06040200u) ^ 0x80808080u); // low bytes - 128
        hi[i] = (int)__builtin_amdgcn_perm(p[2 * i + 1], p[2 * i], 0x07050301u);                 // high bytes, signed
    }
    v16i_t z = {};
    v16i_t acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, tB, z, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < 16; ++w) acc[w] = (acc[w] << 8) + k1; // + 128 S[y] + 64 (:2499-2543)
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, tB, acc, 0, 0, 0);
    int V[16]; // V[y = r][x(w, h)], x(w, h) = 8 (w / 4) + 4 h + w % 4
#pragma unroll
    for (int w = 0; w < 16; ++w) V[w] = min(max(acc[w] >> 7, -32768), 32767);
    // ---- stage 2: bytes of V straight from the registers ----
    v4i_t d0, d1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = (uint32_t)V[4 * i], b = (uint32_t)V[4 * i + 1], cc = (uint32_t)V[4 * i + 2], d = (uint32_t)V[4 * i + 3];
        d0[i] = (int)(gather_byte(a, b, cc, d, 0) ^ 0x80808080u);
        d1[i] = (int)gather_byte(a, b, cc, d, 1);
    }
    const v16i_t k2 = *(const CONST_AS v16i_t*)&c.k->idct32_k2[h][0];
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tP, d1, z, 0, 0, 0);
#pragma unroll
    for (int w = 0; w < 16; ++w) acc[w] = (acc[w] << 8) + k2[w]; // + 128 S[x] + 2048 (:2680-2737)
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tP, d0, acc, 0, 0, 0);
    // lane (y = r, h): residuals at x = 8 q + 4 h + 0..3, q = 0..3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint2 v;
        v.x = ((uint32_t)(acc[4 * q] >> 12) & 0xFFFFu) | ((uint32_t)(acc[4 * q + 1] >> 12) << 16);
        v.y = ((uint32_t)(acc[4 * q + 2] >> 12) & 0xFFFFu) | ((uint32_t)(acc[4 * q + 3] >> 12) << 16);
        *(uint2*)&SH.r1[o1 + r * 32 + 8 * q + 4 * h] = v;
    }
    WSYNC();
}

// o1: where the blocks start in r1 (i16 units, a multiple of 2)
__device__ __forceinline__ void fwd_dct_lg(Ctx c, int lg, int nb, int o1 = 0) {
#ifdef WRENC_EXP_SKIP_DCT // instruction-count experiment only: the residual stays where the coefficients should be
    return;
#endif
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    o1 = uni(o1);
    LDS_AS int32_t* h = (LDS_AS int32_t*)SH.r2;
    constexpr int HB = (int)sizeof(SH.r2);
    switch (lg) {
    case 2: fwd_dct<2, HB>(c, nb, o1, h); break;
    case 3: fwd_dct<3, HB>(c, nb, o1, h); break;
    case 4: fwd_dct<4, HB>(c, nb, o1, h); break;
    default:
        // 32x32 (always a single luma block): the i8-MFMA version, kept on measurement (3.6x in the
        // micro-benchmark, +5.6 % frames/s at max-split-depth 0, +0.9 % at depth 2); its intermediate stays in the
        // accumulators, which is what lets r2 be 2 KB.
        fwd_dct32_mfma(c, o1);
        break;
    }
}
__device__ __forceinline__ void inv_dct_lg(Ctx c, int lg, int nb, int o1 = 0) {
#ifdef WRENC_EXP_SKIP_DCT
    return;
#endif
    c = uni(c);
    lg = uni(lg);
    nb = uni(nb);
    o1 = uni(o1);
    switch (lg) {
    case 2: inv_dct<2>(c, nb, o1); break;
    case 3: inv_dct<3>(c, nb, o1); break;
    case 4: inv_dct<4>(c, nb, o1); break;
    default:
#ifdef WRENC_IDCT32_VDOT2
        inv_dct<5>(c, nb, o1); // the comparison arm of the micro-benchmark (tools/dct_bench.py)
#else
        inv_dct32_mfma(c, o1); // 32x32 is always a single luma block
#endif
        break;
    }
}

} // namespace wrenc
