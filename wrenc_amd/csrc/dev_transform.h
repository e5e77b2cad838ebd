// dev_transform.h -- forward and inverse DCT-2 4..32 (transformer.rs)
// Part of the gfx950 device code of the RD-search path; see wrenc_dev.h for the overall model.
#pragma once

namespace wrenc {

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

} // namespace wrenc
