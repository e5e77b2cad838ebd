// issue_bench.hip -- how fast a gfx950 CU issues scalar and vector instructions.
//
//   hipcc --offload-arch=gfx950 -O2 -o xbuild/issue_bench tools/issue_bench.hip && xbuild/issue_bench
//
// Question: is the scalar ALU ONE unit per CU shared by the four SIMDs (one instruction per cycle per CU, as on GCN), and
// how do waves that issue only SALU, only VALU, or both share a SIMD's issue slots?  The search kernel issues
// 563 k VALU + 314 k SALU + 69 k branches per CTU (round 3): whether its scalar stream competes with its vector stream
// decides what cutting either is worth.
//
// Every wave runs `iters` rounds of 256 instructions of its kind: dependent chains within an accumulator, four
// independent accumulators, no memory access, no LDS.  Workgroups of 8 waves (two per SIMD), `wpc` / 8 of them per CU;
// the kernel uses no LDS and few registers, so up to 32 waves per CU fit.  Rates are instructions per CU per
// shader-clock cycle; the clock is measured in the run itself (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define V4(op) op op op op
#define V16(op) V4(V4(op))
#define V64(op) V4(V16(op))

// kind: 0 VALU only, 1 SALU only, 2 both in every wave (1 : 1), 3 VALU-only waves and SALU-only waves side by side
// (odd / even workgroups), 4 SALU + s_cbranch (a compare and a never-taken branch per 4 SALU)
template <int KIND>
__global__ __launch_bounds__(512) void issue_kernel(unsigned long long* out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    s0 = __builtin_amdgcn_readfirstlane(s0);
    s1 = __builtin_amdgcn_readfirstlane(s1);
    s2 = __builtin_amdgcn_readfirstlane(s2);
    s3 = __builtin_amdgcn_readfirstlane(s3);
    // kind 3: waves 0..3 of a workgroup (one per SIMD) issue VALU only, waves 4..7 SALU only: every SIMD holds both kinds
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool valu_wave = KIND == 0 || KIND == 2 || (KIND == 3 && wave < 4);
    const bool salu_wave = KIND == 1 || KIND == 2 || KIND == 4 || (KIND == 3 && wave >= 4);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 2) {
            asm volatile(V64("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, %2\n s_add_u32 %5, %5, 1\n"
                             "v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, 1\n v_add_u32 %3, %3, %0\n s_add_u32 %7, %7, 1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
        } else if (KIND == 4) {
            asm volatile(V64("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_cmp_eq_u32 %3, 0x7fffffff\n"
                             "s_cbranch_scc1 0\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
        } else {
            if (valu_wave)
                asm volatile(V64("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            if (salu_wave)
                asm volatile(V64("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
        }
    }
    if (a0 + a1 + a2 + a3 + s0 + s1 + s2 + s3 == 0x12345678u) out[0] = 1; // keep the results alive
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[1] = __builtin_readcyclecounter() - c0;
        out[2] = wall_clock64() - r0;
    }
}

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <int KIND>
static int run(int cus, int wpc, int iters, unsigned long long* d_out, double* seconds, double* mhz = nullptr) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = cus * wpc / 8;
    hipLaunchKernelGGL(issue_kernel<KIND>, dim3(grid), dim3(512), 0, 0, d_out, 8); // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(issue_kernel<KIND>, dim3(grid), dim3(512), 0, 0, d_out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    *seconds = ms / 1e3;
    if (mhz) {
        unsigned long long h[3];
        CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        *mhz = 100.0 * (double)h[1] / (double)h[2]; // s_memrealtime counts at 100 MHz
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    unsigned long long* d_out;
    CHECK(hipMalloc((void**)&d_out, 32));
    const int iters = 2000;
    const double per_wave = 256.0 * iters; // instructions of ONE kind per wave (kinds 2 and 4: see below)
    printf("%s: %d CUs, clockRate %.0f MHz (reported)\n", p.gcnArchName, cus, p.clockRate / 1e3);
    // the shader clock under this load: s_memtime ticks per 100 MHz tick of s_memrealtime, in a VALU-only run
    double s_ref, mhz;
    if (run<0>(cus, 16, iters, d_out, &s_ref, &mhz)) return 1;
    const double cyc_per_s = mhz * 1e6;
    printf("VALU only, 16 waves per CU: %.3f ms; s_memtime runs at %.0f MHz (taken as the shader clock)\n", s_ref * 1e3, mhz);
    printf("%-46s %6s %12s %12s %12s\n", "kind", "waves", "VALU/cyc/CU", "SALU/cyc/CU", "per wave/cyc");
    for (int wpc : {8, 16, 24, 32}) {
        double s;
        if (run<0>(cus, wpc, iters, d_out, &s)) return 1;
        printf("%-46s %6d %12.3f %12s %12.3f\n", "VALU only", wpc, wpc * per_wave / s / cyc_per_s, "-", per_wave / s / cyc_per_s);
        if (run<1>(cus, wpc, iters, d_out, &s)) return 1;
        printf("%-46s %6d %12s %12.3f %12.3f\n", "SALU only", wpc, "-", wpc * per_wave / s / cyc_per_s, per_wave / s / cyc_per_s);
        if (run<2>(cus, wpc, iters, d_out, &s)) return 1; // 256 VALU + 256 SALU per round and wave
        printf("%-46s %6d %12.3f %12.3f %12.3f\n", "VALU + SALU 1:1 in every wave", wpc, wpc * per_wave / s / cyc_per_s,
               wpc * per_wave / s / cyc_per_s, 2 * per_wave / s / cyc_per_s);
        if (run<3>(cus, wpc, iters, d_out, &s)) return 1; // half the waves each
        printf("%-46s %6d %12.3f %12.3f %12.3f\n", "VALU-only waves next to SALU-only waves (1:1)", wpc, wpc / 2 * per_wave / s / cyc_per_s,
               wpc / 2 * per_wave / s / cyc_per_s, per_wave / s / cyc_per_s);
        if (run<4>(cus, wpc, iters, d_out, &s)) return 1; // per round: 192 s_add + 64 s_cmp + 64 s_cbranch (not taken)
        printf("%-46s %6d %12s %12.3f %12.3f\n", "SALU with a compare + untaken branch per 3", wpc, "-", wpc * 1.25 * per_wave / s / cyc_per_s,
               1.25 * per_wave / s / cyc_per_s);
    }
    (void)hipFree(d_out);
    return 0;
}
