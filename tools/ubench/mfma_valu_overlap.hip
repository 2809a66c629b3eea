// Do fp32 MFMAs and fp32 VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?  (Round 4: every attempt to put more
// v_mfma_f32_16x16x4_f32 work beside the gather waves of the warp-specialised kernels cost about its own issue time.)
// One 512-thread block per CU = two waves per SIMD: waves 0-3 run a chain-free stream of fp32 MFMAs (4 independent accumulators), waves 4-7
// a stream of independent v_fma_f32.  Timed: MFMA waves alone, VALU waves alone, both together; and the same with bf16 MFMAs.
// "together ~ max(alone)" = separate pipes; "together ~ sum" = one datapath.  usage: mfma_valu_overlap.bin [iters=20000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: fp32 MFMA 16x16x4, 1: bf16 MFMA 16x16x32
__global__ void __launch_bounds__(512) k_mix(float* out, int iters, int run_mfma, int run_valu) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (run_mfma) {
            v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            const float x = 1.0f + threadIdx.x * 1e-6f, y = 0.5f;
            bf16x8 bx, by;
            for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
            for (int i = 0; i < iters; ++i) {
                if (KIND == 0) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a3, 0, 0, 0);
                }
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (run_valu) {
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
        const float m = 1.000001f, c = 1e-7f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {      // 16 independent v_fma_f32 per iteration (32 cycles of issue at 2 cycles each)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(m), "v"(c));
            }
        }
        r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static float run(float* out, int iters, int m, int v) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_mix<KIND><<<256, 512>>>(out, iters, m, v);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_mix<KIND><<<256, 512>>>(out, iters, m, v);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    CHECK(hipMalloc(&out, 4096));
    printf("256 blocks x 8 waves (4 MFMA waves + 4 VALU waves per CU, one of each per SIMD), %d iterations: 4 MFMAs | 16 v_fma_f32 per iteration\n", iters);
    printf("%-28s %12s %12s %12s   %s\n", "MFMA kind", "MFMA alone", "VALU alone", "together", "together / max(alone)");
    const float a0 = run<0>(out, iters, 1, 0), b0 = run<0>(out, iters, 0, 1), c0 = run<0>(out, iters, 1, 1);
    printf("%-28s %9.1f us %9.1f us %9.1f us   %.2f   (%.1f cycles per fp32 MFMA alone at 2.4 GHz)\n", "v_mfma_f32_16x16x4_f32", a0, b0, c0,
           c0 / (a0 > b0 ? a0 : b0), a0 * 2400.0 / (4.0 * iters));
    const float a1 = run<1>(out, iters, 1, 0), b1 = run<1>(out, iters, 0, 1), c1 = run<1>(out, iters, 1, 1);
    printf("%-28s %9.1f us %9.1f us %9.1f us   %.2f   (%.1f cycles per bf16 MFMA alone)\n", "v_mfma_f32_16x16x32_bf16", a1, b1, c1,
           c1 / (a1 > b1 ? a1 : b1), a1 * 2400.0 / (4.0 * iters));
    return 0;
}
