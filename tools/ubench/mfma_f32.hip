// micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate with NACC independent accumulators per wave and
// WAVES waves per SIMD.  usage: ./mfma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wps) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 2);
    const int iters = 4000 / NACC * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), block(256 * wps);
    hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    double flops = 256.0 * 4 * wps * iters * NACC * 2048.0;
    double cyc = ms * 1e-3 * 2.4e9 / (iters * (double)NACC * wps);
    printf("NACC=%2d waves/SIMD=%d: %.1f TF  (%.1f cyc per MFMA per SIMD @2.4GHz)  %.3f ms\n", NACC, wps, flops / ms / 1e9, cyc, ms);
    hipFree(out);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<12>(1); run<4>(2); run<12>(2); run<4>(4);
    return 0;
}
