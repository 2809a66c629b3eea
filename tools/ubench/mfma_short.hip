// micro-benchmark: does the shader clock hold when the work is a stream of SHORT kernels?  Launches N back-to-back
// MFMA kernels of ~`iters` MFMAs per wave and reports the sustained TF/s plus the shader clock seen by
// s_memtime (wall_clock64 = constant 100 MHz counter).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out, long long* clk, int iters, float a0, float b0) {
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
int main() {
    float* out; long long* clk; long long h[2];
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int iters : {50, 200, 1000, 20000}) {
        const int n = iters >= 20000 ? 20 : 2000;
        for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, clk, iters, 1.0f, 2.0f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < n; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, clk, iters, 1.0f, 2.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double us = ms * 1e3 / n, flops = 256.0 * 4 * iters * 4 * 2048.0;
        printf("iters=%6d: %8.2f us/launch  %6.1f TF sustained | in-kernel: %lld shader cycles in %lld x10ns -> %.2f GHz, %.1f cyc/MFMA\n",
               iters, us, flops / us / 1e6, h[0], h[1], h[0] / (h[1] * 10.0), (double)h[0] / (iters * 4));
    }
    return 0;
}
