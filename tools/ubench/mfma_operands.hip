// micro-benchmark: v_mfma_f32_16x16x4_f32 rate when every MFMA takes DIFFERENT source VGPRs (as in a real GEMM
// inner loop: 8 A values x 4 B values per batch) versus constant sources.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, const float* in, int iters) {
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0, 0, 0, 0};
    float pv[8]; float4 qv[8];
    for (int i = 0; i < 8; ++i) { pv[i] = in[threadIdx.x + i * 64]; qv[i] = ((const float4*)in)[threadIdx.x + i * 64]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const float a = MODE == 0 ? pv[0] : pv[st];
            const float4 b = MODE == 0 ? qv[0] : qv[st];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, MODE == 0 ? b.x : b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, MODE == 0 ? b.x : b.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, MODE == 0 ? b.x : b.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, MODE == 0 ? b.x : b.w, acc[3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float* out, float* in) {
    const int iters = 500;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, in, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("mode %d (%s sources): %.1f TF, %.1f cyc/MFMA/SIMD\n", MODE, MODE ? "varying" : "constant",
           256.0 * 4 * iters * 32 * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 32.0));
}
int main() {
    float *out, *in; (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&in, 1 << 20); (void)hipMemset(in, 0, 1 << 20);
    run<0>(out, in); run<1>(out, in);
    return 0;
}
