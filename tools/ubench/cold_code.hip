// What does code cost the FIRST time a CU runs it?  One 64-thread block per CU runs a straight line of N vector instructions (fma chains
// on 8 registers: no memory, no waits) twice inside one launch — the first pass fetches the instructions (cold instruction cache: the
// caches are invalidated at every kernel boundary), the second pass finds them cached.  Also: the same line with a taken branch every
// 16 instructions (small basic blocks, as unrolled role / switch code has them).
// build: hipcc --offload-arch=gfx950 -O3 -o cold_code cold_code.hip ; run: ./cold_code
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define F8 "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n" \
           "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
#define F64 F8 F8 F8 F8 F8 F8 F8 F8
#define F512 F64 F64 F64 F64 F64 F64 F64 F64
#define B16 F8 F8 "s_branch 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n 1:\n"      // 16 instructions, then a taken branch over 8 nops
#define B128 B16 B16 B16 B16 B16 B16 B16 B16
#define B512 B128 B128 B128 B128
template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, float m) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.001f + i;
    long long t[3];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t[pass]) :: "memory");
        if (KIND == 0) asm volatile(F512 F512 F512 F512 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(m));
        else asm volatile(B512 B512 B512 B512 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(m));
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t[2]) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t[1] - t[0]; cyc[blockIdx.x * 2 + 1] = t[2] - t[1]; }
}
// 80 KB of other code on every CU: whatever instruction cache there is holds none of the test kernel afterwards
__global__ void __launch_bounds__(64) k_evict(float* out, float m) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.002f + i;
    asm volatile(F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(m));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND> static void run(const char* name, int blocks, float* out, long long* cyc, bool evict = true) {
    std::vector<long long> h(512);
    for (int r = 0; r < 3; ++r) {
        if (evict) hipLaunchKernelGGL(k_evict, dim3(512), dim3(64), 0, 0, out, 1.0001f);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0001f);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> a, b;
    for (int i = 0; i < blocks; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("  %-34s %3d blocks: first pass median %6lld cycles (%.1f per instruction), second pass %6lld (%.1f)\n", name, blocks, a[blocks / 2], a[blocks / 2] / 2048.0,
           b[blocks / 2], b[blocks / 2] / 2048.0);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 64 * 4); hipMalloc(&cyc, 512 * 8);
    printf("2048 vector instructions (16 KB of code), one wave per CU:\n");
    printf(" behind 80 KB of other code:\n");
    run<0>("straight line", 256, out, cyc);
    run<0>("straight line", 1, out, cyc);
    run<1>("a taken branch every 16", 256, out, cyc);
    run<1>("a taken branch every 16", 1, out, cyc);
    printf(" the same kernel again and again:\n");
    run<0>("straight line", 256, out, cyc, false);
    run<1>("a taken branch every 16", 256, out, cyc, false);
    return 0;
}
