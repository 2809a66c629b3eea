// A consumer wave's weight-slice prologue: 24 loads of 16 bytes per lane.  (a) the address pattern of w_load8 on a k_ts_gemm image —
// lane (c, kb) reads position (c & 3) * 16 + 4 w + (c >> 2) of row group 8 s + 2 kb (+1): per instruction 16 runs of 64 bytes, 3 KB
// apart; (b) the same 24 KB per wave read as contiguous 1 KB per instruction.  4 such waves per block (+ 4 idle), one block per CU; cycles
// from the first load to the last byte.
// build: hipcc --offload-arch=gfx950 -O3 -o slice_loads slice_loads.hip ; run: ./slice_loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
template <int PATTERN>
__global__ void __launch_bounds__(512) k(const float4* __restrict__ img_a, const float4* __restrict__ img_b, float* sink, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) return;
    const int c = lane & 15, kb = lane >> 4, MP = 192;
    float4 v[24];
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const float4* img = (i & 1) ? img_b : img_a;
        const int f = i >> 1, g = f % 3, s = (f / 3) & 1, half = f / 6;      // 3 gates x 2 k steps x 2 halves of 8 k
        if (PATTERN == 0) {
            const int col = g * 60 + 16 * wave + c, pos = (col & ~63) + (col & 3) * 16 + ((col >> 2) & 15);
            v[i] = img[(size_t)(8 * s + 2 * kb + half) * MP + pos];
        } else {
            v[i] = img[(size_t)(wave * 12 + f) * 64 + lane];
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) acc += v[i].x + v[i].w;
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (acc == 12345.678f) sink[0] = acc;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
__global__ void k_other(const float4* p, int n, float* sink) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 12345.678f) sink[0] = acc;
}
template <int PATTERN> static void run(const char* name, const float4* a, const float4* b, const float4* big, float* sink, long long* cyc) {
    std::vector<long long> h(1024);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_other, dim3(1024), dim3(256), 0, 0, big, 4 << 20, sink);      // 64 MB of other traffic between launches
        hipLaunchKernelGGL(k<PATTERN>, dim3(256), dim3(512), 0, 0, a, b, sink, cyc);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 1024 * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-44s median %6lld cycles per wave, p90 %6lld, max %6lld\n", name, h[512], h[920], h[1023]);
}
int main() {
    float4 *a, *b, *big; float* sink; long long* cyc;
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMalloc(&big, 64u << 20); hipMalloc(&sink, 64); hipMalloc(&cyc, 1024 * 8);
    hipMemset(a, 0, 1 << 20); hipMemset(b, 0, 1 << 20); hipMemset(big, 0, 64u << 20);
    run<0>("w_load8 pattern (16 runs of 64 B per load)", a, b, big, sink, cyc);
    run<1>("contiguous (1 KB per load)", a, b, big, sink, cyc);
    run<0>("w_load8 pattern (16 runs of 64 B per load)", a, b, big, sink, cyc);
    run<1>("contiguous (1 KB per load)", a, b, big, sink, cyc);
    return 0;
}
