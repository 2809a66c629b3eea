// What does the first touch of a buffer cost a wave at the start of a launch?  One 64-thread block per CU issues one 16-byte load per lane
// from each of K buffers (separate 8 MB allocations) back to back, then waits: cycles from the first load to the last byte, K = 1 .. 16.
// Between measurements an "other" kernel touches 64 other buffers (whatever translation / cache state a CU keeps belongs to them).
// The same K loads from ONE buffer (K x 1 KB apart) for comparison.
// build: hipcc --offload-arch=gfx950 -O3 -o first_touch first_touch.hip ; run: ./first_touch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
struct Ptrs { const float4* p[16]; };
template <int K>
__global__ void __launch_bounds__(64) k(Ptrs a, long long block_stride_f4, float* sink, long long* cyc) {
    float4 v[K];
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = a.p[i][blockIdx.x * block_stride_f4 + threadIdx.x];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) acc += v[i].x + v[i].w;
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (acc == 12345.678f) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(64) k_other(const float4* const* bufs, int n, float* sink) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += bufs[i][blockIdx.x * 64 + threadIdx.x].x;
    if (acc == 12345.678f) sink[0] = acc;
}
template <int K> static void run(const Ptrs& a, const char* what, const float4* const* others, float* sink, long long* cyc, long long bs = 64) {
    std::vector<long long> h(256);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_other, dim3(256), dim3(64), 0, 0, others, 64, sink);
        hipLaunchKernelGGL(k<K>, dim3(256), dim3(64), 0, 0, a, bs, sink, cyc);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  K = %2d %-18s (blocks %4lld KB apart) median %6lld cycles, p90 %6lld, max %6lld\n", K, what, bs * 16 / 1024, h[128], h[230], h[255]);
}
int main() {
    const size_t sz = 40u << 20;
    std::vector<float4*> bufs(80);
    for (auto& b : bufs) { hipMalloc(&b, sz); hipMemset(b, 0, sz); }
    const float4** others; hipMalloc(&others, 64 * sizeof(void*)); hipMemcpy(others, bufs.data() + 16, 64 * sizeof(void*), hipMemcpyHostToDevice);
    float* sink; long long* cyc; hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
    Ptrs many{}, one{};
    for (int i = 0; i < 16; ++i) { many.p[i] = bufs[i]; one.p[i] = bufs[0] + i * 256 * 64; }      // one buffer: 256 KB apart (the far runs need 256 x 128 KB = 32 MB per buffer)
    run<1>(many, "buffers", others, sink, cyc);
    run<2>(many, "buffers", others, sink, cyc);
    run<4>(many, "buffers", others, sink, cyc);
    run<8>(many, "buffers", others, sink, cyc);
    run<16>(many, "buffers", others, sink, cyc);
    run<4>(one, "pieces of one", others, sink, cyc);
    run<16>(one, "pieces of one", others, sink, cyc);
    // every block in its own part of every buffer (a tile per CU): 128 KB apart
    const long long far = 128 * 1024 / 16;
    run<1>(many, "buffers", others, sink, cyc, far);
    run<2>(many, "buffers", others, sink, cyc, far);
    run<4>(many, "buffers", others, sink, cyc, far);
    run<8>(many, "buffers", others, sink, cyc, far);
    run<16>(many, "buffers", others, sink, cyc, far);
    return 0;
}
