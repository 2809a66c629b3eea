// How fast do the 256 CUs take in the SAME bytes (a weight image every block reads in its prologue) compared with their OWN bytes?
// One 512-thread block per CU; every lane issues `per_lane` 16-byte loads back to back (all in flight), then waits; cycles from the first
// load to the last byte (s_memtime, per block), for
//   own       block b reads its own KB        (a streaming prologue: different lines per CU)
//   same      every block reads the same KB   (a weight image)
//   copies=R  block b reads copy b % R of the same KB (the image replicated R times in memory)
// Run twice per case inside one launch sequence: the first touch comes from HBM, the later ones from the LLC / the XCDs' L2s (the figures
// printed are the medians over blocks of the LAST of 5 launches).
// build: hipcc --offload-arch=gfx950 -O3 -o hot_lines hot_lines.hip ; run: ./hot_lines
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
template <int PER>
__global__ void __launch_bounds__(512) k(const float4* __restrict__ src, long long stride_f4, int copies, float* sink, long long* cyc) {
    const float4* p = src + (long long)(copies > 0 ? blockIdx.x % copies : blockIdx.x) * stride_f4 + threadIdx.x;
    float4 v[PER];
    const long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = p[i * 512];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (acc == 12345.678f) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PER> static void run(const char* name, const float4* src, long long stride_f4, int copies, float* sink, long long* cyc) {
    std::vector<long long> h(256);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<PER>, dim3(256), dim3(512), 0, 0, src, stride_f4, copies, sink, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double kb = PER * 512 * 16 / 1024.0;
    printf("  %-10s %5.0f KB per block: median %6lld cycles (%.1f B/clk/CU), p90 %6lld, max %6lld\n", name, kb, h[128], kb * 1024 / h[128], h[230], h[255]);
}
int main() {
    const size_t cap = 256u << 20;
    float4* src; float* sink; long long* cyc;
    hipMalloc(&src, cap); hipMemset(src, 0, cap); hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8);
    printf("96 KB per block (12 loads of 16 bytes per lane):\n");
    const long long s96 = 12 * 512;      // float4 elements per 96 KB
    run<12>("own", src, s96, 0, sink, cyc);
    run<12>("same", src, s96, 1, sink, cyc);
    run<12>("copies=2", src, s96, 2, sink, cyc);
    run<12>("copies=8", src, s96, 8, sink, cyc);
    run<12>("copies=32", src, s96, 32, sink, cyc);
    printf("48 KB per block:\n");
    const long long s48 = 6 * 512;
    run<6>("own", src, s48, 0, sink, cyc);
    run<6>("same", src, s48, 1, sink, cyc);
    run<6>("copies=8", src, s48, 8, sink, cyc);
    run<6>("copies=32", src, s48, 32, sink, cyc);
    return 0;
}
