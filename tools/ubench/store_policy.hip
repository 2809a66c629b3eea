// What does a launch pay, at its end, for the lines its stores left dirty in the XCDs' L2s?  A chain of dependent launches inside one
// hipGraph (the way the step runs); each launch reads R MB and writes W MB with one store policy:
//   plain        global_store_dwordx4                    (write-back L2: the dirty lines go out with the end-of-kernel release)
//   nt           ... nt          (__builtin_nontemporal_store)
//   sc1          ... sc1         (agent scope)
//   sc0 sc1      ... sc0 sc1     (system scope)
// Reported: us per launch (graph replay, 64 launches per graph) and, from the device-wide 100 MHz counter, the span first wave entry ->
// last wave exit — the difference is what happens outside the waves (dispatch, end-of-kernel cache write-back / invalidate).
// build: hipcc --offload-arch=gfx950 -O3 -o store_policy store_policy.hip ; run: ./store_policy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int POL> __device__ __forceinline__ void store(v4* p, v4 v) {
    if constexpr (POL == 0) *p = v;
    else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ unsigned long long g_first, g_last;
template <int POL>
__global__ void __launch_bounds__(768) k(const v4* __restrict__ in, v4* __restrict__ out, int nr, int nw, int stamp) {
    const unsigned long long t0 = wall_clock64();
    const int tid = blockIdx.x * 768 + threadIdx.x, nt = gridDim.x * 768;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < nr; i += nt) acc += in[i];
    for (int i = tid; i < nw; i += nt) store<POL>(out + i, acc + (float)i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp && (threadIdx.x & 63) == 0) { atomicMin(&g_first, t0); atomicMax(&g_last, wall_clock64()); }
}
template <int POL> static void run(const char* name, const v4* in, v4* a, v4* b, int nr, int nw, hipStream_t s) {
    const int per = 64;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < per; ++i) hipLaunchKernelGGL(k<POL>, dim3(256), dim3(768), 0, s, i & 1 ? b : (nr ? in : a), i & 1 ? a : b, nr, nw, 0);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipEventRecord(e0, s);
    for (int w = 0; w < 10; ++w) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // span of the waves of one launch
    std::vector<double> sp;
    for (int r = 0; r < 20; ++r) {
        unsigned long long f = ~0ull, l = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_first), &f, 8); hipMemcpyToSymbol(HIP_SYMBOL(g_last), &l, 8);
        hipLaunchKernelGGL(k<POL>, dim3(256), dim3(768), 0, s, nr ? in : a, b, nr, nw, 1);
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(&f, HIP_SYMBOL(g_first), 8); hipMemcpyFromSymbol(&l, HIP_SYMBOL(g_last), 8);
        sp.push_back((l - f) / 100.0);
    }
    std::sort(sp.begin(), sp.end());
    printf("  %-8s %7.2f us per launch in the graph | waves alive %6.2f us (median of 20 eager launches)\n", name, ms * 1e3 / (10 * per), sp[10]);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    const size_t cap = 64u << 20;
    v4 *in, *a, *b; hipMalloc(&in, cap); hipMalloc(&a, cap); hipMalloc(&b, cap);
    hipMemset(in, 0, cap); hipMemset(a, 0, cap); hipMemset(b, 0, cap);
    const int cases[][2] = {{0, 20}, {20, 20}, {36, 20}, {20, 0}, {0, 2}};
    for (auto& c : cases) {
        const int nr = c[0] * (1 << 16), nw = c[1] * (1 << 16);      // float4 elements per MB = 65536
        printf("read %d MB, write %d MB per launch (256 blocks x 768 threads):\n", c[0], c[1]);
        run<0>("plain", in, a, b, nr, nw, s);
        run<1>("nt", in, a, b, nr, nw, s);
        run<2>("sc1", in, a, b, nr, nw, s);
        run<3>("sc0 sc1", in, a, b, nr, nw, s);
    }
    return 0;
}
