// How long does a wave wait for its kernel arguments?  One block per CU; lane 0 of wave 0 stamps s_memtime at entry and again once a
// value that depends on an argument (a pointer passed by value in a 256-byte struct) is in a register; a second stamp pair measures
// one dependent global load through that pointer (cold) for scale.  Eager launches and hipGraph replays, 200 launches each.
// build: hipcc --offload-arch=gfx950 -O3 -o kernarg_latency kernarg_latency.hip ; run: ./kernarg_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
struct Args { const int* p; long long* out; int pad[56]; int n; };
__global__ void __launch_bounds__(256) k(Args a) {
    const long long t0 = clock64();
    const int n = a.n;                                   // first use of an argument
    asm volatile("" :: "s"(n));
    long long t1 = clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t1 = clock64();
    const int v = a.p[(blockIdx.x * 64) & 1023];         // one dependent global load
    asm volatile("" :: "v"(v));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = clock64();
    if (threadIdx.x == 0) { a.out[blockIdx.x * 4 + 0] = t1 - t0; a.out[blockIdx.x * 4 + 1] = t2 - t1; a.out[blockIdx.x * 4 + 2] = n + v; }
}
int main() {
    const int nb = 256, reps = 200;
    int* p; long long* out;
    hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    hipMalloc(&out, nb * 4 * sizeof(long long));
    hipStream_t s; hipStreamCreate(&s);
    Args a{}; a.p = p; a.out = out; a.n = 7;
    std::vector<long long> h(nb * 4);
    auto report = [&](const char* name) {
        hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<long long> ka, ld;
        for (int b = 0; b < nb; ++b) { ka.push_back(h[b * 4]); ld.push_back(h[b * 4 + 1]); }
        std::sort(ka.begin(), ka.end()); std::sort(ld.begin(), ld.end());
        printf("%-28s kernel-argument wait: median %lld  p10 %lld  p90 %lld cycles | one cold global load behind it: median %lld\n", name,
               ka[nb / 2], ka[nb / 10], ka[nb * 9 / 10], ld[nb / 2]);
    };
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, s, a);
    hipStreamSynchronize(s);
    report("eager launch (last of 200)");
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, s, a);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    report("graph replay, 16 per launch");
    return 0;
}
