// What does a device-wide phase barrier cost against a kernel boundary on this part?  ("one persistent launch for the
// B = 1024 step, or a committed table showing where a grid barrier costs more than a launch".)
// The headline step is a chain of six dependent launches, 256 blocks (one per CU) each.  This program times, under hipGraph replay:
//   chain     six dependent launches of a phase kernel (each block writes `bytes` floats and reads what its two neighbours wrote in the
//             previous phase: the dependency shape of the step — gathered rows come from the same or an adjacent node tile)
//   flat      ONE launch running the six phases behind a device-wide counter barrier (lane-0 agent release, relaxed sc1 polling with
//             s_sleep, agent acquire: cdna_hip_programming.md G16 / MI355X_MICROARCH.md "barrier-counter")
//   xcd       ONE launch with the XCD-hierarchical barrier (per-XCC counter -> leader -> top counter -> per-XCC generation: "barrier-xcd")
// for a light phase (64 B per block) and a phase that leaves 64 KB per block dirty.  usage: grid_barrier.bin [blocks=256] [reps=200]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)
constexpr int kPhases = 6, kThreads = 256;

__device__ __forceinline__ void phase_body(float* buf, int nfloat, int phase, int b, int nb, int tid) {
    // read what the two neighbours wrote in the previous phase, write this block's slab
    const float* prev = buf + (size_t)((phase + 1) & 1) * nb * nfloat;
    float* cur = buf + (size_t)(phase & 1) * nb * nfloat;
    const int l = (b + nb - 1) % nb, r = (b + 1) % nb;
    for (int i = tid; i < nfloat; i += kThreads) cur[(size_t)b * nfloat + i] = 0.5f * (prev[(size_t)l * nfloat + i] + prev[(size_t)r * nfloat + i]) + 1.f;
}

__global__ void __launch_bounds__(kThreads) k_phase(float* buf, int nfloat, int phase) {
    phase_body(buf, nfloat, phase, blockIdx.x, gridDim.x, threadIdx.x);
}

typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned ld_rlx(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// monotonic counter barrier: arrive = one atomic per block behind an agent release; wait until count reaches nb * epoch
__device__ __forceinline__ void barrier_flat(unsigned* cnt, unsigned epoch, unsigned nb) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_rlx(cnt) < nb * epoch) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// XCD-hierarchical: block b arrives at its XCC's counter (b % 8 = the XCC it runs on is used only to spread the counters; correctness
// does not depend on it); the last arriver of an XCC arrives at the top counter; the last of those bumps the generation word
__device__ __forceinline__ void barrier_xcd(unsigned* st, unsigned epoch, unsigned nb) {
    // st[0] = generation, st[32] = top counter, st[64 + 32 x] = counter of group x (each on its own 128-byte line)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7, per = (nb + 7 - x) / 8;        // blocks in group x
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(st + 64 + 32 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == per * epoch) {
            const unsigned t = __hip_atomic_fetch_add(st + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned groups = nb < 8 ? nb : 8;
            if (t + 1 == groups * epoch) __hip_atomic_store(st, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (ld_rlx(st) < epoch) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int KIND>
__global__ void __launch_bounds__(kThreads) k_persistent(float* buf, int nfloat, unsigned* st) {
    for (int phase = 0; phase < kPhases; ++phase) {
        phase_body(buf, nfloat, phase, blockIdx.x, gridDim.x, threadIdx.x);
        if (phase + 1 < kPhases) {
            if (KIND == 0) barrier_flat(st + 32, phase + 1, gridDim.x);
            else barrier_xcd(st, phase + 1, gridDim.x);
        }
    }
}

static float time_graph(hipGraphExec_t ge, hipStream_t s, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 200;
    const int kSteps = 16;                   // chains per graph launch (as bench.py: sixteen steps per launch)
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    unsigned* st;
    CHECK(hipMalloc(&st, 4096));
    printf("blocks = %d (one per CU at 256), %d phases per step, %d steps per graph launch; us per STEP\n", nb, kPhases, kSteps);
    printf("%-28s %10s %10s %10s   %s\n", "phase writes per block", "chain", "flat", "xcd", "(barrier - boundary) per phase transition: flat / xcd");
    for (int nfloat : {16, 16384}) {
        float* buf;
        CHECK(hipMalloc(&buf, (size_t)2 * nb * nfloat * sizeof(float)));
        CHECK(hipMemset(buf, 0, (size_t)2 * nb * nfloat * sizeof(float)));
        float us[3];
        for (int kind = 0; kind < 3; ++kind) {
            hipGraph_t g; hipGraphExec_t ge;
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int step = 0; step < kSteps; ++step) {
                if (kind == 0) {
                    for (int p = 0; p < kPhases; ++p) k_phase<<<nb, kThreads, 0, s>>>(buf, nfloat, p);
                } else {
                    CHECK(hipMemsetAsync(st, 0, 4096, s));           // barrier state is re-initialised by every call (G16)
                    if (kind == 1) k_persistent<0><<<nb, kThreads, 0, s>>>(buf, nfloat, st);
                    else k_persistent<1><<<nb, kThreads, 0, s>>>(buf, nfloat, st);
                }
            }
            CHECK(hipStreamEndCapture(s, &g));
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            us[kind] = time_graph(ge, s, reps) / kSteps;
            CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
        }
        // check: the three forms computed the same thing
        std::vector<float> h((size_t)nb * nfloat);
        CHECK(hipMemcpy(h.data(), buf + (size_t)((kPhases - 1) & 1) * nb * nfloat, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        char label[64];
        snprintf(label, sizeof label, "%d B", nfloat * 4);
        printf("%-28s %10.2f %10.2f %10.2f   %+.2f / %+.2f us   (sample %.4f)\n", label, us[0], us[1], us[2], (us[1] - us[0]) / (kPhases - 1),
               (us[2] - us[0]) / (kPhases - 1), h[h.size() / 2]);
        CHECK(hipFree(buf));
    }
    return 0;
}
