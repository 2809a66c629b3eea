// The optimizer step of the training loop that drives the path: `Adam(self.model.parameters(), lr=args.lr)` (reference
// src_1gp/trainer.py:49-50, stepped once per batch at trainer.py:301; lr moved by ReduceLROnPlateau, trainer.py:55,85).
// A default-shaped model has 36 parameter tensors between 1 and 307 200 elements (355 k in all).  The library's multi-tensor kernel cuts
// them into 65 536-element chunks — 9 workgroups on a 256-CU device, in double precision because its hyper-parameters are doubles — and
// took 45 us per step (6 % of a 0.79 ms step before this file existed).  Here: ONE launch over all tensors, 1 024-element
// chunks (one float4 per thread, one round trip), fp32 arithmetic: 5.3 us (tools/bench_adam.py).
//   step s = *step + 1                                   (device counter: a captured launch replays correctly)
//   g' = g + weight_decay * p
//   m  = m + (1 - beta1) (g' - m)          v = beta2 v + (1 - beta2) g'^2
//   p  = p - lr / (1 - beta1^s) * m / (sqrt(v) / sqrt(1 - beta2^s) + eps)
// The tensor addresses travel by value in the kernel arguments (gradients are fresh allocations every eager step: no table in device
// memory to keep in sync).  The step counter is written by the LAST workgroup to finish (a two-level ticket), after every workgroup has read it.
#include "common.h"

namespace glam {

constexpr int kAdamMaxTensors = 40;
constexpr int kAdamChunk = 4 * kBlock;

struct AdamArgs {
    float* p[kAdamMaxTensors]; const float* g[kAdamMaxTensors]; float* m[kAdamMaxTensors]; float* v[kAdamMaxTensors];
    int numel[kAdamMaxTensors];
    int chunk_end[kAdamMaxTensors];      // running count of chunks: tensor t owns workgroups [chunk_end[t-1], chunk_end[t])
    int n;
    float* step; const float* lr_dev; unsigned* ticket;
    double lr, beta1, beta2, eps, weight_decay;
    int bump;                            // last launch of a step: advance *step
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2, float step_size,
                                         float bc2_sqrt, float eps, float wd) {
    if (wd != 0.f) g = g + wd * p;
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(kBlock) k_adam(AdamArgs a) {
    const int tid = threadIdx.x, b = blockIdx.x;
    // owner of this workgroup: lane l compares against chunk_end[l] (one vector load from the argument block), the count of passed
    // boundaries is the tensor index — a scalar walk over the table is a chain of up to 40 dependent scalar loads
    const int lane = tid & 63;
    const bool passed = lane < a.n - 1 && b >= a.chunk_end[lane];
    const int t = __builtin_popcountll(__ballot(passed));
    const int first = t ? a.chunk_end[t - 1] : 0;
    const int off = (b - first) * kAdamChunk + 4 * tid, numel = a.numel[t];
    float* p = a.p[t]; const float* g = a.g[t]; float* m = a.m[t]; float* v = a.v[t];
    // agent-scope atomic loads, as the RNG position is read (rng.h): the previous launch's last workgroup published the count with an
    // agent-scope store and no fence, and a plain load may be served a line cached before that update — a stale count would give
    // some workgroups the bias correction of step s - 1, and a stale ticket winner would store the old count again
    const float s = __hip_atomic_load(a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.f;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && off + 4 <= numel;
    float4 pv = f4zero(), gv = f4zero(), mv = f4zero(), vv = f4zero();
    if (vec) { pv = ld4(p + off); gv = ld4(g + off); mv = ld4(m + off); vv = ld4(v + off); }
    else {
        float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (off + i < numel) { pp[i] = p[off + i]; gp[i] = g[off + i]; mp[i] = m[off + i]; vp[i] = v[off + i]; }
    }
    const double lr = a.lr_dev ? (double)__hip_atomic_load(a.lr_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.lr;
    // 1 - beta^s = -expm1(s ln beta): fp32 keeps 1e-7 relative accuracy at every s (1 - powf(beta, s) would lose four digits at s = 1),
    // and the double-precision pow / divide / sqrt of the direct form cost 0.8 us of a 6 us launch
    const float step_size = (float)lr / -expm1f(s * (float)log(a.beta1));
    const float bc2_sqrt = sqrtf(-expm1f(s * (float)log(a.beta2)));
    const float omb1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, omb2 = (float)(1.0 - a.beta2), eps = (float)a.eps,
                wd = (float)a.weight_decay;
    adam_one(pv.x, gv.x, mv.x, vv.x, omb1, b2, omb2, step_size, bc2_sqrt, eps, wd);
    adam_one(pv.y, gv.y, mv.y, vv.y, omb1, b2, omb2, step_size, bc2_sqrt, eps, wd);
    adam_one(pv.z, gv.z, mv.z, vv.z, omb1, b2, omb2, step_size, bc2_sqrt, eps, wd);
    adam_one(pv.w, gv.w, mv.w, vv.w, omb1, b2, omb2, step_size, bc2_sqrt, eps, wd);
    if (vec) { st4(p + off, pv); st4(m + off, mv); st4(v + off, vv); }
    else {
        const float* pp = &pv.x; const float* mp = &mv.x; const float* vp = &vv.x;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (off + i < numel) { p[off + i] = pp[i]; m[off + i] = mp[i]; v[off + i] = vp[i]; }
    }
    if (a.bump) {
        // Every workgroup has consumed *step (it fed the values stored above) before it takes its ticket; the last one to arrive writes the
        // new count and re-arms the tickets.  Two levels as in rng_end (rng.h): workgroup b checks in at sub-counter b % 16 (one 128-byte
        // line each), the last of a sub-group at the main ticket — a single counter serialises every workgroup of the launch at the
        // coherent point (13 of this launch's 16 us), and a release fence per workgroup costs an L2 write-back each.
        __syncthreads();
        if (tid == 0) {
            const unsigned g = gridDim.x, sidx = blockIdx.x & 15u;
            const unsigned in_sub = (g - sidx + 15u) >> 4, nsub = g < 16u ? g : 16u;
            unsigned* sub = a.ticket + 32 * (1 + sidx);
            if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_sub - 1) {
                __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
                    __hip_atomic_store(a.step, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

}  // namespace glam

using namespace glam;

extern "C" int glam_adam_max_tensors(void) { return kAdamMaxTensors; }

extern "C" int glam_adam_step(const uint64_t* table, const int64_t* numel, int n, float* step, unsigned* ticket, const float* lr_dev,
                              double lr, double beta1, double beta2, double eps, double weight_decay, void* stream) {
    GLAM_REQUIRE(n >= 0 && (n == 0 || (table && numel)) && step && ticket, "glam_adam_step: null pointer / negative count");
    GLAM_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && weight_decay >= 0, "glam_adam_step: bad hyper-parameters");
    int done = 0;
    // tensors with no elements take no workgroup; the step counter advances even when nothing is left to launch for
    int last_nonempty = -1;
    for (int i = 0; i < n; ++i) {
        GLAM_REQUIRE(numel[i] >= 0 && numel[i] < ((int64_t)1 << 31) - kAdamChunk, "glam_adam_step: tensor too large for 32-bit offsets");
        if (numel[i] > 0) {
            GLAM_REQUIRE(table[4 * i] && table[4 * i + 1] && table[4 * i + 2] && table[4 * i + 3], "glam_adam_step: null tensor pointer");
            last_nonempty = i;
        }
    }
    if (last_nonempty < 0) return GLAM_OK;
    while (done <= last_nonempty) {
        AdamArgs a{};
        int k = 0, chunks = 0;
        while (done <= last_nonempty && k < kAdamMaxTensors) {
            const int i = done++;
            if (numel[i] == 0) continue;
            a.p[k] = (float*)table[4 * i]; a.g[k] = (const float*)table[4 * i + 1];
            a.m[k] = (float*)table[4 * i + 2]; a.v[k] = (float*)table[4 * i + 3];
            a.numel[k] = (int)numel[i];
            chunks += (int)((numel[i] + kAdamChunk - 1) / kAdamChunk);
            a.chunk_end[k] = chunks;
            ++k;
        }
        if (k == 0) break;
        a.n = k; a.step = step; a.lr_dev = lr_dev; a.ticket = ticket;
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
        a.bump = done > last_nonempty ? 1 : 0;
        hipLaunchKernelGGL(k_adam, dim3(chunks), dim3(kBlock), 0, (hipStream_t)stream, a);
        GLAM_LAUNCH_CHECK("glam_adam_step");
    }
    return GLAM_OK;
}
