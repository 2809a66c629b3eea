// Counter-based random numbers for the training-mode layers of the reference's default configuration
// (src_1gp/model.py:30-31: RReLU activations and Dropout(0.2)): Philox4x32-10 (Salmon et al., SC'11 — the public algorithm
// torch's CUDA generator also uses), keyed by a 64-bit seed, counter = (element-quad index, per-launch offset).
//
// hipGraph-safe stream position: the (seed, offset) pair lives in DEVICE memory (`state` int64[kRngStateWords = 288]: [0] seed, [1] offset,
// [16] a 32-bit ticket and [32 + 16 s] sixteen sub-tickets, each on its own cache line).  Every RNG-consuming launch reads the pair first, uses `offset` as its private stream id, and the LAST block
// of the launch to finish (ticket counter) stores offset + 1 for the next launch — no host round trip, no extra launch, and a
// replayed graph continues the sequence exactly where the previous replay (or eager step) left it.  Block 0 also records the pair
// it used in `eff` (int64[2]) so that the backward kernel regenerates the very same numbers instead of reading saved masks.
#pragma once
#include "common.h"

namespace glam {

struct Philox {
    unsigned k0, k1;
    unsigned o0, o1;     // per-launch offset (counter words 2, 3)
};

__device__ __forceinline__ Philox philox_init(const long long* pair) {
    const unsigned long long seed = (unsigned long long)pair[0], off = (unsigned long long)pair[1];
    return Philox{(unsigned)seed, (unsigned)(seed >> 32), (unsigned)off, (unsigned)(off >> 32)};
}

// four uniform 32-bit words for element quad `q`
__device__ __forceinline__ uint4 philox4(const Philox& p, unsigned long long q) {
    unsigned c0 = (unsigned)q, c1 = (unsigned)(q >> 32), c2 = p.o0, c3 = p.o1;
    unsigned k0 = p.k0, k1 = p.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// uniform in [0, 1): 24 random bits (every value exactly representable)
__device__ __forceinline__ float u01(unsigned w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ unsigned philox_word(const uint4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// forward prologue: the pair this launch uses; block 0 records it for the backward pass.  Every access to `state` is an
// agent-scope atomic (served at the device-coherent point, not from a per-XCD L2 line): the ticket atomics of the same
// launch sequence hit memory behind the L2s, and a plain load could return an offset cached before the previous launch's
// update — some blocks of one launch would then draw from another stream position than block 0 recorded (seen as a rare
// forward / backward mask mismatch).  The tickets live on their own 128-byte lines (state[16], state[32 + 16 s]).
__device__ __forceinline__ Philox rng_begin(long long* state, long long* eff) {
    const long long seed = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long off = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (eff && blockIdx.x == 0 && threadIdx.x == 0) { eff[0] = seed; eff[1] = off; }
    const long long pair[2] = {seed, off};
    return philox_init(pair);
}
// forward epilogue: the last block to arrive advances the stream position.  A block takes its ticket after it has CONSUMED the
// values it loaded from `state` (they fed its Philox key), so every read of the pair precedes the one store.  No fences: a
// release fence per block (an L2 write-back each) made the 2048-block tail kernels 5x slower than their arithmetic.
__device__ __forceinline__ void rng_ticket(long long* state, const Philox& p);
__device__ __forceinline__ void rng_end(long long* state, const Philox& p) {
    __syncthreads();
    if (threadIdx.x == 0) rng_ticket(state, p);
}
// the block's ticket, by ONE thread, after every thread of the block that read the pair has used it (rng_end: the block's barrier; a
// warp-specialised kernel whose other role has left: the last wave of the role that draws, counted in LDS)
__device__ __forceinline__ void rng_ticket(long long* state, const Philox& p) {
    {
        // Two-level ticket: block b first checks in at sub-counter b % 16 (each on its own 128-byte line: state[32 + 16 s]); the last
        // block of a sub-group checks in at the main ticket (state[16]).  One counter for all blocks serialised ~10 ns per block at the
        // coherent point — 5 us of a 512-block launch; now at most grid / 16 + 16 same-address operations are in any chain.
        const unsigned g = gridDim.x, sidx = blockIdx.x & 15u;
        const unsigned in_sub = (g - sidx + 15u) >> 4, nsub = g < 16u ? g : 16u;
        unsigned* sub = reinterpret_cast<unsigned*>(state + 32 + 16 * sidx);
        if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_sub - 1) {
            __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* ticket = reinterpret_cast<unsigned*>(state + 16);
            if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
                __hip_atomic_store(state + 1, (long long)((((unsigned long long)p.o1 << 32) | p.o0) + 1ull), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// One Philox word per element: an RReLU slope comes from its high 16 bits, a Dropout decision from its low 16 bits (independent halves
// of one uniform word)
__device__ __forceinline__ float rrelu_slope_w(unsigned w, float lo, float hi) { return fmaf(hi - lo, (float)(w >> 16) * (1.f / 65536.f), lo); }
__device__ __forceinline__ float drop_scale_w(unsigned w, float p) { return (float)(w & 0xffffu) * (1.f / 65536.f) >= p ? 1.f / (1.f - p) : 0.f; }

}  // namespace glam
