// Shared host/device helpers for libglam_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/glam_hip.h"

#include <hip/hip_ext.h>

namespace glam {

int fail(int code, const char* fmt, ...);

// ---- per-launch kernel timing (glam_prof_* in include/glam_hip.h) ----------------------------------------------------------
// Every kernel of the library is launched through hipLaunchKernelGGL.  While profiling is on, the same launch goes through
// hipExtLaunchKernelGGL with a (start, stop) event pair bound to THAT dispatch: the runtime stamps them with the dispatch's
// own begin / end timestamps — the figures a rocprofv3 kernel trace reports — so bench.py can quote the duration of each
// kernel of the very step it timed.  Off (the default): one predictable branch, the plain launch, hipGraph-capturable.
extern bool g_prof_on;
extern const char* g_prof_label;      // optional name for the NEXT timed launch (template launches stringify unresolved)
#define GLAM_PROF_LABEL(text) do { if (__builtin_expect(::glam::g_prof_on, 0)) ::glam::g_prof_label = (text); } while (0)
bool prof_slot(const char* kernel, unsigned grid, hipEvent_t* start, hipEvent_t* stop);

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                  \
    do {                                                                                                                     \
        hipEvent_t e0__, e1__;                                                                                               \
        if (__builtin_expect(::glam::g_prof_on, 0) && ::glam::prof_slot(#kernelName, dim3(numBlocks).x, &e0__, &e1__))       \
            hipExtLaunchKernelGGL(kernelName, dim3(numBlocks), dim3(numThreads), (memPerBlock), (streamId), e0__, e1__, 0,    \
                                  __VA_ARGS__);                                                                              \
        else                                                                                                                 \
            kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                               \
    } while (0)

#define GLAM_LAUNCH_CHECK(name)                                                    \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) return ::glam::fail(GLAM_E_HIP, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define GLAM_REQUIRE(cond, ...)                                        \
    do {                                                               \
        if (!(cond)) return ::glam::fail(GLAM_E_INVALID, __VA_ARGS__); \
    } while (0)

constexpr int kBlock = 256;        // 4 wavefronts of 64 lanes
constexpr int kMaxBlocks = 2048;   // 256 CUs x 8 resident blocks; grid-stride beyond
constexpr int kBwdBlocks = 512;    // cap on the blocks of the backward-by-target kernels = rows of their d_W_edge partial buffer

static inline int grid_for(int64_t work_items, int items_per_block, int cap = kMaxBlocks) {
    int64_t g = (work_items + items_per_block - 1) / items_per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers -------------------------------------------------------------------------
// Sum over the G consecutive lanes that share a node (G in {4,8,16,64}); every lane gets the total.  The butterfly
// runs on DPP modifiers (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: one v_add_f32_dpp per step) instead of
// ds_bpermute round trips through the LDS crossbar; fp add commutes, so the sums are bit-equal to the xor butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group_sum: lanes per node");
    v += dpp_move<0xB1>(v);                           // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);                           // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_move<0x141>(v);    // row_half_mirror: lane i <-> 7 - i of each 8
    if constexpr (G >= 16) v += dpp_move<0x140>(v);   // row_mirror: lane i <-> 15 - i of each 16
    if constexpr (G == 32) v += __shfl_xor(v, 16, 64);                                    // two DPP rows
    if constexpr (G == 64) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); }   // across the 4 DPP rows
    return v;
}

// Sum over the 64/G groups of a wavefront, lane-slot wise (lanes l, l+G, l+2G, ... are added).
template <int G>
__device__ __forceinline__ float cross_group_sum(float v) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// ... from / to an address KNOWN to be global memory.  A pointer that reaches a load through a run-time choice (a job struct picked by
// block index, a select against the address of a __device__ constant) is a generic pointer to the compiler and the access becomes a
// flat_load / flat_store: those count in BOTH wait counters, so every wait for an LDS operation (lgkmcnt) also waits for the global loads
// in flight — in a kernel that stages through LDS this serialises its prefetch.
typedef float glam_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) glam_v4f glam_gv4;
typedef __attribute__((address_space(1))) float glam_gf1;
__device__ __forceinline__ float4 ld4g(const float* p) {
    const glam_v4f v = *(const glam_gv4*)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld1g(const float* p) { return *(const glam_gf1*)(p); }
// ... with the non-temporal policy: an operand ONE workgroup reads ONCE (a streamed row block) need not displace the lines other waves
// gather from the XCD's L2 (MI355X_MICROARCH.md, nt-weights row: issued -> landed -18 % on streamed-once data)
__device__ __forceinline__ float4 ld4nt(const float* p) {
    const glam_v4f v = __builtin_nontemporal_load((const glam_gv4*)(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4g(float* p, float4 v) { *(glam_gv4*)(p) = (glam_v4f){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st1g(float* p, float v) { *(glam_gf1*)(p) = v; }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// base + 32-bit BYTE offset: lets the compiler use the scalar-base + 32-bit-VGPR-offset addressing mode instead of
// 64-bit pointer arithmetic on the vector ALU (the hosts check that every tensor stays below 4 GiB)
__device__ __forceinline__ float4 ld4o(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st4o(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// The same store WRITTEN THROUGH the XCD's L2 (`sc1`, agent scope) — for the large output tensors of a launch.  Each XCD has its own
// write-back L2 and the L2s are not coherent with each other: what a launch leaves dirty there goes out with the release at its end,
// after the last wave, before the next launch may start (tools/ubench/store_policy.hip: 20 MB of plain stores cost a launch 0.6-0.8 us
// that `sc1` stores do not; `nt` does not help).  The next launch starts with an invalidated L2 either way, so nothing is lost.
// A raw buffer store rather than inline assembly: the compiler keeps counting it in vmcnt.
typedef unsigned glam_v4u __attribute__((ext_vector_type(4)));
// (`base` and `uniform_off` must be wave-uniform: they travel in scalar registers; byte_off is the lane's part)
__device__ __forceinline__ void st4o_wt(float* base, unsigned byte_off, float4 v, unsigned uniform_off = 0) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(glam_v4u{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, (int)byte_off,
                                           (int)uniform_off, 16);
}
__device__ __forceinline__ void stfo_wt(float* base, unsigned byte_off, float v, unsigned uniform_off = 0) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, (int)uniform_off, 16);
}
// WHEN: only launches over at most kWtMaxRows rows (nodes).  There the write-back at the end is a visible share of the launch (the
// headline step at B = 1 024: 68.0 -> 62.6 us); over more rows the L2 merges the partial lines a tile's stores leave before it evicts
// them, and the same stores written through cost MORE (B = 4 096: +5 %, B = 16 384: +15-30 % per kernel; B = 2 048: even) —
// profiles/r5_store_policy.txt.  The choice is a wave-uniform branch on a kernel argument (st4o_sel) or a template argument (st4o_t).
constexpr int kWtMaxRows = 32768;
__device__ __forceinline__ void st4o_sel(bool wt, float* base, unsigned byte_off, float4 v, unsigned uniform_off = 0) {
    if (wt) st4o_wt(base, byte_off, v, uniform_off);
    else st4o(base, byte_off + uniform_off, v);
}
// the same choice made at compile time (the warp-specialised kernels: a branch around the stores of their steady loops cost them half
// of what the policy gains — the hosts pick the instantiation)
template <bool WT> __device__ __forceinline__ void st4o_t(float* base, unsigned byte_off, float4 v) {
    if constexpr (WT) st4o_wt(base, byte_off, v);
    else st4o(base, byte_off, v);
}
__device__ __forceinline__ int ldio(const int* base, unsigned byte_off) {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// A gathered row chunk of 4 channels as it sits in registers between the load and its use
struct XwRow {
    typedef float4 T;
    static constexpr unsigned kElem = 4u;
    static __device__ __forceinline__ T load(const float* base, unsigned byte_off) { return ld4o(base, byte_off); }
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float4 get(T v) { return v; }
};
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ void fma4(float4& acc, float s, float4 a) {
    acc.x = fmaf(s, a.x, acc.x); acc.y = fmaf(s, a.y, acc.y); acc.z = fmaf(s, a.z, acc.z); acc.w = fmaf(s, a.w, acc.w);
}
// fma4 spelled as two <2 x float> fmas: v_pk_fma_f32 (two fp32 lanes per instruction, same rounding).  The SLP vectoriser
// finds these pairs by itself in the forward kernel (forcing them there costs registers: 10.4 vs 9.7 us) but leaves B1's
// inner loop half scalar (231 v_fmac vs 120 v_pk_fma), and B1 is the kernel that is VALU bound.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma4_pk(float4& acc, float s, float4 a) {
    const v2f sv = {s, s};
    const v2f lo = __builtin_elementwise_fma(sv, (v2f){a.x, a.y}, (v2f){acc.x, acc.y});
    const v2f hi = __builtin_elementwise_fma(sv, (v2f){a.z, a.w}, (v2f){acc.z, acc.w});
    acc = make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// exp of a non-positive softmax argument (logit - segment max): one multiply + v_exp_f32 instead of the
// ~18-instruction libm sequence (the aggregate kernels are VALU-issue bound for ~45 % of their time).  Error:
// |x| * 2^-24 relative from the rounded x*log2(e) plus 1 ulp from the instruction, i.e. < 1e-6 on a softmax weight
// for |x| <= 16 (the 1e-5 parity tests run on this path); -DGLAM_EXACT_EXP restores expf.  Forward and backward
// share the function, so the recomputed alpha equals the forward pass's.
__device__ __forceinline__ float softmax_exp(float x) {
#ifdef GLAM_EXACT_EXP
    return expf(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#endif
}

// Contiguous global -> LDS copy of n4 float4 (n4 % 64 == 0) by a block of BLOCK threads: LDS-DMA, 1 KB per wave
// instruction, no staging registers.  Completion is tracked by vmcnt: drain it (s_waitcnt vmcnt(0), which
// __syncthreads() does) before another wave reads the destination.
template <int BLOCK>
__device__ __forceinline__ void lds_copy_async(const float* g, float* l, int n4, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int b = wave * 64; b < n4; b += BLOCK)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 4 * (size_t)(b + lane)),
                                         (__attribute__((address_space(3))) void*)(l + 4 * b), 16, 0, 0);
}

// One LDS-DMA piece issued by hand: lane l's 16 bytes from gbase + byte_off land at lds_base + 16 l (wave-uniform base, in M0).
// Inline asm because the builtin makes the compiler drain vmcnt before every later LDS read; the piece is invisible to its
// waitcnt bookkeeping, which only makes its own vmcnt(n) waits conservative (loads return in order) — the consumer must
// s_waitcnt vmcnt(0) + barrier before reading the destination.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // "m0 is a reserved register": it is the instruction's LDS base operand
__device__ __forceinline__ void dma16(const void* gbase, unsigned byte_off, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(gbase), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)p);
}

// CELU(alpha = 1) and its derivative (torch.celu: max(0,x) + min(0, exp(x) - 1))
__device__ __forceinline__ float celu1(float x) { return x > 0.f ? x : expf(x) - 1.f; }
__device__ __forceinline__ float celu1_grad(float x) { return x > 0.f ? 1.f : expf(x); }
__device__ __forceinline__ float4 celu4(float4 v) { return make_float4(celu1(v.x), celu1(v.y), celu1(v.z), celu1(v.w)); }

// Column sums of rows [r0, r1) of base[., D] into dst[0..D): 16 row groups x 16 float4 column chunks (D % 4 == 0,
// D <= 64) with every load of a thread independent of the others, then a fixed-order sum of the 16 partials; scalar
// fallback otherwise.  (A residue segment has hundreds of rows: the obvious one-thread-per-column loop is a serial chain
// of that many dependent round trips and was 0.7 ms of the two-tower step.)  Ends with a barrier.
__device__ __forceinline__ void block_colsum(const float* base, int r0, int r1, int D, float* s_part, float* dst) {
    const int tid = threadIdx.x;
    if ((D & 3) == 0 && D <= 64) {
        const int c4 = tid & 15, rg = tid >> 4;
        float4 acc = f4zero();
        if (4 * c4 < D) {
            int r = r0 + rg;
            for (; r + 48 < r1; r += 64) {
                const float4 v0 = ld4(base + (size_t)r * D + 4 * c4), v1 = ld4(base + (size_t)(r + 16) * D + 4 * c4),
                             v2 = ld4(base + (size_t)(r + 32) * D + 4 * c4), v3 = ld4(base + (size_t)(r + 48) * D + 4 * c4);
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
            }
            for (; r < r1; r += 16) {
                const float4 v = ld4(base + (size_t)r * D + 4 * c4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        st4(s_part + (rg * 16 + c4) * 4, acc);
        __syncthreads();
        if (tid < D) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sum += s_part[(g * 16 + (tid >> 2)) * 4 + (tid & 3)];
            dst[tid] = sum;
        }
    } else {
        for (int c = tid; c < D; c += kBlock) {
            float sum = 0.f;
            for (int r = r0; r < r1; ++r) sum += base[(size_t)r * D + c];
            dst[c] = sum;
        }
    }
    __syncthreads();
}

// ts_gemm weight-image column order: position p = cg*64 + t*16 + c holds logical column cg*64 + 4c + t
__host__ __device__ inline int ts_col_of_pos(int p) { return (p & ~63) + 4 * (p & 15) + ((p >> 4) & 3); }

}  // namespace glam
