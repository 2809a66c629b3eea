// Shared pieces of the software-pipelined (ELL-record) aggregate kernels: csrc/triplet_dma.hip (every wave runs the whole
// pipeline) and csrc/triplet_ws.hip (warp-specialised: producer waves run the pipeline, consumer waves the MFMA epilogue).
#pragma once
#include "triplet_kernels.h"
#include "bf16x3.h"

#include <stdlib.h>

namespace glam {

constexpr int kMetaSlots = 16;           // a_j / edge_attr slots of the one-piece side table (>= 11 packed edges, all written every pass)

struct FwdDmaArgs {
    const float* xw; const float* a_ij; const float* edge_attr; const float* w_edge; const float* M;
    const int* ell_src; const int* ell_eid;      // [N][4] each
    int N; int Cp; float slope;
    float* aggr; float* stats;
    const float* img_upd; const float* bias_p; float* out;     // fused update epilogue (k_triplet_fwd_pipe<..., FUSE = true>, k_triplet_fwd_ws)
};

struct PassMeta { int deg; int off; int tot; int dmax; };   // per lane: its node's degree, packed slot offset; wave-wide edge
                                                            // count and largest degree (wave-uniform)

// ---- warp-specialised kernels (triplet_ws.hip, triplet_ws_b1.hip): ring constants, DPP broadcasts, LDS flags ----
constexpr int kWsCons = 4;                // consumer waves: one 16-column tile of `out` each
#ifndef GLAM_WS_RING
#define GLAM_WS_RING 8
#endif
constexpr int kWsRing = GLAM_WS_RING;     // tile slots between producers and consumers
// LDS pitch (floats) of the W_edge rows (one row of H * Cp floats per bond type) in the warp-specialised kernels: a multiple of 64
__host__ __device__ constexpr int ws_wedge_pitch(int HC) { return (HC + 63) & ~63; }
// ... as a function of the head count alone (H * Cp <= 64 H): a compile-time constant in the kernels
__host__ __device__ constexpr int ws_wedge_pitch_h(int H) { return 64 * H; }

// lane n of the caller's 16-lane row (the lanes of one node) -> every lane of the row: one v_mov_b32_dpp row_newbcast (no LDS)
template <int CTRL>
__device__ __forceinline__ int dpp_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_int<CTRL>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int row_bcast_i(int v, int n) {
    switch (n & 15) {   // n is a compile-time constant after unrolling: the switch folds
        case 0: return dpp_int<0x150>(v);   case 1: return dpp_int<0x151>(v);   case 2: return dpp_int<0x152>(v);   case 3: return dpp_int<0x153>(v);
        case 4: return dpp_int<0x154>(v);   case 5: return dpp_int<0x155>(v);   case 6: return dpp_int<0x156>(v);   case 7: return dpp_int<0x157>(v);
        case 8: return dpp_int<0x158>(v);   case 9: return dpp_int<0x159>(v);   case 10: return dpp_int<0x15A>(v);  case 11: return dpp_int<0x15B>(v);
        case 12: return dpp_int<0x15C>(v);  case 13: return dpp_int<0x15D>(v);  case 14: return dpp_int<0x15E>(v);  default: return dpp_int<0x15F>(v);
    }
}
__device__ __forceinline__ float row_bcast(float v, int n) { return __builtin_bit_cast(float, row_bcast_i(__builtin_bit_cast(int, v), n)); }

// ---- 3 x bf16 tiles (bf16x3.h): what the producers of the warp-specialised kernels publish when the consumers' product runs on the
// bf16 matrix cores.  A tile is three planes (hi, mid, lo) of bf16 [16 rows][192 k]; a row takes 416 bytes: 104 words = 40 mod 64 makes
// the consumers' ds_read_b128 fragment reads (lane = row | 16-byte k block) conflict free in all four 16-lane groups of the instruction
constexpr int kX3RowBytes = 416;
constexpr int kX3PlaneBytes = 16 * kX3RowBytes;
constexpr int kX3TileBytes = 3 * kX3PlaneBytes;        // 19 968
constexpr int kWsRingX3 = 5;                           // tile slots of an x3 ring (19.5 KB each; B2 carries 48 KB of side tables besides)
bool ts_x3_enabled();                                  // GLAM_X3 (gemm.hip)

// GLAM_WS_GRID (developer knob: blocks of a warp-specialised launch, default one 12-wave block per CU), clamped to [1, max_blocks];
// a value that does not parse as a positive number is ignored
static inline int ws_grid_cap(int max_blocks) {
    const char* e = getenv("GLAM_WS_GRID");
    int v = e ? atoi(e) : 256;
    if (v < 1) v = 256;
    return v > max_blocks ? max_blocks : v;
}
// The warp-specialised kernels use 130-153 KB of dynamic LDS: the opt-in is a per-DEVICE function attribute, so it is set once per
// (kernel instantiation, device) and its return code is reported (a launch without it fails with a generic error).  `done` is the
// instantiation's own flag array.
static inline int ws_opt_in_lds(const void* fn, bool (&done)[64], const char* name) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;      // (an unknown device: set the attribute every time)
    if (done[dev] && dev != 63) return GLAM_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(GLAM_E_HIP, "%s: opting into 160 KB of dynamic LDS failed: %s", name, hipGetErrorString(e));
    done[dev] = true;
    return GLAM_OK;
}

__device__ __forceinline__ int flag_load(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void flag_bump(int* p) { (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }


}  // namespace glam
