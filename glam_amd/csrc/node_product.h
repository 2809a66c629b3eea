// The node product of a TripletMessage, x[16, K <= 64] @ [W_node | Wa] (src_1gp/layer.py:37 + the separable attention columns), taken by
// the PRODUCER waves of the kernel that writes x — the GRU step of the block's previous application (block.hip: k_gru_fwd_ws<.., NODE>) or
// the input embedding in front of its first (tall_x3.hip: k_tall_x3<.., NODE>) — while the tile is still in LDS: the TripletMessage then
// starts at its aggregate launch (glam_triplet_layer_fwd_ell with x = NULL).
//   The consumer waves of those kernels (four, 16 output channels each) leave every finished tile in a second LDS ring as fp32 rows
//   (kNodeXPitch bytes per row, kNodeXRing tiles; s_xready[slot]: consumer waves that wrote, s_xtaken[slot]: producer waves that hold
//   the fragments); producer wave p owns the product's columns 48 p .. 48 p + 47 — three 16-column fragments, W as the FIRST matrix
//   operand, from the layer's pre-split fragment image (layer.hip: kNodePreFloats; 18 coalesced 1 KB loads per wave).  The partial
//   products are k_ts_gemm_x3_sw's in its order (there x is the first operand): the same bits as the stand-alone launch.
#pragma once
#include "bf16x3.h"
#include "triplet_pipe.h"

namespace glam {

constexpr int kNodeXPitch = 272, kNodeXTile = 16 * kNodeXPitch, kNodeXRing = 4;      // fp32 rows: 64 channels + 4
constexpr size_t kNodeXBytes = (size_t)kNodeXRing * kNodeXTile;

struct NodeOut { float* xw; float* a_ij; int m1; int N; };

// this wave's 18 fragments of the image (`wave` = producer index 0 .. 3)
__device__ __forceinline__ void node_load_fragments(Bf16x3 (&wn)[2][3], const void* node_pre, int wave, int lane) {
    const char* base = reinterpret_cast<const char*>(node_pre) + (size_t)wave * (18 * 1024) + lane * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j3 = 0; j3 < 3; ++j3) {
            const char* f = base + (s * 3 + j3) * 3072;
            wn[s][j3].hi = ldfrag(f); wn[s][j3].mid = ldfrag(f + 1024); wn[s][j3].lo = ldfrag(f + 2048);
        }
}

// one finished tile (ring slot `xs`, rows 16 tile ..): fragments out of LDS (fp32, split here: the consumers set the pace and write one
// 16-byte piece per lane), the slot handed back, 36 matrix instructions, three 16-byte stores per lane
__device__ __forceinline__ void node_product_tile(const Bf16x3 (&wn)[2][3], const char* s_xn, int* s_xtaken, int xs, int tile, int wave, int lane,
                                                  const NodeOut& o) {
    const int nc = lane & 15, nkb = lane >> 4;
    asm volatile("" ::: "memory");
    const char* xb = s_xn + xs * kNodeXTile + nc * kNodeXPitch + nkb * 32;       // row nc, k = 32 s + 8 nkb ..
    float4 xr[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { xr[s][0] = *reinterpret_cast<const float4*>(xb + 128 * s); xr[s][1] = *reinterpret_cast<const float4*>(xb + 128 * s + 16); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) flag_bump(s_xtaken + xs);
    Bf16x3 xv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) xv[s] = split8(xr[s][0], xr[s][1]);
    v4f_t acc[3], accb[3];
#pragma unroll
    for (int j3 = 0; j3 < 3; ++j3) { acc[j3] = (v4f_t){0.f, 0.f, 0.f, 0.f}; accb[j3] = acc[j3]; }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j3 = 0; j3 < 3; ++j3) {      // x.mid w.mid, x.hi w.lo, x.lo w.hi
            acc[j3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[s][j3].mid, xv[s].mid, acc[j3], 0, 0, 0);
            acc[j3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[s][j3].lo, xv[s].hi, acc[j3], 0, 0, 0);
            acc[j3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[s][j3].hi, xv[s].lo, acc[j3], 0, 0, 0);
        }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j3 = 0; j3 < 3; ++j3) {      // x.hi w.mid, x.mid w.hi
            acc[j3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[s][j3].mid, xv[s].hi, acc[j3], 0, 0, 0);
            acc[j3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[s][j3].hi, xv[s].mid, acc[j3], 0, 0, 0);
        }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j3 = 0; j3 < 3; ++j3) accb[j3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn[s][j3].hi, xv[s].hi, accb[j3], 0, 0, 0);
    const int row = 16 * tile + nc;
    if (row < o.N) {
#pragma unroll
        for (int j3 = 0; j3 < 3; ++j3) {
            const int col0 = 16 * (3 * wave + j3) + 4 * nkb;
            const float4 v = make_float4(acc[j3][0] + accb[j3][0], acc[j3][1] + accb[j3][1], acc[j3][2] + accb[j3][2], acc[j3][3] + accb[j3][3]);
            if (col0 < o.m1) st4(o.xw + (size_t)row * o.m1 + col0, v);
            else if (col0 < o.m1 + 8) st4(o.a_ij + (size_t)row * 8 + (col0 - o.m1), v);
        }
    }
}

// host side: the arguments as the C entry points take them (all four or none)
struct NodeArgs { const void* pre; int cols; float* xw; float* a_ij; };

}  // namespace glam
