// Fused neighbour-gather / attention-logit / segment-softmax / weighted scatter-add kernels for
// TripletMessage and TripletMessageLight (reference: src_1gp/layer.py:40-55 and :88-97, executed
// there as PyG propagate -> message -> torch_scatter aggregate).
//
// Mapping (gfx950, wave64): a target node is owned by a GROUP of G consecutive lanes (G in
// {4,8,16}); lane l of the group owns the float4 channel chunks q = l + G*it (it < ITER) of EVERY
// head, so one 16-byte load per (edge, head, it) streams the neighbour row xw[src,h,:] and a
// wavefront processes 64/G nodes at once.  Degree-1..4 molecular segments and degree-100 protein
// segments take the same code path (the group walks its CSR segment serially; groups of one wave
// diverge only in trip count).  The attention logits are separable:
//   logit = leaky(a_i[dst] + <edge_attr[e], M> + a_j[src]),
// so no [E,H,3C] triplet tensor is ever formed; e_ij = edge_attr[e] @ W_edge is recomputed from the
// LDS-resident W_edge (De*H*Cp floats) instead of materialising ew[E,H*C].
// No atomics anywhere: forward reduces a CSR-by-target segment in registers, backward B1 walks the
// same segments, backward B2 walks the CSR transpose (by source).  Results are bit-reproducible.
#pragma once
#include "dense.h"

// occupancy targets (waves per SIMD) the register allocator is held to; tuned on MI355X
#ifndef GLAM_FWD_WAVES
#define GLAM_FWD_WAVES 1
#endif
#ifndef GLAM_FWD_CH
#define GLAM_FWD_CH 4
#endif
#ifndef GLAM_B1_CH
#define GLAM_B1_CH 4
#endif
#ifndef GLAM_B1_WAVES
#define GLAM_B1_WAVES 1
#endif

namespace glam {

#ifdef GLAM_B1_PROF   // developer aid (tools/b1_prof.py): where a B1 wave spends its cycles
__device__ long long g_b1_prof[1024 * 32];
#if GLAM_B1_PROF == 2     // stamps without draining the queues: the overlapped picture
#define B1_WAIT() do { } while (0)
#else
#define B1_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#define B1_STAMP(k) do { B1_WAIT(); const long long now__ = clock64(); prof_acc[k] += now__ - prof_last; prof_last = now__; } while (0)
#else
#define B1_STAMP(k) do { } while (0)
#endif

#ifdef GLAM_FWD_PROF   // developer aid (tools/fwd_prof.py): cycle breakdown of the fused forward's tiles
__device__ long long g_fwd_prof[512 * 8];
#define FWD_STAMP(k) do { if (tid == 0 && blockIdx.x < 512) { const long long n__ = clock64(); fp_acc[k] += n__ - fp_last; fp_last = n__; } } while (0)
#else
#define FWD_STAMP(k) do { } while (0)
#endif

struct FwdArgs {
    const float* xw; const float* a_ij; const float* edge_attr; const float* w_edge; const float* M;
    const int* rowptr; const int* nbr; const int* eid;
    int N; int Cp; float slope;
    float* aggr; float* stats;
    // optional fused update (G == 16 only): out[N,Cp] = aggr @ W_scale + bias, W_scale as a k_ts_gemm image
    const float* img_upd; const float* bias_p; float* out;
};

template <int DE>
__device__ __forceinline__ void load_edge_attr(const float* edge_attr, int id, float (&ea)[DE]) {
    const float* p = edge_attr + (size_t)id * DE;
#pragma unroll
    for (int i = 0; i < DE / 4; ++i) {
        float4 v = ld4(p + 4 * i);
        ea[4 * i + 0] = v.x; ea[4 * i + 1] = v.y; ea[4 * i + 2] = v.z; ea[4 * i + 3] = v.w;
    }
}

// pre-activation attention logit of one edge for every head
template <int H, int DE>
__device__ __forceinline__ void edge_pre(const float (&ai)[H], const float4 aj, const float (&ea)[DE],
                                         const float (&Mr)[DE][H], float (&pre)[H]) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float ee = 0.f;
#pragma unroll
        for (int k = 0; k < DE; ++k) ee = fmaf(ea[k], Mr[k][h], ee);
        pre[h] = ai[h] + ee + f4get(aj, h);
    }
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// e_ij chunk = sum_k edge_attr[e,k] * W_edge[k,h,chunk]  (W_edge in LDS)
template <int H, int DE>
__device__ __forceinline__ float4 edge_chunk(const float* s_w, const float (&ea)[DE], int h, int Cp, int q) {
    float4 e4 = f4zero();
#pragma unroll
    for (int k = 0; k < DE; ++k) fma4(e4, ea[k], ld4(s_w + (k * H + h) * Cp + q * 4));
    return e4;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int H, int G, int ITER, int DE, bool EMUL>
__global__ void __launch_bounds__(kBlock, GLAM_FWD_WAVES) k_triplet_fwd(FwdArgs a) {
    typedef XwRow XR;
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const int tid = threadIdx.x;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    constexpr int GPB = kBlock / G;
    // The first node's row pointers and a_i are requested BEFORE W_edge is staged: the staging round trip and the
    // barrier then overlap the first link of the node's dependent chain instead of preceding it.
    const int n_first = min((int)blockIdx.x * GPB + tid / G, a.N - 1);
    int beg_first = ldio(a.rowptr, (unsigned)n_first * 4u), end_first = ldio(a.rowptr, (unsigned)n_first * 4u + 4u);
    float4 aiv_first = ld4o(a.a_ij, (unsigned)n_first * 32u);
    if constexpr (EMUL) {
        for (int i = tid; i < DE * HC / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
        __syncthreads();
    }
    float Mr[DE][H];
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) Mr[k][h] = a.M[k * 4 + h];

    const int lg = tid % G;
    int q[ITER];
    bool ok[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        q[it] = lg + G * it;
        ok[it] = q[it] < Q;
        if (!ok[it]) q[it] = 0;
    }

    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;   // 32-bit byte offsets (see ld4o)
    const unsigned xrow_bytes = (unsigned)HC * XR::kElem, xhead_bytes = (unsigned)Cp * XR::kElem;   // of the gathered xw rows
    unsigned chunk_off[ITER], xchunk_off[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) { chunk_off[it] = (unsigned)q[it] * 16u; xchunk_off[it] = (unsigned)q[it] * 4u * XR::kElem; }
    // Edges are processed CH at a time with every load of a chunk in flight together (indices -> {edge
    // features, a_j, neighbour rows}): a degree <= CH segment (all molecular nodes) costs 3 dependent
    // memory round trips instead of 2 + 2*deg.
    constexpr int CH = ITER == 1 ? GLAM_FWD_CH : 2;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const bool fuse_upd = (G == 16) && a.img_upd != nullptr;
    const int LDT = HC + 4;                                    // LDS row pitch of the 16-node aggr tile
    float* s_tile = s_w + (EMUL ? DE * HC : 0);
    float* s_out = s_tile + 16 * LDT;
    float* s_img = s_out + 16 * 64;                            // update-GEMM weight image, resident for the whole block
    float4 bias_v = f4zero();          // this thread's four columns of the out tile: loaded once, not once per tile behind the stores
    if constexpr (G == 16) {
        // LDS-DMA: lands while the first tile is aggregated (the barrier before the first MFMA drains vmcnt)
        if (fuse_upd) {
            lds_copy_async<kBlock>(a.img_upd, s_img, ((HC + 15) >> 4) * 256, tid);
            if ((tid & 15) * 4 < Cp) bias_v = ld4(a.bias_p + (tid & 15) * 4);
        }
    }
#ifdef GLAM_FWD_PROF
    long long fp_acc[8] = {}, fp_last = clock64();
    const long long fp_t0 = fp_last;
#endif
    for (int base = blockIdx.x * GPB; base < a.N; base += gridDim.x * GPB) {
      const int n = base + tid / G;
      FWD_STAMP(0);
      if (n < a.N) {
        int beg, end;
        float4 aiv;
        if (base == (int)blockIdx.x * GPB) { beg = beg_first; end = end_first; aiv = aiv_first; }     // uniform per block
        else {
            beg = ldio(a.rowptr, (unsigned)n * 4u); end = ldio(a.rowptr, (unsigned)n * 4u + 4u);
            aiv = ld4o(a.a_ij, (unsigned)n * 32u);
        }
        float ai[H], m[H], ssum[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { ai[h] = f4get(aiv, h); m[h] = -INFINITY; ssum[h] = 0.f; }
        float4 acc[H][ITER];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int it = 0; it < ITER; ++it) acc[h][it] = f4zero();

        int sidx[CH], eidx[CH];
        bool val[CH];
        float ea[CH][DE], lk[CH][H];
        typename XR::T rows[CH][H][ITER];
        auto load_idx = [&](int e0) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                val[k] = e0 + k < end;
                const int e = val[k] ? e0 + k : end - 1;
                sidx[k] = ldio(a.nbr, (unsigned)e * 4u);
                eidx[k] = ldio(a.eid, (unsigned)e * 4u);
            }
        };
        auto load_rows = [&]() {
            // rows of invalid slots are loaded from the clamped (valid) index and never used: accumulate skips them
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const unsigned ro = (unsigned)sidx[k] * xrow_bytes;
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int it = 0; it < ITER; ++it) rows[k][h][it] = XR::load(a.xw, ro + xchunk_off[it] + (unsigned)h * xhead_bytes);
            }
        };
        auto load_logits = [&]() {
            float4 aj[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
#pragma unroll
                for (int u = 0; u < DE / 4; ++u) {
                    const float4 v = ld4o(a.edge_attr, (unsigned)eidx[k] * (unsigned)(DE * 4) + 16u * u);
                    ea[k][4 * u] = v.x; ea[k][4 * u + 1] = v.y; ea[k][4 * u + 2] = v.z; ea[k][4 * u + 3] = v.w;
                }
                aj[k] = ld4o(a.a_ij, (unsigned)sidx[k] * 32u + 16u);
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                float pre[H];
                edge_pre<H, DE>(ai, aj[k], ea[k], Mr, pre);
#pragma unroll
                for (int h = 0; h < H; ++h) lk[k][h] = leaky(pre[h], a.slope);
            }
        };
        auto take_max = [&]() {
#pragma unroll
            for (int k = 0; k < CH; ++k)
#pragma unroll
                for (int h = 0; h < H; ++h)
                    if (val[k]) m[h] = fmaxf(m[h], lk[k][h]);
        };
        auto accumulate = [&]() {
            // head-major: the W_edge chunk of one head (DE float4 from LDS) is live for one head only
#pragma unroll
            for (int h = 0; h < H; ++h) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    float4 wv[EMUL ? DE : 1];
                    if constexpr (EMUL) {
#pragma unroll
                        for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_w + (kk * H + h) * Cp + q[it] * 4);
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (!val[k]) continue;
                        const float p = softmax_exp(lk[k][h] - m[h]);
                        if (it == 0) ssum[h] += p;
                        float4 xj = XR::get(rows[k][h][it]);
                        if constexpr (EMUL) {
                            float4 e4 = f4zero();
#pragma unroll
                            for (int kk = 0; kk < DE; ++kk) fma4(e4, ea[k][kk], wv[kk]);
                            xj = e4 * xj;
                        }
                        fma4(acc[h][it], p, xj);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (end - beg <= CH) {
            if (end > beg) {
                load_idx(beg);
                load_rows();
                load_logits();
                take_max();
                accumulate();
            }
        } else {
            for (int e0 = beg; e0 < end; e0 += CH) { load_idx(e0); load_logits(); take_max(); }   // pass 1: segment max
            for (int e0 = beg; e0 < end; e0 += CH) { load_idx(e0); load_rows(); load_logits(); accumulate(); }
        }
        const unsigned orow = (unsigned)n * row_bytes;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float inv = 1.f / (ssum[h] + 1e-16f);
#pragma unroll
            for (int it = 0; it < ITER; ++it)
                if (ok[it]) {
                    const float4 v = inv * acc[h][it];
                    if (a.aggr) st4o(a.aggr, orow + chunk_off[it] + (unsigned)h * head_bytes, v);      // (NULL: the inference forward of the fused-update launch)
                    if constexpr (G == 16) {
                        if (fuse_upd) st4(s_tile + (tid / G) * LDT + h * Cp + q[it] * 4, v);
                    }
                }
        }
        if (lg == 0 && a.stats) {
            float4 mv = f4zero(), sv = f4zero();
            float* mp = &mv.x; float* sp = &sv.x;
#pragma unroll
            for (int h = 0; h < H; ++h) { mp[h] = (end > beg) ? m[h] : 0.f; sp[h] = ssum[h]; }
            st4o(a.stats, (unsigned)n * 32u, mv);
            st4o(a.stats, (unsigned)n * 32u + 16u, sv);
        }
      }
      if constexpr (G == 16) {
        // ---- fused update: out[16 nodes, Cp] = aggr_tile[16, HC] @ W_scale + bias on the fp32 matrix cores ----
        // wave w owns output column tile w (logical columns 4c + w); B fragments from the LDS-resident weight image
        // (one ds_read_b128 per 16-k group), A fragments from the LDS tile.  The launch is capped at two blocks per CU
        // and strides over the 16-node tiles, so the 48 KB image is fetched 512 times instead of once per tile.
        if (fuse_upd) {
            FWD_STAMP(1);
            __syncthreads();
            FWD_STAMP(2);
            const int wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
            const int GK = (HC + 15) >> 4;
            v4f cacc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int g0 = 0; g0 < GK; g0 += 4) {          // 4 k-groups (64 k values) per batch: 8 loads in flight
                float4 bf[4], af[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = g0 + u, k0 = 16 * g + 4 * kq;
                    bf[u] = g < GK ? ld4(s_img + ((4 * g + kq) * 64 + wave * 16 + c) * 4) : f4zero();
                    af[u] = (g < GK && k0 < HC) ? ld4(s_tile + c * LDT + k0) : f4zero();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(af[u], j), f4get(bf[u], j), cacc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_out[(kq * 4 + i) * 64 + 4 * c + wave] = cacc[i];
            FWD_STAMP(3);
            __syncthreads();
            FWD_STAMP(4);
            const int row = tid >> 4, c4 = (tid & 15) * 4;
            if (c4 < Cp && base + row < a.N) {
                float4 v = ld4(s_out + row * 64 + c4);
                v.x += bias_v.x; v.y += bias_v.y; v.z += bias_v.z; v.w += bias_v.w;
                st4(a.out + (size_t)(base + row) * Cp + c4, v);
            }
            FWD_STAMP(5);
        }
      }
    }
#ifdef GLAM_FWD_PROF
    if (tid == 0 && blockIdx.x < 512) {
        for (int k = 0; k < 6; ++k) g_fwd_prof[blockIdx.x * 8 + k] = fp_acc[k];
        g_fwd_prof[blockIdx.x * 8 + 6] = fp_t0;
        g_fwd_prof[blockIdx.x * 8 + 7] = clock64();
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// backward B1: walk CSR-by-target.  Per edge: recompute alpha, d_alpha = <d_aggr[dst], e_ij*xw[src]>,
// softmax + leaky backward -> dpre; accumulate d_a_i per node and d_W_edge / d_M per lane.
// ------------------------------------------------------------------------------------------------
struct BwdDstArgs {
    const float* xw; const float* a_ij; const float* edge_attr; const float* w_edge; const float* M;
    const float* aggr; const float* stats; const float* d_aggr;
    const int* rowptr; const int* nbr; const int* eid;
    int N; int Cp; float slope;
    float* alpha_e; float* dpre_e; float* d_a_ij; float* partial;
    int red_groups;   // rows of the LDS reduction buffer: kBlock / G (every lane group stores its own partial) or 4
    // optional fused prologue (FD variants: G == 16, ITER == 1): d_aggr[16-node tile] = d_out[tile, Cp] @ W_scale^T on the fp32
    // matrix cores, W_scale^T as a k_ts_gemm image (K = Cp, M = H*Cp).  d_aggr is then an OUTPUT (B2 reads it back).
    const float* img_dagg; const float* d_out; float* d_aggr_w;
};

template <int H, int G, int ITER, int DE, bool EMUL, bool FD = false>
__global__ void __launch_bounds__(kBlock, (FD && DE == 4) ? 2 : GLAM_B1_WAVES) k_triplet_bwd_dst(BwdDstArgs a) {
    typedef XwRow XR;
    typedef float v4f __attribute__((ext_vector_type(4)));
    static_assert(!FD || (G == 16 && ITER == 1), "the fused d_aggr prologue maps one 16-lane group to one MFMA tile row");
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int tid = threadIdx.x;
#ifdef GLAM_B1_PROF
    const long long prof_k0 = clock64();
#endif
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    const int WSZ = EMUL ? DE * HC : 0;          // floats of staged W_edge
    const int P = WSZ + DE * 4;                  // floats of one block partial: d_W_edge | d_M
    constexpr int GPB = kBlock / G;
    float* s_w = s_mem;
    float* s_red = s_mem + WSZ;                  // [4 waves][P]
    // FD: the weight image, the 16-node d_aggr tile and the A tile share the space of the reduction buffer, which is only used
    // after the node loop (every wave has passed the loop's last barrier by then)
    const int MP = HC <= 64 ? 64 : 192, GK = (Cp + 15) >> 4, LDT = HC + 4;
    float* s_img = s_red;
    float* s_tile = s_img + GK * 16 * MP;
    float* s_a = s_tile + 16 * LDT;              // d_out rows of the NEXT tile, [16][16 chunks], chunk j of row r at (j - r) & 15
    // The A operand (16 rows of d_out = 4 KB) is one LDS-DMA piece per wave, issued one tile ahead: no registers, and the round
    // trip hides under the previous tile's edge phase.  Rotating the chunks by the row index makes the fragment reads
    // (16 lanes = 16 rows, same logical chunk) hit 16 different bank groups.
    auto fetch_a = [&](int base) {
        const int wave = tid >> 6, lane = tid & 63, r = wave * 4 + (lane >> 4), j = lane & 15;
        const int row = min(base + r, a.N - 1), ch = (j + r) & 15;
        if (4 * ch < Cp) dma16(a.d_out, ((unsigned)row * (unsigned)Cp + 4u * ch) * 4u, lds_addr(s_a + wave * 256));
    };
    if constexpr (FD) {
        fetch_a((int)blockIdx.x * GPB);
        // image: global -> registers -> LDS with every load in flight at once (an LDS-DMA copy of 48 KB takes ~2x as long and
        // the first tile's MFMAs wait for it)
        constexpr int kMaxImg4 = 4 * 16 * 192 / 4 / kBlock;     // float4 per thread of the largest image (Cp = 64, H = 3)
        const int n4 = GK * 4 * MP;
        float4 buf[kMaxImg4];
#pragma unroll
        for (int i = 0; i < kMaxImg4; ++i) {
            const int idx = tid + i * kBlock;
            if (idx < n4) buf[i] = ld4(a.img_dagg + 4 * idx);
        }
        if constexpr (EMUL) {
            for (int i = tid; i < DE * HC / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
        }
#pragma unroll
        for (int i = 0; i < kMaxImg4; ++i) {
            const int idx = tid + i * kBlock;
            if (idx < n4) st4(s_img + 4 * idx, buf[i]);
        }
        __syncthreads();
    } else if constexpr (EMUL) {
        for (int i = tid; i < DE * HC / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
        __syncthreads();
    }
    float Mr[DE][H];
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) Mr[k][h] = a.M[k * 4 + h];

    const int lg = tid % G;
    int q[ITER];
    bool ok[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        q[it] = lg + G * it;
        ok[it] = q[it] < Q;
        if (!ok[it]) q[it] = 0;
    }

    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;   // 32-bit byte offsets (see ld4o)
    unsigned chunk_off[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) chunk_off[it] = (unsigned)q[it] * 16u;
    float4 dw[EMUL ? DE : 1][H][ITER];
    float dM[DE][H];
#pragma unroll
    for (int k = 0; k < (EMUL ? DE : 1); ++k)
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int it = 0; it < ITER; ++it) dw[k][h][it] = f4zero();
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) dM[k][h] = 0.f;

#ifdef GLAM_B1_PROF
    long long prof_acc[32] = {}, prof_last = clock64();
    const long long prof_t0 = prof_last;
#endif
    for (int base = blockIdx.x * GPB; base < a.N; base += gridDim.x * GPB) {
      const int n = base + tid / G;
      if constexpr (!FD) { if (n >= a.N) break; }
      if constexpr (FD) {
        B1_STAMP(8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's piece of the A tile has landed (issued a tile ago)
        B1_STAMP(9);
        __syncthreads();
        B1_STAMP(10);
      }
      // the node's own loads: with FD they fly under the MFMA phase
      const int nc = min(n, a.N - 1);
      const int beg = ldio(a.rowptr, (unsigned)nc * 4u), end = ldio(a.rowptr, (unsigned)nc * 4u + 4u);
      const float4 aiv = ld4o(a.a_ij, (unsigned)nc * 32u);
      const float4 mv = ld4o(a.stats, (unsigned)nc * 32u), sv = ld4o(a.stats, (unsigned)nc * 32u + 16u);
      // sum_e alpha_e * d_alpha_e of the softmax backward: a node whose edges fit ONE chunk (every atom of a molecule) sums the products
      // where they are computed, in the chunk loop below; only a node with more edges takes the identity
      // sum_e alpha_e d_alpha_e == <d_aggr[n,h,:], aggr[n,h,:]> and reads its aggr row for it (round 5: B1 no longer reads aggr for
      // molecular graphs, a quarter of its bytes; the warp-specialised kernel does the same in its quad lanes)
      constexpr int CH = ITER == 1 ? GLAM_B1_CH : 2;   // edges per chunk: all loads of a chunk in flight together
      const bool many = end - beg > CH;
      float4 dag[H][ITER], agr[H][ITER];
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
          for (int it = 0; it < ITER; ++it) {
              agr[h][it] = f4zero();
              if constexpr (!FD) dag[h][it] = ok[it] ? ld4o(a.d_aggr, (unsigned)nc * row_bytes + (unsigned)h * head_bytes + chunk_off[it]) : f4zero();
          }
      if (many) {                                               // (one branch around all of them: in flight together)
#pragma unroll
          for (int h = 0; h < H; ++h)
#pragma unroll
              for (int it = 0; it < ITER; ++it) agr[h][it] = ld4o(a.aggr, (unsigned)nc * row_bytes + (unsigned)h * head_bytes + chunk_off[it]);
      }
      if constexpr (FD) {
        // ---- d_aggr tile = d_out[base .. base+16, :] @ W_scale^T: wave w owns column tile t = w of every 64-column group ----
        const int wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
        B1_STAMP(16);
        float4 af[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 4 * g + kq;                          // logical 16-byte chunk of row c
            af[g] = (base + c < a.N && 4 * ch < Cp) ? ld4(s_a + c * 64 + ((ch - c) & 15) * 4) : f4zero();
        }
        // k-group outer, column group inner: H independent accumulator chains keep the matrix pipe busy (one chain of 16
        // dependent MFMAs stalls on its own latency), and only H B operands are live at a time
        v4f acc[H];
#pragma unroll
        for (int cg = 0; cg < H; ++cg) acc[cg] = (v4f){0.f, 0.f, 0.f, 0.f};
        B1_STAMP(17);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < GK) {
                float4 bv[H];
#pragma unroll
                for (int cg = 0; cg < H; ++cg)
                    bv[cg] = cg * 64 < HC ? ld4(s_img + ((4 * g + kq) * MP + cg * 64 + wave * 16 + c) * 4) : f4zero();
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cg = 0; cg < H; ++cg)
                        acc[cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(af[g], j), f4get(bv[cg], j), acc[cg], 0, 0, 0);
            }
        }
#ifdef GLAM_B1_PROF
        asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[H - 1]));   // the stamp waits for the matrix pipe
#endif
        B1_STAMP(18);
#pragma unroll
        for (int cg = 0; cg < H; ++cg) {
            const int mcol = cg * 64 + 4 * c + wave;
            if (mcol < HC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s_tile[(kq * 4 + i) * LDT + mcol] = acc[cg][i];
            }
        }
        B1_STAMP(11);
        __syncthreads();
        B1_STAMP(12);
#pragma unroll
        for (int h = 0; h < H; ++h) dag[h][0] = ok[0] ? ld4(s_tile + (tid / G) * LDT + h * Cp + q[0] * 4) : f4zero();
        __syncthreads();
        if (base + (int)gridDim.x * GPB < a.N) fetch_a(base + (int)gridDim.x * GPB);   // next tile's A rows (block-uniform)
        B1_STAMP(13);
      }
      if (n < a.N) {
        B1_STAMP(0);
        float ai[H], m[H], inv[H], dot[H], dai[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            ai[h] = f4get(aiv, h); m[h] = f4get(mv, h); inv[h] = 1.f / (f4get(sv, h) + 1e-16f); dai[h] = 0.f;
            float part = 0.f;
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                if constexpr (FD) {
                    const unsigned off = (unsigned)n * row_bytes + (unsigned)h * head_bytes + chunk_off[it];
                    if (ok[it]) st4o(a.d_aggr_w, off, dag[h][it]);
                }
                part += dot4(dag[h][it], agr[h][it]);
            }
            // (nodes with more than one chunk of edges) sum_e alpha_e * d_alpha_e == <d_aggr[n,h,:], aggr[n,h,:]>
            dot[h] = group_sum<G>(part);
        }
        B1_STAMP(1);
        for (int e0 = beg; e0 < end; e0 += CH) {
            int sidx[CH], eidx[CH];
            bool val[CH];
            float eav[CH][DE];
            float4 ajv[CH];
            typename XR::T rows[CH][H][ITER];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                val[k] = e0 + k < end;
                const int e = val[k] ? e0 + k : end - 1;
                sidx[k] = ldio(a.nbr, (unsigned)e * 4u);
                eidx[k] = ldio(a.eid, (unsigned)e * 4u);
            }
            B1_STAMP(2);
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const unsigned ro = (unsigned)sidx[k] * (row_bytes / 4u * XR::kElem);
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int it = 0; it < ITER; ++it)
                        rows[k][h][it] = val[k] ? XR::load(a.xw, ro + ((unsigned)h * head_bytes + chunk_off[it]) / 4u * XR::kElem) : XR::zero();
#pragma unroll
                for (int u = 0; u < DE / 4; ++u) {
                    const float4 v = ld4o(a.edge_attr, (unsigned)eidx[k] * (unsigned)(DE * 4) + 16u * u);
                    eav[k][4 * u] = v.x; eav[k][4 * u + 1] = v.y; eav[k][4 * u + 2] = v.z; eav[k][4 * u + 3] = v.w;
                }
                ajv[k] = ld4o(a.a_ij, (unsigned)sidx[k] * 32u + 16u);
            }
            B1_STAMP(3);
            float pre[CH][H], alpha[CH][H], dp[CH][H];
#pragma unroll
            for (int k = 0; k < CH; ++k) edge_pre<H, DE>(ai, ajv[k], eav[k], Mr, pre[k]);
            // head-major: the W_edge chunk of one head is live for one head only
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float part[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    part[k] = 0.f;
                    alpha[k][h] = softmax_exp(leaky(pre[k][h], a.slope) - m[h]) * inv[h];
                }
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    float4 wv[EMUL ? DE : 1];
                    if constexpr (EMUL) {
#pragma unroll
                        for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_w + (kk * H + h) * Cp + q[it] * 4);
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (!val[k]) continue;
                        const float4 t = dag[h][it] * XR::get(rows[k][h][it]);   // d_aggr * x_j
                        if constexpr (EMUL) {
                            float4 e4 = f4zero();
#pragma unroll
                            for (int kk = 0; kk < DE; ++kk) {
                                fma4_pk(e4, eav[k][kk], wv[kk]);
                                fma4_pk(dw[kk][h][it], eav[k][kk] * alpha[k][h], t);
                            }
                            part[k] += dot4(t, e4);
                        } else {
                            part[k] += t.x + t.y + t.z + t.w;
                        }
                    }
                }
                float dal[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) dal[k] = val[k] ? group_sum<G>(part[k]) : 0.f;
                float S = dot[h];
                if (!many) {                                    // ((p0 + p1) + p2) + p3, an absent edge adds an exact zero
                    S = val[0] ? alpha[0][h] * dal[0] : 0.f;
#pragma unroll
                    for (int k = 1; k < CH; ++k) S += val[k] ? alpha[k][h] * dal[k] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (!val[k]) continue;
                    const float dl = alpha[k][h] * (dal[k] - S);
                    dp[k][h] = pre[k][h] > 0.f ? dl : dl * a.slope;
                    dai[h] += dp[k][h];
#pragma unroll
                    for (int kk = 0; kk < DE; ++kk) dM[kk][h] = fmaf(eav[k][kk], dp[k][h], dM[kk][h]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            B1_STAMP(4);
            if (lg == 0) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (!val[k]) continue;
                    float4 av = f4zero(), dv = f4zero();
                    float* ap = &av.x; float* dpp = &dv.x;
#pragma unroll
                    for (int h = 0; h < H; ++h) { ap[h] = alpha[k][h]; dpp[h] = dp[k][h]; }
                    st4o(a.alpha_e, (unsigned)eidx[k] * 16u, av);
                    st4o(a.dpre_e, (unsigned)eidx[k] * 16u, dv);
                }
            }
        }
        if (lg == 0) {
            float4 dv = f4zero();
            float* dpp = &dv.x;
#pragma unroll
            for (int h = 0; h < H; ++h) dpp[h] = dai[h];
            st4o(a.d_a_ij, (unsigned)n * 32u, dv);
        }
        B1_STAMP(5);
      }
    }
#ifdef GLAM_B1_PROF
    if (tid == 0 && blockIdx.x < 1024) {
        for (int k = 0; k < 32; ++k) g_b1_prof[blockIdx.x * 32 + k] = prof_acc[k];
        g_b1_prof[blockIdx.x * 32 + 6] = prof_t0;
        g_b1_prof[blockIdx.x * 32 + 14] = prof_k0;
    }
#endif

    // ---- block partial of d_W_edge | d_M, summed in a fixed order ----
    const int wave = tid >> 6, lane = tid & 63;
    float* out = a.partial + (size_t)blockIdx.x * P;
    if (a.red_groups == GPB) {
        // every lane group parks its own partial in LDS (the host grants GPB rows when they fit): no cross-lane traffic
        float* red = s_red + (tid / G) * P;
        if constexpr (EMUL) {
#pragma unroll
            for (int k = 0; k < DE; ++k)
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int it = 0; it < ITER; ++it)
                        if (ok[it]) st4(red + (k * H + h) * Cp + q[it] * 4, dw[k][h][it]);
        }
        if (lg == 0) {
#pragma unroll
            for (int k = 0; k < DE; ++k) {
                float4 v = f4zero();
                float* vp = &v.x;
#pragma unroll
                for (int h = 0; h < H; ++h) vp[h] = dM[k][h];
                st4(red + WSZ + k * 4, v);
            }
        }
        __syncthreads();
        for (int i = tid; i < P; i += kBlock) {
            float sum = 0.f;
#pragma unroll 8
            for (int g = 0; g < GPB; ++g) sum += s_red[g * P + i];
            out[i] = sum;
        }
    } else {
        // wide layers: wave shuffle across groups, then 4 waves via LDS
        float* red = s_red + wave * P;
        if constexpr (EMUL) {
#pragma unroll
            for (int k = 0; k < DE; ++k)
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int it = 0; it < ITER; ++it) {
                        float4 v = dw[k][h][it];
                        v.x = cross_group_sum<G>(v.x); v.y = cross_group_sum<G>(v.y);
                        v.z = cross_group_sum<G>(v.z); v.w = cross_group_sum<G>(v.w);
                        if (lane < G && ok[it]) st4(red + (k * H + h) * Cp + q[it] * 4, v);
                    }
        }
#pragma unroll
        for (int k = 0; k < DE; ++k) {
            float4 v = f4zero();
            float* vp = &v.x;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                // every lane of a group holds the same dM; pick lane 0 of each group before the sum
                vp[h] = cross_group_sum<G>(dM[k][h]);
            }
            if (lane == 0) st4(red + WSZ + k * 4, v);
        }
        __syncthreads();
        for (int i = tid; i < P; i += kBlock)
            out[i] = (s_red[i] + s_red[P + i]) + (s_red[2 * P + i] + s_red[3 * P + i]);
    }
#ifdef GLAM_B1_PROF
    if (tid == 0 && blockIdx.x < 1024) g_b1_prof[blockIdx.x * 32 + 7] = clock64();
#endif
}

// ------------------------------------------------------------------------------------------------
// optional: gradient w.r.t. edge_attr (edge features are data in every reference configuration, so this
// only runs when a caller asks for it).  Runs after B1 (needs alpha_e / dpre_e):
//   d_edge_attr[e,k] = sum_h dpre[e,h] M[k,h] + sum_{h,c} alpha[e,h] d_aggr[dst,h,c] xw[src,h,c] W_edge[k,h,c]
// ------------------------------------------------------------------------------------------------
struct BwdDeaArgs {
    const float* xw; const float* w_edge; const float* M; const float* d_aggr; const float* alpha_e; const float* dpre_e;
    const int* rowptr; const int* nbr; const int* eid;
    int N; int Cp;
    float* d_edge_attr;
};

template <int H, int G, int ITER, int DE, bool EMUL>
__global__ void __launch_bounds__(kBlock) k_triplet_bwd_dea(BwdDeaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const int tid = threadIdx.x;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    if constexpr (EMUL) {
        for (int i = tid; i < DE * HC / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
        __syncthreads();
    }
    const int lg = tid % G;
    constexpr int GPB = kBlock / G;
    for (int n = blockIdx.x * GPB + tid / G; n < a.N; n += gridDim.x * GPB) {
        const int beg = a.rowptr[n], end = a.rowptr[n + 1];
        for (int e = beg; e < end; ++e) {
            const int sidx = a.nbr[e], id = a.eid[e];
            const float4 al = ld4(a.alpha_e + (size_t)id * 4), dpv = ld4(a.dpre_e + (size_t)id * 4);
            float dea[DE];
#pragma unroll
            for (int kk = 0; kk < DE; ++kk) dea[kk] = 0.f;
            if constexpr (EMUL) {
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int it = 0; it < ITER; ++it) {
                        const int qq = lg + G * it;
                        if (qq >= Q) continue;
                        const float4 t = ld4(a.d_aggr + (size_t)n * HC + h * Cp + qq * 4) *
                                         ld4(a.xw + (size_t)sidx * HC + h * Cp + qq * 4);
#pragma unroll
                        for (int kk = 0; kk < DE; ++kk)
                            dea[kk] = fmaf(f4get(al, h), dot4(t, ld4(s_w + (kk * H + h) * Cp + qq * 4)), dea[kk]);
                    }
            }
#pragma unroll
            for (int kk = 0; kk < DE; ++kk) {
                float v = EMUL ? group_sum<G>(dea[kk]) : 0.f;
#pragma unroll
                for (int h = 0; h < H; ++h) v = fmaf(f4get(dpv, h), a.M[kk * 4 + h], v);
                dea[kk] = v;
            }
            if (lg == 0) {
#pragma unroll
                for (int i = 0; i < DE / 4; ++i)
                    st4(a.d_edge_attr + (size_t)id * DE + 4 * i, make_float4(dea[4 * i], dea[4 * i + 1], dea[4 * i + 2], dea[4 * i + 3]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward B2: walk the CSR transpose (by source).  d_xw[j] = sum_{e: src=j} alpha_e * e_ij * d_aggr[dst],
// d_a_j[j] = sum dpre_e.
// ------------------------------------------------------------------------------------------------
struct BwdSrcArgs {
    const float* edge_attr; const float* w_edge; const float* d_aggr; const float* alpha_e; const float* dpre_e;
    const int* colptr; const int* nbr; const int* eid;
    int N; int Cp;
    float* d_xw; float* d_a_ij;
    // optional fused input gradient (G == 16 only): d_x[N,Cp] = [d_xw | d_a_i | d_a_j] @ Wcat^T, Wcat^T as a k_ts_gemm
    // image (K = HC + 8, 64 columns); d_a_i is read back from d_a_ij (written by B1)
    const float* img_dx; float* d_x;
};

template <int H, int G, int ITER, int DE, bool EMUL>
__global__ void __launch_bounds__(kBlock, GLAM_FWD_WAVES) k_triplet_bwd_src(BwdSrcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const int tid = threadIdx.x;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    if constexpr (EMUL) {
        for (int i = tid; i < DE * HC / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
        __syncthreads();
    }
    const int lg = tid % G;
    constexpr int GPB = kBlock / G;
    int q[ITER];
    bool ok[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        q[it] = lg + G * it;
        ok[it] = q[it] < Q;
        if (!ok[it]) q[it] = 0;
    }
    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;   // 32-bit byte offsets (see ld4o)
    unsigned chunk_off[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) chunk_off[it] = (unsigned)q[it] * 16u;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const bool fuse_dx = (G == 16) && a.img_dx != nullptr;
    const int KX = HC + 8;                                      // columns of [d_xw | d_a_i | d_a_j]
    const int LDT = KX + ((68 - (KX & 63)) & 63);               // LDS row pitch = 4 mod 64 words: conflict-free A reads
    float* s_tile = s_w + (EMUL ? DE * HC : 0);
    float* s_out = s_tile + 16 * LDT;
    float* s_img = s_out + 16 * 64;                             // Wcat^T weight image, resident for the whole block
    if constexpr (G == 16) {
        if (fuse_dx) lds_copy_async<kBlock>(a.img_dx, s_img, ((KX + 15) >> 4) * 256, tid);
    }
    for (int base = blockIdx.x * GPB; base < a.N; base += gridDim.x * GPB) {
      const int j = base + tid / G;
      if (j < a.N) {
        const int beg = ldio(a.colptr, (unsigned)j * 4u), end = ldio(a.colptr, (unsigned)j * 4u + 4u);
        // d_a_i (written by B1) for the d_x tile: requested here, in front of every store of the node — issued behind the d_xw stores
        // the load made the compiler wait for vmcnt(0), i.e. for those stores to land
        float4 dai_v = f4zero();
        if constexpr (G == 16) {
            if (fuse_dx && lg == 0) dai_v = ld4o(a.d_a_ij, (unsigned)j * 32u);
        }
        float4 acc[H][ITER];
        float4 daj = f4zero();
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int it = 0; it < ITER; ++it) acc[h][it] = f4zero();
        constexpr int CH = ITER == 1 ? 4 : 2;   // edges per chunk: all loads of a chunk in flight together
        for (int e0 = beg; e0 < end; e0 += CH) {
            int nidx[CH], eidx[CH];
            bool val[CH];
            float4 alv[CH], dpv[CH], rows[CH][H][ITER];
            float eav[CH][DE];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                val[k] = e0 + k < end;
                const int e = val[k] ? e0 + k : end - 1;
                nidx[k] = ldio(a.nbr, (unsigned)e * 4u);
                eidx[k] = ldio(a.eid, (unsigned)e * 4u);
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const unsigned ro = (unsigned)nidx[k] * row_bytes;
#pragma unroll
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int it = 0; it < ITER; ++it)
                        rows[k][h][it] = val[k] ? ld4o(a.d_aggr, ro + (unsigned)h * head_bytes + chunk_off[it]) : f4zero();
                alv[k] = ld4o(a.alpha_e, (unsigned)eidx[k] * 16u);
                dpv[k] = ld4o(a.dpre_e, (unsigned)eidx[k] * 16u);
                if constexpr (EMUL) {
#pragma unroll
                    for (int u = 0; u < DE / 4; ++u) {
                        const float4 v = ld4o(a.edge_attr, (unsigned)eidx[k] * (unsigned)(DE * 4) + 16u * u);
                        eav[k][4 * u] = v.x; eav[k][4 * u + 1] = v.y; eav[k][4 * u + 2] = v.z; eav[k][4 * u + 3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (val[k]) { daj.x += dpv[k].x; daj.y += dpv[k].y; daj.z += dpv[k].z; daj.w += dpv[k].w; }
#pragma unroll
            for (int h = 0; h < H; ++h) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    float4 wv[EMUL ? DE : 1];
                    if constexpr (EMUL) {
#pragma unroll
                        for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_w + (kk * H + h) * Cp + q[it] * 4);
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (!val[k]) continue;
                        float4 dg = rows[k][h][it];
                        if constexpr (EMUL) {
                            float4 e4 = f4zero();
#pragma unroll
                            for (int kk = 0; kk < DE; ++kk) fma4(e4, eav[k][kk], wv[kk]);
                            dg = e4 * dg;
                        }
                        fma4(acc[h][it], f4get(alv[k], h), dg);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned orow = (unsigned)j * row_bytes;
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int it = 0; it < ITER; ++it)
                if (ok[it]) {
                    st4o(a.d_xw, orow + (unsigned)h * head_bytes + chunk_off[it], acc[h][it]);
                    if constexpr (G == 16) {
                        if (fuse_dx) st4(s_tile + (tid / G) * LDT + h * Cp + q[it] * 4, acc[h][it]);
                    }
                }
        if (lg == 0) {
            st4o(a.d_a_ij, (unsigned)j * 32u + 16u, daj);
            if constexpr (G == 16) {
                if (fuse_dx) {
                    st4(s_tile + (tid / G) * LDT + HC, dai_v);
                    st4(s_tile + (tid / G) * LDT + HC + 4, daj);
                }
            }
        }
      }
      if constexpr (G == 16) {
        // ---- fused input gradient: d_x[16 nodes, Cp] = tile[16, HC+8] @ Wcat^T on the fp32 matrix cores ----
        // wave w owns output column tile w (logical columns 4c + w); B fragments from the LDS-resident weight image,
        // A fragments from the LDS tile (same scheme as the forward pass's fused update).
        if (fuse_dx) {
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63, c = lane & 15, kq = lane >> 4;
            const int GK = (KX + 15) >> 4;
            const bool rok = base + c < a.N;
            v4f cacc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int g0 = 0; g0 < GK; g0 += 4) {          // 4 k-groups (64 k values) per batch: 8 loads in flight
                float4 bf[4], af[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = g0 + u, k0 = 16 * g + 4 * kq;
                    bf[u] = g < GK ? ld4(s_img + ((4 * g + kq) * 64 + wave * 16 + c) * 4) : f4zero();
                    af[u] = (g < GK && k0 < KX && rok) ? ld4(s_tile + c * LDT + k0) : f4zero();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(af[u], jj), f4get(bf[u], jj), cacc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_out[(kq * 4 + i) * 64 + 4 * c + wave] = cacc[i];
            __syncthreads();
            const int row = tid >> 4, c4 = (tid & 15) * 4;
            if (c4 < Cp && base + row < a.N) st4(a.d_x + (size_t)(base + row) * Cp + c4, ld4(s_out + row * 64 + c4));
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
template <int H, int G, int ITER, int DE, bool EMUL>
struct FwdOp {
    static void run(const FwdArgs& a, int grid, size_t lds, hipStream_t s) {
        static bool big_lds = false;      // > 64 KB of dynamic LDS (fused update: resident weight image) is opted into once
        if (lds > 64 * 1024 && !big_lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_triplet_fwd<H, G, ITER, DE, EMUL>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            big_lds = true;
        }
        hipLaunchKernelGGL((k_triplet_fwd<H, G, ITER, DE, EMUL>), dim3(grid), dim3(kBlock), lds, s, a);
    }
};
template <int H, int G, int ITER, int DE, bool EMUL>
struct BwdDstOp {
    static void run(const BwdDstArgs& a, int grid, size_t lds, hipStream_t s) {
        if constexpr (G == 16 && ITER == 1 && EMUL) {
            if (a.img_dagg) {
                static bool big_lds = false;
                if (lds > 64 * 1024 && !big_lds) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_triplet_bwd_dst<H, G, ITER, DE, EMUL, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                    big_lds = true;
                }
                hipLaunchKernelGGL((k_triplet_bwd_dst<H, G, ITER, DE, EMUL, true>), dim3(grid), dim3(kBlock), lds, s, a);
                return;
            }
        }
        hipLaunchKernelGGL((k_triplet_bwd_dst<H, G, ITER, DE, EMUL>), dim3(grid), dim3(kBlock), lds, s, a);
    }
};
template <int H, int G, int ITER, int DE, bool EMUL>
struct BwdDeaOp {
    static void run(const BwdDeaArgs& a, int grid, size_t lds, hipStream_t s) {
        hipLaunchKernelGGL((k_triplet_bwd_dea<H, G, ITER, DE, EMUL>), dim3(grid), dim3(kBlock), lds, s, a);
    }
};
template <int H, int G, int ITER, int DE, bool EMUL>
struct BwdSrcOp {
    static void run(const BwdSrcArgs& a, int grid, size_t lds, hipStream_t s) {
        static bool big_lds = false;
        if (lds > 64 * 1024 && !big_lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_triplet_bwd_src<H, G, ITER, DE, EMUL>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            big_lds = true;
        }
        hipLaunchKernelGGL((k_triplet_bwd_src<H, G, ITER, DE, EMUL>), dim3(grid), dim3(kBlock), lds, s, a);
    }
};

struct Shape { int G, ITER; };
// smallest G*ITER that covers Q = Cp/4 chunks per head
inline bool pick_shape(int Q, Shape* s) {
    if (Q <= 4) *s = {4, 1};
    else if (Q <= 8) *s = {8, 1};        // (16 lanes with 8 idle measured slower at C = 30: 106.7 vs 104.4 us/step)
    else if (Q <= 16) *s = {16, 1};      // Q = 9..12 (C = 45): 16 lanes with 4 idle + the fused GEMM epilogues beat 4 lanes x 3
                                         // chunks (119 vs 151 us/step)
    else if (Q <= 24) *s = {8, 3};
    else if (Q <= 32) *s = {16, 2};
    else if (Q <= 64) *s = {16, 4};
    else return false;
    return true;
}

template <template <int, int, int, int, bool> class Op, int H, int DE, bool EMUL, typename Args>
inline bool dispatch_shape(Shape sh, const Args& a, int nodes, size_t lds, hipStream_t s, int cap, int* grid_out) {
#define GLAM_CASE(G_, IT_)                                                   \
    if (sh.G == G_ && sh.ITER == IT_) {                                      \
        const int grid = grid_for(nodes, kBlock / G_, cap);                  \
        if (grid_out) *grid_out = grid;                                      \
        Op<H, G_, IT_, DE, EMUL>::run(a, grid, lds, s);                      \
        return true;                                                         \
    }
    GLAM_CASE(4, 1) GLAM_CASE(8, 1) GLAM_CASE(4, 3) GLAM_CASE(16, 1) GLAM_CASE(8, 3) GLAM_CASE(16, 2) GLAM_CASE(16, 4)
#undef GLAM_CASE
    return false;
}


// One translation unit per head count (triplet_h1..h4.hip) instantiates the kernels; these are their entry
// points: kind 0 = forward, 1 = backward by target (B1), 2 = backward by source (B2), 3 = d_edge_attr.
enum { kTripletFwd = 0, kTripletBwdDst = 1, kTripletBwdSrc = 2, kTripletBwdDea = 3 };
#define GLAM_DECLARE_TRIPLET_H(HH)                                                                                  \
    bool triplet_launch_h##HH(int kind, int De, int emul, Shape sh, const void* args, int nodes, size_t lds,        \
                              hipStream_t s, int cap, int* grid_out);
GLAM_DECLARE_TRIPLET_H(1) GLAM_DECLARE_TRIPLET_H(2) GLAM_DECLARE_TRIPLET_H(3) GLAM_DECLARE_TRIPLET_H(4)
#undef GLAM_DECLARE_TRIPLET_H

template <int HH>
inline bool triplet_launch_impl(int kind, int De, int emul, Shape sh, const void* args, int nodes, size_t lds, hipStream_t s,
                                int cap, int* grid_out) {
#define GLAM_KIND(K_, Op_, Args_)                                                                                         \
    if (kind == K_) {                                                                                                     \
        const Args_& a = *static_cast<const Args_*>(args);                                                                \
        if (emul && De == 4) return dispatch_shape<Op_, HH, 4, true>(sh, a, nodes, lds, s, cap, grid_out);               \
        if (emul && De == 8) return dispatch_shape<Op_, HH, 8, true>(sh, a, nodes, lds, s, cap, grid_out);               \
        if constexpr (HH == 1) {  /* TripletMessageLight / GAT: single head, message alpha * x_j */                       \
            if (!emul && De == 4) return dispatch_shape<Op_, 1, 4, false>(sh, a, nodes, lds, s, cap, grid_out);          \
            if (!emul && De == 8) return dispatch_shape<Op_, 1, 8, false>(sh, a, nodes, lds, s, cap, grid_out);          \
        }                                                                                                                 \
        return false;                                                                                                     \
    }
    GLAM_KIND(kTripletFwd, FwdOp, FwdArgs) GLAM_KIND(kTripletBwdDst, BwdDstOp, BwdDstArgs)
    GLAM_KIND(kTripletBwdSrc, BwdSrcOp, BwdSrcArgs) GLAM_KIND(kTripletBwdDea, BwdDeaOp, BwdDeaArgs)
#undef GLAM_KIND
    return false;
}

}  // namespace glam
