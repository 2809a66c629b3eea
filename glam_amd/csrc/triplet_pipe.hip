// Software-pipelined forward aggregate for low-degree (molecular) graphs — glam_triplet_fwd_ell, the aggregate-only operator's kernel
// beyond the LLC — and the ELL index records (glam_ell_build) every molecular kernel of the library reads, the warp-specialised ones
// included (triplet_ws*.hip).  Same arithmetic as k_triplet_fwd (triplet_kernels.h; reference: src_1gp/layer.py:42-55 executed
// through PyG propagate -> message -> scatter-add), bit-identical results (same operation order per lane).
//
// Why: the general kernel holds every in-flight row in VGPRs, so a wave can have ONE node pass in flight, and its three dependent
// memory phases (row pointers -> indices -> rows) sit in front of every pass: measured, it behaves like t = t_VALU + t_memory (59 +
// 80 us at B = 16 384), i.e. with no overlap.  Here every wave runs a three-deep pipeline over its own passes (4 nodes each, one
// 16-lane group per node):
//     iteration p:   wait vmcnt(0)                 -> rows of pass p are in registers, the index record of pass p+1 too
//                    store the results of pass p-1  (held in 14 registers across the wait: a store issued before it would be waited for)
//                    issue the row loads + the side-table piece of pass p+1, then the index-record load of pass p+2
//                    compute pass p (logits, segment softmax, weighted sum) while both are in flight
// Index records: ELL tables built once per edge list next to the CSR (src[4] | eid[4] per node, -1 = empty slot); graphs with an
// in-degree above 4 keep the general kernel (the host asks glam_ell_build's overflow flag once per edge list).
// (Round 4 removed the siblings that no route selected any more — the all-LDS-DMA gather k_triplet_fwd_dma, the barrier-coupled fused
// forward k_triplet_fwd_pipe<FUSE> and the pipelined backward by source k_triplet_bwd_src_pipe: molecular graphs with one-hot bond
// features run the warp-specialised kernels at every size.  History and measurements: DESIGN.md §4.)
#include "triplet_pipe.h"

#include <stdlib.h>
#include <string.h>

namespace glam {

#ifdef GLAM_DMA_PROF
__device__ long long g_dma_prof[64 * 8];
#define DSTAMP(k) do { const long long now__ = clock64(); pacc[k] += now__ - plast; plast = now__; } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// k_triplet_fwd_pipe: the same software pipeline with the gathered ROWS prefetched one pass ahead into VGPRs (plain 16-byte loads:
// L1 / L2 hits move at 64 B/clk/CU) and only the small side table (a_j, edge_attr, a_i: one 64-lane piece per pass) staged by
// LDS-DMA.  Cycle stamps of the all-DMA version (tools/dma_prof.py, B = 16 384, per pass and wave): 1 884 cycles issuing the nine
// 1 KiB gather pieces (~190 cycles each: the LDS-DMA path moves ~10-13 B/clk/CU — sized for an HBM stream, and this gather
// re-reads every row 2.05 times out of L2), 2 607 computing, 8 waiting.  Registers: +48 for the rows in flight (2 waves / SIMD).
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef GLAM_PIPE_WAVES
#define GLAM_PIPE_WAVES 2      // 3 (<= 168 VGPRs) spills: 117 vs 112 us at B = 16 384
#endif
template <int H, int DE, bool ONEHOT>
__global__ void __launch_bounds__(kBlock, GLAM_PIPE_WAVES) k_triplet_fwd_pipe(FwdDmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane >> 4, q = lane & 15;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    const int WSZ = DE * HC;
    constexpr int kMetaF = 64 * 4;
    float* s_w = smem;
    float* wbase = smem + WSZ + wave * (2 * kMetaF);
    for (int i = tid; i < WSZ / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
    __syncthreads();
    float Mr[DE][H];
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) Mr[k][h] = a.M[k * 4 + h];

    const int npass = (a.N + 3) >> 2;
    const int gw = blockIdx.x * (kBlock / 64) + wave, GW = gridDim.x * (kBlock / 64);
    const bool qok = q < Q;
    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;
    const unsigned qoff = (unsigned)(qok ? q : 0) * 16u;
    constexpr int kEaLanes = kMetaSlots * (DE / 4);
    const int mt_kind = lane < kMetaSlots ? 0 : lane < kMetaSlots + kEaLanes ? 1 : lane < kMetaSlots + kEaLanes + 4 ? 2 : 3;
    const int mt_slot = mt_kind == 0 ? lane : mt_kind == 1 ? (lane - kMetaSlots) / (DE / 4) : 0;
    const unsigned mt_sub = mt_kind == 1 ? (unsigned)((lane - kMetaSlots) % (DE / 4)) * 16u : 0u;
    constexpr int CH = 4;

    auto load_rec = [&](int pass, int& rs, int& re) {
        const int n = 4 * pass + j;
        rs = -1; re = -1;
        if (q < 4 && pass < npass && n < a.N) { rs = a.ell_src[4 * n + q]; re = a.ell_eid[4 * n + q]; }
    };
    // issue everything pass `pass` needs: its rows into `rows` (registers), its side table into LDS buffer `sel`
    auto prefetch = [&](int pass, int rs, int re, int sel, float4 (&rows)[CH][H]) -> PassMeta {
        const unsigned long long bal = __ballot(rs >= 0);
        const int d0 = __popc((unsigned)(bal & 0xF)), d1 = __popc((unsigned)((bal >> 16) & 0xF)),
                  d2 = __popc((unsigned)((bal >> 32) & 0xF)), d3 = __popc((unsigned)((bal >> 48) & 0xF));
        PassMeta pm;
        pm.deg = j == 0 ? d0 : j == 1 ? d1 : j == 2 ? d2 : d3;
        pm.off = j == 0 ? 0 : j == 1 ? d0 : j == 2 ? d0 + d1 : d0 + d1 + d2;
        pm.tot = __builtin_amdgcn_readfirstlane(d0 + d1 + d2 + d3);
        pm.dmax = __builtin_amdgcn_readfirstlane(max(max(d0, d1), max(d2, d3)));   // provably scalar: the slot loops branch on it
        if (pm.tot == 0) return pm;
        // side table piece: packed slot t is owned by lane 16 g + (t - off_g)
        const int t = min(mt_slot, pm.tot - 1);
        const int og = t < d0 ? 0 : t < d0 + d1 ? 1 : t < d0 + d1 + d2 ? 2 : 3;
        const int ooff = og == 0 ? 0 : og == 1 ? d0 : og == 2 ? d0 + d1 : d0 + d1 + d2;
        const int owner = 16 * og + (t - ooff);
        // every cross-lane read of the record in flight at once (six ds_bpermute, ONE wait): behind the scalar slot branches below the
        // compiler would issue them one dependent round trip at a time
        const int m_src = __shfl(rs, owner, 64), m_eid = __shfl(re, owner, 64);
        int sk[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) sk[k] = __shfl(rs, 16 * j + k, 64);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sk[0]), "+v"(sk[1]), "+v"(sk[2]), "+v"(sk[3]) : : "memory");
        const unsigned dst = lds_addr(wbase + sel * kMetaF);
        const int n_i = min(4 * pass + (lane - kMetaSlots - kEaLanes), a.N - 1);
        const unsigned off = mt_kind == 1 ? (unsigned)m_eid * (unsigned)(DE * 4) + mt_sub
                           : mt_kind == 2 ? (unsigned)max(n_i, 0) * 32u : (unsigned)m_src * 32u + 16u;
        if (mt_kind == 1) dma16(a.edge_attr, off, dst);
        else dma16(a.a_ij, off, dst);
        // rows of this lane's node: slot k's source sits in lane 16 j + k; an empty slot re-reads the first edge's row (weight 0)
        // (the rows are "settled" by input-only asm uses: an in/out operand split their live ranges, and the compiler then loaded
        // into temporaries and copied them home behind an s_waitcnt vmcnt right after the issue: 3 800 cycles per pass)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (k < pm.dmax) {                            // scalar branch: slot k is empty in all four nodes otherwise
                const unsigned ro = (unsigned)max(sk[k] >= 0 ? sk[k] : sk[0], 0) * row_bytes + qoff;
#pragma unroll
                for (int h = 0; h < H; ++h) rows[k][h] = ld4o(a.xw, ro + (unsigned)h * head_bytes);
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) rows[k][h] = f4zero();
            }
        }
        return pm;
    };

    float4 r_acc[H];
    float4 r_m = f4zero(), r_s = f4zero();
    int r_n = -1;
    auto compute = [&](int pass, const PassMeta& pm, int sel, const float4 (&rows)[CH][H]) {
        const int n = 4 * pass + j;
        if (n >= a.N || pass >= npass) { r_n = -1; return; }
        r_n = n;
        const float* meta = wbase + sel * kMetaF;
        float m[H], ssum[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { m[h] = -INFINITY; ssum[h] = 0.f; r_acc[h] = f4zero(); }
        if (pm.deg > 0) {
            const float4 aiv = ld4(meta + (kMetaSlots + kEaLanes + j) * 4);
            float ai[H];
#pragma unroll
            for (int h = 0; h < H; ++h) ai[h] = f4get(aiv, h);
            bool val[CH];
            float ea[CH][DE], lk[CH][H];
            float4 aj[CH];
            int wrow[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                val[k] = k < pm.deg;
                if (k < pm.dmax) {
                    const int slot = pm.off + (val[k] ? k : 0);
                    aj[k] = ld4(meta + slot * 4);
#pragma unroll
                    for (int u = 0; u < DE / 4; ++u) {
                        const float4 v = ld4(meta + (kMetaSlots + slot * (DE / 4) + u) * 4);
                        ea[k][4 * u] = v.x; ea[k][4 * u + 1] = v.y; ea[k][4 * u + 2] = v.z; ea[k][4 * u + 3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (k < pm.dmax) {
                    float pre[H];
                    edge_pre<H, DE>(ai, aj[k], ea[k], Mr, pre);
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        lk[k][h] = leaky(pre[h], a.slope);
                        m[h] = val[k] ? fmaxf(m[h], lk[k][h]) : m[h];
                    }
                    if constexpr (ONEHOT) {
                        int t = 0;
#pragma unroll
                        for (int kk = 1; kk < DE; ++kk) t = ea[k][kk] != 0.f ? kk : t;
                        wrow[k] = t * HC + (qok ? q : 0) * 4;
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float4 wv[DE], er[CH];
                if constexpr (!ONEHOT) {
#pragma unroll
                    for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_w + (kk * H + h) * Cp + (qok ? q : 0) * 4);
                } else {
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (k < pm.dmax) er[k] = ld4(s_w + wrow[k] + h * Cp);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (k < pm.dmax) {
                        const float p = val[k] ? softmax_exp(lk[k][h] - m[h]) : 0.f;
                        ssum[h] += p;
                        float4 e4;
                        if constexpr (ONEHOT) {
                            e4 = er[k];
                        } else {
                            e4 = f4zero();
#pragma unroll
                            for (int kk = 0; kk < DE; ++kk) fma4(e4, ea[k][kk], wv[kk]);
                        }
                        const float4 xj = e4 * rows[k][h];
                        fma4(r_acc[h], p, xj);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float inv = 1.f / (ssum[h] + 1e-16f);
            r_acc[h] = inv * r_acc[h];
            (&r_m.x)[h] = pm.deg > 0 ? m[h] : 0.f;
            (&r_s.x)[h] = ssum[h];
        }
    };
    auto store_results = [&]() {
        if (r_n < 0) return;
        if (qok) {
            const unsigned orow = (unsigned)r_n * row_bytes + (unsigned)q * 16u;
#pragma unroll
            for (int h = 0; h < H; ++h) st4o(a.aggr, orow + (unsigned)h * head_bytes, r_acc[h]);
        }
        if (q < 2) st4o(a.stats, (unsigned)r_n * 32u + (unsigned)q * 16u, q == 0 ? r_m : r_s);   // one store piece for both halves
        r_n = -1;
    };
    // the rows in flight are (re)defined by an empty asm right after the pipeline's own vmcnt(0): the compiler retires its count of
    // those loads there, and never again behind the stores / loads issued later in the iteration
    auto settle = [&](float4 (&rows)[CH][H]) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int h = 0; h < H; ++h)
                asm volatile("" : : "v"(rows[k][h].x), "v"(rows[k][h].y), "v"(rows[k][h].z), "v"(rows[k][h].w));   // a USE: no new live range
    };

    float4 rows_a[CH][H], rows_b[CH][H];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) { rows_a[k][h] = f4zero(); rows_b[k][h] = f4zero(); }
    int rs_nxt, re_nxt;
    int pass = gw;
    load_rec(pass, rs_nxt, re_nxt);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
    PassMeta pm_cur = prefetch(pass, rs_nxt, re_nxt, 0, rows_a);
    load_rec(pass + GW, rs_nxt, re_nxt);
    // two passes per trip: the register sets swap roles instead of being copied
#ifdef GLAM_DMA_PROF
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = clock64();
#endif
    const int pass_end = ((a.N + 15) >> 4) << 2;         // whole 16-node tiles: the four waves of a block leave the loop together
    for (; pass - wave < pass_end; pass += 2 * GW) {
        DSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_a);
        DSTAMP(1);
        store_results();
        DSTAMP(2);
        PassMeta pm_nxt = prefetch(pass + GW, rs_nxt, re_nxt, 1, rows_b);
        DSTAMP(3);
        load_rec(pass + 2 * GW, rs_nxt, re_nxt);
        DSTAMP(4);
        compute(pass, pm_cur, 0, rows_a);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DSTAMP(5);

        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_b);
        store_results();
        pm_cur = prefetch(pass + 2 * GW, rs_nxt, re_nxt, 0, rows_a);
        load_rec(pass + 3 * GW, rs_nxt, re_nxt);
        if (pass + GW - wave < pass_end) {
            compute(pass + GW, pm_nxt, 1, rows_b);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DSTAMP(6);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_results();
#ifdef GLAM_DMA_PROF
    if (lane == 0 && gw < 64) for (int k = 0; k < 8; ++k) g_dma_prof[gw * 8 + k] = pacc[k];
#endif
}

template <int H, int DE, bool ONEHOT>
static void launch_pipe(const FwdDmaArgs& a, int grid, hipStream_t s) {
    const size_t lds = ((size_t)DE * H * a.Cp + (size_t)(kBlock / 64) * 2 * 64 * 4) * sizeof(float);
    GLAM_PROF_LABEL("k_triplet_fwd_pipe");
    hipLaunchKernelGGL((k_triplet_fwd_pipe<H, DE, ONEHOT>), dim3(grid), dim3(kBlock), lds, s, a);
}

__global__ void __launch_bounds__(kBlock) k_ell_build(const int* rowptr, const int* nbr, const int* eid, int N, int4* ell_src,
                                                     int4* ell_eid, int* overflow) {   // one int4 per node and table
    for (int n = blockIdx.x * kBlock + threadIdx.x; n < N; n += gridDim.x * kBlock) {
        const int beg = rowptr[n], deg = rowptr[n + 1] - beg;
        int s[4] = {-1, -1, -1, -1}, e[4] = {-1, -1, -1, -1};
        for (int k = 0; k < min(deg, 4); ++k) { s[k] = nbr[beg + k]; e[k] = eid[beg + k]; }
        ell_src[n] = make_int4(s[0], s[1], s[2], s[3]);
        ell_eid[n] = make_int4(e[0], e[1], e[2], e[3]);
        if (deg > 4) *overflow = 1;
    }
}

}  // namespace glam

using namespace glam;

extern "C" int glam_ell_build(const int32_t* rowptr, const int32_t* nbr, const int32_t* eid, int64_t N, int32_t* ell_src,
                              int32_t* ell_eid, int32_t* overflow_flag, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_ell_build: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(rowptr && ell_src && ell_eid && overflow_flag && aligned16(ell_src) && aligned16(ell_eid), "glam_ell_build: null / misaligned pointer");
    hipLaunchKernelGGL(k_ell_build, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, rowptr, nbr, eid, (int)N,
                       reinterpret_cast<int4*>(ell_src), reinterpret_cast<int4*>(ell_eid), overflow_flag);
    GLAM_LAUNCH_CHECK("glam_ell_build");
    return GLAM_OK;
}

extern "C" int glam_triplet_fwd_ell_supported(int H, int Cp, int De) {
    return H >= 1 && H <= 4 && Cp >= 4 && Cp <= 64 && (Cp & 3) == 0 && (De == 4 || De == 8);
}

extern "C" int glam_triplet_fwd_ell(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                                    const int32_t* ell_src, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De,
                                    float slope, int edge_onehot, float* aggr, float* stats, int grid_blocks, void* stream) {
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX && E >= 0, "glam_triplet_fwd_ell: N / E out of range");
    if (!glam_triplet_fwd_ell_supported(H, Cp, De))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd_ell: H=%d Cp=%d De=%d outside the kernel table (Cp <= 64, H <= 4, De in {4, 8})", H, Cp, De);
    if ((uint64_t)N * H * Cp * 4 >= (1ull << 32) || (uint64_t)E * De * 4 >= (1ull << 32))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd_ell: a tensor exceeds 4 GiB (32-bit offsets)");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(xw && a_ij && w_edge && M && ell_src && ell_eid && aggr && stats && (E == 0 || edge_attr), "glam_triplet_fwd_ell: null pointer");
    GLAM_REQUIRE(aligned16(xw) && aligned16(a_ij) && aligned16(edge_attr) && aligned16(w_edge) && aligned16(aggr) && aligned16(stats) &&
                     aligned16(ell_src) && aligned16(ell_eid), "glam_triplet_fwd_ell: pointers must be 16-byte aligned");
    FwdDmaArgs a{xw, a_ij, edge_attr, w_edge, M, ell_src, ell_eid,
                 (int)N, Cp, slope, aggr, stats, nullptr, nullptr, nullptr};
    const int npass = (int)((N + 3) / 4);
    int grid = grid_blocks > 0 ? grid_blocks : 512;                 // two 4-wave blocks per CU, every wave pipelines over its passes
    if (grid > (npass + 3) / 4) grid = (npass + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
#define GLAM_PIPE_CASE(HH, DD)                                                                \
    if (H == HH && De == DD) {                                                            \
        if (edge_onehot) launch_pipe<HH, DD, true>(a, grid, s);                           \
        else launch_pipe<HH, DD, false>(a, grid, s);                                      \
        GLAM_LAUNCH_CHECK("glam_triplet_fwd_ell");                                        \
        return GLAM_OK;                                                                   \
    }
    GLAM_PIPE_CASE(1, 4) GLAM_PIPE_CASE(2, 4) GLAM_PIPE_CASE(3, 4) GLAM_PIPE_CASE(4, 4)
    GLAM_PIPE_CASE(1, 8) GLAM_PIPE_CASE(2, 8) GLAM_PIPE_CASE(3, 8) GLAM_PIPE_CASE(4, 8)
#undef GLAM_PIPE_CASE
    return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd_ell: no kernel for H=%d De=%d", H, De);
}
