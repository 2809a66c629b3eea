// Software-pipelined forward aggregate for low-degree (molecular) graphs: the same arithmetic as k_triplet_fwd
// (triplet_kernels.h; reference: src_1gp/layer.py:42-55 executed through PyG propagate -> message -> scatter-add), with the
// gathered neighbour rows travelling global -> LDS by LDS-DMA (global_load_lds_dwordx4: per-lane source address, lane-linear
// LDS destination, NO destination registers) instead of global -> VGPR.
//
// Why: the register-staged kernel holds every in-flight row in VGPRs (48 of its 152), so a wave can have ONE node pass in
// flight, and its three dependent memory phases (row pointers -> indices -> rows) sit in front of every pass: measured, the
// kernel behaves like t = t_VALU + t_memory (59 + 80 us at B = 16 384), i.e. with no overlap.  Here every wave runs a
// three-deep pipeline over its own passes (4 nodes each, one 16-lane group per node):
//     iteration p:   wait vmcnt(0)                 -> rows of pass p are in LDS, the index record of pass p+1 is in registers
//                    store the results of pass p-1  (held in 14 registers across the wait: a store issued before it would be waited for)
//                    issue the LDS-DMA gather of pass p+1 into the other buffer, then the index-record load of pass p+2
//                    compute pass p out of LDS (logits, segment softmax, weighted sum) while both are in flight
// The dependent chain per pass is one trip (the record) + one trip (the rows), both hidden behind a whole pass of arithmetic.
// Index records: ELL tables built once per edge list next to the CSR (src[4] | eid[4] per node, -1 = empty slot); graphs with
// an in-degree above 4 keep the general kernel (the host asks glam_ell_build's overflow flag once per edge list).
// LDS per wave: 2 buffers x kEC edge rows x (H*Q + 1 + DE/4) 16-byte chunks [row chunks | a_j | edge_attr] + 2 x 4 a_i; a pass
// whose four nodes have more than kEC edges together (4 x 4 = 16 is the worst case; 3e-5 of the molecular passes) is processed
// node by node without prefetch.  Results are bit-identical to k_triplet_fwd (same operation order per lane).
#include "triplet_kernels.h"

namespace glam {

constexpr int kEC = 12;                  // edge rows per pass buffer

struct FwdDmaArgs {
    const float* xw; const float* a_ij; const float* edge_attr; const float* w_edge; const float* M;
    const int* ell_src; const int* ell_eid;      // [N][4] each
    int N; int Cp; float slope;
    float* aggr; float* stats;
};

struct PassMeta { int deg; int off; int tot; };

// One LDS-DMA piece: 64 lanes x 16 bytes, lane l lands at lds_base + 16 l (lds_base wave-uniform, in M0).  Written as inline
// assembly on purpose: through the builtin the compiler knows the instruction writes LDS and — unable to prove that the pass
// buffer being READ is not the one being staged — drains vmcnt(0) in front of the next ds_read, which serialises the gather with
// the arithmetic it is meant to hide behind (198 vs 148 us for the register-staged kernel at B = 16 384).  The pipeline below
// orders every read behind its own counted s_waitcnt instead.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)p);
}     // per lane: its node's degree, packed slot offset; wave-wide edge count

// QQ: compile-time chunks per head (Cp / 4) — the staging loop divides by the row length; 0 = run-time.  ONEHOT: every edge_attr
// row is one-hot (bond types, src_1gp/dataset.py:82): e_ij is then exactly one W_edge row (sum_k ea_k W_k with ea in {0, 1}
// adds zeros: bit-identical) and is read from LDS instead of being contracted.
template <int H, int DE, int QQ, bool ONEHOT>
__global__ void __launch_bounds__(kBlock, 2) k_triplet_fwd_dma(FwdDmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane >> 4, q = lane & 15;
    const int Cp = QQ ? 4 * QQ : a.Cp, Q = QQ ? QQ : (Cp >> 2), HC = H * Cp, HQ = H * Q;
    const int RC = HQ + 1 + DE / 4;                       // 16-byte chunks per staged edge row
    const int WSZ = DE * HC;
    float* s_w = smem;
    const int buf_floats = kEC * RC * 4;
    float* wbase = smem + WSZ + wave * (2 * buf_floats + 2 * 16 + kEC * 2);
    float* s_buf0 = wbase;
    float* s_node0 = wbase + 2 * buf_floats;              // [2][4 nodes][4 floats] a_i rows
    int* s_idx = reinterpret_cast<int*>(s_node0 + 2 * 16);   // [kEC][2] (src, eid) of the pass being staged

    for (int i = tid; i < WSZ / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
    __syncthreads();                                      // the only block-wide barrier: waves are independent from here on
    float Mr[DE][H];
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) Mr[k][h] = a.M[k * 4 + h];

    const int npass = (a.N + 3) >> 2;
    const int gw = blockIdx.x * (kBlock / 64) + wave, GW = gridDim.x * (kBlock / 64);
    const bool qok = q < Q;
    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;

    // ---- pipeline stages ---------------------------------------------------------------------------------------------
    // lane q < 4 of a group holds slot q of its node's record: (source node, original edge id), -1 = empty
    auto load_rec = [&](int pass, int& rs, int& re) {
        const int n = 4 * pass + j;
        rs = -1; re = -1;
        if (q < 4 && pass < npass && n < a.N) { rs = a.ell_src[4 * n + q]; re = a.ell_eid[4 * n + q]; }
    };
    // Stage the rows of the nodes selected by `mask` (bit per group) of `pass` into buffer `sel`.  Returns the packing.
    auto issue_dma = [&](int pass, int rs, int re, int sel, int mask) -> PassMeta {
        const unsigned long long bal = __ballot(rs >= 0);          // bits 16 g .. 16 g + 3: occupied slots of group g
        const int d0 = (mask & 1) ? __popc((unsigned)(bal & 0xF)) : 0, d1 = (mask & 2) ? __popc((unsigned)((bal >> 16) & 0xF)) : 0,
                  d2 = (mask & 4) ? __popc((unsigned)((bal >> 32) & 0xF)) : 0, d3 = (mask & 8) ? __popc((unsigned)((bal >> 48) & 0xF)) : 0;
        PassMeta pm;
        pm.deg = j == 0 ? d0 : j == 1 ? d1 : j == 2 ? d2 : d3;
        pm.off = j == 0 ? 0 : j == 1 ? d0 : j == 2 ? d0 + d1 : d0 + d1 + d2;
        pm.tot = d0 + d1 + d2 + d3;
        if (pm.tot > kEC) return pm;                      // overflow: the caller stages it node by node
        if (q < pm.deg) {                                 // occupied slots are the first `deg` ones: publish the group's edges
            s_idx[2 * (pm.off + q)] = rs;
            s_idx[2 * (pm.off + q) + 1] = re;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned dst = lds_addr(s_buf0 + sel * buf_floats);
        const int nchunk = pm.tot * RC;
        constexpr int NI = QQ ? (kEC * (H * QQ + 1 + DE / 4) + 63) / 64 : (kEC * (H * 16 + 1 + DE / 4) + 63) / 64;
        int sv[NI], ev[NI], cc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {                    // every table read of the pass in flight before the first address is formed
            const int g = i * 64 + lane, e = min(g / RC, kEC - 1);
            cc[i] = g - (g / RC) * RC;
            sv[i] = s_idx[2 * e];
            ev[i] = s_idx[2 * e + 1];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i * 64 < nchunk) {                        // wave-uniform
                const int g = i * 64 + lane, c = cc[i];
                if (g < nchunk) {
                    const char* src;
                    if (c < HQ) src = reinterpret_cast<const char*>(a.xw) + (size_t)((unsigned)sv[i] * row_bytes + (unsigned)c * 16u);
                    else if (c == HQ) src = reinterpret_cast<const char*>(a.a_ij) + (size_t)((unsigned)sv[i] * 32u + 16u);
                    else src = reinterpret_cast<const char*>(a.edge_attr) + (size_t)((unsigned)ev[i] * (unsigned)(DE * 4) + (unsigned)(c - HQ - 1) * 16u);
                    dma16(src, dst + (unsigned)i * 1024u);
                }
            }
        }
        if (lane < 4) {                                   // a_i of the pass's four nodes (clamped: unused rows are never read)
            const int n = min(4 * pass + lane, a.N - 1);
            dma16(a.a_ij + (size_t)n * 8, lds_addr(s_node0 + sel * 16));
        }
        return pm;
    };

    // results of a pass, held in registers until the next iteration's stores
    float4 r_acc[H];
    float4 r_m = f4zero(), r_s = f4zero();
    int r_n = -1;

    auto compute = [&](int pass, const PassMeta& pm, int sel) {
        const int n = 4 * pass + j;
        if (n >= a.N || pass >= npass) { r_n = -1; return; }
        r_n = n;
        const float* buf = s_buf0 + sel * buf_floats;
        const float4 aiv = ld4(s_node0 + sel * 16 + j * 4);
        float ai[H], m[H], ssum[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { ai[h] = f4get(aiv, h); m[h] = -INFINITY; ssum[h] = 0.f; r_acc[h] = f4zero(); }
        constexpr int CH = 4;
        bool val[CH];
        float ea[CH][DE], lk[CH][H];
        int rowo[CH];                                                  // float offsets into the pass buffer
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            val[k] = k < pm.deg;
            rowo[k] = (pm.off + (val[k] ? k : 0)) * RC * 4;            // a clamped slot is a valid address inside the buffer
        }
        if (pm.deg > 0) {
            float4 aj[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                aj[k] = ld4(buf + rowo[k] + HQ * 4);
#pragma unroll
                for (int u = 0; u < DE / 4; ++u) {
                    const float4 v = ld4(buf + rowo[k] + (HQ + 1 + u) * 4);
                    ea[k][4 * u] = v.x; ea[k][4 * u + 1] = v.y; ea[k][4 * u + 2] = v.z; ea[k][4 * u + 3] = v.w;
                }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                float pre[H];
                edge_pre<H, DE>(ai, aj[k], ea[k], Mr, pre);
#pragma unroll
                for (int h = 0; h < H; ++h) lk[k][h] = leaky(pre[h], a.slope);
            }
#pragma unroll
            for (int k = 0; k < CH; ++k)
#pragma unroll
                for (int h = 0; h < H; ++h)
                    if (val[k]) m[h] = fmaxf(m[h], lk[k][h]);
            int wrow[CH];                                              // ONEHOT: float offset of the edge's W_edge row (head 0)
            if constexpr (ONEHOT) {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    int t = 0;
#pragma unroll
                    for (int kk = 1; kk < DE; ++kk) t = ea[k][kk] != 0.f ? kk : t;
                    wrow[k] = t * HC + (qok ? q : 0) * 4;
                }
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float4 wv[DE];
                float4 xr[CH], er[CH];
                if constexpr (ONEHOT) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) er[k] = ld4(s_w + wrow[k] + h * Cp);
                } else {
#pragma unroll
                    for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_w + (kk * H + h) * Cp + (qok ? q : 0) * 4);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) xr[k] = ld4(buf + rowo[k] + (h * Q + (qok ? q : 0)) * 4);
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (!val[k]) continue;
                    const float p = softmax_exp(lk[k][h] - m[h]);
                    ssum[h] += p;
                    float4 e4;
                    if constexpr (ONEHOT) {
                        e4 = er[k];
                    } else {
                        e4 = f4zero();
#pragma unroll
                        for (int kk = 0; kk < DE; ++kk) fma4(e4, ea[k][kk], wv[kk]);
                    }
                    const float4 xj = e4 * xr[k];
                    fma4(r_acc[h], p, xj);
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
        if (q == 0) {
            st4o(a.stats, (unsigned)r_n * 32u, r_m);
            st4o(a.stats, (unsigned)r_n * 32u + 16u, r_s);
        }
        r_n = -1;
    };

    // ---- prologue: record 0, its rows, record 1 ----
    int rs_nxt, re_nxt;                                   // record of the pass whose rows are staged NEXT
    int pass = gw;
    load_rec(pass, rs_nxt, re_nxt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PassMeta pm_cur = issue_dma(pass, rs_nxt, re_nxt, 0, 0xF);
    int rs_cur = rs_nxt, re_cur = re_nxt;                 // kept for the (rare) node-by-node path of the current pass
    load_rec(pass + GW, rs_nxt, re_nxt);
    int sel = 0;
    for (; pass < npass; pass += GW, sel ^= 1) {
        // rows of `pass` have landed; the record of pass + GW is in registers.  The record is an in/out operand so that the compiler
        // retires ITS count of the record loads here: otherwise it does so at their first use — after the stores below, with a
        // vmcnt(0) that would also wait for those stores.
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        store_results();                                  // of the previous pass (registers) -> in flight during this pass
        const int rs_n1 = rs_nxt, re_n1 = re_nxt;
        PassMeta pm_nxt = issue_dma(pass + GW, rs_n1, re_n1, sel ^ 1, 0xF);
        load_rec(pass + 2 * GW, rs_nxt, re_nxt);
        if (pm_cur.tot <= kEC) {
            compute(pass, pm_cur, sel);
        } else {
            // more than kEC edges in the four segments together: stage and compute node by node into this pass's (unused) buffer
            float4 t_acc[H], t_m = f4zero(), t_s = f4zero();
            int t_n = -1;
            for (int jj = 0; jj < 4; ++jj) {
                const PassMeta pj = issue_dma(pass, rs_cur, re_cur, sel, 1 << jj);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                compute(pass, pj, sel);
                if (j == jj) {
#pragma unroll
                    for (int h = 0; h < H; ++h) t_acc[h] = r_acc[h];
                    t_m = r_m; t_s = r_s; t_n = r_n;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is re-staged by the next node
            }
#pragma unroll
            for (int h = 0; h < H; ++h) r_acc[h] = t_acc[h];
            r_m = t_m; r_s = t_s; r_n = t_n;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this pass's LDS reads are done before its buffer is re-staged
        pm_cur = pm_nxt;
        rs_cur = rs_n1; re_cur = re_n1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_results();
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

static size_t dma_lds_bytes(int H, int Cp, int De) {
    const int RC = H * (Cp >> 2) + 1 + De / 4;
    return ((size_t)De * H * Cp + (size_t)(kBlock / 64) * (2 * kEC * RC * 4 + 2 * 16 + kEC * 2)) * sizeof(float);
}

template <int H, int DE, int QQ, bool ONEHOT>
static void launch_dma(const FwdDmaArgs& a, int grid, size_t lds, hipStream_t s) {
    static bool big = false;
    if (!big) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_triplet_fwd_dma<H, DE, QQ, ONEHOT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        big = true;
    }
    GLAM_PROF_LABEL("k_triplet_fwd_dma");
    hipLaunchKernelGGL((k_triplet_fwd_dma<H, DE, QQ, ONEHOT>), dim3(grid), dim3(kBlock), lds, s, a);
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
    return H >= 1 && H <= 4 && Cp >= 4 && Cp <= 64 && (Cp & 3) == 0 && (De == 4 || De == 8) && dma_lds_bytes(H, Cp, De) <= 80 * 1024;
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
                 (int)N, Cp, slope, aggr, stats};
    const size_t lds = dma_lds_bytes(H, Cp, De);
    const int npass = (int)((N + 3) / 4);
    int grid = grid_blocks > 0 ? grid_blocks : 512;                 // two 4-wave blocks per CU, every wave pipelines over its passes
    if (grid > (npass + 3) / 4) grid = (npass + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
#define GLAM_DMA_CASE(HH, DD, QQ_)                                                                                  \
    if (H == HH && De == DD && (QQ_ == 0 || Cp == 4 * QQ_)) {                                                       \
        if (edge_onehot) launch_dma<HH, DD, QQ_, true>(a, grid, lds, s);                                            \
        else launch_dma<HH, DD, QQ_, false>(a, grid, lds, s);                                                       \
        GLAM_LAUNCH_CHECK("glam_triplet_fwd_ell");                                                                  \
        return GLAM_OK;                                                                                             \
    }
    GLAM_DMA_CASE(3, 4, 15)                          // the reference's default width (hid_dim 60, 3 heads, bond one-hots)
    GLAM_DMA_CASE(1, 4, 0) GLAM_DMA_CASE(2, 4, 0) GLAM_DMA_CASE(3, 4, 0) GLAM_DMA_CASE(4, 4, 0)
    GLAM_DMA_CASE(1, 8, 0) GLAM_DMA_CASE(2, 8, 0) GLAM_DMA_CASE(3, 8, 0) GLAM_DMA_CASE(4, 8, 0)
#undef GLAM_DMA_CASE
    return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd_ell: no kernel for H=%d De=%d", H, De);
}
