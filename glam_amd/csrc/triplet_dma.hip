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

constexpr int kEC = 11;                  // edge rows per pass buffer (11 x 45 chunks fit the 8 x 64-lane staging pieces of the default width)

// One LDS-DMA piece: 64 lanes x 16 bytes, lane l lands at lds_base + 16 l (lds_base wave-uniform, in M0); source = 64-bit
// wave-uniform base (SGPR pair) + per-lane unsigned 32-bit byte offset.  Written as inline assembly on purpose: through the
// builtin the compiler knows the instruction writes LDS and — unable to prove that the pass buffer being READ is not the one
// being staged — drains vmcnt(0) in front of the next ds_read, which serialises the gather with the arithmetic it is meant to
// hide behind (198 vs 148 us for the register-staged kernel at B = 16 384).  The pipeline below orders every read behind its
// own counted s_waitcnt instead.
// QQ: compile-time chunks per head (Cp / 4) — the staging loop divides by the row length; 0 = run-time.  ONEHOT: every edge_attr
// row is one-hot (bond types, src_1gp/dataset.py:82): e_ij is then exactly one W_edge row (sum_k ea_k W_k with ea in {0, 1}
// adds zeros: bit-identical) and is read from LDS instead of being contracted.
template <int H, int DE, int QQ, bool ONEHOT>
__global__ void __launch_bounds__(kBlock, 2) k_triplet_fwd_dma(FwdDmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane >> 4, q = lane & 15;
    const int Cp = QQ ? 4 * QQ : a.Cp, Q = QQ ? QQ : (Cp >> 2), HC = H * Cp, HQ = H * Q;
    const int WSZ = DE * HC;
    constexpr int NI = QQ ? (kEC * H * QQ + 63) / 64 : (kEC * H * 16 + 63) / 64;   // staging pieces of the row region
    // per-wave LDS: 2 x [rows: NI x 64 chunks | meta: a_j x 16, edge_attr x 16 x DE/4, a_i x 4 (one 64-lane piece)] | src table
    constexpr int kRowF = NI * 64 * 4, kMetaF = 64 * 4, kBufF = kRowF + kMetaF;
    float* s_w = smem;
    float* wbase = smem + WSZ + wave * (2 * kBufF + kMetaSlots);
    int* s_src = reinterpret_cast<int*>(wbase + 2 * kBufF);             // [16] source node of every packed slot of the pass being staged

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
    // pass-independent part of the staging addresses: piece i, this lane -> (packed slot, byte offset inside the xw row)
    int st_slot[NI];
    unsigned st_coff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = i * 64 + lane;
        st_slot[i] = g / HQ;
        st_coff[i] = (unsigned)(g - st_slot[i] * HQ) * 16u;
    }
    // the side-table piece: lanes 0..15 a_j of slot l, 16..16+16*DE/4-1 edge_attr chunks, then 4 lanes a_i of the pass's nodes
    constexpr int kEaLanes = kMetaSlots * (DE / 4);
    const int mt_kind = lane < kMetaSlots ? 0 : lane < kMetaSlots + kEaLanes ? 1 : lane < kMetaSlots + kEaLanes + 4 ? 2 : 3;
    const int mt_slot = mt_kind == 0 ? lane : mt_kind == 1 ? (lane - kMetaSlots) / (DE / 4) : 0;
    const unsigned mt_sub = mt_kind == 1 ? (unsigned)((lane - kMetaSlots) % (DE / 4)) * 16u : 0u;

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
        pm.tot = __builtin_amdgcn_readfirstlane(d0 + d1 + d2 + d3);
        pm.dmax = __builtin_amdgcn_readfirstlane(max(max(d0, d1), max(d2, d3)));   // provably scalar: the slot loops branch on it
        if (pm.tot > kEC || pm.tot == 0) return pm;       // overflow: the caller stages it node by node; nothing to stage
        // the side-table piece needs (src, eid) of slot `mt_slot`: fetch them from the owning lanes while the table is written
        if (q < pm.deg) s_src[pm.off + q] = rs;           // occupied slots are the first `deg` ones of a group
        // owner lane of packed slot t: group g with off_g <= t < off_g + deg_g, lane 16 g + (t - off_g)
        const int t = min(mt_slot, pm.tot - 1);
        const int og = t < d0 ? 0 : t < d0 + d1 ? 1 : t < d0 + d1 + d2 ? 2 : 3;
        const int ooff = og == 0 ? 0 : og == 1 ? d0 : og == 2 ? d0 + d1 : d0 + d1 + d2;
        const int owner = 16 * og + (t - ooff);
        const int m_src = __shfl(rs, owner, 64), m_eid = __shfl(re, owner, 64);
        const unsigned dst = lds_addr(wbase + sel * kBufF);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int last = pm.tot - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i * 64 < pm.tot * HQ) {                   // wave-uniform; lanes past the last chunk re-fetch the last row (never read)
                const int sv = s_src[min(st_slot[i], last)];
                dma16(a.xw, (unsigned)sv * row_bytes + st_coff[i], dst + (unsigned)i * 1024u);
            }
        }
        {   // a_j | edge_attr | a_i in one piece (lanes beyond the table fetch a_j of slot 0 into padding)
            const int n_i = min(4 * pass + (lane - kMetaSlots - kEaLanes), a.N - 1);
            const unsigned off = mt_kind == 1 ? (unsigned)m_eid * (unsigned)(DE * 4) + mt_sub
                               : mt_kind == 2 ? (unsigned)max(n_i, 0) * 32u : (unsigned)m_src * 32u + 16u;
            // two wave-uniform bases cannot share one piece: issue the edge_attr lanes and the a_ij lanes as two masked pieces
            if (mt_kind == 1) dma16(a.edge_attr, off, dst + (unsigned)kRowF * 4u);
            else dma16(a.a_ij, off, dst + (unsigned)kRowF * 4u);
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
        const float* buf = wbase + sel * kBufF;
        const float* meta = buf + kRowF;
        float m[H], ssum[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { m[h] = -INFINITY; ssum[h] = 0.f; r_acc[h] = f4zero(); }
        if (pm.deg > 0) {
            const float4 aiv = ld4(meta + (kMetaSlots + kEaLanes + j) * 4);
            float ai[H];
#pragma unroll
            for (int h = 0; h < H; ++h) ai[h] = f4get(aiv, h);
            constexpr int CH = 4;
            bool val[CH];
            int slot[CH];
            float ea[CH][DE], lk[CH][H];
            float4 aj[CH];
            // slots k >= dmax are empty in all four nodes of the pass: skipped by a scalar branch (56 % of the molecular passes have
            // no node of degree 3); a slot that is empty in THIS node only runs branch-free with weight 0
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                val[k] = k < pm.deg;
                slot[k] = pm.off + (val[k] ? k : 0);      // an unused slot aliases the node's first edge: finite data, weight 0
                if (k < pm.dmax) {
                    aj[k] = ld4(meta + slot[k] * 4);
#pragma unroll
                    for (int u = 0; u < DE / 4; ++u) {
                        const float4 v = ld4(meta + (kMetaSlots + slot[k] * (DE / 4) + u) * 4);
                        ea[k][4 * u] = v.x; ea[k][4 * u + 1] = v.y; ea[k][4 * u + 2] = v.z; ea[k][4 * u + 3] = v.w;
                    }
                }
            }
            int wrow[CH];                                              // ONEHOT: float offset of the edge's W_edge row (head 0)
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
                float4 wv[DE];
                float4 xr[CH], er[CH];
                if constexpr (!ONEHOT) {
#pragma unroll
                    for (int kk = 0; kk < DE; ++kk) wv[kk] = ld4(s_w + (kk * H + h) * Cp + (qok ? q : 0) * 4);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (k < pm.dmax) {
                        if constexpr (ONEHOT) er[k] = ld4(s_w + wrow[k] + h * Cp);
                        xr[k] = ld4(buf + (slot[k] * HQ + h * Q + (qok ? q : 0)) * 4);
                    }
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (k < pm.dmax) {
                        // branch-free inside the wave: an unused slot contributes p = 0 times the (finite) row of the first edge
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
                        const float4 xj = e4 * xr[k];
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
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
    PassMeta pm_cur = issue_dma(pass, rs_nxt, re_nxt, 0, 0xF);
    int rs_cur = rs_nxt, re_cur = re_nxt;                 // kept for the (rare) node-by-node path of the current pass
    load_rec(pass + GW, rs_nxt, re_nxt);
    int sel = 0;
#ifdef GLAM_DMA_PROF
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = clock64();
#endif
    for (; pass < npass; pass += GW, sel ^= 1) {
        DSTAMP(0);
        // rows of `pass` have landed; the record of pass + GW is in registers.  The record is an in/out operand so that the compiler
        // retires ITS count of the record loads here: otherwise it does so at their first use — after the stores below, with a
        // vmcnt(0) that would also wait for those stores.
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        DSTAMP(1);
        store_results();                                  // of the previous pass (registers) -> in flight during this pass
        DSTAMP(2);
        const int rs_n1 = rs_nxt, re_n1 = re_nxt;
        PassMeta pm_nxt = issue_dma(pass + GW, rs_n1, re_n1, sel ^ 1, 0xF);
        DSTAMP(3);
        load_rec(pass + 2 * GW, rs_nxt, re_nxt);
        DSTAMP(4);
        if (pm_cur.tot <= kEC) {
            compute(pass, pm_cur, sel);
            DSTAMP(5);
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
#ifdef GLAM_DMA_PROF
    if (lane == 0 && gw < 64) for (int k = 0; k < 8; ++k) g_dma_prof[gw * 8 + k] = pacc[k];
#endif
}

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
// FUSE: the update GEMM out[16 nodes, Cp] = aggr_tile @ W_scale + bias as an MFMA epilogue per 16-node tile (the four waves of a block
// walk the same tile: pass = 4 * tile + wave), weight image resident in LDS — the step's forward kernel with the pipeline inside: the
// next tile's index record, rows and side table are in flight during the current tile's arithmetic AND its epilogue.
template <int H, int DE, bool ONEHOT, bool FUSE>
__global__ void __launch_bounds__(kBlock, GLAM_PIPE_WAVES) k_triplet_fwd_pipe(FwdDmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane >> 4, q = lane & 15;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    const int WSZ = DE * HC;
    constexpr int kMetaF = 64 * 4;
    float* s_w = smem;
    float* wbase = smem + WSZ + wave * (2 * kMetaF);
    // FUSE: 16-node aggr tile, output staging tile and the update GEMM's weight image behind the per-wave side tables
    const int LDT = HC + 4;
    float* s_tile = smem + WSZ + (kBlock / 64) * 2 * kMetaF;
    float* s_out = s_tile + 16 * LDT;
    float* s_img = s_out + 16 * 64;
    if constexpr (FUSE) lds_copy_async<kBlock>(a.img_upd, s_img, ((HC + 15) >> 4) * 256, tid);   // drained by the barrier below
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
    // FUSE: the tile's row of this node goes to LDS for the epilogue (rows past N / without edges are zero: out = bias)
    auto publish_tile = [&]() {
        if (qok) {
#pragma unroll
            for (int h = 0; h < H; ++h) st4(s_tile + (wave * 4 + j) * LDT + h * Cp + q * 4, r_n >= 0 ? r_acc[h] : f4zero());
        }
    };
    float4 o_hold = f4zero();
    int o_row = -1;
    typedef float v4f __attribute__((ext_vector_type(4)));
    auto epilogue = [&](int tile) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // raw barrier: the prefetch of the next tile stays in flight across it
        const int c = lane & 15, kq = lane >> 4;
        const int GK = (HC + 15) >> 4;
        v4f cacc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int g0 = 0; g0 < GK; g0 += 4) {
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
                for (int jj = 0; jj < 4; ++jj)
                    cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(af[u], jj), f4get(bf[u], jj), cacc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s_out[(kq * 4 + i) * 64 + 4 * c + wave] = cacc[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int row = tid >> 4, c4 = (tid & 15) * 4;
        o_row = -1;
        if (c4 < Cp && 16 * tile + row < a.N) {
            float4 v = ld4(s_out + row * 64 + c4);
            const float4 bb = ld4(a.bias_p + c4);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            o_hold = v;
            o_row = 16 * tile + row;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto store_results = [&]() {
        if constexpr (FUSE) {
            if (o_row >= 0) st4(a.out + (size_t)o_row * Cp + (tid & 15) * 4, o_hold);
            o_row = -1;
        }
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
        if constexpr (FUSE) { publish_tile(); epilogue(pass >> 2); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DSTAMP(5);

        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_b);
        store_results();
        pm_cur = prefetch(pass + 2 * GW, rs_nxt, re_nxt, 0, rows_a);
        load_rec(pass + 3 * GW, rs_nxt, re_nxt);
        if (pass + GW - wave < pass_end) {                // block-uniform: both barriers of the second tile or none
            compute(pass + GW, pm_nxt, 1, rows_b);
            if constexpr (FUSE) { publish_tile(); epilogue((pass + GW) >> 2); }
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
    hipLaunchKernelGGL((k_triplet_fwd_pipe<H, DE, ONEHOT, false>), dim3(grid), dim3(kBlock), lds, s, a);
}

template <int H, int DE, bool ONEHOT>
static void launch_pipe_fused(const FwdDmaArgs& a, int grid, hipStream_t s) {
    const int HC = H * a.Cp;
    const size_t lds = ((size_t)DE * HC + (size_t)(kBlock / 64) * 2 * 64 * 4 + 16 * (size_t)(HC + 4) + 16 * 64 + (size_t)((HC + 15) & ~15) * 64) * sizeof(float);
    static bool big = false;
    if (!big) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_triplet_fwd_pipe<H, DE, ONEHOT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        big = true;
    }
    GLAM_PROF_LABEL("k_triplet_fwd_pipe+update");
    hipLaunchKernelGGL((k_triplet_fwd_pipe<H, DE, ONEHOT, true>), dim3(grid), dim3(kBlock), lds, s, a);
}

// forward aggregate + update GEMM for molecular graphs (called by layer.hip when the caller supplied ELL records)
int triplet_fwd_pipe_fused(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                           const int32_t* ell_src, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De, float slope,
                           int edge_onehot, float* aggr, float* stats, const float* img_upd, const float* bias_p, float* out, hipStream_t s) {
    if (N == 0) return GLAM_OK;
    if (!(H >= 1 && H <= 4 && (De == 4 || De == 8) && (Cp >> 2) > 8 && (Cp >> 2) <= 16 && H * Cp <= 192))
        return fail(GLAM_E_UNSUPPORTED, "triplet_fwd_pipe_fused: H=%d Cp=%d De=%d outside the fused table (36 <= Cp <= 64, H*Cp <= 192)", H, Cp, De);
    if ((uint64_t)N * H * Cp * 4 >= (1ull << 32) || (uint64_t)E * De * 4 >= (1ull << 32))
        return fail(GLAM_E_UNSUPPORTED, "triplet_fwd_pipe_fused: a tensor exceeds 4 GiB (32-bit offsets)");
    FwdDmaArgs a{xw, a_ij, edge_attr, w_edge, M, ell_src, ell_eid, (int)N, Cp, slope, aggr, stats, img_upd, bias_p, out};
    const int ntiles = (int)((N + 15) / 16);
    const int grid = ntiles < 512 ? ntiles : 512;         // two blocks per CU (LDS-resident weight image)
#define GLAM_PF_CASE(HH, DD)                                                     \
    if (H == HH && De == DD) {                                                   \
        if (edge_onehot) launch_pipe_fused<HH, DD, true>(a, grid, s);            \
        else launch_pipe_fused<HH, DD, false>(a, grid, s);                       \
        GLAM_LAUNCH_CHECK("triplet_fwd_pipe_fused");                             \
        return GLAM_OK;                                                          \
    }
    GLAM_PF_CASE(1, 4) GLAM_PF_CASE(2, 4) GLAM_PF_CASE(3, 4) GLAM_PF_CASE(4, 4)
    GLAM_PF_CASE(1, 8) GLAM_PF_CASE(2, 8) GLAM_PF_CASE(3, 8) GLAM_PF_CASE(4, 8)
#undef GLAM_PF_CASE
    return fail(GLAM_E_UNSUPPORTED, "triplet_fwd_pipe_fused: no kernel for H=%d De=%d", H, De);
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_triplet_bwd_src_pipe: backward B2 (d_xw[j] = sum over the out-edges of j of alpha_e * e_ij * d_aggr[dst], d_a_j[j] = sum dpre_e;
// the general kernel is k_triplet_bwd_src) with the data flow of k_triplet_fwd_pipe: ELL records BY SOURCE (dst[4] | eid[4] per
// node, glam_ell_build on the CSR transpose), the d_aggr rows of the next pass prefetched into registers, the per-edge scalars
// (alpha_e | dpre_e | edge_attr: three masked LDS-DMA pieces into one 1 KB side table) staged one pass ahead, results stored one
// pass late.  No softmax, so a pass is cheaper than the forward's.  Same arithmetic in the same order: bit-identical.
// ------------------------------------------------------------------------------------------------------------------------------
struct SrcPipeArgs {
    const float* d_aggr; const float* alpha_e; const float* dpre_e; const float* edge_attr; const float* w_edge;
    const int* ell_dst; const int* ell_eid;      // [N][4] each, by source
    int N; int Cp;
    float* d_xw; float* d_a_ij;
};

template <int H, int DE, bool ONEHOT>
__global__ void __launch_bounds__(kBlock, GLAM_PIPE_WAVES) k_triplet_bwd_src_pipe(SrcPipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane >> 4, q = lane & 15;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    const int WSZ = DE * HC;
    constexpr int kMetaF = 64 * 4;
    float* s_w = smem;
    float* wbase = smem + WSZ + wave * (2 * kMetaF);
    for (int i = tid; i < WSZ / 4; i += kBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
    __syncthreads();

    const int npass = (a.N + 3) >> 2;
    const int gw = blockIdx.x * (kBlock / 64) + wave, GW = gridDim.x * (kBlock / 64);
    const bool qok = q < Q;
    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;
    const unsigned qoff = (unsigned)(qok ? q : 0) * 16u;
    // side table: lanes [0, 16) alpha_e of packed slot t, [16, 32) dpre_e, [32, 32 + 16 * DE / 4) edge_attr
    constexpr int kEaLanes = kMetaSlots * (DE / 4);
    const int mt_kind = lane < 16 ? 0 : lane < 32 ? 1 : lane < 32 + kEaLanes ? 2 : 3;
    const int mt_slot = mt_kind == 0 ? lane : mt_kind == 1 ? lane - 16 : mt_kind == 2 ? (lane - 32) / (DE / 4) : 0;
    const unsigned mt_sub = mt_kind == 2 ? (unsigned)((lane - 32) % (DE / 4)) * 16u : 0u;
    constexpr int CH = 4;

    auto load_rec = [&](int pass, int& rs, int& re) {
        const int n = 4 * pass + j;
        rs = -1; re = -1;
        if (q < 4 && pass < npass && n < a.N) { rs = a.ell_dst[4 * n + q]; re = a.ell_eid[4 * n + q]; }
    };
    auto prefetch = [&](int pass, int rs, int re, int sel, float4 (&rows)[CH][H]) -> PassMeta {
        const unsigned long long bal = __ballot(rs >= 0);
        const int d0 = __popc((unsigned)(bal & 0xF)), d1 = __popc((unsigned)((bal >> 16) & 0xF)),
                  d2 = __popc((unsigned)((bal >> 32) & 0xF)), d3 = __popc((unsigned)((bal >> 48) & 0xF));
        PassMeta pm;
        pm.deg = j == 0 ? d0 : j == 1 ? d1 : j == 2 ? d2 : d3;
        pm.off = j == 0 ? 0 : j == 1 ? d0 : j == 2 ? d0 + d1 : d0 + d1 + d2;
        pm.tot = __builtin_amdgcn_readfirstlane(d0 + d1 + d2 + d3);
        pm.dmax = __builtin_amdgcn_readfirstlane(max(max(d0, d1), max(d2, d3)));
        if (pm.tot == 0) return pm;
        const int t = min(mt_slot, pm.tot - 1);
        const int og = t < d0 ? 0 : t < d0 + d1 ? 1 : t < d0 + d1 + d2 ? 2 : 3;
        const int ooff = og == 0 ? 0 : og == 1 ? d0 : og == 2 ? d0 + d1 : d0 + d1 + d2;
        const int owner = 16 * og + (t - ooff);
        const int m_eid = __shfl(re, owner, 64);
        int sk[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) sk[k] = __shfl(rs, 16 * j + k, 64);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sk[0]), "+v"(sk[1]), "+v"(sk[2]), "+v"(sk[3]) : : "memory");
        const unsigned dst = lds_addr(wbase + sel * kMetaF);
        // three wave-uniform bases: three masked pieces into the same 1 KB buffer (lane l always lands at dst + 16 l)
        if (mt_kind == 0) dma16(a.alpha_e, (unsigned)m_eid * 16u, dst);
        else if (mt_kind == 1) dma16(a.dpre_e, (unsigned)m_eid * 16u, dst);
        else if (mt_kind == 2) dma16(a.edge_attr, (unsigned)m_eid * (unsigned)(DE * 4) + mt_sub, dst);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (k < pm.dmax) {                            // scalar branch: slot k is empty in all four nodes otherwise
                const unsigned ro = (unsigned)max(sk[k] >= 0 ? sk[k] : sk[0], 0) * row_bytes + qoff;
#pragma unroll
                for (int h = 0; h < H; ++h) rows[k][h] = ld4o(a.d_aggr, ro + (unsigned)h * head_bytes);
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) rows[k][h] = f4zero();
            }
        }
        return pm;
    };

    float4 r_acc[H];
    float4 r_da = f4zero();
    int r_n = -1;
    auto compute = [&](int pass, const PassMeta& pm, int sel, const float4 (&rows)[CH][H]) {
        const int n = 4 * pass + j;
        if (n >= a.N || pass >= npass) { r_n = -1; return; }
        r_n = n;
        const float* meta = wbase + sel * kMetaF;
#pragma unroll
        for (int h = 0; h < H; ++h) r_acc[h] = f4zero();
        r_da = f4zero();
        if (pm.deg > 0) {
            bool val[CH];
            float ea[CH][DE];
            float4 al[CH], dp[CH];
            int wrow[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                val[k] = k < pm.deg;
                if (k < pm.dmax) {
                    const int slot = pm.off + (val[k] ? k : 0);
                    al[k] = ld4(meta + slot * 4);
                    dp[k] = ld4(meta + (16 + slot) * 4);
#pragma unroll
                    for (int u = 0; u < DE / 4; ++u) {
                        const float4 v = ld4(meta + (32 + slot * (DE / 4) + u) * 4);
                        ea[k][4 * u] = v.x; ea[k][4 * u + 1] = v.y; ea[k][4 * u + 2] = v.z; ea[k][4 * u + 3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (k < pm.dmax) {
                    if (!val[k]) { al[k] = f4zero(); dp[k] = f4zero(); }      // an empty slot runs with weight 0 on a finite row
                    r_da.x += dp[k].x; r_da.y += dp[k].y; r_da.z += dp[k].z; r_da.w += dp[k].w;
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
                        float4 e4;
                        if constexpr (ONEHOT) {
                            e4 = er[k];
                        } else {
                            e4 = f4zero();
#pragma unroll
                            for (int kk = 0; kk < DE; ++kk) fma4(e4, ea[k][kk], wv[kk]);
                        }
                        const float4 dg = e4 * rows[k][h];
                        fma4(r_acc[h], f4get(al[k], h), dg);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto store_results = [&]() {
        if (r_n < 0) return;
        if (qok) {
            const unsigned orow = (unsigned)r_n * row_bytes + (unsigned)q * 16u;
#pragma unroll
            for (int h = 0; h < H; ++h) st4o(a.d_xw, orow + (unsigned)h * head_bytes, r_acc[h]);
        }
        if (q == 0) st4o(a.d_a_ij, (unsigned)r_n * 32u + 16u, r_da);
        r_n = -1;
    };
    auto settle = [&](float4 (&rows)[CH][H]) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int h = 0; h < H; ++h)
                asm volatile("" : : "v"(rows[k][h].x), "v"(rows[k][h].y), "v"(rows[k][h].z), "v"(rows[k][h].w));
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
    for (; pass < npass; pass += 2 * GW) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_a);
        store_results();
        PassMeta pm_nxt = prefetch(pass + GW, rs_nxt, re_nxt, 1, rows_b);
        load_rec(pass + 2 * GW, rs_nxt, re_nxt);
        compute(pass, pm_cur, 0, rows_a);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_b);
        store_results();
        pm_cur = prefetch(pass + 2 * GW, rs_nxt, re_nxt, 0, rows_a);
        load_rec(pass + 3 * GW, rs_nxt, re_nxt);
        compute(pass + GW, pm_nxt, 1, rows_b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_results();
}

template <int H, int DE, bool ONEHOT>
static void launch_src_pipe(const SrcPipeArgs& a, int grid, hipStream_t s) {
    const size_t lds = ((size_t)DE * H * a.Cp + (size_t)(kBlock / 64) * 2 * 64 * 4) * sizeof(float);
    GLAM_PROF_LABEL("k_triplet_bwd_src_pipe");
    hipLaunchKernelGGL((k_triplet_bwd_src_pipe<H, DE, ONEHOT>), dim3(grid), dim3(kBlock), lds, s, a);
}

// B2 over ELL records by source (called by triplet_bwd_impl when the caller supplied them)
int triplet_bwd_src_pipe(const float* d_aggr, const float* alpha_e, const float* dpre_e, const float* edge_attr, const float* w_edge,
                         const int32_t* ell_dst, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De, int edge_onehot,
                         float* d_xw, float* d_a_ij, int grid_blocks, hipStream_t s) {
    if (N == 0) return GLAM_OK;
    if (!(H >= 1 && H <= 4 && (De == 4 || De == 8) && Cp <= 64 && (Cp & 3) == 0))
        return fail(GLAM_E_UNSUPPORTED, "triplet_bwd_src_pipe: H=%d Cp=%d De=%d outside the kernel table", H, Cp, De);
    if ((uint64_t)N * H * Cp * 4 >= (1ull << 32) || (uint64_t)E * De * 4 >= (1ull << 32))
        return fail(GLAM_E_UNSUPPORTED, "triplet_bwd_src_pipe: a tensor exceeds 4 GiB (32-bit offsets)");
    SrcPipeArgs a{d_aggr, alpha_e, dpre_e, edge_attr, w_edge, ell_dst, ell_eid, (int)N, Cp, d_xw, d_a_ij};
    const int npass = (int)((N + 3) / 4);
    int grid = grid_blocks > 0 ? grid_blocks : 512;
    if (grid > (npass + 3) / 4) grid = (npass + 3) / 4;
#define GLAM_SP_CASE(HH, DD)                                                     \
    if (H == HH && De == DD) {                                                   \
        if (edge_onehot) launch_src_pipe<HH, DD, true>(a, grid, s);              \
        else launch_src_pipe<HH, DD, false>(a, grid, s);                         \
        GLAM_LAUNCH_CHECK("triplet_bwd_src_pipe");                               \
        return GLAM_OK;                                                          \
    }
    GLAM_SP_CASE(1, 4) GLAM_SP_CASE(2, 4) GLAM_SP_CASE(3, 4) GLAM_SP_CASE(4, 4)
    GLAM_SP_CASE(1, 8) GLAM_SP_CASE(2, 8) GLAM_SP_CASE(3, 8) GLAM_SP_CASE(4, 8)
#undef GLAM_SP_CASE
    return fail(GLAM_E_UNSUPPORTED, "triplet_bwd_src_pipe: no kernel for H=%d De=%d", H, De);
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
    const int Q = Cp >> 2, NI = Q == 15 && H == 3 && De == 4 ? (kEC * H * 15 + 63) / 64 : (kEC * H * 16 + 63) / 64;
    return ((size_t)De * H * Cp + (size_t)(kBlock / 64) * (2 * (NI * 64 * 4 + 64 * 4) + kMetaSlots)) * sizeof(float);
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
    const size_t lds = dma_lds_bytes(H, Cp, De);
    const int npass = (int)((N + 3) / 4);
    int grid = grid_blocks > 0 ? grid_blocks : 512;                 // two 4-wave blocks per CU, every wave pipelines over its passes
    if (grid > (npass + 3) / 4) grid = (npass + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    static const bool use_dma = [] { const char* e = getenv("GLAM_ELL_VARIANT"); return e && !strcmp(e, "dma"); }();
    if (!use_dma) {   // default: rows prefetched into registers (k_triplet_fwd_pipe)
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
    }
#define GLAM_DMA_CASE(HH, DD, QQ_)                                                                                  \
    if (H == HH && De == DD && (QQ_ == 0 || Cp == 4 * QQ_)) {                                                       \
        if (edge_onehot) launch_dma<HH, DD, QQ_, true>(a, grid, lds, s);                                            \
        else launch_dma<HH, DD, QQ_, false>(a, grid, lds, s);                                                       \
        GLAM_LAUNCH_CHECK("glam_triplet_fwd_ell");                                                                  \
        return GLAM_OK;                                                                                             \
    }
    GLAM_DMA_CASE(3, 4, 15)                          // the reference's default width (hid_dim 60, 3 heads, bond one-hots)
    GLAM_DMA_CASE(1, 4, 0) GLAM_DMA_CASE(2, 4, 0)
    GLAM_DMA_CASE(1, 8, 0) GLAM_DMA_CASE(2, 8, 0)
#undef GLAM_DMA_CASE
    return fail(GLAM_E_UNSUPPORTED, "glam_triplet_fwd_ell: no kernel for H=%d De=%d", H, De);
}

#ifdef GLAM_DMA_PROF
extern "C" int glam_debug_dma_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_dma_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
