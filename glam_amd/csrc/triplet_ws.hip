// Warp-specialised forward of the TripletMessage layer for molecular graphs (ELL index records, in-degree <= 4):
//     producer waves 0..3   the software-pipelined scatter-aggregate of csrc/triplet_dma.hip (reference: src_1gp/layer.py:42-55
//                           through PyG propagate -> message -> scatter-add): index record -> row prefetch -> logits / segment
//                           softmax / weighted sum; besides the stores to aggr / stats each wave PUBLISHES its four rows of the
//                           16-node tile into a ring of LDS tile slots
//     consumer waves 4..7   the update GEMM out = aggr @ W_scale + bias (src_1gp/layer.py:57-61) on the fp32 matrix cores: wave w
//                           owns output columns 16 w .. 16 w + 15, keeps its 180 x 16 slice of W_scale in 48 REGISTERS for the
//                           whole launch (no weight image in LDS), takes a tile's A fragments from the ring and stores its columns
// The two halves meet only at two LDS counters per ring slot (rows published / fragments taken): no block-wide barrier after the
// prologue, so the gather (vector ALU + memory pipe) and the 48-deep dependent MFMA chain of a tile run side by side on every SIMD
// instead of back to back in the same four waves (k_triplet_fwd_pipe<..., FUSE>: fused time = aggregate time + epilogue time).
// One 8-wave block per CU (the producers' register budget, two waves per SIMD, applies to every wave of a launch).
// Same arithmetic in the same order as k_triplet_fwd / k_ts_gemm: outputs are bit-identical (tested).
#include "triplet_pipe.h"

#include <stdlib.h>

namespace glam {

constexpr int kWsCons = 4;                // consumer waves: one 16-column tile of `out` each
#ifndef GLAM_WS_RING
#define GLAM_WS_RING 8
#endif
constexpr int kWsRing = GLAM_WS_RING;     // tile slots between producers and consumers

#ifdef GLAM_WS_PROF   // developer aid (tools/ws_prof.py): where the producer / consumer waves spend their cycles
__device__ long long g_ws_prof[64 * 12 * 8];
#define WSTAMP(k) do { const long long now__ = clock64(); pacc[k] += now__ - plast; plast = now__; } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif

// lane n of the caller's 16-lane row (the lanes of one node) -> every lane of the row: one v_mov_b32_dpp row_newbcast (no LDS)
template <int CTRL>
__device__ __forceinline__ int dpp_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int row_bcast_i(int v, int n) {
    switch (n & 15) {   // n is a compile-time constant after unrolling: the switch folds
        case 0: return dpp_int<0x150>(v);   case 1: return dpp_int<0x151>(v);   case 2: return dpp_int<0x152>(v);   case 3: return dpp_int<0x153>(v);
        case 4: return dpp_int<0x154>(v);   case 5: return dpp_int<0x155>(v);   case 6: return dpp_int<0x156>(v);   case 7: return dpp_int<0x157>(v);
        case 8: return dpp_int<0x158>(v);   case 9: return dpp_int<0x159>(v);   case 10: return dpp_int<0x15A>(v);  case 11: return dpp_int<0x15B>(v);
        case 12: return dpp_int<0x15C>(v);  case 13: return dpp_int<0x15D>(v);  case 14: return dpp_int<0x15E>(v);  default: return dpp_int<0x15F>(v);
    }
}
__device__ __forceinline__ float row_bcast(float v, int n) { return __builtin_bit_cast(float, row_bcast_i(__builtin_bit_cast(int, v), n)); }

__device__ __forceinline__ int flag_load(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void flag_bump(int* p) { (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// P producer waves (4 or 8: one or two groups of four; group g gathers the block's tiles g, g + P/4, ...) + 4 consumer waves
template <int H, int DE, bool ONEHOT, int P>
__global__ void __launch_bounds__((P + kWsCons) * 64, (P + kWsCons) / 4) k_triplet_fwd_ws(FwdDmaArgs a) {
    constexpr int kWsBlock = (P + kWsCons) * 64, kWsProd = P, PG = P / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    const int WSZ = DE * HC, LDT = HC + 4;
    constexpr int kMetaF = 64 * 4;
    float* s_w = smem;
    int* s_ready = reinterpret_cast<int*>(smem + WSZ);        // [kWsRing] producer check-ins per slot (monotonic)
    int* s_taken = s_ready + 32;                              // [kWsRing] consumer check-outs per slot
    float* s_meta = smem + WSZ + 64;                          // per producer wave: 2 side tables of 1 KB
    float* s_ring = s_meta + kWsProd * 2 * kMetaF;            // kWsRing tiles of 16 x LDT floats
    for (int i = tid; i < WSZ / 4; i += kWsBlock) st4(s_w + 4 * i, ld4(a.w_edge + 4 * i));
    float* s_mt = reinterpret_cast<float*>(s_ready + 16);     // M transposed: [head][edge feature] (one ds_read_b128 per lane and pass)
    if (tid < 64) {
        if ((tid >> 4) == 1) s_mt[tid & 15] = a.M[(tid & 3) * 4 + ((tid >> 2) & 3)];
        else s_ready[tid] = 0;
    }
    __syncthreads();                                          // the only block-wide barrier
    const int ntiles = (a.N + 15) >> 4;
#ifdef GLAM_WS_PROF
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = clock64();
#endif

    if (wave >= kWsProd) {
        // ------------------------------------------------------------------------------------------------------------------
        // consumer: out[16 tile .. +15, 16 w .. +15] = aggr_tile[16, HC] @ W_scale[:, 16 w .. +15] + bias
        // ------------------------------------------------------------------------------------------------------------------
        typedef float v4f __attribute__((ext_vector_type(4)));
        const int w = wave - kWsProd, c = lane & 15, kq = lane >> 4;
        const int GK = (HC + 15) >> 4;                        // 16-k groups, <= 12
        const int col = 16 * w + c;
        const int pos = (col & 3) * 16 + (col >> 2);          // position of logical column `col` in a k_ts_gemm image row
        float4 bf[12];
#pragma unroll
        for (int g = 0; g < 12; ++g) bf[g] = g < GK ? ld4(a.img_upd + ((size_t)(4 * g + kq) * 64 + pos) * 4) : f4zero();
        const float bias = col < Cp ? a.bias_p[col] : 0.f;
        int it = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
            const int slot = it % kWsRing, want = 4 * (it / kWsRing + 1);
            while (flag_load(s_ready + slot) < want) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            WSTAMP(0);
            const float* tl = s_ring + slot * 16 * LDT + c * LDT + 4 * kq;
            float4 af[12];
#pragma unroll
            for (int g = 0; g < 12; ++g) af[g] = (g < GK && 16 * g + 4 * kq < HC) ? ld4(tl + 16 * g) : f4zero();
            WSTAMP(1);
            // no wait here: the compiler counts the fragment reads down (lgkmcnt(11), (10), ...) in front of the MFMAs that use them,
            // so the chain starts when the first fragment lands instead of after the twelfth
            v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                if (g < GK) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(af[g], jj), f4get(bf[g], jj), acc, 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_taken + slot);         // every fragment is in registers: the slot may be refilled
            const int r0 = 16 * tile + 4 * kq;
            if (col < Cp) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r0 + i < a.N) a.out[(size_t)(r0 + i) * Cp + col] = acc[i] + bias;
            }
            WSTAMP(2);
        }
#ifdef GLAM_WS_PROF
        if (lane == 0 && blockIdx.x < 64) for (int k = 0; k < 8; ++k) g_ws_prof[(blockIdx.x * 12 + wave) * 8 + k] = pacc[k];
#endif
        return;
    }

    // ----------------------------------------------------------------------------------------------------------------------
    // producer: k_triplet_fwd_pipe's pipeline (triplet_dma.hip), publishing into the ring instead of meeting at barriers
    // ----------------------------------------------------------------------------------------------------------------------
    float Mr[DE][H];
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h) Mr[k][h] = a.M[k * 4 + h];
    float* wbase = s_meta + wave * (2 * kMetaF);
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)smem;     // dynamic LDS base: a constant
    const int npass = (a.N + 3) >> 2;
    const int grp = wave >> 2, rw = wave & 3;                 // tile group of this wave, its four rows of the group's tiles
    const int gw = 4 * (blockIdx.x + grp * gridDim.x) + rw, GW = 4 * PG * gridDim.x;      // first pass, pass stride
    const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;
    constexpr int kEaLanes = kMetaSlots * (DE / 4);
    constexpr int CH = 4;
    // Lane-derived constants (node group j, chunk q, side-table role ...) are recomputed from an OPAQUE copy of the lane id in every
    // half-trip instead of living in a dozen registers across the loop: at three waves per SIMD the allocator spilled exactly those to
    // scratch, and every reload sat behind an s_waitcnt vmcnt(0) that also waited for the stores and the prefetch in flight.
    int lv = lane;
#define LANE_CONSTS()                                                                        \
    asm volatile("" : "+v"(lv));                                                             \
    const int j = lv >> 4, q = lv & 15;                                                      \
    const bool qok = q < Q;                                                                  \
    const unsigned qoff = (unsigned)(qok ? q : 0) * 16u

    auto load_rec = [&](int pass, int& rs, int& re) {
        LANE_CONSTS(); (void)qok; (void)qoff;
        const int n = 4 * pass + j;
        rs = -1; re = -1;
        if (q < 4 && pass < npass && n < a.N) { rs = a.ell_src[4 * n + q]; re = a.ell_eid[4 * n + q]; }
    };
    auto prefetch = [&](int pass, int rs, int re, int sel, float4 (&rows)[CH][H]) -> PassMeta {
        LANE_CONSTS();
        const int mt_kind = lv < kMetaSlots ? 0 : lv < kMetaSlots + kEaLanes ? 1 : lv < kMetaSlots + kEaLanes + 4 ? 2 : 3;
        const int mt_slot = mt_kind == 0 ? lv : mt_kind == 1 ? (lv - kMetaSlots) / (DE / 4) : 0;
        const unsigned mt_sub = mt_kind == 1 ? (unsigned)((lv - kMetaSlots) % (DE / 4)) * 16u : 0u;
        // occupied record slots: bits 16 g .. 16 g + 3 of the ballot belong to node group g (only lanes q < 4 hold a record)
        const unsigned long long bal = __ballot(rs >= 0);
        PassMeta pm;
        // this lane's node: degree | (packed slots of the groups before it) << 8 — one register across the pipeline stage
        pm.deg = __popcll((bal >> (16 * j)) & 0xFull) | (__popcll(bal & ((1ull << (16 * j)) - 1ull)) << 8);
        pm.off = 0;
        pm.tot = __builtin_amdgcn_readfirstlane(__popcll(bal));
        const int d0 = __popc((unsigned)(bal & 0xF)), d1 = __popc((unsigned)((bal >> 16) & 0xF)),
                  d2 = __popc((unsigned)((bal >> 32) & 0xF)), d3 = __popc((unsigned)((bal >> 48) & 0xF));
        pm.dmax = max(max(d0, d1), max(d2, d3));                               // scalar: the slot loops branch on it
        if (pm.tot == 0) return pm;
        // side table piece: packed slot t is owned by lane 16 g + (t - off_g); g = number of group boundaries at or below t
        const int t = min(mt_slot, pm.tot - 1);
        const int og = (t >= d0) + (t >= d0 + d1) + (t >= d0 + d1 + d2);
        const int ooff = (og > 0 ? d0 : 0) + (og > 1 ? d1 : 0) + (og > 2 ? d2 : 0);
        const int owner = 16 * og + (t - ooff);
        const int m_src = __shfl(rs, owner, 64), m_eid = __shfl(re, owner, 64);
        int sk[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) sk[k] = row_bcast_i(rs, k);      // slot k's source sits in lane k of the node's row: DPP, no LDS trip
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sk[0]), "+v"(sk[1]), "+v"(sk[2]), "+v"(sk[3]) : : "memory");
        WSTAMP(3);
        // LDS byte address of the side table as an integer (casting the generic pointer costs a 64-bit value and a null check)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 4u * (unsigned)(WSZ + 64 + (wave * 2 + sel) * kMetaF));
        const int n_i = min(4 * pass + (lv - kMetaSlots - kEaLanes), a.N - 1);
        const unsigned off = mt_kind == 1 ? (unsigned)m_eid * (unsigned)(DE * 4) + mt_sub
                           : mt_kind == 2 ? (unsigned)max(n_i, 0) * 32u : (unsigned)m_src * 32u + 16u;
        if (mt_kind == 1) dma16(a.edge_attr, off, dst);
        else dma16(a.a_ij, off, dst);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (k < pm.dmax) {
                const unsigned ro = (unsigned)max(sk[k] >= 0 ? sk[k] : sk[0], 0) * row_bytes + qoff;
#pragma unroll
                for (int h = 0; h < H; ++h) rows[k][h] = ld4o(a.xw, ro + (unsigned)h * head_bytes);
            }                                                 // slots >= dmax keep stale registers: compute() skips them the same way
        }
        return pm;
    };

    float4 r_acc[H];
    float4 r_ms = f4zero();                                   // lane q = 0: segment maxima, q = 1: exp-sums (the two halves of a stats row)
    int r_n = -1;
    auto compute = [&](int pass, const PassMeta& pm, int sel, const float4 (&rows)[CH][H]) {
        LANE_CONSTS(); (void)qoff;
        const int n = 4 * pass + j;
#pragma unroll
        for (int h = 0; h < H; ++h) r_acc[h] = f4zero();
        if (n >= a.N || pass >= npass) { r_n = -1; return; }
        r_n = n;
        const float* meta = wbase + sel * kMetaF;
        const int deg = pm.deg & 0xFF, off = pm.deg >> 8;
        if constexpr (ONEHOT && DE == 4) {
            // QUAD layout of the per-edge scalar work.  The logits, the segment softmax and the bond type of an edge are the same in
            // all 16 lanes of its node; computing them there costs the vector ALU 16 x the instructions (SQ_INSTS_VALU: 480 per
            // 4-node pass, the VALU pipe of a SIMD 50 % busy beside a 36 % busy matrix pipe).  Here lane (node j, head hh, slot kk) =
            // 16 j + 4 hh + kk computes ITS (edge, head) once; quad DPP gives the segment max and the ordered sum, row_newbcast hands
            // the weights to the node's 16 row lanes.  Same operations on the same operands in the same order: bit-identical.
            const int hh = (lv >> 2) & 3, kk = lv & 3, hc = hh < H ? hh : H - 1;
            float p = 0.f, mq = 0.f, sq = 0.f, inv = 1e16f;
            int wr = 0;
            if (deg > 0) {
                const bool valid = kk < deg;
                const int slot = off + (valid ? kk : 0);
                const float aj = meta[slot * 4 + hc];
                const float4 ea = ld4(meta + (kMetaSlots + slot) * 4);
                const float ai = meta[(kMetaSlots + kEaLanes + j) * 4 + hc];
                const float4 mc = ld4(s_mt + hc * 4);
                float ee = 0.f;
                ee = fmaf(ea.x, mc.x, ee); ee = fmaf(ea.y, mc.y, ee); ee = fmaf(ea.z, mc.z, ee); ee = fmaf(ea.w, mc.w, ee);
                const float lk = leaky(ai + ee + aj, a.slope);
                float m = valid ? lk : -INFINITY;
                m = fmaxf(m, __builtin_bit_cast(float, dpp_int<0xB1>(__builtin_bit_cast(int, m))));     // quad_perm [1,0,3,2]
                m = fmaxf(m, __builtin_bit_cast(float, dpp_int<0x4E>(__builtin_bit_cast(int, m))));     // quad_perm [2,3,0,1]
                p = valid ? softmax_exp(lk - m) : 0.f;
                const int pi = __builtin_bit_cast(int, p);
                // ((p0 + p1) + p2) + p3: the order of the per-slot loop (an empty slot adds an exact zero)
                sq = ((__builtin_bit_cast(float, dpp_int<0x00>(pi)) + __builtin_bit_cast(float, dpp_int<0x55>(pi))) +
                      __builtin_bit_cast(float, dpp_int<0xAA>(pi))) + __builtin_bit_cast(float, dpp_int<0xFF>(pi));
                inv = 1.f / (sq + 1e-16f);
                mq = m;
                int t = 0;
                t = ea.y != 0.f ? 1 : t; t = ea.z != 0.f ? 2 : t; t = ea.w != 0.f ? 3 : t;
                wr = t * HC;                                  // float offset of the edge's W_edge row (head 0): e_ij is that row
                int tk[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) tk[k] = row_bcast_i(wr, k) + (qok ? q : 0) * 4;       // slot k's row: lane (hh = 0, kk = k)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float4 er[CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (k < pm.dmax) er[k] = ld4(s_w + tk[k] + h * Cp);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (k < pm.dmax) {
                            const float pw = row_bcast(p, 4 * h + k);
                            const float4 xj = er[k] * rows[k][h];
                            fma4(r_acc[h], pw, xj);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                r_acc[h] = row_bcast(inv, 4 * h) * r_acc[h];
                const float mh = row_bcast(mq, 4 * h), sh = row_bcast(sq, 4 * h);
                (&r_ms.x)[h] = q == 0 ? mh : sh;
            }
            return;
        }
        float m[H], ssum[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { m[h] = -INFINITY; ssum[h] = 0.f; }
        if (deg > 0) {
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
                val[k] = k < deg;
                if (k < pm.dmax) {
                    const int slot = off + (val[k] ? k : 0);
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
            (&r_ms.x)[h] = q == 0 ? (deg > 0 ? m[h] : 0.f) : ssum[h];
        }
    };
    // this wave's four rows of local tile `it` go into ring slot it % kWsRing (rows past N are zero: out = bias, never stored)
    auto publish = [&](int it) {
        LANE_CONSTS(); (void)qoff;
        const int slot = it % kWsRing;
        if (it >= kWsRing) {                                  // the consumers must have taken the slot's previous tile
            const int want = kWsCons * (it / kWsRing);
            while (flag_load(s_taken + slot) < want) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (qok) {
            float* tl = s_ring + slot * 16 * LDT + (rw * 4 + j) * LDT + q * 4;
#pragma unroll
            for (int h = 0; h < H; ++h) st4(tl + h * Cp, r_acc[h]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this pass's LDS traffic is done (also guards the side-table buffer)
        if (lv == 0) flag_bump(s_ready + slot);
    };
    auto store_results = [&]() {
        if (r_n < 0) return;
        LANE_CONSTS(); (void)qoff; (void)j;
        if (qok) {
            const unsigned orow = (unsigned)r_n * row_bytes + (unsigned)q * 16u;
#pragma unroll
            for (int h = 0; h < H; ++h) st4o(a.aggr, orow + (unsigned)h * head_bytes, r_acc[h]);
        }
        if (q < 2) st4o(a.stats, (unsigned)r_n * 32u + (unsigned)q * 16u, r_ms);
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
    const int pass_end = ntiles << 2;                         // whole 16-node tiles: every producer publishes every tile of its block
    int it = grp;                                             // local tile index of `pass`
    for (; pass - rw < pass_end; pass += 2 * GW, it += 2 * PG) {
        WSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_a);
        WSTAMP(1);
        store_results();
        WSTAMP(2);
        PassMeta pm_nxt = prefetch(pass + GW, rs_nxt, re_nxt, 1, rows_b);
        WSTAMP(4);
        load_rec(pass + 2 * GW, rs_nxt, re_nxt);
        WSTAMP(5);
        compute(pass, pm_cur, 0, rows_a);
        WSTAMP(6);
        publish(it);
        WSTAMP(7);

        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs_nxt), "+v"(re_nxt) : : "memory");
        settle(rows_b);
        WSTAMP(1);
        store_results();
        WSTAMP(2);
        pm_cur = prefetch(pass + 2 * GW, rs_nxt, re_nxt, 0, rows_a);
        WSTAMP(4);
        load_rec(pass + 3 * GW, rs_nxt, re_nxt);
        WSTAMP(5);
        if (pass + GW - rw < pass_end) {
            compute(pass + GW, pm_nxt, 1, rows_b);
            WSTAMP(6);
            publish(it + PG);
            WSTAMP(7);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_results();
#ifdef GLAM_WS_PROF
    if (lane == 0 && blockIdx.x < 64) for (int k = 0; k < 8; ++k) g_ws_prof[(blockIdx.x * 12 + wave) * 8 + k] = pacc[k];
#endif
}

static size_t ws_lds_bytes(int H, int Cp, int De, int P) {
    const int HC = H * Cp;
    return ((size_t)De * HC + 64 + (size_t)P * 2 * 64 * 4 + (size_t)kWsRing * 16 * (HC + 4)) * sizeof(float);
}

template <int H, int DE, bool ONEHOT, int P>
static void launch_ws_p(const FwdDmaArgs& a, int grid, hipStream_t s) {
    static bool big = false;
    if (!big) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_triplet_fwd_ws<H, DE, ONEHOT, P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        big = true;
    }
    GLAM_PROF_LABEL("k_triplet_fwd_ws+update");
    hipLaunchKernelGGL((k_triplet_fwd_ws<H, DE, ONEHOT, P>), dim3(grid), dim3((P + kWsCons) * 64), ws_lds_bytes(H, a.Cp, DE, P), s, a);
}
template <int H, int DE, bool ONEHOT>
static void launch_ws(const FwdDmaArgs& a, int grid, hipStream_t s) {
    // eight producers (three waves per SIMD) where the kernel fits 168 registers without scratch, four otherwise
    const char* e = getenv("GLAM_WS_PROD");     // developer A/B: producer waves per block
    if constexpr (H <= 3) {
        if (!(e && atoi(e) == 4)) { launch_ws_p<H, DE, ONEHOT, 8>(a, grid, s); return; }
    }
    launch_ws_p<H, DE, ONEHOT, 4>(a, grid, s);
}

// the warp-specialised kernel exists for one-hot bond features of width 4 (src_1gp/dataset.py:82: every molecular dataset of the reference)
bool triplet_fwd_ws_supported(int H, int Cp, int De, int edge_onehot) {
    return edge_onehot && De == 4 && H >= 1 && H <= 4 && (Cp >> 2) > 8 && (Cp >> 2) <= 16 && H * Cp <= 192;
}

bool triplet_fwd_ws_enabled() {
    const char* e = getenv("GLAM_FWD_WS");     // read per call: an A/B switch for experiments (a captured graph keeps what it captured)
    return !e || atoi(e) != 0;
}

// forward aggregate + update GEMM for molecular graphs, warp-specialised (same contract as triplet_fwd_pipe_fused)
int triplet_fwd_ws(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                   const int32_t* ell_src, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De, float slope,
                   int edge_onehot, float* aggr, float* stats, const float* img_upd, const float* bias_p, float* out, hipStream_t s) {
    if (N == 0) return GLAM_OK;
    if (!(H >= 1 && H <= 4 && (De == 4 || De == 8) && (Cp >> 2) > 8 && (Cp >> 2) <= 16 && H * Cp <= 192))
        return fail(GLAM_E_UNSUPPORTED, "triplet_fwd_ws: H=%d Cp=%d De=%d outside the fused table (36 <= Cp <= 64, H*Cp <= 192)", H, Cp, De);
    if ((uint64_t)N * H * Cp * 4 >= (1ull << 32) || (uint64_t)E * De * 4 >= (1ull << 32))
        return fail(GLAM_E_UNSUPPORTED, "triplet_fwd_ws: a tensor exceeds 4 GiB (32-bit offsets)");
    FwdDmaArgs a{xw, a_ij, edge_attr, w_edge, M, ell_src, ell_eid, (int)N, Cp, slope, aggr, stats, img_upd, bias_p, out};
    const int ntiles = (int)((N + 15) / 16);
    const char* ge = getenv("GLAM_WS_GRID");
    const int cap = ge ? atoi(ge) : 256;
    const int grid = ntiles < cap ? ntiles : cap;           // one 8-wave block per CU
    if (!triplet_fwd_ws_supported(H, Cp, De, edge_onehot))
        return fail(GLAM_E_UNSUPPORTED, "triplet_fwd_ws: needs one-hot edge features of width 4 (H=%d Cp=%d De=%d onehot=%d)", H, Cp, De, edge_onehot);
    switch (H) {
        case 1: launch_ws<1, 4, true>(a, grid, s); break;
        case 2: launch_ws<2, 4, true>(a, grid, s); break;
        case 3: launch_ws<3, 4, true>(a, grid, s); break;
        default: launch_ws<4, 4, true>(a, grid, s); break;
    }
    GLAM_LAUNCH_CHECK("triplet_fwd_ws");
    return GLAM_OK;
}

}  // namespace glam

#ifdef GLAM_WS_PROF
extern "C" int glam_debug_ws_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_ws_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
