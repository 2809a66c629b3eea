// Warp-specialised backward by target (B1) of the TripletMessage layer for molecular graphs (ELL index records by target, one-hot bond
// features of width 4), with d_aggr = d_out @ W_scale^T produced inside the launch.  The roles of k_triplet_fwd_ws (triplet_ws.hip) are
// mirrored:
//     matrix waves V..V+3   produce the 16-node d_aggr tiles on the fp32 matrix cores, AHEAD of their use: wave w owns column tiles
//                           3 w .. 3 w + 2 (16 columns each), keeps that 64 x 48 slice of W_scale^T in 48 registers for the whole
//                           launch, reads the d_out rows of a tile as plain (prefetched) loads and writes the tile into an LDS ring slot
//     vector waves 0..V-1   consume the tiles: per pass of 4 nodes read the rows of d_aggr from the slot (and write them to HBM for B2),
//                           gather the neighbours' xw rows (prefetched into registers right after the previous pass's last use of
//                           them), recompute alpha, d_alpha = <d_aggr[dst], e_ij * xw[src]>, softmax + leaky backward -> dpre; store
//                           alpha_e / dpre_e per edge and d_a_i per node; accumulate d_W_edge / d_M
// (general kernel: k_triplet_bwd_dst<..., FD = true>; reference: the autograd of src_1gp/layer.py:42-55).
// Per-edge scalar work runs in the QUAD lane layout (lane 16 j + 4 h + k = node j, head h, slot k computes its (edge, head) once) on
// operands each lane loads ITSELF one pass ahead (plain 4-byte gathers: no side table, no LDS-DMA, every wait counted by the compiler);
// the channel work (d_aggr * xw chunks, the 60-channel dots) runs in the row layout (16 lanes per node).
// d_W_edge accumulates in LDS, one private array [4 bond types][H*Cp] per (vector wave, node row of the pass), by plain
// read - fma - write of the row of the edge's bond type (LDS executes a wave's operations in order, so two slots of one node with the
// same type chain correctly) instead of 48 registers per lane — which is what kept the general kernel at one pass in flight and 256
// registers.  (ds_add_f32 into one array per wave was tried first: 54 vs 21 us at B = 1 024 — the LDS retires float atomics a lane
// at a time.)  d_M in four registers per lane.
// alpha_e, dpre_e, d_a_i and d_aggr are bit-identical to the general kernel (same operations, same order); the d_W_edge / d_M block
// partials are sums in a different (still fixed) order.
#include "triplet_pipe.h"

#include <stdlib.h>
#include <type_traits>

#ifndef GLAM_B1_EARLY
#define GLAM_B1_EARLY 0      // 1: the matrix waves' first product AHEAD of the block's barrier (measured even: 12.6 us either way at B = 1 024 — the
                             // barrier moves from 6.2 k to 7.1 k cycles, the first tile is out 1.6 k behind it in both forms; bit-identical)
#endif
#ifndef GLAM_B1_MPRIO
#define GLAM_B1_MPRIO 0      // s_setprio of the matrix waves (0: none; 3 measured: B = 1 024 even, B = 16 384 +5 %: NEGATIVES.md)
#endif

namespace glam {

#ifdef GLAM_WS_TL      // timeline stamps (tools/ws_timeline.py; see triplet_ws.hip): [block][wave][stamp]
__device__ long long g_b1_tl[256 * 12 * 6];
__device__ long long g_b1_rt[256 * 12 * 6];      // the same stamps on the device-wide 100 MHz counter
#define B1_TL(k) do { unsigned long long c_, r_; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_), "=s"(r_) :: "memory"); \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) { const int i_ = (blockIdx.x * 12 + (threadIdx.x >> 6)) * 6 + (k); \
    g_b1_tl[i_] = (long long)c_; g_b1_rt[i_] = (long long)r_; } } while (0)
#else
#define B1_TL(k) do { } while (0)
#endif


// part[k][h] (k < DM slots, h < H heads) summed over the 16 lanes of a node's row, sum (k, h) DELIVERED TO LANE 4 h + k of the row — the
// quad layout of the per-edge scalars — instead of DM * H full butterflies (every lane ending with every total) and as many selects.
// The same pairings in the same order as group_sum<16> (xor 1, xor 2, row_half_mirror, row_mirror), so every total is bit-equal to it;
// at each step a lane keeps the half of its values its position asks for and hands the other half to its partner: DM * H + ... -> 1
// values per lane.  The mirror steps pair lane i with 7 - i / 15 - i, whose slot index is the reverse one: the odd quads work on
// reversed slots (kc = 3 - kk) and are turned round at the end.  Lanes whose (k, h) does not exist end with finite garbage.
template <int DM, int H>
__device__ __forceinline__ float row_sums_to_quads(const float (&part)[DM][H], int lv) {
    const int qd = (lv >> 2) & 3, kk = lv & 3;
    const bool oddq = qd & 1;
    const int kc = oddq ? 3 - kk : kk;
    const bool p1 = kc & 1, p2 = kc & 2;
    float w0[H], w1[H], u[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        if constexpr (DM >= 2) {
            const float keep = p1 ? part[1][h] : part[0][h], give = p1 ? part[0][h] : part[1][h];
            w0[h] = keep + dpp_f<0xB1>(give);                 // quad_perm [1,0,3,2]
        } else {
            w0[h] = part[0][h] + dpp_f<0xB1>(part[0][h]);
        }
        if constexpr (DM >= 4) {
            const float keep = p1 ? part[3][h] : part[2][h], give = p1 ? part[2][h] : part[3][h];
            w1[h] = keep + dpp_f<0xB1>(give);
        } else if constexpr (DM == 3) {
            w1[h] = part[2][h] + dpp_f<0xB1>(part[2][h]);
        } else {
            w1[h] = 0.f;
        }
        if constexpr (DM >= 3) {
            const float keep = p2 ? w1[h] : w0[h], give = p2 ? w0[h] : w1[h];
            u[h] = keep + dpp_f<0x4E>(give);                  // quad_perm [2,3,0,1]
        } else {
            u[h] = w0[h] + dpp_f<0x4E>(w0[h]);
        }
    }
    // heads: quads 0 and 3 keep head 0, quads 1 and 2 head 1 across the 8-lane halves; then quads 0, 1 keep that and quads 2, 3 head 2
    float r;
    if constexpr (H == 1) {
        const float a8 = u[0] + dpp_f<0x141>(u[0]);           // row_half_mirror
        r = a8 + dpp_f<0x140>(a8);                            // row_mirror
    } else {
        const bool p3 = qd == 0 || qd == 3;
        const float keep = p3 ? u[0] : u[1], give = p3 ? u[1] : u[0];
        const float a8 = keep + dpp_f<0x141>(give);
        if constexpr (H == 2) {
            r = a8 + dpp_f<0x140>(a8);
        } else {
            const float b8 = u[2] + dpp_f<0x141>(u[2]);
            const bool p4 = qd < 2;
            const float keep4 = p4 ? a8 : b8, give4 = p4 ? b8 : a8;
            r = keep4 + dpp_f<0x140>(give4);
        }
    }
    const float rr = dpp_f<0x1B>(r);                          // quad_perm [3,2,1,0]
    return oddq ? rr : r;
}

struct DstWsArgs {
    const float* xw; const float* a_ij; const float* edge_attr; const float* w_edge; const float* M;
    const float* aggr; const float* stats; const float* d_out; const float* img_dagg;
    const int* ell_src; const int* ell_eid;      // [N][4] each, by target
    int N; int Cp; float slope;
    float* d_aggr; float* alpha_e; float* dpre_e; float* d_a_ij; float* partial;      // partial[gridDim.x][4 * H * Cp + 16]
    const float* dagg_pre;      // may be null: W_scale^T as the matrix waves' operand fragments, split already (layer.hip: Staged::dagg_pre)
};

template <int H, int V, bool X3, bool WT = false>      // WT: the outputs are written through the L2 (small launches; common.h: st4o_wt)
__global__ void __launch_bounds__((V + 4) * 64, (V + 4) / 4) k_triplet_bwd_dst_ws(DstWsArgs a) {
    constexpr int kBlockT = (V + 4) * 64, VG = V / 4, CH = 4;
    typedef float v4f __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Cp = a.Cp, Q = Cp >> 2, HC = H * Cp;
    // W_edge and the d_W_edge arrays keep one row per bond type at a pitch of WP floats, a multiple of 64 (and so is the array size WL):
    // rows of different types / different node rows start on the same bank, so a 16-lane ds_read_b128 / ds_write_b128 group that mixes
    // lanes of two nodes is conflict free (pitch HC = 180: 35-43 % of this kernel's LDS cycles were bank conflicts)
    // The d_W_edge arrays take a head per 64 floats (WD = 64 H per bond type): every one of a row's 16 lanes owns a float4 of every head —
    // also the lanes beyond the row's Cp / 4 chunks, whose products are zero — so the accumulator update below is unconditional
    constexpr int WP = ws_wedge_pitch_h(H), WL = 4 * WP, WD = WP, WLD = 4 * WD;
    const int WSZ = 4 * HC, LDT = HC + 8, P = WSZ + 16;
    constexpr int kRing = 4;                                  // tile slots (the d_W_edge arrays take the rest of the LDS)
    float* s_w = smem;
    int* s_ready = reinterpret_cast<int*>(smem + WL);         // [kRing] matrix-wave check-ins per slot
    float* s_mt = reinterpret_cast<float*>(s_ready + 16);     // M transposed: [head][edge feature]
    int* s_taken = s_ready + 32;                              // [kRing] vector-wave check-outs per slot
    float* s_dw = smem + WL + 64;                             // per (vector wave, node row): d_W_edge [4][H][64]
    float* s_ring = s_dw + V * 4 * WLD;                       // kRing tiles of 16 x LDT floats
    B1_TL(0);
    // LDS staging, called by each role BEHIND its first global loads (see k_triplet_fwd_ws): the matrix waves' weight slice and first
    // d_out rows, the vector waves' first record
    auto stage_lds = [&]() {
        const float4 wv = tid < WSZ / 4 ? ld4(a.w_edge + 4 * tid) : f4zero();         // (WSZ / 4 <= 192 < the block)
        const float mv = (tid >> 4) == 1 ? a.M[(tid & 3) * 4 + ((tid >> 2) & 3)] : 0.f;
        for (int i = tid; i < V * WLD; i += kBlockT) st4(s_dw + 4 * i, f4zero());      // V * 4 arrays of WLD floats: under the loads
        if (tid < WSZ / 4) st4(s_w + (4 * tid) / HC * WP + (4 * tid) % HC, wv);
        if (tid < 64) {
            if ((tid >> 4) == 1) s_mt[tid & 15] = mv;
            else s_ready[tid] = 0;
        }
    };
    const int ntiles = (a.N + 15) >> 4;       // (the barrier that publishes the LDS initialisation sits behind each role's first global loads)

    if (wave >= V) {
        // ------------------------------------------------------------------------------------------------------------------
        // matrix waves: d_aggr tile[16, HC] = d_out[16 tile .. +15, Cp] @ W_scale^T, columns 48 w .. 48 w + 47 in this wave
        // ------------------------------------------------------------------------------------------------------------------
#if GLAM_B1_MPRIO
        // The vector waves WAIT for these tiles (r6_ws_timeline_b*.txt: a tile every 2.9 k cycles from the matrix waves against 1.9 k the
        // two gather groups could take: the launch runs at the matrix waves' pace at both sizes).  Their vector work — the operand split,
        // the addresses of the tile stores — shares a SIMD's issue slots with two gather waves: static priority puts it first.
        __builtin_amdgcn_s_setprio(GLAM_B1_MPRIO);
#endif
        const int w = wave - V, c = lane & 15, kq = lane >> 4;
        const int MP = HC <= 64 ? 64 : 192;                   // positions per image row
        auto publish_tile = [&](int it_, const v4f (&acc)[3]) {
            const int slot = it_ % kRing;
            float* tl = s_ring + slot * 16 * LDT;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const int mcol = 16 * (3 * w + ct) + c;
                if (mcol < HC) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) tl[(4 * kq + i) * LDT + mcol] = acc[ct][i];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_ready + slot);
        };
        // the X3 product's result layout (W as the first operand): lane (node c, column block kq) holds columns 16 (3 w + ct) + 4 kq .. + 3
        auto publish_tile4 = [&](int it_, const v4f (&acc)[3]) {
            const int slot = it_ % kRing;
            float* tl = s_ring + slot * 16 * LDT + c * LDT + 4 * kq;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const int m0 = 16 * (3 * w + ct);
                if (m0 + 4 * kq < HC) st4(tl + m0, make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_ready + slot);
        };
        auto wait_slot = [&](int it_) {
            const int slot = it_ % kRing;
            if (it_ >= kRing) {                               // the vector waves must have taken the slot's previous tile
                const int want = 4 * (it_ / kRing);
                while (flag_load(s_taken + slot) < want) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        };
        if constexpr (X3) {
            // ---- 3 x bf16 form (bf16x3.h, round 4): lane (row c, k block kq) holds 8 consecutive k of a d_out row per 32-k step; the wave's
            //      64 x 48 slice of W_scale^T is split once into 72 registers; 36 v_mfma_f32_16x16x32_bf16 per tile (three column tiles x
            //      two steps x six partial products, small ones first) replace 48 fp32 MFMAs that ran INSTEAD of the vector waves'
            //      instructions on the SIMD's one fp32 datapath ----
            const int Kp = (Cp + 15) & ~15;
            Bf16x3 wreg[2][3];
            WRaw8 raw[2][3];
            const bool pre = a.dagg_pre != nullptr;           // (block-uniform)
            if (pre) {
                // the fragments split already, in lane order: 36 coalesced 1 KB loads, no vector work (the splits — 400 instructions per
                // wave — stood between the weight loads and the FIRST tile every vector wave of the block waits for)
                const char* fp = reinterpret_cast<const char*>(a.dagg_pre) + ((size_t)(w * 6) * 3 * 64 + lane) * 16;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) {
                        const char* q = fp + (size_t)((st * 3 + ct) * 3) * 1024;
                        wreg[st][ct].hi = ldfrag(q);
                        wreg[st][ct].mid = ldfrag(q + 1024);
                        wreg[st][ct].lo = ldfrag(q + 2048);
                    }
            } else {
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const int mcol = min(16 * (3 * w + ct) + c, MP - 1);
                const int pos = (mcol & ~63) + (mcol & 3) * 16 + ((mcol >> 2) & 15);     // ts_pos_of_col
#pragma unroll
                for (int st = 0; st < 2; ++st) raw[st][ct] = w_load8(a.img_dagg, MP, pos, 32 * st + 8 * kq, Kp);
            }
            }
            auto load_a = [&](int tile, float4 (&af)[2][2]) {
                const int row = 16 * tile + c;
                const bool rok = tile < ntiles && row < a.N;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k0 = 32 * st + 8 * kq + 4 * u;
                        af[st][u] = (rok && k0 < Cp) ? ld4(a.d_out + (size_t)row * Cp + k0) : f4zero();
                    }
            };
            float4 af[2][2];
            int tile = blockIdx.x, it = 0;
            load_a(tile, af);
            stage_lds();
            if (!pre) {
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int st = 0; st < 2; ++st) wreg[st][ct] = w_split8(raw[st][ct], 32 * st + 8 * kq, Kp, 16 * (3 * w + ct) + c < HC);
            }
            // the product of one tile out of `af` (whose registers take the NEXT tile's rows under the matrix instructions)
            auto product = [&](int next_tile, v4f (&acc)[3]) {
#ifdef GLAM_B1_NOMM      // timing experiment only (wrong numbers): the matrix waves without their operand split and matrix instructions
                load_a(next_tile, af);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) acc[ct] = (v4f){af[0][0].x, af[0][1].y, af[1][0].z, af[1][1].w};
                return;
#endif
                Bf16x3 as[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) as[st] = split8(af[st][0], af[st][1]);
                load_a(next_tile, af);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) acc[ct] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[ct] = mfma_x3_small(wreg[st][ct], as[st], acc[ct]);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[ct] = mfma_x3_mid(wreg[st][ct], as[st], acc[ct]);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[ct] = mfma_x3_big(wreg[st][ct], as[st], acc[ct]);
            };
            // (GLAM_B1_EARLY: the FIRST tile's product ahead of the block's barrier — it needs nothing the barrier publishes, only its store
            //  into the ring and the check-in flag do; measured even, see the macro)
#if GLAM_B1_EARLY
            v4f acc0[3];
            product(tile + gridDim.x, acc0);                  // (every block has a first tile: grid <= ntiles)
            __syncthreads();                                  // LDS initialised
            B1_TL(1);
            publish_tile4(0, acc0);
            B1_TL(2);                                         // (timeline builds: the first tile is out; the fourth)
            tile += gridDim.x;
            it = 1;
#else
            __syncthreads();
            B1_TL(1);
#endif
            for (; tile < ntiles; tile += gridDim.x, ++it) {
                v4f acc[3];
                wait_slot(it);
                product(tile + gridDim.x, acc);
                publish_tile4(it, acc);
                if (it == 0) B1_TL(2);
                if (it == 3) B1_TL(4);
            }
        } else {
        const int GK = (Cp + 15) >> 4;                        // k groups (<= 4)
        float4 bf[3][4];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int mcol = 16 * (3 * w + ct) + c;
            const int pos = (mcol & ~63) + (mcol & 3) * 16 + ((mcol >> 2) & 15);     // ts_pos_of_col
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bf[ct][g] = (g < GK && mcol < HC) ? ld4(a.img_dagg + ((size_t)(4 * g + kq) * MP + pos) * 4) : f4zero();
        }
        auto load_a = [&](int tile, float4 (&af)[4]) {
            const int row = 16 * tile + c;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 4 * g + kq;                    // 16-byte chunk of the row
                af[g] = (tile < ntiles && row < a.N && 4 * ch < Cp) ? ld4(a.d_out + (size_t)row * Cp + 4 * ch) : f4zero();
            }
        };
        float4 af_a[4], af_b[4];
        int tile = blockIdx.x, it = 0;
        load_a(tile, af_a);
        stage_lds();
        __syncthreads();                                      // LDS initialised; the weight slice and the first tile's rows are in flight
        auto one_tile = [&](int it_, const float4 (&af)[4]) {
            wait_slot(it_);
            v4f acc[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[ct] = (v4f){0.f, 0.f, 0.f, 0.f};
            // k group outer, column tile inner: three independent accumulator chains
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < GK) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int ct = 0; ct < 3; ++ct)
                            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(af[g], jj), f4get(bf[ct][g], jj), acc[ct], 0, 0, 0);
                }
            }
            publish_tile(it_, acc);
        };
        for (; tile < ntiles; tile += 2 * gridDim.x, it += 2) {
            load_a(tile + gridDim.x, af_b);                   // next tile's rows in flight under this tile's MFMAs
            one_tile(it, af_a);
            if (tile + (int)gridDim.x < ntiles) {
                load_a(tile + 2 * gridDim.x, af_a);
                one_tile(it + 1, af_b);
            }
        }
        }
    } else {
        // ------------------------------------------------------------------------------------------------------------------
        // vector waves
        // ------------------------------------------------------------------------------------------------------------------
        float* wave_dw = s_dw + wave * 4 * WLD;
        const int npass = (a.N + 3) >> 2;
        const int grp = wave >> 2, rw = wave & 3;
        const int gw = 4 * (blockIdx.x + grp * gridDim.x) + rw, GW = 4 * VG * gridDim.x;
        const unsigned row_bytes = (unsigned)HC * 4u, head_bytes = (unsigned)Cp * 4u;
        int lv = lane;
#define LANE_CONSTS()                                                                        \
        asm volatile("" : "+v"(lv));                                                         \
        const int j = lv >> 4, q = lv & 15;                                                  \
        const bool qok = q < Q;                                                              \
        const unsigned qoff = (unsigned)(qok ? q : 0) * 16u;                                 \
        const int hh = (lv >> 2) & 3, kk = lv & 3, hc = hh < H ? hh : H - 1

        // record in the QUAD layout: lane (j, hh, kk) holds slot kk of node j (every quad of a row loads the same four words)
        auto load_rec = [&](int pass, int& rs, int& re) {
            LANE_CONSTS(); (void)qok; (void)qoff; (void)q; (void)hc; (void)hh;
            const int n = 4 * pass + j;
            rs = -1; re = -1;
            if (pass < npass && n < a.N) { rs = a.ell_src[4 * n + kk]; re = a.ell_eid[4 * n + kk]; }
        };
        // per-edge / per-node scalars of the lane's (edge, head), loaded by the lane itself one pass ahead
        struct Scal { float aj, ai, m, s; float4 ea; };
        float4 rows[CH][H];
        Scal sc;
        sc.aj = sc.ai = sc.m = sc.s = 0.f; sc.ea = f4zero();
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int h = 0; h < H; ++h) rows[k][h] = f4zero();
        auto ldf = [](const float* base, unsigned byte_off) {      // scalar-base + 32-bit offset addressing (see ld4o)
            return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
        };
        auto prefetch = [&](int pass, int rs, int re) {
            LANE_CONSTS(); (void)kk; (void)hh;
            if (pass >= npass) return;
            const unsigned n = (unsigned)min(4 * pass + j, a.N - 1);
            const unsigned long long bal = __ballot(rs >= 0) & 0x000F000F000F000Full;        // the hh = 0 quads carry the degree
            const int d0 = __popc((unsigned)(bal & 0xF)), d1 = __popc((unsigned)((bal >> 16) & 0xF)),
                      d2 = __popc((unsigned)((bal >> 32) & 0xF)), d3 = __popc((unsigned)((bal >> 48) & 0xF));
            const int dmax = max(max(d0, d1), max(d2, d3));
            const int src0 = row_bcast_i(rs, 0), eid0 = row_bcast_i(re, 0);
            // an empty slot aliases the node's first edge (finite data, weight 0); a node without edges reads nothing (E may be 0)
            sc.aj = 0.f; sc.ea = f4zero();
            if (src0 >= 0) {
                sc.aj = ldf(a.a_ij, (unsigned)(rs >= 0 ? rs : src0) * 32u + 16u + 4u * (unsigned)hc);
                sc.ea = ld4o(a.edge_attr, (unsigned)(re >= 0 ? re : eid0) * 16u);
            }
            sc.ai = ldf(a.a_ij, n * 32u + 4u * (unsigned)hc);
            sc.m = ldf(a.stats, n * 32u + 4u * (unsigned)hc);
            sc.s = ldf(a.stats, n * 32u + 16u + 4u * (unsigned)hc);
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (k < dmax) {
                    const int sk = row_bcast_i(rs, k);
                    const unsigned ro = (unsigned)max(sk >= 0 ? sk : src0, 0) * row_bytes + qoff;
#pragma unroll
                    for (int h = 0; h < H; ++h) rows[k][h] = ld4o(a.xw, ro + (unsigned)h * head_bytes);
                }
            }
        };

        float dMq[4] = {0.f, 0.f, 0.f, 0.f};                  // d_M[bond type][head hh of this lane]
        // what the first half of a pass hands to the second (across the prefetch of the next pass)
        float h_pre = 0.f, h_alpha = 0.f, h_dal = 0.f;
        int h_t = 0;
        // first half: phases that need the gathered rows
        auto compute_a = [&](auto dm_tag, int pass, int it_, int rs) {
            constexpr int DM = decltype(dm_tag)::value;
            LANE_CONSTS(); (void)qoff;
            const int n = 4 * pass + j;
            const bool node_ok = pass < npass && n < a.N;
            const unsigned long long bal = __ballot(rs >= 0) & 0x000F000F000F000Full;
            const int deg = __popcll((bal >> (16 * j)) & 0xFull);
            // ---- quad phase 1: logit, alpha, bond type of this lane's (edge, head) ----
            const float4 mc = ld4(s_mt + hc * 4);
            float ee = 0.f;
            ee = fmaf(sc.ea.x, mc.x, ee); ee = fmaf(sc.ea.y, mc.y, ee); ee = fmaf(sc.ea.z, mc.z, ee); ee = fmaf(sc.ea.w, mc.w, ee);
            const float pre = sc.ai + ee + sc.aj;
            const float inv = 1.f / (sc.s + 1e-16f);
            const float alpha = softmax_exp(leaky(pre, a.slope) - sc.m) * inv;
            int t = 0;
            t = sc.ea.y != 0.f ? 1 : t; t = sc.ea.z != 0.f ? 2 : t; t = sc.ea.w != 0.f ? 3 : t;
            h_pre = pre; h_alpha = alpha; h_t = t;
            // ---- the node's d_aggr row out of the ring; on to HBM for B2 ----
            const int slot = it_ % kRing;
            {
                const int want = 4 * (it_ / kRing + 1);
                while (flag_load(s_ready + slot) < want) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            float4 dag[H];
            {
                const float* tl = s_ring + slot * 16 * LDT + (rw * 4 + j) * LDT + (qok ? q : 0) * 4;
#pragma unroll
                for (int h = 0; h < H; ++h) dag[h] = ld4(tl + h * Cp);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lv == 0) flag_bump(s_taken + slot);
            if (!qok) {
#pragma unroll
                for (int h = 0; h < H; ++h) dag[h] = f4zero();
            }
            float dalq = 0.f;
            if (node_ok) {
                if (qok) {
                    const unsigned orow = (unsigned)n * row_bytes + (unsigned)q * 16u;
#pragma unroll
                    for (int h = 0; h < H; ++h) st4o_t<WT>(a.d_aggr, orow + (unsigned)h * head_bytes, dag[h]);
                }
                // ---- row phase: d_alpha[k][h] = <d_aggr[n,h,:], e_ij * xw[src_k,h,:]>, d_W_edge[type_k][h] += alpha * d_aggr * xw ----
                if (deg > 0) {
                    // the update's weight: an empty slot (it aliases the node's first edge: finite data) adds an exact zero, so every lane
                    // of the row runs the read - fma - write of every slot — no exec-mask region around each LDS access
                    const float alpha_w = kk < deg ? alpha : 0.f;
                    float part[DM][H];
                    // slot outer, head inner: the three accumulator rows of a slot (one per head: distinct addresses) are read together,
                    // updated and written back, so a pass costs DM LDS round trips for d_W_edge instead of DM * H dependent ones
#pragma unroll
                    for (int k = 0; k < DM; ++k) {
                        float4 er[H], dacc[H];
                        const int tkk = row_bcast_i(t, k);
                        const float* e = s_w + tkk * WP + (qok ? q : 0) * 4;
                        float* d = wave_dw + j * WLD + tkk * WD + q * 4;
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            er[h] = ld4(e + h * Cp);
                            dacc[h] = ld4(d + h * 64);
                        }
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            const float4 tv = dag[h] * rows[k][h];                   // d_aggr * x_j
                            part[k][h] = dot4(tv, er[h]);
                            fma4(dacc[h], row_bcast(alpha_w, 4 * h + k), tv);        // d_W_edge[type_k][h] += alpha * (d_aggr * x_j)
                        }
#pragma unroll
                        for (int h = 0; h < H; ++h) st4(d + h * 64, dacc[h]);
                    }
                    dalq = row_sums_to_quads<DM, H>(part, lv);
                }
            }
            h_dal = dalq;
        };
        auto compute_a_any = [&](int pass, int it_, int rs) {
            const unsigned long long bal = __ballot(rs >= 0) & 0x000F000F000F000Full;
            const int d0 = __popc((unsigned)(bal & 0xF)), d1 = __popc((unsigned)((bal >> 16) & 0xF)),
                      d2 = __popc((unsigned)((bal >> 32) & 0xF)), d3 = __popc((unsigned)((bal >> 48) & 0xF));
            const int dmax = max(max(d0, d1), max(d2, d3));
            if (dmax <= 1) compute_a(std::integral_constant<int, 1>{}, pass, it_, rs);
            else if (dmax == 2) compute_a(std::integral_constant<int, 2>{}, pass, it_, rs);
            else if (dmax == 3) compute_a(std::integral_constant<int, 3>{}, pass, it_, rs);
            else compute_a(std::integral_constant<int, 4>{}, pass, it_, rs);
        };
        // second half: softmax + leaky backward in the quad layout, the per-edge / per-node stores
        auto compute_b = [&](int pass, int rs, int re) {
            LANE_CONSTS(); (void)qok; (void)qoff; (void)hc;
            const int n = 4 * pass + j;
            const bool node_ok = pass < npass && n < a.N;
            const unsigned long long bal = __ballot(rs >= 0) & 0x000F000F000F000Full;
            const int deg = __popcll((bal >> (16 * j)) & 0xFull);
            const bool valid = node_ok && kk < deg;
            const bool live = valid && hh < H;
            // softmax backward: d_logit_e = alpha_e (d_alpha_e - S), S = sum over the node's edges of alpha_e' d_alpha_e', summed where the
            // (at most four) products already sit: the quad of lanes (j, hh, 0..3), in the order of the general kernel's one-chunk nodes
            // (bit-identical to it with GLAM_X3=0).  Until round 5 S was taken as <d_aggr[n,h,:], aggr[n,h,:]> — the same number, aggr being
            // that weighted sum — at the price of a second [N, H, Cp] row read per node: B1 no longer reads aggr (-28 % of its bytes)
            const float pq = live ? h_alpha * h_dal : 0.f;
            const float S = ((dpp_f<0x00>(pq) + dpp_f<0x55>(pq)) + dpp_f<0xAA>(pq)) + dpp_f<0xFF>(pq);
            const float dl = h_alpha * (h_dal - S);
            float dp = h_pre > 0.f ? dl : dl * a.slope;
            dp = live ? dp : 0.f;
            const float al_st = live ? h_alpha : 0.f;
            // d_a_i[n, hh] = ((dp_0 + dp_1) + dp_2) + dp_3 (an empty slot adds an exact zero)
            const float dai = ((dpp_f<0x00>(dp) + dpp_f<0x55>(dp)) + dpp_f<0xAA>(dp)) + dpp_f<0xFF>(dp);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) dMq[tt] += (h_t == tt) ? dp : 0.f;
            // lanes q < 4 (hh = 0) gather the four heads of their edge: row_shl by 4 h brings lane 4 h + kk to lane kk
            float4 av, dv, da;
            av.x = al_st; av.y = dpp_f<0x104>(al_st); av.z = dpp_f<0x108>(al_st); av.w = dpp_f<0x10C>(al_st);
            dv.x = dp; dv.y = dpp_f<0x104>(dp); dv.z = dpp_f<0x108>(dp); dv.w = dpp_f<0x10C>(dp);
            da.x = dai; da.y = dpp_f<0x104>(dai); da.z = dpp_f<0x108>(dai); da.w = dpp_f<0x10C>(dai);
            if (q < 4 && valid) {
                st4o_t<WT>(a.alpha_e, (unsigned)re * 16u, av);
                st4o_t<WT>(a.dpre_e, (unsigned)re * 16u, dv);
            }
            if (q == 0 && node_ok) st4o_t<WT>(a.d_a_ij, (unsigned)n * 32u, da);
        };

        int rs, re, rs_n, re_n;
        int pass = gw;
        load_rec(pass, rs, re);
        stage_lds();
        prefetch(pass, rs, re);
        load_rec(pass + GW, rs_n, re_n);
        __syncthreads();                                      // LDS initialised; the first pass's operands are in flight
        B1_TL(1);
        const int pass_end = ntiles << 2;
        int it = grp;
        for (; pass - rw < pass_end; pass += GW, it += VG) {
            if (it == grp + VG) B1_TL(2);                     // (the first pass of this wave is done)
            compute_a_any(pass, it, rs);
            __builtin_amdgcn_sched_barrier(0);                // the rows' registers are free from here on: everything of pass p + 1
            prefetch(pass + GW, rs_n, re_n);
            __builtin_amdgcn_sched_barrier(0);
            compute_b(pass, rs, re);
            rs = rs_n; re = re_n;
            load_rec(pass + 2 * GW, rs_n, re_n);
        }
        // every tile of this wave is done: the wave's total of d_M[type][head] — over the quad's slots by DPP, over the four node rows
        // through the crossbar — parked by lane 4 hh in the wave's 16 floats of the scratch (the dead ring)
        B1_TL(3);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float x = dMq[tt];
            x += dpp_f<0xB1>(x);
            x += dpp_f<0x4E>(x);
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            dMq[tt] = x;
        }
        __syncthreads();                                      // (1) the ring is dead: its memory becomes the d_M scratch
        if (lane < 16 && (lane & 3) == 0) st4(s_ring + (wave * 4 + (lane >> 2)) * 4, make_float4(dMq[0], dMq[1], dMq[2], dMq[3]));
#undef LANE_CONSTS
    }
    if (wave >= V) { B1_TL(3); __syncthreads(); }             // (1) for the matrix waves
    __syncthreads();                                          // (2)
    B1_TL(4);
    // ---- block partial of d_W_edge | d_M, every sum in a fixed order: thread (array group g, float4 column c4) adds its 8 of the 32
    //      arrays, the four groups meet in the dead ring behind the d_M scratch, thread c4 adds them in group order and stores 16 bytes ----
    float* out = a.partial + (size_t)blockIdx.x * P;
    {
        constexpr int NC4 = 4 * H * 16, NG = 4, PER = V * 4 / NG;                      // float4 columns of an array; array groups
        static_assert(V * 4 % NG == 0 && NG * NC4 <= kBlockT, "reduction geometry");
        float* s_part = s_ring + V * 64 * 4;
        if (tid < NG * NC4) {
            const int g = tid / NC4, c4 = tid - g * NC4;
            float4 acc = ld4(s_dw + (size_t)(g * PER) * WLD + 4 * c4);
#pragma unroll
            for (int v = 1; v < PER; ++v) {
                const float4 x = ld4(s_dw + (size_t)(g * PER + v) * WLD + 4 * c4);
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
            st4(s_part + 4 * tid, acc);
        }
        __syncthreads();
        if (tid < NC4) {
            const int tt = tid / (H * 16), hq = tid - tt * (H * 16), h = hq >> 4, q = hq & 15;
            if (q < Q) {
                float4 acc = ld4(s_part + 4 * tid);
#pragma unroll
                for (int g = 1; g < NG; ++g) {
                    const float4 x = ld4(s_part + 4 * (g * NC4 + tid));
                    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                }
                st4(out + tt * HC + h * Cp + 4 * q, acc);
            }
        }
    }
    if (tid < 16) {                                           // d_M[type tt][head hh]: the vector waves' totals in wave order
        const int tt = tid >> 2, hh = tid & 3;
        float sum = 0.f;
        if (hh < H) {
#pragma unroll
            for (int v = 0; v < V; ++v) sum += s_ring[(v * 4 + hh) * 4 + tt];
        }
        out[WSZ + tt * 4 + hh] = sum;
    }
#ifdef GLAM_WS_TL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B1_TL(5);
#endif
}

static size_t b1ws_lds_bytes(int H, int Cp, int V) {
    const int HC = H * Cp;
    const int WL = 4 * ws_wedge_pitch_h(H), WLD = 4 * 64 * H;     // W_edge rows | the d_W_edge arrays [4 types][H][64]
    return ((size_t)WL + 64 + (size_t)V * 4 * WLD + (size_t)4 * 16 * (HC + 8)) * sizeof(float);
}

template <int H, int V, bool X3, bool WT = false>
static int launch_b1ws_x(const DstWsArgs& a, int grid, hipStream_t s) {
    if constexpr (X3 && !WT) { if (a.N <= kWtMaxRows) return launch_b1ws_x<H, V, X3, true>(a, grid, s); }
    static bool big[64] = {};
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_triplet_bwd_dst_ws<H, V, X3, WT>), big, "triplet_bwd_dst_ws")) return rc;
    GLAM_PROF_LABEL("d_aggr+k_triplet_bwd_dst_ws");
    hipLaunchKernelGGL((k_triplet_bwd_dst_ws<H, V, X3, WT>), dim3(grid), dim3((V + 4) * 64), b1ws_lds_bytes(H, a.Cp, V), s, a);
    return GLAM_OK;
}
template <int H, int V>
static int launch_b1ws(const DstWsArgs& a, int grid, hipStream_t s) {
    // (with fewer heads the 72-register weight slice would cost the launch its fourth wave per SIMD: those keep the fp32 form)
    if constexpr (H == 3) { if (ts_x3_enabled()) return launch_b1ws_x<H, V, true>(a, grid, s); }
    return launch_b1ws_x<H, V, false>(a, grid, s);
}

// GLAM_B1_PRE=0: the matrix waves split the plain d_aggr image in every block's prologue (A/B switch; bit-identical)
static bool b1_pre_enabled() { const char* e = getenv("GLAM_B1_PRE"); return !e || atoi(e) != 0; }

bool triplet_bwd_dst_ws_supported(int H, int Cp, int De, int edge_onehot) {
    return triplet_fwd_ws_enabled() && triplet_fwd_ws_supported(H, Cp, De, edge_onehot) && H <= 3;
}
int triplet_bwd_dst_ws_blocks(int64_t N) {
    const int ntiles = (int)((N + 15) / 16);
    const int cap = ws_grid_cap(kBwdBlocks);      // the partial workspace holds kBwdBlocks block partials
    return ntiles < cap ? ntiles : cap;
}

// B1 with the d_aggr GEMM inside, warp-specialised (called by triplet_bwd_impl).  Writes gridDim.x block partials [4 * H * Cp + 16].
int triplet_bwd_dst_ws(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M, const float* aggr,
                       const float* stats, const float* d_out, const float* img_dagg, const int32_t* ell_src, const int32_t* ell_eid,
                       int64_t N, int64_t E, int H, int Cp, int De, int edge_onehot, float slope, float* d_aggr, float* alpha_e,
                       float* dpre_e, float* d_a_ij, float* partial, int* nblk_out, hipStream_t s, const float* dagg_pre) {
    if (!triplet_bwd_dst_ws_supported(H, Cp, De, edge_onehot))
        return fail(GLAM_E_UNSUPPORTED, "triplet_bwd_dst_ws: H=%d Cp=%d De=%d onehot=%d outside the kernel table", H, Cp, De, edge_onehot);
    if ((uint64_t)N * H * Cp * 4 >= (1ull << 32) || (uint64_t)E * De * 4 >= (1ull << 32))
        return fail(GLAM_E_UNSUPPORTED, "triplet_bwd_dst_ws: a tensor exceeds 4 GiB (32-bit offsets)");
    DstWsArgs a{xw, a_ij, edge_attr, w_edge, M, aggr, stats, d_out, img_dagg, ell_src, ell_eid, (int)N, Cp, slope,
                d_aggr, alpha_e, dpre_e, d_a_ij, partial, (H == 3 && b1_pre_enabled()) ? dagg_pre : nullptr};
    const int grid = triplet_bwd_dst_ws_blocks(N);
    int rc = GLAM_OK;
    switch (H) {
        case 1: rc = launch_b1ws<1, 8>(a, grid, s); break;
        case 2: rc = launch_b1ws<2, 8>(a, grid, s); break;
        default: rc = launch_b1ws<3, 8>(a, grid, s); break;
    }
    if (rc) return rc;
    GLAM_LAUNCH_CHECK("triplet_bwd_dst_ws");
    *nblk_out = grid;
    return GLAM_OK;
}

}  // namespace glam

#ifdef GLAM_WS_TL
extern "C" int glam_debug_b1_tl(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_b1_tl), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
extern "C" int glam_debug_b1_rt(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_b1_rt), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
