// k_wgrad_x3: the weight-gradient products G[I, J] = [P1 | P2 | 1]^T[I, N] @ Q[N, J] (I <= 320, J <= 64: the reduction runs over the
// N node rows) on the bf16 matrix cores in 3 x bf16 form (bf16x3.h: fp32 accuracy), warp-specialised.  These are the parameter
// gradients of TripletMessage (autograd of src_1gp/layer.py:37, 58-60: d_weight_scale / d_bias = [aggr | 1]^T d_out, d_weight_node +
// the attention rows = [d_xw | d_a]^T x), of the GRU gate linears of MessageBlock (src_1gp/layer.py:262) and of every LinearBlock
// (src_1gp/layer.py:232-237).  k_wgrad (gemm.hip) runs them on v_mfma_f32_16x16x4_f32 — the fp32 VECTOR rate, 34.6 cycles per 4 rows of
// one 16 x 16 tile.  Here every operand row is loaded and split ONCE per block:
//   block   = (group of up to three 64-column slabs of P, row split): 4 producer + 4 consumer waves (one of each per SIMD, <= 256
//             registers), one block per CU; 32-row steps through a ring of three 48 KB stages in LDS
//   producer wave g: column group g (0 .. 2: a slab of P, 3: the <= 64 columns of Q) of EVERY step; lane (c, kb) loads rows 8 kb .. 8 kb + 7
//             of its four columns 4 c .. 4 c + 3 (eight float4: 256-byte runs per row), so the t-th components of its eight registers ARE
//             eight consecutive k of column 4 c + t — the reduction index of the matrix instruction is the row —, splits them into
//             (hi, mid, lo) (5.5 vector instructions per value: bf16x3.h split2s) and writes 16 bytes per term and column: the LDS image
//             is already the fragment layout, no transpose anywhere.  Unit (t, c, kb) sits at t * 1024 + c * 64 + 16 (kb ^ (c >> 1 & 3)):
//             the 8-lane groups of ds_write_b128 and the four 16-lane groups of ds_read_b128 each cover every bank exactly once.
//   consumer wave w: the tile rows 3 w .. 3 w + 2 of the group's twelve (tile row T = 4 slab + ti) x all four tile columns — tile
//             (ti, tj) = rows 4 m + ti of the slab x columns 4 n + tj, the stride-4 permutation of k_wgrad, so the partial slabs keep
//             its format (k_final_reduce, k_param_grads) —: 21 fragment reads and 72 matrix instructions per step in two accumulator
//             chains per tile (small + middle partial products | hi x hi: DESIGN §4.4); every kWxFlush steps both are added into a
//             master accumulator by the vector ALU (round to nearest — the matrix instruction aligns by truncation, and a chain of
//             thousands of steps would carry that bias into the sums).  No cross-wave reduction: the consumers' tiles are disjoint.
// Producers check in per stage (s_ready), consumers release it (s_taken); one barrier (flag initialisation).
// What the forms before this one measured (ablation builds, N = 326 400; profiles/r5_wgrad_x3_forms.txt): one slab per block with 8 + 8
// waves (commit 613b9b6) took 160 us of which 79 were the handshake skeleton and 49 the LDS traffic — Q split and staged three times, 72
// fragment reads per 96 matrix instructions —, 13 the splits and matrix instructions together; sync-free forms with every wave splitting
// its own operands (one 64 x 64 slab per wave at one wave per SIMD: 183 us; 64 x 32 at two per SIMD: 205 us) are bound by the CU's
// 64-byte-per-clock load path (P or Q fetched twice per CU) and by one wave's issue slots.
#include "dense.h"
#include "triplet_pipe.h"

#include <type_traits>

namespace glam {

#ifndef GLAM_WX_NT
#define GLAM_WX_NT 0      // 1: the producers' row loads with the non-temporal policy (every operand row is read once, by one block)
#endif
constexpr int kWxProd = 4, kWxCons = 4, kWxSlabs = 3;
constexpr int kWxThreads = (kWxProd + kWxCons) * 64;
constexpr int kWxUnit = 1024;                  // one column class t of one group: [16 c][4 kb] x 16 bytes
constexpr int kWxGroup = 4 * kWxUnit;          // [4 t]
constexpr int kWxPlane = kWxProd * kWxGroup;   // [P slab 0 | P slab 1 | P slab 2 | Q]
constexpr int kWxStage = 3 * kWxPlane;         // hi | mid | lo: 48 KB per 32-row step
constexpr int kWxRing = 3;
#ifndef GLAM_WX_DEPTH
#define GLAM_WX_DEPTH 3
#endif
constexpr int kWxDepth = GLAM_WX_DEPTH;        // register stages of a producer: steps of loads in flight
constexpr int kWxHeader = 256;                 // s_ready[16] | s_taken[16]
constexpr int kWxFlush = 8;                    // steps between two master-accumulator updates
constexpr size_t kWxLds = kWxHeader + (size_t)kWxRing * kWxStage;
static_assert(kWxLds <= 160 * 1024, "ring exceeds the LDS of a CU");

// lanes whose four columns are the virtual all-ones column / zero padding read here with a row stride of 0
__device__ float4 g_wx_const[2] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

template <bool CELU, bool SEG>
__global__ void __launch_bounds__(kWxThreads) k_wgrad_x3(WgArgs2 two) {
    extern __shared__ __attribute__((aligned(16))) char s_wx[];
    int* s_ready = reinterpret_cast<int*>(s_wx);
    int* s_taken = s_ready + 16;
    char* s_ring = s_wx + kWxHeader;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool second = (int)blockIdx.x >= two.first_b;
    const WgArgs a = second ? two.b : two.a;
    // XCD-aware decode (blockIdx % 8 = XCD): the slab groups of one row split share an XCD, so Q is fetched into one L2
    const int bid = second ? (int)blockIdx.x - two.first_b : (int)blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int ngrp = (a.ntile + kWxSlabs - 1) / kWxSlabs;
    const int grp = loc % ngrp, split = (loc / ngrp) * 8 + xcd;
    if (split >= a.nsplit) return;
    // the block's rows: [row0, row1) of operand set `set` (SEG: the splits are dealt set by set, so a block never straddles two sets)
    int set = 0, lsplit = split, nrows = a.N;
    if (SEG) { const int per = a.nsplit / max(a.nseg, 1); set = split / per; lsplit = split - set * per; nrows = a.seg_rows; }
    const int row0 = min(lsplit * a.rows_per_split, nrows), row1 = min(row0 + a.rows_per_split, nrows);
    const int nsteps = (row1 - row0 + 31) >> 5, nfull = (row1 - row0) >> 5;
    if (tid < 32) s_ready[tid] = 0;
    __syncthreads();

    if (wave < kWxProd) {
        if (nsteps == 0) return;
        const int g = wave, c = lane & 15, kb = lane >> 4;
        // this lane's source column block: base pointer + row stride in floats (I1, I2, J are multiples of 4: no straddling)
        const float* P1 = a.P1; const float* P2 = a.P2; const float* Q = a.Q;
        // (static indices: `a.segP1[set - 1]` made the compiler put the whole argument struct into scratch memory — 168 bytes per lane —
        // to index it)
        if (SEG && set == 1) { P1 = a.segP1[0]; P2 = a.segP2[0]; Q = a.segQ[0]; }
        if (SEG && set >= 2) { P1 = a.segP1[1]; P2 = a.segP2[1]; Q = a.segQ[1]; }
        const float* src;
        int ld = 0;
        if (g < kWxSlabs) {
            const int pcol = (grp * kWxSlabs + g) * 64 + 4 * c, I12 = a.I1 + a.I2;
            src = reinterpret_cast<const float*>(&g_wx_const[(a.ones && pcol == I12) ? 0 : 1]);
            if (pcol < a.I1) { src = P1 + pcol; ld = a.ldp1; }
            else if (pcol < I12) { src = P2 + (pcol - a.I1); ld = a.ldp2; }
        } else {
            const int qcol = 4 * c;
            src = reinterpret_cast<const float*>(&g_wx_const[(a.qones && qcol == a.J) ? 0 : 1]);
            if (qcol < a.J) { src = Q + qcol; ld = a.ldq; }
        }
        const bool celu = CELU && a.q_celu && g == kWxSlabs;       // wave-uniform
        // every load is unconditional (a row beyond the block re-reads its last row and is zeroed when it is used)
        auto load = [&](int s, float4 (&v)[8]) {
            const int rb = row0 + 32 * s + 8 * kb;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = GLAM_WX_NT ? ld4nt(src + (size_t)min(rb + i, row1 - 1) * ld) : ld4(src + (size_t)min(rb + i, row1 - 1) * ld);
#ifdef GLAM_WX_NOLOAD       // timing experiment only (wrong numbers): every step re-reads the block's first rows (cache hits)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ld4(src + (size_t)min(row0 + 8 * kb + i, row1 - 1) * ld);
#endif
        };
        char* const mine = s_ring + g * kWxGroup + c * 64 + 16 * (kb ^ ((c >> 1) & 3));
        // RAGGED: the step may be the block's last one, ending inside its 32 rows (rows beyond row1 are zeroed)
        auto emit = [&](int s, float4 (&v)[8], auto ragged) {
            const int slot = s % kWxRing, round = s / kWxRing;
            if constexpr (decltype(ragged)::value) {
                const int rb = row0 + 32 * s + 8 * kb;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (rb + i >= row1) v[i] = f4zero();
            }
            while (flag_load(s_taken + slot) < kWxCons * round) { }
            asm volatile("" ::: "memory");
            char* tl = mine + slot * kWxStage;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint4 h, m, l;
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = f4get(v[i], t);
                if (celu) {
                    // one wave applies the CELU to all of Q: with expf (~18 instructions) it would set the block's pace (measured:
                    // 55 us against k_wgrad's 42 for the GRU's three operand sets); one multiply + v_exp_f32 (common.h softmax_exp:
                    // |x| 2^-24 + 1 ulp on exp(x) <= 1, i.e. the same absolute error as expf(x) - 1 rounds with)
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = x[i] > 0.f ? x[i] : softmax_exp(x[i]) - 1.f;
                }
#ifdef GLAM_WX_NOSPLIT      // timing experiment only (wrong numbers): the producers without their vector work
                h = make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
                m = make_uint4(__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7]));
                l = h;
#else
                split2s(x[0], x[1], h.x, m.x, l.x);
                split2s(x[2], x[3], h.y, m.y, l.y);
                split2s(x[4], x[5], h.z, m.z, l.z);
                split2s(x[6], x[7], h.w, m.w, l.w);
#endif
                *reinterpret_cast<uint4*>(tl + t * kWxUnit) = h;
                *reinterpret_cast<uint4*>(tl + t * kWxUnit + kWxPlane) = m;
                *reinterpret_cast<uint4*>(tl + t * kWxUnit + 2 * kWxPlane) = l;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_ready + slot);
        };
        // kWxDepth steps of loads in flight per wave (registers; the LDS ring behind them has three stages): 32 KB per wave and step of
        // depth.  The steady loop has no condition between a step's loads and their use, so the wait in front of a split is for exactly
        // that register set; the last steps of the block — the ragged one among them — run behind it
        float4 buf[kWxDepth][8];
#pragma unroll
        for (int d = 0; d < kWxDepth; ++d) load(d, buf[d]);
        int s = 0;
        for (; s + kWxDepth - 1 < nfull; s += kWxDepth) {
#pragma unroll
            for (int d = 0; d < kWxDepth; ++d) {
                emit(s + d, buf[d], std::false_type{});
                load(s + d + kWxDepth, buf[d]);            // (beyond the block: clamped re-reads, never used)
            }
        }
#pragma unroll
        for (int d = 0; d < kWxDepth; ++d)
            if (s + d < nsteps) emit(s + d, buf[d], std::true_type{});
        return;
    }

    // ---- consumers ----
    const int w = wave - kWxProd, m = lane & 15, kq = lane >> 4;
    const int off = m * 64 + 16 * (kq ^ ((m >> 1) & 3));
    v4f_t master[3][4], ca[3][4], cb[3][4];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) master[u][tj] = ca[u][tj] = cb[u][tj] = (v4f_t){0.f, 0.f, 0.f, 0.f};
    // unit offsets of this wave's three tile rows T = 3 w + u: group (slab) T / 4, column class T % 4
    int poff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) poff[u] = ((3 * w + u) >> 2) * kWxGroup + ((3 * w + u) & 3) * kWxUnit;
    auto rd3 = [&](const char* p) {
        Bf16x3 f;
        f.hi = *reinterpret_cast<const bf16x8_t*>(p);
        f.mid = *reinterpret_cast<const bf16x8_t*>(p + kWxPlane);
        f.lo = *reinterpret_cast<const bf16x8_t*>(p + 2 * kWxPlane);
        return f;
    };
    for (int s = 0; s < nsteps; ++s) {
        const int slot = s % kWxRing, want = kWxProd * (s / kWxRing + 1);
        while (flag_load(s_ready + slot) < want) { }
        asm volatile("" ::: "memory");
        const char* tl = s_ring + slot * kWxStage + off;
        Bf16x3 qb[4], pa[3];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) qb[tj] = rd3(tl + kWxSlabs * kWxGroup + tj * kWxUnit);
#pragma unroll
        for (int u = 0; u < 3; ++u) pa[u] = rd3(tl + poff[u]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flag_bump(s_taken + slot);          // every fragment is in registers: the stage may be refilled
#ifdef GLAM_WX_NOMFMA       // timing experiment only (wrong numbers): the consumers without their matrix instructions
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                ca[u][tj] += __builtin_bit_cast(v4f_t, pa[u].hi) + __builtin_bit_cast(v4f_t, qb[tj].lo) + __builtin_bit_cast(v4f_t, pa[u].mid);
                cb[u][tj] += __builtin_bit_cast(v4f_t, pa[u].lo) + __builtin_bit_cast(v4f_t, qb[tj].hi) + __builtin_bit_cast(v4f_t, qb[tj].mid);
            }
        if (false)
#endif
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) ca[u][tj] = mfma_x3_small(pa[u], qb[tj], ca[u][tj]);
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) ca[u][tj] = mfma_x3_mid(pa[u], qb[tj], ca[u][tj]);
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) cb[u][tj] = mfma_x3_big(pa[u], qb[tj], cb[u][tj]);
        }
        if ((s + 1) % kWxFlush == 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    master[u][tj] += ca[u][tj] + cb[u][tj];
                    ca[u][tj] = cb[u][tj] = (v4f_t){0.f, 0.f, 0.f, 0.f};
                }
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int T = 3 * w + u, slab = grp * kWxSlabs + (T >> 2), ti = T & 3;
        if (slab >= a.ntile) continue;                     // (a group's slabs beyond the product: zero columns, nothing to store)
        float* out = a.partial + ((size_t)__builtin_amdgcn_readfirstlane(slab) * a.nsplit + split) * kWgSlabStride;      // (wave-uniform)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const v4f_t v = master[u][tj] + (ca[u][tj] + cb[u][tj]);
            st4o_sel(a.N <= kWtMaxRows, out, (unsigned)(((ti * 4 + tj) * 64 + lane) * 16), make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}

template <bool CELU, bool SEG>
static int launch_wx(const WgArgs2& two, int blocks, hipStream_t s) {
    static bool big[64] = {};
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_wgrad_x3<CELU, SEG>), big, "wgrad_x3")) return rc;
    hipLaunchKernelGGL((k_wgrad_x3<CELU, SEG>), dim3(blocks), dim3(kWxThreads), kWxLds, s, two);
    return GLAM_OK;
}

// the geometry is plan_wgrad_x3's (gemm.hip): rows_per_split a multiple of 32, nsplit a multiple of nseg, blocks per (slab group, split)
int launch_wgrad_x3(const WgArgs2& two, int blocks, hipStream_t s) {
    const bool seg = two.a.nseg > 1, celu = two.a.q_celu || two.b.q_celu;
    int rc;
    if (seg && !celu) { GLAM_PROF_LABEL("k_wgrad_x3<false, sets>"); rc = launch_wx<false, true>(two, blocks, s); }
    else if (seg) { GLAM_PROF_LABEL("k_wgrad_x3<true, sets>"); rc = launch_wx<true, true>(two, blocks, s); }
    else if (celu) { GLAM_PROF_LABEL("k_wgrad_x3<true>"); rc = launch_wx<true, false>(two, blocks, s); }
    else { GLAM_PROF_LABEL("k_wgrad_x3<false>"); rc = launch_wx<false, false>(two, blocks, s); }
    if (rc) return rc;
    GLAM_LAUNCH_CHECK("wgrad_x3");
    return GLAM_OK;
}

}  // namespace glam
