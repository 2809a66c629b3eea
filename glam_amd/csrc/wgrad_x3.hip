// k_wgrad_x3: the weight-gradient products G[I, J] = [P1 | P2 | 1]^T[I, N] @ Q[N, J] (I <= 320, J <= 64: the reduction runs over the
// N node rows) on the bf16 matrix cores in 3 x bf16 form (bf16x3.h: fp32 accuracy), warp-specialised.  These are the parameter
// gradients of TripletMessage (autograd of src_1gp/layer.py:37, 58-60: d_weight_scale / d_bias = [aggr | 1]^T d_out, d_weight_node +
// the attention rows = [d_xw | d_a]^T x), of the GRU gate linears of MessageBlock (src_1gp/layer.py:262) and of every LinearBlock
// (src_1gp/layer.py:232-237).  k_wgrad (gemm.hip) runs them on v_mfma_f32_16x16x4_f32 — the fp32 VECTOR rate, 34.6 cycles per 4 rows and
// 64 x 64 outputs on a datapath it shares with every other wave of the SIMD; two earlier 3 x bf16 forms of it lost because every wave
// split its own operands.  Here a row block is split ONCE:
//   block   = (64-column slab of P, row split) as in k_wgrad, 8 producer + 8 consumer waves, one block per CU
//   producer wave (g, q): column group g (0: the slab's 64 columns of P, 1: the <= 64 columns of Q), every fourth 32-row step; lane
//           (c, kb) loads rows 8 kb .. 8 kb + 7 of its four columns 4 c .. 4 c + 3 (eight float4: 256-byte runs per row), so the t-th
//           components of its eight registers ARE eight consecutive k of column 4 c + t — the reduction index of the matrix
//           instruction is the row —, splits them into (hi, mid, lo) and writes 16 bytes per term and column: the LDS image is already
//           the fragment layout, no transpose anywhere.  Unit (t, c, kb) sits at t * 1024 + c * 64 + 16 (kb ^ (c >> 1 & 3)): the 8-lane
//           groups of ds_write_b128 and the four 16-lane groups of ds_read_b128 each cover every bank exactly once.
//   consumer wave (ti pair, tj): output tiles (2 ti', tj), (2 ti' + 1, tj) of the slab's 4 x 4 — tile (ti, tj) = rows 4 m + ti of the
//           slab x columns 4 n + tj, the stride-4 permutation of k_wgrad, so the partial slabs keep its format (k_final_reduce,
//           k_param_grads) —: 9 fragment reads and 12 matrix instructions per step in three accumulator chains per tile (small /
//           middle / large partial products); every kFlush steps the chains are added into a master accumulator by
//           the vector ALU (round to nearest: the matrix instruction aligns by truncation, and a chain of thousands of steps would
//           carry that bias into the sums).  No cross-wave reduction: the tiles of the consumers are disjoint.
// Ring of RING 24 KB stages, producers check in per stage (s_ready), consumers release it (s_taken); one barrier (flag initialisation).
#include "dense.h"
#include "triplet_pipe.h"

#include <type_traits>

namespace glam {

constexpr int kWxProd = 8, kWxCons = 8;
constexpr int kWxThreads = (kWxProd + kWxCons) * 64;
constexpr int kWxUnit = 1024;                  // one column class t of one group: [16 c][4 kb] x 16 bytes
constexpr int kWxGroup = 4 * kWxUnit;          // [4 t]
constexpr int kWxPlane = 2 * kWxGroup;         // [P columns | Q columns]
constexpr int kWxStage = 3 * kWxPlane;         // hi | mid | lo: 24 KB per 32-row step
constexpr int kWxHeader = 256;                 // s_ready[16] | s_taken[16]
constexpr int kWxFlush = 8;                    // steps between two master-accumulator updates

// lanes whose four columns are the virtual all-ones column / zero padding read here with a row stride of 0
__device__ float4 g_wx_const[2] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

#ifdef GLAM_WX_NOSLEEP
#define WX_PAUSE() do { } while (0)
#else
#define WX_PAUSE() __builtin_amdgcn_s_sleep(1)
#endif

#ifdef GLAM_WX_PROF    // developer aid (tools/wx_prof.py): s_memtime stamps of lane 0 of (block, wave), 8 per wave
__device__ long long g_wx_prof[512 * 16 * 8];
#define WX_STAMP(k) do { if (lane == 0 && blockIdx.x < 512) g_wx_prof[(blockIdx.x * 16 + wave) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#define WX_CLK() ((long long)__builtin_readcyclecounter())
#define WX_ACC(k, v) do { if (lane == 0 && blockIdx.x < 512) g_wx_prof[(blockIdx.x * 16 + wave) * 8 + (k)] = (v); } while (0)
#else
#define WX_STAMP(k) do { } while (0)
#endif

template <bool CELU, bool SEG, int RING>
__global__ void __launch_bounds__(kWxThreads) k_wgrad_x3(WgArgs2 two) {
    extern __shared__ __attribute__((aligned(16))) char s_wx[];
    int* s_ready = reinterpret_cast<int*>(s_wx);
    int* s_taken = s_ready + 16;
    char* s_ring = s_wx + kWxHeader;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    WX_STAMP(0);
    const bool second = (int)blockIdx.x >= two.first_b;
    const WgArgs a = second ? two.b : two.a;
    // XCD-aware decode (blockIdx % 8 = XCD): the slabs of one row split share an XCD, so Q is fetched into one L2
    const int bid = second ? (int)blockIdx.x - two.first_b : (int)blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int slab = loc % a.ntile, split = (loc / a.ntile) * 8 + xcd;
    if (split >= a.nsplit) return;
    // the block's rows: [row0, row1) of operand set `set` (SEG: the splits are dealt set by set, so a block never straddles two sets)
    int set = 0, lsplit = split, nrows = a.N;
    if (SEG) { const int per = a.nsplit / max(a.nseg, 1); set = split / per; lsplit = split - set * per; nrows = a.seg_rows; }
    const int row0 = min(lsplit * a.rows_per_split, nrows), row1 = min(row0 + a.rows_per_split, nrows);
    const int nsteps = (row1 - row0 + 31) >> 5;
    WX_STAMP(5);
    if (tid < 32) s_ready[tid] = 0;
    __syncthreads();
    WX_STAMP(6);

    if (wave < kWxProd) {
        if (nsteps == 0) return;
        const int g = wave & 1, q = wave >> 1, c = lane & 15, kb = lane >> 4;
        // this lane's source column block: base pointer + row stride in floats (I1, I2, J are multiples of 4: no straddling)
        const float* P1 = a.P1; const float* P2 = a.P2; const float* Q = a.Q;
        if (SEG && set > 0) { P1 = a.segP1[set - 1]; P2 = a.segP2[set - 1]; Q = a.segQ[set - 1]; }
        const float* src;
        int ld = 0;
        if (g == 0) {
            const int pcol = slab * 64 + 4 * c, I12 = a.I1 + a.I2;
            src = reinterpret_cast<const float*>(&g_wx_const[(a.ones && pcol == I12) ? 0 : 1]);
            if (pcol < a.I1) { src = P1 + pcol; ld = a.ldp1; }
            else if (pcol < I12) { src = P2 + (pcol - a.I1); ld = a.ldp2; }
        } else {
            const int qcol = 4 * c;
            src = reinterpret_cast<const float*>(&g_wx_const[(a.qones && qcol == a.J) ? 0 : 1]);
            if (qcol < a.J) { src = Q + qcol; ld = a.ldq; }
        }
        const bool celu = CELU && a.q_celu && g == 1;       // wave-uniform
        // every load is unconditional (a row beyond the block re-reads its last row and is zeroed when it is used)
        auto load = [&](int s, float4 (&v)[8]) {
            const int rb = row0 + 32 * s + 8 * kb;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ld4(src + (size_t)min(rb + i, row1 - 1) * ld);
#ifdef GLAM_WX_NOLOAD       // timing experiment only (wrong numbers): every step re-reads the block's first rows (cache hits)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ld4(src + (size_t)min(row0 + 8 * kb + i, row1 - 1) * ld);
#endif
        };
        char* const mine = s_ring + g * kWxGroup + c * 64 + 16 * (kb ^ ((c >> 1) & 3));
#ifdef GLAM_WX_PROF
        long long t_data = 0, t_slot = 0, t_pub = 0;
#endif
        // MASK: the step may be the ragged last one (rows beyond row1 are zeroed)
        auto emit = [&](int s, float4 (&v)[8], auto mask) {
            const int slot = s % RING, round = s / RING;
            if (celu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = celu4(v[i]);
            }
            if constexpr (decltype(mask)::value) {
                const int rb = row0 + 32 * s + 8 * kb;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (rb + i >= row1) v[i] = f4zero();
            }
#ifdef GLAM_WX_PROF
            const long long c0 = WX_CLK();
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            const long long c1 = WX_CLK();
            t_data += c1 - c0;
#endif
#ifndef GLAM_WX_NOPOLLP
            while (flag_load(s_taken + slot) < kWxCons * round) WX_PAUSE();
#endif
            asm volatile("" ::: "memory");
#ifdef GLAM_WX_PROF
            const long long c2 = WX_CLK();
            t_slot += c2 - c1;
#endif
            char* tl = mine + slot * kWxStage;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint4 h, m, l;
#ifdef GLAM_WX_NOSPLIT      // timing experiment only (wrong numbers): the producers without their vector work
                h = make_uint4(__float_as_uint(f4get(v[0], t)), __float_as_uint(f4get(v[1], t)), __float_as_uint(f4get(v[2], t)), __float_as_uint(f4get(v[3], t)));
                m = make_uint4(__float_as_uint(f4get(v[4], t)), __float_as_uint(f4get(v[5], t)), __float_as_uint(f4get(v[6], t)), __float_as_uint(f4get(v[7], t)));
                l = h;
#else
                split2s(f4get(v[0], t), f4get(v[1], t), h.x, m.x, l.x);
                split2s(f4get(v[2], t), f4get(v[3], t), h.y, m.y, l.y);
                split2s(f4get(v[4], t), f4get(v[5], t), h.z, m.z, l.z);
                split2s(f4get(v[6], t), f4get(v[7], t), h.w, m.w, l.w);
#endif
#ifdef GLAM_WX_NOWRITE      // timing experiment only (wrong numbers): one of the twelve LDS writes per step
                if (t == 0) *reinterpret_cast<uint4*>(tl) = make_uint4(h.x ^ m.x ^ l.x, h.y ^ m.y ^ l.y, h.z ^ m.z ^ l.z, h.w ^ m.w ^ l.w);
                else asm volatile("" :: "v"(h.x ^ m.x ^ l.x), "v"(h.y ^ m.y ^ l.y), "v"(h.z ^ m.z ^ l.z), "v"(h.w ^ m.w ^ l.w));
#else
                *reinterpret_cast<uint4*>(tl + t * kWxUnit) = h;
                *reinterpret_cast<uint4*>(tl + t * kWxUnit + kWxPlane) = m;
                *reinterpret_cast<uint4*>(tl + t * kWxUnit + 2 * kWxPlane) = l;
#endif
            }
        };
        auto publish = [&](int s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) flag_bump(s_ready + s % RING);
#ifdef GLAM_WX_PROF
            t_pub = WX_CLK();
#endif
        };
        // two steps of loads in flight per wave (registers): 16 KB per wave, 128 KB per CU.  The steady loop has no condition between a
        // step's loads and their use, so the wait in front of a split is for exactly that register set (vmcnt(8): the other set stays in
        // flight); the last one or two steps of the wave — the ragged one among them — run behind it
        const int nfull = (row1 - row0) >> 5;
        float4 buf[2][8];
        load(q, buf[0]);
        load(q + 4, buf[1]);
        WX_STAMP(1);
        int s = q;
        for (; s + 4 < nfull; s += 8) {
            emit(s, buf[0], std::false_type{});
            if (s == q) WX_STAMP(3);                       // the first step's rows have arrived and are split
            publish(s);                                    // (before the next loads: their issue can stall on a full memory queue)
            load(s + 8, buf[0]);                           // (beyond the block: clamped re-reads, never used)
            emit(s + 4, buf[1], std::false_type{});
            publish(s + 4);
            load(s + 12, buf[1]);
        }
        if (s < nsteps) { emit(s, buf[0], std::true_type{}); publish(s); }
        if (s + 4 < nsteps) { emit(s + 4, buf[1], std::true_type{}); publish(s + 4); }
#ifdef GLAM_WX_PROF
        WX_ACC(4, t_data); WX_ACC(7, t_slot); (void)t_pub;
#endif
        WX_STAMP(2);
        return;
    }

    // ---- consumers ----
    const int w = wave - kWxProd, tp = w >> 2, tj = w & 3, m = lane & 15, kq = lane >> 4;
    const int off = m * 64 + 16 * (kq ^ ((m >> 1) & 3));
    v4f_t master[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    v4f_t cs[2], cm[2], cb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) cs[u] = cm[u] = cb[u] = (v4f_t){0.f, 0.f, 0.f, 0.f};
    WX_STAMP(1);
#ifdef GLAM_WX_PROF
    long long t_ready = 0, t_lds = 0;
#endif
    struct Frag { Bf16x3 pa[2], qb; };
    // the fragment reads of step s are ISSUED one step ahead — behind the wait for the previous step's reads, in front of its matrix
    // instructions — so a wave's LDS latency runs under its own 12 matrix instructions
    auto fetch = [&](int s, Frag& f) {
        const int slot = s % RING, want = 2 * (s / RING + 1);
#ifdef GLAM_WX_PROF
        const long long c0 = WX_CLK();
#endif
#ifndef GLAM_WX_NOPOLLC
        while (flag_load(s_ready + slot) < want) WX_PAUSE();
#endif
        asm volatile("" ::: "memory");
        if (s == 0) WX_STAMP(4);                           // the first stage is there
#ifdef GLAM_WX_PROF
        if (s > 0) t_ready += WX_CLK() - c0;
#endif
        const char* tl = s_ring + slot * kWxStage + off;
#ifdef GLAM_WX_NOREAD       // timing experiment only (wrong numbers): one of the nine fragment reads per step
        f.pa[0].hi = *reinterpret_cast<const bf16x8_t*>(tl);
        f.pa[0].mid = f.pa[0].lo = f.pa[1].hi = f.pa[1].mid = f.pa[1].lo = f.qb.hi = f.qb.mid = f.qb.lo = f.pa[0].hi;
        return;
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const char* p = tl + (2 * tp + u) * kWxUnit;
            f.pa[u].hi = *reinterpret_cast<const bf16x8_t*>(p);
            f.pa[u].mid = *reinterpret_cast<const bf16x8_t*>(p + kWxPlane);
            f.pa[u].lo = *reinterpret_cast<const bf16x8_t*>(p + 2 * kWxPlane);
        }
        const char* p = tl + kWxGroup + tj * kWxUnit;
        f.qb.hi = *reinterpret_cast<const bf16x8_t*>(p);
        f.qb.mid = *reinterpret_cast<const bf16x8_t*>(p + kWxPlane);
        f.qb.lo = *reinterpret_cast<const bf16x8_t*>(p + 2 * kWxPlane);
    };
    auto step = [&](int s, Frag& cur, Frag& nxt) {
#ifdef GLAM_WX_PROF
        const long long c1 = WX_CLK();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // cur has landed
#ifdef GLAM_WX_PROF
        t_lds += WX_CLK() - c1;
#endif
        if (lane == 0) flag_bump(s_taken + s % RING);          // every fragment of the stage is in registers: it may be refilled
        if (s + 1 < nsteps) fetch(s + 1, nxt);
#ifdef GLAM_WX_NOMFMA       // timing experiment only (wrong numbers): the consumers without their matrix instructions
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            cs[u] += __builtin_bit_cast(v4f_t, cur.pa[u].hi) + __builtin_bit_cast(v4f_t, cur.qb.lo);
            cm[u] += __builtin_bit_cast(v4f_t, cur.pa[u].mid) + __builtin_bit_cast(v4f_t, cur.qb.hi);
            cb[u] += __builtin_bit_cast(v4f_t, cur.pa[u].lo) + __builtin_bit_cast(v4f_t, cur.qb.mid);
        }
#else
#pragma unroll
        for (int u = 0; u < 2; ++u) cs[u] = mfma_x3_small(cur.pa[u], cur.qb, cs[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) cm[u] = mfma_x3_mid(cur.pa[u], cur.qb, cm[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) cb[u] = mfma_x3_big(cur.pa[u], cur.qb, cb[u]);
#endif
        if ((s + 1) % kWxFlush == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                master[u] += (cs[u] + cm[u]) + cb[u];
                cs[u] = cm[u] = cb[u] = (v4f_t){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    Frag fa, fb;
    if (nsteps > 0) fetch(0, fa);
    for (int s = 0; s < nsteps; s += 2) {
        step(s, fa, fb);
        if (s + 1 < nsteps) step(s + 1, fb, fa);
    }
    WX_STAMP(2);
#ifdef GLAM_WX_PROF
    WX_ACC(1, t_ready); WX_ACC(7, t_lds);
#endif
    float* out = a.partial + ((size_t)slab * a.nsplit + split) * kWgSlabStride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const v4f_t v = master[u] + ((cs[u] + cm[u]) + cb[u]);
        st4(out + (((2 * tp + u) * 4 + tj) * 64 + lane) * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
    WX_STAMP(3);
}

#ifndef GLAM_WX_RING
#define GLAM_WX_RING 4
#endif

template <bool CELU, bool SEG>
static int launch_wx(const WgArgs2& two, int blocks, hipStream_t s) {
    constexpr int RING = GLAM_WX_RING;
    static bool big[64] = {};
    constexpr size_t lds = kWxHeader + (size_t)RING * kWxStage;
    static_assert(lds <= 160 * 1024, "ring exceeds the LDS of a CU");
    if (int rc = ws_opt_in_lds(reinterpret_cast<const void*>(&k_wgrad_x3<CELU, SEG, RING>), big, "wgrad_x3")) return rc;
    hipLaunchKernelGGL((k_wgrad_x3<CELU, SEG, RING>), dim3(blocks), dim3(kWxThreads), lds, s, two);
    return GLAM_OK;
}

// the geometry is plan_wgrad_x3's (gemm.hip): rows_per_split a multiple of 32, nsplit a multiple of nseg
int launch_wgrad_x3(const WgArgs2& two, int blocks, hipStream_t s) {
    const bool seg = two.a.nseg > 1, celu = two.a.q_celu || two.b.q_celu;
    int rc;
    if (seg) { GLAM_PROF_LABEL("k_wgrad_x3<true, sets>"); rc = launch_wx<true, true>(two, blocks, s); }
    else if (celu) { GLAM_PROF_LABEL("k_wgrad_x3<true>"); rc = launch_wx<true, false>(two, blocks, s); }
    else { GLAM_PROF_LABEL("k_wgrad_x3<false>"); rc = launch_wx<false, false>(two, blocks, s); }
    if (rc) return rc;
    GLAM_LAUNCH_CHECK("wgrad_x3");
    return GLAM_OK;
}

}  // namespace glam

#ifdef GLAM_WX_PROF
extern "C" int glam_debug_wx_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_wx_prof), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
