// Square-ish fp32 matrix products on the bf16 matrix cores in 3 x bf16 form (bf16x3.h): the readout MLP of the model
// (`mol_flat` 5 * hid_dim -> e_dim = 1024; src_1gp/model.py:43-45, 60-61; layer.py LinearBlock) — the products the tall-skinny kernels
// of gemm.hip / tall_x3.hip do not cover ([B, 300] x [300, 1024] has B = 1024 rows, not 20 000) and that ran on the GEMM library.
//
//   C[R, Cn] = act( A . B + bias ),   A(r, k) = A[r * a_rs + k * a_ks] (* gate),   B(k, c) = B[k * b_ks + c * b_cs]
//
// with ONE stride of each pair equal to 1: a block stages a (TR x 32) tile of A and a (64 x 32) tile of B into LDS as three bf16 planes
// each, k-contiguous rows of 80 bytes (20 words: the 16 lanes of a quarter wave read their 16-byte fragments from 16 different
// 4-bank groups), whatever the layout in memory — an operand whose rows run along k is written as 8-byte pieces, one whose rows run
// ACROSS k (the transposed operands of the backward products: dy^T, x) is transposed on the way in: a thread takes the same four
// columns of two consecutive k, packs (k, k + 1) pairs and writes one word per column and plane, lanes running along k so that a
// wave's 64 words fall into 64 different banks.  So the three products of a linear layer — y = x W^T, dx = dy W, dW = dy^T x — are
// one kernel, no transposed copy of anything is made, and the backward pair shares ONE launch (blocks of both products side by side).
//
// Fused into the staging / epilogue, each a launch of its own before: the bias and ReLU / LeakyReLU of the forward (epilogue), the
// activation's derivative on dy (a gate read beside A: the saved OUTPUT y, y > 0 <=> pre-activation > 0), the bias gradient (a
// virtual all-ones column of B at index Cn whose result column goes to `rowsum`).
//
// A block is two groups of four waves that take alternate chunks half a step apart (one splits and writes while the other
// multiplies).  Measured (clock stamps of a -DGLAM_DENSE_STAMP build, B = 1024, 300 -> 1024): a step of one group is ~1000 cycles of
// split + LDS writes and ~950 of fragment reads + 24 matrix instructions (504 cycles by themselves) — the LDS is the bound: three planes
// per operand are 74 KB of LDS traffic per chunk and tile.  12.8 us forward (library: 14.1 + 5.1 for the ReLU inside a model step),
// 26.6 us for the backward pair (library: 13.4 + 11.8 + mask 5.1 + bias gradient 5.7).
#include "bf16x3.h"
#include "dense.h"
#include <stdlib.h>
#include <type_traits>

namespace glam {
#ifdef GLAM_DENSE_STAMP
__device__ long long g_dense_prof[8 * 2 * 64];
#endif
constexpr int kGK = 32;                 // k per chunk
#ifndef GLAM_DENSE_DEPTH
#define GLAM_DENSE_DEPTH 2
#endif
constexpr int kDepth = GLAM_DENSE_DEPTH; // chunks in flight per block
constexpr int kGPitch = 80;             // bytes per LDS row of one plane (32 bf16 + 16 B)
constexpr int kGPlane = 64 * kGPitch;   // one plane of one operand (rows beyond TR unused)
constexpr int kGLds = 2 * 2 * 3 * kGPlane;     // [buffer][operand][plane]

struct GemmJob {
    const float* A; long long a_rs, a_ks;
    const float* gate; float gate_slope;       // same strides as A; A(r, k) *= gate(r, k) > 0 ? 1 : gate_slope
    const float* B; long long b_ks, b_cs;
    const float* bias;                          // [Cn], may be null
    int act; float act_slope;                   // 0 none, 1 ReLU, 2 LeakyReLU(act_slope)
    float* C; long long ldc;
    float* rowsum;                              // non-null: virtual B column Cn == 1, its result goes to rowsum[r]
    int R, Cn, K;
    int tiles_r, tiles_c, first_tile;
    int vec, c_vec;                             // the 16-byte load path / 16-byte stores of C (aligned, ldc % 4 == 0)
    // k split across blocks (few tiles, long reductions: dx of a batch of 32 is 5 tiles of 32 chunks): unit = (tile, split); split sp takes
    // chunks [sp * cps, (sp + 1) * cps) and writes its (TR x 64) partial tile to part[(tile * splits + sp)], its piece of the all-ones
    // column to part_rs[...]; k_dense_splitk_reduce adds the partials in split order and applies bias / activation.  first_tile counts UNITS.
    int splits, cps;
    float* part; float* part_rs;
};
struct GemmArgs { GemmJob job[2]; int njobs; int ntiles; int per_xcd; };
constexpr int kSplitMaxUnits = 512;                                // workspace: part_rs [units][64] | part [units][64 x 64]
constexpr size_t kSplitWsBytes = (size_t)kSplitMaxUnits * 64 * 4 + (size_t)kSplitMaxUnits * 64 * 64 * 4;

#ifdef GLAM_DENSE_STAMP   // developer aid: clock stamps of thread 0 of each wave group of the first 8 blocks
#define DSTAMP(k) do { if ((threadIdx.x & 255) == 0 && blockIdx.x < 8 && (k) < 64) g_dense_prof[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 64 + (k)] = (long long)clock64(); } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif

struct Quad { float4 v; int n; };       // four memory-consecutive elements and how many of them exist

// Four consecutive floats of line `line` (a row of the matrix in memory) starting at `pos`.  The loads are unconditional (clamped
// addresses, zeroed at use: a load under a condition is waited for on the spot).
template <bool VEC>
__device__ __forceinline__ Quad ld_quad(const float* __restrict__ p, long long ls, int line, int pos, int nlines, int npos) {
    Quad q;
    const int lc = min(line, nlines - 1);
    const float* row = p + (long long)lc * ls;
    if (VEC) {
        q.v = ld4g(row + max(min(pos, npos - 4), 0));
        q.n = (line < nlines && pos < npos) ? 4 : 0;
    } else {
        const int last = npos - 1;
        q.v = make_float4(ld1g(row + min(pos, last)), ld1g(row + min(pos + 1, last)), ld1g(row + min(pos + 2, last)), ld1g(row + min(pos + 3, last)));
        q.n = line < nlines ? max(min(npos - pos, 4), 0) : 0;
    }
    return q;
}
// (the asm pins the USE of the loaded registers to this point of the program: without it the compiler moves the zeroing selects up to
// the loads — of all stages, at the loop's head — and waits for every load in flight there)
__device__ __forceinline__ float4 pinned(float4 v) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return v;
}
__device__ __forceinline__ float4 quad_value(const Quad& q0) {
    Quad q = q0;
    q.v = pinned(q.v);
    return make_float4(q.n > 0 ? q.v.x : 0.f, q.n > 1 ? q.v.y : 0.f, q.n > 2 ? q.v.z : 0.f, q.n > 3 ? q.v.w : 0.f);
}
__device__ __forceinline__ float4 gated(float4 v, float4 g, float slope) {
    return make_float4(g.x > 0.f ? v.x : v.x * slope, g.y > 0.f ? v.y : v.y * slope, g.z > 0.f ? v.z : v.z * slope,
                       g.w > 0.f ? v.w : v.w * slope);
}

// What a thread holds of one operand's next chunk between the loads and the LDS writes
template <int ROWS>
struct Stage {
    static constexpr int kItemsKC = ROWS * 8 / 256;      // (row, k quad) items per thread when rows run along k
    Quad q[2], g[2];
};

// ---- loads ---------------------------------------------------------------------------------------------------------------
// rows along k (ks == 1): item = (row, quad of k); ROWS * 8 items, thread t takes items t, t + 256
// rows across k (rs == 1): item = (k pair, quad of rows); 16 * ROWS / 4 items, thread t takes (kp = t % 16, rq = t / 16): two quads
// (every thread loads, also where a 32-row tile has items for half of them only: a load under a condition is waited for on the spot,
// and so is one behind a run-time branch on the layout — KC and VEC are template parameters, the kernel branches once, around the tile)
template <int ROWS, bool GATE, bool KC, bool VEC>
__device__ __forceinline__ void stage_load(Stage<ROWS>& s, const float* __restrict__ p, const float* __restrict__ gate, long long ls,
                                           int row0, int k0, int nrows, int K, int t) {
    if (KC) {
#pragma unroll
        for (int j = 0; j < (ROWS == 64 ? 2 : 1); ++j) {
            const int item = t + 256 * j, row = row0 + (item >> 3), k = k0 + (item & 7) * 4;
            s.q[j] = ld_quad<VEC>(p, ls, row, k, nrows, K);
            if (GATE) s.g[j] = ld_quad<VEC>(gate, ls, row, k, nrows, K);
        }
    } else {
        const int kp = t & 15, rq = (t >> 4) & (ROWS / 4 - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s.q[j] = ld_quad<VEC>(p, ls, k0 + 2 * kp + j, row0 + 4 * rq, K, nrows);
            if (GATE) s.g[j] = ld_quad<VEC>(gate, ls, k0 + 2 * kp + j, row0 + 4 * rq, K, nrows);
        }
    }
}

// ---- split + LDS writes ----------------------------------------------------------------------------------------------------
// ones_at >= 0 (rows across k only): the row with global index ones_at is all ones for k < K (the bias-gradient column)
template <int ROWS, bool GATE, bool KC>
__device__ __forceinline__ void stage_store(const Stage<ROWS>& s, char* planes, float gate_slope, int row0, int k0, int K, int ones_at,
                                            int t) {
    if (KC) {
#pragma unroll
        for (int j = 0; j < (ROWS == 64 ? 2 : 1); ++j) {
            const int item = t + 256 * j, row = item >> 3, kq = item & 7;
            float4 v = quad_value(s.q[j]);
            if (GATE) v = gated(v, pinned(s.g[j].v), gate_slope);
            unsigned h[2], m[2], l[2];
            split2(v.x, v.y, h[0], m[0], l[0]);
            split2(v.z, v.w, h[1], m[1], l[1]);
            char* dst = planes + row * kGPitch + kq * 8;
            *reinterpret_cast<uint2*>(dst) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2*>(dst + kGPlane) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2*>(dst + 2 * kGPlane) = make_uint2(l[0], l[1]);
        }
    } else {
        const int kp = t & 15, rq = t >> 4;
        if (ROWS == 64 || rq < ROWS / 4) {
            float4 v0 = quad_value(s.q[0]), v1 = quad_value(s.q[1]);
            if (GATE) { v0 = gated(v0, pinned(s.g[0].v), gate_slope); v1 = gated(v1, pinned(s.g[1].v), gate_slope); }
            if (ones_at >= 0) {
                const int r = row0 + 4 * rq, k = k0 + 2 * kp;
                const float o0 = k < K ? 1.f : 0.f, o1 = k + 1 < K ? 1.f : 0.f;
                if (r == ones_at) { v0.x = o0; v1.x = o1; }
                if (r + 1 == ones_at) { v0.y = o0; v1.y = o1; }
                if (r + 2 == ones_at) { v0.z = o0; v1.z = o1; }
                if (r + 3 == ones_at) { v0.w = o0; v1.w = o1; }
            }
            char* dst = planes + (4 * rq) * kGPitch + kp * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned h, m, l;
                split2(f4get(v0, i), f4get(v1, i), h, m, l);      // element 0 (even k) in the low half
                *reinterpret_cast<unsigned*>(dst + i * kGPitch) = h;
                *reinterpret_cast<unsigned*>(dst + i * kGPitch + kGPlane) = m;
                *reinterpret_cast<unsigned*>(dst + i * kGPitch + 2 * kGPlane) = l;
            }
        }
    }
}

__device__ __forceinline__ Bf16x3 frag_read(const char* planes, int row, int g) {
    const char* src = planes + row * kGPitch + g * 16;
    Bf16x3 f;
    f.hi = *reinterpret_cast<const bf16x8_t*>(src);
    f.mid = *reinterpret_cast<const bf16x8_t*>(src + kGPlane);
    f.lo = *reinterpret_cast<const bf16x8_t*>(src + 2 * kGPlane);
    return f;
}

// ---- the 16-byte path: nothing but the loads, the splits and the LDS writes in the loop ------------------------------------------
// A thread's items keep their place for the whole tile, so everything about them is decided once: the pointer (rows / columns beyond
// the matrix read a 16-byte block of zeros instead, with step 0), the step per chunk, whether the item lies beyond K in the LAST,
// partial chunk.  The loop then issues ld4(pointer), advances it, splits and writes — no selects, no bounds, no 64-bit index math.
__device__ const float4 g_zero_quad = {0.f, 0.f, 0.f, 0.f};

// A quad whose last 4 - nv elements lie beyond the row (or beyond K) is loaded 4 - nv elements EARLIER — it ends where the data ends, no
// byte outside the matrix is touched — and moved into place here: r[i] = e[i + 4 - nv] for i < nv, 0 beyond.
__device__ __forceinline__ float4 shifted(float4 e, int nv) {
    return make_float4(nv == 4 ? e.x : nv == 3 ? e.y : nv == 2 ? e.z : e.w, nv == 4 ? e.y : nv == 3 ? e.z : nv == 2 ? e.w : 0.f,
                       nv == 4 ? e.z : nv == 3 ? e.w : 0.f, nv == 4 ? e.w : 0.f);
}

template <int ROWS, bool GATE, bool KC, bool PART>
struct VecStage {
    static constexpr int NI = (KC && ROWS == 32) ? 1 : 2;
    const float* p[NI];
    const float* gp[NI];
    long long step[NI];
    bool dead[NI];         // beyond K in the partial last chunk
    int nv[NI];            // elements of the quad that exist (rows along k: in the partial last chunk; rows across k: in every chunk)
    float4 q[NI], g[NI];
    int ones_e;            // rows across k only: which of the item's four rows is the all-ones row (-1 none)

    __device__ __forceinline__ void init(const float* __restrict__ base, const float* __restrict__ gate, long long ls, int row0, int nrows,
                                         int K, int ones_at, int t) {
        const float* zeros = reinterpret_cast<const float*>(&g_zero_quad);
        const int ktail = K & ~(kGK - 1);
        ones_e = -1;
        if (KC) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int item = t + 256 * j, row = row0 + (item >> 3), k = (item & 7) * 4;
                const bool ok = row < nrows;
                const long long off = (long long)row * ls + k;
                p[j] = ok ? base + off : zeros;
                gp[j] = (GATE && ok) ? gate + off : zeros;
                step[j] = ok ? kGK : 0;
                nv[j] = ok ? max(min(K - (ktail + k), 4), 0) : 4;        // (rows beyond the matrix read the zero block as it is)
                dead[j] = nv[j] == 0;
            }
        } else {
            const int kp = t & 15, rq = (t >> 4) & (ROWS / 4 - 1), pos = row0 + 4 * rq;
            const int nvp = max(min(nrows - pos, 4), 0);
            const bool ok = nvp > 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = 2 * kp + j;
                const long long off = (long long)k * ls + pos - (4 - nvp);       // (a partial quad starts earlier: see shifted())
                p[j] = ok ? base + off : zeros;
                gp[j] = (GATE && ok) ? gate + off : zeros;
                step[j] = ok ? kGK * ls : 0;
                nv[j] = ok ? nvp : 4;
                dead[j] = ktail + k >= K;
            }
            if (ones_at >= pos && ones_at < pos + 4) ones_e = ones_at - pos;
        }
    }
    // the loads of chunk cn (nfull = K / 32 full chunks; chunk nfull is the partial one, chunks beyond it are all zeros)
    __device__ __forceinline__ void load(int cn, int nfull, bool has_tail) {
        const float* zeros = reinterpret_cast<const float*>(&g_zero_quad);
        const bool tailc = cn >= nfull, allz = cn > nfull || (cn == nfull && !has_tail);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const bool z = allz || (tailc && dead[j]);
            const int back = (PART && KC && tailc) ? 4 - nv[j] : 0;  // rows along k: the partial quad of the last chunk
            q[j] = ld4g(z ? zeros : p[j] - back);
            if (GATE) g[j] = ld4g(z ? zeros : gp[j] - back);
            p[j] += step[j];
            if (GATE) gp[j] += step[j];
        }
    }
    // partial (uniform): this chunk / tile has quads that reach beyond the data (rows along k: the last chunk; across k: the last tile)
    __device__ __forceinline__ void store(char* planes, float gate_slope, bool ones_live, bool partial, int k0, int K, int t) const {
        if (KC) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int item = t + 256 * j, row = item >> 3, kq = item & 7;
                float4 v = pinned(q[j]);
                float4 gv = GATE ? pinned(g[j]) : v;
                if (PART && partial && !dead[j]) { v = shifted(v, nv[j]); if (GATE) gv = shifted(gv, nv[j]); }
                if (GATE) v = gated(v, gv, gate_slope);
                unsigned h[2], m[2], l[2];
                split2(v.x, v.y, h[0], m[0], l[0]);
                split2(v.z, v.w, h[1], m[1], l[1]);
                char* dst = planes + row * kGPitch + kq * 8;
                *reinterpret_cast<uint2*>(dst) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(dst + kGPlane) = make_uint2(m[0], m[1]);
                *reinterpret_cast<uint2*>(dst + 2 * kGPlane) = make_uint2(l[0], l[1]);
            }
        } else {
            const int kp = t & 15, rq = t >> 4;
            if (ROWS == 64 || rq < ROWS / 4) {
                float4 v0 = pinned(q[0]), v1 = pinned(q[1]);
                float4 g0 = GATE ? pinned(g[0]) : v0, g1 = GATE ? pinned(g[1]) : v1;
                if (PART && partial) {
                    v0 = shifted(v0, nv[0]); v1 = shifted(v1, nv[1]);
                    if (GATE) { g0 = shifted(g0, nv[0]); g1 = shifted(g1, nv[1]); }
                }
                if (GATE) { v0 = gated(v0, g0, gate_slope); v1 = gated(v1, g1, gate_slope); }
                if (ones_live && ones_e >= 0) {          // (ones_live is uniform: only the last column tile has the row)
                    const int k = k0 + 2 * kp;
                    const float o0 = k < K ? 1.f : 0.f, o1 = k + 1 < K ? 1.f : 0.f;
                    if (ones_e == 0) { v0.x = o0; v1.x = o1; }
                    if (ones_e == 1) { v0.y = o0; v1.y = o1; }
                    if (ones_e == 2) { v0.z = o0; v1.z = o1; }
                    if (ones_e == 3) { v0.w = o0; v1.w = o1; }
                }
                char* dst = planes + (4 * rq) * kGPitch + kp * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned h, m, l;
                    split2(f4get(v0, i), f4get(v1, i), h, m, l);      // element 0 (even k) in the low half
                    *reinterpret_cast<unsigned*>(dst + i * kGPitch) = h;
                    *reinterpret_cast<unsigned*>(dst + i * kGPitch + kGPlane) = m;
                    *reinterpret_cast<unsigned*>(dst + i * kGPitch + 2 * kGPlane) = l;
                }
            }
        }
    }
};

// the six partial products of one chunk for the wave's (TR/2 x 32) sub-tile.  Three accumulator chains — the small terms (2^-16 of the
// product and below), the middle ones (2^-8), the large one — summed once at the end: the bf16 matrix instruction aligns its 32
// products and the accumulator by truncation, so small terms added to a large accumulator lose their low bits towards -inf every time
// — a BIAS (measured: mean error -0.1 rms with one chain, z = -80 over 6e5 elements) that the sums downstream of these products
// (bias gradients over 40 k nodes) add up coherently.
template <int MI>
__device__ __forceinline__ void mma_chunk(const char* bufA, const char* bufB, int arow, int bcol, int g, v4f_t (&acc)[3][MI][2]) {
    Bf16x3 fa[MI], fb[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = frag_read(bufA, arow + 16 * i, g);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = frag_read(bufB, bcol + 16 * j, g);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[0][i][j] = mfma_x3_small(fb[j], fa[i], acc[0][i][j]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[1][i][j] = mfma_x3_mid(fb[j], fa[i], acc[1][i][j]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[2][i][j] = mfma_x3_big(fb[j], fa[i], acc[2][i][j]);
}

// One (TR x 64) tile of one product.  Waves 2 x 2: wave (wr, wc) owns rows wr * TR/2 .. and columns wc * 32 ..; B's fragment is the
// FIRST operand of the matrix instruction, so a lane ends up with four consecutive COLUMNS of one row of C (16-byte stores).
template <int TR, bool GATE, bool AKC, bool BKC, int VEC>      // VEC: 0 scalar loads, 1 16-byte loads of whole quads, 2 ... with partial quads
__device__ __forceinline__ void gemm_tile(const GemmJob& jb, int tile, char* lds) {
    constexpr int MI = TR / 32;
    // Two groups of four waves take alternate chunks, half a step apart: while one group splits and writes its next chunk (vector
    // pipe, LDS writes), the other multiplies its current one (matrix pipe, LDS reads) — a wave of each on every SIMD.  With one group
    // the chain load -> split -> write -> barrier -> read -> multiply runs in series, 3-4 x the matrix instructions' own time.
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1, g = lane >> 4, r = lane & 15;
    const int tr = tile / jb.tiles_c, tc = tile - tr * jb.tiles_c;
    const int row0 = tr * TR, col0 = tc * 64;
    const int ones_at = jb.rowsum ? jb.Cn : -1;
    const int ncols_b = jb.Cn;
    const int nchunks = (jb.K + kGK - 1) / kGK;
    const long long a_ls = AKC ? jb.a_rs : jb.a_ks, b_ls = BKC ? jb.b_cs : jb.b_ks;
    const int arow = wr * (TR / 2) + r, bcol = wc * 32 + r;

    v4f_t acc3[3][MI][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < MI; ++i) { acc3[c][i][0] = (v4f_t){0.f, 0.f, 0.f, 0.f}; acc3[c][i][1] = acc3[c][i][0]; }
    // the bias of the lane's columns, requested before the loop (unconditional loads: columns beyond Cn and "no bias" read zeros)
    float4 bias4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + wc * 32 + 16 * j + 4 * g;
        const float* zeros = reinterpret_cast<const float*>(&g_zero_quad);
        const float* bp = jb.bias ? jb.bias : zeros;
        const int last = jb.bias ? jb.Cn - 1 : 0;
        bias4[j] = make_float4(ld1g(bp + min(col, last)), ld1g(bp + min(col + 1, last)), ld1g(bp + min(col + 2, last)), ld1g(bp + min(col + 3, last)));
    }
    DSTAMP(0);
    char* bufA = lds + grp * (6 * kGPlane);            // one buffer per group (its write and its reads are a barrier apart)
    char* bufB = bufA + 3 * kGPlane;
    const int nsteps = (nchunks + 1) / 2;              // chunks per group (chunks beyond K read zeros)
    if (grp == 1) __syncthreads();                     // half a step behind
    if (VEC) {
        // kDepth of the group's chunks in flight in registers; the loop runs whole rounds of kDepth (the stages are indexed statically)
        const int nfull = jb.K / kGK;
        const bool has_tail = (jb.K & (kGK - 1)) != 0;
        const bool ones_live = ones_at >= col0 && ones_at < col0 + 64;
        const bool a_edge = row0 + TR > jb.R, b_edge = col0 + 64 > ncols_b;
        VecStage<TR, GATE, AKC, VEC == 2> sa[kDepth];
        VecStage<64, false, BKC, VEC == 2> sb[kDepth];
#pragma unroll
        for (int s = 0; s < kDepth; ++s) {
            sa[s].init(jb.A, jb.gate, a_ls, row0, jb.R, jb.K, -1, t);
            sb[s].init(jb.B, nullptr, b_ls, col0, ncols_b, jb.K, ones_at, t);
            const int first = 2 * s + grp;              // stage s of group grp starts at chunk 2 s + grp and advances by 2 kDepth
#pragma unroll
            for (int j = 0; j < VecStage<TR, GATE, AKC, false>::NI; ++j) {
                sa[s].p[j] += first * sa[s].step[j]; sa[s].gp[j] += first * sa[s].step[j]; sa[s].step[j] *= 2 * kDepth;
            }
#pragma unroll
            for (int j = 0; j < VecStage<64, false, BKC, false>::NI; ++j) { sb[s].p[j] += first * sb[s].step[j]; sb[s].step[j] *= 2 * kDepth; }
            sa[s].load(first, nfull, has_tail);
            sb[s].load(first, nfull, has_tail);
        }
        DSTAMP(1);
        for (int i0 = 0; i0 < nsteps; i0 += kDepth) {
#pragma unroll
            for (int s = 0; s < kDepth; ++s) {
                const int c = 2 * (i0 + s) + grp;
                const bool tail = has_tail && c == nfull;
                sa[s].store(bufA, jb.gate_slope, false, AKC ? tail : a_edge, c * kGK, jb.K, t);
                sb[s].store(bufB, 0.f, ones_live, BKC ? tail : b_edge, c * kGK, jb.K, t);
                DSTAMP(2 + 4 * (i0 + s));
                __syncthreads();
                DSTAMP(3 + 4 * (i0 + s));
                __builtin_amdgcn_sched_barrier(0);     // the reloads go out AFTER the stage's registers are free: they land in place (no
                                                       // copies at the loop's end, which would have to wait for them)
                sa[s].load(c + 2 * kDepth, nfull, has_tail);
                sb[s].load(c + 2 * kDepth, nfull, has_tail);
                mma_chunk<MI>(bufA, bufB, arow, bcol, g, acc3);
                DSTAMP(4 + 4 * (i0 + s));
                __syncthreads();
                DSTAMP(5 + 4 * (i0 + s));
            }
        }
    } else {
        // any alignment: scalar loads with per-element bounds, one chunk in flight
        Stage<TR> sa;
        Stage<64> sb;
        stage_load<TR, GATE, AKC, false>(sa, jb.A, jb.gate, a_ls, row0, grp * kGK, jb.R, jb.K, t);
        stage_load<64, false, BKC, false>(sb, jb.B, nullptr, b_ls, col0, grp * kGK, ncols_b, jb.K, t);
        const int nrounds = (nsteps + kDepth - 1) / kDepth * kDepth;       // as many barriers as the 16-byte path of the other job
        for (int i = 0; i < nrounds; ++i) {
            const int c = 2 * i + grp;
            stage_store<TR, GATE, AKC>(sa, bufA, jb.gate_slope, row0, c * kGK, jb.K, -1, t);
            stage_store<64, false, BKC>(sb, bufB, 0.f, col0, c * kGK, jb.K, ones_at, t);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            stage_load<TR, GATE, AKC, false>(sa, jb.A, jb.gate, a_ls, row0, (c + 2) * kGK, jb.R, jb.K, t);
            stage_load<64, false, BKC, false>(sb, jb.B, nullptr, b_ls, col0, (c + 2) * kGK, ncols_b, jb.K, t);
            mma_chunk<MI>(bufA, bufB, arow, bcol, g, acc3);
            __syncthreads();
        }
    }
    DSTAMP(60);
    v4f_t acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (acc3[0][i][j] + acc3[1][i][j]) + acc3[2][i][j];
    if (grp == 0) __syncthreads();                     // (the barrier group 1 started with)
    // group 1 hands its sums to group 0
    {
        float* red = reinterpret_cast<float*>(lds);
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<float4*>(red + ((i * 2 + j) * 256 + t) * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 o = *reinterpret_cast<const float4*>(red + ((i * 2 + j) * 256 + t) * 4);
                acc[i][j][0] += o.x; acc[i][j][1] += o.y; acc[i][j][2] += o.z; acc[i][j][3] += o.w;
            }
    }

    // epilogue: lane (g, r) of tile (i, j) holds C[row0 + wr * TR/2 + 16 i + r][col0 + wc * 32 + 16 j + 4 g .. + 3]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = row0 + wr * (TR / 2) + 16 * i + r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wc * 32 + 16 * j + 4 * g;
            if (row >= jb.R) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (ones_at >= 0 && col <= ones_at && ones_at < col + 4) {
                const int e1 = ones_at - col;
                st1g(jb.rowsum + row, e1 == 0 ? v[0] : e1 == 1 ? v[1] : e1 == 2 ? v[2] : v[3]);
            }
            if (col >= jb.Cn) continue;
            v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (jb.act == 1) v[e] = v[e] > 0.f ? v[e] : 0.f;
                else if (jb.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * jb.act_slope;
            }
            float* dst = jb.C + (long long)row * jb.ldc + col;
            if (jb.c_vec && col + 3 < jb.Cn) st4g(dst, make_float4(v[0], v[1], v[2], v[3]));
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < jb.Cn) st1g(dst + e, v[e]);
            }
        }
    }
    DSTAMP(62);
}

// Blocks go to the eight XCDs round robin; consecutive tiles (same rows of A, neighbouring columns) are given to the SAME XCD so that
// the A tile they share is read into one L2.
template <int TR>
__global__ void __launch_bounds__(512) k_dense_x3(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x;
    const int unit = (b & 7) * a.per_xcd + (b >> 3);
    if (unit >= a.ntiles) return;
    const bool second = a.njobs > 1 && unit >= a.job[1].first_tile;
    GemmJob jb = second ? a.job[1] : a.job[0];
    const int S = jb.splits, lu = unit - jb.first_tile, tl = S > 1 ? lu / S : lu, sp = lu - tl * S;
    const int Kall = jb.K;
    const int trow = tl / jb.tiles_c, row0 = trow * TR, col0 = (tl - trow * jb.tiles_c) * 64;
    if (S > 1) {
        const long long k0 = (long long)sp * jb.cps * kGK;
        jb.A += k0 * jb.a_ks; if (jb.gate) jb.gate += k0 * jb.a_ks;
        jb.B += k0 * jb.b_ks;
        jb.K = min(Kall - (int)k0, jb.cps * kGK);
        jb.C = jb.part + ((size_t)tl * S + sp) * (TR * 64) - ((long long)row0 * 64 + col0);
        jb.ldc = 64; jb.c_vec = 1; jb.bias = nullptr; jb.act = 0;
        if (jb.rowsum) jb.rowsum = jb.part_rs + ((size_t)tl * S + sp) * TR - row0;
    }
    // one branch around the whole tile: layout of A, layout of B, gate, 16-byte accesses
#define GLAM_DENSE_CASE(G, AK, BK)                                                     \
    case ((G) * 4 + (AK) * 2 + (BK)):                                                  \
        if (jb.vec == 1) gemm_tile<TR, G, AK, BK, 1>(jb, tl, lds);                     \
        else if (jb.vec == 2) gemm_tile<TR, G, AK, BK, 2>(jb, tl, lds);                \
        else gemm_tile<TR, G, AK, BK, 0>(jb, tl, lds);                                 \
        break;
    switch ((jb.gate ? 4 : 0) + (jb.a_ks == 1 ? 2 : 0) + (jb.b_ks == 1 ? 1 : 0)) {
        GLAM_DENSE_CASE(false, false, false) GLAM_DENSE_CASE(false, false, true) GLAM_DENSE_CASE(false, true, false)
        GLAM_DENSE_CASE(false, true, true) GLAM_DENSE_CASE(true, false, false) GLAM_DENSE_CASE(true, false, true)
        GLAM_DENSE_CASE(true, true, false) GLAM_DENSE_CASE(true, true, true)
    }
#undef GLAM_DENSE_CASE
}

// The second launch of a k-split product: block = one tile of one job, adds the tile's partials in split order and applies the bias /
// activation (a launch boundary instead of tickets and device-wide fences between the blocks of one launch: on this chip a fence
// writes back and invalidates a whole L2 — the fenced form took 18.6 us where the unsplit product takes 8.2).
template <int TR>
__global__ void __launch_bounds__(256) k_dense_splitk_reduce(GemmArgs a) {
    int t = blockIdx.x, q = 0;
    while (q < a.njobs && (a.job[q].splits <= 1 || t >= a.job[q].tiles_r * a.job[q].tiles_c)) {
        if (a.job[q].splits > 1) t -= a.job[q].tiles_r * a.job[q].tiles_c;
        ++q;
    }
    if (q >= a.njobs) return;
    const GemmJob& jb = a.job[q];
    const int S = jb.splits, tl = t, trow = tl / jb.tiles_c, row0 = trow * TR, col0 = (tl - trow * jb.tiles_c) * 64;
    const float* pbase = jb.part + (size_t)tl * S * (TR * 64);
    for (int e = threadIdx.x; e < TR * 16; e += 256) {
        const int r = e >> 4, c4 = (e & 15) * 4, row = row0 + r, col = col0 + c4;
        if (row >= jb.R || col >= jb.Cn) continue;
        float4 acc = ld4g(pbase + r * 64 + c4);
        for (int s = 1; s < S; ++s) {
            const float4 v = ld4g(pbase + (size_t)s * (TR * 64) + r * 64 + c4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (jb.bias && col + i < jb.Cn) v[i] += jb.bias[col + i];
            if (jb.act == 1) v[i] = v[i] > 0.f ? v[i] : 0.f;
            else if (jb.act == 2) v[i] = v[i] > 0.f ? v[i] : v[i] * jb.act_slope;
        }
        float* dst = jb.C + (long long)row * jb.ldc + col;
        if (jb.c_vec && col + 3 < jb.Cn) st4g(dst, make_float4(v[0], v[1], v[2], v[3]));
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (col + i < jb.Cn) st1g(dst + i, v[i]);
        }
    }
    if (jb.rowsum && jb.Cn >= col0 && jb.Cn < col0 + 64 && (int)threadIdx.x < TR && row0 + (int)threadIdx.x < jb.R) {
        const float* pr = jb.part_rs + (size_t)tl * S * TR + threadIdx.x;
        float acc = ld1g(pr);
        for (int s = 1; s < S; ++s) acc += ld1g(pr + (size_t)s * TR);
        st1g(jb.rowsum + row0 + threadIdx.x, acc);
    }
}

// ---- both operands with rows along k (the forward y = x W^T): fragments straight from memory into registers -------------------------
// A lane of the 16 x 16 x 32 matrix instruction holds eight consecutive k of one row: with rows along k that is 32 contiguous bytes of
// the operand itself, so the tile needs no LDS, no transposition and no barrier — each wave loads, splits and multiplies its own
// (32 x 32) sub-tile, two chunks of 32 k in flight in registers.  The two waves that share a row (column) tile load it twice (L1).
// [1024, 300] x [300, 1024]: 11.9 us against 13.3 for the staged kernel above, [2039, 300]: 18.3 against 24.5 (two waves per sub-tile on
// alternate chunks — a second wave per SIMD to overlap split and multiply — measured 12.7 / 22.7: twice the operand loads).  With fewer
// than 128 tiles the staged kernel's 32-row tiles fill the chip better (B = 32: 8.8 against 10.8 us).
struct KcArgs {
    const float* A; long long a_rs; const float* B; long long b_cs; const float* bias; int act; float act_slope;
    float* C; long long ldc; int R, Cn, K, tiles_c, ntiles, per_xcd, c_vec;
};

__device__ __forceinline__ void kc_load(float4 (&q)[2], const float* row, int k, int K, bool row_ok) {
    // (unconditional loads from clamped addresses; what lies beyond K or the matrix is zeroed at the split)
    q[0] = ld4g(row + min(k, K - 4));
    q[1] = ld4g(row + min(k + 4, K - 4));
    (void)row_ok;
}
// `full`: the whole chunk lies inside K (uniform): no select at all.  Rows / columns beyond the matrix keep whatever the clamped loads
// brought — their products only reach output elements that are never stored.  (The selects of the ragged last chunk stand OUTSIDE the
// asm that pins the use of the loaded registers: inside the conditional the compiler made each a divergent branch with an
// s_waitcnt vmcnt(0) in it — every step waited for the prefetch of the step after the next.)
__device__ __forceinline__ Bf16x3 kc_split(const float4 (&q)[2], int k, int K, bool full) {
    if (full) return split8(q[0], q[1]);
    const float4 a = pinned(q[0]), b = pinned(q[1]), z = f4zero();
    return split8(k < K ? a : z, k + 4 < K ? b : z);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_dense_kc(KcArgs a) {
    constexpr int KS = 1;      // (wave groups taking alternate chunks of the same tile — k split inside the block, partial sums through LDS —
                               //  were slower: 12.4 against 10.8 us with two groups; the step is bound by its instruction count)
    const int b = blockIdx.x;
    const int tile = (b & 7) * a.per_xcd + (b >> 3);
    if (tile >= a.ntiles) return;
    const int t = threadIdx.x, grp = 0, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1, g = lane >> 4, r = lane & 15;
    const int tr = tile / a.tiles_c, tc = tile - tr * a.tiles_c;
    const int row0 = tr * 64 + wr * 32, col0 = tc * 64 + wc * 32;
    const float* arow[2];
    const float* brow[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = row0 + 16 * i + r, cb = col0 + 16 * i + r;
        aok[i] = ra < a.R; bok[i] = cb < a.Cn;
        arow[i] = a.A + (long long)min(ra, a.R - 1) * a.a_rs;
        brow[i] = a.B + (long long)min(cb, a.Cn - 1) * a.b_cs;
    }
    float4 bias4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + 16 * j + 4 * g;
        const float* zeros = reinterpret_cast<const float*>(&g_zero_quad);
        const float* bp = a.bias ? a.bias : zeros;
        const int last = a.bias ? a.Cn - 1 : 0;
        bias4[j] = make_float4(ld1g(bp + min(col, last)), ld1g(bp + min(col + 1, last)), ld1g(bp + min(col + 2, last)), ld1g(bp + min(col + 3, last)));
    }
    v4f_t acc3[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc3[c][i][0] = (v4f_t){0.f, 0.f, 0.f, 0.f}; acc3[c][i][1] = acc3[c][i][0]; }
    const int nchunks = (a.K + kGK - 1) / kGK;
    float4 qa[2][2][2], qb[2][2][2];           // [stage][tile][half]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            kc_load(qa[s][i], arow[i], (grp + KS * s) * kGK + 8 * g, a.K, aok[i]);
            kc_load(qb[s][i], brow[i], (grp + KS * s) * kGK + 8 * g, a.K, bok[i]);
        }
    // one step: split stage s (chunk c), refill the stage with chunk c + 2 KS, multiply.  FULL: the chunk lies inside K — no select
    // anywhere in the step (the main loop: a select in it, divergent or not, cost the precise wait counts — every step waited for ALL
    // loads in flight, the prefetch included)
    auto step = [&](int s, int c, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int k = c * kGK + 8 * g;
        Bf16x3 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[i] = kc_split(qa[s][i], k, a.K, FULL); fb[i] = kc_split(qb[s][i], k, a.K, FULL); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) { kc_load(qa[s][i], arow[i], k + 2 * KS * kGK, a.K, aok[i]); kc_load(qb[s][i], brow[i], k + 2 * KS * kGK, a.K, bok[i]); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc3[0][i][j] = mfma_x3_small(fb[j], fa[i], acc3[0][i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc3[1][i][j] = mfma_x3_mid(fb[j], fa[i], acc3[1][i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc3[2][i][j] = mfma_x3_big(fb[j], fa[i], acc3[2][i][j]);
    };
    const int nfull = a.K / kGK;
    int c0 = grp;                                             // this group's chunks: grp, grp + KS, ... (two in flight)
    for (; c0 + KS < nfull; c0 += 2 * KS) {                   // pairs of chunks that both lie inside K
        step(0, c0, std::true_type{});
        step(1, c0 + KS, std::true_type{});
    }
    for (; c0 < nchunks; c0 += 2 * KS) {                      // the last pair: a ragged chunk, one beyond K (all zero)
        step(0, c0, std::false_type{});
        step(1, c0 + KS, std::false_type{});
    }
    v4f_t accs[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accs[i][j] = (acc3[0][i][j] + acc3[1][i][j]) + acc3[2][i][j];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + 16 * i + r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + 16 * j + 4 * g;
            if (row >= a.R || col >= a.Cn) continue;
            const v4f_t s3 = accs[i][j];
            float v[4] = {s3[0] + bias4[j].x, s3[1] + bias4[j].y, s3[2] + bias4[j].z, s3[3] + bias4[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a.act == 1) v[e] = v[e] > 0.f ? v[e] : 0.f;
                else if (a.act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * a.act_slope;
            }
            float* dst = a.C + (long long)row * a.ldc + col;
            if (a.c_vec && col + 3 < a.Cn) st4g(dst, make_float4(v[0], v[1], v[2], v[3]));
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < a.Cn) st1g(dst + e, v[e]);
            }
        }
    }
}

struct Product {
    const float* A; int64_t a_rs, a_ks; const float* gate; float gate_slope;
    const float* B; int64_t b_ks, b_cs; const float* bias; int act; float act_slope;
    float* C; int64_t ldc; float* rowsum; int R, Cn, K;
};

static int check_product(const Product& p, const char* what) {
    GLAM_REQUIRE(p.R >= 1 && p.Cn >= 1 && p.K >= 4, "%s: R = %d, Cn = %d must be >= 1 and K = %d >= 4", what, p.R, p.Cn, p.K);
    GLAM_REQUIRE(p.A && p.B && p.C, "%s: null operand", what);
    // (both strides 1: a single row / column — read as rows along k)
    GLAM_REQUIRE(p.a_rs == 1 || p.a_ks == 1, "%s: one of A's strides (%lld, %lld) must be 1", what, (long long)p.a_rs, (long long)p.a_ks);
    GLAM_REQUIRE(p.b_ks == 1 || p.b_cs == 1, "%s: one of B's strides (%lld, %lld) must be 1", what, (long long)p.b_ks, (long long)p.b_cs);
    GLAM_REQUIRE(p.act >= 0 && p.act <= 2, "%s: act %d (0 none, 1 ReLU, 2 LeakyReLU)", what, p.act);
    GLAM_REQUIRE(!p.rowsum || (p.b_cs == 1 && p.b_ks != 1), "%s: the all-ones column needs B's rows to run across k (b_cs == 1, b_ks > 1)",
                 what);
    GLAM_REQUIRE(p.ldc >= p.Cn, "%s: ldc %lld < Cn %d", what, (long long)p.ldc, p.Cn);
    return 0;
}

static void fill_job(GemmJob& j, const Product& p, int TR, int first_tile) {
    j.A = p.A; j.a_rs = p.a_rs; j.a_ks = p.a_ks; j.gate = p.gate; j.gate_slope = p.gate_slope;
    j.B = p.B; j.b_ks = p.b_ks; j.b_cs = p.b_cs; j.bias = p.bias; j.act = p.act; j.act_slope = p.act_slope;
    j.C = p.C; j.ldc = p.ldc; j.rowsum = p.rowsum; j.R = p.R; j.Cn = p.Cn; j.K = p.K;
    j.tiles_r = (p.R + TR - 1) / TR;
    j.tiles_c = (p.Cn + (p.rowsum ? 1 : 0) + 63) / 64;
    j.first_tile = first_tile;
    // 16-byte loads need dword alignment only on this hardware; the fast path needs room to start a partial quad early (shifted()):
    // at least 4 elements along the unit-stride dimension of both operands (K >= 4 is required anyway)
    j.vec = (p.a_ks == 1 || p.R >= 4) && (p.b_ks == 1 || p.Cn >= 4);
    // 2: some quad of some row is partial (the reduction length for rows along k, the row / column count for rows across k)
    if (j.vec && (((p.a_ks == 1 || p.b_ks == 1) && p.K % 4) || (p.a_ks != 1 && p.R % 4) || (p.b_ks != 1 && p.Cn % 4))) j.vec = 2;
    j.c_vec = aligned16(p.C) && p.ldc % 4 == 0;
    j.splits = 1; j.cps = 0; j.part = nullptr; j.part_rs = nullptr;
}

static bool kc_route(const Product& p) {
    static const bool on = true;
    return on && p.a_ks == 1 && p.b_ks == 1 && !p.gate && !p.rowsum && p.K >= 8 && p.K % 4 == 0 && p.a_rs % 4 == 0 && p.b_cs % 4 == 0 &&
           aligned16(p.A) && aligned16(p.B);
}

static int launch_products(const Product* p, int n, hipStream_t s, void* ws = nullptr, size_t ws_bytes = 0) {
    if (n == 1 && kc_route(p[0]) && (long long)((p[0].R + 63) / 64) * ((p[0].Cn + 63) / 64) >= 128) {      // (few tiles: the 32-row staged tiles fill the chip better)
        KcArgs a{p[0].A, p[0].a_rs, p[0].B, p[0].b_cs, p[0].bias, p[0].act, p[0].act_slope, p[0].C, p[0].ldc, p[0].R, p[0].Cn, p[0].K,
                 (p[0].Cn + 63) / 64, 0, 0, aligned16(p[0].C) && p[0].ldc % 4 == 0};
        a.ntiles = ((p[0].R + 63) / 64) * a.tiles_c;
        a.per_xcd = (a.ntiles + 7) / 8;
        hipLaunchKernelGGL(k_dense_kc, dim3(a.per_xcd * 8), dim3(256), 0, s, a);
        GLAM_LAUNCH_CHECK("k_dense_kc");
        return 0;
    }
    // 64-row tiles; 32-row tiles (twice the blocks) for the small batches that would leave most of the chip idle
    long long t64 = 0;
    for (int i = 0; i < n; ++i) t64 += (long long)((p[i].R + 63) / 64) * ((p[i].Cn + (p[i].rowsum ? 1 : 0) + 63) / 64);
    const int TR = t64 >= 100 ? 64 : 32;
    GemmArgs a{};
    a.njobs = n;
    int first = 0;
    // a job of few tiles and a long reduction: k split across blocks, up to 8 ways, at least four chunks per split, within the CUs the other
    // job leaves (needs the caller's workspace; a second launch adds the partials)
    long long tiles_all = 0;
    for (int i = 0; i < n; ++i) tiles_all += (long long)((p[i].R + TR - 1) / TR) * ((p[i].Cn + (p[i].rowsum ? 1 : 0) + 63) / 64);
    int reduce_tiles = 0;
    for (int i = 0; i < n; ++i) {
        fill_job(a.job[i], p[i], TR, first);
        long long tl = (long long)a.job[i].tiles_r * a.job[i].tiles_c;
        const int nch = (p[i].K + kGK - 1) / kGK;
        if (ws && ws_bytes >= kSplitWsBytes && tl <= 32 && nch >= 16) {
            const long long room = 256 - (tiles_all - tl);
            int S = (int)((room > tl ? room : tl) / tl);
            if (S > 8) S = 8;
            if (S > nch / 4) S = nch / 4;
            if (S > 1) {
                const int cps = (nch + S - 1) / S;
                S = (nch + cps - 1) / cps;
                if (S > 1 && first + tl * S <= kSplitMaxUnits) {
                    GemmJob& j = a.job[i];
                    j.splits = S; j.cps = cps;
                    j.part_rs = reinterpret_cast<float*>(ws) + (size_t)first * 64;
                    j.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)kSplitMaxUnits * 64 * 4) + (size_t)first * 64 * 64;
                    reduce_tiles += (int)tl;
                    tl *= S;
                }
            }
        }
        GLAM_REQUIRE(first + tl < (1 << 24), "glam_dense_gemm: %lld tiles", first + tl);
        first += (int)tl;
    }
    a.ntiles = first;
    a.per_xcd = (first + 7) / 8;
    const int grid = a.per_xcd * 8;
    if (TR == 64) hipLaunchKernelGGL(k_dense_x3<64>, dim3(grid), dim3(512), kGLds, s, a);
    else hipLaunchKernelGGL(k_dense_x3<32>, dim3(grid), dim3(512), kGLds, s, a);
    GLAM_LAUNCH_CHECK("k_dense_x3");
    if (reduce_tiles) {
        if (TR == 64) hipLaunchKernelGGL(k_dense_splitk_reduce<64>, dim3(reduce_tiles), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_dense_splitk_reduce<32>, dim3(reduce_tiles), dim3(256), 0, s, a);
        GLAM_LAUNCH_CHECK("k_dense_splitk_reduce");
    }
    return 0;
}

}  // namespace glam

using namespace glam;

#ifdef GLAM_DENSE_STAMP
extern "C" int glam_debug_dense_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_dense_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int glam_dense_gemm(const float* A, int64_t a_rs, int64_t a_ks, const float* gate, float gate_slope, const float* B,
                               int64_t b_ks, int64_t b_cs, const float* bias, int act, float act_slope, float* C, int64_t ldc,
                               float* rowsum, int R, int Cn, int K, void* stream) {
    const Product p{A, a_rs, a_ks, gate, gate_slope, B, b_ks, b_cs, bias, act, act_slope, C, ldc, rowsum, R, Cn, K};
    if (int rc = check_product(p, "glam_dense_gemm")) return rc;
    return launch_products(&p, 1, (hipStream_t)stream);
}

extern "C" int glam_linear_dense_fwd(const float* x, const float* w, const float* b, int64_t N, int K, int M, int act, float slope,
                                     float* y, void* stream) {
    GLAM_REQUIRE(N >= 1 && N < (1 << 30), "glam_linear_dense_fwd: N = %lld", (long long)N);
    const Product p{x, K, 1, nullptr, 0.f, w, 1, K, b, act, slope, y, M, nullptr, (int)N, M, K};
    if (int rc = check_product(p, "glam_linear_dense_fwd")) return rc;
    return launch_products(&p, 1, (hipStream_t)stream);
}

extern "C" size_t glam_dense_ws_bytes(void) { return kSplitWsBytes; }

static int ws_ok(const char* fn, const void* ws, size_t ws_bytes) {
    GLAM_REQUIRE(!ws || (aligned16(ws) && ws_bytes >= kSplitWsBytes), "%s: the workspace needs glam_dense_ws_bytes() = %zu bytes, 16-byte aligned", fn,
                 kSplitWsBytes);
    return 0;
}

extern "C" int glam_linear_dense_fwd_ws(const float* x, const float* w, const float* b, int64_t N, int K, int M, int act, float slope,
                                        float* y, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(N >= 1 && N < (1 << 30), "glam_linear_dense_fwd_ws: N = %lld", (long long)N);
    if (int rc = ws_ok("glam_linear_dense_fwd_ws", ws, ws_bytes)) return rc;
    const Product p{x, K, 1, nullptr, 0.f, w, 1, K, b, act, slope, y, M, nullptr, (int)N, M, K};
    if (int rc = check_product(p, "glam_linear_dense_fwd_ws")) return rc;
    return launch_products(&p, 1, (hipStream_t)stream, ws, ws_bytes);
}

static int linear_dense_bwd_impl(const char* fn, const float* x, const float* w, const float* dy, const float* y_gate, float gate_slope,
                                 int64_t N, int K, int M, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t s);
extern "C" int glam_linear_dense_bwd_ws(const float* x, const float* w, const float* dy, const float* y_gate, float gate_slope, int64_t N,
                                        int K, int M, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = ws_ok("glam_linear_dense_bwd_ws", ws, ws_bytes)) return rc;
    return linear_dense_bwd_impl("glam_linear_dense_bwd_ws", x, w, dy, y_gate, gate_slope, N, K, M, dx, dw, db, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int glam_linear_dense_bwd(const float* x, const float* w, const float* dy, const float* y_gate, float gate_slope, int64_t N,
                                     int K, int M, float* dx, float* dw, float* db, void* stream) {
    return linear_dense_bwd_impl("glam_linear_dense_bwd", x, w, dy, y_gate, gate_slope, N, K, M, dx, dw, db, nullptr, 0, (hipStream_t)stream);
}
static int linear_dense_bwd_impl(const char* fn, const float* x, const float* w, const float* dy, const float* y_gate, float gate_slope,
                                 int64_t N, int K, int M, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t stream) {
    (void)fn;
    GLAM_REQUIRE(N >= 4 && N < (1 << 30), "glam_linear_dense_bwd: N = %lld (the weight gradient reduces over N >= 4 rows)", (long long)N);
    GLAM_REQUIRE(dw || dx, "glam_linear_dense_bwd: nothing to compute");
    GLAM_REQUIRE(dw || !db, "glam_linear_dense_bwd: db comes with dw");
    Product p[2];
    int n = 0;
    // dW[M, K] = (dy . gate)^T x: A(i, n) = dy[n, i], B(n, j) = x[n, j]; the bias gradient is the all-ones column
    if (dw) p[n++] = Product{dy, 1, M, y_gate, gate_slope, x, K, 1, nullptr, 0, 0.f, dw, K, db, M, K, (int)N};
    // dx[N, K] = (dy . gate) w: B(k, j) = w[k, j]
    if (dx) p[n++] = Product{dy, M, 1, y_gate, gate_slope, w, K, 1, nullptr, 0, 0.f, dx, K, nullptr, (int)N, K, M};
    for (int i = 0; i < n; ++i)
        if (int rc = check_product(p[i], "glam_linear_dense_bwd")) return rc;
    return launch_products(p, n, stream, ws, ws_bytes);
}
