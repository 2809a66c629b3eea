// Per-pair fusion of the two-tower (ligand + protein) models: for every pair i,
//   S_i = mol[seg_i] @ pro[seg_i]^T  ([n_mol, n_res]),  out[i] = [max(S_i), mean(S_i)]
// Reference: dot_and_global_pool2 (src_2gi_dti_scr/layer.py:270-283) — a Python loop with two .item() syncs and a
// matmul per pair.  Here the ligand rows are staged in LDS and each lane owns a residue row in registers; a pair's residues
// are split over kPairSplit blocks (a batch of 32 pairs would otherwise light 32 of 256 CUs), whose (max, argmax, column
// sum) partials a one-block-per-pair kernel combines in a fixed order.  The mean needs no pairwise work at all:
// sum(S_i) = <sum_a mol_a, sum_b pro_b>; the two column sums are kept for the backward pass.
#include "common.h"

namespace glam {

constexpr int kMolTile = 32;      // ligand rows staged per LDS pass
constexpr int kMaxD = 256;
constexpr int kPairSplit = 16;    // residue chunks (blocks) per pair in the split path
constexpr int kResChunk = 64;     // residues per chunk pass: one per lane
constexpr int kPartStride = 68;   // floats per (pair, split) partial: val, idx, pad, pad, colsum[64]

__device__ __forceinline__ bool better(float v, int ix, float best, int bidx) { return v > best || (v == best && ix < bidx); }

// Split path (D % 4 == 0, D <= 64).  Block (i, s): residues s*64 + k*16*64 + lane of pair i; wave q takes the ligand rows
// a = q (mod 4), two at a time (independent dot-product chains; each dot keeps the channel order of the scalar path, so
// values and argmax are those of the one-block kernel).
__global__ void __launch_bounds__(kBlock) k_pair_max_partial(const float* mol, const float* pro, const int* mptr,
                                                            const int* pptr, int D, float* part) {
    __shared__ __attribute__((aligned(16))) float s_mol[kMolTile * 64];
    __shared__ float s_val[kBlock];
    __shared__ int s_idx[kBlock];
    const int i = blockIdx.x / kPairSplit, sp = blockIdx.x % kPairSplit, tid = threadIdx.x;
    const int lane = tid & 63, q = tid >> 6;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], p1 = pptr[i + 1];
    const int nm = m1 - m0, np = p1 - p0;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    float4 cs[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) cs[u] = f4zero();
    for (int b0 = sp * kResChunk; b0 < np; b0 += kPairSplit * kResChunk) {
        const int b = b0 + lane;
        const bool valid = b < np;
        const float* prow = pro + (size_t)(p0 + (valid ? b : 0)) * D;
        float4 pr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            pr[u] = (valid && 4 * u < D) ? ld4(prow + 4 * u) : f4zero();
            cs[u].x += pr[u].x; cs[u].y += pr[u].y; cs[u].z += pr[u].z; cs[u].w += pr[u].w;
        }
        for (int t0 = 0; t0 < nm; t0 += kMolTile) {
            const int tn = min(kMolTile, nm - t0);
            __syncthreads();
            for (int k = tid; k < tn * D; k += kBlock) s_mol[k] = mol[(size_t)(m0 + t0) * D + k];
            __syncthreads();
            for (int a = q; a < tn; a += 8) {
                const bool two = a + 4 < tn;
                const float* r0 = s_mol + a * D;
                const float* r1 = s_mol + (two ? a + 4 : a) * D;
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (4 * u < D) {
                        const float4 v0 = ld4(r0 + 4 * u), v1 = ld4(r1 + 4 * u);
                        d0 = fmaf(v0.x, pr[u].x, d0); d0 = fmaf(v0.y, pr[u].y, d0); d0 = fmaf(v0.z, pr[u].z, d0); d0 = fmaf(v0.w, pr[u].w, d0);
                        d1 = fmaf(v1.x, pr[u].x, d1); d1 = fmaf(v1.y, pr[u].y, d1); d1 = fmaf(v1.z, pr[u].z, d1); d1 = fmaf(v1.w, pr[u].w, d1);
                    }
                }
                if (valid) {
                    const int i0 = (t0 + a) * np + b, i1 = (t0 + a + 4) * np + b;
                    if (better(d0, i0, best, bidx)) { best = d0; bidx = i0; }
                    if (two && better(d1, i1, best, bidx)) { best = d1; bidx = i1; }
                }
            }
        }
    }
    s_val[tid] = best;
    s_idx[tid] = bidx;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if (tid < o && better(s_val[tid + o], s_idx[tid + o], s_val[tid], s_idx[tid])) { s_val[tid] = s_val[tid + o]; s_idx[tid] = s_idx[tid + o]; }
        __syncthreads();
    }
    float* dst = part + (size_t)blockIdx.x * kPartStride;
    if (tid == 0) { dst[0] = s_val[0]; reinterpret_cast<int*>(dst)[1] = s_idx[0]; }
    if (q == 0) {                    // column sums of this block's residues (every wave loaded the same rows)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float x = group_sum<64>(cs[u].x), y = group_sum<64>(cs[u].y), z = group_sum<64>(cs[u].z), w = group_sum<64>(cs[u].w);
            if (lane == 0) st4(dst + 4 + 4 * u, make_float4(x, y, z, w));
        }
    }
}

__global__ void __launch_bounds__(64) k_pair_finish(const float* mol, const int* mptr, const int* pptr, const float* part,
                                                   int D, float* out, int* arg, float* sums) {
    const int i = blockIdx.x, c = threadIdx.x;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], np = pptr[i + 1] - p0, nm = m1 - m0;
    const float* pp = part + (size_t)i * kPairSplit * kPartStride;
    float ps = 0.f, ms = 0.f;
    if (c < D) {
        for (int s = 0; s < kPairSplit; ++s) ps += pp[s * kPartStride + 4 + c];
        int a = m0;
        for (; a + 3 < m1; a += 4) {
            const float v0 = mol[(size_t)a * D + c], v1 = mol[(size_t)(a + 1) * D + c], v2 = mol[(size_t)(a + 2) * D + c],
                        v3 = mol[(size_t)(a + 3) * D + c];
            ms += v0; ms += v1; ms += v2; ms += v3;
        }
        for (; a < m1; ++a) ms += mol[(size_t)a * D + c];
        sums[(size_t)i * 2 * D + c] = ms;
        sums[(size_t)i * 2 * D + D + c] = ps;
    }
    const float tot = group_sum<64>(ms * ps);
    if (c == 0) {
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        for (int s = 0; s < kPairSplit; ++s) {
            const float v = pp[s * kPartStride];
            const int ix = reinterpret_cast<const int*>(pp + s * kPartStride)[1];
            if (better(v, ix, best, bidx)) { best = v; bidx = ix; }
        }
        const bool empty = nm <= 0 || np <= 0;
        out[2 * i] = empty ? 0.f : best;
        out[2 * i + 1] = empty ? 0.f : tot / ((float)nm * (float)np);
        arg[2 * i] = empty ? -1 : m0 + bidx / np;
        arg[2 * i + 1] = empty ? -1 : p0 + bidx % np;
    }
}

// d_mol[a] = g_max * [a == a*] * pro[b*] + g_mean / (nm np) * sum_b pro_b ;  d_pro symmetric.  Block (i, s): the residue
// rows s*16 + k*16*16 + rg of pair i (and, for s == 0, its ligand rows); column sums come from the forward pass.
__global__ void __launch_bounds__(kBlock) k_pair_pool_bwd_split(const float* mol, const float* pro, const int* mptr,
                                                               const int* pptr, const int* arg, const float* sums,
                                                               const float* d_out, int D, float* d_mol, float* d_pro,
                                                               const float* add_mol, const float* add_pro) {
    // add_mol / add_pro (may be null): a second gradient path into the two feature matrices (the next message step's, when the
    // caller took them back from this node), added last — what an add launch of the autograd engine would compute
    const int i = blockIdx.x / kPairSplit, sp = blockIdx.x % kPairSplit, tid = threadIdx.x;
    const int c4 = tid & 15, rg = tid >> 4;
    if (4 * c4 >= D) return;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], p1 = pptr[i + 1];
    const int nm = m1 - m0, np = p1 - p0;
    const bool empty = nm <= 0 || np <= 0;
    const float gmax = empty ? 0.f : d_out[2 * i], gmean = empty ? 0.f : d_out[2 * i + 1] / ((float)nm * (float)np);
    const int am = empty ? -1 : arg[2 * i], ap = empty ? -1 : arg[2 * i + 1];
    const float4 msum = empty ? f4zero() : gmean * ld4(sums + (size_t)i * 2 * D + 4 * c4);
    const float4 psum = empty ? f4zero() : gmean * ld4(sums + (size_t)i * 2 * D + D + 4 * c4);
    for (int b = p0 + sp * 16 + rg; b < p1; b += kPairSplit * 16) {
        float4 v = msum;
        if (b == ap) {
            const float4 t = ld4(mol + (size_t)am * D + 4 * c4);
            v.x = fmaf(gmax, t.x, v.x); v.y = fmaf(gmax, t.y, v.y); v.z = fmaf(gmax, t.z, v.z); v.w = fmaf(gmax, t.w, v.w);
        }
        if (add_pro) { const float4 t = ld4(add_pro + (size_t)b * D + 4 * c4); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        st4(d_pro + (size_t)b * D + 4 * c4, v);
    }
    if (sp == 0)
        for (int a = m0 + rg; a < m1; a += 16) {
            float4 v = psum;
            if (a == am) {
                const float4 t = ld4(pro + (size_t)ap * D + 4 * c4);
                v.x = fmaf(gmax, t.x, v.x); v.y = fmaf(gmax, t.y, v.y); v.z = fmaf(gmax, t.z, v.z); v.w = fmaf(gmax, t.w, v.w);
            }
            if (add_mol) { const float4 t = ld4(add_mol + (size_t)a * D + 4 * c4); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            st4(d_mol + (size_t)a * D + 4 * c4, v);
        }
}

__global__ void __launch_bounds__(kBlock) k_pair_pool_fwd(const float* mol, const float* pro, const int* mptr,
                                                         const int* pptr, int D, float* out, int* arg) {
    __shared__ __attribute__((aligned(16))) float s_mol[kMolTile * kMaxD];
    __shared__ float s_val[kBlock];
    __shared__ int s_idx[kBlock];
    __shared__ float s_sum[2 * kMaxD];
    __shared__ __attribute__((aligned(16))) float s_part[16 * 16 * 4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], p1 = pptr[i + 1];
    const int nm = m1 - m0, np = p1 - p0;
    // column sums of both segments (mean)
    block_colsum(mol, m0, m1, D, s_part, s_sum);
    block_colsum(pro, p0, p1, D, s_part, s_sum + D);
    float best = -INFINITY;
    int bidx = 0x7fffffff;          // flattened (a * np + b): first occurrence wins ties, like a flattened argmax
    const bool vec = (D & 3) == 0 && D <= 64;
    for (int t0 = 0; t0 < nm; t0 += kMolTile) {
        const int tn = min(kMolTile, nm - t0);
        __syncthreads();
        for (int k = tid; k < tn * D; k += kBlock) s_mol[k] = mol[(size_t)(m0 + t0) * D + k];
        __syncthreads();
        for (int b = tid; b < np; b += kBlock) {
            const float* prow = pro + (size_t)(p0 + b) * D;
            if (vec) {
                // the residue row lives in registers (one set of loads per residue instead of one per ligand row); the ligand
                // rows are LDS broadcasts.  Same c order as the scalar path: identical dot products and argmax.
                float4 pr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pr[u] = 4 * u < D ? ld4(prow + 4 * u) : f4zero();
                for (int a = 0; a < tn; ++a) {
                    float d = 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        if (4 * u < D) {
                            const float4 mv = ld4(s_mol + a * D + 4 * u);
                            d = fmaf(mv.x, pr[u].x, d); d = fmaf(mv.y, pr[u].y, d); d = fmaf(mv.z, pr[u].z, d); d = fmaf(mv.w, pr[u].w, d);
                        }
                    }
                    const int idx = (t0 + a) * np + b;
                    if (d > best || (d == best && idx < bidx)) { best = d; bidx = idx; }
                }
            } else {
                for (int a = 0; a < tn; ++a) {
                    float d = 0.f;
                    for (int c = 0; c < D; ++c) d = fmaf(s_mol[a * D + c], prow[c], d);
                    const int idx = (t0 + a) * np + b;
                    if (d > best || (d == best && idx < bidx)) { best = d; bidx = idx; }
                }
            }
        }
    }
    s_val[tid] = best;
    s_idx[tid] = bidx;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if (tid < o) {
            const float v = s_val[tid + o];
            const int ix = s_idx[tid + o];
            if (v > s_val[tid] || (v == s_val[tid] && ix < s_idx[tid])) { s_val[tid] = v; s_idx[tid] = ix; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int c = 0; c < D; ++c) tot = fmaf(s_sum[c], s_sum[D + c], tot);
        const bool empty = nm <= 0 || np <= 0;
        out[2 * i] = empty ? 0.f : s_val[0];
        out[2 * i + 1] = empty ? 0.f : tot / ((float)nm * (float)np);
        arg[2 * i] = empty ? -1 : m0 + s_idx[0] / np;
        arg[2 * i + 1] = empty ? -1 : p0 + s_idx[0] % np;
    }
}

// d_mol[a] = g_max * [a == a*] * pro[b*] + g_mean / (nm np) * sum_b pro_b ;  d_pro symmetric
__global__ void __launch_bounds__(kBlock) k_pair_pool_bwd(const float* mol, const float* pro, const int* mptr,
                                                         const int* pptr, const int* arg, const float* d_out, int D,
                                                         float* d_mol, float* d_pro) {
    __shared__ float s_sum[2 * kMaxD];
    __shared__ __attribute__((aligned(16))) float s_part[16 * 16 * 4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], p1 = pptr[i + 1];
    const int nm = m1 - m0, np = p1 - p0;
    if (nm <= 0 || np <= 0) {
        for (int k = tid; k < max(nm, 0) * D; k += kBlock) d_mol[(size_t)m0 * D + k] = 0.f;
        for (int k = tid; k < max(np, 0) * D; k += kBlock) d_pro[(size_t)p0 * D + k] = 0.f;
        return;
    }
    block_colsum(mol, m0, m1, D, s_part, s_sum);
    block_colsum(pro, p0, p1, D, s_part, s_sum + D);
    const float gmax = d_out[2 * i], gmean = d_out[2 * i + 1] / ((float)nm * (float)np);
    const int am = arg[2 * i], ap = arg[2 * i + 1];
    for (int k = tid; k < nm * D; k += kBlock) {
        const int a = m0 + k / D, c = k % D;
        d_mol[(size_t)a * D + c] = gmean * s_sum[D + c] + (a == am ? gmax * pro[(size_t)ap * D + c] : 0.f);
    }
    for (int k = tid; k < np * D; k += kBlock) {
        const int b = p0 + k / D, c = k % D;
        d_pro[(size_t)b * D + c] = gmean * s_sum[c] + (b == ap ? gmax * mol[(size_t)am * D + c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// dot_and_global_pool5 (src_1gp/layer.py:270-283): out[i] = [max, mean, median, min, std] of S_i = mol[seg_i] @ pro[seg_i]^T.
// The reference loops over pairs in Python (matmul + five reductions + .item() syncs per pair).  Here: one block per pair, no
// score matrix in memory — every pass recomputes the scores with the same instruction sequence (bit-identical values in every
// pass): a 16-lane group owns a residue row (its float4 chunks in registers), walks the ligand rows staged in LDS and forms
// s = <mol_a, pro_b> with a DPP butterfly.
//   pass 1  max / min with first-occurrence arg (flattened a * np + b order, like a flattened argmax);  mean from the column
//           sums: sum(S) = <sum_a mol_a, sum_b pro_b>
//   pass 2  sum (s - mean)^2 in a fixed order -> unbiased std (torch.std default)
//   pass 3..6  exact lower median (torch.median of a flattened tensor: rank (n - 1) / 2) by radix select on the order-preserving
//           integer image of the float, 8 bits per pass (LDS histogram, integer atomics: order independent)
//   pass 7  first flattened index holding the median value (for the backward pass)
constexpr int kP5Groups = kBlock / 16;

__device__ __forceinline__ unsigned p5_key(float v) {        // monotone float -> uint map
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float p5_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// f(flat index a * np + b, score) for every score of the pair, called by ALL 16 lanes of the owning group (same value in each);
// contains block barriers: every thread of the block must call it.
template <typename F>
__device__ __forceinline__ void p5_for_each_score(const float* mol, const float* pro, int m0, int nm, int p0, int np, int D,
                                                  float* s_mol, F&& f) {
    const int tid = threadIdx.x, lg = tid & 15, gid = tid >> 4, Q = D >> 2;
    const bool ok0 = lg < Q, ok1 = lg + 16 < Q;
    for (int t0 = 0; t0 < nm; t0 += kMolTile) {
        const int tn = min(kMolTile, nm - t0);
        __syncthreads();
        for (int k = tid; k < tn * D; k += kBlock) s_mol[k] = mol[(size_t)(m0 + t0) * D + k];
        __syncthreads();
        for (int b = gid; b < np; b += kP5Groups) {
            const float* prow = pro + (size_t)(p0 + b) * D;
            const float4 pr0 = ok0 ? ld4(prow + 4 * lg) : f4zero(), pr1 = ok1 ? ld4(prow + 4 * (lg + 16)) : f4zero();
            for (int a = 0; a < tn; ++a) {
                float part = ok0 ? dot4(ld4(s_mol + a * D + 4 * lg), pr0) : 0.f;
                if (ok1) part += dot4(ld4(s_mol + a * D + 4 * (lg + 16)), pr1);
                f((t0 + a) * np + b, group_sum<16>(part));
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_pair_stats5(const float* mol, const float* pro, const int* mptr, const int* pptr,
                                                       int D, float* out, int* arg) {
    __shared__ __attribute__((aligned(16))) float s_mol[kMolTile * 128];
    __shared__ float s_sum[2 * 128];
    __shared__ __attribute__((aligned(16))) float s_part[16 * 16 * 4];
    __shared__ float s_gv[2][kP5Groups];
    __shared__ int s_gi[2][kP5Groups];
    __shared__ int s_hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_rank;
    __shared__ float s_mean;
    const int i = blockIdx.x, tid = threadIdx.x, lg = tid & 15, gid = tid >> 4;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], p1 = pptr[i + 1];
    const int nm = m1 - m0, np = p1 - p0;
    if (nm <= 0 || np <= 0) {          // the reference raises on an empty pair; a sharded batch may hold one: zeros
        if (tid < 5) out[5 * i + tid] = 0.f;
        if (tid < 6) arg[6 * i + tid] = -1;
        return;
    }
    const float n = (float)nm * (float)np;
    if ((D & 3) == 0 && D <= 64) {
        block_colsum(mol, m0, m1, D, s_part, s_sum);
        block_colsum(pro, p0, p1, D, s_part, s_sum + 128);
    } else {
        for (int c = tid; c < D; c += kBlock) {
            float a = 0.f, b = 0.f;
            for (int r = m0; r < m1; ++r) a += mol[(size_t)r * D + c];
            for (int r = p0; r < p1; ++r) b += pro[(size_t)r * D + c];
            s_sum[c] = a; s_sum[128 + c] = b;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int c = 0; c < D; ++c) tot = fmaf(s_sum[c], s_sum[128 + c], tot);
        s_mean = tot / n;
    }
    // ---- pass 1: max / min ----
    float vmax = -INFINITY, vmin = INFINITY;
    int imax = 0x7fffffff, imin = 0x7fffffff;
    p5_for_each_score(mol, pro, m0, nm, p0, np, D, s_mol, [&](int idx, float v) {
        if (v > vmax || (v == vmax && idx < imax)) { vmax = v; imax = idx; }
        if (v < vmin || (v == vmin && idx < imin)) { vmin = v; imin = idx; }
    });
    if (lg == 0) { s_gv[0][gid] = vmax; s_gi[0][gid] = imax; s_gv[1][gid] = vmin; s_gi[1][gid] = imin; }
    __syncthreads();
    if (tid == 0) {
        for (int g = 0; g < kP5Groups; ++g) {
            if (s_gv[0][g] > vmax || (s_gv[0][g] == vmax && s_gi[0][g] < imax)) { vmax = s_gv[0][g]; imax = s_gi[0][g]; }
            if (s_gv[1][g] < vmin || (s_gv[1][g] == vmin && s_gi[1][g] < imin)) { vmin = s_gv[1][g]; imin = s_gi[1][g]; }
        }
        out[5 * i + 0] = vmax; out[5 * i + 1] = s_mean; out[5 * i + 3] = vmin;
        arg[6 * i + 0] = m0 + imax / np; arg[6 * i + 1] = p0 + imax % np;
        arg[6 * i + 4] = m0 + imin / np; arg[6 * i + 5] = p0 + imin % np;
    }
    __syncthreads();
    // ---- pass 2: unbiased variance about the mean ----
    const float mean = s_mean;
    float ssq = 0.f;
    p5_for_each_score(mol, pro, m0, nm, p0, np, D, s_mol, [&](int, float v) { const float d = v - mean; ssq = fmaf(d, d, ssq); });
    if (lg == 0) s_gv[0][gid] = ssq;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int g = 0; g < kP5Groups; ++g) t += s_gv[0][g];
        out[5 * i + 4] = sqrtf(t / (n - 1.f));            // n == 1: 0 / 0 = NaN, as torch.std
        s_prefix = 0u;
        s_rank = (nm * np - 1) / 2;                        // lower median (torch.median of the flattened scores)
    }
    // ---- passes 3..6: radix select ----
    for (int pass = 0; pass < 4; ++pass) {
        s_hist[tid] = 0;                                   // kBlock == 256 bins
        __syncthreads();
        const int shift = 24 - 8 * pass;
        const unsigned prefix = s_prefix;
        p5_for_each_score(mol, pro, m0, nm, p0, np, D, s_mol, [&](int, float v) {
            const unsigned k = p5_key(v);
            if (lg == 0 && (pass == 0 || (k >> (shift + 8)) == prefix)) atomicAdd(&s_hist[(k >> shift) & 255u], 1);
        });
        __syncthreads();
        if (tid == 0) {
            int r = s_rank, bin = 0;
            for (; bin < 255; ++bin) {
                if (r < s_hist[bin]) break;
                r -= s_hist[bin];
            }
            s_rank = r;
            s_prefix = (prefix << 8) | (unsigned)bin;
        }
        __syncthreads();
    }
    // ---- pass 7: first flattened index of the median value ----
    const unsigned medkey = s_prefix;
    int imed = 0x7fffffff;
    p5_for_each_score(mol, pro, m0, nm, p0, np, D, s_mol, [&](int idx, float v) { if (p5_key(v) == medkey && idx < imed) imed = idx; });
    if (lg == 0) s_gi[0][gid] = imed;
    __syncthreads();
    if (tid == 0) {
        for (int g = 0; g < kP5Groups; ++g) imed = min(imed, s_gi[0][g]);
        out[5 * i + 2] = p5_unkey(medkey);
        arg[6 * i + 2] = m0 + imed / np; arg[6 * i + 3] = p0 + imed % np;
    }
}

// Backward of the five statistics: dS[a,b] = g_mean / n + g_std (S[a,b] - mean) / ((n - 1) std) + g_max [arg max] + g_med [arg med]
// + g_min [arg min];  d_mol[a] = sum_b dS[a,b] pro[b],  d_pro[b] = sum_a dS[a,b] mol[a].  A 16-lane group owns an output row and
// walks the rows of the other side in index order (deterministic), recomputing each score.
__global__ void __launch_bounds__(kBlock) k_pair_stats5_bwd(const float* mol, const float* pro, const int* mptr, const int* pptr,
                                                           const float* out, const int* arg, const float* d_out, int D,
                                                           float* d_mol, float* d_pro) {
    const int i = blockIdx.x, tid = threadIdx.x, lg = tid & 15, gid = tid >> 4, Q = D >> 2;
    const int m0 = mptr[i], m1 = mptr[i + 1], p0 = pptr[i], p1 = pptr[i + 1];
    const int nm = m1 - m0, np = p1 - p0;
    const bool ok0 = lg < Q, ok1 = lg + 16 < Q;
    if (nm <= 0 || np <= 0) {
        for (int k = tid; k < max(nm, 0) * D; k += kBlock) d_mol[(size_t)m0 * D + k] = 0.f;
        for (int k = tid; k < max(np, 0) * D; k += kBlock) d_pro[(size_t)p0 * D + k] = 0.f;
        return;
    }
    const float n = (float)nm * (float)np, mean = out[5 * i + 1], sd = out[5 * i + 4];
    const float gmax = d_out[5 * i], gmean = d_out[5 * i + 1] / n, gmed = d_out[5 * i + 2], gmin = d_out[5 * i + 3];
    const float cstd = d_out[5 * i + 4] / ((n - 1.f) * sd);
    const int amax = arg[6 * i], bmax = arg[6 * i + 1], amed = arg[6 * i + 2], bmed = arg[6 * i + 3], amin = arg[6 * i + 4],
              bmin = arg[6 * i + 5];
    auto sweep = [&](const float* own, int o0, int o1, const float* oth, int t0, int t1, float* d_own, bool own_is_mol) {
        for (int r = o0 + gid; r < o1; r += kP5Groups) {
            const float* orow = own + (size_t)r * D;
            const float4 x0 = ok0 ? ld4(orow + 4 * lg) : f4zero(), x1 = ok1 ? ld4(orow + 4 * (lg + 16)) : f4zero();
            float4 acc0 = f4zero(), acc1 = f4zero();
            for (int t = t0; t < t1; ++t) {
                const float* trow = oth + (size_t)t * D;
                const float4 y0 = ok0 ? ld4(trow + 4 * lg) : f4zero(), y1 = ok1 ? ld4(trow + 4 * (lg + 16)) : f4zero();
                // the same expression as the forward pass: part = dot4(mol chunk, pro chunk) (+ second chunk)
                float part = ok0 ? (own_is_mol ? dot4(x0, y0) : dot4(y0, x0)) : 0.f;
                if (ok1) part += own_is_mol ? dot4(x1, y1) : dot4(y1, x1);
                const float sc = group_sum<16>(part);
                const int a = own_is_mol ? r : t, b = own_is_mol ? t : r;
                float w = fmaf(cstd, sc - mean, gmean);
                if (a == amax && b == bmax) w += gmax;
                if (a == amed && b == bmed) w += gmed;
                if (a == amin && b == bmin) w += gmin;
                fma4(acc0, w, y0);
                fma4(acc1, w, y1);
            }
            if (ok0) st4(d_own + (size_t)r * D + 4 * lg, acc0);
            if (ok1) st4(d_own + (size_t)r * D + 4 * (lg + 16), acc1);
        }
    };
    sweep(pro, p0, p1, mol, m0, m1, d_pro, false);
    sweep(mol, m0, m1, pro, p0, p1, d_mol, true);
}

}  // namespace glam

using namespace glam;

extern "C" int glam_pair_pool5_fwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr, int64_t P,
                                   int D, float* out, int32_t* arg, void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX, "glam_pair_pool5_fwd: P out of range");
    if (D <= 0 || D > 128 || (D & 3)) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool5_fwd: D=%d must be a multiple of 4 in 4..128", D);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && out && arg && aligned16(mol) && aligned16(pro), "glam_pair_pool5_fwd: null / misaligned pointer");
    hipLaunchKernelGGL(k_pair_stats5, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, D, out, arg);
    GLAM_LAUNCH_CHECK("glam_pair_pool5_fwd");
    return GLAM_OK;
}

extern "C" int glam_pair_pool5_bwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                                   const float* out, const int32_t* arg, const float* d_out, int64_t P, int D, float* d_mol,
                                   float* d_pro, void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX, "glam_pair_pool5_bwd: P out of range");
    if (D <= 0 || D > 128 || (D & 3)) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool5_bwd: D=%d must be a multiple of 4 in 4..128", D);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && out && arg && d_out && d_mol && d_pro && aligned16(mol) && aligned16(pro) &&
                     aligned16(d_mol) && aligned16(d_pro), "glam_pair_pool5_bwd: null / misaligned pointer");
    hipLaunchKernelGGL(k_pair_stats5_bwd, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, out, arg,
                       d_out, D, d_mol, d_pro);
    GLAM_LAUNCH_CHECK("glam_pair_pool5_bwd");
    return GLAM_OK;
}

static bool pair_split(int D) { return (D & 3) == 0 && D <= 64; }

extern "C" size_t glam_pair_pool_workspace_bytes(int64_t P, int D) {
    return pair_split(D) && P > 0 ? (size_t)P * kPairSplit * kPartStride * sizeof(float) : 0;
}

extern "C" int glam_pair_pool_fwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                                  int64_t P, int D, float* out, int32_t* argmax, float* sums, void* ws, size_t ws_bytes,
                                  void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX / kPairSplit, "glam_pair_pool_fwd: P out of range");
    if (D <= 0 || D > kMaxD) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool_fwd: D=%d not in 1..%d", D, kMaxD);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && out && argmax && sums, "glam_pair_pool_fwd: null pointer");
    if (pair_split(D)) {
        GLAM_REQUIRE(ws && ws_bytes >= glam_pair_pool_workspace_bytes(P, D), "glam_pair_pool_fwd: workspace too small");
        hipLaunchKernelGGL(k_pair_max_partial, dim3((int)P * kPairSplit), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr,
                           pro_ptr, D, (float*)ws);
        hipLaunchKernelGGL(k_pair_finish, dim3((int)P), dim3(64), 0, (hipStream_t)stream, mol, mol_ptr, pro_ptr, (const float*)ws, D,
                           out, argmax, sums);
    } else {
        hipLaunchKernelGGL(k_pair_pool_fwd, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, D, out,
                           argmax);
    }
    GLAM_LAUNCH_CHECK("glam_pair_pool_fwd");
    return GLAM_OK;
}

extern "C" int glam_pair_pool_bwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                                  const int32_t* argmax, const float* sums, const float* d_out, int64_t P, int D, float* d_mol,
                                  float* d_pro, void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX / kPairSplit, "glam_pair_pool_bwd: P out of range");
    if (D <= 0 || D > kMaxD) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool_bwd: D=%d not in 1..%d", D, kMaxD);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && argmax && sums && d_out && d_mol && d_pro, "glam_pair_pool_bwd: null pointer");
    if (pair_split(D))
        hipLaunchKernelGGL(k_pair_pool_bwd_split, dim3((int)P * kPairSplit), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr,
                           pro_ptr, argmax, sums, d_out, D, d_mol, d_pro, (const float*)nullptr, (const float*)nullptr);
    else
        hipLaunchKernelGGL(k_pair_pool_bwd, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, argmax,
                           d_out, D, d_mol, d_pro);
    GLAM_LAUNCH_CHECK("glam_pair_pool_bwd");
    return GLAM_OK;
}

// ... d_mol += add_mol, d_pro += add_pro (either may be NULL): the gradient of the NEXT message step's use of the two matrices, when
// the caller took them back from this node (one add launch per tower and step less).  Only the widths of the split kernel
// (pair_split(D)); every row must belong to a pair (the segments cover both matrices).
extern "C" int glam_pair_pool_bwd_add(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                                      const int32_t* argmax, const float* sums, const float* d_out, int64_t P, int D,
                                      const float* add_mol, const float* add_pro, float* d_mol, float* d_pro, void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX / kPairSplit, "glam_pair_pool_bwd_add: P out of range");
    if (D <= 0 || D > kMaxD || !pair_split(D)) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool_bwd_add: D=%d outside the split kernel's widths", D);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && argmax && sums && d_out && d_mol && d_pro, "glam_pair_pool_bwd_add: null pointer");
    GLAM_REQUIRE(aligned16(add_mol) && aligned16(add_pro), "glam_pair_pool_bwd_add: addends must be 16-byte aligned");
    hipLaunchKernelGGL(k_pair_pool_bwd_split, dim3((int)P * kPairSplit), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr,
                       argmax, sums, d_out, D, d_mol, d_pro, add_mol, add_pro);
    GLAM_LAUNCH_CHECK("glam_pair_pool_bwd_add");
    return GLAM_OK;
}
extern "C" int glam_pair_pool_add_supported(int D) { return D > 0 && D <= kMaxD && pair_split(D) ? 1 : 0; }
