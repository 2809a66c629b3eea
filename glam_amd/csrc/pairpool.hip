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
                                                               const float* d_out, int D, float* d_mol, float* d_pro) {
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
        st4(d_pro + (size_t)b * D + 4 * c4, v);
    }
    if (sp == 0)
        for (int a = m0 + rg; a < m1; a += 16) {
            float4 v = psum;
            if (a == am) {
                const float4 t = ld4(pro + (size_t)ap * D + 4 * c4);
                v.x = fmaf(gmax, t.x, v.x); v.y = fmaf(gmax, t.y, v.y); v.z = fmaf(gmax, t.z, v.z); v.w = fmaf(gmax, t.w, v.w);
            }
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

}  // namespace glam

using namespace glam;

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
                           pro_ptr, argmax, sums, d_out, D, d_mol, d_pro);
    else
        hipLaunchKernelGGL(k_pair_pool_bwd, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, argmax,
                           d_out, D, d_mol, d_pro);
    GLAM_LAUNCH_CHECK("glam_pair_pool_bwd");
    return GLAM_OK;
}
