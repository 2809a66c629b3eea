// Per-pair fusion of the two-tower (ligand + protein) models: for every pair i,
//   S_i = mol[seg_i] @ pro[seg_i]^T  ([n_mol, n_res]),  out[i] = [max(S_i), mean(S_i)]
// Reference: dot_and_global_pool2 (src_2gi_dti_scr/layer.py:270-283) — a Python loop with two .item() syncs and a
// matmul per pair.  Here: one block per pair, the ligand rows staged in LDS, each thread sweeping residues; the
// mean needs no pairwise work at all: sum(S_i) = <sum_a mol_a, sum_b pro_b>.
#include "common.h"

namespace glam {

constexpr int kMolTile = 32;      // ligand rows staged per LDS pass
constexpr int kMaxD = 256;

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

extern "C" int glam_pair_pool_fwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                                  int64_t P, int D, float* out, int32_t* argmax, void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX, "glam_pair_pool_fwd: P out of range");
    if (D <= 0 || D > kMaxD) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool_fwd: D=%d not in 1..%d", D, kMaxD);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && out && argmax, "glam_pair_pool_fwd: null pointer");
    hipLaunchKernelGGL(k_pair_pool_fwd, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, D, out, argmax);
    GLAM_LAUNCH_CHECK("glam_pair_pool_fwd");
    return GLAM_OK;
}

extern "C" int glam_pair_pool_bwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                                  const int32_t* argmax, const float* d_out, int64_t P, int D, float* d_mol, float* d_pro,
                                  void* stream) {
    GLAM_REQUIRE(P >= 0 && P < INT32_MAX, "glam_pair_pool_bwd: P out of range");
    if (D <= 0 || D > kMaxD) return fail(GLAM_E_UNSUPPORTED, "glam_pair_pool_bwd: D=%d not in 1..%d", D, kMaxD);
    if (P == 0) return GLAM_OK;
    GLAM_REQUIRE(mol && pro && mol_ptr && pro_ptr && argmax && d_out && d_mol && d_pro, "glam_pair_pool_bwd: null pointer");
    hipLaunchKernelGGL(k_pair_pool_bwd, dim3((int)P), dim3(kBlock), 0, (hipStream_t)stream, mol, pro, mol_ptr, pro_ptr, argmax,
                       d_out, D, d_mol, d_pro);
    GLAM_LAUNCH_CHECK("glam_pair_pool_bwd");
    return GLAM_OK;
}
