// The relation weights of NNConv with one-hot bond features (reference: src_1gp/layer.py:115-122, `nn = Linear(De, 32) -> ReLU ->
// Linear(32, C*C)` inside PyG's NNConv): nn(e_ij) takes only De distinct values, nn(eye(De)), a parameter-only [De, C*C] table.
// Through torch that table and its backward were a dozen launches on 4-row operands (addmm / relu / mm with K = 3600 on four rows:
// 19 us for one of them) — ~50 us of a 780 us training step.  Here: one launch forward, two backward.
//   forward : h[k, j] = relu(w1[j, k] + b1[j]);  out[k, m] = b2[m] + sum_j h[k, j] w2[m, j]
//   backward: d_w2[m, j] = sum_k d_out[k, m] h[k, j];  d_b2[m] = sum_k d_out[k, m];
//             d_h[k, j] = sum_m d_out[k, m] w2[m, j]  (block partials, summed in block order by the second launch)
//             d_pre = d_h * (h > 0);  d_w1[j, k] = d_pre[k, j];  d_b1[j] = sum_k d_pre[k, j]
#include "common.h"

namespace glam {

// Every thread owns four consecutive hidden units of one output row m (one 16-byte piece of w2, read and written coalesced); the
// hidden/4 lanes of a row sit side by side in a wavefront.  All of a thread's global loads are issued before the first use: the
// launches are tiny (w2 is 460 KB at C = 60), so their duration is the number of dependent memory round trips, not bytes.
constexpr int kRelMaxDe = 8, kRelMaxHd = 64;
constexpr int kRelBwdBlock = 1024;

__device__ __forceinline__ float f4dot(float4 a, float4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

template <int DE>
__global__ void __launch_bounds__(kBlock) k_relmlp_fwd(const float* w1, const float* b1, const float* w2, const float* b2, int Hd, int M,
                                                       float* h_out, float* out) {
    const int L = Hd >> 2;                                     // lanes per row (a power of two, 1..16)
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const bool live = gid < M * L;
    const int m = live ? gid / L : 0, jq = gid & (L - 1);
    const float4 w = live ? ld4(w2 + (size_t)gid * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bb = b2[m];
    float w1r[4 * DE];                                         // w1[4 jq .. 4 jq + 3, 0 .. De): 4 De consecutive floats
#pragma unroll
    for (int i = 0; i < 4 * DE; ++i) w1r[i] = w1[(size_t)jq * 4 * DE + i];
    const float4 bv = ld4(b1 + 4 * jq);
    float acc[DE];
#pragma unroll
    for (int k = 0; k < DE; ++k) {
        const float4 h = make_float4(fmaxf(w1r[k] + bv.x, 0.f), fmaxf(w1r[DE + k] + bv.y, 0.f), fmaxf(w1r[2 * DE + k] + bv.z, 0.f),
                                     fmaxf(w1r[3 * DE + k] + bv.w, 0.f));
        if (gid < L) st4(h_out + k * Hd + 4 * jq, h);
        acc[k] = f4dot(h, w);
    }
    for (int off = 1; off < L; off <<= 1)
#pragma unroll
        for (int k = 0; k < DE; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
    if (live && jq == 0)
#pragma unroll
        for (int k = 0; k < DE; ++k) out[(size_t)k * M + m] = acc[k] + bb;
}

// d_w2 rows, d_b2, and this block's partial of d_h [De * Hd] (rows in a fixed order: lanes by xor tree, then wavefronts in order)
template <int DE>
__global__ void __launch_bounds__(kRelBwdBlock) k_relmlp_bwd1(const float* d_out, const float* h, const float* w2, int Hd, int M, float* d_w2,
                                                              float* d_b2, float* part) {
    __shared__ float s_part[kRelBwdBlock / 64][kRelMaxDe * kRelMaxHd];
    const int L = Hd >> 2;
    const int gid = blockIdx.x * kRelBwdBlock + threadIdx.x;
    const bool live = gid < M * L;
    const int m = live ? gid / L : 0, jq = gid & (L - 1);
    const float4 w = live ? ld4(w2 + (size_t)gid * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float dv[DE];
    float4 hv[DE];
#pragma unroll
    for (int k = 0; k < DE; ++k) {
        dv[k] = live ? d_out[(size_t)k * M + m] : 0.f;
        hv[k] = ld4(h + k * Hd + 4 * jq);
    }
    float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
    float db = 0.f;
#pragma unroll
    for (int k = 0; k < DE; ++k) {
        dw.x = fmaf(dv[k], hv[k].x, dw.x), dw.y = fmaf(dv[k], hv[k].y, dw.y), dw.z = fmaf(dv[k], hv[k].z, dw.z), dw.w = fmaf(dv[k], hv[k].w, dw.w);
        db += dv[k];
    }
    if (live) {
        st4(d_w2 + (size_t)gid * 4, dw);
        if (jq == 0) d_b2[m] = db;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < DE; ++k) {
        float4 p = make_float4(dv[k] * w.x, dv[k] * w.y, dv[k] * w.z, dv[k] * w.w);
        for (int off = L; off < 64; off <<= 1)
            p.x += __shfl_xor(p.x, off, 64), p.y += __shfl_xor(p.y, off, 64), p.z += __shfl_xor(p.z, off, 64), p.w += __shfl_xor(p.w, off, 64);
        if (lane < L) *reinterpret_cast<float4*>(&s_part[wave][k * Hd + 4 * lane]) = p;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DE * Hd; i += kRelBwdBlock) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < kRelBwdBlock / 64; ++wv) v += s_part[wv][i];
        part[(size_t)blockIdx.x * DE * Hd + i] = v;
    }
}

// d_pre = (sum of the block partials, in block order) * (h > 0);  d_w1 = d_pre^T;  d_b1 = its column sums
__global__ void __launch_bounds__(kBlock) k_relmlp_bwd2(const float* part, int nblk, const float* h, int De, int Hd, float* d_w1, float* d_b1) {
    __shared__ float s_dp[kRelMaxDe * kRelMaxHd];
    const int n = De * Hd;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        float v = 0.f;
        int b = 0;
        for (; b + 8 <= nblk; b += 8) {                        // eight loads in flight, added in block order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(b + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; b < nblk; ++b) v += part[(size_t)b * n + i];
        const float dp = h[i] > 0.f ? v : 0.f;
        s_dp[i] = dp;
        const int k = i / Hd, j = i - k * Hd;
        d_w1[j * De + k] = dp;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Hd; j += kBlock) {
        float v = 0.f;
        for (int k = 0; k < De; ++k) v += s_dp[k * Hd + j];
        d_b1[j] = v;
    }
}

static bool relmlp_shape_ok(int De, int Hd, int64_t M) {
    return De >= 1 && De <= kRelMaxDe && Hd >= 4 && Hd <= kRelMaxHd && (Hd & (Hd - 1)) == 0 && M >= 1 && M <= (1 << 22);
}

template <int DE>
static void relmlp_launch_fwd(const float* w1, const float* b1, const float* w2, const float* b2, int Hd, int M, float* h, float* out, hipStream_t s) {
    const int64_t threads = (int64_t)M * (Hd >> 2);
    hipLaunchKernelGGL(k_relmlp_fwd<DE>, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, w1, b1, w2, b2, Hd, M, h, out);
}

template <int DE>
static void relmlp_launch_bwd1(const float* d_out, const float* h, const float* w2, int Hd, int M, float* d_w2, float* d_b2, float* part, int nblk,
                               hipStream_t s) {
    hipLaunchKernelGGL(k_relmlp_bwd1<DE>, dim3(nblk), dim3(kRelBwdBlock), 0, s, d_out, h, w2, Hd, M, d_w2, d_b2, part);
}

static int relmlp_bwd_blocks(int Hd, int64_t M) { return (int)((M * (Hd >> 2) + kRelBwdBlock - 1) / kRelBwdBlock); }

}  // namespace glam

using namespace glam;

#define GLAM_REL_DISPATCH(De, CALL)                                                                                                           \
    switch (De) {                                                                                                                             \
        case 1: { constexpr int DE = 1; CALL; } break;                                                                                        \
        case 2: { constexpr int DE = 2; CALL; } break;                                                                                        \
        case 3: { constexpr int DE = 3; CALL; } break;                                                                                        \
        case 4: { constexpr int DE = 4; CALL; } break;                                                                                        \
        case 5: { constexpr int DE = 5; CALL; } break;                                                                                        \
        case 6: { constexpr int DE = 6; CALL; } break;                                                                                        \
        case 7: { constexpr int DE = 7; CALL; } break;                                                                                        \
        default: { constexpr int DE = 8; CALL; } break;                                                                                       \
    }

extern "C" int glam_relation_mlp_supported(int De, int Hd, int64_t M) { return relmlp_shape_ok(De, Hd, M) ? 1 : 0; }

extern "C" size_t glam_relation_mlp_workspace_bytes(int De, int Hd, int64_t M) {
    if (!relmlp_shape_ok(De, Hd, M)) return 0;
    return (size_t)relmlp_bwd_blocks(Hd, M) * De * Hd * sizeof(float);
}

extern "C" int glam_relation_mlp_fwd(const float* w1, const float* b1, const float* w2, const float* b2, int De, int Hd, int64_t M, float* h,
                                     float* out, void* stream) {
    GLAM_REQUIRE(w1 && b1 && w2 && b2 && h && out, "glam_relation_mlp_fwd: null pointer");
    if (!relmlp_shape_ok(De, Hd, M))
        return fail(GLAM_E_UNSUPPORTED, "glam_relation_mlp_fwd: De=%d (<= 8), hidden=%d (a power of two in 4..64), M=%lld outside the kernel", De,
                    Hd, (long long)M);
    GLAM_REQUIRE(aligned16(w2) && aligned16(b1) && aligned16(h), "glam_relation_mlp_fwd: w2, b1 and h must be 16-byte aligned");
    GLAM_REL_DISPATCH(De, relmlp_launch_fwd<DE>(w1, b1, w2, b2, Hd, (int)M, h, out, (hipStream_t)stream));
    GLAM_LAUNCH_CHECK("glam_relation_mlp_fwd");
    return GLAM_OK;
}

extern "C" int glam_relation_mlp_bwd(const float* d_out, const float* h, const float* w2, int De, int Hd, int64_t M, float* d_w1, float* d_b1,
                                     float* d_w2, float* d_b2, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(d_out && h && w2 && d_w1 && d_b1 && d_w2 && d_b2 && ws, "glam_relation_mlp_bwd: null pointer");
    if (!relmlp_shape_ok(De, Hd, M))
        return fail(GLAM_E_UNSUPPORTED, "glam_relation_mlp_bwd: De=%d (<= 8), hidden=%d (a power of two in 4..64), M=%lld outside the kernel", De,
                    Hd, (long long)M);
    GLAM_REQUIRE(ws_bytes >= glam_relation_mlp_workspace_bytes(De, Hd, M), "glam_relation_mlp_bwd: workspace too small");
    GLAM_REQUIRE(aligned16(w2) && aligned16(d_w2) && aligned16(h), "glam_relation_mlp_bwd: w2, d_w2 and h must be 16-byte aligned");
    const int nblk = relmlp_bwd_blocks(Hd, M);
    hipStream_t s = (hipStream_t)stream;
    GLAM_REL_DISPATCH(De, relmlp_launch_bwd1<DE>(d_out, h, w2, Hd, (int)M, d_w2, d_b2, reinterpret_cast<float*>(ws), nblk, s));
    hipLaunchKernelGGL(k_relmlp_bwd2, dim3(1), dim3(kBlock), 0, s, reinterpret_cast<const float*>(ws), nblk, h, De, Hd, d_w1, d_b1);
    GLAM_LAUNCH_CHECK("glam_relation_mlp_bwd");
    return GLAM_OK;
}
