// The loss of the training step that drives the path, forward and gradient in one launch:
//   regression      `loss = self.criterion(output, y_true)` with nn.MSELoss            (reference src_1gp/trainer.py:296, loss.py:42)
//   classification  `self.criterion(y_score[y_true >= 0], y_true[y_true >= 0].float())` with nn.BCEWithLogitsLoss
//                   (trainer.py:244-245, loss.py:48): the mean over the labels that are present (-1 = missing, dataset.py:138)
// Through torch these are 7-10 launches of a few hundred to a few hundred thousand elements each (forward map, reduction, the
// autograd root's fill, backward maps; the masked form also needs boolean indexing, which a hipGraph cannot capture) — 25-50 us of a
// 330-830 us step.  Here: one launch writes the loss, 1 / count and the un-normalised gradient; the backward is one scale launch.
//   kind 0  l = (x - y)^2                              dl/dx = 2 (x - y)
//   kind 1  l = max(x, 0) - x y + log1p(exp(-|x|))     dl/dx = sigmoid(x) - y
//   masked  only elements with y >= 0 count (others contribute neither loss nor gradient)
// Sums run in a fixed order (thread-sequential, then a block tree, then the block partials in index order by the last block to finish):
// bit-reproducible run to run.  count = 0 gives nan, as the mean over an empty selection does in the reference.
#include "common.h"

namespace glam {

constexpr int kLossMaxBlocks = 512;

__device__ __forceinline__ void loss_elem(float x, float y, int kind, bool ok, float& l, float& g) {
    if (!ok) { l = 0.f; g = 0.f; return; }
    if (kind == 0) {
        const float d = x - y;
        l = d * d; g = 2.f * d;
    } else {
        const float e = expf(-fabsf(x));
        l = fmaxf(x, 0.f) - x * y + log1pf(e);
        const float s = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        g = s - y;
    }
}

// block sum of (a, b) in a fixed order: wave butterfly, then wave 0 adds the waves' results in wave order
__device__ __forceinline__ void block_sum2(float& a, float& b, float* s_a, float* s_b) {
    a = group_sum<64>(a); b = group_sum<64>(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_a[wave] = a; s_b[wave] = b; }
    __syncthreads();
    float ra = 0.f, rb = 0.f;
    for (int w = 0; w < kBlock / 64; ++w) { ra += s_a[w]; rb += s_b[w]; }
    a = ra; b = rb;
}

__global__ void __launch_bounds__(kBlock) k_loss_fwd(const float* pred, const float* target, int n, int kind, int masked, float* loss,
                                                    float* inv_count, float* grad, float* partial, unsigned* ticket) {
    __shared__ float s_a[kBlock / 64], s_b[kBlock / 64];
    __shared__ int s_last;
    const int tid = threadIdx.x, nb = gridDim.x;
    float sum = 0.f, cnt = 0.f;
    for (int i = blockIdx.x * kBlock + tid; i < n; i += nb * kBlock) {
        const float x = pred[i], y = target[i];
        const bool ok = !masked || y >= 0.f;
        float l, g;
        loss_elem(x, y, kind, ok, l, g);
        grad[i] = g;
        sum += l; cnt += ok ? 1.f : 0.f;
    }
    block_sum2(sum, cnt, s_a, s_b);
    if (nb == 1) {
        if (tid == 0) { loss[0] = sum / cnt; inv_count[0] = 1.f / cnt; }
        return;
    }
    if (tid == 0) {
        __hip_atomic_store(partial + 2 * blockIdx.x, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(partial + 2 * blockIdx.x + 1, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // two-level ticket (rng.h): the partials above are ordered before the check-in by the release of the fetch_add
        const unsigned g = gridDim.x, sidx = blockIdx.x & 15u;
        const unsigned in_sub = (g - sidx + 15u) >> 4, nsub = g < 16u ? g : 16u;
        unsigned* sub = ticket + 32 * (1 + sidx);
        int last = 0;
        if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == in_sub - 1) {
            __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // the last block: the block partials in index order (one wave, lane-strided, then the butterfly)
    if (tid < 64) {
        float a = 0.f, b = 0.f;
        for (int q = tid; q < nb; q += 64) {
            a += __hip_atomic_load(partial + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b += __hip_atomic_load(partial + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a = group_sum<64>(a); b = group_sum<64>(b);
        if (tid == 0) { loss[0] = a / b; inv_count[0] = 1.f / b; }
    }
}

// d_pred = grad * (g_up * inv_count): the gradient of the mean, scaled by whatever arrives at the loss (1 for loss.backward())
__global__ void __launch_bounds__(kBlock) k_loss_bwd(const float* grad, const float* inv_count, const float* g_up, int n, float* d_pred) {
    const float s = g_up[0] * inv_count[0];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) d_pred[i] = grad[i] * s;
}

}  // namespace glam

using namespace glam;

extern "C" size_t glam_loss_workspace_bytes(void) { return (size_t)2 * kLossMaxBlocks * sizeof(float); }

extern "C" int glam_loss_fwd(const float* pred, const float* target, int64_t n, int kind, int masked, float* loss, float* inv_count,
                             float* grad, void* ws, size_t ws_bytes, unsigned* ticket, void* stream) {
    GLAM_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && (kind == 0 || kind == 1), "glam_loss_fwd: bad size / kind");
    GLAM_REQUIRE(loss && inv_count && (n == 0 || (pred && target && grad)), "glam_loss_fwd: null pointer");
    int blocks = (int)((n + 4 * kBlock - 1) / (4 * kBlock));
    if (blocks < 1) blocks = 1;
    if (blocks > kLossMaxBlocks) blocks = kLossMaxBlocks;
    if (blocks > 1) GLAM_REQUIRE(ws && ws_bytes >= glam_loss_workspace_bytes() && ticket, "glam_loss_fwd: workspace / ticket missing");
    hipLaunchKernelGGL(k_loss_fwd, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, pred, target, (int)n, kind, masked, loss,
                       inv_count, grad, (float*)ws, ticket);
    GLAM_LAUNCH_CHECK("glam_loss_fwd");
    return GLAM_OK;
}

extern "C" int glam_loss_bwd(const float* grad, const float* inv_count, const float* g_up, int64_t n, float* d_pred, void* stream) {
    GLAM_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "glam_loss_bwd: bad size");
    if (n == 0) return GLAM_OK;
    GLAM_REQUIRE(grad && inv_count && g_up && d_pred, "glam_loss_bwd: null pointer");
    hipLaunchKernelGGL(k_loss_bwd, dim3(grid_for(n, kBlock, 1024)), dim3(kBlock), 0, (hipStream_t)stream, grad, inv_count, g_up, (int)n,
                       d_pred);
    GLAM_LAUNCH_CHECK("glam_loss_bwd");
    return GLAM_OK;
}
