// Per-graph normalisations over contiguous node segments (one wavefront per graph, lanes over channels).
// Reference semantics replaced (PyG 1.7.2 classes wrapped at src_1gp/layer.py:161-194):
//   mode 0  PairNorm(scale=1, eps=1e-5)(x, batch):   x' = x - mean_n x (per channel);
//           y = scale * x' / sqrt(eps + mean_n ||x'_n||^2)
//   mode 1  graph LayerNorm(eps=1e-5)(x, batch) without the affine part:  x' = x - mean_{n,c} x;
//           y = x' / sqrt(mean_{n,c} x'^2 + eps)        (the per-channel weight/bias stay in the host mirror)
// Statistics are recomputed in the backward pass from x (saved), so nothing but x is kept.
#include "common.h"
#include "rng.h"

namespace glam {

constexpr int kWavesPerBlockN = kBlock / 64;
constexpr int kMaxChunks = 4;   // D <= 256 channels

// sum over the nodes of one graph of column c: 4 loads in flight (graphs are 10-30 nodes, the loop is latency bound)
__device__ __forceinline__ float colsum(const float* x, int beg, int end, int D, int c) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int n = beg;
    for (; n + 4 <= end; n += 4) {
        const float v0 = x[(size_t)n * D + c], v1 = x[(size_t)(n + 1) * D + c];
        const float v2 = x[(size_t)(n + 2) * D + c], v3 = x[(size_t)(n + 3) * D + c];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; n < end; ++n) s0 += x[(size_t)n * D + c];
    return (s0 + s1) + (s2 + s3);
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_graph_norm_fwd(const float* x, const int* ptr, int B, int D, float scale,
                                                          float eps, float* y) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlockN + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlockN;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        if (end <= beg) continue;
        const float inv_cnt = 1.f / (float)(end - beg);
        float mean[kMaxChunks];
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int c = lane + 64 * k;
            mean[k] = c < D ? colsum(x, beg, end, D, c) * inv_cnt : 0.f;
        }
        if (MODE == 1) {   // scalar mean over nodes x channels
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxChunks; ++k) t += (lane + 64 * k < D) ? mean[k] : 0.f;
            t = group_sum<64>(t) / (float)D;
#pragma unroll
            for (int k = 0; k < kMaxChunks; ++k) mean[k] = t;
        }
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
#pragma unroll 4
                for (int n = beg; n < end; ++n) { const float d = x[(size_t)n * D + c] - mean[k]; sq = fmaf(d, d, sq); }
            }
        }
        sq = group_sum<64>(sq) * inv_cnt;
        const float a = MODE == 0 ? scale / sqrtf(eps + sq) : 1.f / sqrtf(sq / (float)D + eps);
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
#pragma unroll 4
                for (int n = beg; n < end; ++n) y[(size_t)n * D + c] = (x[(size_t)n * D + c] - mean[k]) * a;
            }
        }
    }
}

// d_x = a (g - gbar) - a^3 T x' / M      with x' the centred input, a the scale above,
//   PairNorm : gbar = per-channel mean of g over the graph's nodes, T = sum <g, x'>, M = cnt, a = scale/sqrt(eps+s)
//              (d_x = a (g - gbar) - (a^3 / scale^2) T x' / cnt)
//   LayerNorm: gbar = scalar mean of g, T = sum g x', M = cnt*D
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_graph_norm_bwd(const float* x, const float* gy, const int* ptr, int B, int D,
                                                          float scale, float eps, float* dx, const float* addend) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlockN + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlockN;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        if (end <= beg) continue;
        const float inv_cnt = 1.f / (float)(end - beg);
        float mean[kMaxChunks], gbar[kMaxChunks];
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int c = lane + 64 * k;
            mean[k] = c < D ? colsum(x, beg, end, D, c) * inv_cnt : 0.f;
            gbar[k] = c < D ? colsum(gy, beg, end, D, c) * inv_cnt : 0.f;
        }
        if (MODE == 1) {
            float s = 0.f, t = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxChunks; ++k)
                if (lane + 64 * k < D) { s += mean[k]; t += gbar[k]; }
            s = group_sum<64>(s) / (float)D;
            t = group_sum<64>(t) / (float)D;
#pragma unroll
            for (int k = 0; k < kMaxChunks; ++k) { mean[k] = s; gbar[k] = t; }
        }
        float sq = 0.f, T = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
#pragma unroll 4
                for (int n = beg; n < end; ++n) {
                    const float d = x[(size_t)n * D + c] - mean[k];
                    sq = fmaf(d, d, sq);
                    T = fmaf(gy[(size_t)n * D + c], d, T);
                }
            }
        }
        sq = group_sum<64>(sq) * inv_cnt;
        T = group_sum<64>(T);
        float a, coef;
        if (MODE == 0) { a = scale / sqrtf(eps + sq); coef = a * a * a / (scale * scale) * T * inv_cnt; }
        else { a = 1.f / sqrtf(sq / (float)D + eps); coef = a * a * a * T * inv_cnt / (float)D; }
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
#pragma unroll 4
                for (int n = beg; n < end; ++n) {
                    const size_t i = (size_t)n * D + c;
                    const float v = a * (gy[i] - gbar[k]) - coef * (x[i] - mean[k]);
                    dx[i] = addend ? v + addend[i] : v;
                }
            }
        }
    }
}


// ---- wave-per-graph, D % 4 == 0 and D <= 64 (molecule-sized graphs at the padded hidden widths) ----
// The kernels above are three chains of n/4 dependent round trips per graph (10 / 14 us for 20-atom molecules at any
// batch size).  Here a lane owns (row group rg = lane / 16, float4 chunk c4 = lane % 16) and keeps its rows in registers:
// one round of loads per 32 nodes and per pass, the first pass usually being the only one that leaves the CU.
constexpr int kRowsPerLane = 8;     // 4 row groups x 8 = 32 nodes per register pass

__device__ __forceinline__ float4 rg_sum(float4 v) {      // over the 4 row groups of a wave (lanes l, l^16, l^32, l^48)
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    return v;
}

// DROP: the training-mode Dropout(p) that follows the norm in a MessageBlock (src_1gp/layer.py:255-256) from the same launch: y_drop =
// y * mask / (1 - p) on the device-side Philox stream (element i draws word i % 4 of philox4(i / 4), like every RNG kernel of the
// library); y itself is only stored when somebody wants it.
struct NormDrop { long long* state; long long* eff; float p; float* y_drop; };
struct NormDropB { const long long* eff; float p; const float* gy_drop; };

template <int MODE, bool DROP>
__global__ void __launch_bounds__(kBlock) k_graph_norm_fwd_v4(const float* x, const int* ptr, int B, int D, float scale,
                                                             float eps, float* y, NormDrop nd) {
    Philox ph{};
    if constexpr (DROP) ph = rng_begin(nd.state, nd.eff);
    const int lane = threadIdx.x & 63, c4 = lane & 15, rg = lane >> 4;
    const int wave = blockIdx.x * kWavesPerBlockN + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlockN;
    const bool act = 4 * c4 < D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        if (end <= beg) continue;
        const float inv_cnt = 1.f / (float)(end - beg);
        const bool small = end - beg <= 4 * kRowsPerLane;       // wave-uniform: the graph fits the register pass
        float4 row[kRowsPerLane];
        float4 acc = f4zero();
        for (int b0 = beg; b0 < end; b0 += 4 * kRowsPerLane) {
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                const int n = b0 + rg + 4 * u;
                row[u] = (act && n < end) ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) { acc.x += row[u].x; acc.y += row[u].y; acc.z += row[u].z; acc.w += row[u].w; }
        }
        float4 mean = inv_cnt * rg_sum(acc);
        if (MODE == 1) {
            const float t = group_sum<16>(act ? (mean.x + mean.y) + (mean.z + mean.w) : 0.f) / (float)D;
            mean = make_float4(t, t, t, t);
        }
        float sq = 0.f;
        for (int b0 = beg; b0 < end; b0 += 4 * kRowsPerLane) {
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                const int n = b0 + rg + 4 * u;
                if (!small) row[u] = (act && n < end) ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
                if (act && n < end) {
                    const float d0 = row[u].x - mean.x, d1 = row[u].y - mean.y, d2 = row[u].z - mean.z, d3 = row[u].w - mean.w;
                    sq = fmaf(d0, d0, sq); sq = fmaf(d1, d1, sq); sq = fmaf(d2, d2, sq); sq = fmaf(d3, d3, sq);
                }
            }
        }
        sq = group_sum<64>(sq) * inv_cnt;
        const float a = MODE == 0 ? scale / sqrtf(eps + sq) : 1.f / sqrtf(sq / (float)D + eps);
        for (int b0 = beg; b0 < end; b0 += 4 * kRowsPerLane) {
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                const int n = b0 + rg + 4 * u;
                if (!small) row[u] = (act && n < end) ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
                if (act && n < end) {
                    const size_t off = (size_t)n * D + 4 * c4;
                    const float4 v = make_float4((row[u].x - mean.x) * a, (row[u].y - mean.y) * a, (row[u].z - mean.z) * a, (row[u].w - mean.w) * a);
                    if (!DROP || y) st4(y + off, v);
                    if constexpr (DROP) {
                        const uint4 w4 = philox4(ph, off >> 2);
                        st4(nd.y_drop + off, make_float4(v.x * drop_scale_w(w4.x, nd.p), v.y * drop_scale_w(w4.y, nd.p),
                                                         v.z * drop_scale_w(w4.z, nd.p), v.w * drop_scale_w(w4.w, nd.p)));
                    }
                }
            }
        }
    }
    if constexpr (DROP) rng_end(nd.state, ph);
}

template <int MODE, bool DROP>
__global__ void __launch_bounds__(kBlock) k_graph_norm_bwd_v4(const float* x, const float* gy, const int* ptr, int B, int D,
                                                             float scale, float eps, float* dx, const float* addend, NormDropB nd) {
    Philox ph{};
    if constexpr (DROP) ph = philox_init(nd.eff);
    // the gradient of y: gy, plus (DROP) the dropped twin's gradient through the regenerated mask (gy may then be null)
    auto gload = [&](size_t off) -> float4 {
        float4 g = (!DROP || gy) ? ld4(gy + off) : f4zero();
        if constexpr (DROP) {
            const float4 gd = ld4(nd.gy_drop + off);
            const uint4 w4 = philox4(ph, off >> 2);
            g.x = fmaf(gd.x, drop_scale_w(w4.x, nd.p), g.x); g.y = fmaf(gd.y, drop_scale_w(w4.y, nd.p), g.y);
            g.z = fmaf(gd.z, drop_scale_w(w4.z, nd.p), g.z); g.w = fmaf(gd.w, drop_scale_w(w4.w, nd.p), g.w);
        }
        return g;
    };
    const int lane = threadIdx.x & 63, c4 = lane & 15, rg = lane >> 4;
    const int wave = blockIdx.x * kWavesPerBlockN + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlockN;
    const bool act = 4 * c4 < D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        if (end <= beg) continue;
        const float inv_cnt = 1.f / (float)(end - beg);
        const bool small = end - beg <= 4 * kRowsPerLane;
        float4 xr[kRowsPerLane], gr[kRowsPerLane];
        float4 ax = f4zero(), ag = f4zero();
        for (int b0 = beg; b0 < end; b0 += 4 * kRowsPerLane) {
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                const int n = b0 + rg + 4 * u;
                const bool okr = act && n < end;
                xr[u] = okr ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
                gr[u] = okr ? gload((size_t)n * D + 4 * c4) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                ax.x += xr[u].x; ax.y += xr[u].y; ax.z += xr[u].z; ax.w += xr[u].w;
                ag.x += gr[u].x; ag.y += gr[u].y; ag.z += gr[u].z; ag.w += gr[u].w;
            }
        }
        float4 mean = inv_cnt * rg_sum(ax), gbar = inv_cnt * rg_sum(ag);
        if (MODE == 1) {
            const float sm = group_sum<16>(act ? (mean.x + mean.y) + (mean.z + mean.w) : 0.f) / (float)D;
            const float tg = group_sum<16>(act ? (gbar.x + gbar.y) + (gbar.z + gbar.w) : 0.f) / (float)D;
            mean = make_float4(sm, sm, sm, sm); gbar = make_float4(tg, tg, tg, tg);
        }
        float sq = 0.f, T = 0.f;
        for (int b0 = beg; b0 < end; b0 += 4 * kRowsPerLane) {
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                const int n = b0 + rg + 4 * u;
                const bool okr = act && n < end;
                if (!small) {
                    xr[u] = okr ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
                    gr[u] = okr ? gload((size_t)n * D + 4 * c4) : f4zero();
                }
                if (okr) {
                    const float d0 = xr[u].x - mean.x, d1 = xr[u].y - mean.y, d2 = xr[u].z - mean.z, d3 = xr[u].w - mean.w;
                    sq = fmaf(d0, d0, sq); sq = fmaf(d1, d1, sq); sq = fmaf(d2, d2, sq); sq = fmaf(d3, d3, sq);
                    T = fmaf(gr[u].x, d0, T); T = fmaf(gr[u].y, d1, T); T = fmaf(gr[u].z, d2, T); T = fmaf(gr[u].w, d3, T);
                }
            }
        }
        sq = group_sum<64>(sq) * inv_cnt;
        T = group_sum<64>(T);
        float a, coef;
        if (MODE == 0) { a = scale / sqrtf(eps + sq); coef = a * a * a / (scale * scale) * T * inv_cnt; }
        else { a = 1.f / sqrtf(sq / (float)D + eps); coef = a * a * a * T * inv_cnt / (float)D; }
        for (int b0 = beg; b0 < end; b0 += 4 * kRowsPerLane) {
#pragma unroll
            for (int u = 0; u < kRowsPerLane; ++u) {
                const int n = b0 + rg + 4 * u;
                const bool okr = act && n < end;
                if (!small) {
                    xr[u] = okr ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
                    gr[u] = okr ? gload((size_t)n * D + 4 * c4) : f4zero();
                }
                if (okr) {
                    float4 v = make_float4(a * (gr[u].x - gbar.x) - coef * (xr[u].x - mean.x), a * (gr[u].y - gbar.y) - coef * (xr[u].y - mean.y),
                                           a * (gr[u].z - gbar.z) - coef * (xr[u].z - mean.z), a * (gr[u].w - gbar.w) - coef * (xr[u].w - mean.w));
                    if (addend) {      // a second gradient path into x (the block's residual): summed here instead of by an add launch
                        const float4 ad = ld4(addend + (size_t)n * D + 4 * c4);
                        v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                    }
                    st4(dx + (size_t)n * D + 4 * c4, v);
                }
            }
        }
    }
}

// ---- block-per-graph variants for graphs of hundreds of nodes (proteins): the kernels above walk a graph's nodes serially
//      per lane; here 256 threads = 16 row groups x 16 float4 channel chunks (D % 4 == 0, D <= 64) ----
__device__ __forceinline__ float block_sum256(float v, float* s_red) {      // fixed-order tree; returns the total to every thread
    const int tid = threadIdx.x;
    s_red[tid] = v;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    const float r = s_red[0];
    __syncthreads();
    return r;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_graph_norm_fwd_block(const float* x, const int* ptr, int B, int D, float scale,
                                                                float eps, float* y) {
    __shared__ __attribute__((aligned(16))) float s_part[16 * 16 * 4];
    __shared__ __attribute__((aligned(16))) float s_mean[64];
    __shared__ float s_red[kBlock];
    const int tid = threadIdx.x, c4 = tid & 15, rg = tid >> 4;
    const bool act = 4 * c4 < D;
    for (int g = blockIdx.x; g < B; g += gridDim.x) {
        const int beg = ptr[g], end = ptr[g + 1];
        if (end <= beg) continue;
        const float inv_cnt = 1.f / (float)(end - beg);
        block_colsum(x, beg, end, D, s_part, s_mean);
        float4 mean = f4zero();
        if (MODE == 1) {
            float t = 0.f;
            for (int c = 0; c < D; ++c) t += s_mean[c];
            t = t * inv_cnt / (float)D;
            mean = make_float4(t, t, t, t);
        } else if (act) {
            mean = inv_cnt * ld4(s_mean + 4 * c4);
        }
        float sq = 0.f;
        if (act)
            for (int n = beg + rg; n < end; n += 16) {
                const float4 v = ld4(x + (size_t)n * D + 4 * c4);
                const float dx0 = v.x - mean.x, dx1 = v.y - mean.y, dx2 = v.z - mean.z, dx3 = v.w - mean.w;
                sq = fmaf(dx0, dx0, sq); sq = fmaf(dx1, dx1, sq); sq = fmaf(dx2, dx2, sq); sq = fmaf(dx3, dx3, sq);
            }
        sq = block_sum256(sq, s_red) * inv_cnt;
        const float a = MODE == 0 ? scale / sqrtf(eps + sq) : 1.f / sqrtf(sq / (float)D + eps);
        if (act)
            for (int n = beg + rg; n < end; n += 16) {
                const float4 v = ld4(x + (size_t)n * D + 4 * c4);
                st4(y + (size_t)n * D + 4 * c4, make_float4((v.x - mean.x) * a, (v.y - mean.y) * a, (v.z - mean.z) * a, (v.w - mean.w) * a));
            }
        __syncthreads();
    }
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_graph_norm_bwd_block(const float* x, const float* gy, const int* ptr, int B, int D,
                                                                float scale, float eps, float* dx, const float* addend) {
    __shared__ __attribute__((aligned(16))) float s_part[16 * 16 * 4];
    __shared__ __attribute__((aligned(16))) float s_mean[64];
    __shared__ __attribute__((aligned(16))) float s_gbar[64];
    __shared__ float s_red[kBlock];
    const int tid = threadIdx.x, c4 = tid & 15, rg = tid >> 4;
    const bool act = 4 * c4 < D;
    for (int g = blockIdx.x; g < B; g += gridDim.x) {
        const int beg = ptr[g], end = ptr[g + 1];
        if (end <= beg) continue;
        const float inv_cnt = 1.f / (float)(end - beg);
        block_colsum(x, beg, end, D, s_part, s_mean);
        block_colsum(gy, beg, end, D, s_part, s_gbar);
        float4 mean = f4zero(), gbar = f4zero();
        if (MODE == 1) {
            float sm = 0.f, tg = 0.f;
            for (int c = 0; c < D; ++c) { sm += s_mean[c]; tg += s_gbar[c]; }
            sm = sm * inv_cnt / (float)D; tg = tg * inv_cnt / (float)D;
            mean = make_float4(sm, sm, sm, sm); gbar = make_float4(tg, tg, tg, tg);
        } else if (act) {
            mean = inv_cnt * ld4(s_mean + 4 * c4);
            gbar = inv_cnt * ld4(s_gbar + 4 * c4);
        }
        float sq = 0.f, T = 0.f;
        if (act)
            for (int n = beg + rg; n < end; n += 16) {
                const float4 v = ld4(x + (size_t)n * D + 4 * c4), gv = ld4(gy + (size_t)n * D + 4 * c4);
                const float d0 = v.x - mean.x, d1 = v.y - mean.y, d2 = v.z - mean.z, d3 = v.w - mean.w;
                sq = fmaf(d0, d0, sq); sq = fmaf(d1, d1, sq); sq = fmaf(d2, d2, sq); sq = fmaf(d3, d3, sq);
                T = fmaf(gv.x, d0, T); T = fmaf(gv.y, d1, T); T = fmaf(gv.z, d2, T); T = fmaf(gv.w, d3, T);
            }
        sq = block_sum256(sq, s_red) * inv_cnt;
        T = block_sum256(T, s_red);
        float a, coef;
        if (MODE == 0) { a = scale / sqrtf(eps + sq); coef = a * a * a / (scale * scale) * T * inv_cnt; }
        else { a = 1.f / sqrtf(sq / (float)D + eps); coef = a * a * a * T * inv_cnt / (float)D; }
        if (act)
            for (int n = beg + rg; n < end; n += 16) {
                const float4 v = ld4(x + (size_t)n * D + 4 * c4), gv = ld4(gy + (size_t)n * D + 4 * c4);
                float4 o = make_float4(a * (gv.x - gbar.x) - coef * (v.x - mean.x), a * (gv.y - gbar.y) - coef * (v.y - mean.y),
                                       a * (gv.z - gbar.z) - coef * (v.z - mean.z), a * (gv.w - gbar.w) - coef * (v.w - mean.w));
                if (addend) {
                    const float4 ad = ld4(addend + (size_t)n * D + 4 * c4);
                    o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                }
                st4(dx + (size_t)n * D + 4 * c4, o);
            }
        __syncthreads();
    }
}

}  // namespace glam

using namespace glam;

static int norm_dims(const char* fn, int64_t N, int64_t B, int D, int mode) {
    if (N < 0 || B < 0 || N >= INT32_MAX || B >= INT32_MAX) return fail(GLAM_E_INVALID, "%s: N/B out of range", fn);
    if (D <= 0 || D > 64 * kMaxChunks) return fail(GLAM_E_UNSUPPORTED, "%s: D=%d not in 1..%d", fn, D, 64 * kMaxChunks);
    if (mode != 0 && mode != 1) return fail(GLAM_E_INVALID, "%s: mode=%d", fn, mode);
    return GLAM_OK;
}

extern "C" int glam_graph_norm_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int mode, float scale,
                                   float eps, float* y, void* stream) {
    if (int rc = norm_dims("glam_graph_norm_fwd", N, B, D, mode)) return rc;
    if (N == 0 || B == 0) return GLAM_OK;
    GLAM_REQUIRE(x && ptr && y, "glam_graph_norm_fwd: null pointer");
    const dim3 block(kBlock);
    if (N / B >= 64 && (D & 3) == 0 && D <= 64) {       // large graphs: a block per graph
        const dim3 grid(grid_for(B, 1));
        if (mode == 0) hipLaunchKernelGGL(k_graph_norm_fwd_block<0>, grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y);
        else hipLaunchKernelGGL(k_graph_norm_fwd_block<1>, grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y);
        GLAM_LAUNCH_CHECK("glam_graph_norm_fwd");
        return GLAM_OK;
    }
    const dim3 grid(grid_for(B, kWavesPerBlockN));
    if ((D & 3) == 0 && D <= 64) {
        if (mode == 0) hipLaunchKernelGGL((k_graph_norm_fwd_v4<0, false>), grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y, NormDrop{});
        else hipLaunchKernelGGL((k_graph_norm_fwd_v4<1, false>), grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y, NormDrop{});
    } else if (mode == 0) hipLaunchKernelGGL(k_graph_norm_fwd<0>, grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y);
    else hipLaunchKernelGGL(k_graph_norm_fwd<1>, grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y);
    GLAM_LAUNCH_CHECK("glam_graph_norm_fwd");
    return GLAM_OK;
}

static int graph_norm_bwd_impl(const float* x, const float* gy, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                               float scale, float eps, const float* addend, float* dx, void* stream) {
    if (int rc = norm_dims("glam_graph_norm_bwd", N, B, D, mode)) return rc;
    if (N == 0 || B == 0) return GLAM_OK;
    GLAM_REQUIRE(x && gy && ptr && dx, "glam_graph_norm_bwd: null pointer");
    GLAM_REQUIRE(!addend || (D & 3) || aligned16(addend), "glam_graph_norm_bwd_add: addend must be 16-byte aligned");
    const dim3 block(kBlock);
    if (N / B >= 64 && (D & 3) == 0 && D <= 64) {
        const dim3 grid(grid_for(B, 1));
        if (mode == 0) hipLaunchKernelGGL(k_graph_norm_bwd_block<0>, grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend);
        else hipLaunchKernelGGL(k_graph_norm_bwd_block<1>, grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend);
        GLAM_LAUNCH_CHECK("glam_graph_norm_bwd");
        return GLAM_OK;
    }
    const dim3 grid(grid_for(B, kWavesPerBlockN));
    if ((D & 3) == 0 && D <= 64) {
        if (mode == 0) hipLaunchKernelGGL((k_graph_norm_bwd_v4<0, false>), grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend, NormDropB{});
        else hipLaunchKernelGGL((k_graph_norm_bwd_v4<1, false>), grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend, NormDropB{});
    } else if (mode == 0) hipLaunchKernelGGL(k_graph_norm_bwd<0>, grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend);
    else hipLaunchKernelGGL(k_graph_norm_bwd<1>, grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend);
    GLAM_LAUNCH_CHECK("glam_graph_norm_bwd");
    return GLAM_OK;
}

extern "C" int glam_graph_norm_bwd(const float* x, const float* gy, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                                   float scale, float eps, float* dx, void* stream) {
    return graph_norm_bwd_impl(x, gy, ptr, N, B, D, mode, scale, eps, nullptr, dx, stream);
}

// dx = the normalisation's input gradient + addend (f32[N, D]): a second gradient path into the same x — the residual of a MessageBlock,
// src_1gp/layer.py:253-265 — summed in the store instead of by an add launch.  Every node must belong to a graph (ptr[B] = N).
extern "C" int glam_graph_norm_bwd_add(const float* x, const float* gy, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                                       float scale, float eps, const float* addend, float* dx, void* stream) {
    GLAM_REQUIRE(addend, "glam_graph_norm_bwd_add: null addend");
    return graph_norm_bwd_impl(x, gy, ptr, N, B, D, mode, scale, eps, addend, dx, stream);
}

// ---- the norm + the Dropout(p) behind it (MessageBlock: x = norm(x); x = dropout(x); src_1gp/layer.py:255-256) in one launch each way ----
extern "C" int glam_graph_norm_drop_supported(int64_t N, int64_t B, int D) {
    return B > 0 && N > 0 && N / B < 64 && (D & 3) == 0 && D <= 64;       // the lane-per-row kernels (molecule-sized graphs)
}

extern "C" int glam_graph_norm_drop_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int mode, float scale, float eps,
                                        float drop_p, int64_t* rng_state, int64_t* rng_eff, float* y, float* y_drop, void* stream) {
    if (int rc = norm_dims("glam_graph_norm_drop_fwd", N, B, D, mode)) return rc;
    GLAM_REQUIRE(drop_p > 0.f && drop_p < 1.f, "glam_graph_norm_drop_fwd: dropout p = %g outside (0, 1)", drop_p);
    if (!glam_graph_norm_drop_supported(N, B, D))
        return fail(GLAM_E_UNSUPPORTED, "glam_graph_norm_drop_fwd: N=%lld B=%lld D=%d outside the fused kernels (D %% 4 == 0, D <= 64, < 64 nodes per graph)",
                    (long long)N, (long long)B, D);
    GLAM_REQUIRE(x && ptr && y_drop && rng_state && rng_eff && aligned16(x) && aligned16(y) && aligned16(y_drop),
                 "glam_graph_norm_drop_fwd: null / misaligned pointer");
    const dim3 block(kBlock), grid(grid_for(B, kWavesPerBlockN));
    NormDrop nd{reinterpret_cast<long long*>(rng_state), reinterpret_cast<long long*>(rng_eff), drop_p, y_drop};
    if (mode == 0) hipLaunchKernelGGL((k_graph_norm_fwd_v4<0, true>), grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y, nd);
    else hipLaunchKernelGGL((k_graph_norm_fwd_v4<1, true>), grid, block, 0, (hipStream_t)stream, x, ptr, (int)B, D, scale, eps, y, nd);
    GLAM_LAUNCH_CHECK("glam_graph_norm_drop_fwd");
    return GLAM_OK;
}

extern "C" int glam_graph_norm_drop_bwd(const float* x, const float* gy, const float* gy_drop, const int32_t* ptr, int64_t N, int64_t B, int D,
                                        int mode, float scale, float eps, float drop_p, const int64_t* rng_eff, const float* addend, float* dx,
                                        void* stream) {
    if (int rc = norm_dims("glam_graph_norm_drop_bwd", N, B, D, mode)) return rc;
    GLAM_REQUIRE(drop_p > 0.f && drop_p < 1.f, "glam_graph_norm_drop_bwd: dropout p = %g outside (0, 1)", drop_p);
    if (!glam_graph_norm_drop_supported(N, B, D))
        return fail(GLAM_E_UNSUPPORTED, "glam_graph_norm_drop_bwd: N=%lld B=%lld D=%d outside the fused kernels", (long long)N, (long long)B, D);
    GLAM_REQUIRE(x && gy_drop && ptr && dx && rng_eff && aligned16(x) && aligned16(gy) && aligned16(gy_drop) && aligned16(addend) && aligned16(dx),
                 "glam_graph_norm_drop_bwd: null / misaligned pointer");
    const dim3 block(kBlock), grid(grid_for(B, kWavesPerBlockN));
    NormDropB nd{reinterpret_cast<const long long*>(rng_eff), drop_p, gy_drop};
    if (mode == 0) hipLaunchKernelGGL((k_graph_norm_bwd_v4<0, true>), grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend, nd);
    else hipLaunchKernelGGL((k_graph_norm_bwd_v4<1, true>), grid, block, 0, (hipStream_t)stream, x, gy, ptr, (int)B, D, scale, eps, dx, addend, nd);
    GLAM_LAUNCH_CHECK("glam_graph_norm_drop_bwd");
    return GLAM_OK;
}

