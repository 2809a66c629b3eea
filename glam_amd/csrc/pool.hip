// Graph readouts over contiguous node segments (one wavefront per graph, lanes over channels).
// Reference semantics replaced: PyG global_mean_pool / global_add_pool / global_max_pool /
// global_sort_pool(k=3) behind GlobalPool5 (src_1gp/layer.py:197-203), GlobalAttention behind
// GlobalLAPool (src_1gp/layer.py:206-220) and the attention step of Set2Set (src_1gp/model.py:41).
// Segment sums run in node order inside one lane => deterministic, no atomics.
#include "common.h"

namespace glam {

constexpr int kMaxK = 8;
constexpr int kWavesPerBlock = kBlock / 64;

// ---- GlobalPool5: mean | add | top-k rows by last channel ---------------------------------------
__global__ void __launch_bounds__(kBlock) k_pool5_fwd(const float* x, const int* ptr, int B, int D, int K, float* out,
                                                     int* topk_idx) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int OD = (2 + K) * D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        // top-K on the last channel; every lane keeps the same list (broadcast loads)
        float tv[kMaxK];
        int ti[kMaxK];
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) { tv[r] = -INFINITY; ti[r] = -1; }
#pragma unroll 4
        for (int n = beg; n < end; ++n) {
            float v = x[(size_t)n * D + (D - 1)];
            int vi = n;
            bool shifting = false;   // once inserted, the displaced entries just move down one slot
#pragma unroll
            for (int r = 0; r < kMaxK; ++r) {
                if (r < K) {
                    // strict '>' keeps the lower node index first on ties (stable descending sort);
                    // an empty slot (ti < 0) always accepts
                    const bool take = shifting || ti[r] < 0 || v > tv[r];
                    if (take && vi >= 0) {
                        const float ov = tv[r]; const int oi = ti[r];
                        tv[r] = v; ti[r] = vi; v = ov; vi = oi;
                        shifting = true;
                    }
                }
            }
        }
        const float inv_cnt = 1.f / (float)max(end - beg, 1);
        for (int c = lane; c < D; c += 64) {
            float s = 0.f;
#pragma unroll 4
            for (int n = beg; n < end; ++n) s += x[(size_t)n * D + c];
            float* o = out + (size_t)g * OD;
            o[c] = s * inv_cnt;
            o[D + c] = s;
#pragma unroll
            for (int r = 0; r < kMaxK; ++r)
                if (r < K) o[(2 + r) * D + c] = ti[r] >= 0 ? x[(size_t)ti[r] * D + c] : 0.f;
        }
        if (lane < K) {
            int sel = -1;
#pragma unroll
            for (int r = 0; r < kMaxK; ++r) if (r == lane) sel = ti[r];
            topk_idx[(size_t)g * K + lane] = sel;
        }
    }
}

// Wave-per-graph over float4 rows (every hidden width in its padded form; the readout of the default model): the
// kernel above is a chain of ~3 n/4 dependent round trips per graph (22 us for 20-atom molecules at ANY batch size).
// Here a lane owns (row group rg = lane / LPR, float4 column chunk c4 = lane % LPR), LPR = 16 lanes per row (ld <= 64) or 32
// (ld <= 128): all row loads of a graph are in flight together; the top-K rows come from per-lane candidates merged by K
// rounds of a wave-wide arg-max (value descending, node index ascending on ties: the stable order of the serial kernel).
// Rows are ld floats apart (ld % 4 == 0) and hold D <= ld channels, the rest zero padding (hid_dim 15 / 30 / 45 / 90 arrive as
// 16 / 32 / 48 / 92); the output is the compact [B, (2 + K) * D] the reference produces.
__device__ __forceinline__ void put_cols(float* o, int c0, int D, float4 v) {
    if ((D & 3) == 0) { st4(o + c0, v); return; }       // (c0 < ld = D rounded up: c0 + 3 < D when D % 4 == 0)
    if (c0 < D) o[c0] = v.x;
    if (c0 + 1 < D) o[c0 + 1] = v.y;
    if (c0 + 2 < D) o[c0 + 2] = v.z;
    if (c0 + 3 < D) o[c0 + 3] = v.w;
}
__device__ __forceinline__ float4 get_cols(const float* o, int c0, int D, bool vec) {
    if (vec) return ld4(o + c0);
    return make_float4(c0 < D ? o[c0] : 0.f, c0 + 1 < D ? o[c0 + 1] : 0.f, c0 + 2 < D ? o[c0 + 2] : 0.f, c0 + 3 < D ? o[c0 + 3] : 0.f);
}

template <int LPR>
__global__ void __launch_bounds__(kBlock) k_pool5_fwd_v4(const float* x, const int* ptr, int B, int ld, int D, int K, float* out,
                                                        int* topk_idx) {
    constexpr int RG = 64 / LPR;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, rg = lane / LPR;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int OD = (2 + K) * D;
    const bool act = 4 * c4 < ld;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        // candidates: lane l holds the top-K of nodes beg + l, beg + l + 64, ...
        float tv[kMaxK];
        int ti[kMaxK];
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) { tv[r] = -INFINITY; ti[r] = 0x7fffffff; }
        for (int n = beg + lane; n < end; n += 64) {
            float v = x[(size_t)n * ld + (D - 1)];
            int vi = n;
#pragma unroll
            for (int r = 0; r < kMaxK; ++r) {
                if (r < K) {
                    const bool take = v > tv[r] || (v == tv[r] && vi < ti[r]);
                    if (take) { const float ov = tv[r]; const int oi = ti[r]; tv[r] = v; ti[r] = vi; v = ov; vi = oi; }
                }
            }
        }
        // column sums: 8 rows of this lane's row group in flight per step, the last step masked — every load unconditional (clamped
        // row, zeroed afterwards: a load under a condition is waited for on the spot), so a molecule of up to 8 RG nodes is ONE round trip
        // (the row-at-a-time tail this replaces was a chain of five for 20 atoms)
        float4 acc = f4zero();
        if (act) {
            for (int n = beg + rg; n < end; n += 8 * RG) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld4(x + (size_t)min(n + RG * u, end - 1) * ld + 4 * c4);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (n + RG * u < end) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
#pragma unroll
        for (int off = LPR; off <= 32; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        float* o = out + (size_t)g * OD;
        const float inv_cnt = 1.f / (float)max(end - beg, 1);
        if (act && rg == 0) {
            put_cols(o, 4 * c4, D, inv_cnt * acc);
            put_cols(o + D, 4 * c4, D, acc);
        }
        // K rounds of wave-wide arg-max over the lanes' best remaining candidate
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) {
            if (r < K) {
                float bv = tv[0];
                int bi = ti[0];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(bv, off);
                    const int oi = __shfl_xor(bi, off);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                const int sel = bi == 0x7fffffff ? -1 : bi;
                if (sel >= 0 && ti[0] == bi) {           // the owner pops its list
#pragma unroll
                    for (int q = 0; q + 1 < kMaxK; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
                    tv[kMaxK - 1] = -INFINITY; ti[kMaxK - 1] = 0x7fffffff;
                }
                if (act && rg == (r & (RG - 1)))
                    put_cols(o + (2 + r) * D, 4 * c4, D, sel >= 0 ? ld4(x + (size_t)sel * ld + 4 * c4) : f4zero());
                if (lane == 0) topk_idx[(size_t)g * K + r] = sel;
            }
        }
    }
}

// ---- GlobalAttention read (reference: GlobalLAPool, src_1gp/layer.py:206-220) for D % 4 == 0, D <= 128: a lane owns
//      (row group, float4 chunk) with LPR = 16 or 32 lanes per row; all row loads of a pass in flight (the kernels above
//      are chains of n dependent round trips per graph: 25 / 29 us for 20-atom molecules). ----
template <int LPR>
__global__ void __launch_bounds__(kBlock) k_segment_attn_fwd_v4(const float* gate, const float* v, const int* ptr, int B, int D,
                                                               float* out, float* stats) {
    constexpr int RG = 64 / LPR, R = 8;                 // row groups per wave, rows per lane and pass
    const int lane = threadIdx.x & 63, c4 = lane % LPR, rg = lane / LPR;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const bool act = 4 * c4 < D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const bool small = end - beg <= RG * R;
        float4 row[R];
        float e[R];
        float m = -INFINITY;
        for (int b0 = beg; b0 < end; b0 += RG * R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int n = b0 + rg + RG * u;
                const bool okr = n < end;
                e[u] = okr ? gate[n] : -INFINITY;
                row[u] = (act && okr) ? ld4(v + (size_t)n * D + 4 * c4) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < R; ++u) m = fmaxf(m, e[u]);
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (end <= beg) m = 0.f;
        float ssum = 0.f;
        float4 acc = f4zero();
        for (int b0 = beg; b0 < end; b0 += RG * R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int n = b0 + rg + RG * u;
                const bool okr = n < end;
                if (!small) {
                    e[u] = okr ? gate[n] : -INFINITY;
                    row[u] = (act && okr) ? ld4(v + (size_t)n * D + 4 * c4) : f4zero();
                }
                if (okr) {
                    const float p = expf(e[u] - m);
                    ssum += p;
                    fma4(acc, p, row[u]);
                }
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            ssum += __shfl_xor(ssum, off);
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        const float inv = 1.f / (ssum + 1e-16f);
        if (act && rg == 0) st4(out + (size_t)g * D + 4 * c4, inv * acc);
        if (lane == 0) { stats[2 * g] = m; stats[2 * g + 1] = ssum; }
    }
}

template <int LPR>
__global__ void __launch_bounds__(kBlock) k_segment_attn_bwd_v4(const float* gate, const float* v, const float* out,
                                                               const float* stats, const float* d_out, const int* ptr, int B,
                                                               int D, float* d_gate, float* d_v) {
    constexpr int RG = 64 / LPR, R = 8;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, rg = lane / LPR;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const bool act = 4 * c4 < D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float m = stats[2 * g], inv = 1.f / (stats[2 * g + 1] + 1e-16f);
        const float4 gv = act ? ld4(d_out + (size_t)g * D + 4 * c4) : f4zero();
        const float4 ov = act ? ld4(out + (size_t)g * D + 4 * c4) : f4zero();
        const float dot_out = group_sum<LPR>(dot4(gv, ov));
        for (int b0 = beg; b0 < end; b0 += RG * R) {
            float4 row[R];
            float e[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int n = b0 + rg + RG * u;
                const bool okr = n < end;
                e[u] = okr ? gate[n] : 0.f;
                row[u] = (act && okr) ? ld4(v + (size_t)n * D + 4 * c4) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int n = b0 + rg + RG * u;
                const float dv = group_sum<LPR>(dot4(gv, row[u]));
                if (n < end) {
                    const float a = expf(e[u] - m) * inv;
                    if (act) st4(d_v + (size_t)n * D + 4 * c4, a * gv);
                    if (c4 == 0) d_gate[n] = a * (dv - dot_out);
                }
            }
        }
    }
}

// ---- Set2Set attention read (PyG Set2Set.forward: e = <x_n, q_g>, a = softmax over the graph, r_g = sum_n a_n x_n) with
//      the logits computed in the kernel from the per-graph query q[B, D]: no [N, D] gather of q, no [N] logit tensor.
//      Wave per graph, lane = (row group rg, float4 chunk c4), D % 4 == 0 and D <= 64; 32 nodes per register pass. ----
constexpr int kS2sRows = 8;

template <int LPR>     // lanes per row: 16 (D <= 64) or 32 (D <= 128)
__global__ void __launch_bounds__(kBlock) k_s2s_attn_fwd(const float* x, const float* q, const int* ptr, int B, int D, float* r,
                                                        float* stats) {
    constexpr int RG = 64 / LPR;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, rg = lane / LPR;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const bool act = 4 * c4 < D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float4 qv = act ? ld4(q + (size_t)g * D + 4 * c4) : f4zero();
        const bool small = end - beg <= RG * kS2sRows;
        float4 row[kS2sRows];
        float e[kS2sRows];
        float m = -INFINITY;
        for (int b0 = beg; b0 < end; b0 += RG * kS2sRows) {
#pragma unroll
            for (int u = 0; u < kS2sRows; ++u) {
                const int n = b0 + rg + RG * u;
                row[u] = (act && n < end) ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < kS2sRows; ++u) {
                e[u] = group_sum<LPR>(dot4(row[u], qv));
                if (b0 + rg + RG * u < end) m = fmaxf(m, e[u]);
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (end <= beg) m = 0.f;
        float ssum = 0.f;
        float4 acc = f4zero();
        for (int b0 = beg; b0 < end; b0 += RG * kS2sRows) {
#pragma unroll
            for (int u = 0; u < kS2sRows; ++u) {
                const int n = b0 + rg + RG * u;
                if (!small) {
                    row[u] = (act && n < end) ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
                    e[u] = group_sum<LPR>(dot4(row[u], qv));
                }
                if (n < end) {
                    const float p = expf(e[u] - m);
                    ssum += p;
                    fma4(acc, p, row[u]);
                }
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            ssum += __shfl_xor(ssum, off);
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        const float inv = 1.f / (ssum + 1e-16f);
        if (act && rg == 0) st4(r + (size_t)g * D + 4 * c4, inv * acc);
        if (lane == 0) { stats[2 * g] = m; stats[2 * g + 1] = ssum; }
    }
}

// d_x_n = a_n d_r + de_n q,  d_q = sum_n de_n x_n,  de_n = a_n (<d_r, x_n> - <d_r, r>)
template <int LPR>
__global__ void __launch_bounds__(kBlock) k_s2s_attn_bwd(const float* x, const float* q, const float* r, const float* stats,
                                                        const float* d_r, const int* ptr, int B, int D, float* d_x, float* d_q) {
    constexpr int RG = 64 / LPR;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, rg = lane / LPR;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const bool act = 4 * c4 < D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float4 qv = act ? ld4(q + (size_t)g * D + 4 * c4) : f4zero();
        const float4 gv = act ? ld4(d_r + (size_t)g * D + 4 * c4) : f4zero();
        const float4 rv = act ? ld4(r + (size_t)g * D + 4 * c4) : f4zero();
        const float m = stats[2 * g], inv = 1.f / (stats[2 * g + 1] + 1e-16f);
        const float dot_out = group_sum<LPR>(dot4(gv, rv));
        float4 dq = f4zero();
        for (int b0 = beg; b0 < end; b0 += RG * kS2sRows) {
            float4 row[kS2sRows];
#pragma unroll
            for (int u = 0; u < kS2sRows; ++u) {
                const int n = b0 + rg + RG * u;
                row[u] = (act && n < end) ? ld4(x + (size_t)n * D + 4 * c4) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < kS2sRows; ++u) {
                const int n = b0 + rg + RG * u;
                const float e = group_sum<LPR>(dot4(row[u], qv));
                const float da = group_sum<LPR>(dot4(row[u], gv));
                if (n < end) {
                    const float a = expf(e - m) * inv;
                    const float de = a * (da - dot_out);
                    fma4(dq, de, row[u]);
                    if (act) {
                        float4 v = a * gv;
                        fma4(v, de, qv);
                        st4(d_x + (size_t)n * D + 4 * c4, v);
                    }
                }
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            dq.x += __shfl_xor(dq.x, off); dq.y += __shfl_xor(dq.y, off);
            dq.z += __shfl_xor(dq.z, off); dq.w += __shfl_xor(dq.w, off);
        }
        if (act && rg == 0) st4(d_q + (size_t)g * D + 4 * c4, dq);
    }
}

// Block-per-graph variant for graphs of hundreds of nodes (proteins): the wave-per-graph kernel above walks a graph's
// nodes serially, fine for 20 atoms, a chain of 500 dependent round trips for 500 residues (0.55 ms per call with only
// B = 32 waves on the chip).  256 threads split the rows; D % 4 == 0 and D <= 64.
__global__ void __launch_bounds__(kBlock) k_pool5_fwd_block(const float* x, const int* ptr, int B, int D, int K, float* out,
                                                           int* topk_idx) {
    __shared__ __attribute__((aligned(16))) float s_part[16 * 16 * 4];
    __shared__ float s_sum[64];
    __shared__ float s_val[kBlock];
    __shared__ int s_idx[kBlock];
    __shared__ int s_own[kBlock];
    __shared__ int s_top[kMaxK];
    const int tid = threadIdx.x;
    const int OD = (2 + K) * D;
    for (int g = blockIdx.x; g < B; g += gridDim.x) {
        const int beg = ptr[g], end = ptr[g + 1];
        block_colsum(x, beg, end, D, s_part, s_sum);
        // thread-local top-K of the last channel over rows beg + tid, beg + tid + 256, ... (ascending: stable on ties)
        float tv[kMaxK];
        int ti[kMaxK];
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) { tv[r] = -INFINITY; ti[r] = -1; }
        for (int n = beg + tid; n < end; n += kBlock) {
            float v = x[(size_t)n * D + (D - 1)];
            int vi = n;
            bool shifting = false;
#pragma unroll
            for (int r = 0; r < kMaxK; ++r) {
                if (r < K) {
                    const bool take = shifting || ti[r] < 0 || v > tv[r];
                    if (take && vi >= 0) {
                        const float ov = tv[r]; const int oi = ti[r];
                        tv[r] = v; ti[r] = vi; v = ov; vi = oi;
                        shifting = true;
                    }
                }
            }
        }
        // K rounds of a block-wide argmax over the heads of the local lists (value descending, node index ascending)
        int head = 0;
        for (int round = 0; round < K; ++round) {
            float cv = -INFINITY;
            int ci = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < kMaxK; ++r)
                if (r == head && r < K && ti[r] >= 0) { cv = tv[r]; ci = ti[r]; }
            s_val[tid] = cv; s_idx[tid] = ci; s_own[tid] = tid;
            __syncthreads();
            for (int o = kBlock / 2; o > 0; o >>= 1) {
                if (tid < o) {
                    const float v = s_val[tid + o];
                    const int ix = s_idx[tid + o];
                    if (ix != 0x7fffffff && (s_idx[tid] == 0x7fffffff || v > s_val[tid] || (v == s_val[tid] && ix < s_idx[tid]))) {
                        s_val[tid] = v; s_idx[tid] = ix; s_own[tid] = s_own[tid + o];
                    }
                }
                __syncthreads();
            }
            if (tid == 0) s_top[round] = s_idx[0] == 0x7fffffff ? -1 : s_idx[0];
            if (s_idx[0] != 0x7fffffff && s_own[0] == tid) ++head;
            __syncthreads();
        }
        const float inv_cnt = 1.f / (float)max(end - beg, 1);
        float* o = out + (size_t)g * OD;
        for (int e = tid; e < OD; e += kBlock) {
            const int part = e / D, c = e - part * D;
            float v;
            if (part == 0) v = s_sum[c] * inv_cnt;
            else if (part == 1) v = s_sum[c];
            else v = s_top[part - 2] >= 0 ? x[(size_t)s_top[part - 2] * D + c] : 0.f;
            o[e] = v;
        }
        if (tid < K) topk_idx[(size_t)g * K + tid] = s_top[tid];
        __syncthreads();
    }
}

// grid = (graph, row chunk): a protein's 500 x 60 gradient block is spread over kPool5BwdChunks blocks, rows by 16-row
// groups and channels by float4 (D % 4 == 0, D <= 64), no per-element division
constexpr int kPool5BwdChunks = 8;
__global__ void __launch_bounds__(kBlock) k_pool5_bwd_block(const float* d_out, const int* ptr, const int* topk_idx, int B,
                                                           int D, int K, float* d_x) {
    const int tid = threadIdx.x, c4 = tid & 15, rl = tid >> 4;
    const int OD = (2 + K) * D;
    const int g = blockIdx.x / kPool5BwdChunks, chunk = blockIdx.x % kPool5BwdChunks;
    if (g >= B || 4 * c4 >= D) return;
    const int beg = ptr[g], end = ptr[g + 1];
    const float inv_cnt = 1.f / (float)max(end - beg, 1);
    int ti[kMaxK];
#pragma unroll
    for (int r = 0; r < kMaxK; ++r) ti[r] = r < K ? topk_idx[(size_t)g * K + r] : -1;
    const float* go = d_out + (size_t)g * OD;
    const float4 gm = ld4(go + 4 * c4), ga = ld4(go + D + 4 * c4);
    const float4 base = make_float4(gm.x * inv_cnt + ga.x, gm.y * inv_cnt + ga.y, gm.z * inv_cnt + ga.z, gm.w * inv_cnt + ga.w);
    for (int n = beg + chunk * 16 + rl; n < end; n += kPool5BwdChunks * 16) {
        float4 v = base;
#pragma unroll
        for (int r = 0; r < kMaxK; ++r)
            if (r < K && ti[r] == n) {
                const float4 t = ld4(go + (2 + r) * D + 4 * c4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
        st4(d_x + (size_t)n * D + 4 * c4, v);
    }
}

__global__ void __launch_bounds__(kBlock) k_pool5_bwd(const float* d_out, const int* ptr, const int* topk_idx, int B,
                                                     int D, int K, float* d_x) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int OD = (2 + K) * D;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float inv_cnt = 1.f / (float)max(end - beg, 1);
        int ti[kMaxK];
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) ti[r] = r < K ? topk_idx[(size_t)g * K + r] : -1;
        const float* go = d_out + (size_t)g * OD;
        for (int c = lane; c < D; c += 64) {
            const float base = go[c] * inv_cnt + go[D + c];
            for (int n = beg; n < end; ++n) {
                float v = base;
#pragma unroll
                for (int r = 0; r < kMaxK; ++r)
                    if (r < K && ti[r] == n) v += go[(2 + r) * D + c];
                d_x[(size_t)n * D + c] = v;
            }
        }
    }
}

// float4 rows of ld floats (D channels + zero padding), 64 / LPR rows of a graph per wave instruction; d_out is the compact
// [B, (2 + K) * D] (vec: D % 4 == 0 and 16-byte aligned, else scalar reads); the pad columns of d_x are written as zeros
template <int LPR>
__global__ void __launch_bounds__(kBlock) k_pool5_bwd_v4(const float* d_out, const int* ptr, const int* topk_idx, int B, int ld, int D,
                                                        int K, int vec, float* d_x) {
    constexpr int RG = 64 / LPR;
    const int lane = threadIdx.x & 63, c4 = lane % LPR, rg = lane / LPR;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int OD = (2 + K) * D;
    if (4 * c4 >= ld) return;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float inv_cnt = 1.f / (float)max(end - beg, 1);
        const float* go = d_out + (size_t)g * OD;
        const float4 gm = get_cols(go, 4 * c4, D, vec), ga = get_cols(go + D, 4 * c4, D, vec);
        int ti[kMaxK];
        float4 gk[kMaxK];
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) {
            ti[r] = r < K ? topk_idx[(size_t)g * K + r] : -1;
            gk[r] = r < K ? get_cols(go + (2 + r) * D, 4 * c4, D, vec) : f4zero();
        }
        float4 base = ga;
        fma4(base, inv_cnt, gm);
        for (int n = beg + rg; n < end; n += RG) {
            float4 v = base;
#pragma unroll
            for (int r = 0; r < kMaxK; ++r)
                if (r < K && ti[r] == n) { v.x += gk[r].x; v.y += gk[r].y; v.z += gk[r].z; v.w += gk[r].w; }
            st4(d_x + (size_t)n * ld + 4 * c4, v);
        }
    }
}

// ---- scatter(x, batch, reduce = sum | mean | max) over sorted batch -----------------------------
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_segment_pool_fwd(const float* x, const int* ptr, int B, int D, float* out,
                                                            int* argmax) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        for (int c = lane; c < D; c += 64) {
            if (MODE == 2) {
                float best = 0.f;
                int bi = -1;
                for (int n = beg; n < end; ++n) {
                    const float v = x[(size_t)n * D + c];
                    if (bi < 0 || v > best) { best = v; bi = n; }
                }
                out[(size_t)g * D + c] = best;   // empty segment -> 0 (torch_scatter convention)
                argmax[(size_t)g * D + c] = bi;
            } else {
                float s = 0.f;
                for (int n = beg; n < end; ++n) s += x[(size_t)n * D + c];
                if (MODE == 1) s *= 1.f / (float)max(end - beg, 1);
                out[(size_t)g * D + c] = s;
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_segment_pool_bwd(const float* d_out, const int* ptr, const int* argmax,
                                                            int B, int D, float* d_x) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float scale = MODE == 1 ? 1.f / (float)max(end - beg, 1) : 1.f;
        for (int c = lane; c < D; c += 64) {
            const float go = d_out[(size_t)g * D + c] * scale;
            const int am = MODE == 2 ? argmax[(size_t)g * D + c] : -1;
            for (int n = beg; n < end; ++n) d_x[(size_t)n * D + c] = (MODE == 2) ? (n == am ? go : 0.f) : go;
        }
    }
}

// ---- segment-softmax attention readout -----------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_segment_attn_fwd(const float* gate, const float* v, const int* ptr, int B,
                                                            int D, float* out, float* stats) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        float m = -INFINITY;
        for (int n = beg + lane; n < end; n += 64) m = fmaxf(m, gate[n]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (end <= beg) m = 0.f;
        float ssum = 0.f;   // identical on every lane (sequential node order)
        for (int n = beg; n < end; ++n) ssum += expf(gate[n] - m);
        const float inv = 1.f / (ssum + 1e-16f);
        for (int c = lane; c < D; c += 64) {
            float acc = 0.f;
            for (int n = beg; n < end; ++n) acc = fmaf(expf(gate[n] - m), v[(size_t)n * D + c], acc);
            out[(size_t)g * D + c] = acc * inv;
        }
        if (lane == 0) { stats[2 * g] = m; stats[2 * g + 1] = ssum; }
    }
}

__global__ void __launch_bounds__(kBlock) k_segment_attn_bwd(const float* gate, const float* v, const float* out,
                                                            const float* stats, const float* d_out, const int* ptr,
                                                            int B, int D, float* d_gate, float* d_v) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int g = wave; g < B; g += nwaves) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float m = stats[2 * g], inv = 1.f / (stats[2 * g + 1] + 1e-16f);
        float part = 0.f;
        for (int c = lane; c < D; c += 64) part = fmaf(d_out[(size_t)g * D + c], out[(size_t)g * D + c], part);
        const float dot_out = group_sum<64>(part);
        for (int n = beg; n < end; ++n) {
            const float a = expf(gate[n] - m) * inv;
            float p = 0.f;
            for (int c = lane; c < D; c += 64) {
                const float go = d_out[(size_t)g * D + c];
                p = fmaf(go, v[(size_t)n * D + c], p);
                d_v[(size_t)n * D + c] = a * go;
            }
            const float dv = group_sum<64>(p);
            if (lane == 0) d_gate[n] = a * (dv - dot_out);
        }
    }
}

// ---- generic edge -> node reduction over CSR segments (one thread per output element) ------------
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_edge_reduce_fwd(const float* msg, const int* rowptr, const int* eid, int N,
                                                           int D, float* out, int* argmax) {
    const size_t total = (size_t)N * D;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int n = (int)(i / D), c = (int)(i % D);
        const int beg = rowptr[n], end = rowptr[n + 1];
        if (MODE == 2) {
            float best = 0.f;
            int bi = -1;
            for (int e = beg; e < end; ++e) {
                const int id = eid[e];
                const float v = msg[(size_t)id * D + c];
                if (bi < 0 || v > best) { best = v; bi = id; }
            }
            out[i] = best;
            argmax[i] = bi;
        } else {
            float s = 0.f;
            for (int e = beg; e < end; ++e) s += msg[(size_t)eid[e] * D + c];
            if (MODE == 1) s *= 1.f / (float)max(end - beg, 1);
            out[i] = s;
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_edge_reduce_bwd(const float* d_out, const int* rowptr, const int* eid,
                                                           const int* argmax, int N, int D, float* d_msg) {
    const size_t total = (size_t)N * D;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int n = (int)(i / D), c = (int)(i % D);
        const int beg = rowptr[n], end = rowptr[n + 1];
        const float go = d_out[i] * (MODE == 1 ? 1.f / (float)max(end - beg, 1) : 1.f);
        const int am = MODE == 2 ? argmax[i] : -1;
        for (int e = beg; e < end; ++e) {
            const int id = eid[e];
            d_msg[(size_t)id * D + c] = (MODE == 2) ? (id == am ? go : 0.f) : go;
        }
    }
}

// ---- edge-weighted neighbour sums: S[n,k,:] = (1/deg_n) sum_{e -> n} w[e,k] * x[src_e,:] -----------------------
// With one-hot edge features this is the per-relation neighbour sum that turns NNConv (per-edge [C,C] weights
// nn(e_ij), src_1gp/layer.py:115-122) into a 4-relation R-GCN: out = S.view(N, K*C) @ stack_k nn(onehot_k).
template <int K>
__global__ void __launch_bounds__(kBlock) k_edge_wsum_fwd(const float* x, const float* w, const int* rowptr,
                                                         const int* nbr, const int* eid, int N, int D, int mean,
                                                         float* out) {
    const size_t total = (size_t)N * D;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int n = (int)(i / D), c = (int)(i % D);
        const int beg = rowptr[n], end = rowptr[n + 1];
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
        for (int e = beg; e < end; ++e) {
            const float xv = x[(size_t)nbr[e] * D + c];
            const float* we = w + (size_t)eid[e] * K;
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = fmaf(we[k], xv, acc[k]);
        }
        const float sc = mean ? 1.f / (float)max(end - beg, 1) : 1.f;
#pragma unroll
        for (int k = 0; k < K; ++k) out[((size_t)n * K + k) * D + c] = acc[k] * sc;
    }
}

// d_x[j,:] = sum_{e: src = j} (1/deg_dst) sum_k w[e,k] * d_S[dst_e,k,:]   (walks the CSR transpose)
template <int K>
__global__ void __launch_bounds__(kBlock) k_edge_wsum_bwd(const float* d_out, const float* w, const int* colptr,
                                                         const int* dst, const int* eid_t, const int* rowptr, int N,
                                                         int D, int mean, float* dx) {
    const size_t total = (size_t)N * D;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int j = (int)(i / D), c = (int)(i % D);
        float acc = 0.f;
        for (int e = colptr[j]; e < colptr[j + 1]; ++e) {
            const int n = dst[e];
            const float sc = mean ? 1.f / (float)max(rowptr[n + 1] - rowptr[n], 1) : 1.f;
            const float* we = w + (size_t)eid_t[e] * K;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) t = fmaf(we[k], d_out[((size_t)n * K + k) * D + c], t);
            acc = fmaf(sc, t, acc);
        }
        dx[i] = acc;
    }
}


// K in {4, 8}, D % 4 == 0 (NNConv's relation sums at the padded hidden widths): a thread per (node, float4 of channels),
// the K weights of an edge as float4 loads, two edges in flight; same edge order and fma sequence as the scalar kernels.
template <int K>
__global__ void __launch_bounds__(kBlock) k_edge_wsum_fwd_v4(const float* x, const float* w, const int* rowptr, const int* nbr,
                                                            const int* eid, int N, int D, int mean, int self_slot, float* out) {
    const int KS = K + self_slot;     // self_slot: slot K of every node is its own row x[n] (NNConv's root term as one more relation)
    const int D4 = D >> 2;
    const size_t total = (size_t)N * D4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int n = (int)(i / D4), c = 4 * (int)(i % D4);
        const int beg = rowptr[n], end = rowptr[n + 1];
        float4 acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = f4zero();
        int e = beg;
        for (; e + 1 < end; e += 2) {
            const int s0 = nbr[e], s1 = nbr[e + 1], i0 = eid[e], i1 = eid[e + 1];
            float4 w0[K / 4], w1[K / 4];
#pragma unroll
            for (int u = 0; u < K / 4; ++u) { w0[u] = ld4(w + (size_t)i0 * K + 4 * u); w1[u] = ld4(w + (size_t)i1 * K + 4 * u); }
            const float4 v0 = ld4(x + (size_t)s0 * D + c), v1 = ld4(x + (size_t)s1 * D + c);
#pragma unroll
            for (int k = 0; k < K; ++k) fma4(acc[k], f4get(w0[k >> 2], k & 3), v0);
#pragma unroll
            for (int k = 0; k < K; ++k) fma4(acc[k], f4get(w1[k >> 2], k & 3), v1);
        }
        if (e < end) {
            const int i0 = eid[e];
            const float4 v0 = ld4(x + (size_t)nbr[e] * D + c);
#pragma unroll
            for (int k = 0; k < K; ++k) fma4(acc[k], w[(size_t)i0 * K + k], v0);
        }
        const float sc = mean ? 1.f / (float)max(end - beg, 1) : 1.f;
#pragma unroll
        for (int k = 0; k < K; ++k) st4(out + ((size_t)n * KS + k) * D + c, sc * acc[k]);
        if (self_slot) st4(out + ((size_t)n * KS + K) * D + c, ld4(x + (size_t)n * D + c));
    }
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_edge_wsum_bwd_v4(const float* d_out, const float* w, const int* colptr, const int* dst,
                                                            const int* eid_t, const int* rowptr, int N, int D, int mean,
                                                            int self_slot, float* dx, const float* addend) {
    const int D4 = D >> 2, KS = K + self_slot;
    const size_t total = (size_t)N * D4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int j = (int)(i / D4), c = 4 * (int)(i % D4);
        float4 acc = self_slot ? ld4(d_out + ((size_t)j * KS + K) * D + c) : f4zero();
        for (int e = colptr[j]; e < colptr[j + 1]; ++e) {
            const int n = dst[e], id = eid_t[e];
            const float sc = mean ? 1.f / (float)max(rowptr[n + 1] - rowptr[n], 1) : 1.f;
            float4 wv[K / 4], g[K];
#pragma unroll
            for (int u = 0; u < K / 4; ++u) wv[u] = ld4(w + (size_t)id * K + 4 * u);
#pragma unroll
            for (int k = 0; k < K; ++k) g[k] = ld4(d_out + ((size_t)n * KS + k) * D + c);
            float4 t = f4zero();
#pragma unroll
            for (int k = 0; k < K; ++k) fma4(t, f4get(wv[k >> 2], k & 3), g[k]);
            fma4(acc, sc, t);
        }
        if (addend) {      // a second gradient path into x (the skip connection around the conv), added last: what the autograd engine's add computes
            const float4 ad = ld4(addend + (size_t)j * D + c);
            acc.x += ad.x; acc.y += ad.y; acc.z += ad.z; acc.w += ad.w;
        }
        st4(dx + (size_t)j * D + c, acc);
    }
}

// K = 1, D % 4 == 0 (GCNConv's propagate on residue graphs): a thread per (node, float4 of channels), two edges in flight;
// same edge order and fma sequence as the scalar kernels.
__global__ void __launch_bounds__(kBlock) k_edge_wsum1_fwd_v4(const float* x, const float* w, const int* rowptr, const int* nbr,
                                                             const int* eid, int N, int D, int mean, float* out) {
    const int D4 = D >> 2;
    const size_t total = (size_t)N * D4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int n = (int)(i / D4), c = 4 * (int)(i % D4);
        const int beg = rowptr[n], end = rowptr[n + 1];
        float4 acc = f4zero();
        int e = beg;
        for (; e + 1 < end; e += 2) {
            const int s0 = nbr[e], s1 = nbr[e + 1];
            const float w0 = w[eid[e]], w1 = w[eid[e + 1]];
            const float4 v0 = ld4(x + (size_t)s0 * D + c), v1 = ld4(x + (size_t)s1 * D + c);
            acc.x = fmaf(w0, v0.x, acc.x); acc.y = fmaf(w0, v0.y, acc.y); acc.z = fmaf(w0, v0.z, acc.z); acc.w = fmaf(w0, v0.w, acc.w);
            acc.x = fmaf(w1, v1.x, acc.x); acc.y = fmaf(w1, v1.y, acc.y); acc.z = fmaf(w1, v1.z, acc.z); acc.w = fmaf(w1, v1.w, acc.w);
        }
        if (e < end) {
            const float w0 = w[eid[e]];
            const float4 v0 = ld4(x + (size_t)nbr[e] * D + c);
            acc.x = fmaf(w0, v0.x, acc.x); acc.y = fmaf(w0, v0.y, acc.y); acc.z = fmaf(w0, v0.z, acc.z); acc.w = fmaf(w0, v0.w, acc.w);
        }
        const float sc = mean ? 1.f / (float)max(end - beg, 1) : 1.f;
        st4(out + (size_t)n * D + c, make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc));
    }
}

__global__ void __launch_bounds__(kBlock) k_edge_wsum1_bwd_v4(const float* d_out, const float* w, const int* colptr, const int* dst,
                                                             const int* eid_t, const int* rowptr, int N, int D, int mean,
                                                             float* dx) {
    const int D4 = D >> 2;
    const size_t total = (size_t)N * D4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        const int j = (int)(i / D4), c = 4 * (int)(i % D4);
        const int beg = colptr[j], end = colptr[j + 1];
        float4 acc = f4zero();
        int e = beg;
        for (; e + 1 < end; e += 2) {
            const int n0 = dst[e], n1 = dst[e + 1];
            const float sc0 = mean ? 1.f / (float)max(rowptr[n0 + 1] - rowptr[n0], 1) : 1.f;
            const float sc1 = mean ? 1.f / (float)max(rowptr[n1 + 1] - rowptr[n1], 1) : 1.f;
            const float w0 = w[eid_t[e]], w1 = w[eid_t[e + 1]];
            const float4 g0 = ld4(d_out + (size_t)n0 * D + c), g1 = ld4(d_out + (size_t)n1 * D + c);
            acc.x = fmaf(sc0, w0 * g0.x, acc.x); acc.y = fmaf(sc0, w0 * g0.y, acc.y); acc.z = fmaf(sc0, w0 * g0.z, acc.z); acc.w = fmaf(sc0, w0 * g0.w, acc.w);
            acc.x = fmaf(sc1, w1 * g1.x, acc.x); acc.y = fmaf(sc1, w1 * g1.y, acc.y); acc.z = fmaf(sc1, w1 * g1.z, acc.z); acc.w = fmaf(sc1, w1 * g1.w, acc.w);
        }
        if (e < end) {
            const int n0 = dst[e];
            const float sc0 = mean ? 1.f / (float)max(rowptr[n0 + 1] - rowptr[n0], 1) : 1.f;
            const float w0 = w[eid_t[e]];
            const float4 g0 = ld4(d_out + (size_t)n0 * D + c);
            acc.x = fmaf(sc0, w0 * g0.x, acc.x); acc.y = fmaf(sc0, w0 * g0.y, acc.y); acc.z = fmaf(sc0, w0 * g0.z, acc.z); acc.w = fmaf(sc0, w0 * g0.w, acc.w);
        }
        st4(dx + (size_t)j * D + c, acc);
    }
}

}  // namespace glam

using namespace glam;

static int pool_dims(const char* fn, int64_t N, int64_t B, int D) {
    if (N < 0 || B < 0 || N >= INT32_MAX || B >= INT32_MAX) return fail(GLAM_E_INVALID, "%s: N/B out of range", fn);
    if (D <= 0) return fail(GLAM_E_INVALID, "%s: D=%d", fn, D);
    return GLAM_OK;
}

// x rows are ld floats apart and hold D channels (ld == D, or ld = D rounded up to a multiple of four with zero padding)
extern "C" int glam_pool5_padded_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int ld, int D, int k, float* out,
                                     int32_t* topk_idx, void* stream) {
    if (int rc = pool_dims("glam_pool5_fwd", N, B, D)) return rc;
    if (k < 1 || k > kMaxK) return fail(GLAM_E_UNSUPPORTED, "glam_pool5_fwd: k=%d not in 1..%d", k, kMaxK);
    GLAM_REQUIRE(ld >= D && (ld == D || ((ld & 3) == 0 && ld - D < 4)), "glam_pool5_fwd: ld=%d for D=%d (ld == D or D rounded up to a multiple of 4)", ld, D);
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && out && topk_idx && (N == 0 || x), "glam_pool5_fwd: null pointer");
    const hipStream_t s = (hipStream_t)stream;
    const bool v4 = (ld & 3) == 0 && aligned16(x) && ((D & 3) != 0 || aligned16(out));
    if (ld == D && N / B >= 64 && (D & 3) == 0 && D <= 64)      // large graphs: a block per graph
        hipLaunchKernelGGL(k_pool5_fwd_block, dim3(grid_for(B, 1)), dim3(kBlock), 0, s, x, ptr, (int)B, D, k, out, topk_idx);
    else if (v4 && ld <= 64)
        hipLaunchKernelGGL(k_pool5_fwd_v4<16>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, s, x, ptr, (int)B, ld, D, k, out, topk_idx);
    else if (v4 && ld <= 128)
        hipLaunchKernelGGL(k_pool5_fwd_v4<32>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, s, x, ptr, (int)B, ld, D, k, out, topk_idx);
    else if (ld == D)
        hipLaunchKernelGGL(k_pool5_fwd, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, s, x, ptr, (int)B, D, k, out, topk_idx);
    else
        return fail(GLAM_E_UNSUPPORTED, "glam_pool5_fwd: padded rows need ld <= 128 and a 16-byte aligned x (ld=%d)", ld);
    GLAM_LAUNCH_CHECK("glam_pool5_fwd");
    return GLAM_OK;
}

extern "C" int glam_pool5_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int k, float* out,
                              int32_t* topk_idx, void* stream) {
    return glam_pool5_padded_fwd(x, ptr, N, B, D, D, k, out, topk_idx, stream);
}

// d_x rows are ld floats apart; their pad columns (D..ld) are written as zeros
extern "C" int glam_pool5_padded_bwd(const float* d_out, const int32_t* ptr, const int32_t* topk_idx, int64_t N, int64_t B,
                                     int ld, int D, int k, float* d_x, void* stream) {
    if (int rc = pool_dims("glam_pool5_bwd", N, B, D)) return rc;
    if (k < 1 || k > kMaxK) return fail(GLAM_E_UNSUPPORTED, "glam_pool5_bwd: k=%d not in 1..%d", k, kMaxK);
    GLAM_REQUIRE(ld >= D && (ld == D || ((ld & 3) == 0 && ld - D < 4)), "glam_pool5_bwd: ld=%d for D=%d (ld == D or D rounded up to a multiple of 4)", ld, D);
    if (B == 0 || N == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && d_out && topk_idx && d_x, "glam_pool5_bwd: null pointer");
    const hipStream_t s = (hipStream_t)stream;
    const int vec = (D & 3) == 0 && aligned16(d_out);
    const bool v4 = (ld & 3) == 0 && aligned16(d_x) && (vec || ld != D);
    if (ld == D && N / B >= 64 && (D & 3) == 0 && D <= 64 && B * (int64_t)kPool5BwdChunks < 65536)
        hipLaunchKernelGGL(k_pool5_bwd_block, dim3((int)B * kPool5BwdChunks), dim3(kBlock), 0, s, d_out, ptr, topk_idx, (int)B, D, k, d_x);
    else if (v4 && ld <= 64)
        hipLaunchKernelGGL(k_pool5_bwd_v4<16>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, s, d_out, ptr, topk_idx, (int)B, ld, D, k, vec, d_x);
    else if (v4 && ld <= 128)
        hipLaunchKernelGGL(k_pool5_bwd_v4<32>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, s, d_out, ptr, topk_idx, (int)B, ld, D, k, vec, d_x);
    else if (ld == D)
        hipLaunchKernelGGL(k_pool5_bwd, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, s, d_out, ptr, topk_idx, (int)B, D, k, d_x);
    else
        return fail(GLAM_E_UNSUPPORTED, "glam_pool5_bwd: padded rows need ld <= 128 and a 16-byte aligned d_x (ld=%d)", ld);
    GLAM_LAUNCH_CHECK("glam_pool5_bwd");
    return GLAM_OK;
}

extern "C" int glam_pool5_bwd(const float* d_out, const int32_t* ptr, const int32_t* topk_idx, int64_t N, int64_t B,
                              int D, int k, float* d_x, void* stream) {
    return glam_pool5_padded_bwd(d_out, ptr, topk_idx, N, B, D, D, k, d_x, stream);
}

extern "C" int glam_segment_pool_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                                     float* out, int32_t* argmax, void* stream) {
    if (int rc = pool_dims("glam_segment_pool_fwd", N, B, D)) return rc;
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && out && (N == 0 || x) && (mode != 2 || argmax), "glam_segment_pool_fwd: null pointer");
    const dim3 grid(grid_for(B, kWavesPerBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_segment_pool_fwd<0>, grid, block, 0, s, x, ptr, (int)B, D, out, argmax);
    else if (mode == 1) hipLaunchKernelGGL(k_segment_pool_fwd<1>, grid, block, 0, s, x, ptr, (int)B, D, out, argmax);
    else if (mode == 2) hipLaunchKernelGGL(k_segment_pool_fwd<2>, grid, block, 0, s, x, ptr, (int)B, D, out, argmax);
    else return fail(GLAM_E_INVALID, "glam_segment_pool_fwd: mode=%d", mode);
    GLAM_LAUNCH_CHECK("glam_segment_pool_fwd");
    return GLAM_OK;
}

extern "C" int glam_segment_pool_bwd(const float* d_out, const int32_t* ptr, const int32_t* argmax, int64_t N,
                                     int64_t B, int D, int mode, float* d_x, void* stream) {
    if (int rc = pool_dims("glam_segment_pool_bwd", N, B, D)) return rc;
    if (B == 0 || N == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && d_out && d_x && (mode != 2 || argmax), "glam_segment_pool_bwd: null pointer");
    const dim3 grid(grid_for(B, kWavesPerBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_segment_pool_bwd<0>, grid, block, 0, s, d_out, ptr, argmax, (int)B, D, d_x);
    else if (mode == 1) hipLaunchKernelGGL(k_segment_pool_bwd<1>, grid, block, 0, s, d_out, ptr, argmax, (int)B, D, d_x);
    else if (mode == 2) hipLaunchKernelGGL(k_segment_pool_bwd<2>, grid, block, 0, s, d_out, ptr, argmax, (int)B, D, d_x);
    else return fail(GLAM_E_INVALID, "glam_segment_pool_bwd: mode=%d", mode);
    GLAM_LAUNCH_CHECK("glam_segment_pool_bwd");
    return GLAM_OK;
}

extern "C" int glam_segment_attn_fwd(const float* gate, const float* v, const int32_t* ptr, int64_t N, int64_t B, int D,
                                     float* out, float* stats, void* stream) {
    if (int rc = pool_dims("glam_segment_attn_fwd", N, B, D)) return rc;
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && out && stats && (N == 0 || (gate && v)), "glam_segment_attn_fwd: null pointer");
    const dim3 grid(grid_for(B, kWavesPerBlock)), block(kBlock);
    if ((D & 3) == 0 && D <= 64 && aligned16(v) && aligned16(out))
        hipLaunchKernelGGL(k_segment_attn_fwd_v4<16>, grid, block, 0, (hipStream_t)stream, gate, v, ptr, (int)B, D, out, stats);
    else if ((D & 3) == 0 && D <= 128 && aligned16(v) && aligned16(out))
        hipLaunchKernelGGL(k_segment_attn_fwd_v4<32>, grid, block, 0, (hipStream_t)stream, gate, v, ptr, (int)B, D, out, stats);
    else
        hipLaunchKernelGGL(k_segment_attn_fwd, grid, block, 0, (hipStream_t)stream, gate, v, ptr, (int)B, D, out, stats);
    GLAM_LAUNCH_CHECK("glam_segment_attn_fwd");
    return GLAM_OK;
}

extern "C" int glam_segment_attn_bwd(const float* gate, const float* v, const float* out, const float* stats,
                                     const float* d_out, const int32_t* ptr, int64_t N, int64_t B, int D,
                                     float* d_gate, float* d_v, void* stream) {
    if (int rc = pool_dims("glam_segment_attn_bwd", N, B, D)) return rc;
    if (B == 0 || N == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && gate && v && out && stats && d_out && d_gate && d_v, "glam_segment_attn_bwd: null pointer");
    const dim3 grid(grid_for(B, kWavesPerBlock)), block(kBlock);
    const bool al = aligned16(v) && aligned16(out) && aligned16(d_out) && aligned16(d_v);
    if ((D & 3) == 0 && D <= 64 && al)
        hipLaunchKernelGGL(k_segment_attn_bwd_v4<16>, grid, block, 0, (hipStream_t)stream, gate, v, out, stats, d_out, ptr, (int)B, D, d_gate, d_v);
    else if ((D & 3) == 0 && D <= 128 && al)
        hipLaunchKernelGGL(k_segment_attn_bwd_v4<32>, grid, block, 0, (hipStream_t)stream, gate, v, out, stats, d_out, ptr, (int)B, D, d_gate, d_v);
    else
        hipLaunchKernelGGL(k_segment_attn_bwd, grid, block, 0, (hipStream_t)stream, gate, v, out, stats, d_out, ptr, (int)B, D, d_gate, d_v);
    GLAM_LAUNCH_CHECK("glam_segment_attn_bwd");
    return GLAM_OK;
}

extern "C" int glam_edge_reduce_fwd(const float* msg, const int32_t* rowptr, const int32_t* eid, int64_t N, int64_t E,
                                    int D, int mode, float* out, int32_t* argmax, void* stream) {
    if (int rc = pool_dims("glam_edge_reduce_fwd", N, E, D)) return rc;
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(rowptr && out && (E == 0 || (msg && eid)) && (mode != 2 || argmax), "glam_edge_reduce_fwd: null pointer");
    const dim3 grid(grid_for(N * D, kBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_edge_reduce_fwd<0>, grid, block, 0, s, msg, rowptr, eid, (int)N, D, out, argmax);
    else if (mode == 1) hipLaunchKernelGGL(k_edge_reduce_fwd<1>, grid, block, 0, s, msg, rowptr, eid, (int)N, D, out, argmax);
    else if (mode == 2) hipLaunchKernelGGL(k_edge_reduce_fwd<2>, grid, block, 0, s, msg, rowptr, eid, (int)N, D, out, argmax);
    else return fail(GLAM_E_INVALID, "glam_edge_reduce_fwd: mode=%d", mode);
    GLAM_LAUNCH_CHECK("glam_edge_reduce_fwd");
    return GLAM_OK;
}

extern "C" int glam_edge_reduce_bwd(const float* d_out, const int32_t* rowptr, const int32_t* eid, const int32_t* argmax,
                                    int64_t N, int64_t E, int D, int mode, float* d_msg, void* stream) {
    if (int rc = pool_dims("glam_edge_reduce_bwd", N, E, D)) return rc;
    if (N == 0 || E == 0) return GLAM_OK;
    GLAM_REQUIRE(rowptr && eid && d_out && d_msg && (mode != 2 || argmax), "glam_edge_reduce_bwd: null pointer");
    const dim3 grid(grid_for(N * D, kBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_edge_reduce_bwd<0>, grid, block, 0, s, d_out, rowptr, eid, argmax, (int)N, D, d_msg);
    else if (mode == 1) hipLaunchKernelGGL(k_edge_reduce_bwd<1>, grid, block, 0, s, d_out, rowptr, eid, argmax, (int)N, D, d_msg);
    else if (mode == 2) hipLaunchKernelGGL(k_edge_reduce_bwd<2>, grid, block, 0, s, d_out, rowptr, eid, argmax, (int)N, D, d_msg);
    else return fail(GLAM_E_INVALID, "glam_edge_reduce_bwd: mode=%d", mode);
    GLAM_LAUNCH_CHECK("glam_edge_reduce_bwd");
    return GLAM_OK;
}

extern "C" int glam_edge_wsum_fwd(const float* x, const float* w, const int32_t* rowptr, const int32_t* src,
                                  const int32_t* eid, int64_t N, int64_t E, int D, int K, int mean, int self_slot, float* out,
                                  void* stream) {
    if (int rc = pool_dims("glam_edge_wsum_fwd", N, E, D)) return rc;
    if (self_slot && !((K == 4 || K == 8) && (D & 3) == 0 && aligned16(x) && aligned16(w) && aligned16(out)))
        return fail(GLAM_E_UNSUPPORTED, "glam_edge_wsum_fwd: self_slot needs K in {4, 8}, D %% 4 == 0 and 16-byte aligned tensors");
    if (K != 1 && K != 4 && K != 8) return fail(GLAM_E_UNSUPPORTED, "glam_edge_wsum_fwd: K=%d (1, or edge features padded to 4 or 8)", K);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(x && rowptr && out && (E == 0 || (w && src && eid)), "glam_edge_wsum_fwd: null pointer");
    const dim3 grid(grid_for(N * D, kBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (K == 1 && (D & 3) == 0)
        hipLaunchKernelGGL(k_edge_wsum1_fwd_v4, dim3(grid_for(N * (D / 4), kBlock)), block, 0, s, x, w, rowptr, src, eid, (int)N, D, mean, out);
    else if (K == 1) hipLaunchKernelGGL(k_edge_wsum_fwd<1>, grid, block, 0, s, x, w, rowptr, src, eid, (int)N, D, mean, out);
    else if ((D & 3) == 0 && aligned16(x) && aligned16(w) && aligned16(out)) {
        const dim3 g4(grid_for(N * (D / 4), kBlock));
        if (K == 4) hipLaunchKernelGGL(k_edge_wsum_fwd_v4<4>, g4, block, 0, s, x, w, rowptr, src, eid, (int)N, D, mean, self_slot, out);
        else hipLaunchKernelGGL(k_edge_wsum_fwd_v4<8>, g4, block, 0, s, x, w, rowptr, src, eid, (int)N, D, mean, self_slot, out);
    } else if (K == 4) hipLaunchKernelGGL(k_edge_wsum_fwd<4>, grid, block, 0, s, x, w, rowptr, src, eid, (int)N, D, mean, out);
    else hipLaunchKernelGGL(k_edge_wsum_fwd<8>, grid, block, 0, s, x, w, rowptr, src, eid, (int)N, D, mean, out);
    GLAM_LAUNCH_CHECK("glam_edge_wsum_fwd");
    return GLAM_OK;
}

extern "C" int glam_edge_wsum_bwd(const float* d_out, const float* w, const int32_t* colptr, const int32_t* dst,
                                  const int32_t* eid_t, const int32_t* rowptr, int64_t N, int64_t E, int D, int K,
                                  int mean, int self_slot, float* dx, void* stream) {
    if (int rc = pool_dims("glam_edge_wsum_bwd", N, E, D)) return rc;
    if (self_slot && !((K == 4 || K == 8) && (D & 3) == 0 && aligned16(d_out) && aligned16(w) && aligned16(dx)))
        return fail(GLAM_E_UNSUPPORTED, "glam_edge_wsum_bwd: self_slot needs K in {4, 8}, D %% 4 == 0 and 16-byte aligned tensors");
    if (K != 1 && K != 4 && K != 8) return fail(GLAM_E_UNSUPPORTED, "glam_edge_wsum_bwd: K=%d (1, or edge features padded to 4 or 8)", K);
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(d_out && colptr && rowptr && dx && (E == 0 || (w && dst && eid_t)), "glam_edge_wsum_bwd: null pointer");
    const dim3 grid(grid_for(N * D, kBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (K == 1 && (D & 3) == 0)
        hipLaunchKernelGGL(k_edge_wsum1_bwd_v4, dim3(grid_for(N * (D / 4), kBlock)), block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, dx);
    else if (K == 1) hipLaunchKernelGGL(k_edge_wsum_bwd<1>, grid, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, dx);
    else if ((D & 3) == 0 && aligned16(d_out) && aligned16(w) && aligned16(dx)) {
        const dim3 g4(grid_for(N * (D / 4), kBlock));
        if (K == 4) hipLaunchKernelGGL(k_edge_wsum_bwd_v4<4>, g4, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, self_slot, dx, (const float*)nullptr);
        else hipLaunchKernelGGL(k_edge_wsum_bwd_v4<8>, g4, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, self_slot, dx, (const float*)nullptr);
    } else if (K == 4) hipLaunchKernelGGL(k_edge_wsum_bwd<4>, grid, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, dx);
    else hipLaunchKernelGGL(k_edge_wsum_bwd<8>, grid, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, dx);
    GLAM_LAUNCH_CHECK("glam_edge_wsum_bwd");
    return GLAM_OK;
}

// ... + addend[N, D] (may be NULL): dx = (the sums) + addend, the add launch of a skip connection around the layer folded into this
// one.  Only the 16-byte form (K in {4, 8}, D % 4 == 0, aligned tensors); anything else is refused (the caller adds afterwards).
extern "C" int glam_edge_wsum_bwd_add(const float* d_out, const float* w, const int32_t* colptr, const int32_t* dst,
                                      const int32_t* eid_t, const int32_t* rowptr, int64_t N, int64_t E, int D, int K,
                                      int mean, int self_slot, const float* addend, float* dx, void* stream) {
    if (int rc = pool_dims("glam_edge_wsum_bwd_add", N, E, D)) return rc;
    if (!((K == 4 || K == 8) && (D & 3) == 0 && aligned16(d_out) && aligned16(w) && aligned16(dx) && aligned16(addend)))
        return fail(GLAM_E_UNSUPPORTED, "glam_edge_wsum_bwd_add: K in {4, 8}, D %% 4 == 0 and 16-byte aligned tensors only");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(d_out && colptr && rowptr && dx && (E == 0 || (w && dst && eid_t)), "glam_edge_wsum_bwd_add: null pointer");
    const dim3 g4(grid_for(N * (D / 4), kBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (K == 4) hipLaunchKernelGGL(k_edge_wsum_bwd_v4<4>, g4, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, self_slot ? 1 : 0, dx, addend);
    else hipLaunchKernelGGL(k_edge_wsum_bwd_v4<8>, g4, block, 0, s, d_out, w, colptr, dst, eid_t, rowptr, (int)N, D, mean, self_slot ? 1 : 0, dx, addend);
    GLAM_LAUNCH_CHECK("glam_edge_wsum_bwd_add");
    return GLAM_OK;
}

extern "C" int glam_s2s_attn_fwd(const float* x, const float* q, const int32_t* ptr, int64_t N, int64_t B, int D, float* r,
                                 float* stats, void* stream) {
    if (int rc = pool_dims("glam_s2s_attn_fwd", N, B, D)) return rc;
    if ((D & 3) || D > 128) return fail(GLAM_E_UNSUPPORTED, "glam_s2s_attn_fwd: D=%d (multiple of 4, <= 128)", D);
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && q && r && stats && (N == 0 || x), "glam_s2s_attn_fwd: null pointer");
    GLAM_REQUIRE(aligned16(x) && aligned16(q) && aligned16(r), "glam_s2s_attn_fwd: 16-byte alignment");
    if (D <= 64) hipLaunchKernelGGL(k_s2s_attn_fwd<16>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, q, ptr, (int)B, D, r, stats);
    else hipLaunchKernelGGL(k_s2s_attn_fwd<32>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, q, ptr, (int)B, D, r, stats);
    GLAM_LAUNCH_CHECK("glam_s2s_attn_fwd");
    return GLAM_OK;
}

extern "C" int glam_s2s_attn_bwd(const float* x, const float* q, const float* r, const float* stats, const float* d_r,
                                 const int32_t* ptr, int64_t N, int64_t B, int D, float* d_x, float* d_q, void* stream) {
    if (int rc = pool_dims("glam_s2s_attn_bwd", N, B, D)) return rc;
    if ((D & 3) || D > 128) return fail(GLAM_E_UNSUPPORTED, "glam_s2s_attn_bwd: D=%d (multiple of 4, <= 128)", D);
    if (B == 0) return GLAM_OK;
    GLAM_REQUIRE(ptr && q && r && stats && d_r && d_q && (N == 0 || (x && d_x)), "glam_s2s_attn_bwd: null pointer");
    GLAM_REQUIRE(aligned16(x) && aligned16(q) && aligned16(r) && aligned16(d_r) && aligned16(d_x) && aligned16(d_q),
                 "glam_s2s_attn_bwd: 16-byte alignment");
    if (D <= 64) hipLaunchKernelGGL(k_s2s_attn_bwd<16>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, q, r, stats, d_r, ptr,
                                    (int)B, D, d_x, d_q);
    else hipLaunchKernelGGL(k_s2s_attn_bwd<32>, dim3(grid_for(B, kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, q, r, stats, d_r, ptr,
                            (int)B, D, d_x, d_q);
    GLAM_LAUNCH_CHECK("glam_s2s_attn_bwd");
    return GLAM_OK;
}
