// Whole TripletMessage layer (reference: src_1gp/layer.py:15-64) as a fixed sequence of launches on one
// stream, plus the parameter staging kernels.  No host synchronisation, no allocation: hipGraph safe.
//
//   forward : k_stage_params -> k_ts_gemm (x @ [W_node | Wa_i | Wa_j] -> xw, a_ij)
//             -> k_triplet_fwd (gather / softmax / scatter-add -> aggr) -> k_ts_gemm (aggr @ W_scale + bias)
//   backward: k_ts_gemm (d_aggr = d_out @ W_scale^T) ; k_wgrad ([aggr|1]^T d_out -> d_W_scale, d_bias)
//             -> k_triplet_bwd_dst / reduce / k_triplet_bwd_src -> k_ts_gemm (d_x = [d_xw|d_a] @ Wcat^T)
//             ; k_wgrad ([d_xw|d_a]^T x -> d_Wcat) -> k_stage_params_bwd
//
// Separable attention: a_i = x @ Wa_i, a_j = x @ Wa_j with
// Wa_i[k,h] = sum_c W_node[k,h,c] att[h,c], Wa_j[k,h] = sum_c W_node[k,h,c] att[h,2C+c], and
// M[k,h] = sum_c W_edge[k,h,c] att[h,C+c]; they ride along as 8 extra columns of the node GEMM.
// All node-feature widths are padded to Cp (multiple of 4) so every row is 16-byte aligned.
#include "dense.h"

#include <stdlib.h>

namespace glam {

// layout of the staged-parameter buffer (floats; every offset is a multiple of 4)
struct Staged {
    size_t img_node, img_upd, img_dagg, img_dx, we_p, m, bias_p, dagg_pre, node_pre, total;
};
// W_scale^T as the operand fragments of B1's four matrix waves, ALREADY split into their three bf16 terms, in lane order:
// [wave w][k step st][column tile ct][term][lane] x 16 bytes (kDaggPreFloats floats; H = 3 only).  The matrix waves produce the tiles the
// vector waves wait for, and their prologue — 24 scattered weight loads, then 400 vector instructions of splits — stood in front of the
// first tile (r6_ws_timeline_b1024.txt: first tile 4.3 k cycles behind the block's barrier); 36 coalesced 1 KB loads replace it.
constexpr int kDaggPreFloats = 4 * 2 * 3 * 3 * 64 * 4;
// [W_node | Wa] as the operand fragments of the node product inside k_gru_fwd_ws (block.hip: the GRU step of a block that is applied again
// writes the next application's x @ [W_node | Wa]): [producer wave p][k step s][column tile j][term][lane] x 16 bytes — lane (c, kb) of
// fragment (p, s, j) holds W[32 s + 8 kb .. + 7][16 (3 p + j) + c], split into its three bf16 terms (zero beyond the matrix).  The same
// size as the B1 image; present when the node image has 192 column positions (64 < H * Cp + 8 <= 192).
constexpr int kNodePreFloats = 4 * 2 * 3 * 3 * 64 * 4;
static Staged staged_layout(int H, int Cp, int Dp) {
    const int HC = H * Cp;
    Staged s;
    size_t o = 0;
    s.img_node = o; o += ts_image_floats(Cp, HC + 8);     // x[N,Cp]       @ Wcat[Cp, HC+8]
    s.img_upd = o;  o += ts_image_floats(HC, Cp);         // aggr[N,HC]    @ Ws_p[HC, Cp]
    s.img_dagg = o; o += ts_image_floats(Cp, HC);         // d_out[N,Cp]   @ Ws_p^T
    s.img_dx = o;   o += ts_image_floats(HC + 8, Cp);     // [d_xw|d_a]    @ Wcat^T
    s.we_p = o;     o += (size_t)Dp * HC;
    s.m = o;        o += (size_t)Dp * 4;
    s.bias_p = o;   o += (size_t)Cp;
    o = (o + 63) & ~(size_t)63;                          // (256-byte aligned: 1 KB coalesced fragment loads)
    s.dagg_pre = o; o += (H == 3 && HC <= 192) ? kDaggPreFloats : 0;
    s.node_pre = o; o += (HC + 8 > 64 && HC + 8 <= 192) ? kNodePreFloats : 0;
    s.total = o;
    return s;
}
// gradient buffer: d_Wcat[Cp, HC+8] | d_WsB[HC+1, Cp] (last row = d_bias) | d_We_p[Dp, HC] | d_M[Dp, 4]
struct DStaged {
    size_t d_wcat, d_wsb, d_we_p, d_m, total;
};
static DStaged dstaged_layout(int H, int Cp, int Dp) {
    const int HC = H * Cp;
    DStaged s;
    size_t o = 0;
    s.d_wcat = o; o += (size_t)Cp * (HC + 8);
    s.d_wsb = o;  o += (size_t)(HC + 1) * Cp; o = (o + 3) & ~(size_t)3;
    s.d_we_p = o; o += (size_t)Dp * HC;
    s.d_m = o;    o += (size_t)Dp * 4;
    s.total = o;
    return s;
}

// sum_i a[i*sa] * b[i*sb]; 16 products per step with all 32 loads in flight (these dots sit on the critical
// path of a latency-bound launch)
__device__ __forceinline__ float dot_strided(const float* a, int sa, const float* b, int sb, int n) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { av[u] = a[(size_t)(i + u) * sa]; bv[u] = b[(size_t)(i + u) * sb]; }
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            s0 = fmaf(av[u], bv[u], s0); s1 = fmaf(av[u + 1], bv[u + 1], s1);
            s2 = fmaf(av[u + 2], bv[u + 2], s2); s3 = fmaf(av[u + 3], bv[u + 3], s3);
        }
    }
    {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool okk = i + u < n;
            av[u] = okk ? a[(size_t)(i + u) * sa] : 0.f;
            bv[u] = okk ? b[(size_t)(i + u) * sb] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            s0 = fmaf(av[u], bv[u], s0); s1 = fmaf(av[u + 1], bv[u + 1], s1);
            s2 = fmaf(av[u + 2], bv[u + 2], s2); s3 = fmaf(av[u + 3], bv[u + 3], s3);
        }
    }
    return (s0 + s1) + (s2 + s3);
}

struct StageArgs {
    const float* wn; const float* we; const float* att; const float* wsc; const float* bias;
    int C, H, De, Cp, Dp;
    float* base; Staged L;
};

// logical Wcat[k][m], k < Cp, m < H*Cp + 8.  The separable-attention columns (m >= H*Cp, a 60-term contraction each)
// are produced by the dot blocks of k_stage_params; *is_dot tells the copy loop to leave the element alone.
__device__ __forceinline__ float wcat_val(const StageArgs& a, int k, int m, bool* is_dot) {
    const int C = a.C, H = a.H, Cp = a.Cp;
    *is_dot = false;
    if (k >= C) return 0.f;
    if (m < H * Cp) {
        const int h = m / Cp, c = m % Cp;
        return c < C ? a.wn[(size_t)k * H * C + h * C + c] : 0.f;
    }
    const int s = m - H * Cp, h = s & 3;
    if (h >= H) return 0.f;
    *is_dot = true;
    return 0.f;
}
// position of logical column m inside a k_ts_gemm image row (inverse of ts_col_of_pos)
__device__ __forceinline__ int ts_pos_of_col(int m) { return (m & ~63) + (m & 3) * 16 + ((m >> 2) & 15); }
// logical Ws_p[k][m], k < H*Cp, m < Cp
__device__ __forceinline__ float wsp_val(const StageArgs& a, int k, int m) {
    const int h = k / a.Cp, c = k % a.Cp;
    return (c < a.C && m < a.C) ? a.wsc[(size_t)(h * a.C + c) * a.C + m] : 0.f;
}

// (bid, nblocks: the block's place in the staging part of the launch — glam_prestage appends image blocks behind it)
__device__ __forceinline__ void stage_params_block(const StageArgs& a, int copy_blocks, int bid, int nblocks) {
    const int C = a.C, H = a.H, De = a.De, Cp = a.Cp, Dp = a.Dp, HC = H * Cp;
    const int Kp1 = (Cp + 15) & ~15, Kp2 = (HC + 15) & ~15, Kp4 = (HC + 8 + 15) & ~15;
    const int P1 = HC + 8 <= 64 ? 64 : 192, P3 = HC <= 64 ? 64 : 192;   // image column count 16*MT
    if (bid >= copy_blocks) {
        // ---- dot blocks: one 16-lane group per contraction (one round trip of loads + a DPP butterfly) ----
        //   item < 8*C        : Wa[k][s] = sum_c W_node[k,h,c] att[h, side*2C + c]   (s = side*4 + h) -> node image + d_x image
        //   item >= 8*C       : M[kk][h] = sum_c W_edge[kk,h,c] att[h, C + c]
        const int lg = threadIdx.x & 15;
        const int item = (bid - copy_blocks) * (kBlock / 16) + (threadIdx.x >> 4);
        const bool is_m = item >= 8 * C;
        const int r = is_m ? (item - 8 * C) >> 2 : item >> 3;          // k or kk
        const int sx = is_m ? (item - 8 * C) & 3 : item & 7, h = sx & 3, side = sx >> 2;
        if ((is_m && r >= De) || h >= H) return;
        const float* w = (is_m ? a.we : a.wn) + (size_t)r * H * C + h * C;
        const float* t = a.att + (size_t)h * 3 * C + (is_m ? C : (side ? 2 * C : 0));
        float wv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = min(lg + 16 * u, C - 1); wv[u] = w[c]; tv[u] = t[c]; }
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lg + 16 * u < C) v = fmaf(wv[u], tv[u], v);
        for (int c = lg + 64; c < C; c += 16) v = fmaf(w[c], t[c], v);
        v = group_sum<16>(v);
        if (lg == 0) {
            if (is_m) {
                a.base[a.L.m + r * 4 + h] = v;
            } else {
                const int m = HC + sx;                       // Wcat[k = r][m]
                a.base[a.L.img_node + ((size_t)(r >> 2) * P1 + ts_pos_of_col(m)) * 4 + (r & 3)] = v;
                a.base[a.L.img_dx + ((size_t)(m >> 2) * 64 + ts_pos_of_col(r)) * 4 + (m & 3)] = v;   // Wcat^T[m][r]
                if (a.L.total > a.L.node_pre) {              // ... and its three bf16 terms into the node product's fragment image
                    const int f = ((m / 48) * 2 + (r >> 5)) * 3 + (m % 48) / 16, ln = ((r & 31) >> 3) * 16 + (m & 15);
                    unsigned hh, mm, ll;
                    split2(v, 0.f, hh, mm, ll);
                    unsigned short* dst = reinterpret_cast<unsigned short*>(a.base + a.L.node_pre) + ((size_t)(f * 3) * 64 + ln) * 8 + (r & 7);
                    dst[0] = (unsigned short)hh; dst[64 * 8] = (unsigned short)mm; dst[2 * 64 * 8] = (unsigned short)ll;
                }
            }
        }
        return;
    }
    const int n1 = Kp1 * P1, n2 = Kp2 * 64, n3 = Kp1 * P3, n4 = Kp4 * 64, n5 = Dp * HC, n6 = Dp * 4, n7 = Cp;
    const int n8 = (H == 3 && HC <= 192) ? 4 * 2 * 3 * 64 : 0;          // pre-split fragments of B1's matrix waves: one item per (w, st, ct, lane)
    const int n9 = (a.L.total > a.L.node_pre) ? 4 * 2 * 3 * 64 : 0;      // pre-split fragments of the node product inside the GRU step: (p, s, j, lane)
    const int total = n1 + n2 + n3 + n4 + n5 + n6 + n7 + n8 + n9;
    // image element idx -> (k, logical column m): layout [k/4][p][k%4], column order ts_col_of_pos
    for (int idx = bid * kBlock + threadIdx.x; idx < total; idx += copy_blocks * kBlock) {
        int i = idx;
        bool is_dot = false;
        if (i < n1) {            // node image: K = Cp, M = HC + 8
            const int k = (i >> 2) / P1 * 4 + (i & 3), m = ts_col_of_pos((i >> 2) % P1);
            const float v = (k < Cp && m < HC + 8) ? wcat_val(a, k, m, &is_dot) : 0.f;
            if (!is_dot) a.base[a.L.img_node + i] = v;
            continue;
        }
        i -= n1;
        if (i < n2) {            // update image: K = HC, M = Cp
            const int k = (i >> 2) / 64 * 4 + (i & 3), m = ts_col_of_pos((i >> 2) % 64);
            a.base[a.L.img_upd + i] = (k < HC && m < Cp) ? wsp_val(a, k, m) : 0.f;
            continue;
        }
        i -= n2;
        if (i < n3) {            // d_aggr image: K = Cp, M = HC, logical W[k][m] = Ws_p[m][k]
            const int k = (i >> 2) / P3 * 4 + (i & 3), m = ts_col_of_pos((i >> 2) % P3);
            a.base[a.L.img_dagg + i] = (k < Cp && m < HC) ? wsp_val(a, m, k) : 0.f;
            continue;
        }
        i -= n3;
        if (i < n4) {            // d_x image: K = HC + 8, M = Cp, logical W[k][m] = Wcat[m][k]
            const int k = (i >> 2) / 64 * 4 + (i & 3), m = ts_col_of_pos((i >> 2) % 64);
            const float v = (k < HC + 8 && m < Cp) ? wcat_val(a, m, k, &is_dot) : 0.f;
            if (!is_dot) a.base[a.L.img_dx + i] = v;
            continue;
        }
        i -= n4;
        if (i < n5) {
            const int k = i / HC, m = i % HC, h = m / Cp, c = m % Cp;
            a.base[a.L.we_p + i] = (k < De && c < C) ? a.we[(size_t)k * H * C + h * C + c] : 0.f;
            continue;
        }
        i -= n5;
        if (i < n6) {
            const int k = i >> 2, h = i & 3;
            if (!(k < De && h < H)) a.base[a.L.m + i] = 0.f;        // the real entries come from the dot blocks
            continue;
        }
        i -= n6;
        if (i < n7) {
            a.base[a.L.bias_p + i] = i < C ? a.bias[i] : 0.f;
            continue;
        }
        i -= n7;
        if (i >= n8) {
            // (p, s, j, lane = (c, kb)): rows k0 .. k0 + 7 of column mcol of [W_node | Wa].  The attention columns are the dot blocks'
            // (they write their elements' three terms themselves): here only what is zero there — the rows beyond C
            i -= n8;
            const int lane = i & 63, f = i >> 6, j = f % 3, st = (f / 3) & 1, w = f / 6;
            const int c = lane & 15, kb = lane >> 4, mcol = 16 * (3 * w + j) + c, k0 = 32 * st + 8 * kb;
            float v[8];
            bool dot[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dot[u] = false; v[u] = (k0 + u < Cp && mcol < HC + 8) ? wcat_val(a, k0 + u, mcol, &dot[u]) : 0.f; }
            const Bf16x3 fr = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
            float* dst = a.base + a.L.node_pre + ((size_t)(f * 3) * 64 + lane) * 4;
            if (!(dot[0] | dot[1] | dot[2] | dot[3] | dot[4] | dot[5] | dot[6] | dot[7])) {
                *reinterpret_cast<bf16x8_t*>(dst) = fr.hi;
                *reinterpret_cast<bf16x8_t*>(dst + 256) = fr.mid;
                *reinterpret_cast<bf16x8_t*>(dst + 512) = fr.lo;
            } else {
                unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (!dot[u]) { d16[u] = 0; d16[512 + u] = 0; d16[1024 + u] = 0; }
            }
            continue;
        }
        {   // (w, st, ct, lane = (c, kq)): rows k0 .. k0 + 7 of column mcol of W_scale^T — the values k_triplet_bwd_dst_ws's matrix wave w
            // reads out of the d_aggr image (w_load8) and splits (w_split8): the same function on the same numbers, bit-identical
            const int lane = i & 63, f = i >> 6, ct = f % 3, st = (f / 3) & 1, w = f / 6;
            const int c = lane & 15, kq = lane >> 4, mcol = 16 * (3 * w + ct) + c, k0 = 32 * st + 8 * kq;
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (k0 + u < Cp && mcol < HC) ? wsp_val(a, mcol, k0 + u) : 0.f;
            const Bf16x3 fr = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
            float* dst = a.base + a.L.dagg_pre + ((size_t)(f * 3) * 64 + lane) * 4;
            *reinterpret_cast<bf16x8_t*>(dst) = fr.hi;
            *reinterpret_cast<bf16x8_t*>(dst + 256) = fr.mid;
            *reinterpret_cast<bf16x8_t*>(dst + 512) = fr.lo;
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_stage_params(StageArgs a, int copy_blocks) {
    stage_params_block(a, copy_blocks, (int)blockIdx.x, (int)gridDim.x);
}

// The derived weights of a whole model pass in one launch: the TripletMessage's staged images and up to six k_ts_gemm weight images
// (a GRU's four, the input linear's one) — three launches of 5 us each at the head of every training step before.
__global__ void __launch_bounds__(kBlock) k_prestage(StageArgs a, int copy_blocks, int stage_blocks, ImageJobs js) {
    if ((int)blockIdx.x < stage_blocks) stage_params_block(a, copy_blocks, (int)blockIdx.x, stage_blocks);
    else make_images_block(js, (int)blockIdx.x - stage_blocks);
}

// Wide layers (H*Cp + 8 > 192, e.g. hid_dim_alpha = 6): the same derived parameters as plain row-major matrices for
// the library GEMM: Wcat[Cp, HC+8] | Ws_p[HC, Cp] | We_p[Dp, HC] | M[Dp, 4] | bias_p[Cp].  One thread per element; the
// separable-attention entries are C-term dots (dot_strided keeps 32 loads in flight).
__global__ void __launch_bounds__(kBlock) k_stage_plain(const float* wn, const float* we, const float* att, const float* wsc,
                                                       const float* bias, int C, int H, int De, int Cp, int Dp, float* out) {
    const int HC = H * Cp, MC = HC + 8;
    const int n1 = Cp * MC, n2 = HC * Cp, n3 = Dp * HC, n4 = Dp * 4, total = n1 + n2 + n3 + n4 + Cp;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < total; idx += gridDim.x * kBlock) {
        int i = idx;
        float v = 0.f;
        if (i < n1) {
            const int k = i / MC, m = i % MC;
            if (k < C) {
                if (m < HC) {
                    const int h = m / Cp, c = m % Cp;
                    if (c < C) v = wn[(size_t)k * H * C + h * C + c];
                } else {
                    const int sx = m - HC, h = sx & 3, side = sx >> 2;
                    if (h < H) v = dot_strided(wn + (size_t)k * H * C + h * C, 1, att + (size_t)h * 3 * C + (side ? 2 * C : 0), 1, C);
                }
            }
        } else if ((i -= n1) < n2) {
            const int k = i / Cp, m = i % Cp, h = k / Cp, c = k % Cp;
            if (c < C && m < C) v = wsc[(size_t)(h * C + c) * C + m];
        } else if ((i -= n2) < n3) {
            const int k = i / HC, m = i % HC, h = m / Cp, c = m % Cp;
            if (k < De && c < C) v = we[(size_t)k * H * C + h * C + c];
        } else if ((i -= n3) < n4) {
            const int k = i >> 2, h = i & 3;
            if (k < De && h < H) v = dot_strided(we + (size_t)k * H * C + h * C, 1, att + (size_t)h * 3 * C + C, 1, C);
        } else {
            i -= n4;
            if (i < C) v = bias[i];
        }
        out[idx] = v;
    }
}

// chain rule of k_stage_params
__global__ void __launch_bounds__(kBlock) k_stage_params_bwd(const float* wn, const float* we, const float* att,
                                                            const float* d_Wcat, const float* d_WsB,
                                                            const float* d_We_p, const float* d_M, int C, int H,
                                                            int De, int Cp, int Dp, float* d_wn, float* d_we,
                                                            float* d_att, float* d_wsc, float* d_bias) {
    const int MC = H * Cp + 8;
    const int n_wn = C * H * C, n_we = De * H * C, n_att = H * 3 * C, n_ws = H * C * C, n_b = C;
    const int total = n_wn + n_we + n_att + n_ws + n_b;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < total; idx += gridDim.x * kBlock) {
        if (idx < n_wn) {
            const int k = idx / (H * C), m = idx % (H * C), h = m / C, c = m % C;
            const float* dr = d_Wcat + (size_t)k * MC;
            float v = dr[h * Cp + c];
            v = fmaf(dr[H * Cp + h], att[(size_t)h * 3 * C + c], v);
            v = fmaf(dr[H * Cp + 4 + h], att[(size_t)h * 3 * C + 2 * C + c], v);
            d_wn[idx] = v;
        } else if (idx < n_wn + n_we) {
            const int i = idx - n_wn, k = i / (H * C), m = i % (H * C), h = m / C, c = m % C;
            d_we[i] = fmaf(d_M[k * 4 + h], att[(size_t)h * 3 * C + C + c], d_We_p[(size_t)k * H * Cp + h * Cp + c]);
        } else if (idx < n_wn + n_we + n_att) {
            const int i = idx - n_wn - n_we, h = i / (3 * C), t = i % (3 * C), part = t / C, c = t % C;
            float v;
            if (part == 1) v = dot_strided(d_M + h, 4, we + h * C + c, H * C, De);
            else v = dot_strided(d_Wcat + H * Cp + (part == 0 ? 0 : 4) + h, MC, wn + h * C + c, H * C, C);
            d_att[i] = v;
        } else if (idx < n_wn + n_we + n_att + n_ws) {
            const int i = idx - n_wn - n_we - n_att, row = i / C, col = i % C, h = row / C, c = row % C;
            d_wsc[i] = d_WsB[(size_t)(h * Cp + c) * Cp + col];
        } else {
            const int i = idx - n_wn - n_we - n_att - n_ws;
            d_bias[i] = d_WsB[(size_t)H * Cp * Cp + i];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Parameter gradients straight from the partial sets of the backward pass (k_wgrad / k_wgrad_x3 slabs of the two weight-gradient
// products, block partials of B1), i.e. the final fixed-order reductions AND the chain rule of k_stage_params in ONE
// launch (it replaces k_final_reduce -> k_stage_params_bwd).  Block roles:
//   A  d_weight_scale / d_bias        block per (slab, tj, kq) of product 1 ([aggr|1]^T d_out)
//   B  d_weight_node                  block per (slab, tj, kq) of product 2, plus d_Wa_i[k, :], d_Wa_j[k, :] of the block's 16 columns k, then
//                                     d_Wcat[k,h,c] + d_Wa_i[k,h] att_i[h,c] + d_Wa_j[k,h] att_j[h,c]
//   C  d_weight_triplet_att head h    a block per (head, third: att_i | att_e | att_j): d_Wa[:,h] (or d_M[:,h]) into LDS, then the
//                                     contraction with W_node (W_edge), 4 row quarters per column summed in order
//   D  d_weight_edge                  a block per 8 float4 columns of the B1 block partials + d_M[k,h] att_e[h,c]
// Round 6: every partial is read as 16 bytes.  A slab stores the four rows i = 16 kq + 4 r + ti (r = 0..3) of column j = 4 c + tj next to
// one another — one float4 UNIT (ti, tj, kq, c) —, so a role A / B block owns 64 units x nsplit splits: thread (split class sc = wave,
// ti, c) adds the splits s = sc, sc + 4, ... of its unit in ascending order (one 1 KB request per wave and split: four 256-byte runs),
// the four classes meet in LDS in class order, and thread (ti, c, r) finishes its element.  A quarter of the load instructions of the
// scalar form, and with the 128 splits k_wgrad_x3 leaves (one block per CU) two round trips instead of four; the B1 block partials
// ([blocks][P], P a multiple of 4) likewise, 8 float4 columns x 32 row classes per block.  Every sum runs in a fixed order.
struct ParamGradArgs {
    const float* p1; int ns1;            // product 1 partials: i in [0, HC], j in [0, Cp)
    const float* p2; int ns2;            // product 2 partials ([d_xw|d_a]^T x): i in [0, HC+8), j in [0, Cp)
    const float* p3; int ns3; int P;     // B1 block partials [ns3][P]: d_We_p (Dp*HC) | d_M (Dp*4)
    // ns3 rows in up to three buffers of ns3_each rows (the applications of a layer that shares its weights: p3, p3b, p3c)
    const float* wn; const float* we; const float* att;
    int C, H, De, Cp, Dp;
    float* d_wn; float* d_we; float* d_att; float* d_wsc; float* d_bias;
    int blocksA, blocksB, blocksC;
    // optional addends, laid out like the outputs (the gradient carry of a layer applied message_steps times: summed here instead
    // of by a separate add launch)
    const float* c_wn; const float* c_we; const float* c_att; const float* c_wsc; const float* c_bias;
    const float* p3b; const float* p3c; int ns3_each;      // (ns3_each == ns3: one buffer)
};

// float4 unit (ti, tj, kq, c) of a 64 x 64 slab: rows 16 kq + 4 r + ti (r = the component), column 4 c + tj (the decode of k_final_reduce)
__device__ __forceinline__ int wg_unit(int ti, int tj, int kq, int c) { return ((ti * 4 + tj) * 64 + kq * 16 + c) << 2; }
__device__ __forceinline__ const float* wg_unit_ptr(const float* partial, int nsplit, int slab, int ti, int tj, int kq, int c) {
    return partial + (size_t)slab * nsplit * kWgSlabStride + wg_unit(ti, tj, kq, c);
}
__device__ __forceinline__ void add4(float4& a, const float4& v, bool ok) {
    a.x += ok ? v.x : 0.f; a.y += ok ? v.y : 0.f; a.z += ok ? v.z : 0.f; a.w += ok ? v.w : 0.f;
}
// A launch of this kernel is a handful of dependent memory round trips and nothing else, so every role ISSUES all the loads of a
// stage before the first use (clamped addresses, i.e. unconditional loads: a load under a condition is waited for on the spot), and the
// independent sums of a thread share the round trips.
// the splits s = sc, sc + 4, ... < nsplit of unit `p`, added in ascending order (NB requests per round trip)
template <int NB>
__device__ __forceinline__ float4 wg_class_sum(const float* p, int nsplit, int sc) {
    float4 acc = f4zero();
    for (int s0 = sc; s0 < nsplit; s0 += 4 * NB) {
        float4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = ld4g(p + (size_t)min(s0 + 4 * u, nsplit - 1) * kWgSlabStride);
#pragma unroll
        for (int u = 0; u < NB; ++u) add4(acc, v[u], s0 + 4 * u < nsplit);
    }
    return acc;
}
// ... of two or three units, their requests in flight together
template <int NB>
__device__ __forceinline__ void wg_class_sum3(const float* pa, const float* pb, const float* pc, bool three, int nsplit, int sc,
                                              float4& sa, float4& sb, float4& sc3) {
    sa = sb = sc3 = f4zero();
    for (int s0 = sc; s0 < nsplit; s0 += 4 * NB) {
        float4 va[NB], vb[NB], vc[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const size_t o = (size_t)min(s0 + 4 * u, nsplit - 1) * kWgSlabStride;
            va[u] = ld4g(pa + o);
            vb[u] = ld4g(pb + o);
        }
        if (three) {                                          // (block-uniform)
#pragma unroll
            for (int u = 0; u < NB; ++u) vc[u] = ld4g(pc + (size_t)min(s0 + 4 * u, nsplit - 1) * kWgSlabStride);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const bool ok = s0 + 4 * u < nsplit;
            add4(sa, va[u], ok);
            add4(sb, vb[u], ok);
            if (three) add4(sc3, vc[u], ok);
        }
    }
}
// component r of a unit, r known at run time only (block-uniform).  Three selects: indexing the float4 (f4get's chain is folded into that)
// puts it into scratch memory
__device__ __forceinline__ float pick4(float4 v, int r) {
    float x = v.x;
    asm volatile("" : "+v"(x));
    x = r == 1 ? v.y : x;
    asm volatile("" : "+v"(x));
    x = r == 2 ? v.z : x;
    asm volatile("" : "+v"(x));
    return r == 3 ? v.w : x;
}
// the row class of a gradient row i: slab, kq and the component r of its units (ti = i & 3)
struct WgRow { int slab, kq, r, ti; };
__device__ __forceinline__ WgRow wg_row(int i) { const int ii = i & 63, t = ii >> 2; return WgRow{i >> 6, t >> 2, t & 3, ii & 3}; }

// sum over the B1 block partials of element e, cooperatively by a 16-lane group (every lane gets the total): lane lg takes blocks
// lg, lg + 16, ...
constexpr int kB1Batch = 16;
struct B1Batch { float v[kB1Batch]; };
// the rows of up to three buffers of `each` rows behind one another (the applications of a layer that shares its weights): the SETS
// instantiation, whose loops run set by set — a wave-uniform base pointer per set.  (A per-lane row-to-buffer select made the compiler
// branch around the loads and wait for each on the spot: role D alone took 18.8 us of k_param_grads<true>, role C 10.7; and as part
// of the one-buffer kernel the selects cost the headline step 4.5 us.)
struct B1Src { const float* p3; const float* p3b; const float* p3c; int each; };
__device__ __forceinline__ const float* b1_set(const B1Src& src, int q, int nset) {
    return q == 1 && nset > 1 ? src.p3b : q == 2 && nset > 2 ? src.p3c : src.p3;      // (a set that is not there re-reads the first; masked)
}
__device__ __forceinline__ void b1_request(B1Batch& b, const float* p3, int ns3, int P, int e, int s0) {
#pragma unroll
    for (int u = 0; u < kB1Batch; ++u) b.v[u] = p3[(size_t)min(s0 + 16 * u, ns3 - 1) * P + e];
}
__device__ __forceinline__ float b1_add(float part, const B1Batch& b, int ns3, int s0) {
#pragma unroll
    for (int u = 0; u < kB1Batch; ++u) part += s0 + 16 * u < ns3 ? b.v[u] : 0.f;
    return part;
}
__device__ __forceinline__ float b1_sum16(const float* p3, int ns3, int P, int e, int lg) {
    float part = 0.f;
    for (int s = lg; s < ns3; s += 16 * kB1Batch) {
        B1Batch b;
        b1_request(b, p3, ns3, P, e, s);
        part = b1_add(part, b, ns3, s);
    }
    return group_sum<16>(part);
}
// the same over the sets: the batches of all three in flight together, added set after set
__device__ __forceinline__ float b1_sum16_sets(const B1Src& src, int ns3, int P, int e, int lg) {
    const int nset = ns3 / src.each;
    float part = 0.f;
    for (int s = lg; s < src.each; s += 16 * kB1Batch) {
        B1Batch b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b1_request(b[q], b1_set(src, q, nset), src.each, P, e, s);
#pragma unroll
        for (int q = 0; q < 3; ++q) part = b1_add(part, b[q], q < nset ? src.each : 0, s);
    }
    return group_sum<16>(part);
}

// Role D's loads, with the grain of the partial rows: lane (col8 = lane & 7) on 8 consecutive float4 columns — one 128-byte run of a
// row — and (rc = tid >> 3) on 32 interleaved row classes; a thread adds its rows s = rc, rc + 32, ... in order, two columns (its weight
// column and a d_M column) sharing the round trips.
template <bool SETS>
__device__ __forceinline__ void b1_cols(const B1Src& src, int ns3, int P, int col_a, int col_b, int rc, float4& sa, float4& sb) {
    constexpr int BATCH = SETS ? 4 : 8, NQ = SETS ? 3 : 1;
    const int each = SETS ? src.each : ns3, nset = SETS ? ns3 / src.each : 1;
    sa = sb = f4zero();
    for (int s0 = rc; s0 < each; s0 += 32 * BATCH) {
        float4 va[NQ][BATCH], vb[NQ][BATCH];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float* base = b1_set(src, q, nset);
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const float* row = base + (size_t)min(s0 + 32 * u, each - 1) * P;
                va[q][u] = ld4g(row + 4 * col_a);
                vb[q][u] = ld4g(row + 4 * col_b);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const bool ok = q < nset && s0 + 32 * u < each;
                add4(sa, va[q][u], ok);
                add4(sb, vb[q][u], ok);
            }
    }
}
// lanes l, l + 8, ..., l + 56 of a wave (the eight row classes it holds per column) summed into every one of them, in a fixed order
__device__ __forceinline__ float4 rows8_sum(float4 v) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
    }
    return v;
}

#ifdef GLAM_PG_PROF   // developer aid (tools/pg_prof.py): cycle stamps of thread 0 of every block, s_memtime = the device-wide clock
__device__ long long g_pg_prof[512 * 8];
#define PG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_pg_prof[blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PG_STAMP(k) do { } while (0)
#endif

// ---- the four roles, one function each (separate register allocations in the compiler's eyes: as one body the kernel held 106 scalar
//      registers, spilled 14 of them and kept a scratch segment) ----
template <int NB>
__device__ __forceinline__ void pg_role_a(const ParamGradArgs& a, int b, float4* s_red) {
    const int tid = threadIdx.x, C = a.C, Cp = a.Cp, HC = a.H * Cp;
    const int slab = b >> 4, tj = (b >> 2) & 3, kq = b & 3;
    // the element this thread finishes: (ti, c, r) -> row i, column col
    const int lr = tid & 3, lc = (tid >> 2) & 15, lti = tid >> 6;
    const int i = slab * 64 + 16 * kq + 4 * lr + lti, col = 4 * lc + tj;
    const int h = i / Cp, c = i - h * Cp;
    const bool is_bias = i == HC && col < C, mine = i < HC && c < C && col < C;
    const int idx = (h * C + c) * C + col;
    float carry = 0.f;                                        // requested before the partials are waited for
    if (is_bias && a.c_bias) carry = a.c_bias[col];
    if (mine && a.c_wsc) carry = a.c_wsc[idx];
    s_red[tid] = wg_class_sum<NB>(wg_unit_ptr(a.p1, a.ns1, slab, (tid >> 4) & 3, tj, kq, tid & 15), a.ns1, tid >> 6);
    __syncthreads();
    const float* sr = reinterpret_cast<const float*>(s_red) + (lti * 16 + lc) * 4 + lr;
    const float v = ((sr[0] + sr[256]) + sr[512]) + sr[768];
    if (is_bias) a.d_bias[col] = v + carry;
    else if (mine) a.d_wsc[idx] = v + carry;
}

template <int NB>
__device__ __forceinline__ void pg_role_b(const ParamGradArgs& a, int b, float4* s_red, float* s_dw) {
    const int tid = threadIdx.x, C = a.C, H = a.H, Cp = a.Cp, HC = H * Cp;
    const int slab = b >> 4, tj = (b >> 2) & 3, kq = b & 3;
    const int lr = tid & 3, lc = (tid >> 2) & 15, lti = tid >> 6;
    const int i = slab * 64 + 16 * kq + 4 * lr + lti, k = 4 * lc + tj;
    const int h = i / Cp, c = i - h * Cp;
    const bool mine = i < HC && c < C && k < C;
    const size_t idx = (size_t)k * H * C + h * C + c;
    const float att_i = mine ? a.att[(size_t)h * 3 * C + c] : 0.f, att_j = mine ? a.att[(size_t)h * 3 * C + 2 * C + c] : 0.f;
    const float carry = (mine && a.c_wn) ? a.c_wn[idx] : 0.f;
    // the attention-gradient rows of this block's 16 columns: d_Wa_i[k, hh] = row HC + hh, d_Wa_j[k, hh] = row HC + 4 + hh (HC is a multiple
    // of 4, so ti = hh); thread (sc, hh, c2) reads their units — ONE unit when both rows share it (components r and r + 1)
    const int sc = tid >> 6, hh = (tid >> 4) & 3, c2 = tid & 15;
    const WgRow ri = wg_row(HC + hh), rj = wg_row(HC + 4 + hh);
    const bool two = ri.slab != rj.slab || ri.kq != rj.kq;    // (block-uniform)
    float4 sm, si, sj;
    wg_class_sum3<NB>(wg_unit_ptr(a.p2, a.ns2, slab, hh, tj, kq, c2), wg_unit_ptr(a.p2, a.ns2, ri.slab, ri.ti, tj, ri.kq, c2),
                      wg_unit_ptr(a.p2, a.ns2, rj.slab, rj.ti, tj, rj.kq, c2), two, a.ns2, sc, sm, si, sj);
    s_red[tid] = sm;
    s_dw[tid] = pick4(si, ri.r);                              // [sc][hh][c2]
    s_dw[256 + tid] = pick4(two ? sj : si, rj.r);
    __syncthreads();
    if (mine) {
        const float* sr = reinterpret_cast<const float*>(s_red) + (lti * 16 + lc) * 4 + lr;
        float v = ((sr[0] + sr[256]) + sr[512]) + sr[768];
        const float* di = s_dw + h * 16 + lc;
        const float dwa_i = ((di[0] + di[64]) + di[128]) + di[192], dwa_j = ((di[256] + di[320]) + di[384]) + di[448];
        v = fmaf(dwa_i, att_i, v);
        v = fmaf(dwa_j, att_j, v);
        a.d_wn[idx] = v + carry;
    }
}

template <bool SETS, int NB>
__device__ __forceinline__ void pg_role_c(const ParamGradArgs& a, int b, float* s_dwa, float (*s_pc)[64], float* s_dm) {
    const int tid = threadIdx.x, lg = tid & 15, grp = tid >> 4;
    const int C = a.C, H = a.H, De = a.De, Cp = a.Cp, HC = H * Cp, WSZ = a.Dp * HC;
    const int h = b / 3, part = b - 3 * h;
    const int c = tid & 63, kq4 = tid >> 6;
    if (part == 1) {                                    // d_att_e[h, c] = sum_kk d_M[kk, h] W_edge[kk, h, c]
        const bool out_mine = tid < C, dm_mine = grp < De;
        float wcol[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wcol[kk] = (out_mine && kk < De) ? a.we[(size_t)kk * H * C + h * C + tid] : 0.f;
        const float carry = (out_mine && a.c_att) ? a.c_att[(size_t)h * 3 * C + C + tid] : 0.f;
        const int e_dm = WSZ + (dm_mine ? grp : 0) * 4 + h;
        const float dm = SETS ? b1_sum16_sets(B1Src{a.p3, a.p3b, a.p3c, a.ns3_each}, a.ns3, a.P, e_dm, lg) : b1_sum16(a.p3, a.ns3, a.P, e_dm, lg);
        if (dm_mine && lg == 0) s_dm[grp] = dm;
        __syncthreads();
        if (out_mine) {
            float v = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (kk < De) v = fmaf(s_dm[kk], wcol[kk], v);
            a.d_att[(size_t)h * 3 * C + C + tid] = v + carry;
        }
    } else {                                            // d_att_i / d_att_j[h, c] = sum_k d_Wa[k, h] W_node[k, h, c]
        const int side = part == 0 ? 0 : 1;
        // thread (kq4, c): the 16 rows k = 16 kq4 .. + 15 of its column of W_node, requested together with the d_Wa splits
        float wcol[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = 16 * kq4 + r;
            wcol[r] = (k < C && c < C) ? a.wn[(size_t)k * H * C + h * C + c] : 0.f;
        }
        const float carry = (tid < C && a.c_att) ? a.c_att[(size_t)h * 3 * C + 2 * C * side + tid] : 0.f;
        // d_Wa[k, h] for every column k = 4 c' + tj: thread (sc, tj, c') adds the splits of its unit of row HC + 4 side + h
        const WgRow rw = wg_row(HC + 4 * side + h);
        const float4 sv = wg_class_sum<NB>(wg_unit_ptr(a.p2, a.ns2, rw.slab, rw.ti, (tid >> 4) & 3, rw.kq, tid & 15), a.ns2, tid >> 6);
        s_pc[tid >> 6][tid & 63] = pick4(sv, rw.r);
        __syncthreads();
        if (tid < 64) {
            const int kcol = 4 * (tid & 15) + ((tid >> 4) & 3);
            s_dwa[kcol] = kcol < C ? ((s_pc[0][tid] + s_pc[1][tid]) + s_pc[2][tid]) + s_pc[3][tid] : 0.f;
        }
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v = fmaf(s_dwa[16 * kq4 + r], wcol[r], v);
        s_pc[kq4][c] = v;
        __syncthreads();
        if (tid < C) a.d_att[(size_t)h * 3 * C + 2 * C * side + tid] = (((s_pc[0][tid] + s_pc[1][tid]) + s_pc[2][tid]) + s_pc[3][tid]) + carry;
    }
}

template <bool SETS>
__device__ __forceinline__ void pg_role_d(const ParamGradArgs& a, int b, float4* s_red) {
    const int tid = threadIdx.x, w = tid >> 6, col8 = tid & 7, rc = tid >> 3;
    const int C = a.C, H = a.H, De = a.De, Cp = a.Cp, HC = H * Cp, WSZ = a.Dp * HC;
    // the element thread (col8' = tid >> 2, comp = tid & 3) of the first 32 finishes
    const int e = 4 * (b * 8 + ((tid >> 2) & 7)) + (tid & 3);
    const int kk = min(e / HC, a.Dp - 1), m = e - kk * HC, h = min(m / Cp, H - 1), c = m - h * Cp;
    const bool mine = tid < 32 && e < WSZ && kk < De && c < C;
    const int o = (kk * H + h) * C + (mine ? c : 0);
    const float att_e = mine ? a.att[(size_t)h * 3 * C + C + c] : 0.f;
    const float carry = (mine && a.c_we) ? a.c_we[o] : 0.f;
    const int col_a = min(b * 8 + col8, WSZ / 4 - 1), col_b = WSZ / 4 + min(col8, a.Dp - 1);      // weight column | d_M row (kk = col8)
    float4 sa, sb;
    b1_cols<SETS>(B1Src{a.p3, a.p3b, a.p3c, a.ns3_each}, a.ns3, a.P, col_a, col_b, rc, sa, sb);
    sa = rows8_sum(sa);
    sb = rows8_sum(sb);
    if ((tid & 63) < 8) { s_red[w * 16 + col8] = sa; s_red[w * 16 + 8 + col8] = sb; }
    __syncthreads();
    if (mine) {
        const float* sr = reinterpret_cast<const float*>(s_red);
        const int ia = ((tid >> 2) & 7) * 4 + (tid & 3), ib = (8 + kk) * 4 + h;
        const float dwe = ((sr[ia] + sr[64 + ia]) + sr[128 + ia]) + sr[192 + ia];
        const float dm = ((sr[ib] + sr[64 + ib]) + sr[128 + ib]) + sr[192 + ib];
        a.d_we[o] = fmaf(dm, att_e, dwe) + carry;
    }
}

template <bool SETS>
__global__ void __launch_bounds__(kBlock) k_param_grads(ParamGradArgs a) {
    PG_STAMP(0);
    __shared__ float4 s_red[256];
    __shared__ float s_dw[512];
    __shared__ float s_pc[4][64];
    __shared__ float s_dwa[64];
    __shared__ float s_dm[8];
    int b = blockIdx.x;
#ifdef GLAM_PG_ONLY      // developer aid (tools/pg_roles.py): time one block role alone (0 = A, 1 = B, 2 = C, 3 = D)
    {
        const int role = b < a.blocksA ? 0 : b < a.blocksA + a.blocksB ? 1 : b < a.blocksA + a.blocksB + a.blocksC ? 2 : 3;
        if (role != GLAM_PG_ONLY) return;
    }
#endif
    // (NB = the requests per round trip and split class: 10 covers the 40 splits of the B = 1 024 step in one trip without masked loads)
    const bool few1 = a.ns1 <= 40, few2 = a.ns2 <= 40;
    if (b < a.blocksA) {
        if (few1) pg_role_a<10>(a, b, s_red); else pg_role_a<16>(a, b, s_red);
    } else if ((b -= a.blocksA) < a.blocksB) {
        if (few2) pg_role_b<10>(a, b, s_red, s_dw); else pg_role_b<16>(a, b, s_red, s_dw);
    } else if ((b -= a.blocksB) < a.blocksC) {
        if (few2) pg_role_c<SETS, 10>(a, b, s_dwa, s_pc, s_dm); else pg_role_c<SETS, 16>(a, b, s_dwa, s_pc, s_dm);
    } else {
        pg_role_d<SETS>(a, b - a.blocksC, s_red);
    }
    PG_STAMP(4);
}

}  // namespace glam

using namespace glam;

#ifdef GLAM_PG_PROF
extern "C" int glam_debug_pg_prof(long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(glam::g_pg_prof), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

static int dims_ok(const char* fn, int C, int H, int De, int Cp, int Dp) {
    if (C <= 0 || H < 1 || H > 4 || De <= 0 || Cp < C || (Cp & 3))
        return fail(GLAM_E_INVALID, "%s: bad dims C=%d H=%d De=%d Cp=%d", fn, C, H, De, Cp);
    if ((Dp != 4 && Dp != 8) || De > Dp) return fail(GLAM_E_UNSUPPORTED, "%s: De=%d Dp=%d", fn, De, Dp);
    if (H * Cp + 8 > 192 || Cp > 64)
        return fail(GLAM_E_UNSUPPORTED, "%s: H*Cp+8=%d (max 192) / Cp=%d (max 64) outside the dense-kernel table", fn, H * Cp + 8, Cp);
    return GLAM_OK;
}

static int dims_plain(const char* fn, int C, int H, int De, int Cp, int Dp) {
    if (C <= 0 || H < 1 || H > 4 || De <= 0 || Cp < C || (Cp & 3) || Cp > 256)
        return fail(GLAM_E_INVALID, "%s: bad dims C=%d H=%d De=%d Cp=%d", fn, C, H, De, Cp);
    if ((Dp != 4 && Dp != 8) || De > Dp) return fail(GLAM_E_UNSUPPORTED, "%s: De=%d Dp=%d", fn, De, Dp);
    return GLAM_OK;
}

extern "C" size_t glam_triplet_plain_floats(int H, int Cp, int Dp) {
    const size_t HC = (size_t)H * Cp;
    return (size_t)Cp * (HC + 8) + HC * Cp + (size_t)Dp * HC + (size_t)Dp * 4 + Cp;
}

extern "C" int glam_triplet_stage_plain(const float* weight_node, const float* weight_edge, const float* att,
                                        const float* weight_scale, const float* bias, int C, int H, int De, int Cp, int Dp,
                                        float* plain, void* stream) {
    if (int rc = dims_plain("glam_triplet_stage_plain", C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(weight_node && weight_edge && att && weight_scale && bias && plain, "glam_triplet_stage_plain: null pointer");
    hipLaunchKernelGGL(k_stage_plain, dim3(grid_for((int64_t)glam_triplet_plain_floats(H, Cp, Dp), kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, weight_node, weight_edge, att, weight_scale, bias, C, H, De, Cp, Dp, plain);
    GLAM_LAUNCH_CHECK("glam_triplet_stage_plain");
    return GLAM_OK;
}

extern "C" size_t glam_triplet_staged_floats(int H, int Cp, int Dp) { return staged_layout(H, Cp, Dp).total; }
extern "C" size_t glam_triplet_dstaged_floats(int H, int Cp, int Dp) { return dstaged_layout(H, Cp, Dp).total; }

extern "C" int glam_triplet_stage_params(const float* weight_node, const float* weight_edge, const float* att,
                                         const float* weight_scale, const float* bias, int C, int H, int De, int Cp,
                                         int Dp, float* staged, void* stream) {
    if (int rc = dims_ok("glam_triplet_stage_params", C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(weight_node && weight_edge && att && weight_scale && bias && staged && aligned16(staged),
                 "glam_triplet_stage_params: null / misaligned pointer");
    StageArgs a{weight_node, weight_edge, att, weight_scale, bias, C, H, De, Cp, Dp, staged, staged_layout(H, Cp, Dp)};
    const int copy_blocks = grid_for((int64_t)a.L.total, kBlock);
    const int dot_blocks = (8 * C + 4 * Dp + kBlock / 16 - 1) / (kBlock / 16);
    hipLaunchKernelGGL(k_stage_params, dim3(copy_blocks + dot_blocks), dim3(kBlock), 0, (hipStream_t)stream, a, copy_blocks);
    GLAM_LAUNCH_CHECK("glam_triplet_stage_params");
    return GLAM_OK;
}

extern "C" int glam_prestage(const float* weight_node, const float* weight_edge, const float* att, const float* weight_scale,
                             const float* bias, int C, int H, int De, int Cp, int Dp, float* staged, int n_images,
                             const float* const* W, const int* dims, float* const* img, void* stream) {
    GLAM_REQUIRE(n_images >= 0 && n_images <= kMaxImageJobs, "glam_prestage: %d images (at most %d per launch)", n_images, kMaxImageJobs);
    GLAM_REQUIRE(staged || n_images > 0, "glam_prestage: nothing to build");
    GLAM_REQUIRE(n_images == 0 || (W && dims && img), "glam_prestage: null image table");
    StageArgs a{};
    int copy_blocks = 0, stage_blocks = 0;
    if (staged) {
        if (int rc = dims_ok("glam_prestage", C, H, De, Cp, Dp)) return rc;
        GLAM_REQUIRE(weight_node && weight_edge && att && weight_scale && bias && aligned16(staged), "glam_prestage: null / misaligned pointer");
        a = StageArgs{weight_node, weight_edge, att, weight_scale, bias, C, H, De, Cp, Dp, staged, staged_layout(H, Cp, Dp)};
        copy_blocks = grid_for((int64_t)a.L.total, kBlock);
        stage_blocks = copy_blocks + (8 * C + 4 * Dp + kBlock / 16 - 1) / (kBlock / 16);
    }
    ImageJobs js{};
    int blocks = 0;
    for (int q = 0; q < n_images; ++q) {
        GLAM_REQUIRE(W[q] && img[q] && aligned16(img[q]), "glam_prestage: image %d: null / misaligned pointer", q);
        int nb;
        if (dims[4 * q + 1] >= 2) {
            // transW = 2 / 3: matrix M (0: weight_ih, 1: weight_hh) of a GRU with K channels into the forward / backward image of
            // glam_gru_ws_make_pre (img = that image, shared by both matrices' jobs)
            const int Cg = dims[4 * q + 2], m = dims[4 * q + 3];
            GLAM_REQUIRE(dims[4 * q + 1] <= 3 && (m == 0 || m == 1) && dims[4 * q] == Cg, "glam_prestage: image %d: bad pre-split job", q);
            if (!(Cg >= 24 && Cg <= 64 && (Cg & 3) == 0)) return fail(GLAM_E_UNSUPPORTED, "glam_prestage: image %d: C=%d must be a multiple of 4 in 24..64", q, Cg);
            js.job[q] = ImageJob{W[q], Cg, 0, Cg, m, 0, img[q], blocks, dims[4 * q + 1] == 2 ? -1 : -2};
            nb = (24 * 64 + kBlock - 1) / kBlock;
        } else {
            nb = image_job(js.job[q], "glam_prestage", W[q], dims[4 * q], dims[4 * q + 1], dims[4 * q + 2], dims[4 * q + 3], img[q], blocks);
        }
        if (nb < 0) return nb;
        blocks += nb;
    }
    js.njobs = n_images;
    hipLaunchKernelGGL(k_prestage, dim3(stage_blocks + blocks), dim3(kBlock), 0, (hipStream_t)stream, a, copy_blocks, stage_blocks, js);
    GLAM_LAUNCH_CHECK("glam_prestage");
    return GLAM_OK;
}

extern "C" int glam_triplet_stage_params_bwd(const float* weight_node, const float* weight_edge, const float* att,
                                             const float* dstaged, int C, int H, int De, int Cp, int Dp,
                                             float* d_weight_node, float* d_weight_edge, float* d_att,
                                             float* d_weight_scale, float* d_bias, void* stream) {
    if (int rc = dims_plain("glam_triplet_stage_params_bwd", C, H, De, Cp, Dp)) return rc;   // no GEMM image involved: any width
    GLAM_REQUIRE(weight_node && weight_edge && att && dstaged && d_weight_node && d_weight_edge && d_att && d_weight_scale &&
                     d_bias, "glam_triplet_stage_params_bwd: null pointer");
    const DStaged L = dstaged_layout(H, Cp, Dp);
    const int total = C * H * C + De * H * C + H * 3 * C + H * C * C + C;
    hipLaunchKernelGGL(k_stage_params_bwd, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       weight_node, weight_edge, att, dstaged + L.d_wcat, dstaged + L.d_wsb, dstaged + L.d_we_p,
                       dstaged + L.d_m, C, H, De, Cp, Dp, d_weight_node, d_weight_edge, d_att, d_weight_scale, d_bias);
    GLAM_LAUNCH_CHECK("glam_triplet_stage_params_bwd");
    return GLAM_OK;
}

// The fused-GEMM variants of the aggregate kernels (forward + update, B2 + d_x) keep a 48 KB weight image per block in LDS: two
// 4-wave blocks per CU.  That wins while the launch is latency bound (B = 1024: 16.6 us against 9.7 + 11 + a launch boundary) and
// loses once the batch is large enough for the aggregate to need its full occupancy (B = 16 384: 249 us fused against 134 + 62).
static int64_t fuse_max_nodes() {
    static const int64_t v = (int64_t)1 << 40;
    return v;
}

extern "C" int glam_triplet_layer_fwd(const float* x, const float* edge_attr, const float* staged, const int32_t* rowptr,
                                      const int32_t* src, const int32_t* eid, int64_t N, int64_t E, int H, int Cp, int Dp,
                                      float slope, float* xw, float* a_ij, float* aggr, float* stats, float* out, void* stream) {
    if (int rc = dims_ok("glam_triplet_layer_fwd", Cp, H, Dp, Cp, Dp)) return rc;
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_triplet_layer_fwd: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(x && staged && xw && a_ij && out && (!aggr) == (!stats), "glam_triplet_layer_fwd: null pointer");
    GLAM_REQUIRE(aligned16(x) && aligned16(xw) && aligned16(a_ij) && aligned16(aggr) && aligned16(out) && aligned16(staged),
                 "glam_triplet_layer_fwd: 16-byte alignment");
    // aggr = stats = NULL: the inference forward (nothing is kept for a backward pass) — where the update GEMM runs inside the aggregate launch
    if (!aggr && !(triplet_fwd_can_fuse_update(H, Cp, Dp) && N <= fuse_max_nodes()))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_layer_fwd: aggr = NULL needs the fused update (glam_triplet_layer_infer_supported; H=%d Cp=%d Dp=%d)", H, Cp, Dp);
    hipStream_t s = (hipStream_t)stream;
    const int HC = H * Cp;
    const Staged L = staged_layout(H, Cp, Dp);
    TsArgs g1{x, Cp, Cp, nullptr, 0, 0, staged + L.img_node, nullptr, xw, HC, HC, a_ij, 8, 8, (int)N};
    if (int rc = launch_ts_gemm(g1, s)) return rc;
    if (triplet_fwd_can_fuse_update(H, Cp, Dp) && N <= fuse_max_nodes())   // aggregate + update GEMM in one launch
        return triplet_fwd_fused_update(xw, a_ij, edge_attr, staged + L.we_p, staged + L.m, rowptr, src, eid, N, E, H, Cp,
                                        Dp, slope, aggr, stats, staged + L.img_upd, staged + L.bias_p, out, s);
    if (int rc = glam_triplet_fwd(xw, a_ij, edge_attr, staged + L.we_p, staged + L.m, rowptr, src, eid, N, E, H, Cp, Dp, 1,
                                  slope, aggr, stats, stream))
        return rc;
    TsArgs g2{aggr, HC, HC, nullptr, 0, 0, staged + L.img_upd, staged + L.bias_p, out, Cp, Cp, nullptr, 0, 0, (int)N};
    return launch_ts_gemm(g2, s);
}

// Molecular graphs (in-degree <= 4: ELL index records from glam_ell_build; one-hot bond features of width 4): node GEMM, then the
// warp-specialised aggregate + update launch (csrc/triplet_ws.hip) — same outputs as glam_triplet_layer_fwd, bit for bit.
extern "C" int glam_triplet_layer_fwd_ell(const float* x, const float* edge_attr, const float* staged, const int32_t* ell_src,
                                          const int32_t* ell_eid, int edge_onehot, int64_t N, int64_t E, int H, int Cp, int Dp,
                                          float slope, float* xw, float* a_ij, float* aggr, float* stats, float* out, void* stream) {
    if (int rc = dims_ok("glam_triplet_layer_fwd_ell", Cp, H, Dp, Cp, Dp)) return rc;
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_triplet_layer_fwd_ell: N out of range");
    if (N == 0) return GLAM_OK;
    GLAM_REQUIRE(staged && ell_src && ell_eid && xw && a_ij && (!aggr) == (!stats) && out && (E == 0 || edge_attr),
                 "glam_triplet_layer_fwd_ell: null pointer");      // (aggr = stats = NULL: the inference forward, nothing kept for a backward pass)
    GLAM_REQUIRE(aligned16(x) && aligned16(xw) && aligned16(a_ij) && aligned16(aggr) && aligned16(out) && aligned16(staged) &&
                     aligned16(edge_attr) && aligned16(stats), "glam_triplet_layer_fwd_ell: 16-byte alignment");
    if (!(triplet_fwd_ws_enabled() && triplet_fwd_ws_supported(H, Cp, Dp, edge_onehot)))
        return fail(GLAM_E_UNSUPPORTED, "glam_triplet_layer_fwd_ell: the ELL route needs one-hot edge features of width 4, 36 <= Cp <= 64, H*Cp <= 192 "
                    "(H=%d Cp=%d Dp=%d onehot=%d; glam_triplet_layer_ws_supported)", H, Cp, Dp, edge_onehot);
    hipStream_t s = (hipStream_t)stream;
    const int HC = H * Cp;
    const Staged L = staged_layout(H, Cp, Dp);
    if (x) {      // (x = NULL: xw and a_ij hold the node product already — glam_gru_ws_*_fwd_pre_node wrote them with the rows themselves)
        TsArgs g1{x, Cp, Cp, nullptr, 0, 0, staged + L.img_node, nullptr, xw, HC, HC, a_ij, 8, 8, (int)N};
        if (int rc = launch_ts_gemm(g1, s)) return rc;
    }
    return triplet_fwd_ws(xw, a_ij, edge_attr, staged + L.we_p, staged + L.m, ell_src, ell_eid, N, E, H, Cp, Dp, slope, edge_onehot,
                          aggr, stats, staged + L.img_upd, staged + L.bias_p, out, s);
}

// where the k_ts_gemm image of [W_node | Wa] (K = Cp, H*Cp + 8 columns) sits inside the staged buffer, in floats: what
// glam_gru_ws_*_fwd_pre_node takes as node_img
extern "C" size_t glam_triplet_staged_node_image(int H, int Cp, int Dp) { return staged_layout(H, Cp, Dp).img_node; }
// ... and of the same matrix as the pre-split operand fragments of that launch's producer waves (72 KB; (size_t)-1: none for this shape)
extern "C" size_t glam_triplet_staged_node_fragments(int H, int Cp, int Dp) {
    const Staged L = staged_layout(H, Cp, Dp);
    return L.total > L.node_pre ? L.node_pre : (size_t)-1;
}

extern "C" int glam_triplet_layer_ws_supported(int H, int Cp, int Dp, int edge_onehot) {
    return (triplet_fwd_ws_enabled() && triplet_fwd_ws_supported(H, Cp, Dp, edge_onehot)) ? 1 : 0;
}

// does glam_triplet_layer_fwd take aggr = stats = NULL for this shape (the ELL route always does)?
extern "C" int glam_triplet_layer_infer_supported(int H, int Cp, int Dp) { return triplet_fwd_can_fuse_update(H, Cp, Dp) ? 1 : 0; }

extern "C" size_t glam_triplet_layer_bwd_workspace_bytes(int64_t N, int64_t E, int H, int Cp, int Dp) {
    const size_t HC = (size_t)H * Cp;
    return (2 * (size_t)N * HC + (size_t)N * 8 + 2 * wgrad_workspace_floats()) * sizeof(float) +
           glam_triplet_bwd_workspace_bytes(N, E, H, Cp, Dp) + 1024;
}

namespace {
struct ParamOut {   // raw parameters and their gradient buffers; null d_wn: the caller wants `dstaged` instead
    const float* wn; const float* we; const float* att; int C, De;
    float* d_wn; float* d_we; float* d_att; float* d_wsc; float* d_bias;
    const float* c_wn; const float* c_we; const float* c_att; const float* c_wsc; const float* c_bias;   // optional addends (carry)
};
}  // namespace

static int layer_bwd_impl(const float* x, const float* edge_attr, const float* staged, const float* xw,
                          const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                          const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                          const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                          int64_t E, int H, int Cp, int Dp, float slope, float* d_x, float* dstaged,
                          float* d_edge_attr, void* ws, size_t ws_bytes, void* stream, const ParamOut* po,
                          const int32_t* ell_dst = nullptr, const int32_t* ell_eid_t = nullptr, int edge_onehot = 0,
                          const int32_t* ell_src = nullptr, const int32_t* ell_eid = nullptr, const float* dx_addend = nullptr,
                          int64_t* defer_info = nullptr) {
    if (int rc = dims_ok("glam_triplet_layer_bwd", Cp, H, Dp, Cp, Dp)) return rc;
    GLAM_REQUIRE(N >= 0 && N < INT32_MAX, "glam_triplet_layer_bwd: N out of range");
    GLAM_REQUIRE(x && staged && xw && a_ij && aggr && stats && d_out && d_x && (dstaged || po) && ws, "glam_triplet_layer_bwd: null pointer");
    GLAM_REQUIRE(ws_bytes >= glam_triplet_layer_bwd_workspace_bytes(N, E, H, Cp, Dp), "glam_triplet_layer_bwd: workspace too small");
    GLAM_REQUIRE(aligned16(x) && aligned16(d_out) && aligned16(aggr) && aligned16(d_x) && aligned16(staged) && aligned16(dstaged),
                 "glam_triplet_layer_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const int HC = H * Cp;
    const Staged L = staged_layout(H, Cp, Dp);
    const DStaged G = dstaged_layout(H, Cp, Dp);
    if (!dstaged) dstaged = reinterpret_cast<float*>(ws);   // never written on the `po` path (offsets below stay in range)
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255;
    float* d_aggr = reinterpret_cast<float*>(base);
    float* d_xw = d_aggr + (size_t)N * HC;
    float* d_a = d_xw + (size_t)N * HC;
    float* wg1 = d_a + (size_t)N * 8;
    float* wg2 = wg1 + wgrad_workspace_floats();
    void* tws = wg2 + wgrad_workspace_floats();
    const size_t tws_bytes = ws_bytes - (reinterpret_cast<uintptr_t>(tws) - reinterpret_cast<uintptr_t>(ws));
    ReduceArgs ra{};
    ra.njobs = 3;

    WgArgs w1{aggr, HC, HC, nullptr, 0, 0, 1, d_out, Cp, Cp, 0, (int)N, 0, wg1, 0, 0};
    WgArgs w2{d_xw, HC, HC, d_a, 8, 8, 0, x, Cp, Cp, 0, (int)N, 0, wg2, 0, 0};
    // main chain: d_aggr = d_out @ Ws_p^T -> B1 -> B2
    // (inside B1 where a fused variant exists: one launch and one [N, HC] round trip less)
    static const bool fuse_dagg_on = true;
    const bool fuse_dagg = fuse_dagg_on && triplet_bwd_can_fuse_dagg(H, Cp, Dp);
    if (!fuse_dagg) {
        TsArgs g1{d_out, Cp, Cp, nullptr, 0, 0, staged + L.img_dagg, nullptr, d_aggr, HC, HC, nullptr, 0, 0, (int)N};
        if (int rc = launch_ts_gemm(g1, s)) return rc;
    }
    const float* tpart = nullptr;
    int tnblk = 0;
    // d_x = [d_xw | d_a] @ Wcat^T inside B2: warp-specialised over the caller's ELL records by source, or as the general kernel's epilogue
    const bool ws_dx = ell_dst && ell_eid_t && Cp <= 64 && triplet_bwd_src_ws_supported(H, Cp, Dp, edge_onehot);
    const bool fuse_dx = ws_dx || (triplet_bwd_can_fuse_dx(H, Cp, Dp) && N <= fuse_max_nodes());
    if (int rc = triplet_bwd_impl(xw, a_ij, edge_attr, staged + L.we_p, staged + L.m, aggr, stats, d_aggr, rowptr, src, eid,
                                  colptr, dst, eid_t, N, E, H, Cp, Dp, 1, slope, d_xw, d_a, dstaged + G.d_we_p,
                                  dstaged + G.d_m, d_edge_attr, tws, tws_bytes, s, false, &tpart, &tnblk,
                                  fuse_dx ? staged + L.img_dx : nullptr, fuse_dx ? d_x : nullptr,
                                  fuse_dagg ? staged + L.img_dagg : nullptr, fuse_dagg ? d_out : nullptr, ws_dx ? ell_dst : nullptr,
                                  ws_dx ? ell_eid_t : nullptr, edge_onehot, ell_src, ell_eid, dx_addend,
                                  (fuse_dagg && L.total > L.dagg_pre) ? staged + L.dagg_pre : nullptr))
        return rc;
    const int WSZ = Dp * HC;
    if (defer_info) {
        // the data half only: the parameter half (both weight-gradient products + k_param_grads) runs later, over the operand sets of all
        // applications of the layer (glam_triplet_layer_param_grads_sets), which finds this application's pieces in `ws` through these
        if (!fuse_dx) {
            TsArgs g2{d_xw, HC, HC, d_a, 8, 8, staged + L.img_dx, nullptr, d_x, Cp, Cp, nullptr, 0, 0, (int)N};
            if (int rc = launch_ts_gemm(g2, s)) return rc;
        }
        defer_info[0] = (int64_t)(reinterpret_cast<const char*>(tpart) - reinterpret_cast<const char*>(ws));
        defer_info[1] = tnblk;
        defer_info[2] = (int64_t)(reinterpret_cast<const char*>(d_xw) - reinterpret_cast<const char*>(ws));
        defer_info[3] = (int64_t)(reinterpret_cast<const char*>(d_a) - reinterpret_cast<const char*>(ws));
        return GLAM_OK;
    }
    ra.job[1] = ReduceJob{1, tpart, tnblk, WSZ + Dp * 4, 0, 0, 0, 0, dstaged + G.d_we_p, dstaged + G.d_m, WSZ, 0};
    //   d_WsB[HC+1, Cp] = [aggr | 1]^T @ d_out (last row = d_bias)
    //   d_Wcat[Cp, HC+8] = x^T @ [d_xw | d_a], computed as ([d_xw|d_a]^T x)^T
    // (many_splits: from 32 768 rows on the products run warp-specialised on the bf16 matrix cores — k_wgrad_x3, one block per CU, 128
    //  partials per element; k_param_grads / k_final_reduce take any split count)
    if (int rc = launch_wgrad_partials2(w1, dstaged + G.d_wsb, Cp, 1, &ra.job[0], w2, dstaged + G.d_wcat, 1, HC + 8, &ra.job[2], s, true))
        return rc;      // both products in ONE launch
    // d_x = [d_xw | d_a] @ Wcat^T
    if (!fuse_dx) {
        TsArgs g2{d_xw, HC, HC, d_a, 8, 8, staged + L.img_dx, nullptr, d_x, Cp, Cp, nullptr, 0, 0, (int)N};
        if (int rc = launch_ts_gemm(g2, s)) return rc;
    }
    if (po) {   // parameter gradients straight from the partial sets: reductions + chain rule, no intermediate dstaged
        const int C = po->C, De = po->De;
        ParamGradArgs pg{ra.job[0].partial, ra.job[0].nsplit, ra.job[2].partial, ra.job[2].nsplit, tpart, tnblk, WSZ + Dp * 4,
                         po->wn, po->we, po->att, C, H, De, Cp, Dp, po->d_wn, po->d_we, po->d_att, po->d_wsc, po->d_bias,
                         (HC + 1 + 63) / 64 * 16, (HC + 8 + 63) / 64 * 16, 3 * H, po->c_wn, po->c_we, po->c_att, po->c_wsc, po->c_bias};
        pg.ns3_each = tnblk;
        const int blocksD = (WSZ / 4 + 7) / 8;
        GLAM_PROF_LABEL("k_param_grads");
        hipLaunchKernelGGL(k_param_grads<false>, dim3(pg.blocksA + pg.blocksB + pg.blocksC + blocksD), dim3(kBlock), 0, s, pg);
        GLAM_LAUNCH_CHECK("glam_triplet_layer_bwd(param grads)");
        return GLAM_OK;
    }
    // one fixed-order reduction for the three partial sets (d_W_scale|d_bias, d_W_edge|d_M, d_Wcat)
    return launch_final_reduce(ra, s);
}

// an empty batch (N = 0, e.g. an empty shard of a data-parallel step): every parameter gradient is zero
static int zero_param_grads(int C, int H, int De, float* d_wn, float* d_we, float* d_att, float* d_wsc, float* d_bias, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(d_wn, 0, (size_t)C * H * C * sizeof(float), s);
    (void)hipMemsetAsync(d_we, 0, (size_t)De * H * C * sizeof(float), s);
    (void)hipMemsetAsync(d_att, 0, (size_t)H * 3 * C * sizeof(float), s);
    (void)hipMemsetAsync(d_wsc, 0, (size_t)H * C * C * sizeof(float), s);
    (void)hipMemsetAsync(d_bias, 0, (size_t)C * sizeof(float), s);
    GLAM_LAUNCH_CHECK("glam_triplet_layer_bwd_params(N = 0)");
    return GLAM_OK;
}

extern "C" int glam_triplet_layer_bwd(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                      const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                      const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                      const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                                      int64_t E, int H, int Cp, int Dp, float slope, float* d_x, float* dstaged,
                                      float* d_edge_attr, void* ws, size_t ws_bytes, void* stream) {
    GLAM_REQUIRE(dstaged, "glam_triplet_layer_bwd: null pointer");
    return layer_bwd_impl(x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr, dst, eid_t, N, E, H, Cp,
                          Dp, slope, d_x, dstaged, d_edge_attr, ws, ws_bytes, stream, nullptr);
}

extern "C" int glam_triplet_layer_bwd_params(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                             const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                             const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                             const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                                             int64_t E, int C, int H, int De, int Cp, int Dp, float slope,
                                             const float* weight_node, const float* weight_edge, const float* att, float* d_x,
                                             float* d_weight_node, float* d_weight_edge, float* d_att, float* d_weight_scale,
                                             float* d_bias, float* d_edge_attr, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = dims_ok("glam_triplet_layer_bwd_params", C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(weight_node && weight_edge && att && d_weight_node && d_weight_edge && d_att && d_weight_scale && d_bias,
                 "glam_triplet_layer_bwd_params: null pointer");
    if (N == 0) return zero_param_grads(C, H, De, d_weight_node, d_weight_edge, d_att, d_weight_scale, d_bias, stream);
    const ParamOut po{weight_node, weight_edge, att, C, De, d_weight_node, d_weight_edge, d_att, d_weight_scale, d_bias};
    return layer_bwd_impl(x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr, dst, eid_t, N, E, H, Cp,
                          Dp, slope, d_x, nullptr, d_edge_attr, ws, ws_bytes, stream, &po);
}

// The same with addends for the five parameter gradients (each laid out like its output, any of them NULL): out = gradient + addend.
// A layer applied message_steps times with shared weights (src_1gp/model.py:53-54) hands the gradient accumulated by its later
// applications in here, so that the sum costs no launch of its own.
extern "C" int glam_triplet_layer_bwd_params_acc(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                                 const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                                 const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                                 const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                                                 int64_t E, int C, int H, int De, int Cp, int Dp, float slope,
                                                 const float* weight_node, const float* weight_edge, const float* att, float* d_x,
                                                 float* d_weight_node, float* d_weight_edge, float* d_att, float* d_weight_scale,
                                                 float* d_bias, const float* add_weight_node, const float* add_weight_edge,
                                                 const float* add_att, const float* add_weight_scale, const float* add_bias,
                                                 const int32_t* ell_dst, const int32_t* ell_eid_t, int edge_onehot,
                                                 float* d_edge_attr, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = dims_ok("glam_triplet_layer_bwd_params_acc", C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(weight_node && weight_edge && att && d_weight_node && d_weight_edge && d_att && d_weight_scale && d_bias,
                 "glam_triplet_layer_bwd_params_acc: null pointer");
    GLAM_REQUIRE(N > 0, "glam_triplet_layer_bwd_params_acc: N = 0 (add on the host side)");
    const ParamOut po{weight_node, weight_edge, att, C, De, d_weight_node, d_weight_edge, d_att, d_weight_scale, d_bias,
                      add_weight_node, add_weight_edge, add_att, add_weight_scale, add_bias};
    return layer_bwd_impl(x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr, dst, eid_t, N, E, H, Cp,
                          Dp, slope, d_x, nullptr, d_edge_attr, ws, ws_bytes, stream, &po, ell_dst, ell_eid_t, edge_onehot);
}

// The same with the ELL index records of BOTH directions (glam_ell_build on the CSR by target and on its transpose; either pair may be
// NULL): molecular graphs with one-hot bond features then run B1 and B2 + d_x on the warp-specialised kernels (csrc/triplet_ws*.hip).
static int layer_bwd_params_ell(const char* fn, const float* x, const float* edge_attr, const float* staged, const float* xw,
                                                 const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                                 const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                                 const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                                                 int64_t E, int C, int H, int De, int Cp, int Dp, float slope,
                                                 const float* weight_node, const float* weight_edge, const float* att, float* d_x,
                                                 float* d_weight_node, float* d_weight_edge, float* d_att, float* d_weight_scale,
                                                 float* d_bias, const float* add_weight_node, const float* add_weight_edge,
                                                 const float* add_att, const float* add_weight_scale, const float* add_bias,
                                                 const int32_t* ell_src, const int32_t* ell_eid, const int32_t* ell_dst,
                                                 const int32_t* ell_eid_t, int edge_onehot, float* d_edge_attr, void* ws, size_t ws_bytes,
                                                 void* stream, const float* dx_addend) {
    if (int rc = dims_ok(fn, C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(weight_node && weight_edge && att && d_weight_node && d_weight_edge && d_att && d_weight_scale && d_bias,
                 "%s: null pointer", fn);
    GLAM_REQUIRE(N > 0, "%s: N = 0 (add on the host side)", fn);
    GLAM_REQUIRE((!ell_src) == (!ell_eid) && (!ell_dst) == (!ell_eid_t) && aligned16(ell_src) && aligned16(ell_eid) && aligned16(ell_dst) &&
                     aligned16(ell_eid_t), "%s: ELL tables come in 16-byte aligned pairs", fn);
    const ParamOut po{weight_node, weight_edge, att, C, De, d_weight_node, d_weight_edge, d_att, d_weight_scale, d_bias,
                      add_weight_node, add_weight_edge, add_att, add_weight_scale, add_bias};
    return layer_bwd_impl(x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr, dst, eid_t, N, E, H, Cp,
                          Dp, slope, d_x, nullptr, d_edge_attr, ws, ws_bytes, stream, &po, ell_dst, ell_eid_t, edge_onehot, ell_src, ell_eid,
                          dx_addend);
}

extern "C" int glam_triplet_layer_bwd_params_ell(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                                 const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                                 const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                                 const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                                                 int64_t E, int C, int H, int De, int Cp, int Dp, float slope,
                                                 const float* weight_node, const float* weight_edge, const float* att, float* d_x,
                                                 float* d_weight_node, float* d_weight_edge, float* d_att, float* d_weight_scale,
                                                 float* d_bias, const float* add_weight_node, const float* add_weight_edge,
                                                 const float* add_att, const float* add_weight_scale, const float* add_bias,
                                                 const int32_t* ell_src, const int32_t* ell_eid, const int32_t* ell_dst,
                                                 const int32_t* ell_eid_t, int edge_onehot, float* d_edge_attr, void* ws, size_t ws_bytes,
                                                 void* stream) {
    return layer_bwd_params_ell("glam_triplet_layer_bwd_params_ell", x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr, dst,
                                eid_t, N, E, C, H, De, Cp, Dp, slope, weight_node, weight_edge, att, d_x, d_weight_node, d_weight_edge, d_att,
                                d_weight_scale, d_bias, add_weight_node, add_weight_edge, add_att, add_weight_scale, add_bias, ell_src, ell_eid,
                                ell_dst, ell_eid_t, edge_onehot, d_edge_attr, ws, ws_bytes, stream, nullptr);
}

// ... with a second gradient path into the layer's input: d_x = (the layer's input gradient) + d_x_addend, summed in the d_x product's
// epilogue.  Warp-specialised route only (glam_triplet_layer_ws_supported, both ELL pairs given): GLAM_E_UNSUPPORTED otherwise.
extern "C" int glam_triplet_layer_bwd_params_ell_add(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                                     const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                                     const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                                     const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N,
                                                     int64_t E, int C, int H, int De, int Cp, int Dp, float slope,
                                                     const float* weight_node, const float* weight_edge, const float* att, float* d_x,
                                                     float* d_weight_node, float* d_weight_edge, float* d_att, float* d_weight_scale,
                                                     float* d_bias, const float* add_weight_node, const float* add_weight_edge,
                                                     const float* add_att, const float* add_weight_scale, const float* add_bias,
                                                     const int32_t* ell_src, const int32_t* ell_eid, const int32_t* ell_dst,
                                                     const int32_t* ell_eid_t, int edge_onehot, float* d_edge_attr, void* ws, size_t ws_bytes,
                                                     const float* d_x_addend, void* stream) {
    GLAM_REQUIRE(aligned16(d_x_addend), "glam_triplet_layer_bwd_params_ell_add: d_x_addend must be 16-byte aligned");
    return layer_bwd_params_ell("glam_triplet_layer_bwd_params_ell_add", x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr,
                                dst, eid_t, N, E, C, H, De, Cp, Dp, slope, weight_node, weight_edge, att, d_x, d_weight_node, d_weight_edge,
                                d_att, d_weight_scale, d_bias, add_weight_node, add_weight_edge, add_att, add_weight_scale, add_bias, ell_src,
                                ell_eid, ell_dst, ell_eid_t, edge_onehot, d_edge_attr, ws, ws_bytes, stream, d_x_addend);
}

// ---- a layer applied message_steps times with shared weights (src_1gp/model.py:53-54): every application's backward runs its DATA half
//      (d_x: B1, B2 with the d_x product) and leaves its operands in its workspace; the PARAMETER half — both weight-gradient products
//      and k_param_grads — runs once, over the operand sets of up to three applications ----
extern "C" int glam_triplet_layer_bwd_data_ell(const float* x, const float* edge_attr, const float* staged, const float* xw, const float* a_ij,
                                               const float* aggr, const float* stats, const float* d_out, const int32_t* rowptr,
                                               const int32_t* src, const int32_t* eid, const int32_t* colptr, const int32_t* dst,
                                               const int32_t* eid_t, int64_t N, int64_t E, int C, int H, int De, int Cp, int Dp, float slope,
                                               float* d_x, const int32_t* ell_src, const int32_t* ell_eid, const int32_t* ell_dst,
                                               const int32_t* ell_eid_t, int edge_onehot, float* d_edge_attr, void* ws, size_t ws_bytes,
                                               const float* d_x_addend, int64_t* info, void* stream) {
    const char* fn = "glam_triplet_layer_bwd_data_ell";
    if (int rc = dims_ok(fn, C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(N > 0 && info, "%s: N = 0 / null info", fn);
    GLAM_REQUIRE((!ell_src) == (!ell_eid) && (!ell_dst) == (!ell_eid_t) && aligned16(ell_src) && aligned16(ell_eid) && aligned16(ell_dst) &&
                     aligned16(ell_eid_t) && aligned16(d_x_addend), "%s: ELL tables come in 16-byte aligned pairs", fn);
    const ParamOut po{};
    return layer_bwd_impl(x, edge_attr, staged, xw, a_ij, aggr, stats, d_out, rowptr, src, eid, colptr, dst, eid_t, N, E, H, Cp, Dp, slope, d_x,
                          nullptr, d_edge_attr, ws, ws_bytes, stream, &po, ell_dst, ell_eid_t, edge_onehot, ell_src, ell_eid, d_x_addend, info);
}

extern "C" int glam_triplet_layer_param_grads_sets(int nseg, const void* const* ws_set, const int64_t* info, const float* const* x,
                                                   const float* const* aggr, const float* const* d_out, int64_t N, int C, int H, int De, int Cp,
                                                   int Dp, const float* weight_node, const float* weight_edge, const float* att,
                                                   float* d_weight_node, float* d_weight_edge, float* d_att, float* d_weight_scale, float* d_bias,
                                                   const float* add_weight_node, const float* add_weight_edge, const float* add_att,
                                                   const float* add_weight_scale, const float* add_bias, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "glam_triplet_layer_param_grads_sets";
    if (int rc = dims_ok(fn, C, H, De, Cp, Dp)) return rc;
    GLAM_REQUIRE(nseg >= 1 && nseg <= 3, "%s: %d operand sets (1..3)", fn, nseg);
    GLAM_REQUIRE(ws_set && info && x && aggr && d_out && weight_node && weight_edge && att && d_weight_node && d_weight_edge && d_att &&
                     d_weight_scale && d_bias && ws, "%s: null pointer", fn);
    GLAM_REQUIRE(N > 0 && N * nseg < INT32_MAX, "%s: N out of range", fn);
    GLAM_REQUIRE(ws_bytes >= 2 * glam_wgrad_workspace_bytes(), "%s: workspace too small (two products)", fn);
    const int HC = H * Cp, WSZ = Dp * HC;
    const int tnblk = (int)info[1];
    const float* tp[3] = {nullptr, nullptr, nullptr};
    const float* dxw[3] = {nullptr, nullptr, nullptr};
    const float* da[3] = {nullptr, nullptr, nullptr};
    for (int q = 0; q < nseg; ++q) {
        GLAM_REQUIRE(ws_set[q] && x[q] && aggr[q] && d_out[q] && aligned16(x[q]) && aligned16(aggr[q]) && aligned16(d_out[q]),
                     "%s: operand set %d: null / misaligned pointer", fn, q);
        GLAM_REQUIRE(info[4 * q + 1] == tnblk, "%s: the sets come from launches of different grids", fn);
        const char* b = reinterpret_cast<const char*>(ws_set[q]);
        tp[q] = reinterpret_cast<const float*>(b + info[4 * q]);
        dxw[q] = reinterpret_cast<const float*>(b + info[4 * q + 2]);
        da[q] = reinterpret_cast<const float*>(b + info[4 * q + 3]);
    }
    float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    //   d_WsB[HC+1, Cp] = [aggr | 1]^T d_out,   d_Wcat[Cp, HC+8] = x^T [d_xw | d_a] (as its transpose), each summed over the sets
    WgArgs w1{aggr[0], HC, HC, nullptr, 0, 0, 1, d_out[0], Cp, Cp, 0, (int)(N * nseg), 0, partial, 0, 0};
    WgArgs w2{dxw[0], HC, HC, da[0], 8, 8, 0, x[0], Cp, Cp, 0, (int)(N * nseg), 0, partial + wgrad_workspace_floats(), 0, 0};
    w1.nseg = w2.nseg = nseg;
    w1.seg_rows = w2.seg_rows = (int)N;
    for (int q = 1; q < nseg; ++q) {
        w1.segP1[q - 1] = aggr[q]; w1.segQ[q - 1] = d_out[q];
        w2.segP1[q - 1] = dxw[q]; w2.segP2[q - 1] = da[q]; w2.segQ[q - 1] = x[q];
    }
    hipStream_t s = (hipStream_t)stream;
    ReduceJob j1{}, j2{};
    if (int rc = launch_wgrad_partials2(w1, nullptr, Cp, 1, &j1, w2, nullptr, 1, HC + 8, &j2, s, true)) return rc;
    ParamGradArgs pg{j1.partial, j1.nsplit, j2.partial, j2.nsplit, tp[0], tnblk * nseg, WSZ + Dp * 4, weight_node, weight_edge, att, C, H, De, Cp,
                     Dp, d_weight_node, d_weight_edge, d_att, d_weight_scale, d_bias, (HC + 1 + 63) / 64 * 16, (HC + 8 + 63) / 64 * 16, 3 * H,
                     add_weight_node, add_weight_edge, add_att, add_weight_scale, add_bias};
    pg.p3b = tp[1]; pg.p3c = tp[2]; pg.ns3_each = tnblk;
    const int blocksD = (WSZ / 4 + 7) / 8;
    GLAM_PROF_LABEL("k_param_grads<sets>");
    if (nseg == 1) hipLaunchKernelGGL(k_param_grads<false>, dim3(pg.blocksA + pg.blocksB + pg.blocksC + blocksD), dim3(kBlock), 0, s, pg);
    else hipLaunchKernelGGL(k_param_grads<true>, dim3(pg.blocksA + pg.blocksB + pg.blocksC + blocksD), dim3(kBlock), 0, s, pg);
    GLAM_LAUNCH_CHECK(fn);
    return GLAM_OK;
}
