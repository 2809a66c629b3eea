// Internal interfaces between gemm.hip (dense MFMA kernels), triplet.hip (aggregate kernels) and
// layer.hip (whole-layer sequencing).  Not part of the C ABI.
#pragma once
#include "common.h"
#include "bf16x3.h"

namespace glam {

struct TsArgs {
    const float* A1; int K1; int lda1;
    const float* A2; int K2; int lda2;
    const float* Wimg;                     // LDS image of W (see k_ts_make_image)
    const float* bias;                     // added to out1 columns, may be null
    float* out1; int M1; int ldo1;         // columns [0, M1)   (M1 % 4 == 0)
    float* out2; int M2; int ldo2;         // columns [M1, M1+M2), may be null
    int N;
    // optional CELU(alpha = 1) folding for the GRU gate linears of MessageBlock (src_1gp/layer.py:261-262):
    int a_celu;                            // 1: the GEMM consumes celu(A) instead of A
    const float* cgrad_src; int ld_cgrad;  // non-null: out1[r, c] *= celu'(cgrad_src[r, c]) (chain rule through a folded CELU)
    const float* addend; int ld_add;       // non-null: out1[r, c] += addend[r, c] last (a second gradient path into the same tensor)
    int out_relu;                          // 1: out = max(out, 0) — the ReLU of a LinearBlock (src_1gp/layer.py:236) in the epilogue (k_tall_x3 only)
    // rng_state non-null (k_tall_x3, one product per launch, out1 contiguous: ldo1 = M1, no out2): the training-mode RReLU(rr_lo, rr_hi)
    // of a LinearBlock in the epilogue — and, out_drop non-null, the dropped twin Dropout(drop_p)(out) the next block starts with — on
    // the Philox words glam_bias_res_act_rng_fwd draws for the same elements of the same stream position (rng.h)
    long long* rng_state; long long* rng_eff; float rr_lo, rr_hi, drop_p; float* out_drop;
    // node_pre non-null (k_tall_x3<2, 4, 1>, one product per launch, M1 <= 64): out1's rows are the input of a TripletMessage — the producers
    // also write node_xw[N, node_m1] | node_a[N, 8] = out1 (out_drop when given) @ [W_node | Wa] (node_product.h)
    const void* node_pre; float* node_xw; float* node_a; int node_m1;
};

// distance between the 64 x 64 partial slabs of k_wgrad (4096 floats of data each): 16 KB + 256 B, so that the splits of one element —
// which the reductions read 40 at a time — fall on different memory channels instead of every fourth one
constexpr int kWgSlabStride = 4096 + 64;

struct WgArgs {
    const float* P1; int I1; int ldp1;
    const float* P2; int I2; int ldp2;
    int ones;                              // 1: virtual all-ones column at index I1+I2 (bias gradient)
    const float* Q; int J; int ldq;        // J % 4 == 0, J <= 64
    int qones;                             // 1: virtual all-ones column at index J (needs J + 1 <= 64)
    int N; int rows_per_wave;              // multiple of 4
    float* partial;                        // [slab][nsplit][16 tiles][64 lanes][4]
    int nsplit; int ntile;
    int q_celu;                            // 1: the product uses celu(Q) (weight gradient of a linear fed through a folded CELU)
    // nseg > 1 (k_wgrad<.., SEG>): nseg operand sets of seg_rows rows each, summed into ONE product (N = nseg * seg_rows): set 0 is
    // (P1, P2, Q), set s > 0 is (segP1[s - 1], segP2[s - 1], segQ[s - 1]) with the same widths and row strides
    int nseg; int seg_rows;
    const float* segP1[2]; const float* segP2[2]; const float* segQ[2];
    int rows_per_split;                    // k_wgrad_x3 (wgrad_x3.hip): rows of a block, a multiple of 32 (plan_wgrad_x3)
    const float* pmask;                    // k_wgrad<.., PMASK>: P1 is used as P1 * (pmask > 0), pmask laid out like P1 (the ReLU behind a linear:
                                           // its backward folded into the weight-gradient product, glam_wgrad_gemm_split_relu)
};

// Two independent products may share one launch (blocks [0, first_b) work on job a, the rest on job b).
struct WgArgs2 { WgArgs a, b; int first_b; };

// see k_final_reduce in gemm.hip
struct ReduceJob {
    int kind; const float* partial; int nsplit; int n; int I, J, si, sj; float* out; float* out2; int split_at;
    int first_block;
    const float* addend;                   // kind 0, non-null: out[i, j] = sum + addend[i, j] (same strides: a gradient carry)
    int jw;                                // kind 0, > 0: only columns j < jw of `out` exist (a Q whose last columns are padding)
    float* out_b; const float* add_b;      // kind 0, out_b non-null: the LAST column (j = J - 1, the bias gradient of a [Q | 1]
                                           // product) goes to out_b[i] (+ add_b[i]) instead of out: weight and bias gradients land
                                           // in separate contiguous tensors (autograd takes them without a copy)
};
struct ReduceArgs { ReduceJob job[3]; int njobs; };

// Up to six images in one launch (the forward and transposed images of a GRU's two gate matrices change together after
// every optimizer step): job j owns blocks [first[j], first[j+1]).
// gate > 0 (the fused GRU step's images, block.hip): M = 3 gates of `gate` channels, each padded to 64 columns — logical column m holds
// channel m % 64 of gate m / 64 (source row (m / 64) * gate + m % 64 of the [3 * gate, K] matrix), k padded to 64.
// gate < 0 (the warp-specialised GRU step's pre-split images, block.hip): matrix M (0: W_ih, 1: W_hh) of a GRU with K channels into its
// 24 fragments of the forward (gate = -1) or the backward (gate = -2) image `img` — gru_pre_fragment below.
struct ImageJob { const float* W; int ldw, transW, K, M, MT; float* img; int first; int gate; };
constexpr int kGruPreFrags = 48, kGruPreBytes = kGruPreFrags * 3 * 1024;
// The gate matrices as the matrix waves of k_gru_fwd_ws / k_gru_bwd_ws hold them: every operand fragment split into its three bf16 terms
// ONCE per weight update, stored in lane order (fragment f, term t, lane l -> 16 bytes at ((3 f + t) * 64 + l) * 16) — the values
// split8 produces in the kernels' own prologue, which 256 blocks x 6 launches of a training step otherwise each redo.
//   forward image : f = 12 w + 2 (2 g + s) + m     wave w (channels 16 w ..), gate g, k step s, matrix m (0: W_ih, 1: W_hh);
//                   lane (c, kb) holds W_m[g C + 16 w + c][32 s + 8 kb .. + 7]
//   backward image: f = 6 (4 m + ct) + s            wave 4 m + ct (columns 16 ct ..), s = 2 gate + k step of the gate's C rows;
//                   lane (c, kb) holds W_m[(s >> 1) C + 32 (s & 1) + 8 kb .. + 7][16 ct + c]
// (zero where the channel / column index reaches C).  fl = 0 .. 23: the matrix's fragments (forward: 6 w + 2 g + s, backward: 6 ct + s).
__device__ __forceinline__ void gru_pre_fragment(const float* W, int C, bool bwd, int m, int fl, int lane, char* dst) {
    const int c = lane & 15, kb = lane >> 4;
    float v[8];
    int f;
    if (!bwd) {
        const int w = fl / 6, gs = fl % 6, g = gs >> 1, s = gs & 1, ch = 16 * w + c;
        f = 2 * fl + m;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = 32 * s + 8 * kb + j; v[j] = (ch < C && k < C) ? W[(size_t)(g * C + ch) * C + k] : 0.f; }
    } else {
        const int ct = fl / 6, s = fl % 6, col = 16 * ct + c;
        f = 24 * m + fl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int ch = 32 * (s & 1) + 8 * kb + j; v[j] = (col < C && ch < C) ? W[(size_t)((s >> 1) * C + ch) * C + col] : 0.f; }
    }
    const Bf16x3 x = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    char* o = dst + (size_t)f * 3072 + lane * 16;
    *reinterpret_cast<bf16x8_t*>(o) = x.hi; *reinterpret_cast<bf16x8_t*>(o + 1024) = x.mid; *reinterpret_cast<bf16x8_t*>(o + 2048) = x.lo;
}
constexpr int kMaxImageJobs = 6;
struct ImageJobs { ImageJob job[kMaxImageJobs]; int njobs; };
__device__ __forceinline__ void make_images_block(const ImageJobs& js, int bid) {
    int jb = 0;
#pragma unroll
    for (int q = 1; q < kMaxImageJobs; ++q)
        if (q < js.njobs && bid >= js.job[q].first) jb = q;
    const ImageJob& J = js.job[jb];
    const int idx = (bid - J.first) * kBlock + threadIdx.x;
    if (J.gate < 0) {
        if (idx < 24 * 64) gru_pre_fragment(J.W, J.K, J.gate == -2, J.M, idx >> 6, idx & 63, reinterpret_cast<char*>(J.img));
        return;
    }
    const int MP = J.MT * 16, Kp = J.gate ? 64 : (J.K + 15) & ~15;
    if (idx >= Kp * MP) return;
    const int j = idx & 3, p = (idx >> 2) % MP, k = (idx >> 2) / MP * 4 + j;
    const int m = ts_col_of_pos(p);
    float v = 0.f;
    if (J.gate) {
        const int g = m >> 6, ch = m & 63;
        if (k < J.K && ch < J.gate) v = J.W[(size_t)(g * J.gate + ch) * J.ldw + k];
    } else if (k < J.K && m < J.M) {
        v = J.transW ? J.W[(size_t)m * J.ldw + k] : J.W[(size_t)k * J.ldw + m];
    }
    J.img[idx] = v;
}
int image_job(ImageJob& j, const char* fn, const float* W, int ldw, int transW, int K, int M, float* img, int first);   // gemm.hip

size_t ts_image_floats(int K, int M);
int launch_ts_make_image(const float* W, int ldw, int transW, int K, int M, float* img, hipStream_t s);
int launch_ts_gemm(const TsArgs& a, hipStream_t s);
int launch_tall_x3(const TsArgs& a, const TsArgs* b, int variant, hipStream_t s);    // tall_x3.hip: warp-specialised 3 x bf16 products
int launch_ts_gemm2(const TsArgs& a, const TsArgs* b, hipStream_t s);   // b: a second product of the same variant in the same launch
size_t wgrad_workspace_floats();
// many_splits: the reduction behind the launch takes any split count in its stride (k_final_reduce) — k_wgrad_x3 leaves one partial
// per CU and product, 128 or 256 per element, against k_wgrad's 40; k_param_grads (layer.hip) reads them in batches of 40 from
// about a hundred blocks and loses more than the product launch gains (B = 1 024: 4.4 -> 12.7 us), so the layer keeps k_wgrad
int launch_wgrad_partials(WgArgs a, float* out, int si, int sj, hipStream_t s, ReduceJob* job, bool many_splits = true);
int launch_wgrad_partials2(WgArgs a, float* out_a, int si_a, int sj_a, ReduceJob* job_a, WgArgs b, float* out_b, int si_b,
                           int sj_b, ReduceJob* job_b, hipStream_t s, bool many_splits = true);
int launch_final_reduce(ReduceArgs ra, hipStream_t s);
int launch_wgrad_x3(const WgArgs2& two, int blocks, hipStream_t s);     // wgrad_x3.hip: the same products, warp-specialised on the bf16 matrix cores

bool triplet_fwd_can_fuse_update(int H, int Cp, int De);
int triplet_fwd_fused_update(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge,
                             const float* M, const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t N,
                             int64_t E, int H, int Cp, int De, float slope, float* aggr, float* stats,
                             const float* img_upd, const float* bias_p, float* out, hipStream_t s);
// forward aggregate + update GEMM over ELL records, warp-specialised (triplet_ws.hip: producer waves gather, consumer waves run the
// update GEMM; bit-identical to triplet_fwd_fused_update)
int triplet_fwd_ws(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                   const int32_t* ell_src, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De, float slope,
                   int edge_onehot, float* aggr, float* stats, const float* img_upd, const float* bias_p, float* out, hipStream_t s);
bool triplet_fwd_ws_enabled();          // GLAM_FWD_WS (default 1)
bool triplet_fwd_ws_supported(int H, int Cp, int De, int edge_onehot);
int triplet_bwd_impl(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                     const float* aggr, const float* stats, const float* d_aggr, const int32_t* rowptr,
                     const int32_t* src, const int32_t* eid, const int32_t* colptr, const int32_t* dst,
                     const int32_t* eid_t, int64_t N, int64_t E, int H, int Cp, int De, int emul, float slope,
                     float* d_xw, float* d_a_ij, float* d_w_edge, float* d_M, float* d_edge_attr, void* ws,
                     size_t ws_bytes, hipStream_t s, bool reduce_now, const float** partial_out, int* nblk_out,
                     const float* img_dx, float* d_x,
                     const float* img_dagg = nullptr, const float* d_out = nullptr, const int32_t* ell_dst = nullptr,
                     const int32_t* ell_eid_t = nullptr, int edge_onehot = 0, const int32_t* ell_src = nullptr,
                     const int32_t* ell_eid = nullptr, const float* dx_addend = nullptr,      // dx_addend: warp-specialised B2 only
                     const float* dagg_pre = nullptr);      // W_scale^T as pre-split fragments of the warp-specialised B1's matrix waves (layer.hip: Staged)
// B1 with the d_aggr GEMM inside, warp-specialised (triplet_ws_b1.hip: matrix waves produce the d_aggr tiles ahead of the vector waves)
bool triplet_bwd_dst_ws_supported(int H, int Cp, int De, int edge_onehot);
int triplet_bwd_dst_ws(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M, const float* aggr,
                       const float* stats, const float* d_out, const float* img_dagg, const int32_t* ell_src, const int32_t* ell_eid,
                       int64_t N, int64_t E, int H, int Cp, int De, int edge_onehot, float slope, float* d_aggr, float* alpha_e,
                       float* dpre_e, float* d_a_ij, float* partial, int* nblk_out, hipStream_t s, const float* dagg_pre = nullptr);
// B2 + d_x = [d_xw | d_a] @ Wcat^T in one warp-specialised launch (triplet_ws.hip; one-hot edge features of width 4)
bool triplet_bwd_src_ws_supported(int H, int Cp, int De, int edge_onehot);
int triplet_bwd_src_ws(const float* d_aggr, const float* alpha_e, const float* dpre_e, const float* edge_attr, const float* w_edge,
                       const int32_t* ell_dst, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De, int edge_onehot,
                       float* d_xw, float* d_a_ij, const float* img_dx, float* d_x, hipStream_t s, const float* dx_addend = nullptr);
bool triplet_bwd_can_fuse_dx(int H, int Cp, int De);
bool triplet_bwd_can_fuse_dagg(int H, int Cp, int De);

}  // namespace glam
