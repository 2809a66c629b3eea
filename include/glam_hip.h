/*
 * glam_hip.h — C ABI of libglam_hip.so, the MI355X (gfx950) implementation of the GLAM
 * message-passing hot path (yvquanli/GLAM src_1gp/layer.py + model.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *   - tensors are dense, row-major, fp32 ("f32") or int32/int64 as typed;
 *   - `stream` is a hipStream_t passed as void*; every entry point only enqueues work on it
 *     (no host synchronisation, no allocation), so calls are hipGraph-capturable;
 *   - return value: 0 = ok, <0 = error (GLAM_E_*); glam_last_error() returns a per-thread message.
 *     Nothing aborts; the Python mirror turns errors into RuntimeError/IndexError the way the
 *     reference raises Python exceptions (src_1gp/trainer.py:54, loss.py:56-57).
 *   - channel padding: node-feature rows are laid out [N, H, Cp] with Cp % 4 == 0 (16-byte
 *     vector loads); the host mirror zero-pads C = 15/30/45/90 to 16/32/48/92.
 *   - per-node scalar pairs are packed as [N, 8] f32: slots 0..3 = per-head "i" value,
 *     slots 4..7 = per-head "j" value (H <= 4).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).
 */
#ifndef GLAM_HIP_H
#define GLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLAM_ABI_VERSION 4   /* bumped whenever an exported signature changes or an entry point goes away */

#define GLAM_OK 0
#define GLAM_E_INVALID (-1)     /* bad argument (null pointer, negative size, misaligned) */
#define GLAM_E_UNSUPPORTED (-2) /* shape outside the compiled kernel table */
#define GLAM_E_HIP (-3)         /* HIP runtime error at launch */

int glam_abi_version(void);
const char* glam_last_error(void);

/* Which of the library's alternative routes are switched on in THIS process (the environment is read once, by the library):
 * "x3" = every dense product of the layer / GRU kernels in 3 x bf16 form on the bf16 matrix cores (GLAM_X3, default 1),
 * "wgrad_x3" = the weight-gradient products on the warp-specialised 3 x bf16 kernel (GLAM_WGRAD_X3, default 1; needs "x3").
 * The host side asks here instead of parsing the variables itself, so that both sides always agree on a route.
 * Replaces: nothing in the reference (it has one route).  Returns 1 / 0, GLAM_E_INVALID for an unknown name. */
int glam_route_enabled(const char* route);

/* Per-launch kernel timing (measurement aid of bench.py; the reference has no profiling hooks at all —
 * only wall-clock suffixes in its log lines, src_1gp/trainer.py:140-147).
 * Between glam_prof_begin(capacity) and glam_prof_end() every kernel this library launches is bound to a
 * (start, stop) event pair stamped with the dispatch's own begin / end timestamps — what a rocprofv3 kernel trace
 * reports.  Timed launches are NOT hipGraph-capturable: profile eager calls only.  glam_prof_end() returns the
 * number of launches recorded (at most `capacity`; launches beyond it go out untimed).  glam_prof_read(i, ...)
 * waits for launch i and returns its kernel name (host buffer), grid size in blocks and duration in microseconds.
 * Process-global, not thread-safe: one profiling session at a time. */
int glam_prof_begin(int capacity);
int glam_prof_end(void);
int glam_prof_read(int i, char* name_host, int name_cap, int32_t* grid_host, float* usec_host);

/* Zero-padded copies of up to 8 parameter tensors in ONE launch, and their gradients back in one.
 * Replaces: nothing in the reference (it has no padded layouts); on this side it replaces the F.pad / slice-copy pairs that
 * re-lay a GRU's gate matrices [3C, C] -> [3Cp, Cp] (src_1gp/layer.py:247) and a Linear's [M, K] -> [Mp, Kp] (src_1gp/layer.py:229)
 * once per model pass when hid_dim = 15 * hid_dim_alpha is not a multiple of four (src_1gp/glam.py:60: four of five widths).
 * Tensor t is row-major [d0, d1, d2], its padded form [d0, p1, p2]; dims = n x {d0, d1, d2, p1, p2}; src / dst are HOST arrays of n
 * device pointers.  backward = 0: dst[t] (padded) <- src[t] (plain), zeros in the pad.  backward = 1: dst[t] (plain gradient) <-
 * src[t] (padded gradient; NULL = that gradient does not exist: zeros). */
int glam_pad_group(int n, const float* const* src, float* const* dst, const int32_t* dims, int backward, void* stream);

/* 64-bit content fingerprint of up to four device buffers (bufs / nbytes: HOST arrays of n device pointers and byte counts, each
 * 4-byte aligned and a multiple of 4 bytes) into the device word out_dev, in one launch; position-sensitive, independent of the
 * launch geometry.  Replaces: nothing in the reference — its trainer collates and copies a fresh batch every iteration
 * (src_1gp/trainer.py:292-295), so a batch that comes back every epoch (the loader does not shuffle: trainer.py:37-38) can only be
 * recognised by content; glam_amd.graphs keys its staged index structures and captured hipGraphs on the fingerprint of
 * (edge_index, batch, edge_attr).  Not for use inside a stream capture by the caller that wants to read the word back. */
int glam_batch_fingerprint(int n, const void* const* bufs, const int64_t* nbytes, uint64_t* out_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CSR staging of a COO edge list.
 * Replaces: the per-call index_select/scatter bookkeeping PyG's MessagePassing.propagate does for
 * `self.propagate(edge_index, ...)` (src_1gp/layer.py:40, :86) — edge_index int64 [2,E], row 0 =
 * source j, row 1 = target i.
 *   by = 0: group edges by TARGET (edge_index[1]); nbr[] holds the source of each grouped edge.
 *   by = 1: group edges by SOURCE (edge_index[0]); nbr[] holds the target (the transpose, used by
 *           the backward pass to scatter to sources without atomics).
 * Within a segment edges keep their original order (stable), so results are bit-reproducible.
 * rowptr int32[N+1], nbr int32[E], eid int32[E] (original edge id of each grouped edge).
 * err_flag int32[1]: set to 1 by the kernels if any index is outside [0,N) (such edges are
 * dropped); the caller decides when to read it back.
 * ws: scratch of at least glam_csr_workspace_bytes(N,E) bytes. */
size_t glam_csr_workspace_bytes(int64_t N, int64_t E);
int glam_csr_build(const int64_t* edge_index, int64_t N, int64_t E, int by, int32_t* rowptr, int32_t* nbr,
                   int32_t* eid, int32_t* err_flag, void* ws, size_t ws_bytes, void* stream);

/* Segment pointer of a sorted graph-id vector: ptr int32[B+1] from batch int64[N] (non-decreasing),
 * as produced by PyG collation and consumed by global_*_pool(x, batch) (src_1gp/layer.py:202). */
int glam_batch_ptr(const int64_t* batch, int64_t N, int64_t B, int32_t* ptr, int32_t* err_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused neighbour gather + attention logits + per-target segment softmax + weighted scatter-add.
 * Replaces: PyG propagate -> TripletMessage.message -> aggregate (src_1gp/layer.py:40-55; softmax
 * at :51 with the +1e-16 denominator; aggr='add' at :17) when emul=1, and
 * TripletMessageLight.message -> aggregate (src_1gp/layer.py:88-97) when emul=0.
 *
 *   logit[e,h] = leaky_relu(a_ij[dst,h] + sum_k edge_attr[e,k]*M[k,h] + a_ij[src,4+h], slope)
 *   alpha      = segment softmax of logit over the incoming edges of dst
 *   emul=1: aggr[n,h,:] = sum_e alpha[e,h] * (sum_k edge_attr[e,k]*w_edge[k,h,:]) * xw[src,h,:]
 *   emul=0: aggr[n,h,:] = sum_e alpha[e,h] * xw[src,h,:]
 *
 * xw f32[N,H,Cp]; a_ij f32[N,8]; edge_attr f32[E,De] (De in {4,8}, zero padded by the host);
 * w_edge f32[De,H,Cp] (ignored when emul=0); M f32[De,4]; CSR by target (rowptr,src,eid).
 * Outputs: aggr f32[N,H,Cp]; stats f32[N,8] = per-head segment max (0..3) and exp-sum (4..7), the
 * only state the backward pass needs to recompute alpha. */
int glam_triplet_fwd(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge,
                     const float* M, const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t N,
                     int64_t E, int H, int Cp, int De, int emul, float slope, float* aggr, float* stats,
                     void* stream);

/* Backward of glam_triplet_fwd (what torch autograd derives for src_1gp/layer.py:42-55 + PyG softmax /
 * scatter).  Two launches + one partial reduction, no atomics:
 *   B1 (by target):  alpha_e f32[E,4], dpre_e f32[E,4] (grad of the pre-activation logit),
 *                    d_a_ij[:,0:4], d_w_edge f32[De,H,Cp], d_M f32[De,4], optional d_edge_attr f32[E,De];
 *   B2 (by source, CSR transpose colptr/dst/eid_t):  d_xw f32[N,H,Cp], d_a_ij[:,4:8].
 * d_edge_attr may be NULL.  ws >= glam_triplet_bwd_workspace_bytes(...) bytes of scratch. */
size_t glam_triplet_bwd_workspace_bytes(int64_t N, int64_t E, int H, int Cp, int De);
int glam_triplet_bwd(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge,
                     const float* M, const float* aggr, const float* stats, const float* d_aggr,
                     const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                     const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int H, int Cp, int De,
                     int emul, float slope, float* d_xw, float* d_a_ij, float* d_w_edge, float* d_M,
                     float* d_edge_attr, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Graph readouts over contiguous node segments ptr int32[B+1].
 * Replaces: global_mean_pool / global_add_pool / global_max_pool / global_sort_pool(k) as called by
 * GlobalPool5.forward (src_1gp/layer.py:201-203).
 *   glam_pool5_fwd: out f32[B, (2+k)*D] = mean | add | the k rows with the largest LAST channel in
 *   descending order (ties: lower node index first; short graphs zero padded); topk_idx int32[B,k]
 *   (node index or -1) is saved for the backward pass.  k <= 8. */
int glam_pool5_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int k, float* out,
                   int32_t* topk_idx, void* stream);
int glam_pool5_bwd(const float* d_out, const int32_t* ptr, const int32_t* topk_idx, int64_t N, int64_t B,
                   int D, int k, float* d_x, void* stream);
/* The same readout on zero-padded rows: x / d_x rows are ld floats apart and hold D channels, ld = D rounded up to a multiple
 * of four (ld <= 128; hid_dim 15 / 30 / 45 / 90 of src_1gp/glam.py:60 flow as 16 / 32 / 48 / 92), columns D..ld zero.  out / d_out
 * stay the compact [B, (2+k)*D] of the reference, the sort key is channel D - 1; d_x's pad columns are written as zeros.  ld == D
 * is glam_pool5_fwd / _bwd. */
int glam_pool5_padded_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int ld, int D, int k, float* out,
                          int32_t* topk_idx, void* stream);
int glam_pool5_padded_bwd(const float* d_out, const int32_t* ptr, const int32_t* topk_idx, int64_t N, int64_t B,
                          int ld, int D, int k, float* d_x, void* stream);

/* mode 0 = sum, 1 = mean, 2 = max (empty segment -> 0; argmax int32[B,D] saved for backward, may be
 * NULL for modes 0/1).  Replaces torch_scatter.scatter(x, batch, dim=0, reduce=...) behind PyG's
 * global_{add,mean,max}_pool. */
int glam_segment_pool_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int mode, float* out,
                          int32_t* argmax, void* stream);
int glam_segment_pool_bwd(const float* d_out, const int32_t* ptr, const int32_t* argmax, int64_t N, int64_t B,
                          int D, int mode, float* d_x, void* stream);

/* Segment-softmax attention readout: out[g,:] = sum_{n in g} softmax_g(gate)[n] * v[n,:]
 * (softmax with the same +1e-16 denominator as PyG's utils.softmax).
 * Replaces: PyG GlobalAttention.forward behind GlobalLAPool (src_1gp/layer.py:206-220) and the
 * attention step of Set2Set (src_1gp/model.py:41).  gate f32[N]; v f32[N,D]; out f32[B,D];
 * stats f32[B,2] = segment max, exp-sum (saved for the backward pass). */
int glam_segment_attn_fwd(const float* gate, const float* v, const int32_t* ptr, int64_t N, int64_t B, int D,
                          float* out, float* stats, void* stream);
int glam_segment_attn_bwd(const float* gate, const float* v, const float* out, const float* stats,
                          const float* d_out, const int32_t* ptr, int64_t N, int64_t B, int D, float* d_gate,
                          float* d_v, void* stream);

/* Generic edge -> node reduction over a CSR-by-target:  out[n,:] = reduce_{e in seg(n)} msg[eid[e],:]
 * mode 0 = sum, 1 = mean, 2 = max (empty segment -> 0; argmax int32[N,D] = winning edge id, saved for
 * backward, may be NULL for modes 0/1).
 * Replaces: PyG MessagePassing.aggregate -> torch_scatter.scatter(msg, edge_index[1], dim=0, dim_size=N,
 * reduce=aggr) for message functions that stay in PyTorch, e.g. NNConv(aggr='mean')
 * (src_1gp/layer.py:119), GCNConv (src_1gp/layer.py:146). */
int glam_edge_reduce_fwd(const float* msg, const int32_t* rowptr, const int32_t* eid, int64_t N, int64_t E, int D,
                         int mode, float* out, int32_t* argmax, void* stream);
int glam_edge_reduce_bwd(const float* d_out, const int32_t* rowptr, const int32_t* eid, const int32_t* argmax,
                         int64_t N, int64_t E, int D, int mode, float* d_msg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense ends of the layer on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32).
 * Replaces: torch.matmul(x, weight_node) (src_1gp/layer.py:37), torch.matmul(aggr, weight_scale) + bias
 * (src_1gp/layer.py:58-60) and the matmuls autograd derives from them.
 *   glam_ts_gemm_make_image: lays a weight W[K, M] (logical W[k][m] = transW ? W[m*ldw+k] : W[k*ldw+m]) out as the
 *                    LDS image the GEMM blocks copy verbatim (glam_ts_gemm_image_bytes(K, M) bytes).
 *   glam_ts_gemm:    out[N, M1|M2] = [A1 | A2][N, K1+K2] @ W (+ bias on the M1 columns); all K*, M*, ld* % 4 == 0;
 *                    K1+K2 <= 192 with M <= 64, or K1+K2 <= 64 with M <= 192.
 *   glam_wgrad_gemm: out[i*stride_i + j*stride_j] = sum_n [P1 | P2 | 1][n, i] * [Q | 1][n, j]  (reduction over the
 *                    N rows; `ones` / `qones` append an all-ones column on either side, i.e. the bias gradient);
 *                    I <= 320 (five 64-column slabs), J <= 64 (ones columns included); also 64 < J (+ qones) <= 128,
 *                    computed as two column chunks in one launch. */
size_t glam_ts_gemm_image_bytes(int K, int M);
int glam_ts_gemm_make_image(const float* W, int ldw, int transW, int K, int M, float* img, void* stream);
/* The four images of a linear pair (W_a, W_b f32[M,K] as torch.nn stores them; e.g. a GRU's weight_ih / weight_hh) in one
 * launch: forward images (x @ W^T: glam_ts_gemm_image_bytes(K, M)) and input-gradient images (dy @ W: ..._bytes(M, K)). */
int glam_ts_gemm_make_image_quad(const float* Wa, const float* Wb, int K, int M, float* img_a_fwd, float* img_b_fwd,
                                 float* img_a_bwd, float* img_b_bwd, void* stream);

/* glam_ts_gemm_make_image_quad for a GRU's gate matrices w_ih, w_hh f32[3 C, C] (K = C, M = 3 C) PLUS the two gate-padded images of
 * glam_gru_fused_make_images in the same launch (fused_*: glam_gru_fused_image_bytes() each): the six re-layouts of one MessageBlock's
 * GRU (src_1gp/layer.py:246, 262) change together after every optimizer step.  C <= 64, a multiple of 4. */
int glam_gru_make_images(const float* w_ih, const float* w_hh, int C, float* img_a_fwd, float* img_b_fwd, float* img_a_bwd,
                         float* img_b_bwd, float* fused_ih, float* fused_hh, void* stream);
int glam_ts_gemm(const float* A1, int K1, int lda1, const float* A2, int K2, int lda2, const float* Wimg,
                 const float* bias, float* out1, int M1, int ldo1, float* out2, int M2, int ldo2, int64_t N,
                 void* stream);
/* glam_wgrad_gemm_pair: two independent products (same N) in one launch + one reduction: the d_W / d_b of two linears
 * that are applied side by side (the input and hidden gate linears of the GRU step, src_1gp/layer.py:262).  Workspace:
 * glam_wgrad_workspace_bytes() (sized for two products). */
int glam_wgrad_gemm_pair(const float* Pa, int Ia, int ldpa, int ones_a, const float* Qa, int Ja, int ldqa, int qones_a,
                         int qcelu_a, float* out_a, int si_a, int sj_a, const float* Pb, int Ib, int ldpb, int ones_b,
                         const float* Qb, int Jb, int ldqb, int qones_b, int qcelu_b, float* out_b, int si_b, int sj_b,
                         int64_t N, void* ws, size_t ws_bytes, void* stream);
/* [d_W | d_b] of two linears y = [x | 1] W^T (the GRU's gate linears, src_1gp/layer.py:247) in one launch + one reduction with
 * weights and biases in SEPARATE contiguous outputs: dw_x f32[I, J] = P_x^T Q_x, db_x f32[I] = column sums of P_x; optional
 * addends laid out like the outputs (gradient carry; N > 0 when one is given).  qcelu_x: Q_x is consumed as celu(Q_x). */
int glam_wgrad_gemm_pair_split(const float* Pa, int Ia, int ldpa, const float* Qa, int Ja, int ldqa, int qcelu_a, float* dw_a,
                               float* db_a, const float* Pb, int Ib, int ldpb, const float* Qb, int Jb, int ldqb, int qcelu_b,
                               float* dw_b, float* db_b, int64_t N, void* ws, size_t ws_bytes, const float* add_w_a,
                               const float* add_b_a, const float* add_w_b, const float* add_b_b, void* stream);

/* glam_wgrad_gemm (one P block, J <= 64) summed over nseg <= 3 operand sets of N rows each (+ an optional addend laid out like out): the
 * weight gradient of a tall matmul that a block applies message_steps times with shared weights — NNConv's relation product
 * [x_r | 1]^T dy, /root/reference/src_1gp/layer.py:115-122 — as ONE launch + ONE reduction.  GLAM_E_UNSUPPORTED when N is shorter than a
 * wave's row range. */
int glam_wgrad_gemm_sets(int nseg, const float* const* P, int I, int ldp, int ones, const float* const* Q, int J, int ldq, int64_t N,
                         float* out, int stride_i, int stride_j, const float* addend, void* ws, size_t ws_bytes, void* stream);

/* glam_wgrad_gemm_pair_split summed over nseg <= 3 operand sets of N rows each — Pa[s], Qa[s], Pb[s], Qb[s] with the same widths and
 * row strides — in ONE launch + ONE reduction: the weight gradients of a block applied message_steps times with shared weights
 * (/root/reference/src_1gp/model.py:53-54; the GRU's two gate matrices, src_1gp/layer.py:247) are one product over all its
 * applications instead of one per application.  N must be at least a wave's row range (GLAM_E_UNSUPPORTED otherwise: run the sets
 * one by one with the addends). */
int glam_wgrad_gemm_pair_split_seg(int nseg, const float* const* Pa, int Ia, int ldpa, const float* const* Qa, int Ja, int ldqa, int qcelu_a,
                                   float* dw_a, float* db_a, const float* const* Pb, int Ib, int ldpb, const float* const* Qb, int Jb,
                                   int ldqb, int qcelu_b, float* dw_b, float* db_b, int64_t N, void* ws, size_t ws_bytes,
                                   const float* add_w_a, const float* add_b_a, const float* add_w_b, const float* add_b_b, void* stream);

/* Both weight gradients of a GRU step (torch.nn.GRU(C, C), /root/reference/src_1gp/layer.py:247, :262) from the ONE gate-gradient
 * matrix glam_gru_bwd_ws writes when its d_gh is NULL: D[n] = [d_pr | d_pz | d_pn | d_pn r], 4C floats per row.  d_gi = D[:, 0:3C] and
 * d_gh = [D[:, 0:2C] | D[:, 3C:4C]] are column blocks of it:
 *   dw_ih[3C, C] = d_gi^T X (celu(X) with qcelu), db_ih[3C] = column sums of d_gi, dw_hh[3C, C] = d_gh^T H, db_hh[3C] = column sums of d_gh,
 * summed over nseg <= 3 operand sets (D[s], X[s], H[s]) of N rows each (X, H with row strides ldx, ldh), + optional addends laid out
 * like the outputs — glam_wgrad_gemm_pair_split_seg on the two expanded matrices, bit for bit, without their 2C duplicated columns
 * in memory.  C a multiple of 4, C + 1 <= 64.  ws >= glam_wgrad_workspace_bytes(). */
int glam_wgrad_gemm_gru_gates_seg(int nseg, const float* const* D, int C, const float* const* X, int ldx, int qcelu,
                                  const float* const* H, int ldh, float* dw_ih, float* db_ih, float* dw_hh, float* db_hh, int64_t N,
                                  void* ws, size_t ws_bytes, const float* add_w_ih, const float* add_b_ih, const float* add_w_hh,
                                  const float* add_b_hh, void* stream);

/* glam_wgrad_gemm in full ([P1 | P2 | 1]^T Q, output strides, J <= 128) summed over nseg <= 3 operand sets of N rows each (+ an
 * optional addend laid out like out): the N-deep weight gradients of the wide TripletMessage over all applications of the layer. */
int glam_wgrad_gemm_sets2(int nseg, const float* const* P1, int I1, int ldp1, const float* const* P2, int I2, int ldp2, int ones,
                          const float* const* Q, int J, int ldq, int64_t N, float* out, int stride_i, int stride_j, const float* addend,
                          void* ws, size_t ws_bytes, void* stream);
/* [d_W | d_b] of ONE linear y = [act(x) | 1] W^T with up to 127 inputs, weight and bias gradients in separate contiguous tensors with
 * optional addends (the gradient carry): dw[I, J] = P^T act(Q) (+ add_w), db[I] = column sums of P (+ add_b); act = CELU(alpha = 1)
 * when q_celu (the CELU in front of a MessageBlock's GRU, /root/reference/src_1gp/layer.py:261, folded into the gate product of a
 * wide layer).  J % 4 == 0, J + 1 <= 128, I <= 320, N >= 1. */
int glam_wgrad_gemm_linear(const float* P, int I, int ldp, const float* Q, int J, int ldq, int q_celu, float* dw, float* db,
                           const float* add_w, const float* add_b, int64_t N, void* ws, size_t ws_bytes, void* stream);
/* ... summed over nseg <= 3 operand sets of N rows each (the applications of a block that shares the linear: see
 * glam_wgrad_gemm_pair_split_seg). */
int glam_wgrad_gemm_linear_sets(int nseg, const float* const* P, int I, int ldp, const float* const* Q, int J, int ldq, int q_celu,
                                float* dw, float* db, const float* add_w, const float* add_b, int64_t N, void* ws, size_t ws_bytes,
                                void* stream);
/* glam_wgrad_gemm for ONE linear y = [x | 1] W^T with the weight and bias gradients in separate contiguous tensors:
 * dw[I, J] = P^T Q (P = dy f32[N, I], Q = x f32[N, J]), db[I] = column sums of P.  ceil4(J) + 1 <= 64, I <= 320; J need not be a
 * multiple of 4 when ldq >= ceil4(J) (a weight narrower than its zero-padded input: dw stays contiguous [I, J]). */
int glam_wgrad_gemm_split(const float* P, int I, int ldp, const float* Q, int J, int ldq, float* dw, float* db, int64_t N, void* ws,
                          size_t ws_bytes, void* stream);
/* ... of a linear whose output went through a ReLU (LinearBlock + ReLU, src_1gp/layer.py:232-237): P = dy * (Y > 0) with Y f32[N, I]
 * (row stride ldp) = the saved output — the ReLU's backward inside the product instead of an elementwise launch in front of it. */
int glam_wgrad_gemm_split_relu(const float* P, const float* Y, int I, int ldp, const float* Q, int J, int ldq, float* dw, float* db,
                               int64_t N, void* ws, size_t ws_bytes, void* stream);
/* out[N, M] = max(A[N, K] @ W + bias, 0): the Linear + ReLU of a LinearBlock (src_1gp/layer.py:232-237) whose shape runs on the
 * warp-specialised tall kernel (glam_ts_gemm_relu_supported(K, M): the input embeddings 15 -> 60 of the parity configuration and of the
 * two-tower models) — the activation in the product's epilogue instead of an elementwise launch behind it.  Wimg, alignment and
 * leading dimensions as glam_ts_gemm. */
int glam_ts_gemm_relu_supported(int K, int M);
int glam_ts_gemm_relu(const float* A, int K, int lda, const float* Wimg, const float* bias, float* out, int M, int ldo, int64_t N,
                      void* stream);
/* ... with the TRAINING-mode RReLU(rr_lower, rr_upper) of /root/reference/src_1gp/model.py:31 (the reference's default activation) in the
 * epilogue and, out_drop non-NULL, out_drop = Dropout(drop_p)(out): the twin the block behind starts with (layer.py:255-256) — from the
 * device-side Philox stream: the words glam_bias_res_act_rng_fwd draws for the same elements at the same stream position, so
 * glam_ts_gemm + glam_bias_res_act_rng_fwd and this ONE launch write the same bits, and glam_bias_res_act_rng_bwd with rng_eff is its
 * activation backward.  out, out_drop: [N, M] contiguous.  Shapes: glam_ts_gemm_rrelu_supported (K <= 64, M <= 64: the input embeddings
 * mol_lin0 / pro_lin0 of model.py:40-49). */
int glam_ts_gemm_rrelu_supported(int K, int M);
/* The input embedding in front of a TripletMessage (/root/reference/src_1gp/model.py:49, :53) with the TripletMessage's node product of
 * its rows in the same launch: out = act(A @ W + bias) — act 0: none, 1: ReLU, 4: training-mode RReLU (+ the dropped twin out_drop,
 * as glam_ts_gemm_rrelu; rng_* unused otherwise) — and xw[N, node_cols] | a_ij[N, 8] = out (out_drop when given) @ [W_node | Wa] from
 * node_pre (staged + glam_triplet_staged_node_fragments): what glam_ts_gemm writes for the same rows, bit for bit.
 * glam_triplet_layer_fwd_ell with x = NULL then starts at its aggregate launch.  Shapes: glam_ts_gemm_rrelu_supported(K, M), M = the
 * layer's Cp >= 24, 56 < node_cols = H * Cp <= 184. */
int glam_ts_gemm_act_node(const float* A, int K, int lda, const float* Wimg, const float* bias, int M, int64_t N, int act, float rr_lower,
                          float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* out, float* out_drop,
                          const void* node_pre, int node_cols, float* xw, float* a_ij, void* stream);
int glam_ts_gemm_rrelu(const float* A, int K, int lda, const float* Wimg, const float* bias, int M, int64_t N, float rr_lower,
                       float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* out, float* out_drop, void* stream);

/* glam_ts_gemm with the CELU(alpha=1) that MessageBlock applies in front of its GRU (src_1gp/layer.py:261) folded in:
 * a_celu = 1: out = celu(A) @ W + bias; cgrad_src non-NULL: out[r,c] *= celu'(cgrad_src[r,c]) (the chain rule of the same
 * fold on the way back); qcelu_* in glam_wgrad_gemm_pair: the weight gradient uses celu(Q). */
int glam_ts_gemm_celu(const float* A, int K, int lda, int a_celu, const float* Wimg, const float* bias, float* out, int M,
                      int ldo, const float* cgrad_src, int ld_cgrad, int64_t N, void* stream);
/* out[N, M] = A[N, K] @ W + bias + addend[N, M] (row pitch ld_add, a multiple of 4): a second gradient path into the same tensor
 * (the GRU backward's d_h = d_gh @ W_hh^T + the direct z * g term) without an add launch. */
int glam_ts_gemm_add(const float* A, int K, int lda, const float* Wimg, const float* bias, float* out, int M, int ldo,
                     const float* addend, int ld_add, int64_t N, void* stream);
/* Two products that share N and the kernel variant in ONE launch (the GRU's gate linears gi = celu(x) W_ih^T + b_ih and
 * gh = h W_hh^T + b_hh of src_1gp/layer.py:261-262, and the two input-gradient products of their backward), each with every option of
 * glam_ts_gemm_celu / glam_ts_gemm_add: out_x = act_x(A_x) @ W_x (+ bias_x) (* celu'(cgrad_x)) (+ addend_x).  Besides the dispatch it
 * saves, the second product's blocks share the CUs with the first's (two 8-wave blocks per CU for the 48 KB-image variants). */
int glam_ts_gemm_pair(const float* Aa, int Ka, int lda, int a_celu_a, const float* Wimg_a, const float* bias_a, float* out_a, int Ma,
                      int ldo_a, const float* cgrad_a, int ld_cgrad_a, const float* addend_a, int ld_add_a, const float* Ab, int Kb,
                      int ldb, int a_celu_b, const float* Wimg_b, const float* bias_b, float* out_b, int Mb, int ldo_b,
                      const float* cgrad_b, int ld_cgrad_b, const float* addend_b, int ld_add_b, int64_t N, void* stream);
size_t glam_wgrad_workspace_bytes(void);
int glam_wgrad_gemm(const float* P1, int I1, int ldp1, const float* P2, int I2, int ldp2, int ones, const float* Q,
                    int J, int ldq, int qones, int64_t N, float* out, int stride_i, int stride_j, void* ws,
                    size_t ws_bytes, void* stream);
/* glam_wgrad_gemm + addend (f32, the strides of out): out = product + addend.  The gradient carry of a weight that a block applies
 * several times per forward (src_1gp/model.py:53-54) is summed by the reduction instead of an add launch.  N > 0. */
int glam_wgrad_gemm_add(const float* P1, int I1, int ldp1, const float* P2, int I2, int ldp2, int ones, const float* Q,
                        int J, int ldq, int qones, int64_t N, float* out, int stride_i, int stride_j, const float* addend,
                        void* ws, size_t ws_bytes, void* stream);

/* The derived weights of a model pass in ONE launch (three launches of ~5 us each at the head of every training step before): the staged
 * images of a TripletMessage (exactly glam_triplet_stage_params; staged == NULL: none) and n_images <= 6 weight images of
 * glam_ts_gemm (exactly glam_ts_gemm_make_image each: dims[4 q ..] = {ldw, transW, K, M} of image q — e.g. the four images of a GRU's
 * gate matrices and the one of the input linear, /root/reference/src_1gp/model.py:40-42).  transW = 2 / 3 makes image q one matrix's
 * half of a glam_gru_ws_make_pre image instead: {ldw = C, 2: forward image | 3: backward image, K = C, M = 0: weight_ih | 1: weight_hh},
 * img[q] = that image (the two matrices' jobs name the same one). */
int glam_prestage(const float* weight_node, const float* weight_edge, const float* att, const float* weight_scale, const float* bias,
                  int C, int H, int De, int Cp, int Dp, float* staged, int n_images, const float* const* W, const int* dims,
                  float* const* img, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole TripletMessage layer (src_1gp/layer.py:36-61) as one enqueue per direction.  All node-feature
 * widths are the padded Cp (x f32[N,Cp], out f32[N,Cp]; the host mirror pads/slices when C % 4 != 0).
 *   glam_triplet_stage_params: parameters -> `staged` (glam_triplet_staged_floats floats): the four GEMM weight
 *     images ([W_node | Wa_i | Wa_j] with the separable-attention columns, W_scale, and their transposes),
 *     W_edge head-padded, M f32[Dp,4], bias padded.
 *   glam_triplet_layer_fwd:  x -> (xw, a_ij) -> aggr, stats -> out          (2 launches)
 *   glam_triplet_layer_bwd:  d_out -> d_x and `dstaged` (glam_triplet_dstaged_floats floats: d_Wcat[Cp,H*Cp+8] |
 *                            d_WsB[H*Cp+1,Cp] (last row = d_bias) | d_We_p | d_M), optional d_edge_attr
 *                            (9 launches, no atomics)
 *   glam_triplet_stage_params_bwd: chain rule from `dstaged` back to weight_node / weight_edge /
 *                            weight_triplet_att / weight_scale / bias.
 * Limits: H*Cp + 8 <= 192 and Cp <= 64 (C <= 60 at H = 3). */
size_t glam_triplet_staged_floats(int H, int Cp, int Dp);
/* offsets (floats) inside `staged` of [W_node | Wa] (K = Cp rows, H * Cp + 8 columns): as its k_ts_gemm image (what the layer's own node
 * GEMM reads), and as the pre-split operand fragments of the node product inside the GRU step (72 KB: node_pre of
 * glam_gru_ws_fwd_pre_node; (size_t)-1 where the shape has none: H * Cp + 8 <= 64) */
size_t glam_triplet_staged_node_image(int H, int Cp, int Dp);
size_t glam_triplet_staged_node_fragments(int H, int Cp, int Dp);
size_t glam_triplet_dstaged_floats(int H, int Cp, int Dp);
int glam_triplet_stage_params(const float* weight_node, const float* weight_edge, const float* att,
                              const float* weight_scale, const float* bias, int C, int H, int De, int Cp, int Dp,
                              float* staged, void* stream);
int glam_triplet_stage_params_bwd(const float* weight_node, const float* weight_edge, const float* att,
                                  const float* dstaged, int C, int H, int De, int Cp, int Dp, float* d_weight_node,
                                  float* d_weight_edge, float* d_att, float* d_weight_scale, float* d_bias,
                                  void* stream);
/* Wide layers (H*Cp + 8 > 192: hid_dim_alpha = 6 of the reference's search space, glam.py:60): the same derived
 * parameters as plain row-major matrices for library GEMMs, `plain` = Wcat f32[Cp, H*Cp+8] | Ws_p f32[H*Cp, Cp] |
 * We_p f32[Dp, H*Cp] | M f32[Dp, 4] | bias_p f32[Cp] (glam_triplet_plain_floats floats).  The host mirror sequences
 * library GEMMs, glam_triplet_fwd / _bwd and glam_wgrad_gemm around it, fills a `dstaged` buffer and calls
 * glam_triplet_stage_params_bwd (which has no width limit). */
size_t glam_triplet_plain_floats(int H, int Cp, int Dp);
int glam_triplet_stage_plain(const float* weight_node, const float* weight_edge, const float* att,
                             const float* weight_scale, const float* bias, int C, int H, int De, int Cp, int Dp,
                             float* plain, void* stream);
int glam_triplet_layer_fwd(const float* x, const float* edge_attr, const float* staged, const int32_t* rowptr,
                           const int32_t* src, const int32_t* eid, int64_t N, int64_t E, int H, int Cp, int Dp, float slope,
                           float* xw, float* a_ij, float* aggr, float* stats, float* out, void* stream);
size_t glam_triplet_layer_bwd_workspace_bytes(int64_t N, int64_t E, int H, int Cp, int Dp);
int glam_triplet_layer_bwd(const float* x, const float* edge_attr, const float* staged, const float* xw,
                           const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                           const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                           const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int H, int Cp, int Dp,
                           float slope, float* d_x, float* dstaged, float* d_edge_attr, void* ws, size_t ws_bytes,
                           void* stream);

/* glam_triplet_layer_bwd with the parameter chain rule folded in: the gradients of weight_node f32[C,H*C],
 * weight_edge f32[De,H*C], weight_triplet_att f32[H,3C] (as one buffer), weight_scale f32[H*C,C] and bias f32[C]
 * come straight out of the backward pass's partial sums (what autograd derives for src_1gp/layer.py:22-61), one
 * launch instead of glam_triplet_layer_bwd's final reduction + glam_triplet_stage_params_bwd.  7 launches. */
int glam_triplet_layer_bwd_params(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                  const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                  const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                                  const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int C, int H, int De,
                                  int Cp, int Dp, float slope, const float* weight_node, const float* weight_edge,
                                  const float* att, float* d_x, float* d_weight_node, float* d_weight_edge, float* d_att,
                                  float* d_weight_scale, float* d_bias, float* d_edge_attr, void* ws, size_t ws_bytes,
                                  void* stream);

/* glam_triplet_layer_bwd_params with addends for the five parameter gradients (each laid out like its output, any of them
 * NULL): out = gradient + addend.  The block is applied message_steps times with shared weights (src_1gp/model.py:53-54); the
 * gradient already accumulated by its later applications enters here instead of through a separate add launch.  N > 0.
 * ell_dst / ell_eid_t (may be NULL): ELL records BY SOURCE — glam_ell_build on (colptr, dst, eid_t); every node of a molecular graph
 * has at most 4 out-edges — select, for one-hot edge features of width 4 (edge_onehot = 1, glam_triplet_layer_ws_supported), the
 * warp-specialised B2 + d_x launch (bit-identical); ignored otherwise.  edge_onehot as in glam_triplet_fwd_ell. */
int glam_triplet_layer_bwd_params_acc(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                      const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                      const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                                      const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int C, int H, int De,
                                      int Cp, int Dp, float slope, const float* weight_node, const float* weight_edge,
                                      const float* att, float* d_x, float* d_weight_node, float* d_weight_edge, float* d_att,
                                      float* d_weight_scale, float* d_bias, const float* add_weight_node,
                                      const float* add_weight_edge, const float* add_att, const float* add_weight_scale,
                                      const float* add_bias, const int32_t* ell_dst, const int32_t* ell_eid_t, int edge_onehot,
                                      float* d_edge_attr, void* ws, size_t ws_bytes, void* stream);
/* glam_triplet_layer_bwd_params_acc with the ELL records of BOTH directions: ell_src / ell_eid BY TARGET (glam_ell_build on
 * (rowptr, src, eid): what the forward used) and ell_dst / ell_eid_t BY SOURCE; either pair may be NULL.  With one-hot edge features of
 * width 4 (edge_onehot = 1, glam_triplet_layer_ws_supported) both aggregate launches of the backward run warp-specialised: B1 with the
 * d_aggr GEMM produced ahead by matrix waves (csrc/triplet_ws_b1.hip), B2 with the d_x GEMM as the consumers' product
 * (csrc/triplet_ws.hip).  d_x, d_weight_node, d_weight_scale, d_bias and every per-node / per-edge intermediate equal the general
 * route bit for bit; d_weight_edge and the edge part of d_att are the same sums in another fixed order (autograd of
 * src_1gp/layer.py:36-61). */
int glam_triplet_layer_bwd_params_ell(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                      const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                      const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                                      const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int C, int H, int De,
                                      int Cp, int Dp, float slope, const float* weight_node, const float* weight_edge,
                                      const float* att, float* d_x, float* d_weight_node, float* d_weight_edge, float* d_att,
                                      float* d_weight_scale, float* d_bias, const float* add_weight_node,
                                      const float* add_weight_edge, const float* add_att, const float* add_weight_scale,
                                      const float* add_bias, const int32_t* ell_src, const int32_t* ell_eid, const int32_t* ell_dst,
                                      const int32_t* ell_eid_t, int edge_onehot, float* d_edge_attr, void* ws, size_t ws_bytes,
                                      void* stream);
/* The same with a second gradient path into the layer's input: MessageBlock keeps `identity = x` (src_1gp/layer.py:253) and adds it back
 * behind the GRU (:264), so x receives d_identity besides the layer's own input gradient.  d_x = (input gradient) + d_x_addend[N, Cp],
 * summed in the epilogue of the d_x product instead of by an add launch per block application.  Warp-specialised route only (both ELL
 * pairs given, glam_triplet_layer_ws_supported): GLAM_E_UNSUPPORTED otherwise.  d_x_addend == NULL is glam_triplet_layer_bwd_params_ell. */
int glam_triplet_layer_bwd_params_ell_add(const float* x, const float* edge_attr, const float* staged, const float* xw,
                                          const float* a_ij, const float* aggr, const float* stats, const float* d_out,
                                          const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int32_t* colptr,
                                          const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E, int C, int H, int De,
                                          int Cp, int Dp, float slope, const float* weight_node, const float* weight_edge,
                                          const float* att, float* d_x, float* d_weight_node, float* d_weight_edge, float* d_att,
                                          float* d_weight_scale, float* d_bias, const float* add_weight_node,
                                          const float* add_weight_edge, const float* add_att, const float* add_weight_scale,
                                          const float* add_bias, const int32_t* ell_src, const int32_t* ell_eid, const int32_t* ell_dst,
                                          const int32_t* ell_eid_t, int edge_onehot, float* d_edge_attr, void* ws, size_t ws_bytes,
                                          const float* d_x_addend, void* stream);

/* A layer applied message_steps times with shared weights (/root/reference/src_1gp/model.py:53-54), warp-specialised route: every
 * application's backward runs its DATA half — glam_triplet_layer_bwd_data_ell: everything of glam_triplet_layer_bwd_params_ell_add
 * that produces d_x (d_x_addend may be NULL) — and leaves its operands (d_xw, d_a, the block partials of d_W_edge / d_M) in ITS
 * workspace `ws`, whose layout it reports in info (host int64[4]: byte offsets and the partial row count).  The PARAMETER half
 * runs once over the operand sets of nseg <= 3 applications: glam_triplet_layer_param_grads_sets(ws_set[s], info[4 s ..], x[s],
 * aggr[s], d_out[s]) = both weight-gradient products as ONE k_wgrad launch over all sets + ONE k_param_grads (optional addends: the
 * gradient carry).  Its own ws: >= 2 * glam_wgrad_workspace_bytes().  The workspaces of the applications must stay untouched until
 * then.  Same numbers as the per-application form up to the order of the fixed-order sums. */
int glam_triplet_layer_bwd_data_ell(const float* x, const float* edge_attr, const float* staged, const float* xw, const float* a_ij,
                                    const float* aggr, const float* stats, const float* d_out, const int32_t* rowptr, const int32_t* src,
                                    const int32_t* eid, const int32_t* colptr, const int32_t* dst, const int32_t* eid_t, int64_t N, int64_t E,
                                    int C, int H, int De, int Cp, int Dp, float slope, float* d_x, const int32_t* ell_src,
                                    const int32_t* ell_eid, const int32_t* ell_dst, const int32_t* ell_eid_t, int edge_onehot,
                                    float* d_edge_attr, void* ws, size_t ws_bytes, const float* d_x_addend, int64_t* info, void* stream);
int glam_triplet_layer_param_grads_sets(int nseg, const void* const* ws_set, const int64_t* info, const float* const* x,
                                        const float* const* aggr, const float* const* d_out, int64_t N, int C, int H, int De, int Cp, int Dp,
                                        const float* weight_node, const float* weight_edge, const float* att, float* d_weight_node,
                                        float* d_weight_edge, float* d_att, float* d_weight_scale, float* d_bias,
                                        const float* add_weight_node, const float* add_weight_edge, const float* add_att,
                                        const float* add_weight_scale, const float* add_bias, void* ws, size_t ws_bytes, void* stream);
/* ---------------------------------------------------------------------------------------------
 * MessageBlock remainder: gate math of one torch.nn.GRU(C, C) step with seq_len 1 (src_1gp/layer.py:247, :262).
 * gi = celu(x) @ W_ih^T + b_ih and gh = h @ W_hh^T + b_hh (f32[N,3C], gate order r|z|n; computed with
 * glam_ts_gemm) -> h_new f32[N,C].  The backward recomputes the gates from gi / gh. */
int glam_gru_gates_fwd(const float* gi, const float* gh, const float* h, int64_t N, int C, float* h_new, void* stream);
/* The same gate math with the rest of MessageBlock.forward folded in (src_1gp/layer.py:263-266): h_new = GRU gates
 * (the state handed to the next step), out = act(h_new + identity) with identity NULL for res=False and act in
 * {0 none, 1 ReLU, 2 LeakyReLU(slope), 3 CELU(alpha=1)}.  Backward: d_out is the gradient of `out`, d_hstate (may be
 * NULL) the gradient reaching h_new through the next step; it returns d_gi, d_gh, d_h and d_identity (if non-NULL). */
int glam_gru_tail_fwd(const float* gi, const float* gh, const float* h, const float* identity, int64_t N, int C, int act,
                      float slope, float* h_new, float* out, void* stream);
int glam_gru_tail_bwd(const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                      const float* d_hstate, int64_t N, int C, int act, float slope, float* d_gi, float* d_gh, float* d_h,
                      float* d_identity, void* stream);
int glam_gru_gates_bwd(const float* gi, const float* gh, const float* h, const float* d_hnew, int64_t N, int C,
                       float* d_gi, float* d_gh, float* d_h, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-graph normalisation over contiguous node segments ptr int32[B+1] (empty graphs are skipped).
 * Replaces: PyG PairNorm()(x, batch) behind _PairNorm (src_1gp/layer.py:179-185; mode 0, scale 1, eps 1e-5) and the
 * statistics of PyG's graph LayerNorm(x, batch) behind _LayerNorm (src_1gp/layer.py:170-176; mode 1: y = (x - mean) /
 * sqrt(var + eps) over all nodes x channels of a graph; the per-channel affine stays in the host mirror).
 * D <= 256.  The backward pass recomputes the statistics from x. */
int glam_graph_norm_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int mode, float scale, float eps,
                        float* y, void* stream);
int glam_graph_norm_bwd(const float* x, const float* gy, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                        float scale, float eps, float* dx, void* stream);
/* dx = glam_graph_norm_bwd's result + addend f32[N, D]: the residual path of a MessageBlock (src_1gp/layer.py:253-265: x feeds the norm
 * AND the block's skip connection) joins in the store instead of in an add launch.  Every node must belong to a graph (ptr[B] = N). */
int glam_graph_norm_bwd_add(const float* x, const float* gy, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                            float scale, float eps, const float* addend, float* dx, void* stream);

/* The norm and the training-mode Dropout(p) a MessageBlock applies right behind it (x = norm(x); x = dropout(x),
 * /root/reference/src_1gp/layer.py:255-256; run.py's default configuration: _PairNorm + Dropout(0.2)) in ONE launch each way, on the
 * device-side Philox stream (rng_state: int64[GLAM_RNG_STATE_WORDS], advanced by the launch; rng_eff: int64[2], the stream position
 * the forward used, handed to the backward, which regenerates the mask).  y_drop = y * mask / (1 - p); y (the plain output) may be
 * NULL when nobody reads it.  Backward: dx = norm'(gy + mask / (1 - p) * gy_drop) (+ addend, may be NULL: see
 * glam_graph_norm_bwd_add); gy may be NULL.  Molecule-sized graphs only (glam_graph_norm_drop_supported: D % 4 == 0, D <= 64, fewer
 * than 64 nodes per graph on average), GLAM_E_UNSUPPORTED otherwise (run the two ops one after the other). */
int glam_graph_norm_drop_supported(int64_t N, int64_t B, int D);
int glam_graph_norm_drop_fwd(const float* x, const int32_t* ptr, int64_t N, int64_t B, int D, int mode, float scale, float eps, float drop_p,
                             int64_t* rng_state, int64_t* rng_eff, float* y, float* y_drop, void* stream);
int glam_graph_norm_drop_bwd(const float* x, const float* gy, const float* gy_drop, const int32_t* ptr, int64_t N, int64_t B, int D, int mode,
                             float scale, float eps, float drop_p, const int64_t* rng_eff, const float* addend, float* dx, void* stream);

/* Edge-weighted neighbour sums over a CSR-by-target: S[n,k,:] = (mean ? 1/deg_n : 1) * sum_{e -> n} w[eid e, k] *
 * x[src e, :]  (x f32[N,D], w f32[E,K], K in {1,4,8}, out f32[N,K,D]; K = 1 with w = the symmetric degree
 * normalisation is GCNConv's propagate, src_2gi_dti_scr/layer.py:143-149).  With one-hot edge features this is the per-relation
 * neighbour sum that evaluates NNConv(aggr='mean') (src_1gp/layer.py:115-122: per-edge [C,C] weights nn(e_ij)) as a
 * K-relation R-GCN without the [E, C*C] tensor.  The backward (w.r.t. x) walks the CSR transpose. */
int glam_edge_wsum_fwd(const float* x, const float* w, const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                       int64_t N, int64_t E, int D, int K, int mean, int self_slot, float* out, void* stream);
int glam_edge_wsum_bwd(const float* d_out, const float* w, const int32_t* colptr, const int32_t* dst,
                       const int32_t* eid_t, const int32_t* rowptr, int64_t N, int64_t E, int D, int K, int mean,
                       int self_slot, float* dx, void* stream);
/* ... dx = (the same sums) + addend[N, D] (NULL: none): the gradient of a skip connection around the layer (MessageBlock,
 * src_1gp/layer.py:253-265) joins here instead of in an add launch of its own.  16-byte form only (K in {4, 8}, D % 4 == 0, aligned). */
int glam_edge_wsum_bwd_add(const float* d_out, const float* w, const int32_t* colptr, const int32_t* dst, const int32_t* eid_t,
                           const int32_t* rowptr, int64_t N, int64_t E, int D, int K, int mean, int self_slot, const float* addend,
                           float* dx, void* stream);
/* self_slot = 1 (K in {4, 8}, D % 4 == 0): out is f32[N, K+1, D] and slot K of node n is x[n] itself — NNConv's root term
 * x_i @ root (src_1gp/layer.py:119) as one more relation, so that the layer is ONE GEMM [N, (K+1) D] x [(K+1) D, out]. */

/* ---------------------------------------------------------------------------------------------
 * Per-pair fusion of the two-tower models: out f32[P,2] = [max, mean] of mol[seg_i] @ pro[seg_i]^T for every pair i
 * (segments mol_ptr / pro_ptr int32[P+1]; empty segment -> 0); argmax int32[P,2] = (ligand row, residue row) of the max;
 * sums f32[P,2,D] = the column sums of both segments, kept by the caller from the forward to the backward call.
 * Replaces: dot_and_global_pool2 (src_2gi_dti_scr/layer.py:270-283: Python loop, .item() syncs, one matmul per pair). */
size_t glam_pair_pool_workspace_bytes(int64_t P, int D);
int glam_pair_pool_fwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr, int64_t P,
                       int D, float* out, int32_t* argmax, float* sums, void* workspace, size_t workspace_bytes,
                       void* stream);
int glam_pair_pool_bwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                       const int32_t* argmax, const float* sums, const float* d_out, int64_t P, int D, float* d_mol,
                       float* d_pro, void* stream);
/* ... d_mol += add_mol, d_pro += add_pro (either may be NULL): the gradient of the next message step's use of the two feature
 * matrices (src_2gi_dti_scr/model.py:66-70: every step's outputs feed this fusion AND the next step), when the caller took them
 * back from this node — one add launch per tower and step less.  glam_pair_pool_add_supported(D): the widths that have it. */
int glam_pair_pool_add_supported(int D);
int glam_pair_pool_bwd_add(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr, const int32_t* argmax,
                           const float* sums, const float* d_out, int64_t P, int D, const float* add_mol, const float* add_pro,
                           float* d_mol, float* d_pro, void* stream);

/* Training-mode block tails of the reference's DEFAULT configuration (src_1gp/model.py:30-31, run.py:35-37: RReLU activations,
 * Dropout(0.2) in front of every conv) — the same launches as glam_gru_tail_* / glam_bias_res_act_* with two additions:
 *   act = 4: torch.nn.RReLU in training mode, out = y > 0 ? y : a * y with a ~ U(rr_lower, rr_upper) per element;
 *   out_drop (may be NULL): a second output Dropout(drop_p)(out) = out * mask / (1 - p), the next conv's input (layer.py:256).
 * Random numbers: Philox4x32-10 keyed by rng_state[0] (seed), stream position rng_state[1] (offset), both int64 in DEVICE memory
 * and ticket words at index 16 and 32 + 16 s, s < 16 (each on its own cache line): rng_state int64[GLAM_RNG_STATE_WORDS = 288], zero
 * everything but the seed once.  Every launch uses the offset it finds
 * and the last block to finish stores offset + 1 — hipGraph replays continue the sequence with no host involvement.  rng_eff
 * int64[2] receives the (seed, offset) pair the launch used; the backward entry points regenerate slopes and masks from it (no
 * mask tensors).  d_out / d_out_drop: gradients of the two outputs (either may be NULL). glam_bias_res_act_rng_fwd accepts
 * out = NULL for act = 0 (a plain Dropout). */
#define GLAM_RNG_STATE_WORDS 288
int glam_gru_tail_rng_fwd(const float* gi, const float* gh, const float* h, const float* identity, int64_t N, int C, int act,
                          float slope, float rr_lower, float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff,
                          float* h_new, float* out, float* out_drop, void* stream);
int glam_gru_tail_rng_bwd(const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                          const float* d_out_drop, const float* d_hstate, int64_t N, int C, int act, float slope, float rr_lower,
                          float rr_upper, float drop_p, const int64_t* rng_eff, float* d_gi, float* d_gh, float* d_h,
                          float* d_identity, void* stream);
int glam_bias_res_act_rng_fwd(const float* y, const float* bias, const float* identity, int64_t N, int C, int act, float slope,
                              float rr_lower, float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* out,
                              float* out_drop, void* stream);
int glam_bias_res_act_rng_bwd(const float* out, const float* d_out, const float* d_out_drop, int64_t N, int C, int act,
                              float slope, float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff, float* d_y,
                              void* stream);

/* Software-pipelined forward aggregate for graphs whose in-degree never exceeds 4 (molecules): the arithmetic of
 * glam_triplet_fwd (multi-head form, emul = 1; src_1gp/layer.py:42-55 through PyG propagate / softmax / scatter-add), bit for
 * bit, with the neighbour rows staged global -> LDS by LDS-DMA and every wave prefetching the rows of its next pass and the
 * index record of the one after while it computes the current one (csrc/triplet_dma.hip).
 * glam_ell_build: index records from the by-target CSR — ell_src / ell_eid int32[N, 4] (source node and original edge id of
 * the node's incoming edges in CSR order, -1 = empty slot); *overflow_flag (int32, zero it first) is set when some node has
 * more than 4 incoming edges: such an edge list must use glam_triplet_fwd.  Built once per edge list, like the CSR.
 * glam_triplet_fwd_ell: same tensors as glam_triplet_fwd; Cp <= 64, H <= 4, De in {4, 8}; edge_onehot = 1 asserts that every
 * edge_attr row is one-hot (e_ij is then read as one W_edge row, bit-identical); grid_blocks <= 0 picks the default persistent
 * grid (two 4-wave blocks per CU). */
int glam_ell_build(const int32_t* rowptr, const int32_t* nbr, const int32_t* eid, int64_t N, int32_t* ell_src,
                   int32_t* ell_eid, int32_t* overflow_flag, void* stream);
int glam_triplet_fwd_ell_supported(int H, int Cp, int De);
int glam_triplet_fwd_ell(const float* xw, const float* a_ij, const float* edge_attr, const float* w_edge, const float* M,
                         const int32_t* ell_src, const int32_t* ell_eid, int64_t N, int64_t E, int H, int Cp, int De,
                         float slope, int edge_onehot, float* aggr, float* stats, int grid_blocks, void* stream);

/* glam_triplet_layer_fwd for molecular graphs — edge lists with an ELL form (glam_ell_build, in-degree <= 4) and one-hot edge features
 * of width 4 (src_1gp/dataset.py:82): node GEMM + the warp-specialised aggregate / update launch (csrc/triplet_ws.hip: producer waves
 * gather, consumer waves run the update GEMM out of an LDS tile ring).  Same tensors and results (bit for bit) as
 * glam_triplet_layer_fwd; the faster route at EVERY batch size.  glam_triplet_layer_ws_supported says whether a shape is inside
 * (36 <= Cp <= 64, H * Cp <= 192, Dp = 4, edge_onehot = 1); other shapes are GLAM_E_UNSUPPORTED here and belong to
 * glam_triplet_layer_fwd.
 * The INFERENCE forward (src_1gp/trainer.py:306-327: @torch.no_grad() evaluation of every split after every epoch): aggr = stats = NULL —
 * both exist only for a backward pass, and the launch then stores neither (15.3 of the 22.5 MB it writes at 1 024 molecules).  Always
 * available here; glam_triplet_layer_fwd takes NULL where glam_triplet_layer_infer_supported says so (the shapes whose update GEMM runs
 * inside the aggregate launch).  xw and a_ij are still written: the aggregate reads them.
 * x = NULL: xw and a_ij HOLD the node product already — the previous application's GRU step wrote them with the rows themselves
 * (glam_gru_ws_fwd_pre_node; /root/reference/src_1gp/model.py:53-54 applies one block message_steps times) — and the call is the
 * aggregate / update launch alone. */
int glam_triplet_layer_ws_supported(int H, int Cp, int Dp, int edge_onehot);
int glam_triplet_layer_infer_supported(int H, int Cp, int Dp);
int glam_triplet_layer_fwd_ell(const float* x, const float* edge_attr, const float* staged, const int32_t* ell_src,
                               const int32_t* ell_eid, int edge_onehot, int64_t N, int64_t E, int H, int Cp, int Dp, float slope,
                               float* xw, float* a_ij, float* aggr, float* stats, float* out, void* stream);

/* The relation-weight table of NNConv with one-hot bond features (src_1gp/layer.py:115-122: nn = Linear(De, hidden) -> ReLU ->
 * Linear(hidden, M), M = in * out): out f32[De, M] = nn(eye(De)), h f32[De, hidden] = the hidden activations (saved for the backward).
 * w1 f32[hidden, De], b1 f32[hidden], w2 f32[M, hidden], b2 f32[M] (torch.nn.Linear layouts).  De <= 8, hidden a power of two in
 * 4..64 (the reference's is 32), M <= 2^22: glam_relation_mlp_supported says whether a shape is inside.  One launch forward, two
 * backward (block partials of d_h summed in block order: bit-reproducible). */
int glam_relation_mlp_supported(int De, int hidden, int64_t M);
size_t glam_relation_mlp_workspace_bytes(int De, int hidden, int64_t M);
int glam_relation_mlp_fwd(const float* w1, const float* b1, const float* w2, const float* b2, int De, int hidden, int64_t M, float* h,
                          float* out, void* stream);
int glam_relation_mlp_bwd(const float* d_out, const float* h, const float* w2, int De, int hidden, int64_t M, float* d_w1, float* d_b1,
                          float* d_w2, float* d_b2, void* ws, size_t ws_bytes, void* stream);

/* dot_and_global_pool5 (src_1gp/layer.py:270-283): for every pair i, [max, mean, median, min, std] of
 * S_i = mol[seg_i] @ pro[seg_i]^T — the reference's Python loop of matmul + max / mean / median / min / std per pair
 * (median = torch.median of the flattened scores: the LOWER median; std unbiased).  One block per pair, the score matrix is
 * never materialised.  D % 4 == 0, D <= 128 (the host zero-pads odd widths).  out f32[P,5]; arg int32[P,6] = global row
 * indices (a, b) of the max, the median and the min element (first occurrence in flattened order), kept for the backward pass.
 * An empty pair yields zeros / -1. */
int glam_pair_pool5_fwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr, int64_t P,
                        int D, float* out, int32_t* arg, void* stream);
int glam_pair_pool5_bwd(const float* mol, const float* pro, const int32_t* mol_ptr, const int32_t* pro_ptr,
                        const float* out, const int32_t* arg, const float* d_out, int64_t P, int D, float* d_mol,
                        float* d_pro, void* stream);

/* Block tail of the convs without a GRU (GCNConv / GATConv blocks, src_1gp/layer.py:248 and :263-266) in one launch per
 * direction: out = act(y + bias + identity) (bias / identity may be NULL; act codes as glam_gru_tail_fwd); the backward
 * returns d_y = d_out * act'(out) (= d_identity; d_bias is its column sum). */
int glam_bias_res_act_fwd(const float* y, const float* bias, const float* identity, int64_t N, int C, int act, float slope,
                          float* out, void* stream);
int glam_bias_res_act_bwd(const float* out, const float* d_out, int64_t N, int C, int act, float slope, float* d_y,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Set2Set readout (src_1gp/model.py:41: PyG Set2Set(C, processing_steps = 3)).
 *   glam_lstm_cell_fwd/bwd: gate math of one torch.nn.LSTM cell step; gates f32[B,4C] = W_ih q* + b_ih + W_hh h + b_hh
 *     (order i|f|g|o) -> h_new, c_new f32[B,C]; the backward recomputes the gates (d_h / d_c may be NULL = zero).
 *   glam_s2s_attn_fwd/bwd: e_n = <x_n, q_g>, a = softmax over the nodes of graph g (denominator + 1e-16), r_g = sum_n a_n x_n
 *     with x f32[N,D], q f32[B,D], ptr int32[B+1] -> r f32[B,D], stats f32[B,2] (segment max, exp-sum); the backward
 *     returns d_x f32[N,D] and d_q f32[B,D].  D % 4 == 0, D <= 128. */
int glam_lstm_cell_fwd(const float* gates, const float* c_prev, int64_t B, int C, float* h_new, float* c_new, void* stream);
int glam_lstm_cell_bwd(const float* gates, const float* c_prev, const float* d_h, const float* d_c, int64_t B, int C,
                       float* d_gates, float* d_c_prev, void* stream);
int glam_s2s_attn_fwd(const float* x, const float* q, const int32_t* ptr, int64_t N, int64_t B, int D, float* r,
                      float* stats, void* stream);
int glam_s2s_attn_bwd(const float* x, const float* q, const float* r, const float* stats, const float* d_r,
                      const int32_t* ptr, int64_t N, int64_t B, int D, float* d_x, float* d_q, void* stream);

/* ---- the GRU step of a MessageBlock in one launch -------------------------------------------------------------------------
 * src_1gp/layer.py:261-266: x = celu(conv out); x, h = GRU(x, h) (one step, seq_len 1); x = act(x + identity).  Both gate linears
 * (gi = celu(x) W_ih^T + b_ih, gh = h W_hh^T + b_hh) on the fp32 matrix cores with the gate math, residual and activation in their
 * epilogue: the same results, bit for bit, as glam_ts_gemm_pair + glam_gru_tail_fwd (resp. glam_gru_tail_rng_fwd), without the round
 * trip of gi / gh between them (both are still written: the backward, glam_gru_tail_*_bwd, recomputes the gates from them).
 * C a multiple of 4, at most 64.  Images: glam_gru_fused_make_images (two buffers of glam_gru_fused_image_bytes(), rebuilt whenever
 * the weights change).  act codes / RNG arguments as glam_gru_tail_fwd / glam_gru_tail_rng_fwd. */
int glam_gru_fused_supported(int C);
size_t glam_gru_fused_image_bytes(void);
int glam_gru_fused_make_images(const float* w_ih, const float* w_hh, int C, float* img_ih, float* img_hh, void* stream);
int glam_gru_fused_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                       const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh,
                       float* h_new, float* out, void* stream);
int glam_gru_fused_rng_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                           const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower,
                           float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new,
                           float* out, float* out_drop, void* stream);

/* The same step (src_1gp/layer.py:261-266) warp-specialised on the bf16 matrix cores in 3 x bf16 form: every fp32 operand split exactly
 * into three bf16 terms, six partial products per product, fp32 accumulation — fp32 accuracy (measured: closer to the fp64 result than
 * the fp32 matrix instructions), roundings differ from glam_gru_fused_fwd at the last bits; RReLU / Dropout use the same Philox words.
 * img_ih / img_hh are the k_ts_gemm images of W_ih^T / W_hh^T (glam_ts_gemm_make_image(w, C, 1, C, 3 C, img): the ones glam_ts_gemm_pair
 * takes).  C a multiple of 4 in 24 .. 64.
 * gh = NULL (every glam_gru_ws_*fwd* form): gi is [N, 4C] and receives THE GATES [r | z | n | gh_n] instead of the two pre-activation
 * matrices — all the backward takes from them (r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n): the values
 * the backward would recompute, bit for bit).  4C instead of 6C floats per row written here and read by glam_gru_bwd_ws* (which takes
 * this matrix when ITS gh is NULL), and no sigmoid / tanh in the backward's vector waves. */
int glam_gru_ws_supported(int C);
int glam_gru_ws_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                    const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh,
                    float* h_new, float* out, void* stream);
int glam_gru_ws_rng_fwd(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                        const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower,
                        float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new,
                        float* out, float* out_drop, void* stream);
/* ... also writing x_celu[N, C] = celu(x), the CELU of src_1gp/layer.py:261 as the launch applies it (celu_in must be set; x_celu may be
 * NULL): the tensor to keep for the backward INSTEAD of x.  glam_gru_bwd_ws(_rng) with celu_in = 2 takes x as celu(x) and forms
 * celu'(x) = x > 0 ? 1 : celu(x) + 1 from it, and the weight gradient of W_ih reads it as Q without a CELU of its own
 * (glam_wgrad_gemm_pair_split(_seg) with qcelu_a = 0): no exponential in either (the one wave that applied the CELU to all of Q set the
 * pace of the weight-gradient launch: 35.3 -> 25.8 us for three applications at N = 20 400). */
int glam_gru_ws_fwd_xc(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                       const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh,
                       float* h_new, float* out, float* x_celu, void* stream);
int glam_gru_ws_rng_fwd_xc(const float* x, const float* h, const float* identity, const float* img_ih, const float* img_hh,
                           const float* b_ih, const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower,
                           float rr_upper, float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new,
                           float* out, float* out_drop, float* x_celu, void* stream);

/* The gate matrices as the matrix waves of the warp-specialised step hold them: every 16 x 8 operand fragment already split into its
 * three bf16 terms, stored in lane order, ONCE per weight update (one launch for both images; either may be NULL).  weight_ih / weight_hh
 * are torch.nn.GRUCell's [3C, C] (layer.py:250), contiguous.  The *_pre entry points below read these instead of the k_ts_gemm images and
 * compute bit-identical results: what they drop is the prologue in which all 256 blocks of every launch re-split the same matrices
 * (~4 k of a block's ~31 k cycles at N = 20 400: forward 18.4 -> 15.2 us, backward 21.6 -> 20.0 us per application).
 * glam_gru_ws_pre_bytes(): size of ONE image (144 KB, 16-byte aligned), any supported C. */
size_t glam_gru_ws_pre_bytes(void);
int glam_gru_ws_make_pre(const float* w_ih, const float* w_hh, int C, void* pre_fwd, void* pre_bwd, void* stream);
int glam_gru_ws_fwd_pre(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih, const float* b_hh,
                        int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh, float* h_new, float* out,
                        float* x_celu, void* stream);
int glam_gru_ws_rng_fwd_pre(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                            const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower, float rr_upper,
                            float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new, float* out,
                            float* out_drop, float* x_celu, void* stream);
/* ... and the node product of the block's NEXT application inside the same launch.  /root/reference/src_1gp/model.py:53-54 applies ONE
 * MessageBlock message_steps times: this step's output rows (`out`; in the rng form the dropped twin `out_drop` when it is given — what
 * layer.py:255-259 hands the conv) are the next TripletMessage's input, and its first product x @ [W_node | Wa] (layer.py:37 + the
 * separable attention columns) needs nothing else.  The producer waves take every finished 16-row tile out of LDS — the consumers leave
 * it there split into bf16 terms — through the 3 x bf16 product against node_pre (that matrix as pre-split operand fragments inside the
 * layer's staged buffer: staged + glam_triplet_staged_node_fragments(H, Cp, Dp); K = C = Cp, node_cols = H * Cp with 56 < node_cols <= 184) while the
 * ring is full, and write xw[N, node_cols] and a_ij[N, 8]: the values glam_ts_gemm / glam_triplet_layer_fwd_ell write for the same
 * rows, bit for bit (same partial products in the same order).  glam_triplet_layer_fwd_ell with x = NULL then starts at its aggregate
 * launch: one launch and one re-read of the rows less per application. */
int glam_gru_ws_fwd_pre_node(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                             const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float* gi, float* gh, float* h_new,
                             float* out, float* x_celu, const void* node_pre, int node_cols, float* xw, float* a_ij, void* stream);
int glam_gru_ws_rng_fwd_pre_node(const float* x, const float* h, const float* identity, const void* pre_fwd, const float* b_ih,
                                 const float* b_hh, int64_t N, int C, int celu_in, int act, float slope, float rr_lower, float rr_upper,
                                 float drop_p, int64_t* rng_state, int64_t* rng_eff, float* gi, float* gh, float* h_new, float* out,
                                 float* out_drop, float* x_celu, const void* node_pre, int node_cols, float* xw, float* a_ij,
                                 void* stream);
int glam_gru_bwd_ws_pre(const float* gi, const float* gh, const float* h, const float* out, const float* d_out, const float* d_hstate,
                        const float* x, const void* pre_bwd, int64_t N, int C, int celu_in, int act, float slope, int merge_identity,
                        float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream);
int glam_gru_bwd_ws_rng_pre(const float* gi, const float* gh, const float* h, const float* out, const float* d_out,
                            const float* d_out_drop, const float* d_hstate, const float* x, const void* pre_bwd, int64_t N, int C,
                            int celu_in, int act, float slope, float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff,
                            int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream);

/* Backward of the same step in ONE launch: glam_gru_tail_bwd (resp. glam_gru_tail_rng_bwd) + the two input-gradient products
 *   d_x = d_gi @ W_ih (* celu'(x) when celu_in), d_h = d_gh @ W_hh + the direct g z path,
 * warp-specialised, the products in 3 x bf16 form.  d_gi / d_gh [N, 3C] are still written (the weight-gradient launch reads them), d_h
 * comes out COMPLETE (the tail kernels' d_h is only the direct part).  img_ih_t / img_hh_t: glam_ts_gemm_make_image(w, C, 0, 3 C, C, img),
 * the images glam_ts_gemm_pair takes for these products.  x is read only with celu_in; d_identity, d_hstate, d_out_drop may be NULL
 * (d_out too in the rng form when d_out_drop is given).  merge_identity = 1: the skip connection and the GRU state are the same tensor
 * (the first application of a block, src_1gp/layer.py:254) — d_h additionally receives d_identity and d_identity is not written.
 * gh = NULL and d_gh = NULL (both or neither; every glam_gru_bwd_ws* form): gi is the gate matrix [N, 4C] of a forward called with
 * gh = NULL, and d_gi is written as [N, 4C] = [d_pr | d_pz | d_pn | d_pn r] — d_gi and d_gh differ in their last block only;
 * glam_wgrad_gemm_gru_gates_seg forms both weight gradients from it.  Same d_x, d_h, d_identity and gate gradients, bit for bit.
 * C a multiple of 4 in 24 .. 64. */
int glam_gru_bwd_ws(const float* gi, const float* gh, const float* h, const float* out, const float* d_out, const float* d_hstate,
                    const float* x, const float* img_ih_t, const float* img_hh_t, int64_t N, int C, int celu_in, int act, float slope,
                    int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream);
int glam_gru_bwd_ws_rng(const float* gi, const float* gh, const float* h, const float* out, const float* d_out, const float* d_out_drop,
                        const float* d_hstate, const float* x, const float* img_ih_t, const float* img_hh_t, int64_t N, int C,
                        int celu_in, int act, float slope, float rr_lower, float rr_upper, float drop_p, const int64_t* rng_eff,
                        int merge_identity, float* d_gi, float* d_gh, float* d_identity, float* d_x, float* d_h, void* stream);

/* ---- narrow-output linear (the model's output head) -------------------------------------------------------------------
 * y[N, M] = x[N, K] @ w[M, K]^T + b for M <= 16, K % 4 == 0: replaces torch.nn.functional.linear / its autograd for
 * `lin_out1 = LinearBlock(e_dim, out_dim)` (reference src_1gp/model.py:47,61; layer.py:223-237), where a GEMM library spends
 * 33 us forward + 22 us backward on what is a row dot product.  b, dx, db may be NULL.  Backward: dx[N, K], dw[M, K], db[M];
 * ws >= glam_linear_narrow_bwd_workspace_bytes(K, M), 16-byte aligned.  Deterministic (fixed-order sums, no atomics). */
int glam_linear_narrow_supported(int K, int M);
int glam_linear_narrow_fwd(const float* x, const float* w, const float* b, int64_t N, int K, int M, float* y, void* stream);
size_t glam_linear_narrow_bwd_workspace_bytes(int K, int M);
int glam_linear_narrow_bwd(const float* x, const float* w, const float* dy, int64_t N, int K, int M, float* dx, float* dw, float* db,
                           void* ws, size_t ws_bytes, void* stream);
/* The head reading the hidden layer's PRE-activation: y = Linear(Dropout(drop_p)(RReLU(rr_lower, rr_upper)(x))) in TRAINING mode — the
 * RReLU of `mol_flat` (/root/reference/src_1gp/model.py:43-45, :60, default activation model.py:31) and the Dropout `lin_out1` starts
 * with (layer.py:232-236, model.py:30) applied to every element as the row dot products read it.  Neither the activated matrix nor its
 * dropped twin is written: the two elementwise launches and their two backward launches are gone.  The words are the ones
 * glam_bias_res_act_rng_fwd draws for the same elements at the same stream position (y equals the three-launch pipeline's bit for
 * bit); rng_eff receives the (seed, offset) pair.  glam_linear_narrow_act_bwd: x is the same pre-activation, dx the gradient of THAT
 * (through the Dropout and the RReLU, words regenerated from rng_eff), dw / db the head's. */
int glam_linear_narrow_act_fwd(const float* x, const float* w, const float* b, int64_t N, int K, int M, float rr_lower, float rr_upper,
                               float drop_p, int64_t* rng_state, int64_t* rng_eff, float* y, void* stream);
int glam_linear_narrow_act_bwd(const float* x, const float* w, const float* dy, int64_t N, int K, int M, float rr_lower, float rr_upper,
                               float drop_p, const int64_t* rng_eff, float* dx, float* dw, float* db, void* ws, size_t ws_bytes,
                               void* stream);

/* Column sums out[D] = sum_n x[n, 0..D) of a row-major f32[N, ld] matrix (D, ld multiples of 4): the bias gradient of a linear whose
 * matrix products stay on the GEMM library (the 300 -> 1024 readout MLP, src_1gp/model.py:44-46).  ws (>= glam_colsum_workspace_bytes)
 * is only touched for N > 2048.  Fixed summation order. */
size_t glam_colsum_workspace_bytes(int D);
int glam_colsum(const float* x, int64_t N, int D, int ld, float* out, void* ws, size_t ws_bytes, void* stream);

/* Square-ish fp32 products on the bf16 matrix cores in 3 x bf16 form (fp32 accuracy; csrc/dense_x3.hip): the readout MLP of the model
 * (`mol_flat` LinearBlock(5 * hid_dim, e_dim = 1024) and wide output heads; /root/reference/src_1gp/model.py:43-45,60-61, LinearBlock
 * src_1gp/layer.py:196-209 = norm -> dropout -> nn.Linear -> act), i.e. torch's F.linear and its autograd for matrices with ~1000 rows.
 *
 *   C[R, Cn] = act( A . B + bias ),  A(r, k) = A[r * a_rs + k * a_ks] * (gate(r, k) > 0 ? 1 : gate_slope),  B(k, c) = B[k * b_ks + c * b_cs]
 *
 * exactly one stride of each pair is 1 (any of the four layout combinations; no transposed copies are made).  gate (same strides as
 * A) and bias[Cn] may be NULL.  act: 0 none, 1 ReLU, 2 LeakyReLU(act_slope), applied after the bias.  rowsum non-NULL (needs
 * b_cs == 1): rowsum[r] = sum_k A(r, k) (a virtual all-ones column of B).  K >= 4; 16-byte accesses are used where alignment and the
 * dimensions allow (multiples of 4), scalar ones elsewhere.  Deterministic (no atomics, no split reductions). */
int glam_dense_gemm(const float* A, int64_t a_rs, int64_t a_ks, const float* gate, float gate_slope, const float* B, int64_t b_ks,
                    int64_t b_cs, const float* bias, int act, float act_slope, float* C, int64_t ldc, float* rowsum, int R, int Cn, int K,
                    void* stream);
/* y[N, M] = act(x[N, K] w[M, K]^T + b) — nn.Linear + a fused ReLU / LeakyReLU — on glam_dense_gemm's kernel (b may be NULL). */
int glam_linear_dense_fwd(const float* x, const float* w, const float* b, int64_t N, int K, int M, int act, float slope, float* y,
                          void* stream);
/* Its backward in ONE launch: with g = dy * (y_gate > 0 ? 1 : gate_slope) (y_gate = the forward's OUTPUT when an activation was
 * fused, NULL otherwise): dx[N, K] = g w, dw[M, K] = g^T x, db[M] = column sums of g.  dx may be NULL; db may be NULL; N >= 4. */
int glam_linear_dense_bwd(const float* x, const float* w, const float* dy, const float* y_gate, float gate_slope, int64_t N, int K, int M,
                          float* dx, float* dw, float* db, void* stream);
/* The same two products with a workspace (glam_dense_ws_bytes() bytes, 16-byte aligned, contents irrelevant; launches that share it must
 * be ordered on one stream) that lets a product of few tiles and a long reduction split k across blocks — dx of the reference's batch
 * of 32 (run.py:40) is 5 tiles of 32 serial chunks beside 160 one-chunk tiles of dw.  Each split writes its partial tile; a second
 * launch adds them in split order and applies bias / activation (run-to-run identical; a launch boundary, not tickets: a device-wide
 * fence between the blocks of one launch writes back and invalidates a whole L2 on this chip).  A product with more than 32 tiles or
 * fewer than 16 chunks of 32 is not split and equals the plain entry points' result bit for bit.  ws == NULL: exactly the plain entry
 * points. */
size_t glam_dense_ws_bytes(void);
int glam_linear_dense_fwd_ws(const float* x, const float* w, const float* b, int64_t N, int K, int M, int act, float slope, float* y,
                             void* ws, size_t ws_bytes, void* stream);
int glam_linear_dense_bwd_ws(const float* x, const float* w, const float* dy, const float* y_gate, float gate_slope, int64_t N, int K,
                             int M, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, void* stream);

/* One Adam step over n parameter tensors in one launch per 40 tensors — the optimizer of the training loop that drives the path
 * (`Adam(self.model.parameters(), lr=args.lr)`, src_1gp/trainer.py:49-50, stepped at trainer.py:301; torch.optim.Adam semantics
 * without amsgrad / maximize; weight_decay is the L2 form).  table: HOST array [n][4] of device addresses {param, grad, exp_avg,
 * exp_avg_sq} (f32, numel[i] elements each, contiguous); step: device f32 count of the steps taken so far, advanced by the launch
 * itself (a captured launch replays correctly); ticket: device u32[GLAM_ADAM_TICKET_WORDS], zero before the first call, owned by the optimizer; lr_dev: device
 * f32 learning rate read by the launch (NULL: the host value `lr`).  In place: param, exp_avg, exp_avg_sq. */
#define GLAM_ADAM_TICKET_WORDS 544
int glam_adam_max_tensors(void);
int glam_adam_step(const uint64_t* table, const int64_t* numel, int n, float* step, unsigned* ticket, const float* lr_dev, double lr,
                   double beta1, double beta2, double eps, double weight_decay, void* stream);

/* The training step's loss, value and gradient in one launch (mean reduction):
 *   kind 0  squared error          — `self.criterion(output, y_true)` with nn.MSELoss, src_1gp/trainer.py:296 / loss.py:42
 *   kind 1  BCE with logits        — nn.BCEWithLogitsLoss, loss.py:48
 *   masked != 0: the mean runs over the elements with target >= 0 only — `criterion(y_score[y_true >= 0], y_true[y_true >= 0])`,
 *   trainer.py:244-245 (labels of -1 are missing, dataset.py:138), without the boolean indexing (not capturable in a hipGraph).
 * pred, target: f32[n]; loss, inv_count: f32[1] (the mean and 1 / count; count = 0 gives nan like the reference's empty mean);
 * grad: f32[n] un-normalised d loss_i / d pred_i; ws >= glam_loss_workspace_bytes(); ticket: device u32[GLAM_ADAM_TICKET_WORDS],
 * zero before the first call, re-armed by every call (n > 1024 only).  Fixed summation order.
 * glam_loss_bwd: d_pred[i] = grad[i] * g_up[0] * inv_count[0] (g_up: the f32[1] gradient arriving at the loss). */
size_t glam_loss_workspace_bytes(void);
int glam_loss_fwd(const float* pred, const float* target, int64_t n, int kind, int masked, float* loss, float* inv_count, float* grad,
                  void* ws, size_t ws_bytes, unsigned* ticket, void* stream);
int glam_loss_bwd(const float* grad, const float* inv_count, const float* g_up, int64_t n, float* d_pred, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLAM_HIP_H */
