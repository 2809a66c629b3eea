// torch extension front end of libglam_hip.so: TORCH_LIBRARY(glam) operators on at::Tensor — the boundary BASELINE.json's
// north_star and SURVEY.md §8(b) name ("the hot path exposed through a torch extension": csr_from_edge_index,
// triplet_aggregate fwd/bwd, segment_pool, segment_softmax_aggregate, sort_pool_topk_last, each wrapped for autograd).
// Every operator only validates (TORCH_CHECK: device, dtype, contiguity, shapes — the reference's convention is a Python
// exception, src_1gp/trainer.py:54), allocates its outputs through the caching allocator, fetches the CURRENT HIP stream and
// forwards to the C ABI of include/glam_hip.h; no arithmetic lives here.  The differentiable operators are C++ autograd nodes
// (torch::autograd::Function), so an eagerly issued step does not pay the Python autograd-node and ctypes marshalling cost.
//
// Host-only C++ (no device code): built by csrc/Makefile with the system compiler against torch's headers.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/glam_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void* cur_stream() { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", glam_last_error()); }

void want(const Tensor& t, at::ScalarType st, const char* name) {
    TORCH_CHECK(t.defined(), name, ": undefined tensor");
    TORCH_CHECK(t.is_cuda(), name, ": glam ops run on an MI355X HIP device only (got a ", t.device(), " tensor); there is no CPU fallback");
    TORCH_CHECK(t.scalar_type() == st, name, ": expected ", st, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": must be contiguous");
}
const float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fpm(Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
const int32_t* ip(const Tensor& t) { return t.defined() ? t.data_ptr<int32_t>() : nullptr; }

// ---- CSR staging (glam_csr_build / glam_batch_ptr) ---------------------------------------------------------------------------
// by = 0: group by target (edge_index[1]); by = 1: by source.  Returns (rowptr i32[N+1], nbr i32[E], eid i32[E], err i32[1]);
// err != 0 <=> an id outside [0, N) (the caller decides when to read it back: no sync here).
std::tuple<Tensor, Tensor, Tensor, Tensor> csr_from_edge_index(const Tensor& edge_index, int64_t N, int64_t by) {
    want(edge_index, at::kLong, "edge_index");
    TORCH_CHECK(edge_index.dim() == 2 && edge_index.size(0) == 2, "edge_index must be int64 [2, E]");
    TORCH_CHECK(N >= 0, "N must be non-negative");
    const int64_t E = edge_index.size(1);
    auto i32 = edge_index.options().dtype(at::kInt);
    Tensor rowptr = at::empty({N + 1}, i32), nbr = at::empty({E}, i32), eid = at::empty({E}, i32), err = at::zeros({1}, i32);
    Tensor ws = at::empty({(int64_t)glam_csr_workspace_bytes(N, E)}, edge_index.options().dtype(at::kByte));
    check_rc(glam_csr_build(edge_index.data_ptr<int64_t>(), N, E, (int)by, rowptr.data_ptr<int32_t>(), nbr.data_ptr<int32_t>(),
                            eid.data_ptr<int32_t>(), err.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
             "glam_csr_build");
    return {rowptr, nbr, eid, err};
}

std::tuple<Tensor, Tensor> batch_ptr(const Tensor& batch, int64_t num_graphs) {
    want(batch, at::kLong, "batch");
    TORCH_CHECK(batch.dim() == 1 && num_graphs >= 0, "batch must be an int64 vector, num_graphs >= 0");
    auto i32 = batch.options().dtype(at::kInt);
    Tensor ptr = at::empty({num_graphs + 1}, i32), err = at::zeros({1}, i32);
    check_rc(glam_batch_ptr(batch.data_ptr<int64_t>(), batch.numel(), num_graphs, ptr.data_ptr<int32_t>(), err.data_ptr<int32_t>(),
                            cur_stream()), "glam_batch_ptr");
    return {ptr, err};
}

// ---- fused gather / attention softmax / scatter-add (glam_triplet_fwd / glam_triplet_bwd) ------------------------------------
struct TripletAggregateFn : public torch::autograd::Function<TripletAggregateFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& xw, const Tensor& a_ij, const Tensor& edge_attr,
                          const c10::optional<Tensor>& w_edge_opt, const Tensor& M, const Tensor& rowptr, const Tensor& src, const Tensor& eid,
                          const Tensor& colptr, const Tensor& dst, const Tensor& eid_t, int64_t H, double slope) {
        const Tensor w_edge = w_edge_opt.has_value() ? *w_edge_opt : Tensor();     // absent: the single-head "light" message alpha * x_j
        want(xw, at::kFloat, "xw"); want(a_ij, at::kFloat, "a_ij"); want(edge_attr, at::kFloat, "edge_attr"); want(M, at::kFloat, "M");
        const bool emul = w_edge.defined() && w_edge.numel() > 0;
        if (emul) want(w_edge, at::kFloat, "w_edge");
        want(rowptr, at::kInt, "rowptr"); want(src, at::kInt, "src"); want(eid, at::kInt, "eid");
        const int64_t N = xw.size(0), E = src.numel(), De = edge_attr.size(1);
        TORCH_CHECK(H >= 1 && H <= 4 && xw.dim() == 2 && xw.size(1) % (4 * H) == 0, "xw must be [N, H*Cp] with Cp % 4 == 0");
        const int Cp = (int)(xw.size(1) / H);
        TORCH_CHECK(a_ij.sizes() == at::IntArrayRef({N, 8}) && edge_attr.size(0) == E && M.sizes() == at::IntArrayRef({De, 4}) &&
                        rowptr.numel() == N + 1, "triplet_aggregate: shape mismatch");
        Tensor aggr = at::empty_like(xw), stats = at::empty({N, 8}, xw.options());
        check_rc(glam_triplet_fwd(fp(xw), fp(a_ij), fp(edge_attr), emul ? fp(w_edge) : nullptr, fp(M), ip(rowptr), ip(src), ip(eid), N, E,
                                  (int)H, Cp, (int)De, emul ? 1 : 0, (float)slope, fpm(aggr), fpm(stats), cur_stream()), "glam_triplet_fwd");
        ctx->save_for_backward({xw, a_ij, edge_attr, emul ? w_edge : Tensor(), M, aggr, stats, rowptr, src, eid, colptr, dst, eid_t});
        ctx->saved_data["H"] = H;
        ctx->saved_data["slope"] = slope;
        return aggr;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto s = ctx->get_saved_variables();
        const Tensor &xw = s[0], &a_ij = s[1], &edge_attr = s[2], &w_edge = s[3], &M = s[4], &aggr = s[5], &stats = s[6], &rowptr = s[7],
                     &src = s[8], &eid = s[9], &colptr = s[10], &dst = s[11], &eid_t = s[12];
        const int64_t H = ctx->saved_data["H"].toInt();
        const double slope = ctx->saved_data["slope"].toDouble();
        const bool emul = w_edge.defined();
        want(colptr, at::kInt, "colptr"); want(dst, at::kInt, "dst"); want(eid_t, at::kInt, "eid_t");
        Tensor d_aggr = grads[0].contiguous();
        const int64_t N = xw.size(0), E = src.numel(), De = edge_attr.size(1);
        const int Cp = (int)(xw.size(1) / H);
        Tensor d_xw = at::empty_like(xw), d_a = at::empty_like(a_ij), d_M = at::empty_like(M);
        Tensor d_we = emul ? at::empty_like(w_edge) : Tensor();
        // the gradient of the bond features only when somebody asks for it (the kernels add into a zeroed buffer)
        Tensor d_ea = ctx->needs_input_grad(2) ? at::zeros_like(edge_attr) : Tensor();
        Tensor ws = at::empty({(int64_t)glam_triplet_bwd_workspace_bytes(N, E, (int)H, Cp, (int)De)}, xw.options().dtype(at::kByte));
        check_rc(glam_triplet_bwd(fp(xw), fp(a_ij), fp(edge_attr), emul ? fp(w_edge) : nullptr, fp(M), fp(aggr), fp(stats), fp(d_aggr),
                                  ip(rowptr), ip(src), ip(eid), ip(colptr), ip(dst), ip(eid_t), N, E, (int)H, Cp, (int)De, emul ? 1 : 0,
                                  (float)slope, fpm(d_xw), fpm(d_a), emul ? fpm(d_we) : nullptr, fpm(d_M), fpm(d_ea), ws.data_ptr(),
                                  (size_t)ws.numel(), cur_stream()), "glam_triplet_bwd");
        return {d_xw, d_a, d_ea, d_we, d_M, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor triplet_aggregate(const Tensor& xw, const Tensor& a_ij, const Tensor& edge_attr, const c10::optional<Tensor>& w_edge, const Tensor& M,
                         const Tensor& rowptr, const Tensor& src, const Tensor& eid, const Tensor& colptr, const Tensor& dst,
                         const Tensor& eid_t, int64_t H, double slope) {
    return TripletAggregateFn::apply(xw, a_ij, edge_attr, w_edge, M, rowptr, src, eid, colptr, dst, eid_t, H, slope);
}

// the forward launches of the layer; `infer`: nothing is kept for a backward pass (aggr = stats = NULL: the kernels store neither)
static Tensor layer_launch(const Tensor& x, const Tensor& edge_attr, const Tensor& wn, const Tensor& we, const Tensor& att, const Tensor& wsc,
                           const Tensor& bias, const Tensor& rowptr, const Tensor& src, const Tensor& eid, int64_t H, double slope, bool ell_f,
                           const c10::optional<Tensor>& ell_src, const c10::optional<Tensor>& ell_eid, bool edge_onehot, bool infer,
                           Tensor& staged, Tensor& xw, Tensor& a_ij, Tensor& aggr, Tensor& stats) {
    const int64_t N = x.size(0), E = src.numel();
    const int C = (int)wn.size(0), De = (int)we.size(0), Cp = (int)x.size(1), Dp = (int)edge_attr.size(1);
    const int HC = (int)H * Cp;
    staged = at::empty({(int64_t)glam_triplet_staged_floats((int)H, Cp, Dp)}, x.options());
    check_rc(glam_triplet_stage_params(fp(wn), fp(we), fp(att), fp(wsc), fp(bias), C, (int)H, De, Cp, Dp, fpm(staged), cur_stream()),
             "glam_triplet_stage_params");
    xw = at::empty({N, HC}, x.options());
    a_ij = at::empty({N, 8}, x.options());
    Tensor out = at::empty({N, Cp}, x.options());
    infer = infer && (ell_f || glam_triplet_layer_infer_supported((int)H, Cp, Dp));
    if (!infer) {
        aggr = at::empty({N, HC}, x.options());
        stats = at::empty({N, 8}, x.options());
    }
    float* aggr_p = infer ? nullptr : fpm(aggr);
    float* stats_p = infer ? nullptr : fpm(stats);
    // molecular graphs (ELL index records, one-hot bond features): the warp-specialised kernels, as the Python autograd node takes them
    if (ell_f) {
        want(*ell_src, at::kInt, "ell_src"); want(*ell_eid, at::kInt, "ell_eid");
        TORCH_CHECK(ell_src->numel() == 4 * N && ell_eid->numel() == 4 * N, "triplet_layer: ELL records are int32 [N, 4]");
        check_rc(glam_triplet_layer_fwd_ell(fp(x), fp(edge_attr), fp(staged), ip(*ell_src), ip(*ell_eid), edge_onehot ? 1 : 0, N, E, (int)H, Cp,
                                            Dp, (float)slope, fpm(xw), fpm(a_ij), aggr_p, stats_p, fpm(out), cur_stream()),
                 "glam_triplet_layer_fwd_ell");
    } else {
        check_rc(glam_triplet_layer_fwd(fp(x), fp(edge_attr), fp(staged), ip(rowptr), ip(src), ip(eid), N, E, (int)H, Cp, Dp,
                                        (float)slope, fpm(xw), fpm(a_ij), aggr_p, stats_p, fpm(out), cur_stream()), "glam_triplet_layer_fwd");
    }
    return out;
}

static void layer_checks(const Tensor& x, const Tensor& edge_attr, const Tensor& wn, const Tensor& we, const Tensor& att, const Tensor& wsc,
                         const Tensor& bias, const Tensor& rowptr, const Tensor& src, const Tensor& eid, int64_t H) {
    want(x, at::kFloat, "x"); want(edge_attr, at::kFloat, "edge_attr"); want(wn, at::kFloat, "weight_node");
    want(we, at::kFloat, "weight_edge"); want(att, at::kFloat, "weight_triplet_att"); want(wsc, at::kFloat, "weight_scale");
    want(bias, at::kFloat, "bias"); want(rowptr, at::kInt, "rowptr"); want(src, at::kInt, "src"); want(eid, at::kInt, "eid");
    const int64_t N = x.size(0), E = src.numel();
    const int C = (int)wn.size(0), De = (int)we.size(0), Cp = (int)x.size(1), Dp = (int)edge_attr.size(1);
    TORCH_CHECK(Cp == (C + 3) / 4 * 4 && (Dp == 4 || Dp == 8) && De <= Dp && wn.size(1) == H * C && wsc.size(0) == H * C && wsc.size(1) == C &&
                    bias.numel() == C && att.numel() == H * 3 * C && edge_attr.size(0) == E && rowptr.numel() == N + 1,
                "triplet_layer: shape mismatch (x must be [N, ceil4(C)], edge_attr [E, 4 | 8])");
}

// ---- the whole TripletMessage layer (glam_triplet_stage_params -> glam_triplet_layer_fwd / glam_triplet_layer_bwd_params) -----
struct TripletLayerFn : public torch::autograd::Function<TripletLayerFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& edge_attr, const Tensor& wn, const Tensor& we, const Tensor& att,
                          const Tensor& wsc, const Tensor& bias, const Tensor& rowptr, const Tensor& src, const Tensor& eid,
                          const Tensor& colptr, const Tensor& dst, const Tensor& eid_t, int64_t H, double slope,
                          const c10::optional<Tensor>& ell_src, const c10::optional<Tensor>& ell_eid, const c10::optional<Tensor>& ell_dst,
                          const c10::optional<Tensor>& ell_eid_t, bool edge_onehot) {
        layer_checks(x, edge_attr, wn, we, att, wsc, bias, rowptr, src, eid, H);
        const int64_t N = x.size(0);
        const int Cp = (int)x.size(1), Dp = (int)edge_attr.size(1);
        Tensor staged, xw, a_ij, aggr, stats;
        const bool ell_f = ell_src.has_value() && ell_src->defined() && ell_eid.has_value() && ell_eid->defined() && N > 0 &&
                           glam_triplet_layer_ws_supported((int)H, Cp, Dp, edge_onehot ? 1 : 0);
        Tensor out = layer_launch(x, edge_attr, wn, we, att, wsc, bias, rowptr, src, eid, H, slope, ell_f, ell_src, ell_eid, edge_onehot, false,
                                  staged, xw, a_ij, aggr, stats);
        const bool ell_b = ell_dst.has_value() && ell_dst->defined() && ell_eid_t.has_value() && ell_eid_t->defined();
        ctx->save_for_backward({x, edge_attr, wn, we, att, staged, xw, a_ij, aggr, stats, rowptr, src, eid, colptr, dst, eid_t,
                                ell_f ? *ell_src : Tensor(), ell_f ? *ell_eid : Tensor(), ell_b ? *ell_dst : Tensor(), ell_b ? *ell_eid_t : Tensor()});
        ctx->saved_data["onehot"] = edge_onehot;
        ctx->saved_data["H"] = H;
        ctx->saved_data["slope"] = slope;
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto s = ctx->get_saved_variables();
        const Tensor &x = s[0], &edge_attr = s[1], &wn = s[2], &we = s[3], &att = s[4], &staged = s[5], &xw = s[6], &a_ij = s[7], &aggr = s[8],
                     &stats = s[9], &rowptr = s[10], &src = s[11], &eid = s[12], &colptr = s[13], &dst = s[14], &eid_t = s[15];
        const int64_t H = ctx->saved_data["H"].toInt();
        const double slope = ctx->saved_data["slope"].toDouble();
        want(colptr, at::kInt, "colptr"); want(dst, at::kInt, "dst"); want(eid_t, at::kInt, "eid_t");
        Tensor d_out = grads[0].contiguous();
        const int64_t N = x.size(0), E = src.numel();
        const int C = (int)wn.size(0), De = (int)we.size(0), Cp = (int)x.size(1), Dp = (int)edge_attr.size(1);
        Tensor d_x = at::empty_like(x);
        // the five parameter gradients are consecutive views of ONE buffer (a data-parallel step all-reduces it as a single bucket)
        const int64_t n_wn = wn.numel(), n_we = we.numel(), n_att = att.numel(), n_ws = H * C * C;
        Tensor flat = at::empty({n_wn + n_we + n_att + n_ws + C}, x.options());
        Tensor d_wn = flat.narrow(0, 0, n_wn).view(wn.sizes()), d_we = flat.narrow(0, n_wn, n_we).view(we.sizes()),
               d_att = flat.narrow(0, n_wn + n_we, n_att).view(att.sizes()), d_wsc = flat.narrow(0, n_wn + n_we + n_att, n_ws).view({H * C, C}),
               d_bias = flat.narrow(0, n_wn + n_we + n_att + n_ws, C);
        Tensor d_ea = ctx->needs_input_grad(1) ? at::zeros_like(edge_attr) : Tensor();
        Tensor ws = at::empty({(int64_t)glam_triplet_layer_bwd_workspace_bytes(N, E, (int)H, Cp, Dp)}, x.options().dtype(at::kByte));
        const Tensor &ls = s[16], &le = s[17], &ld = s[18], &lt = s[19];
        const bool onehot = ctx->saved_data["onehot"].toBool();
        if (ld.defined() && lt.defined() && !d_ea.defined() && N > 0) {
            want(ld, at::kInt, "ell_dst"); want(lt, at::kInt, "ell_eid_t");
            check_rc(glam_triplet_layer_bwd_params_ell(fp(x), fp(edge_attr), fp(staged), fp(xw), fp(a_ij), fp(aggr), fp(stats), fp(d_out),
                                                       ip(rowptr), ip(src), ip(eid), ip(colptr), ip(dst), ip(eid_t), N, E, C, (int)H, De, Cp, Dp,
                                                       (float)slope, fp(wn), fp(we), fp(att), fpm(d_x), fpm(d_wn), fpm(d_we), fpm(d_att),
                                                       fpm(d_wsc), fpm(d_bias), nullptr, nullptr, nullptr, nullptr, nullptr,
                                                       ls.defined() ? ip(ls) : nullptr, le.defined() ? ip(le) : nullptr, ip(ld), ip(lt),
                                                       onehot ? 1 : 0, nullptr, ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
                     "glam_triplet_layer_bwd_params_ell");
        } else {
            check_rc(glam_triplet_layer_bwd_params(fp(x), fp(edge_attr), fp(staged), fp(xw), fp(a_ij), fp(aggr), fp(stats), fp(d_out), ip(rowptr),
                                                   ip(src), ip(eid), ip(colptr), ip(dst), ip(eid_t), N, E, C, (int)H, De, Cp, Dp, (float)slope,
                                                   fp(wn), fp(we), fp(att), fpm(d_x), fpm(d_wn), fpm(d_we), fpm(d_att), fpm(d_wsc), fpm(d_bias),
                                                   fpm(d_ea), ws.data_ptr(), (size_t)ws.numel(), cur_stream()), "glam_triplet_layer_bwd_params");
        }
        return {d_x, d_ea, d_wn, d_we, d_att, d_wsc, d_bias, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor triplet_layer(const Tensor& x, const Tensor& edge_attr, const Tensor& wn, const Tensor& we, const Tensor& att, const Tensor& wsc,
                     const Tensor& bias, const Tensor& rowptr, const Tensor& src, const Tensor& eid, const Tensor& colptr, const Tensor& dst,
                     const Tensor& eid_t, int64_t H, double slope, const c10::optional<Tensor>& ell_src, const c10::optional<Tensor>& ell_eid,
                     const c10::optional<Tensor>& ell_dst, const c10::optional<Tensor>& ell_eid_t, bool edge_onehot) {
    // no backward can follow (torch.no_grad(), or nothing that requires a gradient): the inference forward, outside autograd
    const bool grad = at::GradMode::is_enabled() && (x.requires_grad() || edge_attr.requires_grad() || wn.requires_grad() || we.requires_grad() ||
                                                     att.requires_grad() || wsc.requires_grad() || bias.requires_grad());
    if (!grad) {
        layer_checks(x, edge_attr, wn, we, att, wsc, bias, rowptr, src, eid, H);
        const bool ell_f = ell_src.has_value() && ell_src->defined() && ell_eid.has_value() && ell_eid->defined() && x.size(0) > 0 &&
                           glam_triplet_layer_ws_supported((int)H, (int)x.size(1), (int)edge_attr.size(1), edge_onehot ? 1 : 0);
        Tensor staged, xw, a_ij, aggr, stats;
        return layer_launch(x, edge_attr, wn, we, att, wsc, bias, rowptr, src, eid, H, slope, ell_f, ell_src, ell_eid, edge_onehot, true, staged, xw,
                            a_ij, aggr, stats);
    }
    return TripletLayerFn::apply(x, edge_attr, wn, we, att, wsc, bias, rowptr, src, eid, colptr, dst, eid_t, H, slope, ell_src, ell_eid, ell_dst,
                                 ell_eid_t, edge_onehot);
}

// ---- readouts ------------------------------------------------------------------------------------------------------------------
struct SegmentPoolFn : public torch::autograd::Function<SegmentPoolFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& ptr, int64_t mode) {
        want(x, at::kFloat, "x"); want(ptr, at::kInt, "ptr");
        TORCH_CHECK(x.dim() == 2 && mode >= 0 && mode <= 2, "segment_pool: x must be [N, D], mode in {0 sum, 1 mean, 2 max}");
        const int64_t N = x.size(0), B = ptr.numel() - 1, D = x.size(1);
        Tensor out = at::empty({B, D}, x.options());
        Tensor arg = mode == 2 ? at::empty({B, D}, x.options().dtype(at::kInt)) : Tensor();
        check_rc(glam_segment_pool_fwd(fp(x), ip(ptr), N, B, (int)D, (int)mode, fpm(out), arg.defined() ? arg.data_ptr<int32_t>() : nullptr,
                                       cur_stream()), "glam_segment_pool_fwd");
        ctx->save_for_backward({ptr, arg});
        ctx->saved_data["mode"] = mode;
        ctx->saved_data["N"] = N;
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto s = ctx->get_saved_variables();
        const int64_t mode = ctx->saved_data["mode"].toInt(), N = ctx->saved_data["N"].toInt();
        Tensor d_out = grads[0].contiguous();
        const int64_t B = d_out.size(0), D = d_out.size(1);
        Tensor d_x = at::empty({N, D}, d_out.options());
        check_rc(glam_segment_pool_bwd(fp(d_out), ip(s[0]), s[1].defined() ? s[1].data_ptr<int32_t>() : nullptr, N, B, (int)D, (int)mode,
                                       fpm(d_x), cur_stream()), "glam_segment_pool_bwd");
        return {d_x, Tensor(), Tensor()};
    }
};
Tensor segment_pool(const Tensor& x, const Tensor& ptr, int64_t mode) { return SegmentPoolFn::apply(x, ptr, mode); }

struct SegmentAttnFn : public torch::autograd::Function<SegmentAttnFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& gate, const Tensor& v, const Tensor& ptr) {
        want(gate, at::kFloat, "gate"); want(v, at::kFloat, "v"); want(ptr, at::kInt, "ptr");
        const int64_t N = v.size(0), B = ptr.numel() - 1, D = v.size(1);
        TORCH_CHECK(gate.numel() == N, "segment_softmax_aggregate: one gate per row of v");
        Tensor out = at::empty({B, D}, v.options()), stats = at::empty({B, 2}, v.options());
        check_rc(glam_segment_attn_fwd(fp(gate), fp(v), ip(ptr), N, B, (int)D, fpm(out), fpm(stats), cur_stream()), "glam_segment_attn_fwd");
        ctx->save_for_backward({gate, v, out, stats, ptr});
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto s = ctx->get_saved_variables();
        Tensor d_out = grads[0].contiguous();
        const int64_t N = s[1].size(0), B = s[4].numel() - 1, D = s[1].size(1);
        Tensor d_gate = at::empty_like(s[0]), d_v = at::empty_like(s[1]);
        check_rc(glam_segment_attn_bwd(fp(s[0]), fp(s[1]), fp(s[2]), fp(s[3]), fp(d_out), ip(s[4]), N, B, (int)D, fpm(d_gate), fpm(d_v),
                                       cur_stream()), "glam_segment_attn_bwd");
        return {d_gate, d_v, Tensor()};
    }
};
Tensor segment_softmax_aggregate(const Tensor& gate, const Tensor& v, const Tensor& ptr) { return SegmentAttnFn::apply(gate, v, ptr); }

struct Pool5Fn : public torch::autograd::Function<Pool5Fn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& ptr, int64_t k) {
        want(x, at::kFloat, "x"); want(ptr, at::kInt, "ptr");
        TORCH_CHECK(x.dim() == 2 && k >= 1 && k <= 8, "global_pool5: x must be [N, D], 1 <= k <= 8");
        const int64_t N = x.size(0), B = ptr.numel() - 1, D = x.size(1);
        Tensor out = at::empty({B, (2 + k) * D}, x.options()), topk = at::empty({B, k}, x.options().dtype(at::kInt));
        check_rc(glam_pool5_fwd(fp(x), ip(ptr), N, B, (int)D, (int)k, fpm(out), topk.data_ptr<int32_t>(), cur_stream()), "glam_pool5_fwd");
        ctx->save_for_backward({ptr, topk});
        ctx->saved_data["N"] = N;
        ctx->saved_data["k"] = k;
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto s = ctx->get_saved_variables();
        const int64_t N = ctx->saved_data["N"].toInt(), k = ctx->saved_data["k"].toInt();
        Tensor d_out = grads[0].contiguous();
        const int64_t B = d_out.size(0), D = d_out.size(1) / (2 + k);
        Tensor d_x = at::empty({N, D}, d_out.options());
        check_rc(glam_pool5_bwd(fp(d_out), ip(s[0]), ip(s[1]), N, B, (int)D, (int)k, fpm(d_x), cur_stream()), "glam_pool5_bwd");
        return {d_x, Tensor(), Tensor()};
    }
};
// mean | add | sort-pool(k) readout [B, (2 + k) D]; sort_pool_topk_last is its last k*D columns
Tensor global_pool5(const Tensor& x, const Tensor& ptr, int64_t k) { return Pool5Fn::apply(x, ptr, k); }
Tensor sort_pool_topk_last(const Tensor& x, const Tensor& ptr, int64_t k) { return Pool5Fn::apply(x, ptr, k).slice(1, 2 * x.size(1)); }

}  // namespace

TORCH_LIBRARY(glam, m) {
    // this shim was compiled against one set of C signatures: a libglam_hip.so of another ABI version would be called with shifted arguments
    TORCH_CHECK(glam_abi_version() == GLAM_ABI_VERSION, "_glam_torch.so was built for ABI version ", GLAM_ABI_VERSION,
                " of libglam_hip.so, the loaded library reports ", glam_abi_version(), " (rebuild: make -C glam_amd/csrc)");
    m.def("csr_from_edge_index(Tensor edge_index, int N, int by=0) -> (Tensor, Tensor, Tensor, Tensor)", &csr_from_edge_index);
    m.def("batch_ptr(Tensor batch, int num_graphs) -> (Tensor, Tensor)", &batch_ptr);
    m.def("triplet_aggregate(Tensor xw, Tensor a_ij, Tensor edge_attr, Tensor? w_edge, Tensor M, Tensor rowptr, Tensor src, Tensor eid, "
          "Tensor colptr, Tensor dst, Tensor eid_t, int heads, float slope=0.2) -> Tensor", &triplet_aggregate);
    m.def("triplet_layer(Tensor x, Tensor edge_attr, Tensor weight_node, Tensor weight_edge, Tensor weight_triplet_att, Tensor weight_scale, "
          "Tensor bias, Tensor rowptr, Tensor src, Tensor eid, Tensor colptr, Tensor dst, Tensor eid_t, int heads, float slope=0.2, "
          "Tensor? ell_src=None, Tensor? ell_eid=None, Tensor? ell_dst=None, Tensor? ell_eid_t=None, bool edge_onehot=False) -> Tensor",
          &triplet_layer);
    m.def("segment_pool(Tensor x, Tensor ptr, int mode) -> Tensor", &segment_pool);
    m.def("segment_softmax_aggregate(Tensor gate, Tensor v, Tensor ptr) -> Tensor", &segment_softmax_aggregate);
    m.def("global_pool5(Tensor x, Tensor ptr, int k=3) -> Tensor", &global_pool5);
    m.def("sort_pool_topk_last(Tensor x, Tensor ptr, int k=3) -> Tensor", &sort_pool_topk_last);
}
