"""Readouts, segment reductions, graph norms, block tails and per-pair fusion pools (src_1gp/layer.py:161-220, 270-283;
src_1gp/model.py:41).

Part of ``glam_amd.ops`` (every public name here is re-exported there: ``from glam_amd import ops; ops.pool5(...)``).  Knobs, the weight
scope, the padded-column bookkeeping and the index staging live in ``glam_amd/ops.py`` and are read through ``_o`` at call time."""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib
from . import ops as _o
from ._lib import GlamHipError, check, f32c, ptr, require_device, stream

# --------------------------------------------------------------------------------------
# readouts
# --------------------------------------------------------------------------------------
class _Pool5(torch.autograd.Function):
    """``x`` holds ``D`` channels in rows of ``x.size(1) >= D`` floats (zero padding behind them: the padded flow of odd widths)."""

    @staticmethod
    def forward(ctx, x, sp, k, D):
        require_device(x)
        x = f32c(x, "x")
        N, ld = x.shape
        if N != sp.N:
            raise GlamHipError(f"pool: x has {N} rows but batch has {sp.N}")
        out = torch.empty(sp.B, (2 + k) * D, dtype=torch.float32, device=x.device)
        topk = torch.empty(sp.B, k, dtype=torch.int32, device=x.device)
        check(_lib.load().glam_pool5_padded_fwd(ptr(x), ptr(sp.ptr), N, sp.B, ld, D, k, ptr(out), ptr(topk), stream()),
              "glam_pool5_fwd")
        ctx.save_for_backward(topk)
        ctx.sp, ctx.dims = sp, (N, ld, D, k)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        (topk,) = ctx.saved_tensors
        N, ld, D, k = ctx.dims
        sp = ctx.sp
        d_out = f32c(d_out, "d_out")
        d_x = torch.empty(N, ld, dtype=torch.float32, device=d_out.device)
        check(_lib.load().glam_pool5_padded_bwd(ptr(d_out), ptr(sp.ptr), ptr(topk), N, sp.B, ld, D, k, ptr(d_x), stream()),
              "glam_pool5_bwd")
        return d_x, None, None, None


def pool5(x, sp, k=3):
    """mean || add || sort-pool(k) readout, ``[B, (2+k)*D]``."""
    base = _o.padded_base(x) if x.dim() == 2 else None
    if base is not None and base.size(1) <= 128 and base.size(1) - x.size(1) < 4:
        return _Pool5.apply(base, sp, k, x.size(1))      # odd widths: straight from the zero-padded rows, no compaction copy
    return _Pool5.apply(x, sp, k, x.size(1))


_MODES = {"sum": 0, "add": 0, "mean": 1, "max": 2}


class _SegmentPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sp, mode):
        require_device(x)
        x = f32c(x, "x")
        N, D = x.shape
        if N != sp.N:
            raise GlamHipError(f"pool: x has {N} rows but batch has {sp.N}")
        out = torch.empty(sp.B, D, dtype=torch.float32, device=x.device)
        argmax = torch.empty(sp.B, D, dtype=torch.int32, device=x.device) if mode == 2 else None
        check(_lib.load().glam_segment_pool_fwd(ptr(x), ptr(sp.ptr), N, sp.B, D, mode, ptr(out), ptr(argmax), stream()),
              "glam_segment_pool_fwd")
        ctx.sp, ctx.dims, ctx.argmax = sp, (N, D, mode), argmax
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        N, D, mode = ctx.dims
        sp = ctx.sp
        d_out = f32c(d_out, "d_out")
        d_x = torch.empty(N, D, dtype=torch.float32, device=d_out.device)
        check(_lib.load().glam_segment_pool_bwd(ptr(d_out), ptr(sp.ptr), ptr(ctx.argmax), N, sp.B, D, mode, ptr(d_x),
                                                stream()), "glam_segment_pool_bwd")
        return d_x, None, None


def segment_pool(x, sp, reduce="sum"):
    """``scatter(x, batch, dim=0, reduce)`` over the sorted ``batch`` behind ``sp``."""
    squeeze = x.dim() == 1
    out = _SegmentPool.apply(x.unsqueeze(-1) if squeeze else x, sp, _MODES[reduce])
    return out.squeeze(-1) if squeeze else out


class _SegmentAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, v, sp):
        require_device(gate, v)
        gate, v = f32c(gate.reshape(-1), "gate"), f32c(v, "v")
        N, D = v.shape
        if gate.numel() != N or N != sp.N:
            raise GlamHipError("segment_attention: gate / v / batch disagree on the node count")
        out = torch.empty(sp.B, D, dtype=torch.float32, device=v.device)
        stats = torch.empty(sp.B, 2, dtype=torch.float32, device=v.device)
        check(_lib.load().glam_segment_attn_fwd(ptr(gate), ptr(v), ptr(sp.ptr), N, sp.B, D, ptr(out), ptr(stats), stream()),
              "glam_segment_attn_fwd")
        ctx.save_for_backward(gate, v, out, stats)
        ctx.sp = sp
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        gate, v, out, stats = ctx.saved_tensors
        sp = ctx.sp
        N, D = v.shape
        d_out = f32c(d_out, "d_out")
        d_gate, d_v = torch.empty_like(gate), torch.empty_like(v)
        check(_lib.load().glam_segment_attn_bwd(ptr(gate), ptr(v), ptr(out), ptr(stats), ptr(d_out), ptr(sp.ptr), N,
                                                sp.B, D, ptr(d_gate), ptr(d_v), stream()), "glam_segment_attn_bwd")
        return d_gate, d_v, None


class _BiasResAct(torch.autograd.Function):
    """``act(y + bias + identity)``: the tail of a MessageBlock whose conv has no GRU (GCN / GAT), one launch per direction.
    ``rng = (rr_lower, rr_upper, drop_p)``: training-mode RReLU (``act == 4``) and / or a second output ``Dropout(drop_p)(out)``
    from the device-side Philox stream; ``want_out=False`` with ``act == 0`` is a plain Dropout.  Returns ``(out, out_drop)``."""

    @staticmethod
    def forward(ctx, y, bias, identity, act, slope, rng=None, want_out=True):
        ctx.set_materialize_grads(False)
        require_device(y, bias, identity)
        y = f32c(y, "y")
        bias = None if bias is None else f32c(bias, "bias")
        identity = None if identity is None else f32c(identity, "identity")
        N, C = y.shape
        out = torch.empty_like(y) if want_out else None
        out_drop, eff = None, None
        if rng is None:
            if act == _o.ACT_CODES["rrelu"] or not want_out:
                raise GlamHipError("bias_res_act: 'rrelu' / dropout-only need rng=(lower, upper, drop_p)")
            check(_lib.load().glam_bias_res_act_fwd(ptr(y), ptr(bias), ptr(identity), N, C, act, float(slope), ptr(out), stream()),
                  "glam_bias_res_act_fwd")
        else:
            lo, hi, p = (float(v) for v in rng)
            eff = torch.empty(2, dtype=torch.int64, device=y.device)
            out_drop = torch.empty_like(y) if p > 0 else None
            check(_lib.load().glam_bias_res_act_rng_fwd(ptr(y), ptr(bias), ptr(identity), N, C, act, float(slope), lo, hi, p,
                                                        ptr(_o.rng_state(y.device)), ptr(eff), ptr(out), ptr(out_drop), stream()),
                  "glam_bias_res_act_rng_fwd")
        ctx.save_for_backward(*([out] if out is not None else []))
        ctx.eff = eff
        ctx.shape = (N, C)
        ctx.cfg = (act, float(slope), bias is not None, identity is not None, None if rng is None else tuple(float(v) for v in rng))
        return out, out_drop

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_out_drop=None):
        out = ctx.saved_tensors[0] if ctx.saved_tensors else None
        act, slope, has_bias, has_id, rng = ctx.cfg
        N, C = ctx.shape
        d_out = None if d_out is None else f32c(d_out, "d_out")
        d_out_drop = None if d_out_drop is None else f32c(d_out_drop, "d_out_drop")
        ref = d_out if d_out is not None else d_out_drop
        d_y = torch.empty_like(ref)
        if rng is None:
            check(_lib.load().glam_bias_res_act_bwd(ptr(out), ptr(d_out), N, C, act, slope, ptr(d_y), stream()), "glam_bias_res_act_bwd")
        else:
            check(_lib.load().glam_bias_res_act_rng_bwd(ptr(out), ptr(d_out), ptr(d_out_drop), N, C, act, slope, rng[0], rng[1], rng[2],
                                                        ptr(ctx.eff), ptr(d_y), stream()), "glam_bias_res_act_rng_bwd")
        return d_y, (d_y.sum(0) if has_bias else None), (d_y if has_id else None), None, None, None, None


def bias_res_act(y, bias, identity, act="none", slope=0.0, rng=None):
    out, out_drop = _BiasResAct.apply(y, bias, identity, _o.ACT_CODES[act], slope, rng, True)
    if out_drop is not None:
        _o.register_dropped(out, out_drop, rng[2])
    return out


def rrelu(x, lower=1.0 / 8, upper=1.0 / 3, drop_p=0.0):
    """Training-mode ``torch.nn.RReLU(lower, upper)`` on the device-side Philox stream (one launch per direction, slopes
    regenerated in the backward); ``drop_p > 0`` also writes the dropped twin for a ``Dropout(drop_p)`` that follows."""
    shape = x.shape
    out, out_drop = _BiasResAct.apply(x.reshape(-1, shape[-1]), None, None, _o.ACT_CODES["rrelu"], 0.0, (lower, upper, drop_p), True)
    out = out.view(shape)
    if out_drop is not None:
        _o.register_dropped(out, out_drop.view(shape), drop_p)
    return out


def dropout(x, p):
    """Training-mode ``torch.nn.Dropout(p)``: ``x * mask / (1 - p)``; the mask is regenerated in the backward (no mask tensor)."""
    shape = x.shape
    _, out_drop = _BiasResAct.apply(x.reshape(-1, shape[-1]), None, None, _o.ACT_CODES["none"], 0.0, (1.0, 1.0, float(p)), False)
    return out_drop.view(shape)


class _LstmCell(torch.autograd.Function):
    """Gate math of one ``torch.nn.LSTM`` cell step (Set2Set): ``(gates[B,4C], c[B,C]) -> (h', c')``, one launch each way."""

    @staticmethod
    def forward(ctx, gates, c_prev):
        require_device(gates, c_prev)
        gates, c_prev = f32c(gates, "gates"), f32c(c_prev, "c")
        B, C = c_prev.shape
        h_new, c_new = torch.empty_like(c_prev), torch.empty_like(c_prev)
        check(_lib.load().glam_lstm_cell_fwd(ptr(gates), ptr(c_prev), B, C, ptr(h_new), ptr(c_new), stream()), "glam_lstm_cell_fwd")
        ctx.save_for_backward(gates, c_prev)
        return h_new, c_new

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_h, d_c):
        gates, c_prev = ctx.saved_tensors
        B, C = c_prev.shape
        d_h = None if d_h is None else f32c(d_h, "d_h")
        d_c = None if d_c is None else f32c(d_c, "d_c")
        d_gates, d_cp = torch.empty_like(gates), torch.empty_like(c_prev)
        check(_lib.load().glam_lstm_cell_bwd(ptr(gates), ptr(c_prev), ptr(d_h), ptr(d_c), B, C, ptr(d_gates), ptr(d_cp), stream()),
              "glam_lstm_cell_bwd")
        return d_gates, d_cp


def lstm_cell(gates, c_prev):
    return _LstmCell.apply(gates, c_prev)


class _QueryAttention(torch.autograd.Function):
    """Set2Set's attention read ``r_g = sum_n softmax_n(<x_n, q_g>) x_n`` with the logits formed inside the kernel."""

    @staticmethod
    def forward(ctx, x, q, sp):
        require_device(x, q)
        x, q = f32c(x, "x"), f32c(q, "q")
        N, D = x.shape
        if N != sp.N or q.shape != (sp.B, D):
            raise GlamHipError("query_attention: x / q disagree with the batch vector")
        r = torch.empty(sp.B, D, dtype=torch.float32, device=x.device)
        stats = torch.empty(sp.B, 2, dtype=torch.float32, device=x.device)
        check(_lib.load().glam_s2s_attn_fwd(ptr(x), ptr(q), ptr(sp.ptr), N, sp.B, D, ptr(r), ptr(stats), stream()), "glam_s2s_attn_fwd")
        ctx.save_for_backward(x, q, r, stats)
        ctx.sp = sp
        return r

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_r):
        x, q, r, stats = ctx.saved_tensors
        sp = ctx.sp
        d_r = f32c(d_r, "d_r")
        d_x, d_q = torch.empty_like(x), torch.empty_like(q)
        check(_lib.load().glam_s2s_attn_bwd(ptr(x), ptr(q), ptr(r), ptr(stats), ptr(d_r), ptr(sp.ptr), x.size(0), sp.B, x.size(1),
                                            ptr(d_x), ptr(d_q), stream()), "glam_s2s_attn_bwd")
        return d_x, d_q, None


def query_attention_supported(D):
    return D % 4 == 0 and D <= 128


def query_attention(x, q, sp):
    return _QueryAttention.apply(x, q, sp)


def segment_attention(gate, v, sp):
    """``scatter_add(softmax(gate, batch) * v, batch)`` -> ``[B, D]`` (GlobalAttention / Set2Set)."""
    return _SegmentAttn.apply(gate, v, sp)


class _EdgeReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, msg, gi, mode):
        require_device(msg)
        msg = f32c(msg, "msg")
        E, D = msg.shape
        if E != gi.E:
            raise GlamHipError(f"edge_reduce: {E} messages for {gi.E} edges")
        out = torch.empty(gi.N, D, dtype=torch.float32, device=msg.device)
        argmax = torch.empty(gi.N, D, dtype=torch.int32, device=msg.device) if mode == 2 else None
        check(_lib.load().glam_edge_reduce_fwd(ptr(msg), ptr(gi.rowptr), ptr(gi.eid), gi.N, E, D, mode, ptr(out),
                                               ptr(argmax), stream()), "glam_edge_reduce_fwd")
        ctx.gi, ctx.dims, ctx.argmax = gi, (E, D, mode), argmax
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        E, D, mode = ctx.dims
        gi = ctx.gi
        d_out = f32c(d_out, "d_out")
        d_msg = torch.empty(E, D, dtype=torch.float32, device=d_out.device)
        check(_lib.load().glam_edge_reduce_bwd(ptr(d_out), ptr(gi.rowptr), ptr(gi.eid), ptr(ctx.argmax), gi.N, E, D,
                                               mode, ptr(d_msg), stream()), "glam_edge_reduce_bwd")
        return d_msg, None, None


def edge_reduce(msg, gi, reduce="sum"):
    """``scatter(msg, edge_index[1], dim=0, dim_size=N, reduce)`` over the CSR-by-target in ``gi``."""
    squeeze = msg.dim() == 1
    out = _EdgeReduce.apply(msg.unsqueeze(-1) if squeeze else msg, gi, _MODES[reduce])
    return out.squeeze(-1) if squeeze else out


class _GraphNorm(torch.autograd.Function):
    """``with_identity``: the op also returns ``x`` itself as a second output — the skip connection of a MessageBlock
    (src_1gp/layer.py:253-265: ``x`` feeds the norm AND ``x + identity``).  Both gradient paths then arrive at THIS node and the
    backward kernel sums them in its store (glam_graph_norm_bwd_add) instead of autograd launching an add."""

    @staticmethod
    def forward(ctx, x, sp, mode, scale, eps, with_identity=False, drop_p=0.0):
        require_device(x)
        x = f32c(x, "x")
        N, D = x.shape
        if N != sp.N:
            raise GlamHipError(f"graph_norm: x has {N} rows but batch has {sp.N}")
        ctx.drop_p, ctx.eff = float(drop_p), None
        if drop_p > 0:
            # the training-mode Dropout(drop_p) behind the norm from the same launch: the output IS the dropped tensor (graph_norm_drop_supported)
            y = torch.empty_like(x)
            ctx.eff = torch.empty(2, dtype=torch.int64, device=x.device)
            check(_lib.load().glam_graph_norm_drop_fwd(ptr(x), ptr(sp.ptr), N, sp.B, D, mode, float(scale), float(eps), float(drop_p),
                                                       ptr(_o.rng_state(x.device)), ptr(ctx.eff), None, ptr(y), stream()),
                  "glam_graph_norm_drop_fwd")
        else:
            y = torch.zeros_like(x) if sp.B == 0 else torch.empty_like(x)
            check(_lib.load().glam_graph_norm_fwd(ptr(x), ptr(sp.ptr), N, sp.B, D, mode, float(scale), float(eps), ptr(y), stream()),
                  "glam_graph_norm_fwd")
        ctx.save_for_backward(x)
        ctx.sp, ctx.cfg = sp, (mode, float(scale), float(eps))
        if with_identity:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy, d_id=None):
        (x,) = ctx.saved_tensors
        mode, scale, eps = ctx.cfg
        sp = ctx.sp
        N, D = x.shape
        if gy is None:                  # only the identity output was used
            return d_id, None, None, None, None, None, None
        gy = f32c(gy, "gy")
        dx = torch.empty_like(x)
        lib = _lib.load()
        if ctx.drop_p > 0:              # gy is the gradient of the DROPPED output: the mask is regenerated inside the launch
            check(lib.glam_graph_norm_drop_bwd(ptr(x), None, ptr(gy), ptr(sp.ptr), N, sp.B, D, mode, scale, eps, ctx.drop_p, ptr(ctx.eff),
                                               ptr(None if d_id is None else f32c(d_id, "d_identity")), ptr(dx), stream()),
                  "glam_graph_norm_drop_bwd")
            return dx, None, None, None, None, None, None
        if d_id is not None and N > 0 and sp.B > 0:
            check(lib.glam_graph_norm_bwd_add(ptr(x), ptr(gy), ptr(sp.ptr), N, sp.B, D, mode, scale, eps, ptr(f32c(d_id, "d_identity")), ptr(dx),
                                              stream()), "glam_graph_norm_bwd_add")
            return dx, None, None, None, None, None, None
        check(lib.glam_graph_norm_bwd(ptr(x), ptr(gy), ptr(sp.ptr), N, sp.B, D, mode, scale, eps, ptr(dx), stream()),
              "glam_graph_norm_bwd")
        if d_id is not None:
            dx = d_id if (N == 0 or sp.B == 0) else dx + d_id
        return dx, None, None, None, None, None, None


def graph_norm_drop_supported(x, sp):
    """The norm + Dropout launch pair exists for this shape (molecule-sized graphs, a multiple of 4 channels <= 64)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and sp.B > 0
            and _lib.load().glam_graph_norm_drop_supported(x.size(0), sp.B, x.size(1)) == 1)


def pair_norm(x, sp, scale=1.0, eps=1e-5, with_identity=False, drop_p=0.0):
    """PyG ``PairNorm(scale, eps=1e-5)(x, batch)`` (one kernel per direction); ``with_identity``: ``(y, x)`` — see _GraphNorm;
    ``drop_p > 0`` (``graph_norm_drop_supported``): the result is ``Dropout(drop_p)(PairNorm(x))`` in training mode, from the one launch."""
    return _GraphNorm.apply(x, sp, 0, scale, eps, with_identity, drop_p)


def graph_standardize(x, sp, eps=1e-5, with_identity=False):
    """Statistics part of PyG's graph ``LayerNorm(x, batch)``: zero mean / unit variance per graph."""
    return _GraphNorm.apply(x, sp, 1, 1.0, eps, with_identity, 0.0)


class _EdgeWeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gi, mean, self_slot, with_identity=False):
        require_device(x, w)
        x_in = x
        x, w = f32c(x, "x"), f32c(w, "w")
        N, D = x.shape
        E, K = w.shape
        if N != gi.N or E != gi.E:
            raise GlamHipError("edge_weighted_sum: x / w disagree with the edge list")
        out = torch.empty(N, K + int(self_slot), D, dtype=torch.float32, device=x.device)
        check(_lib.load().glam_edge_wsum_fwd(ptr(x), ptr(w), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), N, E, D, K, int(mean),
                                             int(self_slot), ptr(out), stream()), "glam_edge_wsum_fwd")
        ctx.save_for_backward(w)
        ctx.gi, ctx.cfg = gi, (N, E, D, K, int(mean), int(self_slot))
        ctx.aliased = bool(with_identity)
        if ctx.aliased:
            ctx.set_materialize_grads(False)
            return out, x_in.view_as(x_in)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_alias=None):
        (w,) = ctx.saved_tensors
        gi = ctx.gi
        N, E, D, K, mean, self_slot = ctx.cfg
        if d_out is None:                    # (only with the alias: the sums themselves were not used)
            return d_alias, None, None, None, None, None
        d_out = f32c(d_out, "d_out")
        colptr, dst, eid_t = gi.transpose()
        dx = torch.empty(N, D, dtype=torch.float32, device=d_out.device)
        lib = _lib.load()
        if d_alias is not None and K in (4, 8) and D % 4 == 0:
            # the skip connection's gradient joins the sums in this launch (no add launch of the autograd engine)
            d_alias = f32c(d_alias, "d_identity")
            check(lib.glam_edge_wsum_bwd_add(ptr(d_out), ptr(w), ptr(colptr), ptr(dst), ptr(eid_t), ptr(gi.rowptr), N, E, D, K, mean, self_slot,
                                             ptr(d_alias), ptr(dx), stream()), "glam_edge_wsum_bwd_add")
            return dx, None, None, None, None, None
        check(lib.glam_edge_wsum_bwd(ptr(d_out), ptr(w), ptr(colptr), ptr(dst), ptr(eid_t), ptr(gi.rowptr), N, E, D, K,
                                     mean, self_slot, ptr(dx), stream()), "glam_edge_wsum_bwd")
        return (dx if d_alias is None else dx + d_alias), None, None, None, None, None


def edge_weighted_sum(x, w, gi, mean=False, self_slot=False, with_identity=False):
    """``S[n,k,:] = (1/deg_n) sum_{e->n} w[e,k] * x[src_e,:]`` -> ``[N, K, D]`` (no gradient w.r.t. ``w``: edge data);
    ``self_slot``: ``[N, K+1, D]`` with ``S[n,K,:] = x[n,:]`` (K in {4, 8}, D % 4 == 0).  ``with_identity``: returns ``(S, identity)`` with
    ``identity`` = ``x`` handed back through this node (a skip connection around the caller then sends its gradient here: it is added
    inside the backward launch)."""
    if with_identity:
        if torch.is_grad_enabled() and x.requires_grad:
            return _EdgeWeightedSum.apply(x, w, gi, mean, self_slot, True)
        return _EdgeWeightedSum.apply(x, w, gi, mean, self_slot), x
    return _EdgeWeightedSum.apply(x, w, gi, mean, self_slot)


def self_slot_supported(K, D):
    return K in (4, 8) and D % 4 == 0


_ONEHOT_CACHE: dict = {}


class _PairPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mol, pro, msp, psp, with_identity=False):
        require_device(mol, pro)
        mol_in, pro_in = mol, pro
        mol, pro = f32c(mol, "mol_out"), f32c(pro, "pro_out")
        if msp.B != psp.B or mol.size(1) != pro.size(1) or mol.size(0) != msp.N or pro.size(0) != psp.N:
            raise GlamHipError("pair_pool: the two batches disagree (pair count / width / node count)")
        P, D = msp.B, mol.size(1)
        out = torch.empty(P, 2, dtype=torch.float32, device=mol.device)
        arg = torch.empty(P, 2, dtype=torch.int32, device=mol.device)
        sums = torch.empty(P, 2, D, dtype=torch.float32, device=mol.device)
        lib = _lib.load()
        ws = torch.empty(max(lib.glam_pair_pool_workspace_bytes(P, D), 16), dtype=torch.uint8, device=mol.device)
        check(lib.glam_pair_pool_fwd(ptr(mol), ptr(pro), ptr(msp.ptr), ptr(psp.ptr), P, D, ptr(out), ptr(arg), ptr(sums), ptr(ws),
                                     ws.numel(), stream()), "glam_pair_pool_fwd")
        ctx.save_for_backward(mol, pro, arg, sums)
        ctx.sps = (msp, psp)
        ctx.aliased = bool(with_identity)
        if ctx.aliased:      # the two inputs come back as outputs: what the caller does with them next sends its gradient HERE
            ctx.set_materialize_grads(False)
            return out, mol_in.view_as(mol_in), pro_in.view_as(pro_in)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_ma=None, d_pa=None):
        mol, pro, arg, sums = ctx.saved_tensors
        msp, psp = ctx.sps
        if d_out is None:                    # (only with the aliases: the fusion values themselves were not used)
            return d_ma, d_pa, None, None, None
        d_out = f32c(d_out, "d_out")
        d_mol, d_pro = torch.empty_like(mol), torch.empty_like(pro)
        lib = _lib.load()
        if d_ma is not None or d_pa is not None:
            d_ma = None if d_ma is None else f32c(d_ma, "d_mol (next use)")
            d_pa = None if d_pa is None else f32c(d_pa, "d_pro (next use)")
            check(lib.glam_pair_pool_bwd_add(ptr(mol), ptr(pro), ptr(msp.ptr), ptr(psp.ptr), ptr(arg), ptr(sums), ptr(d_out), msp.B, mol.size(1),
                                             ptr(d_ma), ptr(d_pa), ptr(d_mol), ptr(d_pro), stream()), "glam_pair_pool_bwd_add")
        else:
            check(lib.glam_pair_pool_bwd(ptr(mol), ptr(pro), ptr(msp.ptr), ptr(psp.ptr), ptr(arg), ptr(sums), ptr(d_out), msp.B,
                                         mol.size(1), ptr(d_mol), ptr(d_pro), stream()), "glam_pair_pool_bwd")
        return d_mol, d_pro, None, None, None


def pair_pool(mol_out, pro_out, msp, psp, with_identity=False):
    """``[max, mean]`` of ``mol[seg_i] @ pro[seg_i].T`` per pair -> ``[P, 2]`` (dot_and_global_pool2).  ``with_identity``: returns
    ``(out, mol_out, pro_out)`` with the two matrices handed back through this node where that saves the add launches of their second
    use (the next message step): its gradient is then added inside this node's backward launch."""
    if with_identity:
        D = mol_out.size(1)
        if (torch.is_grad_enabled() and mol_out.requires_grad and pro_out.requires_grad and mol_out.is_cuda
                and mol_out.dtype == torch.float32 and pro_out.dtype == torch.float32 and mol_out.size(1) == pro_out.size(1)
                and _lib.load().glam_pair_pool_add_supported(D) == 1):
            return _PairPool.apply(mol_out, pro_out, msp, psp, True)
        return _PairPool.apply(mol_out, pro_out, msp, psp), mol_out, pro_out
    return _PairPool.apply(mol_out, pro_out, msp, psp)


class _PairPool5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mol, pro, msp, psp):
        require_device(mol, pro)
        mol, pro = f32c(mol, "mol_out"), f32c(pro, "pro_out")
        if msp.B != psp.B or mol.size(1) != pro.size(1) or mol.size(0) != msp.N or pro.size(0) != psp.N:
            raise GlamHipError("pair_pool5: the two batches disagree (pair count / width / node count)")
        P, D = msp.B, mol.size(1)
        out = torch.empty(P, 5, dtype=torch.float32, device=mol.device)
        arg = torch.empty(P, 6, dtype=torch.int32, device=mol.device)
        check(_lib.load().glam_pair_pool5_fwd(ptr(mol), ptr(pro), ptr(msp.ptr), ptr(psp.ptr), P, D, ptr(out), ptr(arg), stream()),
              "glam_pair_pool5_fwd")
        ctx.save_for_backward(mol, pro, out, arg)
        ctx.sps = (msp, psp)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        mol, pro, out, arg = ctx.saved_tensors
        msp, psp = ctx.sps
        d_out = f32c(d_out, "d_out")
        d_mol, d_pro = torch.empty_like(mol), torch.empty_like(pro)
        check(_lib.load().glam_pair_pool5_bwd(ptr(mol), ptr(pro), ptr(msp.ptr), ptr(psp.ptr), ptr(out), ptr(arg), ptr(d_out), msp.B,
                                              mol.size(1), ptr(d_mol), ptr(d_pro), stream()), "glam_pair_pool5_bwd")
        return d_mol, d_pro, None, None


def pair_pool5(mol_out, pro_out, msp, psp):
    """``[max, mean, median, min, std]`` of ``mol[seg_i] @ pro[seg_i].T`` per pair -> ``[P, 5]`` (dot_and_global_pool5);
    widths that are multiples of 4 up to 128 (``pad_cols`` the operands first: zero columns do not change a score)."""
    return _PairPool5.apply(mol_out, pro_out, msp, psp)
