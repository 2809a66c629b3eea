"""Dense operators of the path: linears on the hand-written MFMA kernels (k_ts_gemm / k_wgrad / narrow heads), NNConv's relation
table, the GRU step of MessageBlock (src_1gp/layer.py:223-267).

Part of ``glam_amd.ops`` (every public name here is re-exported there: ``from glam_amd import ops; ops.linear(...)``).  Knobs, the weight
scope, the padded-column bookkeeping and the index staging live in ``glam_amd/ops.py`` and are read through ``_o`` at call time."""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import _lib
from . import ops as _o
from ._lib import GlamHipError, check, f32c, ptr, require_device, stream

# --------------------------------------------------------------------------------------
# dense linear on the fp32 matrix cores + GRU gate math (MessageBlock remainder)
# --------------------------------------------------------------------------------------
def linear_supported(K, M):
    """Shapes the tall-skinny MFMA kernels cover in both directions (gemm.hip: ts_variant 0 / 1) with room for the bias
    ones-column in the weight-gradient kernel."""
    Kp, Mp = (K + 3) // 4 * 4, (M + 3) // 4 * 4
    return (Kp <= 60 and Mp <= 192) or (Kp <= 188 and Mp <= 64)


class _Linear(torch.autograd.Function):
    """y[N,M] = x[N,K] @ w[M,K]^T + b on k_ts_gemm; d_x on k_ts_gemm, d_w / d_b on k_wgrad (K, M multiples of 4).
    ``w`` may have FEWER columns than ``x`` (``w[M, Kw]``, ``Kw <= K``): ``x`` is then a zero-padded data matrix (atom features
    15 -> 16) and the weight image is built straight from the unpadded parameter (the image zero-fills k >= Kw); only for an
    ``x`` that needs no gradient."""

    @staticmethod
    def forward(ctx, x, w, b, relu=False, rrelu=None, node=None):
        """``relu``: ``max(y, 0)`` in the product's epilogue (glam_ts_gemm_relu; the caller checked glam_ts_gemm_relu_supported); the
        backward masks ``dy`` by the saved output first.  ``rrelu = (lower, upper, drop_p)``: the training-mode RReLU in the epilogue
        (glam_ts_gemm_rrelu) and, ``drop_p > 0``, the dropped twin as a second output — returns ``(y, y_drop)``; the backward regenerates
        the slopes / the mask from the recorded stream position (glam_bias_res_act_rng_bwd).  ``node = (staged, H * Cp, took)``: the rows
        feed a TripletMessage — the launch also writes its node product (glam_ts_gemm_act_node; ``took`` receives ``xw`` / ``a_ij``)."""
        require_device(x, w, b)
        x, w = f32c(x, "x"), f32c(w, "weight")
        b = None if b is None else f32c(b, "bias")
        N, K = x.shape
        M, Kw = w.shape
        if Kw > K or (Kw < K and ctx.needs_input_grad[0]):
            raise GlamHipError("linear: weight wider than the input / narrow weight with a differentiable input")
        lib, dev = _lib.load(), x.device
        scope = _o._SCOPE

        def build():
            img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, dtype=torch.float32, device=dev)
            check(lib.glam_ts_gemm_make_image(ptr(w), Kw, 1, Kw, M, ptr(img), stream()), "glam_ts_gemm_make_image")
            return img

        img = _o._scoped(scope.fwd if scope else None, ("lin", id(w)), w, build)
        y = torch.empty(N, M, dtype=torch.float32, device=dev)
        ctx.rrelu = None
        nd = None
        if node is not None and N > 0:
            staged, cols, took = node
            nd = (staged[lib.glam_triplet_staged_node_fragments(cols // M, M, 4):], cols, torch.empty(N, cols, dtype=torch.float32, device=dev),
                  torch.empty(N, 8, dtype=torch.float32, device=dev))
            took["xw"], took["a_ij"] = nd[2], nd[3]
        if rrelu is not None:
            ctx.set_materialize_grads(False)
            lo, hi, p = (float(v) for v in rrelu)
            eff = torch.empty(2, dtype=torch.int64, device=dev)
            y_drop = torch.empty_like(y) if p > 0 else None
            if nd is not None:
                check(lib.glam_ts_gemm_act_node(ptr(x), K, K, ptr(img), ptr(b), M, N, 4, lo, hi, p, ptr(_o.rng_state(dev)), ptr(eff), ptr(y), ptr(y_drop),
                                                ptr(nd[0]), nd[1], ptr(nd[2]), ptr(nd[3]), stream()), "glam_ts_gemm_act_node")
            else:
                check(lib.glam_ts_gemm_rrelu(ptr(x), K, K, ptr(img), ptr(b), M, N, lo, hi, p, ptr(_o.rng_state(dev)), ptr(eff), ptr(y), ptr(y_drop),
                                             stream()), "glam_ts_gemm_rrelu")
            ctx.save_for_backward(x, w, y)
            ctx.rrelu, ctx.eff = (lo, hi, p), eff
            ctx.has_bias = b is not None
            ctx.scope = scope
            return y, y_drop
        if relu and nd is not None:
            check(lib.glam_ts_gemm_act_node(ptr(x), K, K, ptr(img), ptr(b), M, N, 1, 0.0, 0.0, 0.0, None, None, ptr(y), None, ptr(nd[0]), nd[1],
                                            ptr(nd[2]), ptr(nd[3]), stream()), "glam_ts_gemm_act_node")
            ctx.save_for_backward(x, w, y)
        elif relu:
            check(lib.glam_ts_gemm_relu(ptr(x), K, K, ptr(img), ptr(b), ptr(y), M, M, N, stream()), "glam_ts_gemm_relu")
            ctx.save_for_backward(x, w, y)
        else:
            check(lib.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), ptr(b), ptr(y), M, M, None, 0, 0, N, stream()), "glam_ts_gemm")
            ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.scope = scope
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, dy_drop=None):
        x, w = ctx.saved_tensors[:2]
        N, K = x.shape
        M = w.size(0)
        if ctx.rrelu is not None:
            # the activation's backward on the regenerated slopes / mask: d_pre = d_y * (y > 0 ? 1 : slope) + d_y_drop * mask / (1 - p) * ...
            y_act = ctx.saved_tensors[2]
            dy = None if dy is None else f32c(dy, "dy")
            dy_drop = None if dy_drop is None else f32c(dy_drop, "dy_drop")
            d_pre = torch.empty_like(y_act)
            lo, hi, p = ctx.rrelu
            check(_lib.load().glam_bias_res_act_rng_bwd(ptr(y_act), ptr(dy), ptr(dy_drop), N, M, _o.ACT_CODES["rrelu"], 0.0, lo, hi, p, ptr(ctx.eff),
                                                        ptr(d_pre), stream()), "glam_bias_res_act_rng_bwd")
            dy = d_pre
        dy = f32c(dy, "dy")
        y_relu = ctx.saved_tensors[2] if (len(ctx.saved_tensors) == 3 and ctx.rrelu is None) else None
        # the fused ReLU's backward dy * (y > 0): inside the weight-gradient product where that is dy's only consumer (the first linear
        # of the model: atom features need no gradient), an elementwise launch otherwise
        mask_in_product = y_relu is not None and _o.RELU_IN_WGRAD and not ctx.needs_input_grad[0] and K + 1 <= 64
        if y_relu is not None and not mask_in_product:
            dy = torch.ops.aten.threshold_backward(dy, y_relu, 0.0)
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        dx = None
        if ctx.needs_input_grad[0]:
            def build():
                img = torch.empty(lib.glam_ts_gemm_image_bytes(M, K) // 4, **f)
                check(lib.glam_ts_gemm_make_image(ptr(w), K, 0, M, K, ptr(img), stream()), "glam_ts_gemm_make_image")
                return img

            img = _o._scoped(ctx.scope.bwd if ctx.scope else None, ("lin", id(w)), w, build)
            dx = torch.empty(N, K, **f)
            check(lib.glam_ts_gemm(ptr(dy), M, M, None, 0, 0, ptr(img), None, ptr(dx), K, K, None, 0, 0, N, stream()), "glam_ts_gemm")
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        if K + 1 <= 64:
            # weight and bias gradients as separate contiguous tensors: autograd keeps them as they are (views of one [M, K + 1] buffer
            # cost a copy launch each when they become .grad)
            # (a weight narrower than the zero-padded input, 15 -> 16 columns: the reduction writes its real columns only)
            Kw = w.size(1)
            dw, db = torch.empty(M, Kw, **f), torch.empty(M, **f)
            if mask_in_product:
                check(lib.glam_wgrad_gemm_split_relu(ptr(dy), ptr(y_relu), M, M, ptr(x), Kw, K, ptr(dw), ptr(db), N, ptr(ws), ws.numel(), stream()),
                      "glam_wgrad_gemm_split_relu")
            else:
                check(lib.glam_wgrad_gemm_split(ptr(dy), M, M, ptr(x), Kw, K, ptr(dw), ptr(db), N, ptr(ws), ws.numel(), stream()),
                      "glam_wgrad_gemm_split")
            return dx, dw, (db if ctx.has_bias else None), None, None, None
        dwb = torch.empty(M + 1, K + 1, **f)          # [d_w | d_b] (+ a spare row / column for the ones trick)
        if M <= 64:   # out[k, m] = sum_n [x|1][n,k] dy[n,m]  ->  written transposed into dwb[m, k]
            check(lib.glam_wgrad_gemm(ptr(x), K, K, None, 0, 0, 1, ptr(dy), M, M, 0, N, ptr(dwb), 1, K + 1, ptr(ws), ws.numel(),
                                      stream()), "glam_wgrad_gemm")
        else:         # out[m, k] = sum_n dy[n,m] [x|1][n,k]
            check(lib.glam_wgrad_gemm(ptr(dy), M, M, None, 0, 0, 0, ptr(x), K, K, 1, N, ptr(dwb), K + 1, 1, ptr(ws), ws.numel(),
                                      stream()), "glam_wgrad_gemm")
        dw = dwb[:M, :w.size(1)]
        db = dwb[:M, K] if ctx.has_bias else None
        return dx, dw, db, None, None, None


class _RelationMLP(torch.autograd.Function):
    """``nn(eye(De))`` for ``nn = Linear(De, hidden) -> ReLU -> Linear(hidden, M)``: the relation-weight table of NNConv with one-hot
    bond features (src_1gp/layer.py:115-122) — one launch forward, two backward, instead of a dozen library launches on 4-row
    operands (csrc/relmlp.hip)."""

    @staticmethod
    def forward(ctx, w1, b1, w2, b2):
        require_device(w1, b1, w2, b2)
        w1, b1, w2, b2 = f32c(w1, "w1"), f32c(b1, "b1"), f32c(w2, "w2"), f32c(b2, "b2")
        Hd, De = w1.shape
        M = w2.size(0)
        lib = _lib.load()
        h = torch.empty(De, Hd, dtype=torch.float32, device=w1.device)
        out = torch.empty(De, M, dtype=torch.float32, device=w1.device)
        check(lib.glam_relation_mlp_fwd(ptr(w1), ptr(b1), ptr(w2), ptr(b2), De, Hd, M, ptr(h), ptr(out), stream()), "glam_relation_mlp_fwd")
        ctx.save_for_backward(h, w2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        h, w2 = ctx.saved_tensors
        De, Hd = h.shape
        M = w2.size(0)
        lib, dev = _lib.load(), h.device
        d_out = f32c(d_out, "d_out")
        f = dict(dtype=torch.float32, device=dev)
        d_w1, d_b1, d_w2, d_b2 = torch.empty(Hd, De, **f), torch.empty(Hd, **f), torch.empty(M, Hd, **f), torch.empty(M, **f)
        ws = torch.empty(lib.glam_relation_mlp_workspace_bytes(De, Hd, M), dtype=torch.uint8, device=dev)
        check(lib.glam_relation_mlp_bwd(ptr(d_out), ptr(h), ptr(w2), De, Hd, M, ptr(d_w1), ptr(d_b1), ptr(d_w2), ptr(d_b2), ptr(ws),
                                        ws.numel(), stream()), "glam_relation_mlp_bwd")
        return d_w1, d_b1, d_w2, d_b2


def relation_mlp(nn, De):
    """``nn(eye(De))`` — on the HIP kernels when ``nn`` is the reference's ``Sequential(Linear(De, hidden), ReLU(), Linear(hidden, M))``
    (fp32, on the device, a shape ``glam_relation_mlp_supported`` accepts: De <= 8, hidden a power of two up to 64), through torch
    otherwise."""
    mods = list(nn.children()) if isinstance(nn, torch.nn.Sequential) else []
    if (len(mods) == 3 and isinstance(mods[0], torch.nn.Linear) and isinstance(mods[1], torch.nn.ReLU) and isinstance(mods[2], torch.nn.Linear)
            and mods[0].bias is not None and mods[2].bias is not None and mods[0].in_features == De
            and mods[0].weight.is_cuda and mods[0].weight.dtype == torch.float32 and mods[2].weight.dtype == torch.float32
            and _lib.load().glam_relation_mlp_supported(De, mods[0].out_features, mods[2].out_features)):
        return _RelationMLP.apply(mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias)
    p = next(nn.parameters())
    return nn(torch.eye(De, dtype=p.dtype, device=p.device))


class _MatmulTall(torch.autograd.Function):
    """``A[N,K] @ W[K,M] (+ bias)`` for tall A and a small weight whose shape is outside the MFMA forward table: the two
    data-side products stay on the library GEMM, but the WEIGHT gradient ``A^T @ dY`` — a reduction over the N rows for
    which the library's heuristics pick 32x32 tiles (77 us at N = 20 k, K = 240, M = 60) — runs on ``k_wgrad`` (≈12 us),
    the bias gradient riding on its ones column.  ``carry``: the gradient carry of (w, bias) when a block applies them several
    times per forward (see _ParamBundle): [d_w | d_bias] flat, summed by the reduction of the weight-gradient product."""

    @staticmethod
    def forward(ctx, a, w, bias, carry=None, first_app=True, with_identity=False):
        """``with_identity``: ``a`` comes back as a second output — the skip connection of the caller (MessageBlock around a GCNConv,
        layer.py:253-265) takes it from there, so that its gradient arrives HERE and joins ``dy @ w^T`` in that product's epilogue
        instead of in an add launch of the autograd engine."""
        require_device(a, w, bias)
        a_in = a
        a, w = f32c(a, "a"), f32c(w, "w")
        ctx.save_for_backward(a, w)
        ctx.has_bias = bias is not None
        ctx.scope = _o._SCOPE
        ctx.carried = carry is not None
        ctx.aliased = bool(with_identity)
        ctx.first_app = bool(first_app)
        if ctx.carried or ctx.aliased:
            ctx.set_materialize_grads(False)     # the carry of the LAST application has no gradient yet: None, not a zero fill
        N, K = a.shape
        M = w.size(1)
        if K % 4 == 0 and M % 4 == 0 and K <= 320 and M <= 64 and N > 0 and _lib.route_enabled("x3"):
            # NNConv's [N, 300] x [300, 60] relation product (and any K <= 320 x M <= 64): the long-reduction 3 x bf16 kernel (tall_x3.hip)
            # (an 80 KB-image fp32 k_ts_gemm<4, 20, 4> measured 18.9 us against the library's 15 at N = 20 k and was not kept)
            lib = _lib.load()
            scope = ctx.scope
            img = _o._scoped(scope.fwd if scope else None, ("tall-fwd", id(w)), w, lambda: _o._ts_image(w, K, M, False))
            out = torch.empty(N, M, dtype=torch.float32, device=a.device)
            check(lib.glam_ts_gemm(ptr(a), K, K, None, 0, 0, ptr(img), ptr(f32c(bias, "bias")) if bias is not None else None, ptr(out), M, M,
                                   None, 0, 0, N, stream()), "glam_ts_gemm")
        else:
            out = torch.matmul(a, w) if bias is None else torch.addmm(f32c(bias, "bias"), a, w)
        res = (out,) + ((a_in.view_as(a_in),) if ctx.aliased else ()) + ((carry.view(-1),) if ctx.carried else ())
        return res if len(res) > 1 else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *more):
        d_alias = more[0] if ctx.aliased else None
        d_carry = more[-1] if ctx.carried else None
        res = _MatmulTall._backward(ctx, dy, d_alias, d_carry)
        return res + (None,)                              # (with_identity)

    @staticmethod
    def _backward(ctx, dy, d_alias, d_carry):
        a, w = ctx.saved_tensors
        N, K = a.shape
        M = w.size(1)
        if dy is None:      # only with a carry / an alias (grads are not materialised then): the output itself was unused
            return d_alias, None, None, d_carry, None
        dy = f32c(dy, "dy")
        da = None
        if ctx.needs_input_grad[0]:
            if M <= 64 and K <= 64 and M % 4 == 0 and K % 4 == 0 and linear_supported(M, K) and N > 0:
                # a layer-sized product (GCNConv 60 -> 60): k_ts_gemm, the skip connection's gradient added in its epilogue
                lib = _lib.load()
                scope = ctx.scope
                img = _o._scoped(scope.bwd if scope else None, ("tall-dx", id(w)), w, lambda: _o._ts_image(w, M, K, True))
                da = torch.empty(N, K, dtype=torch.float32, device=a.device)
                if d_alias is not None:
                    d_alias = f32c(d_alias, "d_identity")
                    check(lib.glam_ts_gemm_add(ptr(dy), M, M, ptr(img), None, ptr(da), K, K, ptr(d_alias), K, N, stream()), "glam_ts_gemm_add")
                    d_alias = None
                else:
                    check(lib.glam_ts_gemm(ptr(dy), M, M, None, 0, 0, ptr(img), None, ptr(da), K, K, None, 0, 0, N, stream()), "glam_ts_gemm")
            elif M <= 96 and K <= 320 and K > 64:
                # dy[N, M] @ w^T[M, K] with a wide output: the 120 KB-image k_ts_gemm variant (the library GEMM picks 16x256
                # tiles for this shape: 44 us for 60 -> 300 at N = 20 k)
                lib = _lib.load()
                scope = ctx.scope
                img = _o._scoped(scope.bwd if scope else None, ("tall-dx", id(w)), w, lambda: _o._ts_image(w, M, K, True))
                da = torch.empty(N, K, dtype=torch.float32, device=a.device)
                check(lib.glam_ts_gemm(ptr(dy), M, M, None, 0, 0, ptr(img), None, ptr(da), K, K, None, 0, 0, N, stream()), "glam_ts_gemm")
            else:
                da = torch.matmul(dy, w.t())
            if d_alias is not None:
                da = da + d_alias
        elif d_alias is not None:
            da = d_alias
        dw = db = None
        scope = ctx.scope
        if ctx.carried and ctx.has_bias and M <= 64 and scope is not None and _o.GRU_WGRAD_BATCH and N >= _GRU_BATCH_MIN_ROWS:
            # the weight gradient of ALL applications of the block in one launch (see _GruBlock.backward): every application but the first
            # parks (a, dy); the first one — its backward runs last — multiplies the parked sets together, three per launch
            lib = _lib.load()
            parked = scope.bwd.setdefault(("tall-parked", id(w)), (w, []))[1]
            parked.append((a, dy))
            if not ctx.first_app:
                return da, None, None, d_carry, None
            sets = list(parked)
            parked.clear()
            dwb = torch.empty(K + 1, M, dtype=torch.float32, device=a.device)
            add = None if d_carry is None else f32c(d_carry, "d_carry")
            vp = ctypes.c_void_p
            while sets:
                grp, sets = sets[:3], sets[3:]
                n = len(grp)
                ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=a.device)
                check(lib.glam_wgrad_gemm_sets(n, (vp * n)(*[t[0].data_ptr() for t in grp]), K, K, 1, (vp * n)(*[t[1].data_ptr() for t in grp]),
                                               M, M, N, ptr(dwb), M, 1, ptr(add), ptr(ws), ws.numel(), stream()), "glam_wgrad_gemm_sets")
                add = dwb
            return da, None, None, dwb.view(-1), None
        if ctx.needs_input_grad[1] or ctx.has_bias or ctx.carried:
            lib = _lib.load()
            ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=a.device)
            add = f32c(d_carry, "d_carry") if (ctx.carried and d_carry is not None and N > 0) else None

            def product(*args):     # (P, I, ldp, ones, Q, J, ldq, out, si, sj): the carry, laid out like `out`, joins in the reduction
                P, I, ldp, ones, Q, J, ldq, out, si, sj = args
                if add is None:
                    check(lib.glam_wgrad_gemm(ptr(P), I, ldp, None, 0, 0, ones, ptr(Q), J, ldq, 0, N, ptr(out), si, sj, ptr(ws), ws.numel(),
                                              stream()), "glam_wgrad_gemm")
                else:
                    check(lib.glam_wgrad_gemm_add(ptr(P), I, ldp, None, 0, 0, ones, ptr(Q), J, ldq, 0, N, ptr(out), si, sj, ptr(add),
                                                  ptr(ws), ws.numel(), stream()), "glam_wgrad_gemm_add")

            if ctx.has_bias:   # [dw ; db] = [a | 1]^T dy   (K + 1 <= 320, M <= 128: matmul_tall's bias condition)
                dwb = torch.empty(K + 1, M, dtype=torch.float32, device=a.device)
                product(a, K, K, 1, dy, M, M, dwb, M, 1)
                if ctx.carried:
                    flat = dwb.view(-1)
                    return da, None, None, (flat if (add is not None or d_carry is None) else flat.add_(d_carry)), None
                return da, dwb[:K], dwb[K], None, None
            dw = torch.empty(K, M, dtype=torch.float32, device=a.device)
            if M <= 128:      # dw = a^T dy: P = a (up to 320 columns), Q = dy (two 64-column chunks beyond 64)
                product(a, K, K, 0, dy, M, M, dw, M, 1)
            else:             # wide output: dw^T = dy^T a, written through transposed strides
                product(dy, M, M, 0, a, K, K, dw, 1, M)
            if ctx.carried:
                flat = dw.view(-1)
                return da, None, None, (flat if (add is not None or d_carry is None) else flat.add_(d_carry)), None
        return da, dw, db, None, None


def _matmul_tall_node(a, w, bias, with_identity=False):
    """``_MatmulTall`` with the gradients of (w, bias) carried across the applications of a block inside a weight_scope."""
    K, M = w.shape
    total = K * M + (M if bias is not None else 0)
    params = (w,) if bias is None else (w, bias)

    def split(flat):     # [d_w (K x M) | d_bias (M)]: the layout of the [a | 1]^T dy product
        return (flat[:K * M].view(K, M),) if bias is None else (flat[:K * M].view(K, M), flat[K * M:])
    key = ("carry-tall", id(w), id(bias))
    hit = _o._SCOPE.fwd.get(key) if _o._SCOPE is not None else None
    first = not (hit is not None and hit[0] is w)          # the block's first application of this pass: its backward runs LAST
    carry = _o._carry_for(key, params, total, split) if (w.requires_grad or (bias is not None and bias.requires_grad)) else None
    if carry is None:
        return _MatmulTall.apply(a, w, bias, None, True, with_identity)
    res = _MatmulTall.apply(a, w, bias, carry, first, with_identity)
    _o._carry_store(key, w, res[-1])
    return (res[0], res[1]) if with_identity else res[0]


def matmul_tall(a, w, bias=None, with_identity=False):
    """``a @ w (+ bias)`` with the weight gradient on the MFMA reduction kernel when it fits: one of (K, M) <= 320 and the other
    <= 128, multiples of 4 (with a bias: K + 1 <= 320 and M <= 128).  ``with_identity``: returns ``(out, identity)`` with ``identity`` =
    ``a`` handed back through the product's autograd node where that saves the add launch of a skip connection around it (plain ``a``
    elsewhere)."""
    K, M = w.shape
    ok = a.dim() == 2 and a.is_cuda and K % 4 == 0 and M % 4 == 0 and a.size(0) >= 64
    alias = bool(with_identity) and ok and torch.is_grad_enabled() and a.requires_grad and a.dtype == torch.float32
    def fin(out, ident=None):
        return (out, a if ident is None else ident) if with_identity else out
    if ok and bias is not None and K + 1 <= 320 and M <= 128:
        r = _matmul_tall_node(a, w, bias, alias)
        return fin(*r) if alias else fin(r)
    if ok and ((K <= 320 and M <= 128) or (K <= 128 and M <= 320)):
        r = _matmul_tall_node(a, w, None, alias)
        out, ident = r if alias else (r, None)
        return fin(out if bias is None else out + bias, ident)
    out = torch.matmul(a, w)
    return fin(out if bias is None else out + bias)


class _LinearTall(torch.autograd.Function):
    """``y = act(x) @ w^T + b`` for layer widths beyond the MFMA forward table (e.g. the GRU gate linears 92 -> 276 of
    hid_dim_alpha = 6): both data-side products on the warp-specialised 3 x bf16 kernels (tall_x3.hip), ``d_w = dy^T act(x)`` and ``d_b``
    on ``k_wgrad`` as two contiguous tensors (``glam_wgrad_gemm_linear``: no strided views for autograd to copy).  ``celu_in``: act = the
    CELU MessageBlock applies in front of its GRU (src_1gp/layer.py:261), folded into the operand loads of all three products instead
    of a launch each way.  ``carry``: the gradient carry of (w, b) when a block applies them several times per forward (see
    _ParamBundle): ``[d_w | d_b]`` flat, summed by the reduction of the weight-gradient product."""

    @staticmethod
    def forward(ctx, x, w, b, carry=None, celu_in=False, first_app=True):
        require_device(x, w, b)
        x, w, b = f32c(x, "x"), f32c(w, "weight"), f32c(b, "bias")
        ctx.scope = _o._SCOPE
        ctx.carried = carry is not None
        ctx.first_app = bool(first_app)
        if ctx.carried:
            ctx.set_materialize_grads(False)     # the carry of the LAST application has no gradient yet: None, not a zero fill
        N, K = x.shape
        M = w.size(0)
        ctx.fold = bool(celu_in) and K <= 96 and M <= 288 and K + 1 <= 128 and N > 0      # every product has a CELU-aware kernel
        if celu_in and not ctx.fold:
            x = torch.celu(x)
        ctx.save_for_backward(x, w)
        if K <= 96 and M <= 320:       # 92 -> 276: k_tall_x3<3, 4, 5> (15 us at N = 20.4 k; the library 29, the fp32 LDS-image kernel 24)
            lib = _lib.load()
            img = _o._scoped(_o._SCOPE.fwd if _o._SCOPE else None, ("lin", id(w)), w, lambda: _o._ts_image(w, K, M, True))
            y = torch.empty(N, M, dtype=torch.float32, device=x.device)
            if ctx.fold:
                check(lib.glam_ts_gemm_celu(ptr(x), K, K, 1, ptr(img), ptr(b), ptr(y), M, M, None, 0, N, stream()), "glam_ts_gemm_celu")
            else:
                check(lib.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), ptr(b), ptr(y), M, M, None, 0, 0, N, stream()), "glam_ts_gemm")
        else:
            y = torch.addmm(b, x, w.t())
        return (y, carry.view(-1)) if ctx.carried else y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, d_carry=None):
        if dy is None:      # only with a carry (grads are not materialised then): the output itself was unused
            return None, None, None, d_carry, None, None
        x, w = ctx.saved_tensors
        dy = f32c(dy, "dy")
        N, K = x.shape
        M = w.size(0)
        lib = _lib.load()
        f = dict(dtype=torch.float32, device=x.device)
        dx = None
        if ctx.needs_input_grad[0] and M <= 288 and K <= 96 and N > 0:      # dy[N, M] @ w[M, K], long reduction: tall_x3.hip
            scope = ctx.scope
            img = _o._scoped(scope.bwd if scope else None, ("lin-t", id(w)), w, lambda: _o._ts_image(w, M, K, False))
            dx = torch.empty(N, K, **f)
            if ctx.fold:        # ... * celu'(x) in the epilogue
                check(lib.glam_ts_gemm_celu(ptr(dy), M, M, 0, ptr(img), None, ptr(dx), K, K, ptr(x), K, N, stream()), "glam_ts_gemm_celu")
            else:
                check(lib.glam_ts_gemm(ptr(dy), M, M, None, 0, 0, ptr(img), None, ptr(dx), K, K, None, 0, 0, N, stream()), "glam_ts_gemm")
        elif ctx.needs_input_grad[0]:
            dx = torch.matmul(dy, w)
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=x.device)
        scope = ctx.scope
        if (ctx.carried and scope is not None and _o.GRU_WGRAD_BATCH and N >= _GRU_BATCH_MIN_ROWS and K % 4 == 0 and K + 1 <= 128 and M <= 320):
            # the weight gradient of ALL applications of the block in one launch (see _GruBlock.backward): every application but the first
            # parks (dy, x); the first one — its backward runs last — multiplies the parked sets together, three per launch
            parked = scope.bwd.setdefault(("lintall-parked", id(w)), (w, []))[1]
            parked.append((dy, x, bool(ctx.fold)))
            if not ctx.first_app:
                return dx, None, None, d_carry, None, None
            sets = list(parked)
            parked.clear()
            addf = None if d_carry is None else f32c(d_carry, "d_carry")
            vp = ctypes.c_void_p
            while sets:
                grp = [t for t in sets if t[2] == sets[0][2]][:3]
                sets = [t for t in sets if all(t is not u for u in grp)]
                n = len(grp)
                flat = torch.empty(M * (K + 1), **f)
                aw, ab = (addf[:M * K], addf[M * K:]) if addf is not None else (None, None)
                ws2 = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=x.device)
                check(lib.glam_wgrad_gemm_linear_sets(n, (vp * n)(*[t[0].data_ptr() for t in grp]), M, M, (vp * n)(*[t[1].data_ptr() for t in grp]),
                                                      K, K, int(grp[0][2]), ptr(flat[:M * K]), ptr(flat[M * K:]), ptr(aw), ptr(ab), N, ptr(ws2),
                                                      ws2.numel(), stream()), "glam_wgrad_gemm_linear_sets")
                addf = flat
            return dx, None, None, addf, None, None
        add = f32c(d_carry, "d_carry") if (ctx.carried and d_carry is not None and N > 0) else None
        if K % 4 == 0 and K + 1 <= 128 and M <= 320 and N > 0:
            # weight and bias gradients as two contiguous pieces of one buffer [d_w (M x K) | d_b (M)]
            flat = torch.empty(M * (K + 1), **f)
            dw, db = flat[:M * K].view(M, K), flat[M * K:]
            aw, ab = (add[:M * K], add[M * K:]) if add is not None else (None, None)
            check(lib.glam_wgrad_gemm_linear(ptr(dy), M, M, ptr(x), K, K, int(ctx.fold), ptr(dw), ptr(db), ptr(aw), ptr(ab), N, ptr(ws),
                                             ws.numel(), stream()), "glam_wgrad_gemm_linear")
            if ctx.carried:
                return dx, None, None, (flat if (add is not None or d_carry is None) else flat.add_(d_carry)), None, None
            return dx, dw, db, None, None, None
        dwb = torch.empty(M, K + 1, **f)
        check(lib.glam_wgrad_gemm(ptr(dy), M, M, None, 0, 0, 0, ptr(x), K, K, 1, N, ptr(dwb), K + 1, 1, ptr(ws), ws.numel(), stream()),
              "glam_wgrad_gemm")
        if ctx.carried:      # (layout of the carry: [d_w | d_b])
            flat = torch.cat([dwb[:, :K].reshape(-1), dwb[:, K]])
            return dx, None, None, (flat if d_carry is None else flat.add_(d_carry)), None, None
        return dx, dwb[:, :K], dwb[:, K], None, None, None


def _linear_tall_node(x, w, b, celu_in=False):
    """``_LinearTall`` with the gradients of (w, b) carried across the applications of a block inside a weight_scope."""
    M, K = w.shape

    def split(flat):     # [d_w (M x K) | d_b (M)]: two contiguous pieces
        return flat[:M * K].view(M, K), flat[M * K:]
    key = ("carry-lintall", id(w), id(b))
    hit = _o._SCOPE.fwd.get(key) if _o._SCOPE is not None else None
    first = not (hit is not None and hit[0] is w)          # the block's first application of this pass: its backward runs LAST
    carry = _o._carry_for(key, (w, b), M * (K + 1), split) if (w.requires_grad or b.requires_grad) else None
    if carry is None:
        return _LinearTall.apply(x, w, b, None, celu_in)
    y, carry = _LinearTall.apply(x, w, b, carry, celu_in, first)
    _o._carry_store(key, w, carry)
    return y


def linear_tall_supported(K, M):
    return K % 4 == 0 and M % 4 == 0 and M <= 320 and K + 1 <= 128


class _LinearNarrow(torch.autograd.Function):
    """``y[N, M] = x[N, K] @ w[M, K]^T + b`` for a handful of outputs (M <= 16: the model's output head, out_dim 1 / 2 / 12): row dot
    products on ``glam_linear_narrow_fwd``; backward ``d_x``, ``d_w``, ``d_b`` in one pass over ``x`` + a fixed-order reduction
    (``glam_linear_narrow_bwd``) — the GEMM library took 33 + 22 us for 1024 x 1024 -> 1, this takes a few us each way."""

    @staticmethod
    def forward(ctx, x, w, b):
        require_device(x, w, b)
        x, w = f32c(x, "x"), f32c(w, "weight")
        b = None if b is None else f32c(b, "bias")
        N, K = x.shape
        M = w.size(0)
        y = torch.empty(N, M, dtype=torch.float32, device=x.device)
        check(_lib.load().glam_linear_narrow_fwd(ptr(x), ptr(w), ptr(b), N, K, M, ptr(y), stream()), "glam_linear_narrow_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        x, w = ctx.saved_tensors
        dy = f32c(dy, "dy")
        N, K = x.shape
        M = w.size(0)
        lib = _lib.load()
        f = dict(dtype=torch.float32, device=x.device)
        dx = torch.empty(N, K, **f) if ctx.needs_input_grad[0] else None
        dw = torch.empty(M, K, **f)
        db = torch.empty(M, **f) if ctx.has_bias else None
        ws = torch.empty(lib.glam_linear_narrow_bwd_workspace_bytes(K, M), dtype=torch.uint8, device=x.device)
        check(lib.glam_linear_narrow_bwd(ptr(x), ptr(w), ptr(dy), N, K, M, ptr(dx), ptr(dw), ptr(db), ptr(ws), ws.numel(), stream()),
              "glam_linear_narrow_bwd")
        return dx, dw, db


class _LinearNarrowAct(torch.autograd.Function):
    """``y = Linear(Dropout(p)(RReLU(lower, upper)(x)))`` in training mode with ``x`` the PRE-activation of the layer in front of the head
    (``mol_flat`` -> ``lin_out1``, src_1gp/model.py:60-61): the row dot products apply both to every element they read
    (``glam_linear_narrow_act_fwd`` / ``_bwd``); neither the activated matrix nor the dropped twin exists."""

    @staticmethod
    def forward(ctx, x, w, b, lower, upper, p):
        require_device(x, w, b)
        x, w = f32c(x, "x"), f32c(w, "weight")
        b = None if b is None else f32c(b, "bias")
        N, K = x.shape
        M = w.size(0)
        y = torch.empty(N, M, dtype=torch.float32, device=x.device)
        eff = torch.empty(2, dtype=torch.int64, device=x.device)
        check(_lib.load().glam_linear_narrow_act_fwd(ptr(x), ptr(w), ptr(b), N, K, M, float(lower), float(upper), float(p),
                                                     ptr(_o.rng_state(x.device)), ptr(eff), ptr(y), stream()), "glam_linear_narrow_act_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.eff, ctx.cfg = b is not None, eff, (float(lower), float(upper), float(p))
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None, None, None
        x, w = ctx.saved_tensors
        dy = f32c(dy, "dy")
        N, K = x.shape
        M = w.size(0)
        lib = _lib.load()
        f = dict(dtype=torch.float32, device=x.device)
        dx = torch.empty(N, K, **f) if ctx.needs_input_grad[0] else None
        dw = torch.empty(M, K, **f)
        db = torch.empty(M, **f) if ctx.has_bias else None
        ws = torch.empty(lib.glam_linear_narrow_bwd_workspace_bytes(K, M), dtype=torch.uint8, device=x.device)
        lo, hi, p = ctx.cfg
        check(lib.glam_linear_narrow_act_bwd(ptr(x), ptr(w), ptr(dy), N, K, M, lo, hi, p, ptr(ctx.eff), ptr(dx), ptr(dw), ptr(db), ptr(ws),
                                             ws.numel(), stream()), "glam_linear_narrow_act_bwd")
        return dx, dw, db, None, None, None


def rrelu_dropout_linear_narrow(x, weight, bias, lower, upper, p):
    """Training-mode ``F.linear(F.dropout(rrelu(x, lower, upper), p), weight, bias)`` for a head of at most 16 outputs, one launch each
    way (``x``: the pre-activation); ``None`` where the shape is outside the row-dot-product kernels (the caller runs the three steps)."""
    M, K = weight.shape
    f32 = x.dtype == torch.float32 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    if not (_o.HEAD_ACT_FUSED and x.dim() == 2 and x.is_cuda and f32 and M <= 16 and K >= 64 and K % 4 == 0 and x.size(1) == K
            and 0 < lower <= upper and 0 <= p < 1 and _o.padded_base(x) is None):
        return None
    return _LinearNarrowAct.apply(x, weight, bias, lower, upper, p)


class _LinearLib(torch.autograd.Function):
    """``F.linear`` with the matrix products on the GEMM library (layers outside the MFMA kernels' table: the 300 -> 1024 readout MLP)
    and the bias gradient on ``glam_colsum`` (torch's generic column reduction takes 14 us for [1024, 1024]; this takes a few)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.set_materialize_grads(False)
        return torch.addmm(b, x, w.t())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        x, w = ctx.saved_tensors
        dy = f32c(dy, "dy")
        dx = torch.mm(dy, w) if ctx.needs_input_grad[0] else None
        dw = torch.mm(dy.t(), x) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2] and dy.data_ptr() % 16:
            db = dy.sum(0)                   # a contiguous view at a storage offset that is not 16-byte aligned: the kernel loads float4
        elif ctx.needs_input_grad[2]:
            lib = _lib.load()
            N, D = dy.shape
            db = torch.empty(D, dtype=torch.float32, device=dy.device)
            ws = torch.empty(lib.glam_colsum_workspace_bytes(D), dtype=torch.uint8, device=dy.device)   # (touched for N > 2048 only)
            check(lib.glam_colsum(ptr(dy), N, D, D, ptr(db), ptr(ws), ws.numel(), stream()), "glam_colsum")
        return dx, dw, db


def linear_dense_supported(N, K, M):
    """The class of ``csrc/dense_x3.hip`` that beats the GEMM library inside a model step: the readout MLP 5 * hid_dim -> 1024 with rows
    that are multiples of 16 bytes (hid_dim 60: 300 columns).  The kernel takes any shape, but with partial quads (K = 75, 150, 225,
    450: the other hidden widths) or scalar output stores (the 617-task head) the model step measured 1-17 us SLOWER than on the
    library (``tools/bench_model.py --alpha``, A/B ``ops.DENSE_LINEAR``), so those stay there."""
    return K % 4 == 0 and M % 4 == 0 and K >= 32 and N >= 4


_DENSE_WS = {}


def _dense_ws(dev):
    """The split-k workspace of ``glam_linear_dense_fwd_ws`` / ``_bwd_ws`` (``ops.DENSE_SPLITK``; None: switched off): one per device;
    the launches that share it are ordered on the process's one compute stream."""
    if not _o.DENSE_SPLITK:
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ws = _DENSE_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None      # (born inside a capture it would live in that graph's private pool and outlive it here: this launch goes unsplit)
        ws = _DENSE_WS[key] = torch.empty(_lib.load().glam_dense_ws_bytes(), dtype=torch.uint8, device=dev)
    return ws


class _LinearDense(torch.autograd.Function):
    """``act(F.linear(x, w, b))`` with act in {none, ReLU, LeakyReLU} on ``glam_linear_dense_fwd`` (3 x bf16 matrix cores, bias and
    activation in the epilogue) and the whole backward — activation derivative on ``dy`` (read from the saved OUTPUT), ``dx``, ``dw``
    and the bias gradient — in one ``glam_linear_dense_bwd`` launch: the products of the readout MLP (src_1gp/model.py:43-45, 60) that
    ran on the GEMM library with an activation, a mask and a column-sum launch around them."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        require_device(x, w) if b is None else require_device(x, w, b)
        x, w = f32c(x, "x"), f32c(w, "weight")
        b = None if b is None else f32c(b, "bias")
        N, K = x.shape
        M = w.size(0)
        y = torch.empty(N, M, dtype=torch.float32, device=x.device)
        ws = _dense_ws(x.device)
        check(_lib.load().glam_linear_dense_fwd_ws(ptr(x), ptr(w), ptr(b), N, K, M, act, slope, ptr(y), ptr(ws), 0 if ws is None else ws.numel(),
                                                   stream()), "glam_linear_dense_fwd_ws")
        ctx.save_for_backward(x, w, y if act else None)
        ctx.has_bias, ctx.slope = b is not None, (0.0 if act == 1 else slope)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None, None
        x, w, y = ctx.saved_tensors
        dy = f32c(dy, "dy")
        N, K = x.shape
        M = w.size(0)
        f = dict(dtype=torch.float32, device=x.device)
        # only what autograd asks for (a frozen layer: no dy^T x product; the entry point takes NULL for each output)
        dx = torch.empty(N, K, **f) if ctx.needs_input_grad[0] else None
        dw = torch.empty(M, K, **f) if ctx.needs_input_grad[1] else None
        db = torch.empty(M, **f) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if dx is None and dw is None and db is None:
            return None, None, None, None, None
        need_dw = dw is not None
        if db is not None and dw is None:
            dw = torch.empty(M, K, **f)      # (the bias gradient is the all-ones column of the dy^T [x | 1] product: it comes with dw)
        ws = _dense_ws(x.device)
        check(_lib.load().glam_linear_dense_bwd_ws(ptr(x), ptr(w), ptr(dy), ptr(y), ctx.slope, N, K, M, ptr(dx), ptr(dw), ptr(db), ptr(ws),
                                                   0 if ws is None else ws.numel(), stream()), "glam_linear_dense_bwd_ws")
        return dx, (dw if need_dw else None), db, None, None


_DENSE_ACT = {"none": 0, "relu": 1, "leaky": 2}


def _dense_route(x, weight, bias):
    M, K = weight.shape
    f32 = x.dtype == torch.float32 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    return (_o.DENSE_LINEAR and x.dim() == 2 and x.is_cuda and f32 and not linear_supported(K, M) and M > 16 and linear_dense_supported(x.size(0), K, M)
            and _o.padded_base(x) is None)


def linear_act(x, weight, bias, act="none", slope=0.0):
    """``act(F.linear(x, weight, bias))`` for a deterministic elementwise activation (``"none"``, ``"relu"``, ``"leaky"``) in one launch
    each way where the shape is in the dense kernel's class; ``None`` elsewhere (the caller applies ``linear`` and its activation)."""
    if not _dense_route(x, weight, bias):
        return None
    return _LinearDense.apply(x, weight, bias, _DENSE_ACT[act], float(slope))


def linear(x, weight, bias=None):
    """``F.linear`` on the hand-written kernels when the shape is in their table (the layer-sized linears of the
    path: GRU gates 60->180, input embedding 15->60, ... on the MFMA kernels; heads with <= 16 outputs as row dot products);
    larger / odd layers (e.g. the 300->1024 readout MLP) stay on the library GEMM, which is the right tool for them."""
    M, K = weight.shape
    f32 = x.dtype == torch.float32 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    if x.dim() == 2 and x.is_cuda and f32 and M <= 16 and K >= 64 and K % 4 == 0 and not linear_supported(K, M):
        return _LinearNarrow.apply(x, weight, bias)      # (other dtypes — fp64, autocast — fall through to F.linear below)
    if _dense_route(x, weight, bias):
        return _LinearDense.apply(x, weight, bias, 0, 0.0)
    if x.dim() != 2 or not linear_supported(K, M):
        if x.dim() == 2 and x.is_cuda and bias is not None and M % 4 == 0 and x.dtype == torch.float32 and weight.dtype == torch.float32:
            return _LinearLib.apply(x, weight, bias)
        return torch.nn.functional.linear(x, weight, bias)
    Kp, Mp = (K + 3) // 4 * 4, (M + 3) // 4 * 4
    if Kp != K:
        x = _o.pad_cols(x, Kp)
    if Kp != K and Mp == M and not (x.requires_grad and torch.is_grad_enabled()):
        return _Linear.apply(x, weight, bias)    # data input (atom features): the image is built from the unpadded weight
    if Kp != K or Mp != M:
        # padded once per model pass (the block's linears are applied message_steps times), like every derived weight
        w0, b0 = weight, bias
        def build_padded():        # weight and bias from one launch
            items = [(w0, (1, M, K), (Mp, Kp), (Mp, Kp))] + ([] if b0 is None else [(b0, (1, 1, M), (1, Mp), (Mp,))])
            out = _o.pad_group(items)
            return out[0], (None if b0 is None else out[1])
        weight, bias = _o.scoped_weights(("lin-pad", id(w0), None if b0 is None else id(b0)), w0, build_padded)
    y = _Linear.apply(x, weight, bias)
    return _o.slice_cols(y, M)                    # pad columns are x @ 0 + 0


def linear_relu(x, weight, bias=None, node=None):
    """``relu(F.linear(x, weight, bias))`` with the ReLU in the epilogue of the product where its shape runs on ``k_tall_x3`` (the input
    embeddings of the models: 15 -> 60 ...); ``None`` elsewhere (the caller applies ``linear`` and the activation)."""
    M, K = weight.shape
    Kp = (K + 3) // 4 * 4
    f32 = x.dtype == torch.float32 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    if not (x.dim() == 2 and x.is_cuda and f32 and M % 4 == 0 and linear_supported(K, M) and not _dense_route(x, weight, bias)
            and _lib.load().glam_ts_gemm_relu_supported(Kp, M) == 1):
        return None
    if Kp != K:
        if x.requires_grad and torch.is_grad_enabled():
            return None                      # (a padded differentiable input takes the padded-weight route of ``linear``)
        x = _o.pad_cols(x, Kp)
    nd, took = _node_arg(node, M) if _lib.load().glam_ts_gemm_rrelu_supported(Kp, M) == 1 else (None, None)
    y = _Linear.apply(x, weight, bias, True, None, nd)
    if took:
        _o.register_node_product(y, node[0], took["xw"], took["a_ij"])
    return y


def _node_arg(node, M):
    """``node`` of ``linear_relu`` / ``linear_rrelu`` as ``_Linear`` takes it (+ the dict the forward fills), or None where the product's
    width is not the layer's."""
    if node is None or node[1] % M or not _o.NODE_IN_GRU:
        return None, None
    took = {}
    return (node[0], node[1], took), took


def linear_rrelu(x, weight, bias, lower, upper, drop_p=0.0, node=None):
    """Training-mode ``RReLU(lower, upper)(F.linear(x, weight, bias))`` with the activation — and, ``drop_p > 0``, the dropped twin the
    following ``Dropout(drop_p)`` picks up (``ops.take_dropped``) — in the epilogue of the product where its shape runs on ``k_tall_x3``
    with that epilogue (the input embeddings: 15 -> 60 ...); ``None`` elsewhere (the caller applies ``linear`` and ``rrelu``)."""
    M, K = weight.shape
    Kp = (K + 3) // 4 * 4
    f32 = x.dtype == torch.float32 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    if not (_o.RRELU_IN_GEMM and x.dim() == 2 and x.is_cuda and f32 and M % 4 == 0 and linear_supported(K, M) and not _dense_route(x, weight, bias)
            and 0 < lower <= upper and 0 <= drop_p < 1 and _lib.load().glam_ts_gemm_rrelu_supported(Kp, M) == 1):
        return None
    if Kp != K:
        if x.requires_grad and torch.is_grad_enabled():
            return None
        x = _o.pad_cols(x, Kp)
    nd, took = _node_arg(node, M)
    y, y_drop = _Linear.apply(x, weight, bias, False, (float(lower), float(upper), float(drop_p)), nd)
    if y_drop is not None:
        _o.register_dropped(y, y_drop, drop_p)
    if took:      # (``node = (staged, H * Cp)`` of ops.first_node_spec: the TripletMessage behind finds its node product ready)
        _o.register_node_product(y_drop if y_drop is not None else y, node[0], took["xw"], took["a_ij"])
    return y


class _LinearSplit(torch.autograd.Function):
    """(y1[N,M1], y2[N,M2]) = x[N,K] @ wt[K, M1+M2] with the two column blocks written to separate tensors
    (node features + packed attention scalars of the single-head layers).  K, M1, M2 multiples of 4."""

    @staticmethod
    def forward(ctx, x, wt, M1):
        require_device(x, wt)
        x, wt = f32c(x, "x"), f32c(wt, "weight")
        N, K = x.shape
        M = wt.size(1)
        M2 = M - M1
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, **f)
        check(lib.glam_ts_gemm_make_image(ptr(wt), M, 0, K, M, ptr(img), stream()), "glam_ts_gemm_make_image")
        y1, y2 = torch.empty(N, M1, **f), torch.empty(N, M2, **f)
        check(lib.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), None, ptr(y1), M1, M1, ptr(y2), M2, M2, N, stream()), "glam_ts_gemm")
        ctx.save_for_backward(x, wt)
        ctx.M1 = M1
        return y1, y2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy1, dy2):
        x, wt = ctx.saved_tensors
        N, K = x.shape
        M = wt.size(1)
        M1 = ctx.M1
        M2 = M - M1
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        dy1, dy2 = f32c(dy1, "dy1"), f32c(dy2, "dy2")
        dx = None
        if ctx.needs_input_grad[0]:
            img = torch.empty(lib.glam_ts_gemm_image_bytes(M, K) // 4, **f)
            check(lib.glam_ts_gemm_make_image(ptr(wt), M, 1, M, K, ptr(img), stream()), "glam_ts_gemm_make_image")
            dx = torch.empty(N, K, **f)
            check(lib.glam_ts_gemm(ptr(dy1), M1, M1, ptr(dy2), M2, M2, ptr(img), None, ptr(dx), K, K, None, 0, 0, N, stream()),
                  "glam_ts_gemm")
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        dwt = torch.empty(K, M, **f)     # out[i = m, j = k] written at dwt[k, m]
        check(lib.glam_wgrad_gemm(ptr(dy1), M1, M1, ptr(dy2), M2, M2, 0, ptr(x), K, K, 0, N, ptr(dwt), 1, M, ptr(ws), ws.numel(),
                                  stream()), "glam_wgrad_gemm")
        return dx, dwt, None


def linear_split(x, wt, M1):
    """``x @ wt`` split into the first ``M1`` and the remaining columns (both contiguous)."""
    return _LinearSplit.apply(x, wt, M1)


def linear_split_supported(K, M):
    return K % 4 == 0 and M % 4 == 0 and K <= 64 and M <= 192      # K is the J side of the weight-gradient kernel


class _GruGates(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, gh, h):
        require_device(gi, gh, h)
        gi, gh, h = f32c(gi, "gi"), f32c(gh, "gh"), f32c(h, "h")
        N, C = h.shape
        h_new = torch.empty_like(h)
        check(_lib.load().glam_gru_gates_fwd(ptr(gi), ptr(gh), ptr(h), N, C, ptr(h_new), stream()), "glam_gru_gates_fwd")
        ctx.save_for_backward(gi, gh, h)
        return h_new

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_hnew):
        gi, gh, h = ctx.saved_tensors
        N, C = h.shape
        d_hnew = f32c(d_hnew, "d_hnew")
        d_gi, d_gh, d_h = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        check(_lib.load().glam_gru_gates_bwd(ptr(gi), ptr(gh), ptr(h), ptr(d_hnew), N, C, ptr(d_gi), ptr(d_gh), ptr(d_h),
                                             stream()), "glam_gru_gates_bwd")
        return d_gi, d_gh, d_h


class _GruTail(torch.autograd.Function):
    """GRU gates + residual + activation in one launch per direction; returns (out, h_new)."""

    @staticmethod
    def forward(ctx, gi, gh, h, identity, act, slope):
        require_device(gi, gh, h)
        gi, gh, h = f32c(gi, "gi"), f32c(gh, "gh"), f32c(h, "h")
        identity = None if identity is None else f32c(identity, "identity")
        N, C = h.shape
        h_new, out = torch.empty_like(h), torch.empty_like(h)
        check(_lib.load().glam_gru_tail_fwd(ptr(gi), ptr(gh), ptr(h), ptr(identity), N, C, act, float(slope), ptr(h_new), ptr(out),
                                            stream()), "glam_gru_tail_fwd")
        ctx.save_for_backward(gi, gh, h, out)
        ctx.cfg = (act, float(slope), identity is not None)
        return out, h_new

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_hstate):
        gi, gh, h, out = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        N, C = h.shape
        d_out = f32c(d_out, "d_out")
        d_hstate = None if d_hstate is None else f32c(d_hstate, "d_hstate")
        d_gi, d_gh, d_h = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        d_id = torch.empty_like(h) if has_res else None
        check(_lib.load().glam_gru_tail_bwd(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_hstate), N, C, act, slope,
                                            ptr(d_gi), ptr(d_gh), ptr(d_h), ptr(d_id), stream()), "glam_gru_tail_bwd")
        return d_gi, d_gh, d_h, d_id, None, None


ACT_CODES = {"none": 0, "relu": 1, "leaky": 2, "celu": 3, "rrelu": 4}


def gru_tail(x, h, identity, w_ih, w_hh, b_ih, b_hh, act="none", slope=0.0, celu_in=False, rng=None, node=None):
    """``h_new = GRU(celu(x) if celu_in else x, h)`` (one step), ``out = act(h_new + identity)`` (src_1gp/layer.py:261-266):
    the two gate GEMMs plus ONE elementwise launch per direction.  Returns ``(out, h_new)``.
    ``rng = (rr_lower, rr_upper, drop_p)`` (training mode of the reference's defaults): ``act == "rrelu"`` draws its slopes in
    the kernel and, with ``drop_p > 0``, the kernel also writes ``Dropout(drop_p)(out)`` and registers it as the dropped twin
    of ``out`` (``take_dropped``) — available on the one-node path (24 <= C <= 60, odd widths zero-padded); elsewhere the caller applies
    ``ops.rrelu`` / ``ops.dropout`` itself (``rng`` must then be None)."""
    C = h.size(1)
    if rng is not None and not gru_rng_supported(C, w_ih, b_ih, b_hh):
        raise GlamHipError("gru_tail(rng=...) needs the one-node GRU block (24 <= C <= 60)")
    if gru_block_supported(C, w_ih, b_ih, b_hh):
        # node = (staged, H * Cp) of ops.next_node_spec: the launch also writes the node product of the block's next application
        return _gru_block(x, h, identity, w_ih, w_hh, b_ih, b_hh, ACT_CODES[act], slope, celu_in, rng, node)
    Cp = (C + 3) // 4 * 4
    if _gru_padded_supported(C, w_ih, b_ih, b_hh):
        # odd widths: the same node at Cp with gate-wise zero-padded weights (built once per model pass).  Pad channels
        # stay exactly zero through the step: gates r = z = 1/2, n = tanh(0) = 0, h' = z * 0 = 0, act(0 + 0) = 0.
        def build():
            return _gru_padded(w_ih, w_hh, b_ih, b_hh, C, Cp)
        wi, wh, bi, bh = _o.scoped_weights(("gru-pad", id(w_ih), id(w_hh), id(b_ih), id(b_hh)), w_ih, build)
        # (with rng the dropped twin is registered for out_p: layer._apply_dropout finds it through the padded base of the view)
        out_p, hn_p = _gru_block(_o.pad_cols(x, Cp), _o.pad_cols(h, Cp), None if identity is None else _o.pad_cols(identity, Cp),
                                 wi, wh, bi, bh, ACT_CODES[act], slope, celu_in, rng)
        return _o.slice_cols(out_p, C), _o.slice_cols(hn_p, C)
    if b_ih is not None and b_hh is not None and tuple(w_ih.shape) == (3 * C, C) and linear_tall_supported(Cp, 3 * Cp) and h.size(0) >= 64:
        # wide GRU (hid_dim_alpha = 6): library GEMMs for the gate products, k_wgrad for their weight gradients, at Cp
        def build_wide():
            return _gru_padded(w_ih, w_hh, b_ih, b_hh, C, Cp)
        wi, wh, bi, bh = _o.scoped_weights(("gru-pad", id(w_ih), id(w_hh), id(b_ih), id(b_hh)), w_ih, build_wide) if Cp != C else \
            (w_ih, w_hh, b_ih, b_hh)
        x_p, h_p = _o.pad_cols(x, Cp), _o.pad_cols(h, Cp)
        if torch.is_grad_enabled() and Cp <= 96:
            # the four images of the two gate linears (forward and input-gradient products) from one launch instead of four
            Kc, Mc = Cp, 3 * Cp
            _o.prestage(None, [("fwd", ("lin", id(wi)), wi, wi, Kc, 1, Kc, Mc, Kc), ("fwd", ("lin", id(wh)), wh, wh, Kc, 1, Kc, Mc, Kc),
                               ("bwd", ("lin-t", id(wi)), wi, wi, Kc, 0, Mc, Kc, Mc), ("bwd", ("lin-t", id(wh)), wh, wh, Kc, 0, Mc, Kc, Mc)])
        out_p, hn_p = _GruTail.apply(_linear_tall_node(x_p, wi, bi, celu_in), _linear_tall_node(h_p, wh, bh), h_p,
                                     None if identity is None else _o.pad_cols(identity, Cp), ACT_CODES[act], slope)
        return _o.slice_cols(out_p, C), _o.slice_cols(hn_p, C)
    if celu_in:
        x = torch.celu(x)
    return _GruTail.apply(linear(x, w_ih, b_ih), linear(h, w_hh, b_hh), h, identity, ACT_CODES[act], slope)




def _gru_padded(w_ih, w_hh, b_ih, b_hh, C, Cp):
    """Gate-wise zero-padded GRU parameters ``[3C, C] -> [3Cp, Cp]``, ``[3C] -> [3Cp]``: all four from one launch."""
    wsp, bsp = ((3, C, C), (Cp, Cp), (3 * Cp, Cp)), ((1, 3, C), (3, Cp), (3 * Cp,))
    return _o.pad_group([(w_ih, *wsp), (w_hh, *wsp), (b_ih, *bsp), (b_hh, *bsp)])


def _want_gru_ws(lib, N, C):
    """The warp-specialised 3 x bf16 GRU step (``ops.GRU_WS``, default on; GLAM_X3=0 keeps every dense product on the fp32 matrix cores)."""
    return N > 0 and _o.GRU_WS == "1" and _lib.route_enabled("x3") and lib.glam_gru_ws_supported(C) == 1


def _gru_pre(lib, scope, w_ih, w_hh, C, dev, st):
    """[2, bytes] uint8: the forward and the backward image of ``glam_gru_ws_make_pre`` for this pair of gate matrices — one launch per
    weight scope (a training step), shared by every application of the block and by its backward."""
    def build():
        buf = torch.empty(2, lib.glam_gru_ws_pre_bytes(), dtype=torch.uint8, device=dev)
        check(lib.glam_gru_ws_make_pre(ptr(w_ih), ptr(w_hh), C, ptr(buf[0]), ptr(buf[1]), st), "glam_gru_ws_make_pre")
        return buf
    return _o._scoped(scope.fwd if scope else None, ("gru-pre", id(w_ih), id(w_hh)), w_ih, build)


def gru_images_plain(N, C):
    """True when the GRU step of this size runs on the four plain ``k_ts_gemm`` images of its gate matrices (the warp-specialised step,
    or the unfused gate linears) — the set ``ops.prestage`` can build ahead; the fp32 fused step needs its gate-padded images too."""
    lib = _lib.load()
    if N <= 0 or not linear_supported(C, 3 * C) or lib.glam_ts_gemm_image_bytes(3 * C, C) <= 0:
        return False
    if _want_gru_ws(lib, N, C):
        return not _o.GRU_PRE          # (the warp-specialised step reads its own pre-split images: glam_gru_ws_make_pre)
    return not (_want_gru_fused(N) and lib.glam_gru_fused_supported(C))


def gru_images_pre(N, C):
    """True when the GRU step of this size reads the pre-split images of ``glam_gru_ws_make_pre`` (the warp-specialised step, default)."""
    return bool(_o.GRU_PRE) and _want_gru_ws(_lib.load(), N, C)


def _want_gru_fused(N):
    return _o.GRU_FUSED in ("1", True) or (_o.GRU_FUSED == "auto" and N >= _o.GRU_FUSED_MIN_NODES)

def _gru_block(x, h, identity, w_ih, w_hh, b_ih, b_hh, act, slope, celu_in, rng=None, node=None):
    """``_GruBlock`` with the gradients of its four parameters carried across the block's applications (see _ParamBundle)."""
    M, C = w_ih.shape
    def split(flat):      # [d_w_ih | d_b_ih | d_w_hh | d_b_hh], every piece contiguous: autograd takes the views without a copy
        w1, b1, w2, b2 = flat.split([M * C, M, M * C, M])
        return w1.view(M, C), w2.view(M, C), b1, b2
    key = ("carry-gru", id(w_ih))
    hit = _o._SCOPE.fwd.get(key) if _o._SCOPE is not None else None
    first = not (hit is not None and hit[0] is w_ih)        # the block's first application of this pass: its backward runs LAST
    carry = _o._carry_for(key, (w_ih, w_hh, b_ih, b_hh), 2 * M * (C + 1), split)
    took = {} if node is not None else None        # (filled by the forward when its route wrote the product: xw, a_ij)
    out, h_new, out_drop, carry = _GruBlock.apply(x, h, identity, w_ih, w_hh, b_ih, b_hh, act, slope, celu_in, carry, rng, first,
                                                  None if node is None else (node[0], node[1], took))
    if carry is not None:
        _o._carry_store(key, w_ih, carry)
    if out_drop is not None:
        _o.register_dropped(out, out_drop, rng[2])
    if took:
        _o.register_node_product(out_drop if out_drop is not None else out, node[0], took["xw"], took["a_ij"])
    return out, h_new


_GRU_BATCH_MIN_ROWS = 512       # (a wave's row range must fit into one operand set: glam_wgrad_gemm_pair_split_seg)


class _GruBlock(torch.autograd.Function):
    """The whole GRU step of a MessageBlock as ONE autograd node: both gate GEMMs + gates/residual/activation forward;
    gate backward + both input-gradient GEMMs + BOTH weight-gradient products in one launch pair backward.  Besides the
    launches it saves (one weight-gradient launch and one reduction per step) it replaces three Python autograd nodes by
    one, which is what an eagerly issued training step is bound by."""

    @staticmethod
    def forward(ctx, x, h, identity, w_ih, w_hh, b_ih, b_hh, act, slope, celu_in, carry=None, rng=None, first_app=True, node=None):
        ctx.set_materialize_grads(False)     # unused outputs (the last step's h', its dropped twin) arrive as None, not as zero fills
        require_device(x, h, w_ih, w_hh, b_ih, b_hh)
        x, h = f32c(x, "x"), f32c(h, "h")
        w_ih, w_hh, b_ih, b_hh = f32c(w_ih, "weight_ih"), f32c(w_hh, "weight_hh"), f32c(b_ih, "bias_ih"), f32c(b_hh, "bias_hh")
        identity = None if identity is None else f32c(identity, "identity")
        N, C = h.shape
        M = 3 * C
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        scope = _o._SCOPE

        def image(w):
            def build():
                img = torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, **f)
                check(lib.glam_ts_gemm_make_image(ptr(w), C, 1, C, M, ptr(img), stream()), "glam_ts_gemm_make_image")
                return img
            return _o._scoped(scope.fwd if scope else None, ("lin", id(w)), w, build)

        st = stream()
        ws_route = _want_gru_ws(lib, N, C)
        # (the warp-specialised step: gi = the gates [r | z | n | gh_n], no gh — see ops.GRU_GATES)
        gates = bool(ws_route and _o.GRU_GATES)
        gi, gh = (torch.empty(N, 4 * C, **f), None) if gates else (torch.empty(N, M, **f), torch.empty(N, M, **f))
        pre = _gru_pre(lib, scope, w_ih, w_hh, C, dev, st) if (ws_route and _o.GRU_PRE) else None
        if scope is not None and pre is None:
            # all four images of the step (forward and input-gradient images of both gate matrices) in ONE launch, shared by the
            # message_steps applications of the block through the scope tables
            ka, kb = ("lin", id(w_ih)), ("lin", id(w_hh))
            ha, hb = scope.fwd.get(ka), scope.fwd.get(kb)
            if not (ha is not None and ha[0] is w_ih and hb is not None and hb[0] is w_hh):
                nf, nb = lib.glam_ts_gemm_image_bytes(C, M) // 4, lib.glam_ts_gemm_image_bytes(M, C) // 4
                ia, ib, ta, tb = torch.empty(nf, **f), torch.empty(nf, **f), torch.empty(nb, **f), torch.empty(nb, **f)
                if N > 0 and _want_gru_fused(N) and lib.glam_gru_fused_supported(C) and not _want_gru_ws(lib, N, C):
                    # ... and the two gate-padded images of the fused step: six re-layouts of the same two matrices, one launch
                    fused = torch.empty(2, lib.glam_gru_fused_image_bytes() // 4, **f)
                    check(lib.glam_gru_make_images(ptr(w_ih), ptr(w_hh), C, ptr(ia), ptr(ib), ptr(ta), ptr(tb), ptr(fused[0]), ptr(fused[1]),
                                                   st), "glam_gru_make_images")
                    scope.fwd[("gru-fused", id(w_ih), id(w_hh))] = (w_ih, fused)
                else:
                    check(lib.glam_ts_gemm_make_image_quad(ptr(w_ih), ptr(w_hh), C, M, ptr(ia), ptr(ib), ptr(ta), ptr(tb), st),
                          "glam_ts_gemm_make_image_quad")
                scope.fwd[ka], scope.fwd[kb] = (w_ih, ia), (w_hh, ib)
                scope.bwd[ka], scope.bwd[kb] = (w_ih, ta), (w_hh, tb)
        h_new, out = torch.empty_like(h), torch.empty_like(h)
        out_drop, eff = None, None
        if rng is None and act == ACT_CODES["rrelu"]:
            raise GlamHipError("gru block: act 'rrelu' needs rng=(lower, upper, drop_p)")
        if rng is not None:
            lo, hi, p = (float(v) for v in rng)
            eff = torch.empty(2, dtype=torch.int64, device=dev)
            out_drop = torch.empty_like(h) if p > 0 else None
        # celu_in: x is the raw conv output and the CELU of layer.py:261 is applied inside the gate GEMM's operand load
        x_keep, x_is_celu = x, False
        if ws_route:
            # the warp-specialised 3 x bf16 form of the fused step (block.hip: k_gru_fwd_ws).  With the folded CELU the launch also writes
            # celu(x): the backward and the weight gradient read THAT instead of x and apply no exponential of their own
            xc = torch.empty_like(x) if (celu_in and N > 0) else None
            if pre is not None and node is not None and N > 0:
                # ... and with the node product of the block's NEXT application (ops.NODE_IN_GRU): its producers multiply every finished
                # tile by [W_node | Wa] before it leaves LDS
                staged, cols, took = node
                nimg = staged[lib.glam_triplet_staged_node_fragments(cols // C, C, 4):]      # ([W_node | Wa] as the producers' operand fragments)
                xw, a_ij = torch.empty(N, cols, **f), torch.empty(N, 8, **f)
                if rng is None:
                    check(lib.glam_gru_ws_fwd_pre_node(ptr(x), ptr(h), ptr(identity), ptr(pre[0]), ptr(b_ih), ptr(b_hh), N, C, int(celu_in), act,
                                                       float(slope), ptr(gi), ptr(gh), ptr(h_new), ptr(out), ptr(xc), ptr(nimg), cols, ptr(xw),
                                                       ptr(a_ij), st), "glam_gru_ws_fwd_pre_node")
                else:
                    check(lib.glam_gru_ws_rng_fwd_pre_node(ptr(x), ptr(h), ptr(identity), ptr(pre[0]), ptr(b_ih), ptr(b_hh), N, C, int(celu_in),
                                                           act, float(slope), lo, hi, p, ptr(_o.rng_state(dev)), ptr(eff), ptr(gi), ptr(gh),
                                                           ptr(h_new), ptr(out), ptr(out_drop), ptr(xc), ptr(nimg), cols, ptr(xw), ptr(a_ij), st),
                          "glam_gru_ws_rng_fwd_pre_node")
                took["xw"], took["a_ij"] = xw, a_ij
            elif pre is not None:
                # ... on the gate matrices as pre-split operand fragments (glam_gru_ws_make_pre: once per weight update, not per block)
                if rng is None:
                    check(lib.glam_gru_ws_fwd_pre(ptr(x), ptr(h), ptr(identity), ptr(pre[0]), ptr(b_ih), ptr(b_hh), N, C, int(celu_in), act,
                                                  float(slope), ptr(gi), ptr(gh), ptr(h_new), ptr(out), ptr(xc), st), "glam_gru_ws_fwd_pre")
                else:
                    check(lib.glam_gru_ws_rng_fwd_pre(ptr(x), ptr(h), ptr(identity), ptr(pre[0]), ptr(b_ih), ptr(b_hh), N, C, int(celu_in), act,
                                                      float(slope), lo, hi, p, ptr(_o.rng_state(dev)), ptr(eff), ptr(gi), ptr(gh), ptr(h_new),
                                                      ptr(out), ptr(out_drop), ptr(xc), st), "glam_gru_ws_rng_fwd_pre")
            elif rng is None:
                img_a, img_b = image(w_ih), image(w_hh)       # (the plain k_ts_gemm images: GLAM_GRU_PRE=0)
                check(lib.glam_gru_ws_fwd_xc(ptr(x), ptr(h), ptr(identity), ptr(img_a), ptr(img_b), ptr(b_ih), ptr(b_hh), N, C,
                                             int(celu_in), act, float(slope), ptr(gi), ptr(gh), ptr(h_new), ptr(out), ptr(xc), st),
                      "glam_gru_ws_fwd_xc")
            else:
                img_a, img_b = image(w_ih), image(w_hh)
                check(lib.glam_gru_ws_rng_fwd_xc(ptr(x), ptr(h), ptr(identity), ptr(img_a), ptr(img_b), ptr(b_ih), ptr(b_hh), N, C,
                                                 int(celu_in), act, float(slope), lo, hi, p, ptr(_o.rng_state(dev)), ptr(eff), ptr(gi), ptr(gh),
                                                 ptr(h_new), ptr(out), ptr(out_drop), ptr(xc), st), "glam_gru_ws_rng_fwd_xc")
            if xc is not None:
                x_keep, x_is_celu = xc, True
        elif N > 0 and _want_gru_fused(N) and lib.glam_gru_fused_supported(C):
            # both gate linears + gates + residual + activation (+ RReLU / Dropout) in ONE launch (bit-identical to the sequence below)
            def build_fused():
                nb = lib.glam_gru_fused_image_bytes() // 4
                buf = torch.empty(2, nb, **f)
                check(lib.glam_gru_fused_make_images(ptr(w_ih), ptr(w_hh), C, ptr(buf[0]), ptr(buf[1]), st), "glam_gru_fused_make_images")
                return buf
            imgs = _o._scoped(scope.fwd if scope else None, ("gru-fused", id(w_ih), id(w_hh)), w_ih, build_fused)
            if rng is None:
                check(lib.glam_gru_fused_fwd(ptr(x), ptr(h), ptr(identity), ptr(imgs[0]), ptr(imgs[1]), ptr(b_ih), ptr(b_hh), N, C,
                                             int(celu_in), act, float(slope), ptr(gi), ptr(gh), ptr(h_new), ptr(out), st), "glam_gru_fused_fwd")
            else:
                check(lib.glam_gru_fused_rng_fwd(ptr(x), ptr(h), ptr(identity), ptr(imgs[0]), ptr(imgs[1]), ptr(b_ih), ptr(b_hh), N, C,
                                                 int(celu_in), act, float(slope), lo, hi, p, ptr(_o.rng_state(dev)), ptr(eff), ptr(gi), ptr(gh),
                                                 ptr(h_new), ptr(out), ptr(out_drop), st), "glam_gru_fused_rng_fwd")
        else:
            if _o.GEMM_PAIR:     # both gate linears in ONE launch (two products of the same kernel variant share the CUs)
                img_a, img_b = image(w_ih), image(w_hh)     # both alive until the launch is enqueued (outside a scope they are temporaries:
                #                                             the allocator would hand the first one's memory to the second)
                check(lib.glam_ts_gemm_pair(ptr(x), C, C, int(celu_in), ptr(img_a), ptr(b_ih), ptr(gi), M, M, None, 0, None, 0,
                                            ptr(h), C, C, 0, ptr(img_b), ptr(b_hh), ptr(gh), M, M, None, 0, None, 0, N, st),
                      "glam_ts_gemm_pair")
            else:
                check(lib.glam_ts_gemm_celu(ptr(x), C, C, int(celu_in), ptr(image(w_ih)), ptr(b_ih), ptr(gi), M, M, None, 0, N, st),
                      "glam_ts_gemm_celu")
                check(lib.glam_ts_gemm(ptr(h), C, C, None, 0, 0, ptr(image(w_hh)), ptr(b_hh), ptr(gh), M, M, None, 0, 0, N, st), "glam_ts_gemm")
            if rng is None:
                check(lib.glam_gru_tail_fwd(ptr(gi), ptr(gh), ptr(h), ptr(identity), N, C, act, float(slope), ptr(h_new), ptr(out), st),
                      "glam_gru_tail_fwd")
            else:     # training mode: RReLU slopes / the next conv's Dropout mask drawn inside the launch
                check(lib.glam_gru_tail_rng_fwd(ptr(gi), ptr(gh), ptr(h), ptr(identity), N, C, act, float(slope), lo, hi, p,
                                                ptr(_o.rng_state(dev)), ptr(eff), ptr(h_new), ptr(out), ptr(out_drop), st), "glam_gru_tail_rng_fwd")
        ctx.save_for_backward(x_keep, h, gi, gh, out, w_ih, w_hh)
        ctx.x_is_celu = x_is_celu
        ctx.gates = gates
        ctx.pre = pre
        # the first application of a block seeds the GRU state with the block input, which is also the skip connection (layer.py:253-254):
        # one tensor, two roles — the backward then returns ONE gradient for it (k_gru_bwd_ws adds d_identity into d_h)
        ctx.same_h_id = (identity is not None and identity.data_ptr() == h.data_ptr() and identity.shape == h.shape
                         and identity.stride() == h.stride())
        ctx.eff = eff
        ctx.cfg = (act, float(slope), identity is not None, bool(celu_in), None if rng is None else tuple(float(v) for v in rng))
        ctx.scope = scope
        ctx.carried = carry is not None
        ctx.first_app = bool(first_app)
        return out, h_new, out_drop, (carry.view(-1) if ctx.carried else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_hstate, d_out_drop=None, d_carry=None):
        x, h, gi, gh, out, w_ih, w_hh = ctx.saved_tensors
        act, slope, has_res, celu_in, rng = ctx.cfg
        # x_is_celu: `x` holds celu(x) (the warp-specialised forward wrote it): celu' comes from it (flag 2) and Q needs no CELU
        celu_bwd = 2 if ctx.x_is_celu else int(celu_in)
        celu_q = celu_in and not ctx.x_is_celu
        N, C = h.shape
        M = 3 * C
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        st = stream()
        d_out = None if d_out is None else f32c(d_out, "d_out")
        d_out_drop = None if d_out_drop is None else f32c(d_out_drop, "d_out_drop")
        d_hstate = None if d_hstate is None else f32c(d_hstate, "d_hstate")
        gates = ctx.gates                     # gi = [r | z | n | gh_n]; d_gi = [d_pr | d_pz | d_pn | d_pn r], no d_gh
        d_gi, d_gh, d_h = torch.empty_like(gi), (None if gates else torch.empty_like(gh)), torch.empty_like(h)
        d_id = torch.empty_like(h) if has_res else None
        scope = ctx.scope

        def image_t(w):
            def build():
                img = torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, **f)
                check(lib.glam_ts_gemm_make_image(ptr(w), C, 0, M, C, ptr(img), stream()), "glam_ts_gemm_make_image")
                return img
            return _o._scoped(scope.bwd if scope else None, ("lin", id(w)), w, build)

        ws = gates or _want_gru_ws(lib, N, C)
        if ws:
            # gate gradients + both input-gradient products in ONE launch (block.hip: k_gru_bwd_ws); d_h comes out complete
            dx = torch.empty(N, C, **f)
            pre = ctx.pre
            img_a, img_b = (None, None) if pre is not None else (image_t(w_ih), image_t(w_hh))
            merge = int(has_res and ctx.same_h_id)
            if merge:
                d_id = None                   # (its gradient is part of d_h: the two inputs are one tensor)
            if rng is None:
                if d_out is None:
                    d_out = torch.zeros_like(h)
                if pre is not None:
                    check(lib.glam_gru_bwd_ws_pre(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_hstate), ptr(x), ptr(pre[1]), N, C,
                                                  celu_bwd, act, slope, merge, ptr(d_gi), ptr(d_gh), ptr(d_id), ptr(dx), ptr(d_h), st),
                          "glam_gru_bwd_ws_pre")
                else:
                    check(lib.glam_gru_bwd_ws(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_hstate), ptr(x), ptr(img_a), ptr(img_b), N,
                                              C, celu_bwd, act, slope, merge, ptr(d_gi), ptr(d_gh), ptr(d_id), ptr(dx), ptr(d_h), st),
                          "glam_gru_bwd_ws")
            else:
                if d_out is None and d_out_drop is None:
                    d_out = torch.zeros_like(h)
                if pre is not None:
                    check(lib.glam_gru_bwd_ws_rng_pre(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_out_drop), ptr(d_hstate), ptr(x),
                                                      ptr(pre[1]), N, C, celu_bwd, act, slope, rng[0], rng[1], rng[2], ptr(ctx.eff), merge,
                                                      ptr(d_gi), ptr(d_gh), ptr(d_id), ptr(dx), ptr(d_h), st), "glam_gru_bwd_ws_rng_pre")
                else:
                    check(lib.glam_gru_bwd_ws_rng(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_out_drop), ptr(d_hstate), ptr(x),
                                                  ptr(img_a), ptr(img_b), N, C, celu_bwd, act, slope, rng[0], rng[1], rng[2], ptr(ctx.eff), merge,
                                                  ptr(d_gi), ptr(d_gh), ptr(d_id), ptr(dx), ptr(d_h), st), "glam_gru_bwd_ws_rng")
            dh = d_h
        elif rng is None:
            if d_out is None:
                d_out = torch.zeros_like(h)
            check(lib.glam_gru_tail_bwd(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_hstate), N, C, act, slope, ptr(d_gi),
                                        ptr(d_gh), ptr(d_h), ptr(d_id), st), "glam_gru_tail_bwd")
        else:
            if d_out is None and d_out_drop is None:
                d_out = torch.zeros_like(h)
            check(lib.glam_gru_tail_rng_bwd(ptr(gi), ptr(gh), ptr(h), ptr(out), ptr(d_out), ptr(d_out_drop), ptr(d_hstate), N, C, act, slope,
                                            rng[0], rng[1], rng[2], ptr(ctx.eff), ptr(d_gi), ptr(d_gh), ptr(d_h), ptr(d_id), st),
                  "glam_gru_tail_rng_bwd")
        if ws:
            pass
        elif _o.GEMM_PAIR:
            dx, dh = torch.empty(N, C, **f), torch.empty(N, C, **f)
            # with the folded CELU the epilogue multiplies by celu'(x): dx is the gradient of the RAW input
            # d_h = d_gh @ W_hh^T + the direct z * g path of the gate equations (the addend of the GEMM's epilogue); both products in one launch
            img_a, img_b = image_t(w_ih), image_t(w_hh)
            check(lib.glam_ts_gemm_pair(ptr(d_gi), M, M, 0, ptr(img_a), None, ptr(dx), C, C, ptr(x) if celu_in else None, C, None, 0,
                                        ptr(d_gh), M, M, 0, ptr(img_b), None, ptr(dh), C, C, None, 0, ptr(d_h), C, N, st),
                  "glam_ts_gemm_pair")
        else:
            dx, dh = torch.empty(N, C, **f), torch.empty(N, C, **f)
            check(lib.glam_ts_gemm_celu(ptr(d_gi), M, M, 0, ptr(image_t(w_ih)), None, ptr(dx), C, C, ptr(x) if celu_in else None, C, N, st),
                  "glam_ts_gemm_celu")
            check(lib.glam_ts_gemm_add(ptr(d_gh), M, M, ptr(image_t(w_hh)), None, ptr(dh), C, C, ptr(d_h), C, N, st), "glam_ts_gemm_add")
        if ctx.carried and scope is not None and _o.GRU_WGRAD_BATCH and N >= _GRU_BATCH_MIN_ROWS:
            # The weight gradients of ALL applications of the block in one launch pair: every application but the first parks its
            # operands in the scope and passes the carry on untouched; the first one (its backward runs last: everything later in
            # the forward depends on its outputs) multiplies the parked sets together — [d_gi_1; d_gi_2; d_gi_3]^T [x_1; x_2; x_3] —
            # three sets per launch.  3 launches + 3 reductions -> 1 + 1 per training step at message_steps = 3.
            key = ("gru-parked", id(w_ih))
            parked = scope.bwd.setdefault(key, (w_ih, []))[1]
            parked.append((d_gi, x, d_gh, h, bool(celu_q), gates))
            if not ctx.first_app:
                return dx, dh, d_id, None, None, None, None, None, None, None, d_carry, None, None, None
            sets = list(parked)
            parked.clear()
            flat = torch.empty(2 * M * (C + 1), **f)
            dw_ih, db_ih, dw_hh, db_hh = flat.split([M * C, M, M * C, M])
            add = [None] * 4 if d_carry is None else list(f32c(d_carry, "d_carry").split([M * C, M, M * C, M]))
            vp = ctypes.c_void_p
            while sets:
                grp = [t for t in sets if t[4:] == sets[0][4:]][:3]
                sets = [t for t in sets if all(t is not u for u in grp)]
                n = len(grp)
                ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
                arr = lambda i: (vp * n)(*[t[i].data_ptr() for t in grp])
                if grp[0][5]:
                    check(lib.glam_wgrad_gemm_gru_gates_seg(n, arr(0), C, arr(1), C, int(grp[0][4]), arr(3), C, ptr(dw_ih), ptr(db_ih),
                                                            ptr(dw_hh), ptr(db_hh), N, ptr(ws), ws.numel(), ptr(add[0]), ptr(add[1]),
                                                            ptr(add[2]), ptr(add[3]), st), "glam_wgrad_gemm_gru_gates_seg")
                else:
                    check(lib.glam_wgrad_gemm_pair_split_seg(n, arr(0), M, M, arr(1), C, C, int(grp[0][4]), ptr(dw_ih), ptr(db_ih), arr(2), M, M,
                                                             arr(3), C, C, 0, ptr(dw_hh), ptr(db_hh), N, ptr(ws), ws.numel(), ptr(add[0]),
                                                             ptr(add[1]), ptr(add[2]), ptr(add[3]), st), "glam_wgrad_gemm_pair_split_seg")
                add = [dw_ih, db_ih, dw_hh, db_hh]        # a further group adds onto the result in place
            return dx, dh, d_id, None, None, None, None, None, None, None, flat, None, None, None
        # [d_W | d_b] of both linears: out[m, k] = sum_n dy[n, m] * [x | 1][n, k], two products, one launch + one reduction
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        # one buffer [d_w_ih | d_b_ih | d_w_hh | d_b_hh], contiguous pieces; a gradient carry (same layout) is added by the reduction
        flat = torch.empty(2 * M * (C + 1), **f)
        dw_ih, db_ih, dw_hh, db_hh = flat.split([M * C, M, M * C, M])
        dc = [None] * 4
        if ctx.carried and d_carry is not None and N > 0:
            dc = f32c(d_carry, "d_carry").split([M * C, M, M * C, M])
            d_carry = None
        if gates:
            one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
            check(lib.glam_wgrad_gemm_gru_gates_seg(1, one(d_gi), C, one(x), C, int(celu_q), one(h), C, ptr(dw_ih), ptr(db_ih), ptr(dw_hh),
                                                    ptr(db_hh), N, ptr(ws), ws.numel(), ptr(dc[0]), ptr(dc[1]), ptr(dc[2]), ptr(dc[3]), st),
                  "glam_wgrad_gemm_gru_gates_seg")
        else:
            check(lib.glam_wgrad_gemm_pair_split(ptr(d_gi), M, M, ptr(x), C, C, int(celu_q), ptr(dw_ih), ptr(db_ih),
                                                 ptr(d_gh), M, M, ptr(h), C, C, 0, ptr(dw_hh), ptr(db_hh), N, ptr(ws), ws.numel(),
                                                 ptr(dc[0]), ptr(dc[1]), ptr(dc[2]), ptr(dc[3]), st), "glam_wgrad_gemm_pair_split")
        if ctx.carried:
            return dx, dh, d_id, None, None, None, None, None, None, None, (flat if d_carry is None else flat.add_(d_carry)), None, None, None
        return dx, dh, d_id, dw_ih.view(M, C), dw_hh.view(M, C), db_ih, db_hh, None, None, None, None, None, None, None


def gru_block_supported(C, w_ih, b_ih, b_hh):
    return C % 4 == 0 and C + 1 <= 64 and linear_supported(C, 3 * C) and 3 * C > 64 and b_ih is not None and b_hh is not None and \
        tuple(w_ih.shape) == (3 * C, C)        # C + 1 <= 64: both weight gradients in ONE k_wgrad launch


def _gru_padded_supported(C, w_ih, b_ih, b_hh):
    """Odd widths (hid_dim 30 / 45): the one-node block at Cp = ceil4(C) on gate-wise zero-padded parameters."""
    Cp = (C + 3) // 4 * 4
    return Cp != C and b_ih is not None and b_hh is not None and tuple(w_ih.shape) == (3 * C, C) and linear_supported(Cp, 3 * Cp) \
        and 3 * Cp > 64 and Cp + 1 <= 64


def gru_rng_supported(C, w_ih, b_ih, b_hh):
    """Widths whose GRU tail draws RReLU slopes / the next step's Dropout mask inside the kernel (``gru_tail(rng=...)``)."""
    return gru_block_supported(C, w_ih, b_ih, b_hh) or _gru_padded_supported(C, w_ih, b_ih, b_hh)


def gru_step(x, h, w_ih, w_hh, b_ih, b_hh):
    """One ``torch.nn.GRU(C, C)`` step with seq_len 1 on its own parameters (src_1gp/layer.py:247, :262)."""
    return _GruGates.apply(linear(x, w_ih, b_ih), linear(h, w_hh, b_hh), h)
