"""Functional ops over ``libglam_hip.so`` with autograd (host side of the C ABI).

Every function launches hand-written gfx950 kernels on torch's *current* HIP stream; PyTorch
only supplies device memory, the stream and the autograd tape.  There is no CPU path.

``GraphIndex`` is the CSR staging of one ``edge_index`` (by target for the forward gather, by
source — the transpose — for the backward scatter); it is cached per ``edge_index`` tensor
because the reference re-uses one edge list for every message step (``src_1gp/model.py:53-54``)
and its train loader does not shuffle (``src_1gp/trainer.py:37-38``).
"""
from __future__ import annotations

import collections
import contextlib
import os
import weakref

import torch

from . import _lib
from ._lib import GlamHipError, check, f32c, ptr, require_device, stream


# --------------------------------------------------------------------------------------
# CSR staging
# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# id validation of foreign batches
# --------------------------------------------------------------------------------------
# Tensors collated by glam_amd.data carry trust marks and are never read back.  Anything else (a PyG ``Batch``, hand-built
# tensors) has its node / graph ids range-checked ON THE DEVICE while the CSR / ptr is built (bad entries are dropped there: no
# out-of-bounds access either way); what remains is when the host learns about a raised flag:
#   "sync"      (default) read the flag back at once: IndexError at the call that staged the tensor, like torch's own index checks
#               on CPU — one host sync per NEW tensor (cached afterwards);
#   "deferred"  copy the flag to pinned host memory asynchronously and raise at the next staging call, at ``check_pending()`` or at
#               the end of ``Architecture.forward`` once the copy has landed — no host sync on the step (CUDA's own convention for
#               device-side errors: reported late, never lost as long as the program keeps calling into the library);
#   "off"       never look (the reference on a GPU: a device-side assert or silent garbage).
VALIDATE = os.environ.get("GLAM_VALIDATE", "sync")
_PENDING: collections.deque = collections.deque()
_PINNED_FLAGS: list = []


def _check_flag(err, message):
    if VALIDATE == "off":
        return
    if VALIDATE == "deferred" and not torch.cuda.is_current_stream_capturing():
        host = _PINNED_FLAGS.pop() if _PINNED_FLAGS else torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(err, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _PENDING.append((ev, host, message))
        return
    if int(err.item()) != 0:
        raise IndexError(message)


def poll_checks(block=False):
    """Raise the IndexError of a deferred id check whose flag has come back (``block``: wait for all of them)."""
    while _PENDING:
        ev, host, message = _PENDING[0]
        if block:
            ev.synchronize()
        elif not ev.query():
            return
        _PENDING.popleft()
        bad = int(host[0]) != 0
        _PINNED_FLAGS.append(host)
        if bad:
            raise IndexError(message + " (deferred check, GLAM_VALIDATE=deferred: raised after the call that staged the tensor)")


def check_pending():
    """Wait for every outstanding deferred id check and raise if one failed."""
    poll_checks(block=True)


class GraphIndex:
    """CSR-by-target ``(rowptr, src, eid)`` and CSR-by-source ``(colptr, dst, eid_t)`` of an
    int64 ``edge_index[2,E]`` over ``N`` nodes (PyG flow source_to_target: row 0 = source j,
    row 1 = target i).  int32, device resident, stable inside every segment."""

    def __init__(self, edge_index, num_nodes, validate=True):
        require_device(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise GlamHipError(f"edge_index must be int64 [2,E], got {edge_index.dtype} {tuple(edge_index.shape)}")
        ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
        self.N, self.E = int(num_nodes), int(ei.size(1))
        self.device = ei.device
        lib = _lib.load()
        dev = ei.device
        i32 = dict(dtype=torch.int32, device=dev)
        self.rowptr = torch.empty(self.N + 1, **i32)
        self.src = torch.empty(self.E, **i32)
        self.eid = torch.empty(self.E, **i32)
        # a WEAK reference: the index lives in a cache keyed on this tensor, and holding the tensor here would keep both
        # alive for ever (a loader that builds new batch objects every step leaked one CSR per step)
        self._ei_ref = weakref.ref(ei)
        self._t = None
        self._ell = False              # False: not built yet; None: in-degree > 4 somewhere; else (ell_src, ell_eid)
        self._ell_t = False            # the same by source (out-degree), for the pipelined backward B2
        self._err = torch.zeros(1, **i32)
        ws = torch.empty(lib.glam_csr_workspace_bytes(self.N, self.E), dtype=torch.uint8, device=dev)
        check(lib.glam_csr_build(ptr(ei), self.N, self.E, 0, ptr(self.rowptr), ptr(self.src), ptr(self.eid),
                                 ptr(self._err), ptr(ws), ws.numel(), stream()), "glam_csr_build")
        # tensors built by glam_amd.data's collation carry a trust mark — the tensor's version counter at marking time: ids valid by
        # construction (checked once on the host when the dataset was packed) and not written since: no read-back, so a loop over
        # fresh batches has no host sync per step
        poll_checks()
        if validate and getattr(edge_index, "_glam_trusted", None) != edge_index._version:   # same failure class as torch's index_select on CPU
            _check_flag(self._err, f"edge_index holds node ids outside [0, {self.N})")

    # the pipelined forward pays off once xw + aggr (2 * N * H * Cp * 4 bytes) no longer fit the 256 MiB LLC (measured crossover:
    # B = 8 192 -> 0.59 general vs 0.55 pipelined; B = 16 384 -> 0.44-0.48 vs 0.56-0.64 of the HBM peak); ELL_MIN_NODES overrides
    ELL_MIN_NODES = int(os.environ.get("GLAM_ELL_MIN_NODES", "-1"))
    LLC_BYTES = 256 << 20

    @classmethod
    def wants_ell(cls, N, H, Cp):
        return N >= cls.ELL_MIN_NODES if cls.ELL_MIN_NODES >= 0 else 2 * N * H * Cp * 4 > cls.LLC_BYTES

    def ell(self):
        """``(ell_src, ell_eid)`` int32 ``[N, 4]`` index records of the software-pipelined forward aggregate (``glam_ell_build``),
        or ``None`` when some node has more than 4 incoming edges (one host sync, once per edge list — molecular graphs never
        do; protein contact maps always do and keep the general kernel).  While a stream capture is running the answer must
        already be known (a read-back would invalidate the capture): unknown then means ``None``, the general kernels."""
        if self._ell is False and torch.cuda.is_current_stream_capturing():
            return None
        if self._ell is False:
            self._ell = None
            if self.N > 0:
                lib = _lib.load()
                i32 = dict(dtype=torch.int32, device=self.device)
                es, ee, ovf = torch.empty(self.N, 4, **i32), torch.empty(self.N, 4, **i32), torch.zeros(1, **i32)
                check(lib.glam_ell_build(ptr(self.rowptr), ptr(self.src), ptr(self.eid), self.N, ptr(es), ptr(ee), ptr(ovf), stream()),
                      "glam_ell_build")
                if int(ovf.item()) == 0:
                    self._ell = (es, ee)
        return self._ell

    def ell_t(self):
        """``(ell_dst, ell_eid_t)`` int32 ``[N, 4]``: the ELL records BY SOURCE of the software-pipelined backward B2 (``glam_ell_build`` on
        the CSR transpose), or ``None`` when some node has more than 4 outgoing edges (one host sync, once per edge list; inside a
        stream capture an answer not known yet is ``None``)."""
        if self._ell_t is False and torch.cuda.is_current_stream_capturing():
            return None
        if self._ell_t is False:
            self._ell_t = None
            if self.N > 0:
                lib = _lib.load()
                colptr, dst, eid_t = self.transpose()
                i32 = dict(dtype=torch.int32, device=self.device)
                es, ee, ovf = torch.empty(self.N, 4, **i32), torch.empty(self.N, 4, **i32), torch.zeros(1, **i32)
                check(lib.glam_ell_build(ptr(colptr), ptr(dst), ptr(eid_t), self.N, ptr(es), ptr(ee), ptr(ovf), stream()), "glam_ell_build(T)")
                if int(ovf.item()) == 0:
                    self._ell_t = (es, ee)
        return self._ell_t

    def transpose(self):
        """CSR by source (built on first backward)."""
        if self._t is None:
            lib = _lib.load()
            i32 = dict(dtype=torch.int32, device=self.device)
            colptr, dst, eid_t = torch.empty(self.N + 1, **i32), torch.empty(self.E, **i32), torch.empty(self.E, **i32)
            ws = torch.empty(lib.glam_csr_workspace_bytes(self.N, self.E), dtype=torch.uint8, device=self.device)
            ei = self._ei_ref()
            if ei is None:     # the caller dropped edge_index before the first backward: the by-target CSR holds the same edges
                deg = (self.rowptr[1:] - self.rowptr[:-1]).long()
                dst64 = torch.repeat_interleave(torch.arange(self.N, device=self.device), deg, output_size=self.E)
                ei = torch.empty(2, self.E, dtype=torch.int64, device=self.device)
                ei[0, self.eid.long()] = self.src.long()
                ei[1, self.eid.long()] = dst64
            check(lib.glam_csr_build(ptr(ei), self.N, self.E, 1, ptr(colptr), ptr(dst), ptr(eid_t),
                                     ptr(self._err), ptr(ws), ws.numel(), stream()), "glam_csr_build(T)")
            self._t = (colptr, dst, eid_t)
        return self._t


# Whole-layer route for molecular graphs (ELL form, one-hot bond features of width 4): "auto" = the warp-specialised kernels
# (csrc/triplet_ws.hip, csrc/triplet_ws_b1.hip) wherever they exist, at every batch size; "0" = the general kernels (A/B knob of the
# tests: the two routes are pinned to each other).
WS_ROUTE = os.environ.get("GLAM_WS_ROUTE", "auto")


# --------------------------------------------------------------------------------------
# per-forward reuse of staged weights
# --------------------------------------------------------------------------------------
class _WeightScope:
    """Weight re-layouts (GEMM images, the TripletMessage staging buffer) made during ONE model forward, and the
    transposed images made by its backward.  A MessageBlock is applied ``message_steps`` times per forward with the same
    parameters (src_1gp/model.py:53-54), so each re-layout is built once per pass instead of once per application.  The
    scope only lives for one forward (and is kept alive by the autograd nodes for the matching backward): parameters cannot
    change inside it, so nothing can go stale."""

    def __init__(self):
        self.fwd, self.bwd = {}, {}


_SCOPE = None


@contextlib.contextmanager
def weight_scope():
    """``with ops.weight_scope():`` around a model forward (``glam_amd.model.Architecture`` does it)."""
    global _SCOPE
    prev, _SCOPE = _SCOPE, _WeightScope()
    scope = _SCOPE
    try:
        yield
    finally:
        _SCOPE = prev
        # The forward table is only read while the forward runs.  Some of its entries carry autograd history that leads back
        # to nodes holding this scope (for their backward images): dropping the table here leaves no reference cycle, so the
        # pass's activations are released when its backward finishes, not whenever Python's cycle collector next runs (a
        # collection in the middle of a later hipGraph capture brings the capture down).
        scope.fwd.clear()


def scoped_weights(key, owner, build):
    """Derived weights (pure functions of parameters: pads, stacks, small matmuls) built once per model forward when a
    ``weight_scope`` is active; their autograd subgraph is then also shared by the applications of the block."""
    return _scoped(_SCOPE.fwd if _SCOPE else None, key, owner, build)


def _scoped(table, key, owner, build):
    """``build()`` once per (scope table, key); ``owner`` is pinned next to the value so ``id(owner)`` stays unique."""
    if table is None:
        return build()
    hit = table.get(key)
    if hit is not None and hit[0] is owner:
        return hit[1]
    val = build()
    table[key] = (owner, val)
    return val

# ---- zero-padded column layout of odd-width features (hid_dim 15 / 30 / 45 / 90: C % 4 != 0) ----------------------
# The kernels work on rows of Cp = ceil4(C) floats.  An op that produced a padded [N, Cp] tensor whose pad columns are
# zero hands the caller the [N, C] VIEW of it (slice_cols) and remembers the padded tensor; when that view comes back as
# the input of the next op (conv -> GRU -> next message step), pad_cols returns the padded tensor itself instead of
# copying: the per-step pad / slice glue of the odd widths disappears without changing any module interface.
_PADDED: dict = {}      # data_ptr -> (weakref(padded tensor), its version counter at registration)


def slice_cols(x_p, C):
    """``x_p[:, :C]`` of a padded tensor whose columns ``C..`` are zero (the caller guarantees it)."""
    if x_p.size(1) == C:
        return x_p
    key = x_p.data_ptr()

    def _drop(ref, k=key):
        hit = _PADDED.get(k)
        if hit is not None and hit[0] is ref:
            _PADDED.pop(k, None)
    _PADDED[key] = (weakref.ref(x_p, _drop), x_p._version)
    return x_p[:, :C]


def pad_cols(x, Cp):
    """``x[N, C]`` zero-padded to ``Cp`` columns: the registered padded tensor when ``x`` is its untouched view, else one
    ``F.pad`` per tensor and model pass."""
    if x.size(1) == Cp:
        return x
    hit = _PADDED.get(x.data_ptr())
    if hit is not None:
        base = hit[0]()
        if base is not None and base._version == hit[1] and x.dim() == 2 and base.shape == (x.size(0), Cp) and \
                x.stride() == (Cp, 1) and base.data_ptr() == x.data_ptr() and base.dtype == x.dtype:
            return base
    C = x.size(1)
    if not x.requires_grad and x.grad_fn is None:
        # a DATA tensor (e.g. the atom features x[N, 15] of a cached loader batch): padded once per tensor, not once per pass
        key = id(x)
        hit = _PAD_DATA.get(key)
        if hit is not None and hit[0]() is x and hit[1] == x._version and hit[2].size(1) == Cp:
            return hit[2]
        padded = torch.nn.functional.pad(x, (0, Cp - C))
        try:
            _PAD_DATA[key] = (weakref.ref(x, lambda _r, k=key: _PAD_DATA.pop(k, None)), x._version, padded)
        except TypeError:
            pass
        return padded
    return _scoped(_SCOPE.fwd if _SCOPE else None, ("pad-cols", id(x), Cp), x, lambda: torch.nn.functional.pad(x, (0, Cp - C)))


_PAD_DATA: dict = {}


_GI_CACHE: dict = {}


def graph_index(edge_index, num_nodes):
    """Cached ``GraphIndex`` for this very tensor object (dropped when the tensor dies or is
    modified in place)."""
    key = id(edge_index)
    hit = _GI_CACHE.get(key)
    if hit is not None:
        ref, version, n, gi = hit
        if ref() is edge_index and version == edge_index._version and n == int(num_nodes):
            return gi
    gi = GraphIndex(edge_index, num_nodes)
    try:
        ref = weakref.ref(edge_index, lambda _r, k=key, c=_GI_CACHE: c.pop(k, None))
        _GI_CACHE[key] = (ref, edge_index._version, int(num_nodes), gi)
    except TypeError:
        pass
    return gi


class SegmentPtr:
    """``ptr[B+1]`` (int32) of a sorted ``batch`` vector; ``num_graphs`` optional — when absent it
    is read back from ``batch[-1]`` exactly like PyG's ``int(batch.max()) + 1`` (one host sync)."""

    def __init__(self, batch, num_graphs=None, validate=True):
        require_device(batch)
        if batch.dtype != torch.int64 or batch.dim() != 1:
            raise GlamHipError("batch must be an int64 vector")
        self.N = int(batch.numel())
        if num_graphs is None:
            num_graphs = int(batch[-1].item()) + 1 if self.N > 0 else 0
        self.B = int(num_graphs)
        if self.B < 0 or self.B >= 2 ** 31 - 1:      # an unchecked batch[-1] (negative / huge id) must not size the ptr buffer
            raise IndexError(f"batch must be non-decreasing with ids in [0, num_graphs) (num_graphs = {self.B})")
        self.ptr = torch.empty(self.B + 1, dtype=torch.int32, device=batch.device)
        err = torch.zeros(1, dtype=torch.int32, device=batch.device)
        check(_lib.load().glam_batch_ptr(ptr(batch.contiguous()), self.N, self.B, ptr(self.ptr), ptr(err), stream()),
              "glam_batch_ptr")
        poll_checks()
        if validate and getattr(batch, "_glam_trusted", None) != batch._version:
            _check_flag(err, "batch must be non-decreasing with ids in [0, num_graphs)")


_SP_CACHE: dict = {}


def segment_ptr(batch, num_graphs=None):
    key = id(batch)
    hit = _SP_CACHE.get(key)
    if hit is not None:
        ref, version, sp = hit
        if ref() is batch and version == batch._version and (num_graphs is None or sp.B == int(num_graphs)):
            return sp
    sp = SegmentPtr(batch, num_graphs)
    try:
        ref = weakref.ref(batch, lambda _r, k=key, c=_SP_CACHE: c.pop(k, None))
        _SP_CACHE[key] = (ref, batch._version, sp)
    except TypeError:
        pass
    return sp


# --------------------------------------------------------------------------------------
# fused gather / attention softmax / scatter-add  (TripletMessage, TripletMessageLight)
# --------------------------------------------------------------------------------------
class _TripletAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xw, a_ij, edge_attr, w_edge, M, gi, H, Cp, De, emul, slope):
        require_device(xw, a_ij, edge_attr, w_edge, M)
        xw, a_ij, edge_attr, M = f32c(xw, "xw"), f32c(a_ij, "a_ij"), f32c(edge_attr, "edge_attr"), f32c(M, "M")
        w_edge = f32c(w_edge, "w_edge") if emul else None
        N, E = gi.N, gi.E
        if xw.shape != (N, H * Cp) or a_ij.shape != (N, 8) or edge_attr.shape != (E, De) or M.shape != (De, 4):
            raise GlamHipError(f"triplet_aggregate: shape mismatch xw={tuple(xw.shape)} a_ij={tuple(a_ij.shape)} "
                               f"edge_attr={tuple(edge_attr.shape)} M={tuple(M.shape)} for N={N} E={E} H={H} Cp={Cp} De={De}")
        aggr = torch.empty(N, H * Cp, dtype=torch.float32, device=xw.device)
        stats = torch.empty(N, 8, dtype=torch.float32, device=xw.device)
        lib = _lib.load()
        # Batches whose working set leaves the 256 MiB LLC: the software-pipelined kernel (bit-identical; 0.56 vs 0.44 of the HBM
        # peak at B = 16 384).  Below that the general kernel's three waves per SIMD win (9.4 vs 10.4 us at B = 1 024).
        ell = gi.ell() if (emul and GraphIndex.wants_ell(N, H, Cp) and lib.glam_triplet_fwd_ell_supported(H, Cp, De)) else None
        if ell is not None:
            check(lib.glam_triplet_fwd_ell(ptr(xw), ptr(a_ij), ptr(edge_attr), ptr(w_edge), ptr(M), ptr(ell[0]), ptr(ell[1]), N, E, H, Cp, De,
                                           float(slope), int(rows_are_one_hot(edge_attr)), ptr(aggr), ptr(stats), 0, stream()),
                  "glam_triplet_fwd_ell")
        else:
            check(lib.glam_triplet_fwd(ptr(xw), ptr(a_ij), ptr(edge_attr), ptr(w_edge), ptr(M), ptr(gi.rowptr),
                                       ptr(gi.src), ptr(gi.eid), N, E, H, Cp, De, int(emul), float(slope),
                                       ptr(aggr), ptr(stats), stream()), "glam_triplet_fwd")
        ctx.save_for_backward(xw, a_ij, edge_attr, w_edge, M, aggr, stats)
        ctx.gi, ctx.dims = gi, (H, Cp, De, int(emul), float(slope))
        return aggr

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_aggr):
        xw, a_ij, edge_attr, w_edge, M, aggr, stats = ctx.saved_tensors
        gi = ctx.gi
        H, Cp, De, emul, slope = ctx.dims
        N, E = gi.N, gi.E
        lib = _lib.load()
        d_aggr = f32c(d_aggr, "d_aggr")
        colptr, dst, eid_t = gi.transpose()
        dev = xw.device
        d_xw = torch.empty_like(xw)
        d_a_ij = torch.empty_like(a_ij)
        d_w_edge = torch.empty_like(w_edge) if emul else None
        d_M = torch.empty_like(M)
        d_ea = torch.zeros_like(edge_attr) if ctx.needs_input_grad[2] else None
        ws = torch.empty(lib.glam_triplet_bwd_workspace_bytes(N, E, H, Cp, De), dtype=torch.uint8, device=dev)
        check(lib.glam_triplet_bwd(ptr(xw), ptr(a_ij), ptr(edge_attr), ptr(w_edge), ptr(M), ptr(aggr), ptr(stats),
                                   ptr(d_aggr), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), ptr(colptr), ptr(dst),
                                   ptr(eid_t), N, E, H, Cp, De, emul, slope, ptr(d_xw), ptr(d_a_ij), ptr(d_w_edge),
                                   ptr(d_M), ptr(d_ea), ptr(ws), ws.numel(), stream()), "glam_triplet_bwd")
        return d_xw, d_a_ij, d_ea, d_w_edge, d_M, None, None, None, None, None, None


def triplet_aggregate(xw, a_ij, edge_attr, w_edge, M, gi, heads, Cp, slope=0.2):
    """``aggr[N, H*Cp]`` of TripletMessage (see ``glam_triplet_fwd`` in include/glam_hip.h)."""
    return _TripletAggregate.apply(xw, a_ij, edge_attr, w_edge, M, gi, heads, Cp, edge_attr.size(1), True, slope)


def light_aggregate(xw, a_ij, edge_attr, M, gi, Cp, slope=0.2):
    """``aggr[N, Cp]`` of TripletMessageLight (single head, message ``alpha * x_j``)."""
    return _TripletAggregate.apply(xw, a_ij, edge_attr, None, M, gi, 1, Cp, edge_attr.size(1), False, slope)


def fused_layer_supported(C, heads, De):
    """Shapes covered by the dense MFMA kernels behind ``glam_triplet_layer_*`` (C <= 60 at 3 heads)."""
    Cp = (C + 3) // 4 * 4
    return heads * Cp + 8 <= 192 and Cp <= 64 and De <= 8 and 1 <= heads <= 4


# ---- gradient carry of a block's parameters ---------------------------------------------------------------------------
# A MessageBlock applies the SAME parameters message_steps times per forward (src_1gp/model.py:53-54), so autograd would sum
# message_steps gradients per parameter tensor: 17 small add kernels per training step of the default model.  Inside a
# weight_scope the parameters instead enter the graph once, through a `_ParamBundle` node whose output (an uninitialised
# flat tensor — only its GRADIENT matters) is threaded through the applications as an extra input / output: application k's
# backward receives the gradients accumulated by the later applications, adds its own flat gradient buffer (one add), and
# hands the sum on; the bundle's backward splits the total into per-parameter views.  Works for .backward() and
# autograd.grad alike; outside a scope (or without grad) the ops return per-parameter gradients as before.
GRAD_CARRY = os.environ.get("GLAM_GRAD_CARRY", "1") == "1"


class _ParamBundle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, split, total, *params):
        ctx.split = split
        return torch.empty(total, dtype=torch.float32, device=params[0].device)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_flat):
        return (None, None) + tuple(ctx.split(d_flat))


def _carry_for(key, params, total, split):
    """The scope's carry tensor for this parameter set (created on the block's first application of the pass), or None
    when gradients are off / there is no scope."""
    scope = _SCOPE
    if scope is None or not GRAD_CARRY or not torch.is_grad_enabled() or not any(p.requires_grad for p in params):
        return None
    hit = scope.fwd.get(key)
    if hit is not None and hit[0] is params[0]:
        return hit[1]
    return _ParamBundle.apply(split, total, *params)


def _carry_store(key, owner, carry):
    if _SCOPE is not None and carry is not None:
        _SCOPE.fwd[key] = (owner, carry)


# ---- staged parameter images across passes -------------------------------------------------------------------------------
# The weight re-layout of a TripletMessage (k_stage_params: four GEMM images, W_edge, M, bias) depends on the parameters only.  By
# default it is rebuilt in every pass (every model forward inside a weight_scope): always right, also when a captured optimizer
# launch rewrites the parameters between replays, which no host-side bookkeeping can see.  ``with ops.cached_staging():`` is the
# caller's statement that parameters are written only through torch (in-place ops bump the version counter; ``.data`` swaps change
# the address) and NOT by kernels replayed from a graph that also holds this forward: the images are then kept until a parameter's
# (address, version) changes — a forward-backward step whose parameters stand still (bench.py's configs[1] step: no optimizer in it)
# starts at the node GEMM.
# Writers the host cannot see through (address, version counter) — ``glam_amd.optim.Adam`` updates parameters through raw device
# pointers — announce themselves by bumping PARAM_EPOCH, which is part of every entry's stamp.
CACHED_STAGING = False
_STAGED: dict = {}
PARAM_EPOCH = 0


def parameters_written():
    """Tell the staging cache that parameters were (or will be, by a launch just enqueued) written outside torch's version counting."""
    global PARAM_EPOCH
    PARAM_EPOCH += 1


@contextlib.contextmanager
def cached_staging(on=True):
    global CACHED_STAGING
    prev, CACHED_STAGING = CACHED_STAGING, bool(on)
    try:
        yield
    finally:
        CACHED_STAGING = prev


def _staged_cached(kind, params, build):
    key = (kind,) + tuple(id(p) for p in params)
    stamp = (PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in params)
    hit = _STAGED.get(key)
    if hit is not None and hit[0] == stamp and all(r() is p for r, p in zip(hit[1], params)):
        return hit[2]
    val = build()
    if torch.cuda.is_current_stream_capturing():
        # built inside a capture: the buffer lives in the graph's pool and is only filled when the graph is replayed — an eager pass
        # before that would read uninitialised images, so it is never published to the cache
        return val
    try:
        refs = tuple(weakref.ref(p, lambda _r, k=key, c=_STAGED: c.pop(k, None)) for p in params)
    except TypeError:
        return val
    _STAGED[key] = (stamp, refs, val)
    return val


def _ws_route(lib, N, H, Cp, Dp, ea_p):
    """The warp-specialised kernels cover this layer call (shape table in the library + one-hot rows: one cached read-back per tensor)."""
    return WS_ROUTE != "0" and N > 0 and lib.glam_triplet_layer_ws_supported(H, Cp, Dp, 1) == 1 and rows_are_one_hot(ea_p)


class _TripletLayer(torch.autograd.Function):
    """Whole TripletMessage layer: parameter staging, node GEMM (+ separable attention columns), fused
    gather/softmax/scatter-add, update GEMM — and the hand-written backward of all of it."""

    @staticmethod
    def forward(ctx, x_p, ea_p, wn, we, att, wsc, bias, gi, H, slope, carry=None):
        require_device(x_p, ea_p, wn, we, att, wsc, bias)
        x_p, ea_p = f32c(x_p, "x"), f32c(ea_p, "edge_attr")
        wn, we, att, wsc, bias = (f32c(t, n) for t, n in ((wn, "weight_node"), (we, "weight_edge"),
                                                           (att, "weight_triplet_att"), (wsc, "weight_scale"), (bias, "bias")))
        C, De = wn.size(0), we.size(0)
        N, Cp = x_p.shape
        Dp = ea_p.size(1)
        if gi.N != N or ea_p.size(0) != gi.E or wn.shape != (C, H * C) or wsc.shape != (H * C, C) or Cp != (C + 3) // 4 * 4:
            raise GlamHipError("triplet_layer: shape mismatch")
        lib, dev = _lib.load(), x_p.device
        HC = H * Cp
        f = dict(dtype=torch.float32, device=dev)
        ctx.carried = carry is not None
        ctx.set_materialize_grads(False)     # the carry of the block's LAST application has no gradient yet: None, not a zero fill
        def build():
            buf = torch.empty(lib.glam_triplet_staged_floats(H, Cp, Dp), **f)
            check(lib.glam_triplet_stage_params(ptr(wn), ptr(we), ptr(att), ptr(wsc), ptr(bias), C, H, De, Cp, Dp, ptr(buf),
                                                stream()), "glam_triplet_stage_params")
            return buf

        # the same conv is applied message_steps times per model forward: one staging per pass (see _WeightScope); with
        # ops.cached_staging() the staged images additionally survive from pass to pass until a parameter is written
        staged = _staged_cached(("triplet", H, Cp, Dp), (wn, we, att, wsc, bias), build) if CACHED_STAGING else \
            _scoped(_SCOPE.fwd if _SCOPE else None, ("triplet", id(wn), id(we), id(att), id(wsc), id(bias)), wn, build)
        xw, a_ij = torch.empty(N, HC, **f), torch.empty(N, 8, **f)
        aggr, stats, out = torch.empty(N, HC, **f), torch.empty(N, 8, **f), torch.empty(N, Cp, **f)
        # molecular graphs with one-hot bond features take the warp-specialised kernels at every size (13.6 vs 16.8 us at B = 1 024,
        # 136 vs 256 us at B = 16 384 against the general fused kernel); everything else the general kernels
        ell = gi.ell() if _ws_route(lib, N, H, Cp, Dp, ea_p) else None
        if ell is not None:
            check(lib.glam_triplet_layer_fwd_ell(ptr(x_p), ptr(ea_p), ptr(staged), ptr(ell[0]), ptr(ell[1]), 1, N,
                                                 gi.E, H, Cp, Dp, float(slope), ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats), ptr(out), stream()),
                  "glam_triplet_layer_fwd_ell")
        else:
            check(lib.glam_triplet_layer_fwd(ptr(x_p), ptr(ea_p), ptr(staged), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid),
                                             N, gi.E, H, Cp, Dp, float(slope), ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats), ptr(out),
                                             stream()), "glam_triplet_layer_fwd")
        ctx.save_for_backward(x_p, ea_p, wn, we, att, staged, xw, a_ij, aggr, stats)
        ctx.gi, ctx.dims = gi, (C, H, De, Cp, Dp, float(slope))
        return (out, carry.view(-1)) if ctx.carried else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_carry=None):
        if d_out is None:                    # the layer's output was not used: only the carry (if any) passes through
            return (None,) * 10 + (d_carry,)
        x_p, ea_p, wn, we, att, staged, xw, a_ij, aggr, stats = ctx.saved_tensors
        C, H, De, Cp, Dp, slope = ctx.dims
        gi = ctx.gi
        N, E = gi.N, gi.E
        lib, dev = _lib.load(), x_p.device
        d_out = f32c(d_out, "d_out")
        colptr, dst, eid_t = gi.transpose()
        f = dict(dtype=torch.float32, device=dev)
        d_x = torch.empty_like(x_p)
        d_ea = torch.zeros_like(ea_p) if ctx.needs_input_grad[1] else None
        ws = torch.empty(lib.glam_triplet_layer_bwd_workspace_bytes(N, E, H, Cp, Dp), dtype=torch.uint8, device=dev)
        # the five parameter gradients are consecutive views of ONE buffer (parameter order), so a data-parallel
        # step can all-reduce them as a single bucket without a gather copy (parallel.flat_view)
        sizes = [wn.numel(), we.numel(), att.numel(), H * C * C, C]
        flatg = torch.empty(sum(sizes), **f)
        d_wn, d_we, d_att, d_wsc, d_bias = (t.view(s) for t, s in zip(flatg.split(sizes), (wn.shape, we.shape, att.shape, (H * C, C), (C,))))
        # molecular graphs with one-hot bond features: B1 and B2 + d_x warp-specialised over the ELL records of both directions
        ell_t = gi.ell_t() if (d_ea is None and _ws_route(lib, N, H, Cp, Dp, ea_p)) else None
        have_carry = ctx.carried and d_carry is not None and N > 0
        ell_f = gi.ell() if ell_t is not None else None          # (by target: what the forward used)
        if have_carry or ell_t is not None:
            # the gradient accumulated by the later applications of the block is summed by k_param_grads itself
            c_parts = f32c(d_carry, "d_carry").split(sizes) if have_carry else (None,) * 5
            c_wn, c_we, c_att, c_wsc, c_bias = c_parts
            check(lib.glam_triplet_layer_bwd_params_ell(ptr(x_p), ptr(ea_p), ptr(staged), ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats),
                                                        ptr(d_out), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), ptr(colptr), ptr(dst),
                                                        ptr(eid_t), N, E, C, H, De, Cp, Dp, slope, ptr(wn), ptr(we), ptr(att), ptr(d_x),
                                                        ptr(d_wn), ptr(d_we), ptr(d_att), ptr(d_wsc), ptr(d_bias), ptr(c_wn), ptr(c_we),
                                                        ptr(c_att), ptr(c_wsc), ptr(c_bias), ptr(ell_f[0]) if ell_f else None,
                                                        ptr(ell_f[1]) if ell_f else None, ptr(ell_t[0]) if ell_t else None,
                                                        ptr(ell_t[1]) if ell_t else None, 1 if ell_t else 0,
                                                        ptr(d_ea), ptr(ws), ws.numel(), stream()),
                  "glam_triplet_layer_bwd_params_ell")
            if ctx.carried:
                return d_x, d_ea, None, None, None, None, None, None, None, None, (flatg if (have_carry or d_carry is None) else flatg.add_(d_carry))
            return d_x, d_ea, d_wn, d_we, d_att, d_wsc, d_bias, None, None, None, None
        check(lib.glam_triplet_layer_bwd_params(ptr(x_p), ptr(ea_p), ptr(staged), ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats),
                                                ptr(d_out), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), ptr(colptr), ptr(dst),
                                                ptr(eid_t), N, E, C, H, De, Cp, Dp, slope, ptr(wn), ptr(we), ptr(att), ptr(d_x),
                                                ptr(d_wn), ptr(d_we), ptr(d_att), ptr(d_wsc), ptr(d_bias), ptr(d_ea), ptr(ws),
                                                ws.numel(), stream()), "glam_triplet_layer_bwd_params")
        if ctx.carried:
            return d_x, d_ea, None, None, None, None, None, None, None, None, (flatg if d_carry is None else flatg.add_(d_carry))
        return d_x, d_ea, d_wn, d_we, d_att, d_wsc, d_bias, None, None, None, None


# The layer through the torch-extension operator (torch.ops.glam.triplet_layer: C++ autograd node, no ctypes marshalling, no Python
# autograd.Function) is the route for EAGERLY issued steps, which are bound by host time: full-model step issued eagerly 1.21 vs 1.48 ms
# at B = 32, 1.32 vs 1.55 ms at B = 1 024.  The Python node keeps the per-pass weight staging and the gradient carry of a weight_scope,
# which a captured hipGraph replays for free.  GLAM_TORCH_EXT: "auto" (default) = the operator while nothing is being captured and the
# batch is cache resident, "1" / "0" = always / never.  Both routes give
# the same numbers bit for bit (tested), so an eager first visit and a captured second one stay on one trajectory.
_ext_env = os.environ.get("GLAM_TORCH_EXT", "auto")
USE_TORCH_EXT = True if _ext_env == "1" else False if _ext_env == "0" else "auto"
_EXT_OK = None


def _want_torch_ext(N, H, Cp):
    global _EXT_OK
    if USE_TORCH_EXT is False or N <= 0 or CACHED_STAGING:     # (the C++ node stages per call)
        return False
    if USE_TORCH_EXT == "auto":
        if torch.cuda.is_current_stream_capturing() or GraphIndex.wants_ell(N, H, Cp):
            return False
        if _EXT_OK is None and os.environ.get("GLAM_HIP_LIB"):
            _EXT_OK = False            # the shim is linked against the in-tree libglam_hip.so, not against a substituted build
        if _EXT_OK is None:
            try:
                from . import torch_ext
                torch_ext.load()
                _EXT_OK = True
            except Exception:          # noqa: BLE001 - the shim is optional in auto mode: the Python node is the same HIP path
                _EXT_OK = False
        return _EXT_OK
    return True


def triplet_layer(x_p, ea_p, weight_node, weight_edge, att, weight_scale, bias, gi, heads, slope=0.2):
    """``TripletMessage.forward`` (src_1gp/layer.py:36-61) in padded widths: ``x_p[N,Cp] -> out[N,Cp]``."""
    if _want_torch_ext(gi.N, heads, x_p.size(1)):
        from . import torch_ext
        # the one-time read-backs of the ELL routes happen on an eager visit (a later CAPTURED visit of the same batch finds them
        # cached); with them the C++ node launches the same warp-specialised kernels as the Python node: one trajectory, bit for bit
        ell_f = ell_b = None
        onehot = False
        if _ws_route(_lib.load(), gi.N, heads, x_p.size(1), ea_p.size(1), ea_p):
            onehot = True
            ell_f = gi.ell()
            if torch.is_grad_enabled():
                ell_b = gi.ell_t()
        # same checks, same exception type as the Python node (the operator's own TORCH_CHECKs would raise RuntimeError)
        require_device(x_p, ea_p, weight_node, weight_edge, att, weight_scale, bias)
        C = weight_node.size(0)
        if (x_p.dim() != 2 or ea_p.dim() != 2 or gi.N != x_p.size(0) or ea_p.size(0) != gi.E or weight_node.shape != (C, heads * C)
                or weight_scale.shape != (heads * C, C) or x_p.size(1) != (C + 3) // 4 * 4):
            raise GlamHipError("triplet_layer: shape mismatch")
        t_ptr = gi.transpose() if torch.is_grad_enabled() else (gi.rowptr, gi.src, gi.eid)     # any int32 tensors when no backward follows
        return torch_ext.load().triplet_layer(f32c(x_p, "x"), f32c(ea_p, "edge_attr"), weight_node, weight_edge, att, weight_scale, bias,
                                              gi.rowptr, gi.src, gi.eid, t_ptr[0], t_ptr[1], t_ptr[2], heads, float(slope),
                                              ell_f[0] if ell_f else None, ell_f[1] if ell_f else None,
                                              ell_b[0] if ell_b else None, ell_b[1] if ell_b else None, onehot)
    params = (weight_node, weight_edge, att, weight_scale, bias)
    C = weight_node.size(0)
    sizes = [weight_node.numel(), weight_edge.numel(), att.numel(), heads * C * C, C]
    shapes = (weight_node.shape, weight_edge.shape, att.shape, (heads * C, C), (C,))
    key = ("carry-triplet", id(weight_node))
    carry = _carry_for(key, params, sum(sizes), lambda flat: [t.view(sh) for t, sh in zip(flat.split(sizes), shapes)])
    if carry is None:
        return _TripletLayer.apply(x_p, ea_p, weight_node, weight_edge, att, weight_scale, bias, gi, heads, slope)
    # the parameters still enter as inputs (the kernels read them, and the scope's staging cache is keyed on them), but this
    # node returns no gradient for them: it flows through `carry`
    out, carry = _TripletLayer.apply(x_p, ea_p, weight_node, weight_edge, att, weight_scale, bias, gi, heads, slope, carry)
    _carry_store(key, weight_node, carry)
    return out


class _TripletLayerWide(torch.autograd.Function):
    """TripletMessage for widths beyond the fused kernels' table (H*Cp + 8 > 192; hid_dim_alpha = 6 of glam.py:60) as ONE
    autograd node: ``k_stage_plain`` -> library GEMMs for the data-side products -> the aggregate kernels -> ``k_wgrad``
    for both N-deep weight gradients (written straight into the ``dstaged`` layout) -> ``k_stage_params_bwd``.
    No per-parameter torch glue on either pass."""

    @staticmethod
    def forward(ctx, x_p, ea_p, wn, we, att, wsc, bias, gi, H, slope):
        require_device(x_p, ea_p, wn, we, att, wsc, bias)
        x_p, ea_p = f32c(x_p, "x"), f32c(ea_p, "edge_attr")
        wn, we, att, wsc, bias = (f32c(t, n) for t, n in ((wn, "weight_node"), (we, "weight_edge"),
                                                           (att, "weight_triplet_att"), (wsc, "weight_scale"), (bias, "bias")))
        C, De = wn.size(0), we.size(0)
        N, Cp = x_p.shape
        Dp = ea_p.size(1)
        HC = H * Cp
        if gi.N != N or ea_p.size(0) != gi.E or wn.shape != (C, H * C) or wsc.shape != (H * C, C) or Cp != (C + 3) // 4 * 4:
            raise GlamHipError("triplet_layer_wide: shape mismatch")
        lib, dev = _lib.load(), x_p.device
        f = dict(dtype=torch.float32, device=dev)

        def build():
            buf = torch.empty(lib.glam_triplet_plain_floats(H, Cp, Dp), **f)
            check(lib.glam_triplet_stage_plain(ptr(wn), ptr(we), ptr(att), ptr(wsc), ptr(bias), C, H, De, Cp, Dp, ptr(buf),
                                               stream()), "glam_triplet_stage_plain")
            return buf

        scope = _SCOPE
        plain = _scoped(scope.fwd if scope else None, ("triplet-plain", id(wn), id(we), id(att), id(wsc), id(bias)), wn, build)
        Wcat, Ws_p, We_p, M, bias_p = _plain_views(plain, H, Cp, Dp)
        xw, a_ij = torch.empty(N, HC, **f), torch.empty(N, 8, **f)
        aggr, stats, out = torch.empty(N, HC, **f), torch.empty(N, 8, **f), torch.empty(N, Cp, **f)
        mfma = _wide_gemms_supported(H, Cp)
        st = stream()
        if mfma:     # the 120 KB-image k_ts_gemm variant: xw and a_ij in one launch
            img1 = _scoped(scope.fwd if scope else None, ("wide-img-node", id(wn)), wn, lambda: _ts_image(Wcat, Cp, HC + 8, False))
            check(lib.glam_ts_gemm(ptr(x_p), Cp, Cp, None, 0, 0, ptr(img1), None, ptr(xw), HC, HC, ptr(a_ij), 8, 8, N, st), "glam_ts_gemm")
        else:
            torch.matmul(x_p, Wcat[:, :HC], out=xw)                        # layer.py:37
            torch.matmul(x_p, Wcat[:, HC:], out=a_ij)                      # separable attention scalars a_i | a_j
        check(lib.glam_triplet_fwd(ptr(xw), ptr(a_ij), ptr(ea_p), ptr(We_p), ptr(M), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid),
                                   N, gi.E, H, Cp, Dp, 1, float(slope), ptr(aggr), ptr(stats), st), "glam_triplet_fwd")
        torch.addmm(bias_p, aggr, Ws_p, out=out)                           # layer.py:57-61 (276 -> 92: the library GEMM wins)
        ctx.scope = scope
        ctx.save_for_backward(x_p, ea_p, wn, we, att, plain, xw, a_ij, aggr, stats)
        ctx.gi, ctx.dims = gi, (C, H, De, Cp, Dp, float(slope))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        x_p, ea_p, wn, we, att, plain, xw, a_ij, aggr, stats = ctx.saved_tensors
        C, H, De, Cp, Dp, slope = ctx.dims
        gi = ctx.gi
        N, E, HC = gi.N, gi.E, H * Cp
        lib, dev = _lib.load(), x_p.device
        f = dict(dtype=torch.float32, device=dev)
        d_out = f32c(d_out, "d_out")
        if N == 0:        # an empty batch: every gradient is zero
            z = lambda t: torch.zeros_like(t)
            return (torch.zeros_like(x_p), torch.zeros_like(ea_p) if ctx.needs_input_grad[1] else None, z(wn), z(we), z(att),
                    torch.zeros(H * C, C, **f), torch.zeros(C, **f), None, None, None)
        Wcat, Ws_p, We_p, M, _ = _plain_views(plain, H, Cp, Dp)
        colptr, dst, eid_t = gi.transpose()
        # dstaged: d_Wcat[Cp, HC+8] | d_WsB[HC+1, Cp] | d_We_p[Dp, HC] | d_M[Dp, 4]   (include/glam_hip.h)
        o_wsb = Cp * (HC + 8)
        o_we = (o_wsb + (HC + 1) * Cp + 3) // 4 * 4
        o_m = o_we + Dp * HC
        dstaged = torch.empty(lib.glam_triplet_dstaged_floats(H, Cp, Dp), **f)
        mfma = _wide_gemms_supported(H, Cp)
        scope = ctx.scope
        if mfma:
            img3 = _scoped(scope.bwd if scope else None, ("wide-img-dagg", id(wn)), wn, lambda: _ts_image(Ws_p, Cp, HC, True))
            d_aggr = torch.empty(N, HC, **f)
            check(lib.glam_ts_gemm(ptr(d_out), Cp, Cp, None, 0, 0, ptr(img3), None, ptr(d_aggr), HC, HC, None, 0, 0, N, stream()), "glam_ts_gemm")
        else:
            d_aggr = torch.matmul(d_out, Ws_p.t())
        d_xw, d_a = torch.empty(N, HC, **f), torch.empty(N, 8, **f)
        d_ea = torch.zeros_like(ea_p) if ctx.needs_input_grad[1] else None
        ws = torch.empty(lib.glam_triplet_bwd_workspace_bytes(N, E, H, Cp, Dp), dtype=torch.uint8, device=dev)
        check(lib.glam_triplet_bwd(ptr(xw), ptr(a_ij), ptr(ea_p), ptr(We_p), ptr(M), ptr(aggr), ptr(stats), ptr(d_aggr),
                                   ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), ptr(colptr), ptr(dst), ptr(eid_t), N, E, H, Cp, Dp, 1,
                                   slope, ptr(d_xw), ptr(d_a), ptr(dstaged[o_we:]), ptr(dstaged[o_m:]), ptr(d_ea), ptr(ws),
                                   ws.numel(), stream()), "glam_triplet_bwd")
        d_x = torch.matmul(d_xw, Wcat[:, :HC].t())
        d_x.addmm_(d_a, Wcat[:, HC:].t())
        wws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        # d_WsB = [aggr | 1]^T d_out ;  d_Wcat = x^T [d_xw | d_a], computed as ([d_xw | d_a]^T x)^T
        check(lib.glam_wgrad_gemm(ptr(aggr), HC, HC, None, 0, 0, 1, ptr(d_out), Cp, Cp, 0, N, ptr(dstaged[o_wsb:]), Cp, 1,
                                  ptr(wws), wws.numel(), stream()), "glam_wgrad_gemm")
        check(lib.glam_wgrad_gemm(ptr(d_xw), HC, HC, ptr(d_a), 8, 8, 0, ptr(x_p), Cp, Cp, 0, N, ptr(dstaged), 1, HC + 8,
                                  ptr(wws), wws.numel(), stream()), "glam_wgrad_gemm")
        sizes = [wn.numel(), we.numel(), att.numel(), H * C * C, C]
        flatg = torch.empty(sum(sizes), **f)
        d_wn, d_we, d_att, d_wsc, d_bias = (t.view(s) for t, s in zip(flatg.split(sizes), (wn.shape, we.shape, att.shape, (H * C, C), (C,))))
        check(lib.glam_triplet_stage_params_bwd(ptr(wn), ptr(we), ptr(att), ptr(dstaged), C, H, De, Cp, Dp, ptr(d_wn), ptr(d_we),
                                                ptr(d_att), ptr(d_wsc), ptr(d_bias), stream()), "glam_triplet_stage_params_bwd")
        return d_x, d_ea, d_wn, d_we, d_att, d_wsc, d_bias, None, None, None


def _wide_gemms_supported(H, Cp):
    return Cp <= 96 and H * Cp + 8 <= 320


def _ts_image(W, K, M, transposed):
    """k_ts_gemm weight image of the logical ``[K, M]`` matrix ``W`` (or ``W^T`` of the stored ``[M, K]`` matrix)."""
    lib = _lib.load()
    img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, dtype=torch.float32, device=W.device)
    check(lib.glam_ts_gemm_make_image(ptr(W), W.stride(0), int(transposed), K, M, ptr(img), stream()), "glam_ts_gemm_make_image")
    return img


def _plain_views(plain, H, Cp, Dp):
    HC = H * Cp
    n1, n2, n3, n4 = Cp * (HC + 8), HC * Cp, Dp * HC, Dp * 4
    return (plain[:n1].view(Cp, HC + 8), plain[n1:n1 + n2].view(HC, Cp), plain[n1 + n2:n1 + n2 + n3].view(Dp, HC),
            plain[n1 + n2 + n3:n1 + n2 + n3 + n4].view(Dp, 4), plain[n1 + n2 + n3 + n4:])


def wide_layer_supported(C, heads, De):
    """Widths the one-node wide path covers (k_wgrad: up to 320 + 128 columns): C <= 100 at 3 heads."""
    Cp = (C + 3) // 4 * 4
    return heads * Cp + 8 <= 320 and Cp <= 128 and De <= 8 and 1 <= heads <= 4


def triplet_layer_wide(x_p, ea_p, weight_node, weight_edge, att, weight_scale, bias, gi, heads, slope=0.2):
    return _TripletLayerWide.apply(x_p, ea_p, weight_node, weight_edge, att, weight_scale, bias, gi, heads, slope)


# --------------------------------------------------------------------------------------
# device-side random stream of the training-mode layers (RReLU slopes, Dropout masks): csrc/rng.h
# --------------------------------------------------------------------------------------
_RNG_STATE: dict = {}
RNG_STATE_WORDS = 288          # include/glam_hip.h: GLAM_RNG_STATE_WORDS


def rng_state(device):
    """``int64[288]`` on ``device``: [0] Philox seed, [1] stream offset (advanced by every RNG-consuming launch, on the device), [16] and
    [32 + 16 s] tickets (include/glam_hip.h: GLAM_RNG_STATE_WORDS).
    Created on first use from ``torch.initial_seed()`` — so the reference's ``seed_torch`` (``utils.py:22-28``) also fixes this
    stream — and OUTSIDE any hipGraph capture (``GraphedTrainStep`` runs the first visit of a batch eagerly)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _RNG_STATE.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise GlamHipError("the RNG state must exist before a hipGraph capture: run one eager training-mode forward first "
                               "(or call glam_amd.ops.manual_seed)")
        st = _RNG_STATE[key] = torch.tensor([torch.initial_seed() & (2 ** 63 - 1)] + [0] * (RNG_STATE_WORDS - 1), dtype=torch.int64, device=device)
    return st


def manual_seed(seed, device=None):
    """Restart the device-side stream of RReLU / Dropout numbers at ``(seed, offset 0)``."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = rng_state(device)
    st.copy_(torch.tensor([int(seed) & (2 ** 63 - 1)] + [0] * (RNG_STATE_WORDS - 1), dtype=torch.int64))
    return st


# A tail kernel that also wrote Dropout(p)(out) — the input of the NEXT message step's conv (layer.py:255-259) — registers the
# pair here; the block's dropout call on that very tensor then returns the twin instead of launching a kernel.
_DROPPED: dict = {}      # data_ptr(out) -> (weakref(out), out._version, out_drop, p)


def register_dropped(out, out_drop, p):
    key = out.data_ptr()

    def _gone(ref, k=key):
        hit = _DROPPED.get(k)
        if hit is not None and hit[0] is ref:
            _DROPPED.pop(k, None)
    _DROPPED[key] = (weakref.ref(out, _gone), out._version, out_drop, float(p))


def take_dropped(x, p):
    hit = _DROPPED.get(x.data_ptr())
    if hit is not None and hit[0]() is x and hit[1] == x._version and hit[3] == float(p):
        _DROPPED.pop(x.data_ptr(), None)
        return hit[2]
    return None


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
    def forward(ctx, x, w, b):
        require_device(x, w, b)
        x, w = f32c(x, "x"), f32c(w, "weight")
        b = None if b is None else f32c(b, "bias")
        N, K = x.shape
        M, Kw = w.shape
        if Kw > K or (Kw < K and ctx.needs_input_grad[0]):
            raise GlamHipError("linear: weight wider than the input / narrow weight with a differentiable input")
        lib, dev = _lib.load(), x.device
        scope = _SCOPE

        def build():
            img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, dtype=torch.float32, device=dev)
            check(lib.glam_ts_gemm_make_image(ptr(w), Kw, 1, Kw, M, ptr(img), stream()), "glam_ts_gemm_make_image")
            return img

        img = _scoped(scope.fwd if scope else None, ("lin", id(w)), w, build)
        y = torch.empty(N, M, dtype=torch.float32, device=dev)
        check(lib.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), ptr(b), ptr(y), M, M, None, 0, 0, N, stream()), "glam_ts_gemm")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.scope = scope
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = f32c(dy, "dy")
        N, K = x.shape
        M = w.size(0)
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        dx = None
        if ctx.needs_input_grad[0]:
            def build():
                img = torch.empty(lib.glam_ts_gemm_image_bytes(M, K) // 4, **f)
                check(lib.glam_ts_gemm_make_image(ptr(w), K, 0, M, K, ptr(img), stream()), "glam_ts_gemm_make_image")
                return img

            img = _scoped(ctx.scope.bwd if ctx.scope else None, ("lin", id(w)), w, build)
            dx = torch.empty(N, K, **f)
            check(lib.glam_ts_gemm(ptr(dy), M, M, None, 0, 0, ptr(img), None, ptr(dx), K, K, None, 0, 0, N, stream()), "glam_ts_gemm")
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        dwb = torch.empty(M + 1, K + 1, **f)          # [d_w | d_b] (+ a spare row / column for the ones trick)
        if M <= 64:   # out[k, m] = sum_n [x|1][n,k] dy[n,m]  ->  written transposed into dwb[m, k]
            check(lib.glam_wgrad_gemm(ptr(x), K, K, None, 0, 0, 1, ptr(dy), M, M, 0, N, ptr(dwb), 1, K + 1, ptr(ws), ws.numel(),
                                      stream()), "glam_wgrad_gemm")
        else:         # out[m, k] = sum_n dy[n,m] [x|1][n,k]
            check(lib.glam_wgrad_gemm(ptr(dy), M, M, None, 0, 0, 0, ptr(x), K, K, 1, N, ptr(dwb), K + 1, 1, ptr(ws), ws.numel(),
                                      stream()), "glam_wgrad_gemm")
        dw = dwb[:M, :w.size(1)]
        db = dwb[:M, K] if ctx.has_bias else None
        return dx, dw, db


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
    def forward(ctx, a, w, bias, carry=None):
        require_device(a, w, bias)
        a, w = f32c(a, "a"), f32c(w, "w")
        ctx.save_for_backward(a, w)
        ctx.has_bias = bias is not None
        ctx.scope = _SCOPE
        ctx.carried = carry is not None
        if ctx.carried:
            ctx.set_materialize_grads(False)     # the carry of the LAST application has no gradient yet: None, not a zero fill
        # (an 80 KB-image k_ts_gemm<4, 20, 4> for K <= 320 was measured here — NNConv's [N, 300] x [300, 60] relation product —: 18.9 us
        # against the library's 15 at N = 20 k, at 256 registers: not kept)
        out = torch.matmul(a, w) if bias is None else torch.addmm(f32c(bias, "bias"), a, w)
        return (out, carry.view(-1)) if ctx.carried else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, d_carry=None):
        a, w = ctx.saved_tensors
        N, K = a.shape
        M = w.size(1)
        if dy is None:      # only with a carry (grads are not materialised then): the output itself was unused
            return None, None, None, d_carry
        dy = f32c(dy, "dy")
        da = None
        if ctx.needs_input_grad[0]:
            if M <= 96 and K <= 320 and K > 64:
                # dy[N, M] @ w^T[M, K] with a wide output: the 120 KB-image k_ts_gemm variant (the library GEMM picks 16x256
                # tiles for this shape: 44 us for 60 -> 300 at N = 20 k)
                lib = _lib.load()
                scope = ctx.scope
                img = _scoped(scope.bwd if scope else None, ("tall-dx", id(w)), w, lambda: _ts_image(w, M, K, True))
                da = torch.empty(N, K, dtype=torch.float32, device=a.device)
                check(lib.glam_ts_gemm(ptr(dy), M, M, None, 0, 0, ptr(img), None, ptr(da), K, K, None, 0, 0, N, stream()), "glam_ts_gemm")
            else:
                da = torch.matmul(dy, w.t())
        dw = db = None
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
                    return da, None, None, (flat if (add is not None or d_carry is None) else flat.add_(d_carry))
                return da, dwb[:K], dwb[K]
            dw = torch.empty(K, M, dtype=torch.float32, device=a.device)
            if M <= 128:      # dw = a^T dy: P = a (up to 320 columns), Q = dy (two 64-column chunks beyond 64)
                product(a, K, K, 0, dy, M, M, dw, M, 1)
            else:             # wide output: dw^T = dy^T a, written through transposed strides
                product(dy, M, M, 0, a, K, K, dw, 1, M)
            if ctx.carried:
                flat = dw.view(-1)
                return da, None, None, (flat if (add is not None or d_carry is None) else flat.add_(d_carry))
        return da, dw, db


def _matmul_tall_node(a, w, bias):
    """``_MatmulTall`` with the gradients of (w, bias) carried across the applications of a block inside a weight_scope."""
    K, M = w.shape
    total = K * M + (M if bias is not None else 0)
    params = (w,) if bias is None else (w, bias)

    def split(flat):     # [d_w (K x M) | d_bias (M)]: the layout of the [a | 1]^T dy product
        return (flat[:K * M].view(K, M),) if bias is None else (flat[:K * M].view(K, M), flat[K * M:])
    key = ("carry-tall", id(w), id(bias))
    carry = _carry_for(key, params, total, split) if (w.requires_grad or (bias is not None and bias.requires_grad)) else None
    if carry is None:
        return _MatmulTall.apply(a, w, bias)
    out, carry = _MatmulTall.apply(a, w, bias, carry)
    _carry_store(key, w, carry)
    return out


def matmul_tall(a, w, bias=None):
    """``a @ w (+ bias)`` with the weight gradient on the MFMA reduction kernel when it fits: one of (K, M) <= 320 and the other
    <= 128, multiples of 4 (with a bias: K + 1 <= 320 and M <= 128)."""
    K, M = w.shape
    ok = a.dim() == 2 and a.is_cuda and K % 4 == 0 and M % 4 == 0 and a.size(0) >= 64
    if ok and bias is not None and K + 1 <= 320 and M <= 128:
        return _matmul_tall_node(a, w, bias)
    if ok and ((K <= 320 and M <= 128) or (K <= 128 and M <= 320)):
        out = _matmul_tall_node(a, w, None)
        return out if bias is None else out + bias
    out = torch.matmul(a, w)
    return out if bias is None else out + bias


class _LinearTall(torch.autograd.Function):
    """``y = x @ w^T + b`` for layer widths beyond the MFMA forward table (e.g. the GRU gate linears 92 -> 276 of
    hid_dim_alpha = 6): the data-side products on the library GEMM, ``[d_w | d_b] = dy^T [x | 1]`` on ``k_wgrad``."""

    @staticmethod
    def forward(ctx, x, w, b):
        require_device(x, w, b)
        x, w, b = f32c(x, "x"), f32c(w, "weight"), f32c(b, "bias")
        ctx.save_for_backward(x, w)
        N, K = x.shape
        M = w.size(0)
        if K <= 96 and M <= 320:       # the 120 KB-image k_ts_gemm variant (24 vs 29 us for 92 -> 276 at N = 20.7 k)
            lib = _lib.load()
            img = _scoped(_SCOPE.fwd if _SCOPE else None, ("lin", id(w)), w, lambda: _ts_image(w, K, M, True))
            y = torch.empty(N, M, dtype=torch.float32, device=x.device)
            check(lib.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), ptr(b), ptr(y), M, M, None, 0, 0, N, stream()), "glam_ts_gemm")
            return y
        return torch.addmm(b, x, w.t())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = f32c(dy, "dy")
        N, K = x.shape
        M = w.size(0)
        dx = torch.matmul(dy, w) if ctx.needs_input_grad[0] else None
        lib = _lib.load()
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=x.device)
        if K + 1 <= 64:
            # weight and bias gradients as separate contiguous tensors: autograd keeps them as they are (views of one [M, K + 1] buffer
            # cost a copy launch each when they become .grad)
            dw, db = torch.empty(M, K, dtype=torch.float32, device=x.device), torch.empty(M, dtype=torch.float32, device=x.device)
            check(lib.glam_wgrad_gemm_split(ptr(dy), M, M, ptr(x), K, K, ptr(dw), ptr(db), N, ptr(ws), ws.numel(), stream()),
                  "glam_wgrad_gemm_split")
            return dx, dw, db
        dwb = torch.empty(M, K + 1, dtype=torch.float32, device=x.device)
        check(lib.glam_wgrad_gemm(ptr(dy), M, M, None, 0, 0, 0, ptr(x), K, K, 1, N, ptr(dwb), K + 1, 1, ptr(ws), ws.numel(), stream()),
              "glam_wgrad_gemm")
        return dx, dwb[:, :K], dwb[:, K]


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


def linear(x, weight, bias=None):
    """``F.linear`` on the hand-written kernels when the shape is in their table (the layer-sized linears of the
    path: GRU gates 60->180, input embedding 15->60, ... on the MFMA kernels; heads with <= 16 outputs as row dot products);
    larger / odd layers (e.g. the 300->1024 readout MLP) stay on the library GEMM, which is the right tool for them."""
    M, K = weight.shape
    f32 = x.dtype == torch.float32 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
    if x.dim() == 2 and x.is_cuda and f32 and M <= 16 and K >= 64 and K % 4 == 0 and not linear_supported(K, M):
        return _LinearNarrow.apply(x, weight, bias)      # (other dtypes — fp64, autocast — fall through to F.linear below)
    if x.dim() != 2 or not linear_supported(K, M):
        if x.dim() == 2 and x.is_cuda and bias is not None and M % 4 == 0 and x.dtype == torch.float32 and weight.dtype == torch.float32:
            return _LinearLib.apply(x, weight, bias)
        return torch.nn.functional.linear(x, weight, bias)
    Kp, Mp = (K + 3) // 4 * 4, (M + 3) // 4 * 4
    if Kp != K:
        x = pad_cols(x, Kp)
    if Kp != K and Mp == M and not (x.requires_grad and torch.is_grad_enabled()):
        return _Linear.apply(x, weight, bias)    # data input (atom features): the image is built from the unpadded weight
    if Kp != K or Mp != M:
        # padded once per model pass (the block's linears are applied message_steps times), like every derived weight
        w0, b0 = weight, bias
        weight, bias = scoped_weights(("lin-pad", id(w0), None if b0 is None else id(b0)), w0, lambda: (
            torch.nn.functional.pad(w0, (0, Kp - K, 0, Mp - M)),
            None if b0 is None else torch.nn.functional.pad(b0, (0, Mp - M))))
    y = _Linear.apply(x, weight, bias)
    return slice_cols(y, M)                    # pad columns are x @ 0 + 0


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


def gru_tail(x, h, identity, w_ih, w_hh, b_ih, b_hh, act="none", slope=0.0, celu_in=False, rng=None):
    """``h_new = GRU(celu(x) if celu_in else x, h)`` (one step), ``out = act(h_new + identity)`` (src_1gp/layer.py:261-266):
    the two gate GEMMs plus ONE elementwise launch per direction.  Returns ``(out, h_new)``.
    ``rng = (rr_lower, rr_upper, drop_p)`` (training mode of the reference's defaults): ``act == "rrelu"`` draws its slopes in
    the kernel and, with ``drop_p > 0``, the kernel also writes ``Dropout(drop_p)(out)`` and registers it as the dropped twin
    of ``out`` (``take_dropped``) — available on the one-node path (C % 4 == 0, C <= 60); elsewhere the caller applies
    ``ops.rrelu`` / ``ops.dropout`` itself (``rng`` must then be None)."""
    C = h.size(1)
    if rng is not None and not gru_block_supported(C, w_ih, b_ih, b_hh):
        raise GlamHipError("gru_tail(rng=...) needs the one-node GRU block (C % 4 == 0, C <= 60)")
    if gru_block_supported(C, w_ih, b_ih, b_hh):
        return _gru_block(x, h, identity, w_ih, w_hh, b_ih, b_hh, ACT_CODES[act], slope, celu_in, rng)
    Cp = (C + 3) // 4 * 4
    if Cp != C and b_ih is not None and b_hh is not None and tuple(w_ih.shape) == (3 * C, C) and linear_supported(Cp, 3 * Cp) \
            and 3 * Cp > 64:
        # odd widths: the same node at Cp with gate-wise zero-padded weights (built once per model pass).  Pad channels
        # stay exactly zero through the step: gates r = z = 1/2, n = tanh(0) = 0, h' = z * 0 = 0, act(0 + 0) = 0.
        def build():
            pw = lambda w: torch.nn.functional.pad(w.view(3, C, C), (0, Cp - C, 0, Cp - C)).reshape(3 * Cp, Cp)
            pb = lambda b: torch.nn.functional.pad(b.view(3, C), (0, Cp - C)).reshape(3 * Cp)
            return pw(w_ih), pw(w_hh), pb(b_ih), pb(b_hh)
        wi, wh, bi, bh = scoped_weights(("gru-pad", id(w_ih), id(w_hh), id(b_ih), id(b_hh)), w_ih, build)
        out_p, hn_p = _gru_block(pad_cols(x, Cp), pad_cols(h, Cp), None if identity is None else pad_cols(identity, Cp),
                                 wi, wh, bi, bh, ACT_CODES[act], slope, celu_in)
        return slice_cols(out_p, C), slice_cols(hn_p, C)
    if b_ih is not None and b_hh is not None and tuple(w_ih.shape) == (3 * C, C) and linear_tall_supported(Cp, 3 * Cp) and h.size(0) >= 64:
        # wide GRU (hid_dim_alpha = 6): library GEMMs for the gate products, k_wgrad for their weight gradients, at Cp
        def build_wide():
            pw = lambda w: torch.nn.functional.pad(w.view(3, C, C), (0, Cp - C, 0, Cp - C)).reshape(3 * Cp, Cp)
            pb = lambda b: torch.nn.functional.pad(b.view(3, C), (0, Cp - C)).reshape(3 * Cp)
            return pw(w_ih), pw(w_hh), pb(b_ih), pb(b_hh)
        wi, wh, bi, bh = scoped_weights(("gru-pad", id(w_ih), id(w_hh), id(b_ih), id(b_hh)), w_ih, build_wide) if Cp != C else \
            (w_ih, w_hh, b_ih, b_hh)
        x_p, h_p = pad_cols(x, Cp), pad_cols(h, Cp)
        if celu_in:
            x_p = torch.celu(x_p)
        out_p, hn_p = _GruTail.apply(_LinearTall.apply(x_p, wi, bi), _LinearTall.apply(h_p, wh, bh), h_p,
                                     None if identity is None else pad_cols(identity, Cp), ACT_CODES[act], slope)
        return slice_cols(out_p, C), slice_cols(hn_p, C)
    if celu_in:
        x = torch.celu(x)
    return _GruTail.apply(linear(x, w_ih, b_ih), linear(h, w_hh, b_hh), h, identity, ACT_CODES[act], slope)


GEMM_PAIR = os.environ.get("GLAM_GEMM_PAIR", "1") == "1"     # A/B knob: the GRU's two products per direction in one launch
# Gate GEMMs + gate math + tail of the forward GRU step in ONE launch (glam_gru_fused_fwd, bit-identical to the pair launch + tail
# kernel, tested).  Its work item is coarse (a 16-row tile x both products x all three gates) and its epilogue runs at the chip's write
# bandwidth, so it pays once a wave has several tiles to pipeline: model step at B = 8 192 3.90 vs 4.08 ms, B = 1 024 0.807 vs 0.814 ms,
# B = 32 0.384 vs 0.370 ms (the 96 KB of weight images per block dominate).  "auto": from GRU_FUSED_MIN_NODES nodes on.
GRU_FUSED = os.environ.get("GLAM_GRU_FUSED", "auto")
GRU_FUSED_MIN_NODES = 16384


def _want_gru_fused(N):
    return GRU_FUSED in ("1", True) or (GRU_FUSED == "auto" and N >= GRU_FUSED_MIN_NODES)

def _gru_block(x, h, identity, w_ih, w_hh, b_ih, b_hh, act, slope, celu_in, rng=None):
    """``_GruBlock`` with the gradients of its four parameters carried across the block's applications (see _ParamBundle)."""
    M, C = w_ih.shape
    def split(flat):      # [d_w_ih | d_b_ih | d_w_hh | d_b_hh], every piece contiguous: autograd takes the views without a copy
        w1, b1, w2, b2 = flat.split([M * C, M, M * C, M])
        return w1.view(M, C), w2.view(M, C), b1, b2
    key = ("carry-gru", id(w_ih))
    carry = _carry_for(key, (w_ih, w_hh, b_ih, b_hh), 2 * M * (C + 1), split)
    out, h_new, out_drop, carry = _GruBlock.apply(x, h, identity, w_ih, w_hh, b_ih, b_hh, act, slope, celu_in, carry, rng)
    if carry is not None:
        _carry_store(key, w_ih, carry)
    if out_drop is not None:
        register_dropped(out, out_drop, rng[2])
    return out, h_new


class _GruBlock(torch.autograd.Function):
    """The whole GRU step of a MessageBlock as ONE autograd node: both gate GEMMs + gates/residual/activation forward;
    gate backward + both input-gradient GEMMs + BOTH weight-gradient products in one launch pair backward.  Besides the
    launches it saves (one weight-gradient launch and one reduction per step) it replaces three Python autograd nodes by
    one, which is what an eagerly issued training step is bound by."""

    @staticmethod
    def forward(ctx, x, h, identity, w_ih, w_hh, b_ih, b_hh, act, slope, celu_in, carry=None, rng=None):
        ctx.set_materialize_grads(False)     # unused outputs (the last step's h', its dropped twin) arrive as None, not as zero fills
        require_device(x, h, w_ih, w_hh, b_ih, b_hh)
        x, h = f32c(x, "x"), f32c(h, "h")
        w_ih, w_hh, b_ih, b_hh = f32c(w_ih, "weight_ih"), f32c(w_hh, "weight_hh"), f32c(b_ih, "bias_ih"), f32c(b_hh, "bias_hh")
        identity = None if identity is None else f32c(identity, "identity")
        N, C = h.shape
        M = 3 * C
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        scope = _SCOPE

        def image(w):
            def build():
                img = torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, **f)
                check(lib.glam_ts_gemm_make_image(ptr(w), C, 1, C, M, ptr(img), stream()), "glam_ts_gemm_make_image")
                return img
            return _scoped(scope.fwd if scope else None, ("lin", id(w)), w, build)

        gi, gh = torch.empty(N, M, **f), torch.empty(N, M, **f)
        st = stream()
        if scope is not None:
            # all four images of the step (forward and input-gradient images of both gate matrices) in ONE launch, shared by the
            # message_steps applications of the block through the scope tables
            ka, kb = ("lin", id(w_ih)), ("lin", id(w_hh))
            ha, hb = scope.fwd.get(ka), scope.fwd.get(kb)
            if not (ha is not None and ha[0] is w_ih and hb is not None and hb[0] is w_hh):
                nf, nb = lib.glam_ts_gemm_image_bytes(C, M) // 4, lib.glam_ts_gemm_image_bytes(M, C) // 4
                ia, ib, ta, tb = torch.empty(nf, **f), torch.empty(nf, **f), torch.empty(nb, **f), torch.empty(nb, **f)
                if N > 0 and _want_gru_fused(N) and lib.glam_gru_fused_supported(C):
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
        if N > 0 and _want_gru_fused(N) and lib.glam_gru_fused_supported(C):
            # both gate linears + gates + residual + activation (+ RReLU / Dropout) in ONE launch (bit-identical to the sequence below)
            def build_fused():
                nb = lib.glam_gru_fused_image_bytes() // 4
                buf = torch.empty(2, nb, **f)
                check(lib.glam_gru_fused_make_images(ptr(w_ih), ptr(w_hh), C, ptr(buf[0]), ptr(buf[1]), st), "glam_gru_fused_make_images")
                return buf
            imgs = _scoped(scope.fwd if scope else None, ("gru-fused", id(w_ih), id(w_hh)), w_ih, build_fused)
            if rng is None:
                check(lib.glam_gru_fused_fwd(ptr(x), ptr(h), ptr(identity), ptr(imgs[0]), ptr(imgs[1]), ptr(b_ih), ptr(b_hh), N, C,
                                             int(celu_in), act, float(slope), ptr(gi), ptr(gh), ptr(h_new), ptr(out), st), "glam_gru_fused_fwd")
            else:
                check(lib.glam_gru_fused_rng_fwd(ptr(x), ptr(h), ptr(identity), ptr(imgs[0]), ptr(imgs[1]), ptr(b_ih), ptr(b_hh), N, C,
                                                 int(celu_in), act, float(slope), lo, hi, p, ptr(rng_state(dev)), ptr(eff), ptr(gi), ptr(gh),
                                                 ptr(h_new), ptr(out), ptr(out_drop), st), "glam_gru_fused_rng_fwd")
        else:
            if GEMM_PAIR:     # both gate linears in ONE launch (two products of the same kernel variant share the CUs)
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
                                                ptr(rng_state(dev)), ptr(eff), ptr(h_new), ptr(out), ptr(out_drop), st), "glam_gru_tail_rng_fwd")
        ctx.save_for_backward(x, h, gi, gh, out, w_ih, w_hh)
        ctx.eff = eff
        ctx.cfg = (act, float(slope), identity is not None, bool(celu_in), None if rng is None else tuple(float(v) for v in rng))
        ctx.scope = scope
        ctx.carried = carry is not None
        return out, h_new, out_drop, (carry.view(-1) if ctx.carried else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_hstate, d_out_drop=None, d_carry=None):
        x, h, gi, gh, out, w_ih, w_hh = ctx.saved_tensors
        act, slope, has_res, celu_in, rng = ctx.cfg
        N, C = h.shape
        M = 3 * C
        lib, dev = _lib.load(), x.device
        f = dict(dtype=torch.float32, device=dev)
        st = stream()
        d_out = None if d_out is None else f32c(d_out, "d_out")
        d_out_drop = None if d_out_drop is None else f32c(d_out_drop, "d_out_drop")
        d_hstate = None if d_hstate is None else f32c(d_hstate, "d_hstate")
        d_gi, d_gh, d_h = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        d_id = torch.empty_like(h) if has_res else None
        if rng is None:
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
        scope = ctx.scope

        def image_t(w):
            def build():
                img = torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, **f)
                check(lib.glam_ts_gemm_make_image(ptr(w), C, 0, M, C, ptr(img), stream()), "glam_ts_gemm_make_image")
                return img
            return _scoped(scope.bwd if scope else None, ("lin", id(w)), w, build)

        dx, dh = torch.empty(N, C, **f), torch.empty(N, C, **f)
        # with the folded CELU the epilogue multiplies by celu'(x): dx is the gradient of the RAW input
        # d_h = d_gh @ W_hh^T + the direct z * g path of the gate equations (the addend of the GEMM's epilogue); both products in one launch
        if GEMM_PAIR:
            img_a, img_b = image_t(w_ih), image_t(w_hh)
            check(lib.glam_ts_gemm_pair(ptr(d_gi), M, M, 0, ptr(img_a), None, ptr(dx), C, C, ptr(x) if celu_in else None, C, None, 0,
                                        ptr(d_gh), M, M, 0, ptr(img_b), None, ptr(dh), C, C, None, 0, ptr(d_h), C, N, st),
                  "glam_ts_gemm_pair")
        else:
            check(lib.glam_ts_gemm_celu(ptr(d_gi), M, M, 0, ptr(image_t(w_ih)), None, ptr(dx), C, C, ptr(x) if celu_in else None, C, N, st),
                  "glam_ts_gemm_celu")
            check(lib.glam_ts_gemm_add(ptr(d_gh), M, M, ptr(image_t(w_hh)), None, ptr(dh), C, C, ptr(d_h), C, N, st), "glam_ts_gemm_add")
        # [d_W | d_b] of both linears: out[m, k] = sum_n dy[n, m] * [x | 1][n, k], two products, one launch + one reduction
        ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        # one buffer [d_w_ih | d_b_ih | d_w_hh | d_b_hh], contiguous pieces; a gradient carry (same layout) is added by the reduction
        flat = torch.empty(2 * M * (C + 1), **f)
        dw_ih, db_ih, dw_hh, db_hh = flat.split([M * C, M, M * C, M])
        dc = [None] * 4
        if ctx.carried and d_carry is not None and N > 0:
            dc = f32c(d_carry, "d_carry").split([M * C, M, M * C, M])
            d_carry = None
        check(lib.glam_wgrad_gemm_pair_split(ptr(d_gi), M, M, ptr(x), C, C, int(celu_in), ptr(dw_ih), ptr(db_ih),
                                             ptr(d_gh), M, M, ptr(h), C, C, 0, ptr(dw_hh), ptr(db_hh), N, ptr(ws), ws.numel(),
                                             ptr(dc[0]), ptr(dc[1]), ptr(dc[2]), ptr(dc[3]), st), "glam_wgrad_gemm_pair_split")
        if ctx.carried:
            return dx, dh, d_id, None, None, None, None, None, None, None, (flat if d_carry is None else flat.add_(d_carry)), None
        return dx, dh, d_id, dw_ih.view(M, C), dw_hh.view(M, C), db_ih, db_hh, None, None, None, None, None


def gru_block_supported(C, w_ih, b_ih, b_hh):
    return C % 4 == 0 and C + 1 <= 64 and linear_supported(C, 3 * C) and 3 * C > 64 and b_ih is not None and b_hh is not None and \
        tuple(w_ih.shape) == (3 * C, C)        # C + 1 <= 64: both weight gradients in ONE k_wgrad launch


def gru_step(x, h, w_ih, w_hh, b_ih, b_hh):
    """One ``torch.nn.GRU(C, C)`` step with seq_len 1 on its own parameters (src_1gp/layer.py:247, :262)."""
    return _GruGates.apply(linear(x, w_ih, b_ih), linear(h, w_hh, b_hh), h)


# --------------------------------------------------------------------------------------
# readouts
# --------------------------------------------------------------------------------------
class _Pool5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sp, k):
        require_device(x)
        x = f32c(x, "x")
        N, D = x.shape
        if N != sp.N:
            raise GlamHipError(f"pool: x has {N} rows but batch has {sp.N}")
        out = torch.empty(sp.B, (2 + k) * D, dtype=torch.float32, device=x.device)
        topk = torch.empty(sp.B, k, dtype=torch.int32, device=x.device)
        check(_lib.load().glam_pool5_fwd(ptr(x), ptr(sp.ptr), N, sp.B, D, k, ptr(out), ptr(topk), stream()),
              "glam_pool5_fwd")
        ctx.save_for_backward(topk)
        ctx.sp, ctx.dims = sp, (N, D, k)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        (topk,) = ctx.saved_tensors
        N, D, k = ctx.dims
        sp = ctx.sp
        d_out = f32c(d_out, "d_out")
        d_x = torch.empty(N, D, dtype=torch.float32, device=d_out.device)
        check(_lib.load().glam_pool5_bwd(ptr(d_out), ptr(sp.ptr), ptr(topk), N, sp.B, D, k, ptr(d_x), stream()),
              "glam_pool5_bwd")
        return d_x, None, None


def pool5(x, sp, k=3):
    """mean || add || sort-pool(k) readout, ``[B, (2+k)*D]``."""
    return _Pool5.apply(x, sp, k)


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
            if act == ACT_CODES["rrelu"] or not want_out:
                raise GlamHipError("bias_res_act: 'rrelu' / dropout-only need rng=(lower, upper, drop_p)")
            check(_lib.load().glam_bias_res_act_fwd(ptr(y), ptr(bias), ptr(identity), N, C, act, float(slope), ptr(out), stream()),
                  "glam_bias_res_act_fwd")
        else:
            lo, hi, p = (float(v) for v in rng)
            eff = torch.empty(2, dtype=torch.int64, device=y.device)
            out_drop = torch.empty_like(y) if p > 0 else None
            check(_lib.load().glam_bias_res_act_rng_fwd(ptr(y), ptr(bias), ptr(identity), N, C, act, float(slope), lo, hi, p,
                                                        ptr(rng_state(y.device)), ptr(eff), ptr(out), ptr(out_drop), stream()),
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
    out, out_drop = _BiasResAct.apply(y, bias, identity, ACT_CODES[act], slope, rng, True)
    if out_drop is not None:
        register_dropped(out, out_drop, rng[2])
    return out


def rrelu(x, lower=1.0 / 8, upper=1.0 / 3, drop_p=0.0):
    """Training-mode ``torch.nn.RReLU(lower, upper)`` on the device-side Philox stream (one launch per direction, slopes
    regenerated in the backward); ``drop_p > 0`` also writes the dropped twin for a ``Dropout(drop_p)`` that follows."""
    shape = x.shape
    out, out_drop = _BiasResAct.apply(x.reshape(-1, shape[-1]), None, None, ACT_CODES["rrelu"], 0.0, (lower, upper, drop_p), True)
    out = out.view(shape)
    if out_drop is not None:
        register_dropped(out, out_drop.view(shape), drop_p)
    return out


def dropout(x, p):
    """Training-mode ``torch.nn.Dropout(p)``: ``x * mask / (1 - p)``; the mask is regenerated in the backward (no mask tensor)."""
    shape = x.shape
    _, out_drop = _BiasResAct.apply(x.reshape(-1, shape[-1]), None, None, ACT_CODES["none"], 0.0, (1.0, 1.0, float(p)), False)
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
    def forward(ctx, x, sp, mode, scale, eps, with_identity=False):
        require_device(x)
        x = f32c(x, "x")
        N, D = x.shape
        if N != sp.N:
            raise GlamHipError(f"graph_norm: x has {N} rows but batch has {sp.N}")
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
            return d_id, None, None, None, None, None
        gy = f32c(gy, "gy")
        dx = torch.empty_like(x)
        lib = _lib.load()
        if d_id is not None and N > 0 and sp.B > 0:
            check(lib.glam_graph_norm_bwd_add(ptr(x), ptr(gy), ptr(sp.ptr), N, sp.B, D, mode, scale, eps, ptr(f32c(d_id, "d_identity")), ptr(dx),
                                              stream()), "glam_graph_norm_bwd_add")
            return dx, None, None, None, None, None
        check(lib.glam_graph_norm_bwd(ptr(x), ptr(gy), ptr(sp.ptr), N, sp.B, D, mode, scale, eps, ptr(dx), stream()),
              "glam_graph_norm_bwd")
        if d_id is not None:
            dx = d_id if (N == 0 or sp.B == 0) else dx + d_id
        return dx, None, None, None, None, None


def pair_norm(x, sp, scale=1.0, eps=1e-5, with_identity=False):
    """PyG ``PairNorm(scale, eps=1e-5)(x, batch)`` (one kernel per direction); ``with_identity``: ``(y, x)`` — see _GraphNorm."""
    return _GraphNorm.apply(x, sp, 0, scale, eps, with_identity)


def graph_standardize(x, sp, eps=1e-5, with_identity=False):
    """Statistics part of PyG's graph ``LayerNorm(x, batch)``: zero mean / unit variance per graph."""
    return _GraphNorm.apply(x, sp, 1, 1.0, eps, with_identity)


class _EdgeWeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gi, mean, self_slot):
        require_device(x, w)
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
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        (w,) = ctx.saved_tensors
        gi = ctx.gi
        N, E, D, K, mean, self_slot = ctx.cfg
        d_out = f32c(d_out, "d_out")
        colptr, dst, eid_t = gi.transpose()
        dx = torch.empty(N, D, dtype=torch.float32, device=d_out.device)
        check(_lib.load().glam_edge_wsum_bwd(ptr(d_out), ptr(w), ptr(colptr), ptr(dst), ptr(eid_t), ptr(gi.rowptr), N, E, D, K,
                                             mean, self_slot, ptr(dx), stream()), "glam_edge_wsum_bwd")
        return dx, None, None, None, None


def edge_weighted_sum(x, w, gi, mean=False, self_slot=False):
    """``S[n,k,:] = (1/deg_n) sum_{e->n} w[e,k] * x[src_e,:]`` -> ``[N, K, D]`` (no gradient w.r.t. ``w``: edge data);
    ``self_slot``: ``[N, K+1, D]`` with ``S[n,K,:] = x[n,:]`` (K in {4, 8}, D % 4 == 0)."""
    return _EdgeWeightedSum.apply(x, w, gi, mean, self_slot)


def self_slot_supported(K, D):
    return K in (4, 8) and D % 4 == 0


_ONEHOT_CACHE: dict = {}


def rows_are_one_hot(t):
    """True iff every row of ``t`` is one-hot (one host sync, cached per tensor object like the CSR staging).  Inside a stream capture an
    answer that is not cached yet is ``False`` (no read-back there): the contraction path is always correct."""
    key = id(t)
    hit = _ONEHOT_CACHE.get(key)
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return hit[2]
    mark = getattr(t, "_glam_onehot", None)      # (flag, tensor version) known from the host side (data.PackedDataset): no read-back
    if not (mark is not None and mark[1] == t._version) and t.is_cuda and torch.cuda.is_current_stream_capturing():
        return False
    ok = bool(mark[0]) if (mark is not None and mark[1] == t._version) else \
        (bool((((t == 0) | (t == 1)).all() & (t.sum(dim=1) == 1).all()).item()) if t.numel() else True)
    try:
        _ONEHOT_CACHE[key] = (weakref.ref(t, lambda _r, k=key, c=_ONEHOT_CACHE: c.pop(k, None)), t._version, ok)
    except TypeError:
        pass
    return ok


class _PairPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mol, pro, msp, psp):
        require_device(mol, pro)
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
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        mol, pro, arg, sums = ctx.saved_tensors
        msp, psp = ctx.sps
        d_out = f32c(d_out, "d_out")
        d_mol, d_pro = torch.empty_like(mol), torch.empty_like(pro)
        check(_lib.load().glam_pair_pool_bwd(ptr(mol), ptr(pro), ptr(msp.ptr), ptr(psp.ptr), ptr(arg), ptr(sums), ptr(d_out), msp.B,
                                             mol.size(1), ptr(d_mol), ptr(d_pro), stream()), "glam_pair_pool_bwd")
        return d_mol, d_pro, None, None


def pair_pool(mol_out, pro_out, msp, psp):
    """``[max, mean]`` of ``mol[seg_i] @ pro[seg_i].T`` per pair -> ``[P, 2]`` (dot_and_global_pool2)."""
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
