"""hipGraph capture of whole training steps.

The message-passing path is launch bound at molecular batch sizes (≈190 kernels of 5–20 µs per full-model step):
issued eagerly from Python the step costs 2.2 ms, replayed from a hipGraph 1.0 ms.  The reference's
train loader does not shuffle (``src_1gp/trainer.py:37-38``), so with ``glam_amd.data.DataLoader(cache=True)`` every
batch object comes back each epoch with the same tensors at the same addresses — exactly what a captured graph needs.

``GraphedTrainStep`` keeps one graph per batch object:

* first visit: the step runs eagerly (this is also where the one-time host work of a new batch happens: CSR staging and
  its validation sync, one-hot detection, ...);
* second visit: the step is captured and replayed once;
* later visits: one ``hipGraphLaunch``.

Every visit performs exactly one optimizer step, so the parameter trajectory is the eager one (same kernels, same
order; the kernels are deterministic) — to the bit below 512 atoms per batch or with a block applied once; above that a captured
step sums the weight gradients of all applications of a block in one product (``ops.GRU_WGRAD_BATCH``), the eager first visit one
per application: equal within fp32 rounding.  ``GraphedTrainStep.run(batches)`` goes one step further for the epoch loop: up to
``steps_per_graph`` CONSECUTIVE steps (one per batch, in order) are captured into one graph — every ``hipGraphLaunch`` carries a
bubble of ≈7 µs on this stack (a 99 µs step replays in 92 µs at eight steps per launch, bench.py), and an ESOL epoch at the reference's
batch size is 36 launches otherwise.  Requirements: an optimizer created with ``capturable=True`` (Adam/AdamW), a
loss function of ``(output, batch)`` that stays on the device, and no data-dependent Python control flow in the model.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import weakref

import torch


class GraphedTrainStep:
    """See the module docstring.  Hyper-parameters: a captured optimizer launch bakes Python floats in as constants, and the
    reference drives ``lr`` with ``ReduceLROnPlateau`` (``src_1gp/trainer.py:55,85``), which assigns a new float to
    ``param_group['lr']``.  So the learning rate of every group lives in a DEVICE TENSOR that the captured launches read;
    before every step the group's current value (a float a scheduler wrote, or a tensor) is copied into it — replayed steps
    follow the scheduler exactly like eager ones.  ``betas`` / ``eps`` / ``weight_decay`` / flags are snapshotted; if they ever
    change, every graph is dropped and re-captured on its next visit.

    The returned loss is a fresh tensor (a clone of the graph's static output).  After a replay ``p.grad`` refers to the
    gradient buffers of the graph captured LAST, not necessarily the one replayed: inspect or clip gradients in eager mode."""

    def __init__(self, model, optimizer, loss_fn, max_graphs=4096):
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep needs an optimizer created with capturable=True "
                                 "(e.g. Adam(..., capturable=True, fused=True)); fused=True alone is not enough")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.max_graphs = max_graphs
        self._one = {}        # device -> the scalar 1.0 every step's backward starts from
        self._state = {}      # id(batch) -> [weakref, visits, graph, static_loss]
        self._multi = {}      # (id(batch), ...) -> [weakrefs, graph, static_losses]: several consecutive steps per graph launch
        self._pool = None
        self._lr = [None] * len(optimizer.param_groups)
        self._hyper = None

    _BAKED = ("betas", "eps", "weight_decay", "amsgrad", "maximize", "momentum", "dampening", "nesterov", "alpha", "centered")

    def _sync_hyper(self):
        for i, g in enumerate(self.optimizer.param_groups):
            lr, t = g["lr"], self._lr[i]
            if t is None:
                dev = g["params"][0].device
                t = self._lr[i] = (lr.detach().to(device=dev, dtype=torch.float32).clone() if torch.is_tensor(lr)
                                   else torch.tensor(float(lr), dtype=torch.float32, device=dev))
                g["lr"] = t
            elif lr is not t:                 # a scheduler (or the user) assigned a new value since the last step
                if torch.is_tensor(lr):
                    t.copy_(lr)
                else:
                    t.fill_(float(lr))
                g["lr"] = t
        snap = tuple(tuple((k, g[k]) for k in self._BAKED if k in g) for g in self.optimizer.param_groups)
        if self._hyper is not None and snap != self._hyper:
            for st in self._state.values():   # constants of the captured optimizer launches changed: capture again
                st[2] = st[3] = None
            self._multi.clear()
        self._hyper = snap

    def _step(self, batch):
        self.optimizer.zero_grad(set_to_none=True)
        with no_graphed_call():           # the whole step is this class's graph: the model's own graphed-callable route stays out of it
            loss = self.loss_fn(self.model(batch), batch)
        # the root gradient as a tensor kept across steps: `loss.backward()` alone launches a fill for ones_like(loss) in every step
        one = self._one.get(loss.device) if (loss.dim() == 0 and loss.dtype == torch.float32 and loss.is_cuda) else None
        if one is None and loss.dim() == 0 and loss.dtype == torch.float32 and loss.is_cuda and not torch.cuda.is_current_stream_capturing():
            one = self._one[loss.device] = torch.ones((), dtype=torch.float32, device=loss.device)     # (first visits are eager)
        if one is not None:
            loss.backward(gradient=one)
        else:
            loss.backward()
        self.optimizer.step()
        # detached: a caller that keeps the returned loss (a list of per-step losses, say) would otherwise keep the step's autograd
        # graph — and its AccumulateGrad nodes, bound to the stream of THIS step — alive into the capture of a later step on the
        # capture stream, which crashes the runtime at capture end
        return loss.detach()

    def __call__(self, batch):
        """One optimizer step on ``batch``; returns the (device) loss tensor of this step."""
        self._sync_hyper()
        key = id(batch)
        st = self._state.get(key)
        if st is None or st[0]() is not batch:
            if len(self._state) >= self.max_graphs:
                return self._step(batch)
            ref = weakref.ref(batch, lambda _r, k=key, d=self._state: d.pop(k, None))
            self._state[key] = [ref, 1, None, None]
            return self._step(batch)                      # first visit: eager (stages the CSR, syncs once)
        st[1] += 1
        if st[2] is None:                                 # second visit: capture, then fall through to the replay
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self._pool):
                st[3] = self._step(batch)
            if self._pool is None:
                self._pool = graph.pool()                 # all graphs share one memory pool: they never run concurrently
            st[2] = graph
        st[2].replay()
        return st[3].clone()                              # static outputs of different graphs may alias in the shared pool

    def run(self, batches, steps_per_graph=16):
        """``len(batches)`` optimizer steps, one per batch, in order — the parameter trajectory of calling the stepper on each batch in
        turn — with up to ``steps_per_graph`` consecutive steps per graph launch.  A chunk is captured once every batch in it has had
        its eager first visit (so the first epoch runs step by step, the second captures, later ones replay); the same sequence of
        batch objects must come back for the replay to apply (the cached, non-shuffling loader of the reference's training loop).
        Returns the per-step losses as one device tensor ``[len(batches)]``."""
        batches = list(batches)
        losses = []
        for i in range(0, len(batches), max(1, steps_per_graph)):
            losses.extend(self._run_chunk(batches[i:i + max(1, steps_per_graph)]))
        return torch.stack([l.reshape(()) for l in losses]) if losses else torch.empty(0)

    def _run_chunk(self, chunk):
        self._sync_hyper()
        seen = all((st := self._state.get(id(b))) is not None and st[0]() is b for b in chunk)
        if len(chunk) == 1 or not seen or len(self._multi) >= self.max_graphs:
            return [self(b) for b in chunk]
        key = tuple(id(b) for b in chunk)
        mt = self._multi.get(key)
        if mt is None or any(r() is not b for r, b in zip(mt[0], chunk)):
            refs = [weakref.ref(b, lambda _r, k=key, d=self._multi: d.pop(k, None)) for b in chunk]
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self._pool):
                outs = [self._step(b) for b in chunk]
            if self._pool is None:
                self._pool = graph.pool()
            mt = self._multi[key] = [refs, graph, outs]
        mt[1].replay()
        return [o.clone() for o in mt[2]]

    def graphs(self):
        return sum(1 for s in self._state.values() if s[2] is not None) + len(self._multi)


class GraphedForward:
    """Inference counterpart: ``GraphedForward(model)(batch)`` runs ``model(batch)`` under ``torch.no_grad()`` — eagerly on
    the first visit of a batch object, from a hipGraph afterwards (the evaluation loaders of the reference iterate the same
    batches after every epoch: ``src_1gp/trainer.py:39-41``).  The returned tensor is a clone of the graph's static output (the graphs share one memory
    pool, so static outputs of different batches may alias).  Parameters may change between calls (training in between): the
    captured kernels re-read them; the model must be in ``eval()`` mode (or otherwise free of RNG-dependent layers that
    differ between the modes you compare)."""

    def __init__(self, model, max_graphs=4096):
        self.model, self.max_graphs = model, max_graphs
        self._state = {}
        self._pool = None

    @torch.no_grad()
    def __call__(self, *batches):
        key = tuple(id(b) for b in batches)
        st = self._state.get(key)
        if st is None or any(r() is not b for r, b in zip(st[0], batches)):
            if len(self._state) >= self.max_graphs:
                return self.model(*batches)
            refs = [weakref.ref(b, lambda _r, k=key, d=self._state: d.pop(k, None)) for b in batches]
            self._state[key] = [refs, None, None]
            with no_graphed_call():
                return self.model(*batches)
        if st[1] is None:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self._pool), no_graphed_call():
                st[2] = self.model(*batches)
            if self._pool is None:
                self._pool = graph.pool()
            st[1] = graph
        st[1].replay()
        return st[2].clone()


# ---------------------------------------------------------------------------------------------------------------------------------
# The graphed-callable route: ``model(batch)`` itself replays hipGraphs, so the reference's training loop runs UNCHANGED
# (src_1gp/trainer.py:286-304: ``optimizer.zero_grad(); output = self.model(mol_batch); loss.backward(); optimizer.step()``, any
# optimizer, a fresh device copy of the batch every iteration).
# ---------------------------------------------------------------------------------------------------------------------------------
GRAPHED_CALL = os.environ.get("GLAM_GRAPHED_CALL", "1") != "0"      # process-wide switch; per model: ``model.graphed_call = False``
_suspended = 0


@contextlib.contextmanager
def no_graphed_call():
    """Inside: every model takes its eager forward (used by the whole-step graphs above and by the route's own captures)."""
    global _suspended
    _suspended += 1
    try:
        yield
    finally:
        _suspended -= 1


# module-level A/B switches of glam_amd.ops and the library's environment switches: a captured graph bakes the route in, so they are part
# of a graph's key (flipping one between two calls of a model — the parity tests do — must not replay the other route)
_OPS_KNOBS = ("VALIDATE", "WS_ROUTE", "GRAD_CARRY", "CACHED_STAGING", "USE_TORCH_EXT", "GEMM_PAIR", "GRU_FUSED", "GRU_FUSED_MIN_NODES", "GRU_WS",
              "SKIP_THROUGH_CONV", "GRU_WGRAD_BATCH", "GRU_PRE", "GRU_GATES", "NODE_IN_GRU", "NODE_IN_GRU_MAX_ROWS", "RRELU_IN_GEMM", "HEAD_ACT_FUSED", "DENSE_SPLITK", "NORM_DROP", "PRESTAGE", "DENSE_LINEAR", "INFER_FWD", "RELU_IN_WGRAD")
_ENV_KNOBS = ("GLAM_X3", "GLAM_WS", "GLAM_WGRAD_X3", "GLAM_WGRAD_X3_ROWS", "GLAM_TALL_X3", "GLAM_WS_GRID", "GLAM_B1_PRE")


_OPS_GET = None


def _route_signature():
    global _OPS_GET
    from . import ops
    if _OPS_GET is None:
        import operator
        _OPS_GET = operator.attrgetter(*_OPS_KNOBS)      # (one C call for the 22 switches: this runs on every call of a routed model)
    env = os.environ
    return _OPS_GET(ops) + (ops.GraphIndex.ELL_MIN_NODES,) + tuple([env.get(k) for k in _ENV_KNOBS])


class _Replay(torch.autograd.Function):
    """One autograd node for the whole model: forward = replay of the captured forward graph, backward = replay of the captured
    backward graph (the ``torch.cuda.make_graphed_callables`` shape).  The parameters are inputs of the node so that their gradients
    reach ``p.grad`` through autograd like any others (hooks, accumulation, any optimizer)."""

    @staticmethod
    def forward(ctx, st, *params):
        ctx.st = st
        st.gen += 1                                  # this replay owns the state's static activations until its backward has run
        ctx.gen = st.gen
        st.pending = weakref.ref(ctx)
        st.fwd.replay()
        return st.out.detach().clone()               # (static outputs stay private: a caller may keep the result across steps)

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        if ctx.gen != st.gen:
            # (the route itself never gets here: a call that finds an earlier output of this state still waiting for its backward runs
            # eagerly — GraphedCallable.__call__; this guards a state whose graphs were replayed behind the route's back)
            raise RuntimeError("glam_amd.graphs: the captured activations of this output were overwritten by a later forward of the same "
                               "batch before its backward ran; set model.graphed_call = False for this pattern")
        st.pending = None
        # A gradient buffer of this graph that autograd took over as p.grad in an earlier backward (p.grad was None then) and that is
        # still p.grad now — zero_grad(set_to_none=False), or a second backward before the optimizer step — would be overwritten by the
        # replay below and then added to itself: such a p.grad gets a private copy first.  (zero_grad()'s default, set_to_none=True,
        # never comes here: no copies on the common path.)
        for p, buf in zip(st.params, st.grads):
            if buf is not None and p.grad is not None and p.grad.data_ptr() == buf.data_ptr():
                p.grad = p.grad.clone()
        st.gout.copy_(g)
        st.bwd.replay()
        return (None, *[None if buf is None else buf.detach() for buf in st.grads])


class _CallState:
    __slots__ = ("static", "visits", "fwd", "bwd", "out", "gout", "params", "grads", "x_sig", "gen", "pending")

    def __init__(self, static):
        self.static, self.visits, self.gen = static, 0, 0
        self.fwd = self.bwd = self.out = self.gout = self.params = self.grads = self.x_sig = self.pending = None


class GraphedCallable:
    """``route(module, eager_forward, batch)`` = ``eager_forward(batch)``, from hipGraphs once a batch has been seen before.

    A batch is recognised by CONTENT: the shapes plus a 64-bit fingerprint of ``edge_index``, ``batch`` and ``edge_attr``
    (``glam_batch_fingerprint``, one launch and one 8-byte read-back per call; skipped when the very same tensors come back, as
    with ``glam_amd.data.DataLoader(cache=True)``) — the reference's trainer collates and copies every batch anew in every epoch,
    but its loader does not shuffle (``trainer.py:37-38``), so the same batches recur.

    * first visit of a content: the model runs eagerly on the caller's batch, as it always did;
    * second visit: the batch's tensors are copied into private static tensors and the model runs eagerly ON THEM (this is where their
      CSR / ELL staging and its one validation sync happen, cached on the static tensors);
    * third visit: the forward is captured into one hipGraph and — when autograd is recording — the backward into another, through
      ``torch.autograd.grad`` on the static output;
    * from then on ``model(batch)`` copies ``batch.x`` into the static tensor and replays the forward graph; the result carries ONE
      autograd node whose backward replays the backward graph and hands the parameters' gradients to autograd.

    Per-graph memory pools (a forward's saved activations must survive until its backward whatever else runs in between).  The route
    stays eager — silently, it is an optimisation — for CPU tensors, inside someone else's stream capture, when ``batch.x`` requires a
    gradient, for empty batches, when the model has forward hooks, beyond ``max_graphs`` contents, when less than a quarter of the
    device memory is free, and for a content whose previous output is still alive and has not been back-propagated yet (two forwards
    of one batch before the first backward — a consistency loss, say: a graph has ONE set of static activations, so the second
    forward runs eagerly and both backwards are right).  ``model.graphed_call = False`` or ``GLAM_GRAPHED_CALL=0`` switch it off.

    Captured graphs bake the parameters' ADDRESSES and the trainable set in.  They are dropped, and captured again on later visits,
    whenever a parameter object was replaced, a parameter's storage moved (``module.to()`` / ``.cpu()`` / ``.cuda()`` / ``.double()`` /
    ``.float()`` keep the ``Parameter`` objects and swap their ``.data``) or a ``requires_grad`` flag changed (freeze / unfreeze)."""

    def __init__(self, max_graphs=4096):
        self.max_graphs = max_graphs
        self._states = {}          # key -> _CallState
        self._by_obj = {}          # id(batch) -> (weakref, tensor signature, key): the same tensors again need no fingerprint
        self._fp = None
        self._param_key = None     # (id, requires_grad) of every parameter the captures were made with
        self._all = None           # every parameter, in module order
        self._params = self._first = None
        self._probe = None         # (data_ptr, requires_grad) of every parameter at the last call

    def __deepcopy__(self, memo):
        return GraphedCallable(self.max_graphs)      # graphs and static tensors belong to the original module's parameters

    def __getstate__(self):
        return {"max_graphs": self.max_graphs}

    def __setstate__(self, state):
        self.__init__(state.get("max_graphs", 4096))

    def graphs(self):
        return sum((s.fwd is not None) + (s.bwd is not None) for s in self._states.values())

    def clear(self):
        self._states.clear()
        self._by_obj.clear()

    # -- recognition -----------------------------------------------------------------------------------------------------------
    def _fingerprint(self, tensors):
        from . import _lib
        lib = _lib.load()
        dev = tensors[0].device
        if self._fp is None or self._fp.device != dev:
            self._fp = torch.empty(1, dtype=torch.int64, device=dev)
        n = len(tensors)
        bufs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
        sizes = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in tensors])
        _lib.check(lib.glam_batch_fingerprint(n, bufs, sizes, self._fp.data_ptr(), _lib.stream()), "glam_batch_fingerprint")
        return int(self._fp.item())                  # (the one read-back of a recognised call)

    def _key(self, module, datas, index):
        flags = (module.training, torch.is_grad_enabled(), _route_signature())
        okey = tuple(id(d) for d in datas)
        ent = self._by_obj.get(okey)
        # the same batch OBJECT(S) holding the same index tensor OBJECTS, unwritten since (addresses alone would not do: a freed tensor's
        # address comes back with other content)
        if (ent is not None and all(r() is d for r, d in zip(ent[0], datas))
                and all(r() is t and v == t._version for (r, v), t in zip(ent[1], index))):
            return ent[2][:-1] + (flags,)
        sig = tuple((weakref.ref(t), t._version) for t in index)
        fp = tuple(self._fingerprint(index[k:k + 3]) for k in range(0, len(index), 3))
        key = (tuple(tuple(d.x.shape) for d in datas), tuple(tuple(t.shape) for t in index),
               tuple(getattr(d, "num_graphs", None) for d in datas), fp, flags)
        try:
            refs = tuple(weakref.ref(d, lambda _r, k=okey, c=self._by_obj: c.pop(k, None)) for d in datas)
            self._by_obj[okey] = (refs, sig, key)
        except TypeError:                            # a batch type without weak references: fingerprinted every time
            pass
        return key

    # -- the route ---------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _fields(data):
        return tuple(getattr(data, k, None) for k in ("x", "edge_index", "edge_attr", "batch"))

    def __call__(self, module, eager_forward, *datas):
        """``eager_forward(*datas)``; every argument is a batch (``x``, ``edge_index``, ``edge_attr``, ``batch``): one for ``Architecture``,
        ligand + protein for ``ArchitectureDTI``, two drugs for ``ArchitectureDDI``."""
        fields = [self._fields(d) for d in datas]
        ok = (GRAPHED_CALL and not _suspended and getattr(module, "graphed_call", True) and len(datas) > 0
              and not module._forward_hooks and not module._forward_pre_hooks and not torch.cuda.is_current_stream_capturing())
        for x, ei, ea, bv in (fields if ok else ()):
            ok = (ok and all(torch.is_tensor(t) for t in (x, ei, ea, bv)) and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
                  and x.numel() > 0 and ei.numel() > 0 and ei.dtype == torch.int64 and bv.dtype == torch.int64 and ea.dtype == torch.float32
                  and ei.is_contiguous() and ea.is_contiguous() and bv.is_contiguous() and x.is_contiguous())
        if not ok:
            return eager_forward(*datas)
        # The captures bake the parameters' addresses and the trainable set in.  Probe per call, without walking the module tree (~25 us):
        # the first parameter object (a model re-built or loaded into new Parameters), and of every cached parameter its storage address
        # and requires_grad — module.to() / .cpu() / .cuda() / .double() / .float() keep the Parameter OBJECTS and swap their .data, a
        # fine-tuning schedule flips requires_grad on the same objects.  Any difference: walk again, and drop every graph unless the walk
        # finds the very same (object, requires_grad, address) set.
        params = self._params
        # (two list comprehensions: 3 us for the default model's 15 parameters; a generator of pairs costs 7)
        probe = None if self._all is None else ([p.data_ptr() for p in self._all], [p.requires_grad for p in self._all])
        if params is None or next(module.parameters(), None) is not self._first or probe != self._probe:
            self._all = tuple(module.parameters())
            self._first = self._all[0] if self._all else None
            params = self._params = tuple(p for p in self._all if p.requires_grad)
            self._probe = ([p.data_ptr() for p in self._all], [p.requires_grad for p in self._all])
            pkey = tuple((id(p), p.requires_grad, p.data_ptr(), p.dtype, p.device) for p in self._all)
            if pkey != self._param_key:
                self.clear()
                self._param_key = pkey
        key = self._key(module, datas, tuple(t for _x, ei, ea, bv in fields for t in (ei, bv, ea)))
        st = self._states.get(key)
        if st is None:
            if len(self._states) >= self.max_graphs:
                return eager_forward(*datas)
            st = self._states[key] = _CallState(None)
        st.visits += 1
        if st.visits == 1:
            # first visit: the caller's own batch, eagerly — exactly what happened before this route existed (its index tensors get
            # their CSR / ELL staging, so a caller that captures a graph of its own around ``model(batch)`` later finds them staged)
            with no_graphed_call():
                return eager_forward(*datas)
        if st.static is None:
            from .data import Batch
            st.static, st.x_sig = [], [None] * len(datas)
            for d, (x, ei, ea, bv) in zip(datas, fields):
                sb = Batch(x=x.detach().clone(), edge_index=ei.clone(), edge_attr=ea.clone(), batch=bv.clone())
                ng = getattr(d, "num_graphs", None)
                if ng is not None:
                    sb.num_graphs = ng
                st.static.append(sb)
        else:
            for k, (x, _ei, _ea, _bv) in enumerate(fields):
                sig = st.x_sig[k]
                if sig is None or sig[0]() is not x or sig[1] != x._version:
                    st.static[k].x.copy_(x)          # (a cached loader hands the same tensor object back, unwritten: nothing to copy)
        # the OBJECT, not its address: a freed tensor's address comes back with other content
        st.x_sig = [(weakref.ref(f[0]), f[0]._version) for f in fields]
        if st.fwd is None:
            free, total = torch.cuda.mem_get_info(fields[0][0].device)
            if st.visits < 3 or free < total // 4:
                # second visit: eagerly on the private static copy (ITS index tensors are staged here: one validation sync)
                with no_graphed_call():
                    return eager_forward(*st.static)
            self._capture(st, module, eager_forward, params)
        if st.bwd is None:
            st.fwd.replay()
            return st.out.detach().clone()
        if st.pending is not None and st.pending() is not None:
            # an earlier output of this content is alive and has not been back-propagated: its backward needs the static activations a
            # replay would overwrite — this forward runs eagerly on the caller's batch (its own activations, its own autograd graph)
            with no_graphed_call():
                return eager_forward(*datas)
        return _Replay.apply(st, *st.params)

    def _capture(self, st, module, eager_forward, params):
        grad = torch.is_grad_enabled() and len(params) > 0
        # a NEW features tensor for the capture: whatever the eager visit derived from the old one and cached on it (the zero-padded copy
        # of the atom features, ops.pad_cols) must be recomputed INSIDE the graph — later calls bring other features
        for sb in st.static:
            sb.x = sb.x.clone()
        torch.cuda.synchronize()
        fwd = torch.cuda.CUDAGraph()
        if not grad:
            with torch.cuda.graph(fwd), no_graphed_call():
                st.out = eager_forward(*st.static)
            st.fwd = fwd
            return
        # The captured forward runs on PROXY leaves (detached aliases of the parameters: same storage, so optimizer updates show).  A
        # parameter's AccumulateGrad node is bound to the stream it was created on and lives as long as any autograd graph that reaches
        # it — and the reference's loop still holds last iteration's ``loss`` when it calls the model again (trainer.py:296-299): with the
        # parameters themselves as leaves the captured graph would end in nodes of the caller's stream, the backward capture would
        # join that stream, and the runtime crashes at capture end (tools/repro_graphed_call.py, V=4 / V=5).
        names = [n for n, p in module.named_parameters() if p.requires_grad]
        proxies = tuple(p.detach().requires_grad_() for p in params)
        with torch.cuda.graph(fwd), no_graphed_call():
            out = torch.func.functional_call(module, dict(zip(names, proxies)), tuple(st.static))
        st.fwd, st.out = fwd, out
        if out.requires_grad:
            st.gout = torch.zeros_like(out)
            bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(bwd, pool=fwd.pool()):
                grads = torch.autograd.grad(out, proxies, st.gout, allow_unused=True)
            st.bwd, st.params, st.grads = bwd, params, tuple(grads)


def graphed_call(module, eager_forward, *datas):
    """``eager_forward(*datas)`` through ``module``'s graphed-callable route (created on first use, stored outside the module's
    parameters / buffers / submodules: state dicts and ``module.to()`` never see it)."""
    route = module.__dict__.get("_glam_graphed_route")
    if route is None:
        route = module.__dict__["_glam_graphed_route"] = GraphedCallable()
    return route(module, eager_forward, *datas)
