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
            return self.model(*batches)
        if st[1] is None:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self._pool):
                st[2] = self.model(*batches)
            if self._pool is None:
                self._pool = graph.pool()
            st[1] = graph
        st[1].replay()
        return st[2].clone()
