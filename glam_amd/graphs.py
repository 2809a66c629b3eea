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
order; the kernels are deterministic).  Requirements: an optimizer created with ``capturable=True`` (Adam/AdamW), a
loss function of ``(output, batch)`` that stays on the device, and no data-dependent Python control flow in the model.
"""
from __future__ import annotations

import weakref

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, max_graphs=4096):
        for g in optimizer.param_groups:
            if g.get("capturable") is False and not g.get("fused"):
                raise ValueError("GraphedTrainStep needs an optimizer created with capturable=True (or fused=True, capturable=True)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.max_graphs = max_graphs
        self._state = {}      # id(batch) -> [weakref, visits, graph, static_loss]
        self._pool = None

    def _step(self, batch):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(batch), batch)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, batch):
        """One optimizer step on ``batch``; returns the (device) loss tensor of this step."""
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
        return st[3]

    def graphs(self):
        return sum(1 for s in self._state.values() if s[2] is not None)


class GraphedForward:
    """Inference counterpart: ``GraphedForward(model)(batch)`` runs ``model(batch)`` under ``torch.no_grad()`` — eagerly on
    the first visit of a batch object, from a hipGraph afterwards (the evaluation loaders of the reference iterate the same
    batches after every epoch: ``src_1gp/trainer.py:39-41``).  The returned tensor is the graph's static output: read or
    copy it before the next call on the same batch.  Parameters may change between calls (training in between): the
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
        return st[2]
