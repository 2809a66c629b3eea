"""The optimizer of the training loop that drives the path: ``Adam(self.model.parameters(), lr=args.lr)`` (reference
``src_1gp/trainer.py:49-50``, stepped at ``trainer.py:301``, its learning rate moved by ``ReduceLROnPlateau``, ``trainer.py:55,85``).

``glam_amd.optim.Adam`` is a ``torch.optim.Optimizer`` with ``torch.optim.Adam``'s constructor arguments, ``param_groups`` and
``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so schedulers and checkpoints work unchanged.  Its step is
ONE HIP launch over all parameter tensors (``glam_adam_step``: 4 µs where the library's multi-tensor kernel takes 45 µs on a
default-shaped model — 9 workgroups of double-precision arithmetic), and its host side is a cached address table: 36 ``data_ptr()`` reads
per step instead of the library optimizer's per-step list building.  It is always "capturable": the step count lives on the device and
the launch advances it, a learning rate given as a device tensor is read by the launch (``glam_amd.graphs.GraphedTrainStep`` does that).

Differences from ``torch.optim.Adam``, by design: fp32 CUDA/HIP parameters only; no ``amsgrad`` / ``maximize`` / ``differentiable``;
one step count per parameter GROUP (a parameter without a gradient in some step keeps its moments and still sees the group's bias
correction — the library counts per tensor); arithmetic in fp32 with the bias corrections in double (the library's fused kernel works
in double, its single-tensor path in fp32: all three agree to rounding, tested)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch
from torch.optim import optimizer as _topt      # (the module: its global step-hook tables)

from . import _lib
from ._lib import GlamHipError


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False,
                 capturable=True, differentiable=False, foreach=None, fused=None):
        if amsgrad or maximize or differentiable:
            raise GlamHipError("glam_amd.optim.Adam: amsgrad / maximize / differentiable are not implemented")
        if not (torch.is_tensor(lr) or lr >= 0.0) or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("glam_amd.optim.Adam: invalid hyper-parameters")
        # `capturable` is always true here (device-side step count); the key is kept because GraphedTrainStep and user code look for it
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      capturable=True))
        self._plans = {}          # group index -> _Plan

    class _Plan:
        __slots__ = ("params", "table", "numel", "step", "ticket", "flat_m", "flat_v", "sub", "sub_key", "gptrs", "pptrs")

    def _plan(self, gi, group):
        plan = self._plans.get(gi)
        ps = [p for p in group["params"] if p.requires_grad]
        if plan is not None and len(plan.params) == len(ps) and all(a is b for a, b in zip(plan.params, ps)):
            return plan
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise GlamHipError("glam_amd.optim.Adam: parameters must be contiguous fp32 tensors on a HIP device")
        dev = ps[0].device
        if any(p.device != dev for p in ps):
            raise GlamHipError("glam_amd.optim.Adam: one device per parameter group")
        old = self._plans.get(gi)
        plan = Adam._Plan()
        plan.params = ps
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]                       # every view starts 16-byte aligned
        plan.flat_m = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        plan.flat_v = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        plan.step = torch.zeros((), dtype=torch.float32, device=dev)
        plan.ticket = torch.zeros(544, dtype=torch.int32, device=dev)      # GLAM_ADAM_TICKET_WORDS: main ticket + 16 sub-counters, 128 B apart
        plan.table = np.zeros((len(ps), 4), dtype=np.uint64)
        plan.numel = np.array([p.numel() for p in ps], dtype=np.int64)
        plan.sub, plan.sub_key = None, None
        plan.gptrs, plan.pptrs = None, None          # gradient / parameter addresses as the table holds them (step(): one list compare)
        off = 0
        for i, (p, n) in enumerate(zip(ps, sizes)):
            m, v = plan.flat_m[off:off + p.numel()].view_as(p), plan.flat_v[off:off + p.numel()].view_as(p)
            st = self.state[p]
            if "exp_avg" in st:                                              # adopted state (load_state_dict, a re-grouped parameter)
                m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                if i == 0 or float(st["step"]) > float(plan.step):
                    plan.step.fill_(float(st["step"]))
            st["exp_avg"], st["exp_avg_sq"], st["step"] = m, v, plan.step
            plan.table[i] = (p.data_ptr(), 0, m.data_ptr(), v.data_ptr())
            off += n
        del old
        self._plans[gi] = plan
        return plan

    def zero_grad(self, set_to_none: bool = True):
        """``torch.optim.Optimizer.zero_grad``; its default (``set_to_none=True``) as the plain loop it amounts to — the base class's
        goes through a dynamo guard and a profiler scope, 25 us per call where the reference's loop is bound by host time (its batch of
        32: DESIGN.md §7)."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.grad = None

    def step(self, closure=None):
        # (torch wraps an optimizer's step in a profiler scope that also runs the step hooks — Optimizer.profile_hook_step, ≈15 us per
        #  call; `step.hooked` below keeps it off this class, and the hooks, when there are any, are run here)
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _topt._global_optimizer_pre_hooks or _topt._global_optimizer_post_hooks:
            return Adam._hooked_step(self, closure)
        return self._step(closure)

    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        from . import ops
        ops.parameters_written()         # the launch below writes parameters through raw pointers: no version counter moves
        with torch.no_grad():
            self._launch(lib)
        return loss

    def _launch(self, lib):
        for gi, group in enumerate(self.param_groups):
            if not any(p.requires_grad for p in group["params"]):
                continue
            plan = self._plan(gi, group)
            table, numel = plan.table, plan.numel
            missing = None
            grads = [p.grad for p in plan.params]
            complete = all([g is not None for g in grads])      # (`None in grads` would compare TENSORS with None: a torch call each)
            if complete:
                # the common step: every parameter has a gradient — the addresses against last step's, one list compare each (a loop
                # over numpy elements cost 35 us for the default model's 24 tensors)
                gp, pp = [g.data_ptr() for g in grads], [p.data_ptr() for p in plan.params]
                if gp != plan.gptrs:
                    for i, (g, p) in enumerate(zip(grads, plan.params)):
                        if table[i, 1] != gp[i]:                             # a gradient tensor not seen at this address yet
                            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.is_sparse:
                                raise GlamHipError("glam_amd.optim.Adam: gradients must be dense contiguous fp32 tensors on the parameter's device")
                            table[i, 1] = gp[i]
                    plan.gptrs = gp
                if pp != plan.pptrs:                                         # `p.data = ...` since the last step
                    for i in range(len(pp)):
                        table[i, 0] = pp[i]
                    plan.pptrs = pp
            else:
                plan.gptrs = plan.pptrs = None
            for i, p in (() if complete else enumerate(plan.params)):
                g = p.grad
                if g is None:
                    missing = missing or []
                    missing.append(i)
                    continue
                ptr = g.data_ptr()
                if table[i, 1] != ptr:                                       # a gradient tensor not seen at this address yet
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.is_sparse:
                        raise GlamHipError("glam_amd.optim.Adam: gradients must be dense contiguous fp32 tensors on the parameter's device")
                    table[i, 1] = ptr
                if table[i, 0] != p.data_ptr():                              # `p.data = ...` since the last step
                    table[i, 0] = p.data_ptr()
            if missing:                                                      # parameters without a gradient sit this step out
                if len(missing) == len(plan.params):
                    continue
                keep = np.ones(len(plan.params), dtype=bool)
                keep[missing] = False
                table, numel = np.ascontiguousarray(table[keep]), np.ascontiguousarray(numel[keep])
            lr = group["lr"]
            lr_dev = None
            if torch.is_tensor(lr):
                if lr.is_cuda:
                    if lr.dtype != torch.float32 or lr.numel() != 1:
                        raise GlamHipError("glam_amd.optim.Adam: a device learning rate must be one fp32 element")
                    lr_dev, lr = lr, 0.0
                else:
                    lr = float(lr)
            b1, b2 = group["betas"]
            rc = lib.glam_adam_step(table.ctypes.data, numel.ctypes.data, len(numel), plan.step.data_ptr(), plan.ticket.data_ptr(),
                                    lr_dev.data_ptr() if lr_dev is not None else None, float(lr), float(b1), float(b2), float(group["eps"]),
                                    float(group["weight_decay"]), torch.cuda.current_stream(plan.step.device).cuda_stream)
            if rc != 0:
                raise GlamHipError(f"glam_adam_step failed (code {rc}): {lib.glam_last_error().decode()}")

    def state_dict(self):
        """``torch.optim.Adam``'s layout with a PRIVATE ``step`` per parameter: internally every parameter of a group shares one device
        counter, and a checkpoint that kept the sharing would, loaded into ``torch.optim.Adam`` (capturable / fused), be advanced once
        per parameter per step by ``_foreach_add_``."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone()
        return sd

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plans = {}                 # raw device addresses: never carried over by pickling / copying

    def __deepcopy__(self, memo):
        # the plans hold raw addresses of the ORIGINAL moment buffers: a copy rebuilds its own from its (deep-copied) state
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, {} if k == "_plans" else copy.deepcopy(v, memo))
        for gi, group in enumerate(new.param_groups):
            if any(p.requires_grad for p in group["params"]) and any("exp_avg" in new.state.get(p, {}) for p in group["params"]):
                new._plan(gi, group)
        return new

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # adopt the loaded per-parameter tensors into fresh flat buffers NOW: the base class does not copy tensors that already have the
        # parameter's dtype and device, so until then this optimizer's state aliases the one the dictionary came from
        self._plans.clear()
        for gi, group in enumerate(self.param_groups):
            if any(p.requires_grad for p in group["params"]):
                self._plan(gi, group)

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_plans"):
            self._plans.clear()


# torch wraps `cls.step` in Optimizer.profile_hook_step when the first instance is built — unless the function says it is hooked already:
# Adam.step runs the wrapper itself, and only when a step hook is registered (see there)
Adam._hooked_step = torch.optim.Optimizer.profile_hook_step(Adam._step)
Adam.step.hooked = True
