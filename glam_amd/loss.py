"""The loss of the training step, mirroring the reference's ``loss.py`` (``get_loss``, ``src_1gp/loss.py:40-59``).

``get_loss('mse')`` / ``get_loss('bcel')`` — the criteria of the regression and classification trainers (``trainer.py:296`` and
``trainer.py:244-245``) — return modules with ``nn.MSELoss`` / ``nn.BCEWithLogitsLoss`` semantics (mean over all elements) whose value
and gradient come from ONE HIP launch (``glam_loss_fwd``) and whose backward is one scale launch; through torch the same is 7–10 small
launches per step.  ``MaskedBCEWithLogitsLoss`` is the classification trainer's ``criterion(y_score[y_true >= 0], y_true[y_true >= 0])``
(labels of -1 are missing, ``dataset.py:138``) without the boolean indexing, which a hipGraph cannot capture.  Every other name of the
reference's table maps to the same torch module it maps to there."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import GlamHipError, f32c, ptr, require_device

_TICKETS: dict = {}     # device index -> persistent u32[544] ticket buffer (zeroed once, re-armed by every launch)


def _ticket(dev):
    t = _TICKETS.get(dev.index)
    if t is None:
        t = _TICKETS[dev.index] = torch.zeros(544, dtype=torch.int32, device=dev)
    return t


class _MeanLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind, masked):
        require_device(pred, target)
        if pred.shape != target.shape:
            raise GlamHipError(f"loss: prediction {tuple(pred.shape)} and target {tuple(target.shape)} differ in shape")
        # classification labels arrive as a LongTensor (src_1gp/dataset.py:139; the reference casts with .float() at the call site,
        # trainer.py:244-245): cast here so that criterion(y_score, y_true) is the drop-in (the mask y >= 0 survives the cast)
        if target.dtype != torch.float32:
            target = target.to(torch.float32)
        p, t = f32c(pred, "prediction"), f32c(target, "target")
        lib, dev, n = _lib.load(), pred.device, pred.numel()
        out = torch.empty(2, dtype=torch.float32, device=dev)             # loss | 1 / count
        grad = torch.empty_like(p)
        ws = torch.empty(lib.glam_loss_workspace_bytes(), dtype=torch.uint8, device=dev) if n > 1024 else None
        rc = lib.glam_loss_fwd(ptr(p), ptr(t), n, kind, int(masked), out.data_ptr(), out.data_ptr() + 4, ptr(grad), ptr(ws),
                               ws.numel() if ws is not None else 0, ptr(_ticket(dev)) if n > 1024 else None,
                               torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise GlamHipError(f"glam_loss_fwd failed (code {rc}): {lib.glam_last_error().decode()}")
        ctx.save_for_backward(grad, out)
        ctx.shape = pred.shape
        return out[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_up):
        grad, out = ctx.saved_tensors
        lib, dev = _lib.load(), grad.device
        g_up = f32c(g_up.reshape(1), "loss gradient")
        d_pred = torch.empty_like(grad)
        rc = lib.glam_loss_bwd(ptr(grad), out.data_ptr() + 4, ptr(g_up), grad.numel(), ptr(d_pred), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise GlamHipError(f"glam_loss_bwd failed (code {rc}): {lib.glam_last_error().decode()}")
        return d_pred.view(ctx.shape), None, None, None


def mse_loss(pred, target):
    """``F.mse_loss(pred, target)`` (mean)."""
    return _MeanLoss.apply(pred, target, 0, False)


def bce_with_logits(pred, target, masked=False):
    """``F.binary_cross_entropy_with_logits(pred, target)`` (mean); ``masked``: over the elements with ``target >= 0`` only."""
    return _MeanLoss.apply(pred, target, 1, masked)


class MSELoss(torch.nn.Module):
    def forward(self, input, target):
        return mse_loss(input, target)


class BCEWithLogitsLoss(torch.nn.Module):
    def forward(self, input, target):
        return bce_with_logits(input, target)


class MaskedBCEWithLogitsLoss(torch.nn.Module):
    """``criterion(y_score[y_true >= 0], y_true[y_true >= 0].float())`` of ``trainer.py:244-245`` as ``criterion(y_score, y_true)``."""

    def forward(self, input, target):
        return bce_with_logits(input, target, masked=True)


def get_loss(loss_str):
    """``loss.py:40-59``: the same names; 'mse' and 'bcel' on the HIP launch, 'bcel_masked' in addition, the rest as in the reference
    ('focal' / 'mtce' are the reference's own small modules over torch functions and are not on the training loops' default path)."""
    d = {
        'mse': MSELoss, 'bcel': BCEWithLogitsLoss, 'bcel_masked': MaskedBCEWithLogitsLoss,
        'mae': torch.nn.L1Loss, 'huber': torch.nn.SmoothL1Loss, 'smae': torch.nn.SmoothL1Loss, 'bce': torch.nn.BCELoss,
        'bcen': lambda: torch.nn.BCELoss(reduction="none"), 'bceln': lambda: torch.nn.BCEWithLogitsLoss(reduction="none"),
        'kl': torch.nn.KLDivLoss, 'hinge': torch.nn.HingeEmbeddingLoss, 'nll': torch.nn.NLLLoss, 'ce': torch.nn.CrossEntropyLoss,
    }
    if loss_str not in d:
        raise ValueError('loss not found')
    return d[loss_str]()
