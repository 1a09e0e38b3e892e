"""Student losses on libvmc kernels — same names, arguments and error behaviour as the reference's
losses.py (distillation_loss :5-44, classification_loss :47-67, reconstruction_loss :70-81).

Each loss is ONE fused forward pass that also produces the gradient (K10); the autograd backward only
scales that gradient by the incoming scalar.
"""
from __future__ import annotations

import torch

from . import autograd_ops as ag
from ._lib import check, lib, ptr, stream


class _DistillFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, cosine):
        if student.shape != teacher.shape:
            raise RuntimeError(f"distillation_loss: shape mismatch {tuple(student.shape)} vs {tuple(teacher.shape)}")
        E = student.shape[-1]
        rows = student.numel() // E
        s = student.contiguous().float()
        # teacher may be the strided slice rgb_emb[:, :-1] of train.py:98: rows_per_clip rows out of a longer clip
        if teacher.dtype == torch.float32 and teacher.dim() == 3 and teacher.stride(-1) == 1 and teacher.stride(1) == E:
            t = teacher                                 # strided view read in place: strides of the tensor the kernel gets
            rows_per_clip, clip_stride = t.shape[1], t.stride(0)
        else:                                           # other dtypes (f16 / f64 stores): a dense fp32 copy, dense strides
            t = teacher.contiguous().float()
            rows_per_clip, clip_stride = rows, rows * E
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        ds = torch.empty_like(s) if student.requires_grad else None
        nbytes = lib.vmc_loss_workspace_bytes(rows)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=s.device)
        check(lib.vmc_distill_loss(ptr(s), ptr(t), ptr(loss), ptr(ds), rows, E, rows_per_clip, clip_stride, int(cosine),
                                   ptr(ws), nbytes, stream()), "distill_loss")
        ctx.save_for_backward(ds)
        ctx.shape = student.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return ag.scale_by_device_scalar(ds, g).view(ctx.shape), None, None


def distillation_loss(student_embeddings, teacher_embeddings, mode="mse"):
    """losses.py:5-44.  mode 'mse' or 'cosine' (safe cosine with eps clamps); anything else raises ValueError."""
    if mode not in ("mse", "cosine"):
        raise ValueError(f"Unsupported mode '{mode}'. Choose 'mse' or 'cosine'.")
    return _DistillFn.apply(student_embeddings, teacher_embeddings, mode == "cosine")


class _BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, pw):
        x = logits.contiguous().float()
        y = targets.contiguous().float()
        if x.shape != y.shape:
            raise ValueError(f"Target size ({tuple(y.shape)}) must be the same as input size ({tuple(x.shape)})")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if logits.requires_grad else None
        ws = torch.empty(64, dtype=torch.float32, device=x.device)
        check(lib.vmc_bce_loss(ptr(x), ptr(y), ptr(loss), ptr(dx), x.numel(), float(pw), ptr(ws), 256, stream()), "bce_loss")
        ctx.save_for_backward(dx)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return ag.scale_by_device_scalar(dx, g).view(ctx.shape), None, None


def classification_loss(predictions, targets, positive_weight=None):
    """losses.py:47-67: BCE-with-logits, per-element pos_weight = positive_weight * targets + 1 (so the
    effective positive weight is positive_weight + 1), mean over all elements."""
    return _BceFn.apply(predictions, targets, -1.0 if positive_weight is None else float(positive_weight))


def bce_with_logits_loss(predictions, targets):
    """nn.BCEWithLogitsLoss() as used by TFAM/train_and_eval.py:58."""
    return _BceFn.apply(predictions, targets, -1.0)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        x = logits.contiguous().float()
        if x.dim() != 2:
            raise ValueError(f"cross_entropy_loss expects [rows, classes] logits, got {tuple(x.shape)}")
        rows, C = x.shape
        if target.is_floating_point():
            if target.shape != x.shape:
                raise ValueError(f"probability targets must have the logits' shape {tuple(x.shape)}, got {tuple(target.shape)}")
            tidx, tprob = None, target.contiguous().float()
        else:
            if target.shape != (rows,):
                raise ValueError(f"index targets must have shape ({rows},), got {tuple(target.shape)}")
            tidx, tprob = target.contiguous().to(torch.int64), None
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if logits.requires_grad else None
        ws = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.vmc_cross_entropy_loss(ptr(x), ptr(tidx), ptr(tprob), ptr(loss), ptr(dx), rows, C, ptr(ws), rows * 4, stream()),
              "cross_entropy_loss")
        ctx.save_for_backward(dx)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return ag.scale_by_device_scalar(dx, g).view(ctx.shape), None


def cross_entropy_loss(predictions, targets):
    """nn.CrossEntropyLoss() (mean) as used by the MammalNet variants: class-index targets
    (train_frame_diff_mn.py:102 ``labels.argmax(dim=1)``) or float probability rows
    (TFAM/train_and_eval_frame_diff_MN.py:83 passes the one-hot ``labels`` directly)."""
    return _CrossEntropyFn.apply(predictions, targets)


def reconstruction_loss(reconstruction, input):  # noqa: A002  (reference signature)
    """losses.py:70-81."""
    raise NotImplementedError


def loss_and_grad(criterion, logits, targets):
    """(loss, d loss / d logits) of a mean-reduced criterion of this module WITHOUT autograd nodes: the loss kernels produce the
    gradient in the same launch, and ``logits.backward(dlogits)`` is what ``loss.backward()`` computes (implicit upstream gradient 1)
    minus two launches (autograd's ones fill, the scaling by it) -- the training loops of TFAM/train_and_eval.py:81-83 spelled for a
    launch-bound step.  criterion: bce_with_logits_loss or cross_entropy_loss."""
    x = logits.detach().contiguous().float()
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    if criterion is bce_with_logits_loss:
        y = targets.contiguous().float()
        if x.shape != y.shape:
            raise ValueError(f"Target size ({tuple(y.shape)}) must be the same as input size ({tuple(x.shape)})")
        ws = torch.empty(64, dtype=torch.float32, device=x.device)
        check(lib.vmc_bce_loss(ptr(x), ptr(y), ptr(loss), ptr(dx), x.numel(), -1.0, ptr(ws), 256, stream()), "bce_loss")
    elif criterion is cross_entropy_loss:
        rows, C = x.shape
        tidx, tprob = (None, targets.contiguous().float()) if targets.is_floating_point() else (targets.contiguous().to(torch.int64), None)
        ws = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.vmc_cross_entropy_loss(ptr(x), ptr(tidx), ptr(tprob), ptr(loss), ptr(dx), rows, C, ptr(ws), rows * 4, stream()),
              "cross_entropy_loss")
    else:
        raise ValueError("loss_and_grad: criterion must be bce_with_logits_loss or cross_entropy_loss")
    return loss, dx.view(logits.shape)
