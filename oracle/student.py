"""Oracle: MoCLIP student model + losses + train step, fp32 (test infrastructure only).

Follows models/student_model.py:8-98, losses.py:5-67 and train.py:89-107.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import vit


def residual_mlp(sd, x, alpha, prefix="residual_mlp."):
    """models/student_model.py:27-35: x + alpha * fc2(GELU_erf(fc1(x)))."""
    h = F.gelu(x @ sd[prefix + "fc1.weight"].t() + sd[prefix + "fc1.bias"])
    return x + alpha * (h @ sd[prefix + "fc2.weight"].t() + sd[prefix + "fc2.bias"])


def student_forward(sd, flow_videos_u8, heads, alpha=0.1, wrap_quirk=True):
    """models/student_model.py:61-98.  flow_videos [B,T,3,H,W] u8 with H=W=input resolution.

    Returns (embeddings [B,T,E], embeddings_for_distillation [B,T,E], logits [B,C]).
    ``wrap_quirk`` reproduces the float->to_pil_image wrap-around of :74,:78.
    """
    B, T, C, H, W = flow_videos_u8.shape
    frames = flow_videos_u8.reshape(B * T, C, H, W)
    if wrap_quirk:
        frames = vit.to_pil_wrap_u8(frames)
    pix = vit.normalize_u8(frames)
    emb = vit.vit_forward(sd, pix, heads, prefix="visual_encoder.").view(B, T, -1)   # :84-87
    emb_d = residual_mlp(sd, emb, alpha)                                               # :90
    pooled = emb.mean(dim=1)                                                           # :93
    h = torch.relu(pooled @ sd["classification_head.0.weight"].t() + sd["classification_head.0.bias"])
    logits = h @ sd["classification_head.2.weight"].t() + sd["classification_head.2.bias"]  # :96
    return emb, emb_d, logits


def distillation_loss(student, teacher, mode="mse"):
    """losses.py:5-44."""
    if mode == "mse":
        return ((student - teacher) ** 2).mean()
    if mode != "cosine":
        raise ValueError(f"Unsupported mode '{mode}'. Choose 'mse' or 'cosine'.")
    eps = 1e-5
    sn = torch.linalg.vector_norm(student, dim=-1).clamp(min=eps)   # norm has subgradient 0 at 0
    tn = torch.linalg.vector_norm(teacher, dim=-1).clamp(min=eps)
    cos = (student * teacher).sum(-1) / (sn * tn)
    cos = cos.clamp(-1 + eps, 1 - eps)
    return (1 - cos).mean()


def classification_loss(pred, targets, positive_weight=None):
    """losses.py:47-67: BCE-with-logits, per-element pos_weight = pw*y + 1, mean over B*C."""
    y = targets.to(torch.float32)
    w = torch.ones_like(y) if positive_weight is None else positive_weight * y + 1.0
    # l = (1-y) x + (1 + (w-1) y) (log(1+exp(-|x|)) + max(-x,0))   
    lw = 1 + (w - 1) * y
    loss = (1 - y) * pred + lw * (torch.log1p(torch.exp(-pred.abs())) + torch.clamp(-pred, min=0))
    return loss.mean()


def cross_entropy_loss(pred, targets):
    """nn.CrossEntropyLoss() (mean) of the MammalNet variants: train_frame_diff_mn.py:82,102 (class indices) and
    TFAM/train_and_eval_frame_diff_MN.py:59,83 (float one-hot rows = probability targets).  Written out (log-softmax) so
    the oracle does not lean on the fused torch op it checks."""
    logp = pred - torch.logsumexp(pred, dim=1, keepdim=True)
    if targets.is_floating_point():
        return -(targets * logp).sum(dim=1).mean()
    return -logp.gather(1, targets.view(-1, 1).long()).mean()


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False):
    """torch.optim.Adam (train.py:66) / AdamW (TFAM/train_and_eval.py:53) single-tensor update,
    PyTorch default formulation: p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    if decoupled and weight_decay != 0.0:
        p = p * (1 - lr * weight_decay)
    elif weight_decay != 0.0:
        g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
