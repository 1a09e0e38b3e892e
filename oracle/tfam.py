"""Oracle: TFAM fusion block (AMO_CLIP), fp32, eval mode (test infrastructure only).

Follows TFAM/models/AMO_CLIP.py:6-171; the multi-head attention arithmetic is the container's
torch.nn.functional.multi_head_attention_forward (functional.py:6206-6640): q scaled by dh^-1/2,
key-padding mask added as -inf, softmax, out_proj.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def mha(sd, prefix, q_in, kv_in, nhead, key_padding_mask=None):
    """nn.MultiheadAttention(batch_first=True) forward, eval mode (AMO_CLIP.py:19-20,39,43)."""
    W, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    D = q_in.shape[-1]
    dh = D // nhead
    B, Tq, _ = q_in.shape
    Tk = kv_in.shape[1]
    q = q_in @ W[:D].t() + b[:D]
    k = kv_in @ W[D:2 * D].t() + b[D:2 * D]
    v = kv_in @ W[2 * D:].t() + b[2 * D:]
    q = q.view(B, Tq, nhead, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.view(B, Tk, nhead, dh).transpose(1, 2)
    v = v.view(B, Tk, nhead, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                          # [B,H,Tq,Tk]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.view(B, 1, 1, Tk), float("-inf"))
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, Tq, D)
    return o @ sd[prefix + "out_proj.weight"].t() + sd[prefix + "out_proj.bias"]


def _ln(sd, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)


def attention_layer(sd, prefix, x, nhead, cross_src=None, src_kpm=None, cross_kpm=None):
    """AttentionLayer.forward, AMO_CLIP.py:37-51 (dropouts are identities in eval)."""
    x = _ln(sd, prefix + "norm_self.", x + mha(sd, prefix + "self_attn.", x, x, nhead, src_kpm))
    if cross_src is not None:
        x = _ln(sd, prefix + "norm_cross.", x + mha(sd, prefix + "cross_attn.", x, cross_src, nhead, cross_kpm))
    h = torch.relu(x @ sd[prefix + "ffn.0.weight"].t() + sd[prefix + "ffn.0.bias"])   # ReLU: :26 (activation arg never forwarded)
    h = h @ sd[prefix + "ffn.3.weight"].t() + sd[prefix + "ffn.3.bias"]
    return _ln(sd, prefix + "norm_ffn.", x + h)


def positional_encoding(seq_len, d_model):
    """AMO_CLIP.py:88-97."""
    position = torch.arange(seq_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(seq_len, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def amo_clip_forward(sd, rgb_emb, motion_emb, mask_rgb=None, mask_flow=None, *, nhead=8,
                     use_cross_attention=True, use_pe=False, use_only_rgb=False, use_only_flow=False,
                     concat_dim=1):
    """AMO_CLIP.forward, AMO_CLIP.py:99-171 (eval).  Inputs are not modified (the reference adds the
    PE in place, :133-134; the returned logits are the same)."""
    num_layers = 0
    while f"layers.{num_layers}.norm_self.weight" in sd:
        num_layers += 1
    D = rgb_emb.shape[-1]
    attn_rgb = ~mask_rgb if mask_rgb is not None else None
    attn_flow = ~mask_flow if mask_flow is not None else None
    if use_pe:
        rgb_emb = rgb_emb + positional_encoding(rgb_emb.size(1), D).unsqueeze(0)
        motion_emb = motion_emb + positional_encoding(motion_emb.size(1), D).unsqueeze(0)
    if use_only_rgb:
        x = rgb_emb
        for i in range(num_layers):
            x = attention_layer(sd, f"layers.{i}.", x, nhead, src_kpm=attn_rgb)
    elif use_only_flow:
        x = motion_emb
        for i in range(num_layers):
            x = attention_layer(sd, f"layers.{i}.", x, nhead, src_kpm=attn_flow)
    elif use_cross_attention:
        x = rgb_emb
        for i in range(num_layers):
            x = attention_layer(sd, f"layers.{i}.", x, nhead, cross_src=motion_emb, src_kpm=attn_rgb,
                                cross_kpm=attn_flow)
    else:
        rgb_emb = rgb_emb[:, :-1, :]
        attn_rgb = attn_rgb[:, :-1]
        if concat_dim == 1:
            attn_mask = torch.cat([attn_rgb, attn_flow], dim=1)
            x = torch.cat([rgb_emb, motion_emb], dim=1)
        else:
            attn_mask = attn_flow
            x = torch.cat([rgb_emb, motion_emb], dim=-1)
            x = x @ sd["projection_layer.weight"].t() + sd["projection_layer.bias"]
        for i in range(num_layers):
            x = attention_layer(sd, f"layers.{i}.", x, nhead, src_kpm=attn_mask)
    pooled = x.mean(dim=1)                                               # :170 — includes padded rows
    h = _ln(sd, "classifier.0.", pooled)
    h = F.gelu(h @ sd["classifier.1.weight"].t() + sd["classifier.1.bias"])
    return h @ sd["classifier.4.weight"].t() + sd["classifier.4.bias"]


def bce_with_logits_mean(x, y):
    """nn.BCEWithLogitsLoss() (TFAM/train_and_eval.py:58)."""
    return (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()


def cosine_lr(epoch, epochs, base_lr=1e-4, eta_min=1e-6):
    """CosineAnnealingLR closed form (TFAM/train_and_eval.py:54-56,162)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / epochs)) / 2
