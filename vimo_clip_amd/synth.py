"""Counter-based synthetic data / weight generator.

Every value is a pure function of (seed, stream name, element index), so the same tensors are
produced on any machine and any torch version (``torch.manual_seed`` streams are not stable across
versions/devices).  Used by bench.py, the tests and oracle/make_golden.py; there are no datasets or
checkpoints offline (SURVEY.md §7 "Hard parts"), so all weights are random-init with the OpenAI-clip
initialisation scales and all frames are uniform u8 noise.
"""
from __future__ import annotations

import hashlib
import math

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def _stream_key(seed: int, name: str) -> np.uint64:
    h = hashlib.blake2b(f"{seed}/{name}".encode(), digest_size=8).digest()
    return np.uint64(int.from_bytes(h, "little"))


def bits(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n uint64 words of stream (seed, name); ``lane`` selects an independent sub-stream."""
    key = _stream_key(seed, name)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(lane)
        return _mix64(_mix64(ctr ^ key) + key)


def uniform(seed: int, name: str, shape, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(seed, name, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = ((bits(seed, name, n, 0) >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / (1 << 53))
    u2 = (bits(seed, name, n, 1) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy((mean + std * z).astype(np.float32).reshape(shape))


def randint_u8(seed: int, name: str, shape) -> torch.Tensor:
    n = int(np.prod(shape))
    words = bits(seed, name, (n + 7) // 8)
    return torch.from_numpy(words.view(np.uint8)[:n].copy().reshape(shape))


def randint(seed: int, name: str, shape, lo: int, hi: int) -> torch.Tensor:
    """integers in [lo, hi) (int64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    w = bits(seed, name, n) >> np.uint64(16)
    return torch.from_numpy((w % np.uint64(hi - lo)).astype(np.int64).reshape(shape) + lo)


# --------------------------------------------------------------------------------------------------
# CLIP ViT geometries and OpenAI-clip-style random initialisation
# --------------------------------------------------------------------------------------------------
VIT_GEOMETRY = {
    # name: (resolution, patch, width, layers, heads, output_dim)
    "ViT-B/32": (224, 32, 768, 12, 12, 512),
    "ViT-B/16": (224, 16, 768, 12, 12, 512),
    "ViT-L/14": (224, 14, 1024, 24, 16, 768),
    # tiny geometries for CPU-sized parity tests (dh = 64 like the real ones)
    "ViT-tiny/32": (64, 32, 128, 2, 2, 64),
    "ViT-tiny/16": (64, 16, 128, 2, 2, 64),
    "ViT-tiny/14": (56, 14, 128, 3, 2, 96),
}


def vit_state_dict(name: str, seed: int, stress: float = 1.0) -> dict:
    """Random ViT weights under the OpenAI ``clip.model.VisionTransformer`` parameter names.

    Scales follow ``clip.model.CLIP.initialize_parameters`` (attn_std = width^-0.5,
    proj_std = width^-0.5 (2L)^-0.5, fc_std = (2 width)^-0.5); biases and LayerNorm affine terms are
    perturbed so that parity tests exercise them.  ``stress`` multiplies the matrix scales to get
    larger-norm activations (representative reduced-precision error).
    """
    R, p, D, L, H, E = VIT_GEOMETRY[name]
    g = R // p
    N = g * g + 1
    sd = {}
    scale = D ** -0.5
    attn_std = D ** -0.5 * stress
    proj_std = (D ** -0.5) * ((2 * L) ** -0.5) * stress
    fc_std = (2 * D) ** -0.5 * stress
    sd["conv1.weight"] = normal(seed, "conv1.weight", (D, 3, p, p), std=(3 * p * p) ** -0.5 * stress)
    sd["class_embedding"] = normal(seed, "class_embedding", (D,), std=scale)
    sd["positional_embedding"] = normal(seed, "positional_embedding", (N, D), std=scale)
    for ln in ("ln_pre", "ln_post"):
        sd[f"{ln}.weight"] = normal(seed, f"{ln}.weight", (D,), std=0.1, mean=1.0)
        sd[f"{ln}.bias"] = normal(seed, f"{ln}.bias", (D,), std=0.05)
    for i in range(L):
        pre = f"transformer.resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            sd[pre + f"{ln}.weight"] = normal(seed, pre + f"{ln}.weight", (D,), std=0.1, mean=1.0)
            sd[pre + f"{ln}.bias"] = normal(seed, pre + f"{ln}.bias", (D,), std=0.05)
        sd[pre + "attn.in_proj_weight"] = normal(seed, pre + "attn.in_proj_weight", (3 * D, D), std=attn_std)
        sd[pre + "attn.in_proj_bias"] = normal(seed, pre + "attn.in_proj_bias", (3 * D,), std=0.02)
        sd[pre + "attn.out_proj.weight"] = normal(seed, pre + "attn.out_proj.weight", (D, D), std=proj_std)
        sd[pre + "attn.out_proj.bias"] = normal(seed, pre + "attn.out_proj.bias", (D,), std=0.02)
        sd[pre + "mlp.c_fc.weight"] = normal(seed, pre + "mlp.c_fc.weight", (4 * D, D), std=fc_std)
        sd[pre + "mlp.c_fc.bias"] = normal(seed, pre + "mlp.c_fc.bias", (4 * D,), std=0.02)
        sd[pre + "mlp.c_proj.weight"] = normal(seed, pre + "mlp.c_proj.weight", (D, 4 * D), std=proj_std)
        sd[pre + "mlp.c_proj.bias"] = normal(seed, pre + "mlp.c_proj.bias", (D,), std=0.02)
    sd["proj"] = normal(seed, "proj", (D, E), std=scale)
    return sd


def student_state_dict(name: str, seed: int, num_classes: int = 140, zero_fc2: bool = False) -> dict:
    """Weights for FlowStudentModel (models/student_model.py:38-59) under its state_dict keys."""
    E = VIT_GEOMETRY[name][5]
    sd = {"visual_encoder." + k: v for k, v in vit_state_dict(name, seed).items()}
    k = E ** -0.5
    sd["residual_mlp.fc1.weight"] = uniform(seed, "rmlp.fc1.w", (E, E), -k, k)
    sd["residual_mlp.fc1.bias"] = uniform(seed, "rmlp.fc1.b", (E,), -k, k)
    if zero_fc2:  # reference init (student_model.py:24-25)
        sd["residual_mlp.fc2.weight"] = torch.zeros(E, E)
        sd["residual_mlp.fc2.bias"] = torch.zeros(E)
    else:
        sd["residual_mlp.fc2.weight"] = uniform(seed, "rmlp.fc2.w", (E, E), -k, k)
        sd["residual_mlp.fc2.bias"] = uniform(seed, "rmlp.fc2.b", (E,), -k, k)
    sd["classification_head.0.weight"] = uniform(seed, "head.0.w", (E // 2, E), -k, k)
    sd["classification_head.0.bias"] = uniform(seed, "head.0.b", (E // 2,), -k, k)
    k2 = (E // 2) ** -0.5
    sd["classification_head.2.weight"] = uniform(seed, "head.2.w", (num_classes, E // 2), -k2, k2)
    sd["classification_head.2.bias"] = uniform(seed, "head.2.b", (num_classes,), -k2, k2)
    return sd


def tfam_state_dict(d_model: int, nhead: int, num_layers: int, dim_feedforward: int, num_classes: int,
                    seed: int) -> dict:
    """Weights for AMO_CLIP (TFAM/models/AMO_CLIP.py:81-86) under its state_dict keys."""
    D, ff, C = d_model, dim_feedforward, num_classes
    sd = {}
    xav = math.sqrt(6.0 / (D + 3 * D))
    for i in range(num_layers):
        pre = f"layers.{i}."
        for att in ("self_attn", "cross_attn"):
            sd[pre + f"{att}.in_proj_weight"] = uniform(seed, pre + att + ".ipw", (3 * D, D), -xav, xav)
            sd[pre + f"{att}.in_proj_bias"] = normal(seed, pre + att + ".ipb", (3 * D,), std=0.02)
            k = D ** -0.5
            sd[pre + f"{att}.out_proj.weight"] = uniform(seed, pre + att + ".opw", (D, D), -k, k)
            sd[pre + f"{att}.out_proj.bias"] = normal(seed, pre + att + ".opb", (D,), std=0.02)
        k = D ** -0.5
        sd[pre + "ffn.0.weight"] = uniform(seed, pre + "ffn.0.w", (ff, D), -k, k)
        sd[pre + "ffn.0.bias"] = uniform(seed, pre + "ffn.0.b", (ff,), -k, k)
        k = ff ** -0.5
        sd[pre + "ffn.3.weight"] = uniform(seed, pre + "ffn.3.w", (D, ff), -k, k)
        sd[pre + "ffn.3.bias"] = uniform(seed, pre + "ffn.3.b", (D,), -k, k)
        for ln in ("norm_self", "norm_cross", "norm_ffn"):
            sd[pre + f"{ln}.weight"] = normal(seed, pre + ln + ".w", (D,), std=0.1, mean=1.0)
            sd[pre + f"{ln}.bias"] = normal(seed, pre + ln + ".b", (D,), std=0.05)
    sd["classifier.0.weight"] = normal(seed, "cls.0.w", (D,), std=0.1, mean=1.0)
    sd["classifier.0.bias"] = normal(seed, "cls.0.b", (D,), std=0.05)
    k = D ** -0.5
    sd["classifier.1.weight"] = uniform(seed, "cls.1.w", (D // 2, D), -k, k)
    sd["classifier.1.bias"] = uniform(seed, "cls.1.b", (D // 2,), -k, k)
    k = (D // 2) ** -0.5
    sd["classifier.4.weight"] = uniform(seed, "cls.4.w", (C, D // 2), -k, k)
    sd["classifier.4.bias"] = uniform(seed, "cls.4.b", (C,), -k, k)
    k = (2 * D) ** -0.5
    sd["projection_layer.weight"] = uniform(seed, "proj.w", (D, 2 * D), -k, k)
    sd["projection_layer.bias"] = uniform(seed, "proj.b", (D,), -k, k)
    return sd


def multi_hot_labels(seed: int, name: str, batch: int, num_classes: int = 140, max_pos: int = 3) -> torch.Tensor:
    """[batch, num_classes] f32 multi-hot with 1..max_pos positives per row (SURVEY.md §8d cfg 3)."""
    y = torch.zeros(batch, num_classes)
    npos = randint(seed, name + "/n", (batch,), 1, max_pos + 1)
    idx = randint(seed, name + "/i", (batch, max_pos), 0, num_classes)
    for b in range(batch):
        y[b, idx[b, : int(npos[b])]] = 1.0
    return y
