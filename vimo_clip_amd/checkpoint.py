"""Checkpoint plumbing with the reference's file layouts.

* student: ``torch.save(model.state_dict())`` per epoch + ``"<run> - best"/student_best.pth`` (train.py:164-172); the
  reference saves from ``nn.DataParallel`` so every key carries a ``module.`` prefix, and ``inference.py:85-86`` loads it
  back into a DataParallel-wrapped model ``strict=True``.
* TFAM: ``{"epoch", "state_dict", "optimizer", "scheduler", "best_val_loss", "best_val_mAP"}`` written as
  ``best_model.pth`` only when the validation mAP improves (TFAM/train_and_eval.py:133-148), reloaded before testing
  (:186-191).  The ``"optimizer"`` entry is ``FusedAdam.state_dict()`` = ``{"step", "m", "v", "lr"}`` over the flat parameter
  arena, NOT ``torch.optim.AdamW.state_dict()``'s per-parameter layout: the reference's loaders only read ``state_dict``
  (:188-190), so weights interchange and optimiser state does not.

Files written here keep the ``module.`` prefix (so the reference's own loaders accept them); loading accepts both forms.
Reference files are opened with ``weights_only=True`` (nothing in the file is executed).  ``state_dict()`` tensors of a
model whose parameters live in a ``GradArena`` are views into the arena: they are cloned before saving.
"""
from __future__ import annotations

import os

import torch

PREFIX = "module."


def strip_prefix(sd: dict) -> dict:
    return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in sd.items()}


def add_prefix(sd: dict) -> dict:
    return {(k if k.startswith(PREFIX) else PREFIX + k): v for k, v in sd.items()}


def snapshot(model) -> dict:
    """CPU clone of the state dict with the DataParallel prefix the reference's checkpoints carry."""
    return add_prefix({k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items()})


def save_state_dict(model, path: str) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(snapshot(model), path)


def load_file(path: str, map_location="cpu"):
    return torch.load(path, map_location=map_location, weights_only=True)


def load_state_dict(model, path_or_sd, strict: bool = True):
    sd = load_file(path_or_sd) if isinstance(path_or_sd, (str, os.PathLike)) else path_or_sd
    if "state_dict" in sd and all(not torch.is_tensor(v) or k == "state_dict" for k, v in sd.items()):
        sd = sd["state_dict"]                       # TFAM checkpoint dict
    return model.load_state_dict(strip_prefix(sd), strict=strict)


# ---- HF CLIPModel (extract_embeddings.py:17) -> OpenAI clip VisionTransformer key names -----------------------------------
def hf_clip_to_openai_visual(hf_sd: dict) -> dict:
    """State dict of ``transformers.CLIPModel`` / ``CLIPVisionModelWithProjection`` -> the ``visual.*`` names of OpenAI clip
    (SURVEY.md 8c weight map): q/k/v projections are packed into ``attn.in_proj_*``, ``visual_projection.weight`` is
    transposed into ``proj``.  Text-tower keys are ignored."""
    pre = "vision_model."
    sd = {k[len(pre):]: v for k, v in hf_sd.items() if k.startswith(pre)}
    out = {
        "conv1.weight": sd["embeddings.patch_embedding.weight"],
        "class_embedding": sd["embeddings.class_embedding"],
        "positional_embedding": sd["embeddings.position_embedding.weight"],
        "ln_pre.weight": sd["pre_layrnorm.weight"], "ln_pre.bias": sd["pre_layrnorm.bias"],
        "ln_post.weight": sd["post_layernorm.weight"], "ln_post.bias": sd["post_layernorm.bias"],
        "proj": hf_sd["visual_projection.weight"].t().contiguous(),
    }
    i = 0
    while f"encoder.layers.{i}.layer_norm1.weight" in sd:
        s, d = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[d + "ln_1.weight"], out[d + "ln_1.bias"] = sd[s + "layer_norm1.weight"], sd[s + "layer_norm1.bias"]
        out[d + "ln_2.weight"], out[d + "ln_2.bias"] = sd[s + "layer_norm2.weight"], sd[s + "layer_norm2.bias"]
        out[d + "attn.in_proj_weight"] = torch.cat([sd[s + f"self_attn.{n}_proj.weight"] for n in "qkv"], dim=0)
        out[d + "attn.in_proj_bias"] = torch.cat([sd[s + f"self_attn.{n}_proj.bias"] for n in "qkv"], dim=0)
        out[d + "attn.out_proj.weight"], out[d + "attn.out_proj.bias"] = sd[s + "self_attn.out_proj.weight"], sd[s + "self_attn.out_proj.bias"]
        out[d + "mlp.c_fc.weight"], out[d + "mlp.c_fc.bias"] = sd[s + "mlp.fc1.weight"], sd[s + "mlp.fc1.bias"]
        out[d + "mlp.c_proj.weight"], out[d + "mlp.c_proj.bias"] = sd[s + "mlp.fc2.weight"], sd[s + "mlp.fc2.bias"]
        i += 1
    return out
