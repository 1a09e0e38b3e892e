"""Oracle: CLIP ViT image encoder, fp32 (test infrastructure only — see oracle/__init__.py).

Follows the arithmetic that the reference reaches through third-party code that is absent from
/root/reference:
  * OpenAI ``clip`` @ dcba3cb (requirements.txt:5), ``clip.model.VisionTransformer`` — called at
    models/student_model.py:44,48,84;
  * HF ``transformers==4.53.2`` (requirements.txt:63), ``CLIPModel.get_image_features`` — called at
    extract_embeddings.py:17-18,94.
Both are the same published algorithm (pre-LN residual blocks, QuickGELU, LayerNorm eps 1e-5, CLS
pooling, bias-free patch conv and projection); parameter names are OpenAI clip's, which are the
``visual_encoder.*`` state_dict keys of the reference checkpoints (train.py:167).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def normalize_u8(frames_u8: torch.Tensor) -> torch.Tensor:
    """[F,3,H,W] u8 -> f32 (x/255 - mean)/std.  ToTensor + Normalize of clip._transform and the
    rescale+normalize of CLIPImageProcessor (extract_embeddings.py:91); for H=W=224 the bicubic
    resize / centre crop in front of it are identities."""
    x = frames_u8.to(torch.float32) / 255.0
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - mean) / std


def to_pil_wrap_u8(frames: torch.Tensor) -> torch.Tensor:
    """The student's float->PIL quirk (models/student_model.py:74,78):
    u8 frames are cast to float 0..255, ``to_pil_image`` multiplies floats by 255 and casts to u8,
    which wraps modulo 256:  v -> (v*255) mod 256 == (256 - v) mod 256."""
    v = frames.to(torch.int64)
    return ((v * 255) % 256).to(torch.uint8)


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def vit_forward(sd: dict, pixel_values: torch.Tensor, heads: int, prefix: str = "",
                return_tokens: bool = False, trace: list = None) -> torch.Tensor:
    """pixel_values [F,3,R,R] f32 (already normalised) -> [F,E] f32.
    trace (optional list): receives (stage name, residual stream [F,N,D]) after ln_pre and after every residual add."""
    g = lambda k: sd[prefix + k].to(torch.float32)
    conv_w = g("conv1.weight")
    D, _, p, _ = conv_w.shape
    x = F.conv2d(pixel_values.to(torch.float32), conv_w, stride=p)           # [F,D,g,g]
    Fn = x.shape[0]
    x = x.reshape(Fn, D, -1).permute(0, 2, 1)                                 # [F,g*g,D]
    cls = g("class_embedding").view(1, 1, D).expand(Fn, 1, D)
    x = torch.cat([cls, x], dim=1) + g("positional_embedding")
    x = layer_norm(x, g("ln_pre.weight"), g("ln_pre.bias"))
    if trace is not None:
        trace.append(("ln_pre", x.clone()))
    N = x.shape[1]
    dh = D // heads
    L = 0
    while (prefix + f"transformer.resblocks.{L}.ln_1.weight") in sd:
        L += 1
    for i in range(L):
        pre = f"transformer.resblocks.{i}."
        h = layer_norm(x, g(pre + "ln_1.weight"), g(pre + "ln_1.bias"))
        qkv = h @ g(pre + "attn.in_proj_weight").t() + g(pre + "attn.in_proj_bias")
        q, k, v = qkv.split(D, dim=-1)
        q = q.view(Fn, N, heads, dh).transpose(1, 2)
        k = k.view(Fn, N, heads, dh).transpose(1, 2)
        v = v.view(Fn, N, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
        o = torch.softmax(s, dim=-1) @ v
        o = o.transpose(1, 2).reshape(Fn, N, D)
        x = x + o @ g(pre + "attn.out_proj.weight").t() + g(pre + "attn.out_proj.bias")
        if trace is not None:
            trace.append((f"blk{i}.attn", x.clone()))
        h = layer_norm(x, g(pre + "ln_2.weight"), g(pre + "ln_2.bias"))
        h = quick_gelu(h @ g(pre + "mlp.c_fc.weight").t() + g(pre + "mlp.c_fc.bias"))
        x = x + h @ g(pre + "mlp.c_proj.weight").t() + g(pre + "mlp.c_proj.bias")
        if trace is not None:
            trace.append((f"blk{i}.mlp", x.clone()))
    if return_tokens:
        return x
    y = layer_norm(x[:, 0], g("ln_post.weight"), g("ln_post.bias"))
    return y @ g("proj")


def openai_to_hf_vision(sd: dict, heads: int) -> dict:
    """Weight map OpenAI-clip names -> HF CLIPModel names, used only by
    make_golden.py to pin vit_forward against transformers.CLIPModel built from a config."""
    out = {}
    D = sd["class_embedding"].shape[0]
    out["vision_model.embeddings.patch_embedding.weight"] = sd["conv1.weight"]
    out["vision_model.embeddings.class_embedding"] = sd["class_embedding"]
    out["vision_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out["vision_model.pre_layrnorm.weight"] = sd["ln_pre.weight"]
    out["vision_model.pre_layrnorm.bias"] = sd["ln_pre.bias"]
    out["vision_model.post_layernorm.weight"] = sd["ln_post.weight"]
    out["vision_model.post_layernorm.bias"] = sd["ln_post.bias"]
    out["visual_projection.weight"] = sd["proj"].t().contiguous()
    i = 0
    while f"transformer.resblocks.{i}.ln_1.weight" in sd:
        s = f"transformer.resblocks.{i}."
        d = f"vision_model.encoder.layers.{i}."
        out[d + "layer_norm1.weight"] = sd[s + "ln_1.weight"]
        out[d + "layer_norm1.bias"] = sd[s + "ln_1.bias"]
        out[d + "layer_norm2.weight"] = sd[s + "ln_2.weight"]
        out[d + "layer_norm2.bias"] = sd[s + "ln_2.bias"]
        w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            out[d + f"self_attn.{nm}.weight"] = w[j * D:(j + 1) * D]
            out[d + f"self_attn.{nm}.bias"] = b[j * D:(j + 1) * D]
        out[d + "self_attn.out_proj.weight"] = sd[s + "attn.out_proj.weight"]
        out[d + "self_attn.out_proj.bias"] = sd[s + "attn.out_proj.bias"]
        out[d + "mlp.fc1.weight"] = sd[s + "mlp.c_fc.weight"]
        out[d + "mlp.fc1.bias"] = sd[s + "mlp.c_fc.bias"]
        out[d + "mlp.fc2.weight"] = sd[s + "mlp.c_proj.weight"]
        out[d + "mlp.fc2.bias"] = sd[s + "mlp.c_proj.bias"]
        i += 1
    return out
