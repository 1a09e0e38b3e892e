"""MoCLIP student on libvmc — drop-in for the reference's models/student_model.py.

Same constructor / forward signatures, attribute names (``device, preprocess, visual_encoder,
residual_mlp.{fc1,fc2,alpha}, classification_head.{0,2}``) and ``state_dict`` keys as
``FlowStudentModel`` (models/student_model.py:38-98), so reference checkpoints (keys optionally
prefixed ``module.`` by DataParallel, train.py:167) load with ``strict=True``.

Differences that are deliberate and documented:
  * there are no pretrained weights offline: ``clip.load(name)`` (:44) is replaced by building the
    named geometry with OpenAI-clip initialisation; load a checkpoint with ``load_state_dict``;
  * the per-frame CPU loop ``to_pil_image -> CLIP preprocess`` (:77-78) runs as HIP kernels: Pillow-exact
    bicubic resize + centre crop (preprocess.py) when the frames are not already 224x224, then normalisation
    fused with patch extraction; the float->PIL wrap-around (v -> (256 - v) mod 256) is
    reproduced bit-exactly for u8 / integer-valued inputs.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import autograd_ops as ag
from .. import ops
from ..autograd_vit import vit_forward_train
from ..clip_vit import VisionTransformer, _Lin
from ..oracle_free_constants import CLIP_MEAN, CLIP_STD


class _Preprocess:
    """Stand-in for the torchvision ``Compose`` that ``clip.load`` returns (used as ``model.preprocess`` and
    ``model.preprocess.transforms`` by inference.py:90 / student_model.py:77).  The transform list is
    descriptive; the arithmetic runs inside vmc_preprocess_patches_u8."""

    def __init__(self, n_px: int):
        self.n_px = n_px
        self.transforms = [("Resize", n_px, "bicubic"), ("CenterCrop", n_px), ("ToTensor",), ("Normalize", CLIP_MEAN, CLIP_STD)]

    def __repr__(self):
        return f"CLIPPreprocess(n_px={self.n_px}, mean={CLIP_MEAN}, std={CLIP_STD})"


def _init_linear(lin: _Lin):
    nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
    bound = 1 / math.sqrt(lin.weight.shape[1])
    nn.init.uniform_(lin.bias, -bound, bound)


class ResidualMLP(nn.Module):
    """models/student_model.py:8-35: x + alpha * fc2(GELU(fc1(x))), fc2 zero-initialised."""

    def __init__(self, embed_dim, alpha=0.1, compute_dtype=torch.bfloat16):
        super().__init__()
        self.fc1 = _Lin(embed_dim, embed_dim)
        self.fc2 = _Lin(embed_dim, embed_dim)
        self.alpha = alpha
        self.compute_dtype = compute_dtype
        _init_linear(self.fc1)
        nn.init.zeros_(self.fc2.weight)
        nn.init.zeros_(self.fc2.bias)

    def forward(self, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        xa, xb = ag.fork(x2, self.compute_dtype)
        h = ag.linear(ag.cast(xa, self.compute_dtype), self.fc1.weight, self.fc1.bias, act=ops.ACT_GELU_ERF)
        m = ag.linear(h, self.fc2.weight, self.fc2.bias, out_f32=True)
        return ag.AddFn.apply(xb, m, self.alpha).view(shape)


class FlowStudentModel(nn.Module):
    def __init__(self, clip_model_name="ViT-B/32", device="cuda", num_classes=140, alpha=0.1,
                 compute_dtype=torch.bfloat16, residual_dtype=torch.float32):
        super().__init__()
        self.device = device
        self.compute_dtype = compute_dtype
        self.visual_encoder = VisionTransformer.from_name(clip_model_name, compute_dtype=compute_dtype,
                                                          residual_dtype=residual_dtype)
        self.preprocess = _Preprocess(self.visual_encoder.input_resolution)
        embed_dim = self.visual_encoder.output_dim
        self.residual_mlp = ResidualMLP(embed_dim, alpha=alpha, compute_dtype=compute_dtype)
        # nn.Sequential(Linear, ReLU, Linear): indices 0 and 2 carry parameters, as in the reference (:55-59)
        self.classification_head = nn.Sequential(_Lin(embed_dim, embed_dim // 2), nn.Identity(), _Lin(embed_dim // 2, num_classes))
        _init_linear(self.classification_head[0])
        _init_linear(self.classification_head[2])
        self.to(device)

    def _frames_u8(self, flow_videos):
        B, T, C, H, W = flow_videos.shape
        fr = flow_videos.reshape(B * T, C, H, W)      # any H x W: Resize(R, BICUBIC) + CenterCrop(R) run PIL-exact on the GPU
        if fr.dtype != torch.uint8:
            # the reference casts to float and to_pil_image multiplies by 255 and wraps to u8 (:74,:78);
            # integer-valued floats in 0..255 are the same pixels as their u8 cast
            fr = fr.to(torch.uint8)
        return fr.to(self.device)

    def forward(self, flow_videos):
        """flow_videos [B,T,3,H,W] -> (embeddings [B,T,E], embeddings_for_distillation [B,T,E], logits [B,C])."""
        B, T = flow_videos.shape[:2]
        frames = self._frames_u8(flow_videos)
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if train:
            emb = vit_forward_train(self.visual_encoder, frames, wrap_quirk=True)          # [B*T, E] f32
            e_out, e_mlp, e_pool = _fork3(emb, self.compute_dtype)
        else:
            emb = self.visual_encoder.encode_frames_u8(frames, wrap_quirk=True)
            e_out = e_mlp = e_pool = emb
        E = emb.shape[-1]
        emb_distill = self.residual_mlp(e_mlp.view(B, T, E))
        pooled = ag.MeanPoolFn.apply(e_pool, B, T, self.compute_dtype, False)             # [B,E] 16-bit
        h = ag.linear(pooled, self.classification_head[0].weight, self.classification_head[0].bias, act=ops.ACT_RELU)
        logits = ag.linear(h, self.classification_head[2].weight, self.classification_head[2].bias, out_f32=True)
        return e_out.view(B, T, E), emb_distill, logits


def _fork3(x, dt16):
    a, rest = ag.fork(x, dt16)
    b, c = ag.fork(rest, dt16)
    return a, b, c


FrameDiffStudentModel = FlowStudentModel   # models/student_model_frame_diff.py: identical arithmetic, renamed inputs
