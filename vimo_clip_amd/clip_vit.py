"""CLIP ViT image encoder on the libvmc HIP kernels.

Drop-in for the two third-party encoders the reference calls:
  * OpenAI ``clip`` ``model.visual`` (models/student_model.py:44-48,84): ``VisionTransformer`` keeps the
    parameter names (``conv1.weight, class_embedding, positional_embedding, ln_pre.*,
    transformer.resblocks.{i}.{ln_1,attn.in_proj_weight,attn.in_proj_bias,attn.out_proj,ln_2,mlp.c_fc,
    mlp.c_proj}, ln_post.*, proj``) so reference checkpoints load with ``strict=True`` (inference.py:86),
    and the attributes ``output_dim`` / ``input_resolution`` (models/student_model.py:49).
  * HF ``CLIPModel.get_image_features`` (extract_embeddings.py:94): ``CLIPImageEncoder``.

Data layout in HBM (F frames, N = g*g+1 tokens, width D):
  patches [F*g*g, kpad] 16-bit  ->  x [F*N, D] residual stream (fp32 by default)  ->  per block:
  h [F*N, D] 16-bit (LayerNorm out), qkv [F*N, 3D] 16-bit, o [F*N, D] 16-bit, u [F*N, 4D] 16-bit.
All GEMMs are vmc_linear (bf16/f16 MFMA, fp32 accumulate) with bias / QuickGELU / residual /
positional-embedding epilogues fused; there is no PyTorch compute on this path.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from .synth import VIT_GEOMETRY


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Lin(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_out, d_in))
        self.bias = nn.Parameter(torch.zeros(d_out))


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _Lin(d, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = _Attn(d)
        self.ln_1 = _LN(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", _Lin(d, 4 * d)), ("c_proj", _Lin(4 * d, d))]))
        self.ln_2 = _LN(d)
        self.heads = heads


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class _Conv(nn.Module):
    def __init__(self, d, p):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, 3, p, p))


class VisionTransformer(nn.Module):
    """OpenAI-clip ``VisionTransformer`` parameter surface; forward runs on libvmc."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int,
                 compute_dtype: torch.dtype = torch.bfloat16, residual_dtype: torch.dtype = torch.float32):
        super().__init__()
        if width != heads * 64:
            raise ValueError("CLIP ViT geometries have head_dim 64")
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.width, self.layers, self.heads = width, layers, heads
        self.conv1 = _Conv(width, patch_size)
        scale = width ** -0.5
        g = input_resolution // patch_size
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(g * g + 1, width))
        self.ln_pre = _LN(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = _LN(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.compute_dtype = compute_dtype
        self.residual_dtype = residual_dtype
        self._w16 = {}          # name -> (param version, 16-bit copy)
        self.fuse_add_ln = True  # False: x += GEMM in the epilogue (fp32 read-modify-write), plain LayerNorm after it
        self.cls_only_last_block = True  # last block's out_proj / MLP on the F class rows only (what ln_post reads)
        self.cls_query_last_block = True  # ... and its attention for the class token's query only (vmc_attention_vit_cls_fwd)
        # x + attention branch not written back, the add+LN after c_proj redoes it (vmc_add2_layernorm_fwd: 22 instead of 24 B per
        # element and layer, bit-identical).  Measured 41.04 vs 40.98 ms per step: the saved fp32 write is paid back by re-reading a
        # branch that has left the caches -- off by default (profiles/README.md, round 2)
        self.defer_attn_add = False
        self.exact_patch_embed = True   # split-precision patch GEMM (fp32-accurate; +0.2 % FLOPs): see patch_operands()
        self.frame_chunk = 256  # frames per pass (bounds activation memory; F*N*4D*2 B for the MLP buffer)
        self._init_weights()

    @classmethod
    def from_name(cls, name: str, **kw):
        R, p, D, L, H, E = VIT_GEOMETRY[name]
        return cls(R, p, D, L, H, E, **kw)

    def _init_weights(self):
        D, L = self.width, self.layers
        proj_std, attn_std, fc_std = (D ** -0.5) * ((2 * L) ** -0.5), D ** -0.5, (2 * D) ** -0.5
        nn.init.normal_(self.conv1.weight, std=(3 * self.patch_size ** 2) ** -0.5)
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)

    # ---- 16-bit compute copies of the fp32 master weights -------------------------------------------
    def w16(self, name: str, param: torch.Tensor, transposed=False, pad_k=False) -> torch.Tensor:
        from .autograd_ops import weights          # one cache for the inference and training paths
        return weights.get(param, self.compute_dtype, transposed=transposed, pad_k=pad_k)

    def patch_operands(self, mode: str):
        """Weight / bias / alpha of the inference patch-embedding GEMM for a patch layout (ops.patches_u8_exact,
        ops.patches_f32_split or the plain single-slice layout).  The split layouts carry the weight as hi + lo 16-bit halves,
        so the GEMM is fp32-accurate (include/vmc.h, vmc_patches_u8_exact).  One-off weight preparation, cached per weight
        version like every other 16-bit compute copy."""
        from .autograd_ops import weights
        from .oracle_free_constants import CLIP_MEAN, CLIP_STD
        w = self.conv1.weight
        key = (mode, self.compute_dtype, weights.epoch, w._version, w.data_ptr())
        hit = getattr(self, "_patch_ops", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        dt16, p = self.compute_dtype, self.patch_size
        D, k = w.shape[0], 3 * p * p
        kpad = (k + 63) // 64 * 64
        w2 = w.detach().float().reshape(D, 3, p * p)

        def split(wm):          # [D, k] fp32 -> 16-bit hi, lo (zero padded to kpad)
            hi = wm.to(dt16)
            lo = (wm - hi.float()).to(dt16)
            pad = lambda t: torch.nn.functional.pad(t, (0, kpad - k))
            return pad(hi), pad(lo)

        if mode == "u8_exact":
            istd = torch.tensor([1.0 / s_ for s_ in CLIP_STD], device=w.device).view(1, 3, 1)
            mean = torch.tensor(CLIP_MEAN, device=w.device).view(1, 3, 1)
            hi, lo = split((w2 * istd).reshape(D, k))
            bias = (-255.0 * (w2.double() * (mean.double() * istd.double())).sum(dim=(1, 2))).float().contiguous()
            ops_ = (torch.cat([hi, lo], dim=1).contiguous(), bias, 1.0 / 255.0)
        elif mode == "f32_split":
            hi, lo = split(w2.reshape(D, k))
            ops_ = (torch.cat([hi, hi, lo], dim=1).contiguous(), None, 1.0)
        else:
            ops_ = (self.w16("conv1", w, pad_k=True), None, 1.0)
        self._patch_ops = (key, ops_)
        return ops_

    def invalidate_weight_cache(self):
        from .autograd_ops import weights
        weights.clear()

    # ---- inference forward ---------------------------------------------------------------------------
    @torch.no_grad()
    def _encode_patches(self, patches: torch.Tensor, F: int, trace: list = None, patch_mode: str = "plain") -> torch.Tensor:
        """trace (tests only): receives (stage, copy of the residual stream [F*N, D]) after ln_pre and after each residual add
        of every block but the last MLP add (there only the class rows are formed)."""
        dt16, D, H = self.compute_dtype, self.width, self.heads
        g = self.input_resolution // self.patch_size
        g2, N = g * g, g * g + 1
        dev = patches.device
        x = torch.empty((F * N, D), dtype=self.residual_dtype, device=dev)
        pos = self.positional_embedding.detach()
        # K1: patch GEMM, epilogue adds positional_embedding[1 + patch] and scatters into token rows
        w_patch, b_patch, alpha = self.patch_operands(patch_mode)
        ops.linear(patches, w_patch, bias=b_patch, alpha=alpha, res=pos[1:], out=x, out_row_group=g2, res_row_mod=g2)
        ops.set_class_rows(x, self.class_embedding.detach(), pos[0], F, D, N * D, dt16)
        xf32 = self.residual_dtype == torch.float32
        # ln_pre in place on the residual stream
        if xf32:
            ops.layernorm(x, self.ln_pre.weight, self.ln_pre.bias, dt16, out16=False, out32=True, y32=x)
        else:
            self._ln_inplace16(x, self.ln_pre)
        if trace is not None:
            trace.append(("ln_pre", x.float().clone()))
        fused = xf32 and D % 256 == 0 and self.fuse_add_ln   # residual add fused into the next LayerNorm (vmc_add_layernorm_fwd)
        blocks = list(self.transformer.resblocks)
        h = None
        for i, blk in enumerate(blocks):
            pre = f"blk{i}."
            last = i + 1 == len(blocks)
            if h is None:
                h, *_ = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias, dt16)
            cls_query = last and self.cls_only_last_block and self.cls_query_last_block and trace is None
            if cls_query:
                # ... and of its attention only the class token's QUERY is needed (K and V of every token still are): the q third of
                # in_proj runs on the F class rows, the attention on one query tile per (frame, head) instead of ceil(N / 16)
                w_in, b_in = self.w16(pre + "in_proj", blk.attn.in_proj_weight), blk.attn.in_proj_bias
                kv = ops.linear(h, w_in[D:], bias=b_in[D:])
                q_cls = ops.linear(h.view(F, N, D)[:, 0, :], w_in[:D], bias=b_in[:D])
                o = None
                o_cls = ops.attention_vit_cls(q_cls, kv, F, N, H)
                del kv, q_cls
            else:
                qkv = ops.linear(h, self.w16(pre + "in_proj", blk.attn.in_proj_weight), bias=blk.attn.in_proj_bias)
                o, _ = ops.attention_vit(qkv, F, N, H)
                del qkv
            if last and self.cls_only_last_block:
                # Only x[:, 0] of the last block reaches ln_post (`x[:, 0, :]`, modeling_clip.py:650; OpenAI clip model.py
                # `x = self.ln_post(x[:, 0, :])`): its out_proj, ln_2, MLP and residual adds are per-token maps, so they run on
                # the F class rows instead of the F*N token rows -- the same arithmetic on 1/N of the rows (the attention itself
                # still needs every token's K and V).  Saves two of the 96 large GEMMs and two add+LayerNorm passes per step.
                if o is not None:
                    o_cls = o.view(F, N, D)[:, 0, :]                  # strided rows (lda = N*D)
                x_cls = x.view(F, N, D)[:, 0, :]
                if fused:
                    a = ops.linear(o_cls, self.w16(pre + "out_proj", blk.attn.out_proj.weight), bias=blk.attn.out_proj.bias)
                    hc = ops.add_layernorm_(x, a, blk.ln_2.weight, blk.ln_2.bias, rows=F, ldx=N * D, ldb=D)
                    u = ops.linear(hc, self.w16(pre + "c_fc", blk.mlp.c_fc.weight), bias=blk.mlp.c_fc.bias, act=ops.ACT_QUICKGELU)
                    m = ops.linear(u, self.w16(pre + "c_proj", blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias)
                    cls = ops.add_layernorm_(x, m, self.ln_post.weight, self.ln_post.bias, rows=F, ldx=N * D, ldb=D, write_x=False)
                else:
                    ops.linear(o_cls, self.w16(pre + "out_proj", blk.attn.out_proj.weight), bias=blk.attn.out_proj.bias, res=x_cls, out=x_cls)
                    hc, *_ = ops.layernorm(x, blk.ln_2.weight, blk.ln_2.bias, dt16, rows=F, ldx=N * D)
                    u = ops.linear(hc, self.w16(pre + "c_fc", blk.mlp.c_fc.weight), bias=blk.mlp.c_fc.bias, act=ops.ACT_QUICKGELU)
                    ops.linear(u, self.w16(pre + "c_proj", blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias, res=x_cls, out=x_cls)
                    cls, *_ = ops.layernorm(x, self.ln_post.weight, self.ln_post.bias, dt16, rows=F, ldx=N * D)
                del u, o
                break
            defer = fused and trace is None and self.defer_attn_add
            if fused:
                a = ops.linear(o, self.w16(pre + "out_proj", blk.attn.out_proj.weight), bias=blk.attn.out_proj.bias)
                # defer: h = LN(x + a) without writing x + a back; the add+LayerNorm after c_proj redoes (x + a) + m from the
                # kept attention branch -- 8 + 14 instead of 12 + 12 bytes per element, the same fp32 stream bit for bit
                h = ops.add_layernorm_(x, a, blk.ln_2.weight, blk.ln_2.bias, write_x=not defer)
            else:
                ops.linear(o, self.w16(pre + "out_proj", blk.attn.out_proj.weight), bias=blk.attn.out_proj.bias, res=x, out=x)
                h, *_ = ops.layernorm(x, blk.ln_2.weight, blk.ln_2.bias, dt16)
            if trace is not None:
                trace.append((f"blk{i}.attn", x.float().clone()))
            u = ops.linear(h, self.w16(pre + "c_fc", blk.mlp.c_fc.weight), bias=blk.mlp.c_fc.bias, act=ops.ACT_QUICKGELU)
            if fused:
                m = ops.linear(u, self.w16(pre + "c_proj", blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias)
                if last:       # only the class rows are needed after the last block: x[cls] + m[cls] -> ln_post
                    if defer:
                        cls = ops.add_layernorm_(x, m, self.ln_post.weight, self.ln_post.bias, rows=F, ldx=N * D, ldb=N * D, write_x=False,
                                                 branch0=a.view(F, N, D)[:, 0, :].contiguous())
                    else:
                        cls = ops.add_layernorm_(x, m, self.ln_post.weight, self.ln_post.bias, rows=F, ldx=N * D, ldb=N * D, write_x=False)
                else:
                    nxt = blocks[i + 1]
                    h = ops.add_layernorm_(x, m, nxt.ln_1.weight, nxt.ln_1.bias, branch0=a if defer else None)
                del m, a
            else:
                ops.linear(u, self.w16(pre + "c_proj", blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias, res=x, out=x)
                h = None
                if last:
                    cls, *_ = ops.layernorm(x, self.ln_post.weight, self.ln_post.bias, dt16, rows=F, ldx=N * D)
            if trace is not None and not (fused and last):
                trace.append((f"blk{i}.mlp", x.float().clone()))
            del u, o
        return ops.linear(cls, self.w16("proj", self.proj, transposed=True), out_dtype=torch.float32)

    def _ln_inplace16(self, x, ln):
        y, *_ = ops.layernorm(x, ln.weight, ln.bias, self.compute_dtype)
        x.copy_(y)   # device-to-device copy (memory plumbing); only taken for a 16-bit residual stream

    def fit_frames_u8(self, frames_u8: torch.Tensor, wrap_quirk: bool = False, crop_mode: str = "torchvision"):
        """Frames of any size -> [F,3,R,R] u8 by Resize(R, BICUBIC) + CenterCrop(R) with Pillow's exact arithmetic
        (preprocess.resize_center_crop_u8).  Returns (frames, wrap_still_pending)."""
        R = self.input_resolution
        if tuple(frames_u8.shape[-2:]) == (R, R):
            return frames_u8, wrap_quirk
        from .preprocess import resize_center_crop_u8
        return resize_center_crop_u8(frames_u8, R, crop_mode, wrap_quirk)

    def _warm_weight_copies(self, patch_mode: str):
        """Every 16-bit compute copy the inference forward reads, cast on the CURRENT stream (before the forward forks into streams:
        the cache is filled at enqueue time, so a second stream could otherwise read a copy whose cast kernel has not run yet)."""
        self.patch_operands(patch_mode)
        for i, blk in enumerate(self.transformer.resblocks):
            pre = f"blk{i}."
            self.w16(pre + "in_proj", blk.attn.in_proj_weight)
            self.w16(pre + "out_proj", blk.attn.out_proj.weight)
            self.w16(pre + "c_fc", blk.mlp.c_fc.weight)
            self.w16(pre + "c_proj", blk.mlp.c_proj.weight)
        self.w16("proj", self.proj, transposed=True)

    @torch.no_grad()
    def _encode_on_streams(self, frames_u8: torch.Tensor, wrap: bool, n: int) -> torch.Tensor:
        """The frames of one pass as n slices on n streams: while one slice's kernel drains (the last round of a persistent GEMM walk,
        the tail launch, a memory-bound pass that leaves the MFMA pipes idle) another slice's next kernel already runs.  Frames are
        independent and every slice runs the same kernels: same bits as one stream (tests/test_gpu_encoder.py)."""
        F = frames_u8.shape[0]
        mode = "u8_exact" if self.exact_patch_embed else "plain"
        self._warm_weight_copies(mode)
        cur = torch.cuda.current_stream()
        pool = getattr(self, "_slice_streams", None)
        if pool is None or len(pool) < n:
            pool = self._slice_streams = [torch.cuda.Stream() for _ in range(n)]
        cuts = [F * k // n for k in range(n + 1)]
        outs = []
        for st, a, b in zip(pool, cuts[:-1], cuts[1:]):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                fr = frames_u8[a:b]
                if self.exact_patch_embed:
                    patches = ops.patches_u8_exact(fr, self.patch_size, self.compute_dtype, wrap)
                else:
                    patches = ops.preprocess_patches_u8(fr, self.patch_size, self.compute_dtype, wrap)
                outs.append(self._encode_patches(patches, b - a, patch_mode=mode))
        for st in pool[:n]:
            cur.wait_stream(st)
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def encode_frames_u8(self, frames_u8: torch.Tensor, wrap_quirk: bool = False, crop_mode: str = "torchvision") -> torch.Tensor:
        """[F,3,H,W] u8 -> [F,E] f32.  Resize/crop (when H,W != R), CLIP normalisation and patch extraction run on the GPU (K0)."""
        n = int(getattr(self, "slice_streams", 1))
        if n > 1 and frames_u8.is_cuda and 64 * n <= frames_u8.shape[0] <= self.frame_chunk:
            fr, wrap = self.fit_frames_u8(frames_u8, wrap_quirk, crop_mode)
            return self._encode_on_streams(fr, wrap, n)
        outs = []
        for s in range(0, frames_u8.shape[0], self.frame_chunk):
            fr, wrap = self.fit_frames_u8(frames_u8[s:s + self.frame_chunk], wrap_quirk, crop_mode)
            if self.exact_patch_embed:
                patches = ops.patches_u8_exact(fr, self.patch_size, self.compute_dtype, wrap)
                outs.append(self._encode_patches(patches, fr.shape[0], patch_mode="u8_exact"))
            else:
                patches = ops.preprocess_patches_u8(fr, self.patch_size, self.compute_dtype, wrap)
                outs.append(self._encode_patches(patches, fr.shape[0]))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    @torch.no_grad()
    def encode_pixel_values(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[F,3,R,R] normalised floats -> [F,E] f32 (the ``visual_encoder(x)`` call of student_model.py:84)."""
        outs = []
        for s in range(0, pixel_values.shape[0], self.frame_chunk):
            pv = pixel_values[s:s + self.frame_chunk]
            if self.exact_patch_embed:
                outs.append(self._encode_patches(ops.patches_f32_split(pv, self.patch_size, self.compute_dtype), pv.shape[0],
                                                 patch_mode="f32_split"))
            else:
                outs.append(self._encode_patches(ops.patches_f32(pv, self.patch_size, self.compute_dtype), pv.shape[0]))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd_vit import vit_forward_train
            return vit_forward_train(self, x)
        if x.dtype == torch.uint8:
            return self.encode_frames_u8(x)
        return self.encode_pixel_values(x)


class CLIPImageEncoder(nn.Module):
    """The slice of HF ``CLIPModel`` the extractor uses (extract_embeddings.py:17,94):
    ``get_image_features(pixel_values) -> [F, E]`` tensor (transformers 4.53.2 semantics)."""

    def __init__(self, clip_model_name: str = "ViT-B/16", compute_dtype=torch.bfloat16, residual_dtype=torch.float32):
        super().__init__()
        self.visual = VisionTransformer.from_name(clip_model_name, compute_dtype=compute_dtype,
                                                  residual_dtype=residual_dtype)

    @torch.no_grad()
    def get_image_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        return self.visual.encode_pixel_values(pixel_values)
