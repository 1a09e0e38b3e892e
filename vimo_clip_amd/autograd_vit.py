"""Differentiable forward of the CLIP ViT on libvmc kernels (student fine-tuning path, train.py:95-104).

Same arithmetic as VisionTransformer._encode_patches, but every op is an autograd.Function from
autograd_ops so that ``loss.backward()`` runs the hand-written backward kernels (K8).  Activations are
kept in the compute dtype; the residual stream uses ``model.residual_dtype``.
"""
from __future__ import annotations

import torch

from . import autograd_ops as ag
from . import ops


def vit_tokens_train(model, x: torch.Tensor, wrap_quirk: bool = False, cls_only: bool = True):
    """x: u8 [F,3,R,R] frames or float pixel values.  Returns (the final residual stream, F, N, rows_are_cls).
    cls_only: only x[:, 0] of the last block reaches ln_post, so its out_proj / ln_2 / MLP run on the F class rows (forward and,
    through autograd, backward: the dgrad / wgrad GEMMs of those four linears shrink by N as well); the stream returned is then
    [F, D].  The attention of the last block still sees every token (its K / V gradients flow to all rows)."""
    dt16, D, H = model.compute_dtype, model.width, model.heads
    F = x.shape[0]
    g = model.input_resolution // model.patch_size
    N = g * g + 1
    with torch.no_grad():   # the input frames carry no gradient
        if x.dtype == torch.uint8:
            x, wrap_quirk = model.fit_frames_u8(x, wrap_quirk)
            patches = ops.preprocess_patches_u8(x, model.patch_size, dt16, wrap_quirk)
        else:
            patches = ops.patches_f32(x, model.patch_size, dt16)
    xp = ag.linear(patches, model.conv1.weight)                                    # [F*g*g, D]
    xs = ag.AssembleTokensFn.apply(xp, model.class_embedding, model.positional_embedding, F, N, model.residual_dtype)
    xs = ag.layernorm(xs, model.ln_pre.weight, model.ln_pre.bias, dt16, out_f32=(model.residual_dtype == torch.float32))
    if model.residual_dtype != torch.float32:
        xs = ag.cast(xs, model.residual_dtype)
    blocks = list(model.transformer.resblocks)
    for i, blk in enumerate(blocks):
        h, xs = ag.layernorm(xs, blk.ln_1.weight, blk.ln_1.bias, dt16, passthrough=True)
        qkv = ag.linear(h, blk.attn.in_proj_weight, blk.attn.in_proj_bias)
        o = ag.SelfAttnPackedFn.apply(qkv, None, F, N, H)
        if cls_only and i + 1 == len(blocks):
            o, xs = _ClsRowsFn.apply(o, F, N), _ClsRowsFn.apply(xs, F, N)
        xs = ag.linear(o, blk.attn.out_proj.weight, blk.attn.out_proj.bias, res=xs,
                       out_f32=(model.residual_dtype == torch.float32))
        h, xs = ag.layernorm(xs, blk.ln_2.weight, blk.ln_2.bias, dt16, passthrough=True)
        u = ag.linear(h, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, act=ops.ACT_QUICKGELU)
        xs = ag.linear(u, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, res=xs,
                       out_f32=(model.residual_dtype == torch.float32))
    return xs, F, N, cls_only


class _ClsRowsFn(torch.autograd.Function):
    """Select the class-token rows x[f*N] (a strided gather is memory plumbing); the backward scatters the
    gradient back into a zero matrix."""

    @staticmethod
    def forward(ctx, xs, F, N):
        ctx.meta = (F, N, xs.shape[1], xs.dtype)
        return xs.view(F, N, -1)[:, 0].contiguous()

    @staticmethod
    def backward(ctx, dcls):
        F, N, D, dtype = ctx.meta
        dx = torch.zeros((F, N, D), dtype=dtype, device=dcls.device)
        dx[:, 0].copy_(dcls if dcls.dtype == dtype else (ops.cast32(dcls.contiguous()) if dtype == torch.float32
                                                          else ops.cast16(dcls.contiguous(), dtype)))
        return dx.view(F * N, D), None, None


class _ProjFn(torch.autograd.Function):
    """y = x @ proj  with proj stored [D, E] (OpenAI clip ``x @ self.proj``): linear with W = proj^T."""

    @staticmethod
    def forward(ctx, x16, proj):
        wT = ag.weights.get(proj, x16.dtype, transposed=True)                     # [E, D]
        ctx.save_for_backward(x16)
        ctx.proj = proj
        return ops.linear(x16, wT, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, dy):
        (x16,) = ctx.saved_tensors
        proj = ctx.proj
        dt16 = x16.dtype
        M, D = x16.shape
        E = proj.shape[1]
        dy16 = ops.cast16(dy.contiguous(), dt16)
        w = ag.weights.get(proj, dt16, pad_k=(E % 64 != 0))                       # [D, Epad]
        dyp = dy16
        if w.shape[1] != E:
            dyp = torch.zeros((M, w.shape[1]), dtype=dt16, device=dy16.device)
            dyp[:, :E].copy_(dy16)
        dx = ops.linear(dyp, w)                                                    # [M,E] @ [D,E]^T -> [M,D]
        out = ag._grad_out(proj, (D, E))
        if D % 8 == 0 and E % 8 == 0:
            ops.wgrad_tn(x16, dy16, out)                                            # dproj[D,E] = x^T dy
        else:
            Mp = ag._pad64(M)
            xt = torch.zeros((D, Mp), dtype=dt16, device=x16.device)
            ag.check(ag.lib.vmc_transpose16(ag.ptr(x16), ag.ptr(xt), M, D, x16.stride(0), Mp, ag.stream()), "transpose16")
            dyt = torch.zeros((E, Mp), dtype=dt16, device=x16.device)
            ag.check(ag.lib.vmc_transpose16(ag.ptr(dy16), ag.ptr(dyt), M, E, dy16.stride(0), Mp, ag.stream()), "transpose16")
            ops.linear_wgrad(xt, dyt, out)
        return dx, ag._deliver(proj, out)


def vit_forward_train(model, x: torch.Tensor, wrap_quirk: bool = False) -> torch.Tensor:
    """[F,3,R,R] -> [F,E] f32 embeddings, differentiable w.r.t. every ViT parameter."""
    xs, F, N, rows_are_cls = vit_tokens_train(model, x, wrap_quirk, cls_only=getattr(model, "cls_only_last_block", True))
    cls = xs if rows_are_cls else _ClsRowsFn.apply(xs, F, N)
    h = ag.layernorm(cls, model.ln_post.weight, model.ln_post.bias, model.compute_dtype)
    return _ProjFn.apply(h, model.proj)
