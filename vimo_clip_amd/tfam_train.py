"""Fused TFAM training chains (vmc_tfam_train_* / vmc_tfam_layer_bwd / vmc_tfam_head_bwd of include/vmc.h) behind
``AMO_CLIP.forward`` in train mode: the whole model is ONE autograd node.

The reference's training step (TFAM/train_and_eval.py:66-101: ``output = model(...)``, ``loss.backward()``) over
AttentionLayer.forward / AMO_CLIP.forward (TFAM/models/AMO_CLIP.py:37-51,99-171) runs, for short clips, as 27 forward and
38 backward launches instead of ~240 per-op launches.  Nothing is packed or copied for it: the kernels read the 16-bit
compute copies (and their transposes) that ``autograd_ops.weights`` keeps current after every optimiser step, the fp32
masters of biases / LayerNorm parameters, and write gradients straight into the parameters' gradient slots (the flat arena
of optim.GradArena when there is one).  Dropout draws the same seeds, in the same order, with the same element indices as
the per-op path (AttentionLayer.run), so both paths apply identical masks.
"""
from __future__ import annotations

import ctypes

import torch

from . import autograd_ops as ag
from . import tfam_fused
from ._lib import check, dt, lib, ptr, stream

_LAYER_FIELDS = (
    "w_self_in w_self_out w_cross_in w_cross_out w_ffn0 w_ffn3 "
    "wt_self_in wt_self_out wt_cross_in wt_cross_out wt_ffn0 wt_ffn3 "
    "b_self_in b_self_out b_cross_in b_cross_out b_ffn0 b_ffn3 "
    "ln_self_g ln_self_b ln_cross_g ln_cross_b ln_ffn_g ln_ffn_b "
    "gw_self_in gw_self_out gw_cross_in gw_cross_out gw_ffn0 gw_ffn3 "
    "gb_self_in gb_self_out gb_cross_in gb_cross_out gb_ffn0 gb_ffn3 "
    "g_ln_self_g g_ln_self_b g_ln_cross_g g_ln_cross_b g_ln_ffn_g g_ln_ffn_b").split()
_HEAD_FIELDS = ("w_cls1 w_cls4 w32_cls1 w32_cls4 cls_ln_g cls_ln_b b_cls1 b_cls4 "
                "g_cls_ln_g g_cls_ln_b gw_cls1 gb_cls1 gw_cls4 gb_cls4").split()


class LayerParams(ctypes.Structure):          # vmc_tfam_layer_params
    _fields_ = [(n, ctypes.c_void_p) for n in _LAYER_FIELDS]


class HeadParams(ctypes.Structure):           # vmc_tfam_head_params
    _fields_ = [(n, ctypes.c_void_p) for n in _HEAD_FIELDS]


SEEDS_PER_LAYER = 7


def supported(model, B, T, Tk, has_cross) -> bool:
    """Shapes the training chain covers: the eval chain's set, at most 256 token rows (the grouped weight gradient keeps the
    whole contraction in four LDS stages), at most 32 clips, a class count that is a multiple of 4."""
    if not tfam_fused.supported(model, B, T, Tk, has_cross):
        return False
    C = model.classifier[4].weight.shape[0]
    return B * T <= 256 and (not has_cross or B * Tk <= 256) and B <= 32 and C % 4 == 0


def _grad_dst(p, fresh):
    """Gradient destination of a parameter: its arena slot, or a new fp32 tensor handed back to autograd."""
    if p is None or not p.requires_grad:
        return None
    slot = getattr(p, "_vmc_grad", None)
    if slot is not None and slot.is_contiguous():
        return slot
    g = fresh.get(id(p))
    if g is None:
        g = fresh[id(p)] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)      # zeros: unused row ranges stay defined
    return g


class _Tables:
    """ctypes arrays of vmc_tfam_layer_params / vmc_tfam_head_params for one (model, compute dtype)."""

    def __init__(self, model, dtype16, cross):
        self.model, self.dtype16, self.cross = model, dtype16, cross
        self.key = None
        self.keep = []

    def params(self):
        m = self.model
        out = []
        for layer in m.layers:
            sa, ca = layer.self_attn, layer.cross_attn
            out += [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias]
            if self.cross:
                out += [ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias, layer.norm_cross.weight, layer.norm_cross.bias]
            out += [layer.ffn[0].weight, layer.ffn[0].bias, layer.ffn[3].weight, layer.ffn[3].bias,
                    layer.norm_self.weight, layer.norm_self.bias, layer.norm_ffn.weight, layer.norm_ffn.bias]
        c = m.classifier
        return out + [c[0].weight, c[0].bias, c[1].weight, c[1].bias, c[4].weight, c[4].bias]

    def build(self, fresh=None):
        """(layers array, head struct).  Cached; rebuilt when a parameter, a compute copy or a gradient slot moved.  fresh: a dict
        -> parameters without a gradient slot get new fp32 gradient tensors (collected in it; nothing is cached then)."""
        ps = self.params()
        key = (ag.weights.generation,
               tuple((p._version, p.data_ptr(), 0 if getattr(p, "_vmc_grad", None) is None else p._vmc_grad.data_ptr()) for p in ps))
        if fresh is None:
            if key == self.key:
                return self.layers, self.head
            if torch.cuda.is_current_stream_capturing() and self.key is not None:
                raise RuntimeError("TFAM training chain: parameters or their compute copies moved during a graph capture")
            fresh_d = {}
        else:
            fresh_d = fresh
        m, d16 = self.model, self.dtype16
        keep = []

        def w(p):
            t = ag.weights.get(p, d16, both=True)
            keep.append(t)
            return t.data_ptr()

        def wt(p):
            t = ag.weights.get(p, d16, transposed=True, both=True)
            keep.append(t)
            return t.data_ptr()

        def f(p):
            t = p.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("TFAM training chain needs contiguous fp32 parameters")
            return t.data_ptr()

        def g(p):
            if fresh is None:                               # cached tables carry arena slots only
                t = getattr(p, "_vmc_grad", None) if p.requires_grad else None
                t = t if (t is not None and t.is_contiguous()) else None
            else:
                t = _grad_dst(p, fresh_d)
            return None if t is None else t.data_ptr()

        L = len(m.layers)
        layers = (LayerParams * L)()
        for i, layer in enumerate(m.layers):
            sa, ca, s = layer.self_attn, layer.cross_attn, layers[i]
            s.w_self_in, s.wt_self_in, s.b_self_in = w(sa.in_proj_weight), wt(sa.in_proj_weight), f(sa.in_proj_bias)
            s.w_self_out, s.wt_self_out, s.b_self_out = w(sa.out_proj.weight), wt(sa.out_proj.weight), f(sa.out_proj.bias)
            s.gw_self_in, s.gb_self_in = g(sa.in_proj_weight), g(sa.in_proj_bias)
            s.gw_self_out, s.gb_self_out = g(sa.out_proj.weight), g(sa.out_proj.bias)
            if self.cross:
                s.w_cross_in, s.wt_cross_in, s.b_cross_in = w(ca.in_proj_weight), wt(ca.in_proj_weight), f(ca.in_proj_bias)
                s.w_cross_out, s.wt_cross_out, s.b_cross_out = w(ca.out_proj.weight), wt(ca.out_proj.weight), f(ca.out_proj.bias)
                s.gw_cross_in, s.gb_cross_in = g(ca.in_proj_weight), g(ca.in_proj_bias)
                s.gw_cross_out, s.gb_cross_out = g(ca.out_proj.weight), g(ca.out_proj.bias)
                s.ln_cross_g, s.ln_cross_b = f(layer.norm_cross.weight), f(layer.norm_cross.bias)
                s.g_ln_cross_g, s.g_ln_cross_b = g(layer.norm_cross.weight), g(layer.norm_cross.bias)
            s.w_ffn0, s.wt_ffn0, s.b_ffn0 = w(layer.ffn[0].weight), wt(layer.ffn[0].weight), f(layer.ffn[0].bias)
            s.w_ffn3, s.wt_ffn3, s.b_ffn3 = w(layer.ffn[3].weight), wt(layer.ffn[3].weight), f(layer.ffn[3].bias)
            s.gw_ffn0, s.gb_ffn0 = g(layer.ffn[0].weight), g(layer.ffn[0].bias)
            s.gw_ffn3, s.gb_ffn3 = g(layer.ffn[3].weight), g(layer.ffn[3].bias)
            s.ln_self_g, s.ln_self_b = f(layer.norm_self.weight), f(layer.norm_self.bias)
            s.ln_ffn_g, s.ln_ffn_b = f(layer.norm_ffn.weight), f(layer.norm_ffn.bias)
            s.g_ln_self_g, s.g_ln_self_b = g(layer.norm_self.weight), g(layer.norm_self.bias)
            s.g_ln_ffn_g, s.g_ln_ffn_b = g(layer.norm_ffn.weight), g(layer.norm_ffn.bias)
        c, h = m.classifier, HeadParams()
        for p in (c[1].weight, c[4].weight):                 # the head's backward uses the fp32 masters: no transposed copies
            keep.append(ag.weights.get(p, d16))
        h.w_cls1, h.w_cls4 = keep[-2].data_ptr(), keep[-1].data_ptr()
        h.w32_cls1, h.w32_cls4 = f(c[1].weight), f(c[4].weight)
        h.cls_ln_g, h.cls_ln_b, h.b_cls1, h.b_cls4 = f(c[0].weight), f(c[0].bias), f(c[1].bias), f(c[4].bias)
        h.g_cls_ln_g, h.g_cls_ln_b = g(c[0].weight), g(c[0].bias)
        h.gw_cls1, h.gb_cls1, h.gw_cls4, h.gb_cls4 = g(c[1].weight), g(c[1].bias), g(c[4].weight), g(c[4].bias)
        if fresh is None:
            self.key, self.layers, self.head, self.keep = key, layers, h, keep
        else:
            fresh_d["_keep"] = keep
        return layers, h


def _tables(model, dtype16, cross) -> _Tables:
    tabs = model.__dict__.setdefault("_tfam_train_tables", {})
    t = tabs.get((dtype16, cross))
    if t is None:
        t = tabs[(dtype16, cross)] = _Tables(model, dtype16, cross)
    return t


def _dims(model, B, T, Tk, cross):
    D, H, L = model.d_model, model.nhead, len(model.layers)
    return (B, T, Tk, D, H, model.layers[0].ffn[0].weight.shape[0], L, model.classifier[4].weight.shape[0], int(cross))


class TfamTrainFn(torch.autograd.Function):
    """logits = AMO_CLIP(x, motion) in train mode.  Inputs after ``seeds`` are the parameters (so that autograd asks for their
    gradients); their values are read through the pointer tables, not through these tensors."""

    @staticmethod
    def forward(ctx, model, x, motion, mask, mask_kv, cross, p_drop, p_mlp, seeds, *params):
        dt16 = model.compute_dtype
        B, T, D = x.shape
        Tk = motion.shape[1] if cross else 0
        dims = _dims(model, B, T, Tk, cross)
        tab = _tables(model, dt16, cross)
        layers, head = tab.build()
        nbytes = lib.vmc_tfam_train_workspace_bytes(*dims)
        if nbytes == 0:
            raise RuntimeError("vmc_tfam_train_workspace_bytes: unsupported shape")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        logits = torch.empty((B, dims[7]), dtype=torch.float32, device=x.device)
        sarr = (ctypes.c_uint64 * len(seeds))(*seeds) if seeds else None
        check(lib.vmc_tfam_train_fwd(ptr(x), ptr(motion) if cross else None, ptr(mask), ptr(mask_kv) if cross else None, layers,
                                     ctypes.byref(head), ptr(logits), ptr(ws), nbytes, *dims, float(p_drop), float(p_mlp), sarr, dt(dt16),
                                     stream()), "tfam_train_fwd")
        ctx.model, ctx.tab, ctx.ws, ctx.dims, ctx.cross = model, tab, ws, dims, cross
        ctx.masks, ctx.drop, ctx.sarr, ctx.params = (mask, mask_kv), (float(p_drop), float(p_mlp)), sarr, params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model, tab, ws, dims, cross = ctx.model, ctx.tab, ctx.ws, ctx.dims, ctx.cross
        mask, mask_kv = ctx.masks
        p_drop, p_mlp = ctx.drop
        dt16 = model.compute_dtype
        fresh = {}
        need_fresh = any(p.requires_grad and getattr(p, "_vmc_grad", None) is None for p in ctx.params)
        layers, head = tab.build(fresh) if need_fresh else tab.build()      # slot-less parameters: new gradient tensors, returned below
        dlogits = dlogits.contiguous().float()
        L = dims[6]
        sarr, nbytes = ctx.sarr, ws.numel()
        off = ctypes.sizeof(ctypes.c_uint64)
        seed_at = (lambda i: ctypes.cast(ctypes.addressof(sarr) + i * off, ctypes.c_void_p)) if sarr is not None else (lambda i: None)
        done = getattr(model, "grad_group_callback", None)      # e.g. FusedAdam.group_ready: the optimiser step of a finished group
        c = model.classifier
        head_params = (c[4].weight, c[4].bias, c[1].weight, c[1].bias, c[0].weight, c[0].bias)
        if not ag.grad_ready_hooks and done is None:
            # nobody waits for a layer's gradients (single process): the whole backward in one call -- the dgrad chains of all layers,
            # then every weight gradient in ONE grouped launch (vmc_tfam_train_bwd)
            check(lib.vmc_tfam_train_bwd(ptr(dlogits), ptr(mask), ptr(mask_kv) if cross else None, layers, ctypes.byref(head), ptr(ws), nbytes,
                                         *dims, p_drop, p_mlp, sarr, dt(dt16), stream()), "tfam_train_bwd")
        else:
            check(lib.vmc_tfam_head_bwd(ptr(dlogits), layers, ctypes.byref(head), ptr(ws), nbytes, *dims, p_mlp,
                                        int(sarr[SEEDS_PER_LAYER * L]) if sarr is not None else 0, dt(dt16), stream()), "tfam_head_bwd")
            _report(head_params)
            if done is not None:                                 # group L = the classifier
                done(L)
            for l in range(L - 1, -1, -1):
                check(lib.vmc_tfam_layer_bwd(ptr(mask), ptr(mask_kv) if cross else None, layers, l, ptr(ws), nbytes, *dims, p_drop,
                                             seed_at(SEEDS_PER_LAYER * l), dt(dt16), stream()), "tfam_layer_bwd")
                layer = model.layers[l]
                sa, ca = layer.self_attn, layer.cross_attn
                rep = [layer.norm_ffn.weight, layer.norm_ffn.bias, layer.ffn[3].weight, layer.ffn[3].bias, layer.ffn[0].weight, layer.ffn[0].bias]
                if cross:      # the per-op path reports the packed cross in_proj twice (q rows, k|v rows): same counts here (parallel.GradientAllReducer)
                    rep += [layer.norm_cross.weight, layer.norm_cross.bias, ca.out_proj.weight, ca.out_proj.bias, ca.in_proj_weight, ca.in_proj_bias,
                            ca.in_proj_weight, ca.in_proj_bias]
                rep += [layer.norm_self.weight, layer.norm_self.bias, sa.out_proj.weight, sa.out_proj.bias, sa.in_proj_weight, sa.in_proj_bias]
                _report(rep)
                if done is not None:
                    done(l)
        grads = tuple((fresh.get(id(p)) if p.requires_grad and getattr(p, "_vmc_grad", None) is None else None) for p in ctx.params)
        ctx.ws = None
        return (None,) * 9 + grads


def _report(params):
    if ag.grad_ready_hooks:
        for p in params:
            if p.requires_grad and getattr(p, "_vmc_grad", None) is not None:
                for hook in ag.grad_ready_hooks:
                    hook(p)


def forward_train(model, x, motion, mask, mask_kv, cross, seed_fn):
    """Train-mode forward of ``model`` through the fused chain (autograd-tracked); None when the shapes are outside its set."""
    B, T, _ = x.shape
    Tk = motion.shape[1] if cross else 0
    if x.requires_grad or (cross and motion.requires_grad) or not supported(model, B, T, Tk, cross):
        return None
    p = float(model.layers[0].p)
    if any(float(layer.p) != p for layer in model.layers):
        return None
    p_mlp = float(model.mlp_dropout)
    L = len(model.layers)
    seeds = []
    if p > 0.0 or p_mlp > 0.0:
        # the draw order of AttentionLayer.run: self-attention P, self branch, [cross-attention P, cross branch,] FFN inner, FFN trailing,
        # FFN branch; then the classifier's dropout
        for _ in range(L):
            row = [0] * SEEDS_PER_LAYER
            if p > 0.0:
                row[0], row[1] = seed_fn(), seed_fn()
                if cross:
                    row[2], row[3] = seed_fn(), seed_fn()
                row[4], row[5], row[6] = seed_fn(), seed_fn(), seed_fn()
            seeds += row
        seeds.append(seed_fn() if p_mlp > 0.0 else 0)
    tab = _tables(model, model.compute_dtype, cross)
    return TfamTrainFn.apply(model, x, motion, mask, mask_kv, cross, p, p_mlp, tuple(seeds), *tab.params())
