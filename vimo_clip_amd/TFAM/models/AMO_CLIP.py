"""TFAM fusion block + multi-label head on libvmc — drop-in for TFAM/models/AMO_CLIP.py.

``AttentionLayer`` / ``AMO_CLIP`` keep the reference's constructor and forward signatures and its
``state_dict`` keys (``layers.{i}.self_attn.in_proj_weight|in_proj_bias|out_proj.*``, ``layers.{i}.cross_attn.*``,
``layers.{i}.ffn.{0,3}.*``, ``layers.{i}.norm_{self,cross,ffn}.*``, ``classifier.{0,1,4}.*``,
``projection_layer.*``; TFAM/models/AMO_CLIP.py:19-34,81-86) so its checkpoints load ``strict=True``.

Reference behaviours kept on purpose: the mean-pool runs over ALL T rows
including padded ones (:170); with ``use_pe`` the positional encoding is added IN PLACE to the caller's
tensors (:133-134); the FFN activation is ReLU whatever ``activation`` says (:13,81), the classifier uses
exact GELU; in cross-attention mode every layer attends the same raw motion tokens.

Data layout: tokens are flattened to [B*T, D]; LayerNorm outputs are produced in fp32 (residual operand)
and cast once to the compute dtype (GEMM operand); the residual sums are formed in fp32 inside the GEMM
epilogues.  Dropout (train mode) uses counter-based masks inside the kernels.
"""
from __future__ import annotations

import itertools
import math

import torch
import torch.nn as nn

from ... import autograd_ops as ag
from ... import ops
from ...clip_vit import _Lin, _LN


class _MHA(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, d_model, num_heads, dropout):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = d_model, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = _Lin(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.kaiming_uniform_(self.out_proj.weight, a=math.sqrt(5))
        nn.init.zeros_(self.out_proj.bias)


class _Slot(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential indices aligned with the reference."""

    def forward(self, x):
        return x


def _linear_init(lin):
    nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
    bound = 1 / math.sqrt(lin.weight.shape[1])
    nn.init.uniform_(lin.bias, -bound, bound)
    return lin


class AttentionLayer(nn.Module):
    def __init__(self, d_model: int, num_heads: int, dim_feedforward: int, dropout: float = 0.1, activation: str = "relu"):
        super().__init__()
        assert d_model % num_heads == 0, f"d_model ({d_model}) debe ser divisible por num_heads ({num_heads})"
        self.self_attn = _MHA(d_model, num_heads, dropout)
        self.cross_attn = _MHA(d_model, num_heads, dropout)
        # Sequential(Linear, act, Dropout, Linear, Dropout): parameters at indices 0 and 3 (:23-29)
        self.ffn = nn.Sequential(_linear_init(_Lin(d_model, dim_feedforward)), _Slot(), _Slot(),
                                 _linear_init(_Lin(dim_feedforward, d_model)), _Slot())
        self.ffn_act = ops.ACT_GELU_ERF if activation == "gelu" else ops.ACT_RELU
        self.norm_self, self.norm_cross, self.norm_ffn = _LN(d_model), _LN(d_model), _LN(d_model)
        self.p = dropout
        self.d_model, self.num_heads = d_model, num_heads

    # x32 / x16: the token matrix [B*T, D] as fp32 (residual operand) and in the compute dtype (GEMM operand)
    def run(self, x32, x16, B, T, dt16, mask_u8, cross16=None, Tk=0, cross_mask_u8=None, seed_fn=None):
        H, D = self.num_heads, self.d_model
        tr = self.training
        p = self.p if tr else 0.0
        fused = D % 256 == 0 and D <= 2048 and getattr(self, "fuse_tail", True)   # fuse_tail=False: separate dropout / add / LN (tests)

        def tail(xr, branch, norm, extra_drop=False):   # LN(x + dropout(branch)) -> (fp32, 16-bit)
            # extra_drop: the branch first passes the FFN's own trailing nn.Dropout (:28), then the block dropout (:50)
            if fused:      # dropout masks applied inside the add + LayerNorm kernel (same seeds / masks as ag.dropout would use)
                d1 = (p, seed_fn()) if p > 0 and extra_drop else None
                d2 = (p, seed_fn()) if p > 0 else None
                drops = (d1, d2) if d1 else ((d2, (0.0, 0)) if d2 else ((0.0, 0), (0.0, 0)))
                return ag.postnorm(xr, branch, norm.weight, norm.bias, drops)
            if extra_drop:
                branch = ag.dropout(branch, p, tr, seed_fn)
            branch = ag.dropout(branch, p, tr, seed_fn)
            y = ag.layernorm(_add32(xr, branch, dt16), norm.weight, norm.bias, dt16, out_f32=True)
            a, b2 = ag.fork(y, dt16)
            return a, ag.cast(b2, dt16)

        qkv = ag.linear(x16, self.self_attn.in_proj_weight, self.self_attn.in_proj_bias)
        o = _SelfAttn.apply(qkv, mask_u8, B, T, H, p, seed_fn() if p > 0 else 0)
        o = ag.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias)
        x32, x16 = tail(x32, o, self.norm_self)
        if cross16 is not None:
            W, b = self.cross_attn.in_proj_weight, self.cross_attn.in_proj_bias
            q = ag.linear(x16, W, b, rows=(0, D))              # _in_projection_packed: q from rows 0:D,
            kv = ag.linear(cross16, W, b, rows=(D, 3 * D))      # k | v from rows D:3D of the packed in_proj
            o = _CrossAttn.apply(q, kv, cross_mask_u8, B, T, Tk, H, p, seed_fn() if p > 0 else 0)
            o = ag.linear(o, self.cross_attn.out_proj.weight, self.cross_attn.out_proj.bias)
            x32, x16 = tail(x32, o, self.norm_cross)
        h = ag.linear(x16, self.ffn[0].weight, self.ffn[0].bias, act=self.ffn_act)
        h = ag.dropout(h, p, tr, seed_fn)
        f = ag.linear(h, self.ffn[3].weight, self.ffn[3].bias)
        return tail(x32, f, self.norm_ffn, extra_drop=True)                # ffn's trailing Dropout (:28) + self.dropout (:50), both in tail()


class _Add32(torch.autograd.Function):
    """y(f32) = a(f32) + b(16-bit): the residual sum feeding a post-norm LayerNorm."""

    @staticmethod
    def forward(ctx, a, b, dt16):
        ctx.meta = (b.dtype, dt16)
        return ag._add(a, b, torch.float32, dt16)

    @staticmethod
    def backward(ctx, dy):
        bdtype, dt16 = ctx.meta
        dy = dy.contiguous()
        return dy, ops.cast16(dy, bdtype), None


def _add32(a, b, dt16):
    return _Add32.apply(a, b, dt16)


class _SelfAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, mask, B, T, H, p, seed):
        D = qkv.shape[1] // 3
        dh = D // H
        out, lse = ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], mask, B, H, T, T, dh, want_lse=True,
                                 dropout_p=p, dropout_seed=seed)
        ctx.save_for_backward(qkv, out, lse, mask)
        ctx.meta = (B, T, H, D, dh, p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, mask = ctx.saved_tensors
        B, T, H, D, dh, p, seed = ctx.meta
        dout = ops.cast16(dout.contiguous(), qkv.dtype)
        dqkv = torch.empty_like(qkv)
        ag._attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], mask, out, dout, lse,
                     dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, T, T, dh, p, seed)
        return dqkv, None, None, None, None, None, None


class _CrossAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, mask, B, Tq, Tk, H, p, seed):
        D = q.shape[1]
        dh = D // H
        out, lse = ops.attention(q, kv[:, :D], kv[:, D:], mask, B, H, Tq, Tk, dh, want_lse=True, dropout_p=p, dropout_seed=seed)
        ctx.save_for_backward(q, kv, out, lse, mask)
        ctx.meta = (B, Tq, Tk, H, D, dh, p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse, mask = ctx.saved_tensors
        B, Tq, Tk, H, D, dh, p, seed = ctx.meta
        dout = ops.cast16(dout.contiguous(), q.dtype)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ag._attn_bwd(q, kv[:, :D], kv[:, D:], mask, out, dout, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Tq, Tk, dh, p, seed)
        return dq, dkv, None, None, None, None, None, None, None


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _mask_u8(mask, dev):
    """[B,T] mask (True / 1 = real token) as a uint8 device tensor.  torch.bool is one byte holding 0 / 1: it is VIEWED, not
    cast (a cast is one more kernel launch per mask per forward, 8 us of a 210 us forward at the reference batch)."""
    if mask is None:
        return None
    if not mask.is_cuda:
        mask = mask.to(dev)
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask.to(dtype=torch.uint8).contiguous()


class AMO_CLIP(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, use_cross_attention=True,
                 use_pe=False, use_only_rgb=False, use_only_flow=False, concat_dim=1, dropout=0.1, mlp_dropout=0.3,
                 device="cuda", compute_dtype=torch.bfloat16):
        super().__init__()
        self.use_cross_attention, self.use_pe = use_cross_attention, use_pe
        self.use_only_rgb, self.use_only_flow, self.concat_dim = use_only_rgb, use_only_flow, concat_dim
        self.d_model, self.device, self.nhead = d_model, device, nhead
        self.compute_dtype = compute_dtype
        self.mlp_dropout = mlp_dropout
        self.layers = nn.ModuleList([AttentionLayer(d_model, nhead, dim_feedforward, dropout=dropout) for _ in range(num_layers)])
        # Sequential(LayerNorm, Linear, GELU, Dropout, Linear): parameters at indices 0, 1, 4 (:84)
        self.classifier = nn.Sequential(_LN(d_model), _linear_init(_Lin(d_model, d_model // 2)), _Slot(), _Slot(),
                                        _linear_init(_Lin(d_model // 2, num_classes)))
        self.projection_layer = _linear_init(_Lin(2 * d_model, d_model))
        self.fused_inference = True     # eval + no_grad forwards of short clips run the fused launch chain (tfam_fused.py)
        self.fused_training = True      # train-mode forwards of short clips run the fused training chains (tfam_train.py)
        self.set_dropout_seed(0x5EED)

    def set_dropout_seed(self, seed: int):
        """Per-rank / per-run base seed of the counter-based dropout masks; restarts the mask sequence."""
        self._seed_base = int(seed)
        self._seed_iter = itertools.count(1)

    def use_device_seeds(self, optimizer):
        """Draw dropout seeds from ``optimizer``'s device-resident step state (FusedAdam.enable_device_state): every call site
        of a step gets the ADDRESS of its seed, which vmc_train_tick rewrites before each step -- what a captured step needs."""
        self._seed_source = optimizer

    def _next_seed(self):
        src = getattr(self, "_seed_source", None)
        if src is not None and self.training:
            i = self._seed_site = getattr(self, "_seed_site", -1) + 1
            return src.seed_address(i)
        return ((self._seed_base << 24) ^ next(self._seed_iter)) & 0x7FFFFFFFFFFFFFFF      # bit 63 is the pointer tag (include/vmc.h)

    def positional_encoding(self, seq_len):
        """Sinusoidal table [seq_len, d_model] (:88-97), produced by the same kernel that adds it."""
        pe = torch.zeros(1, seq_len, self.d_model, device=self.device)
        return ops.add_sinusoidal_pe_(pe)[0]

    def parameter_groups_by_layer(self):
        """[layer 0, ..., layer L-1, classifier]: the used parameters whose gradients the fused training chains complete together
        (tfam_train.TfamTrainFn.backward reports group i through ``grad_group_callback(i)``; optim.FusedAdam.enable_backward_overlap)."""
        used = {id(p) for p in self.used_parameters()}
        groups = [[p for p in layer.parameters() if id(p) in used] for layer in self.layers]
        return groups + [[p for p in self.classifier.parameters() if id(p) in used]]

    def used_parameters(self):
        """Parameters that receive a gradient in the configured fusion mode (the DDP all-reduce set;
        SURVEY.md §7 'DDP with unused parameters')."""
        cross = self.use_cross_attention and not (self.use_only_rgb or self.use_only_flow)
        proj = (not self.use_only_rgb and not self.use_only_flow and not self.use_cross_attention and self.concat_dim == -1)
        out = []
        for n, p in self.named_parameters():
            if ".cross_attn." in n or ".norm_cross." in n:
                if cross:
                    out.append(p)
            elif n.startswith("projection_layer."):
                if proj:
                    out.append(p)
            else:
                out.append(p)
        return out

    def _fused_inputs(self, rgb_emb, motion_emb, m_rgb, m_flow, training):
        """(x, mask, motion, mask_kv, cross) of the fused chains for the configured fusion mode, or None when the mode / shapes are
        outside their set (the per-op path then runs)."""
        from ... import tfam_fused as tf
        from ... import tfam_train as tt
        dt16 = self.compute_dtype
        motion, m_kv, cross = None, None, False
        if self.use_only_rgb:
            x, m = rgb_emb, m_rgb
        elif self.use_only_flow:
            x, m = motion_emb, m_flow
        elif self.use_cross_attention:
            x, m, motion, m_kv, cross = rgb_emb, m_rgb, motion_emb, m_flow, True
        else:
            rgb_cut = rgb_emb[:, :-1, :]
            m_cut = m_rgb[:, :-1] if m_rgb is not None else None
            if self.concat_dim == 1:
                x = torch.cat([rgb_cut, motion_emb], dim=1)                   # token concat: memory plumbing
                m = torch.cat([m_cut, m_flow], dim=1).contiguous() if m_cut is not None else None
            else:
                if training:                          # the projection layer's gradient needs d(tokens): per-op path
                    return None
                xcat = torch.cat([rgb_cut, motion_emb], dim=-1)
                if not tf.supported(self, xcat.shape[0], xcat.shape[1], 0, False):
                    return None
                x = ag.linear(ag.cast(_f32c(xcat).view(-1, xcat.shape[-1]), dt16), self.projection_layer.weight,
                              self.projection_layer.bias, out_f32=True).view(xcat.shape[0], xcat.shape[1], -1)
                m = m_flow
        ok = tt.supported if training else tf.supported
        if x.shape[-1] != self.d_model or not ok(self, x.shape[0], x.shape[1], motion.shape[1] if cross else 0, cross):
            return None
        return _f32c(x), m, (_f32c(motion) if cross else None), m_kv, cross

    def _forward_fused(self, rgb_emb, motion_emb, m_rgb, m_flow):
        """Eval forward through vmc_tfam_forward (one call: hoisted K|V GEMM + 6 launches per layer + pool + head).
        Returns None when the shapes are outside the fused chain's set; the per-op path below then runs."""
        from ... import tfam_fused as tf
        sel = self._fused_inputs(rgb_emb, motion_emb, m_rgb, m_flow, False)
        if sel is None:
            return None
        x, m, motion, m_kv, cross = sel
        pack = tf.get_pack(self, self.compute_dtype).refresh()
        return pack.forward(x, motion, m, m_kv, cross, slot=getattr(self, "fused_slot", 0))

    def _forward_fused_train(self, rgb_emb, motion_emb, m_rgb, m_flow):
        """Train-mode forward + (through autograd) backward as the fused launch chains of tfam_train.py: one autograd node for the
        whole model.  None when the mode / shapes are outside the chain's set."""
        from ... import tfam_train as tt
        sel = self._fused_inputs(rgb_emb, motion_emb, m_rgb, m_flow, True)
        if sel is None:
            return None
        x, m, motion, m_kv, cross = sel
        return tt.forward_train(self, x, motion, m, m_kv, cross, self._next_seed)

    def forward(self, rgb_emb, motion_emb, mask_rgb=None, mask_flow=None):
        dt16, D = self.compute_dtype, self.d_model
        dev = self.device
        rgb_emb = rgb_emb if rgb_emb.is_cuda else rgb_emb.to(dev)
        motion_emb = motion_emb if motion_emb.is_cuda else motion_emb.to(dev)
        if self.use_pe:                                   # in place on the caller's tensors, as the reference (:133-134)
            ops.add_sinusoidal_pe_(rgb_emb)
            ops.add_sinusoidal_pe_(motion_emb)
        B = rgb_emb.shape[0]
        m_rgb, m_flow = _mask_u8(mask_rgb, dev), _mask_u8(mask_flow, dev)
        seed_fn = self._next_seed
        self._seed_site = -1                              # device-seed mode: call sites are numbered from 0 in every forward
        if self.fused_inference and not self.training and not torch.is_grad_enabled():
            out = self._forward_fused(rgb_emb, motion_emb, m_rgb, m_flow)
            if out is not None:
                return out
        if self.fused_training and self.training and torch.is_grad_enabled():
            out = self._forward_fused_train(rgb_emb, motion_emb, m_rgb, m_flow)
            if out is not None:
                return out
            self._seed_site = -1                          # nothing was drawn on a path that declined

        def flat(t):
            return t.contiguous().float().view(-1, t.shape[-1])

        def pair(x32):
            a, b2 = ag.fork(x32, dt16)
            return a, ag.cast(b2, dt16)

        if self.use_only_rgb:
            (x, x16), T, m = pair(flat(rgb_emb)), rgb_emb.shape[1], m_rgb
            for layer in self.layers:
                x, x16 = layer.run(x, x16, B, T, dt16, m, seed_fn=seed_fn)
        elif self.use_only_flow:
            (x, x16), T, m = pair(flat(motion_emb)), motion_emb.shape[1], m_flow
            for layer in self.layers:
                x, x16 = layer.run(x, x16, B, T, dt16, m, seed_fn=seed_fn)
        elif self.use_cross_attention:
            (x, x16), T, Tk = pair(flat(rgb_emb)), rgb_emb.shape[1], motion_emb.shape[1]
            cross16 = ag.cast(flat(motion_emb), dt16)
            for layer in self.layers:
                x, x16 = layer.run(x, x16, B, T, dt16, m_rgb, cross16=cross16, Tk=Tk, cross_mask_u8=m_flow, seed_fn=seed_fn)
        else:
            rgb_cut = rgb_emb[:, :-1, :]
            m_cut = m_rgb[:, :-1] if m_rgb is not None else None
            if self.concat_dim == 1:
                xcat = torch.cat([rgb_cut, motion_emb], dim=1)                 # token concat: memory plumbing
                m = torch.cat([m_cut, m_flow], dim=1).contiguous() if m_cut is not None else None
                x, T = flat(xcat), xcat.shape[1]
            else:
                xcat = torch.cat([rgb_cut, motion_emb], dim=-1)
                T = xcat.shape[1]
                x = ag.linear(ag.cast(flat(xcat), dt16), self.projection_layer.weight, self.projection_layer.bias, out_f32=True)
                m = m_flow
            x, x16 = pair(x)
            for layer in self.layers:
                x, x16 = layer.run(x, x16, B, T, dt16, m, seed_fn=seed_fn)
        pooled = ag.MeanPoolFn.apply(x, B, T, dt16, True)                                      # [B, D] f32, all T rows
        h = ag.layernorm(pooled, self.classifier[0].weight, self.classifier[0].bias, dt16)
        h = ag.linear(h, self.classifier[1].weight, self.classifier[1].bias, act=ops.ACT_GELU_ERF)
        h = ag.dropout(h, self.mlp_dropout, self.training, seed_fn)
        return ag.linear(h, self.classifier[4].weight, self.classifier[4].bias, out_f32=True)
