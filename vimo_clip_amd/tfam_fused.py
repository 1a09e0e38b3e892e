"""Fused TFAM inference chain (vmc_tfam_* of include/vmc.h) behind ``AMO_CLIP.forward``.

The reference's eval forward (TFAM/models/AMO_CLIP.py:99-171) for short clips: one hoisted K|V GEMM over the raw
motion tokens, six launches per AttentionLayer, pool + two head launches, all from ONE ctypes call.  The module's
fp32 parameters stay the state-dict-compatible masters; this file keeps the two packs the kernels read:

* ``wpack``  16-bit compute copies of every matrix, in the order ``vmc_tfam_pack_offset`` defines,
* ``ppack``  fp32 biases and LayerNorm parameters.

Both live in buffers allocated ONCE per (module, dtype): re-packing after an optimiser step rewrites them in place,
so a captured hipGraph of the forward keeps reading valid, current weights (the packs are refreshed outside the
graph, before a replay).  ``pack_is_current`` compares parameter versions, data pointers and the global weight epoch
that ``optim.invalidate_weight_copies`` bumps when an optimiser kernel rewrites the masters behind autograd's back.
"""
from __future__ import annotations

import torch

from . import autograd_ops as ag
from ._lib import check, dt, lib, ptr, stream

# slot ids of include/vmc.h
W_SELF_IN, W_SELF_OUT, W_CROSS_Q, W_CROSS_OUT, W_FFN0, W_FFN3, W_KV_ALL, W_CLS1, W_CLS4, W_END = range(10)
(P_SELF_IN_B, P_SELF_OUT_B, P_CROSS_Q_B, P_CROSS_OUT_B, P_FFN0_B, P_FFN3_B, P_NORM_SELF, P_NORM_CROSS, P_NORM_FFN,
 P_KV_ALL_B, P_CLS_LN, P_CLS1_B, P_CLS4_B, P_END) = range(16, 30)

MAX_T = 64        # tokens per clip (TF_MAX_T of csrc/tfam_kernels.h): queries in parts of 32 per row block, up to four key tiles
MAX_ROWS = 256      # B*T above which the per-op path (256x256 GEMM tiles) wins: measured crossover B = 16 at T = 16 (profiles/README.md)


def supported(model, B, T, Tk, has_cross) -> bool:
    """Shapes the fused chain covers (everything else takes the per-op path)."""
    D, H = model.d_model, model.nhead
    ff = model.layers[0].ffn[0].weight.shape[0]
    if D not in (512, 768) or D % H or D // H not in (64, 96):
        return False
    if not 0 < T <= MAX_T or (has_cross and not 0 < Tk <= MAX_T) or ff % 512 or B * T > MAX_ROWS:
        return False
    if ((2 if T <= 16 else 1) * H) % 4:
        return False
    return model.layers[0].ffn_act == 3      # ReLU (ops.ACT_RELU): what the reference always runs (:13,81)


class TfamPack:
    MAX_UNPINNED = 16       # scratch buffers of eager forwards kept around (each a few MB)

    def __init__(self, model, dtype16):
        self.model, self.dtype16 = model, dtype16
        self.D, self.H, self.L = model.d_model, model.nhead, len(model.layers)
        self.ff = model.layers[0].ffn[0].weight.shape[0]
        self.C = model.classifier[4].weight.shape[0]
        dev = model.classifier[4].weight.device
        self.wpack = torch.empty(self._off(W_END, 0), dtype=dtype16, device=dev)
        self.ppack = torch.zeros(self._off(P_END, 0), dtype=torch.float32, device=dev)
        self._key = None
        self._ws = {}
        self._pinned = set()

    def _off(self, slot, layer):
        off = lib.vmc_tfam_pack_offset(slot, layer, self.D, self.ff, self.L, self.C)
        if off < 0:
            raise RuntimeError(f"vmc_tfam_pack_offset: bad slot {slot}")
        return int(off)

    def _state_key(self):
        ps = list(self.model.parameters())
        return (ag.weights.epoch, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))

    def pack_is_current(self) -> bool:
        return self._key == self._state_key()

    def _put_w(self, slot, layer, w):
        w = w.detach()
        rows, cols = w.shape
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        off = self._off(slot, layer)
        dst = self.wpack[off:off + rows * cols]
        check(lib.vmc_cast_weight(ptr(w), ptr(dst), None, rows, cols, cols, 0, dt(self.dtype16), stream()), "cast_weight")

    def _put_folded(self, wslot, bslot, layer, w, b, norm):
        """Linear fed by LayerNorm `norm`: weight columns scaled by gamma, bias += W beta (vmc_tfam_fold_layernorm)."""
        w, b = w.detach(), b.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        b = b.float().contiguous()
        rows, cols = w.shape
        woff, boff = self._off(wslot, layer), self._off(bslot, layer)
        check(lib.vmc_tfam_fold_layernorm(ptr(w), ptr(b), ptr(norm.weight.detach()), ptr(norm.bias.detach()),
                                          ptr(self.wpack[woff:woff + rows * cols]), ptr(self.ppack[boff:boff + rows]), rows, cols,
                                          dt(self.dtype16), stream()), "tfam_fold_layernorm")

    def _put_p(self, slot, layer, *vals):
        off = self._off(slot, layer)
        for v in vals:
            v = v.detach().reshape(-1)
            self.ppack[off:off + v.numel()].copy_(v)
            off += v.numel()

    def refresh(self):
        """(Re)fill both packs from the module's current parameters (in place: the buffers never move)."""
        if self.pack_is_current():
            return self
        D, m = self.D, self.model
        cross = m.use_cross_attention and not (m.use_only_rgb or m.use_only_flow)
        for i, layer in enumerate(m.layers):
            sa, ca = layer.self_attn, layer.cross_attn
            if i == 0:                                    # layer 0 reads the raw tokens: nothing to fold
                self._put_w(W_SELF_IN, i, sa.in_proj_weight)
                self._put_p(P_SELF_IN_B, i, sa.in_proj_bias)
            else:
                self._put_folded(W_SELF_IN, P_SELF_IN_B, i, sa.in_proj_weight, sa.in_proj_bias, m.layers[i - 1].norm_ffn)
            self._put_w(W_SELF_OUT, i, sa.out_proj.weight)
            self._put_folded(W_CROSS_Q, P_CROSS_Q_B, i, ca.in_proj_weight[:D], ca.in_proj_bias[:D], layer.norm_self)
            self._put_w(W_CROSS_OUT, i, ca.out_proj.weight)
            self._put_folded(W_FFN0, P_FFN0_B, i, layer.ffn[0].weight, layer.ffn[0].bias, layer.norm_cross if cross else layer.norm_self)
            self._put_w(W_FFN3, i, layer.ffn[3].weight)
            self._put_w(W_KV_ALL, i, ca.in_proj_weight[D:])
            self._put_p(P_SELF_OUT_B, i, sa.out_proj.bias)
            self._put_p(P_CROSS_OUT_B, i, ca.out_proj.bias)
            self._put_p(P_FFN3_B, i, layer.ffn[3].bias)
            self._put_p(P_NORM_SELF, i, layer.norm_self.weight, layer.norm_self.bias)
            self._put_p(P_NORM_CROSS, i, layer.norm_cross.weight, layer.norm_cross.bias)
            self._put_p(P_NORM_FFN, i, layer.norm_ffn.weight, layer.norm_ffn.bias)
            self._put_p(P_KV_ALL_B, i, ca.in_proj_bias[D:])
        self._put_w(W_CLS1, 0, m.classifier[1].weight)
        self._put_w(W_CLS4, 0, m.classifier[4].weight)
        self._put_p(P_CLS_LN, 0, m.classifier[0].weight, m.classifier[0].bias)
        self._put_p(P_CLS1_B, 0, m.classifier[1].bias)
        self._put_p(P_CLS4_B, 0, m.classifier[4].bias)
        self._key = self._state_key()
        return self

    def workspace(self, B, T, Tk, has_cross, slot=0):
        """Scratch of one forward.  `slot`: forwards that may be in flight at the same time (several streams / graphs over
        independent batches) need a scratch buffer each."""
        key = (B, T, Tk, has_cross, slot)
        ws = self._ws.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if ws is None:
            if capturing:
                raise RuntimeError("TfamPack.workspace: scratch must exist before a capture (GraphedCallable's warm-up allocates it)")
            n = lib.vmc_tfam_workspace_bytes(B, T, Tk, self.D, self.ff, self.L, self.C, int(has_cross))
            if len(self._ws) >= self.MAX_UNPINNED + len(self._pinned):
                # only scratch that no hipGraph has baked into its launches may be dropped (clearing everything freed
                # buffers live graphs still wrote to); the oldest unpinned entry goes
                for k in list(self._ws):
                    if k not in self._pinned:
                        del self._ws[k]
                        break
            ws = self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.wpack.device)
        if capturing:
            self._pinned.add(key)                      # its address is now part of a graph: never freed while this pack lives
        return ws

    def forward(self, x, motion, mask, mask_kv, has_cross, slot=0):
        """x [B,T,D] fp32 tokens, motion [B,Tk,D] fp32 (cross mode) or None, masks uint8 [B,T] / [B,Tk] or None."""
        B, T, D = x.shape
        Tk = motion.shape[1] if has_cross else 0
        ws = self.workspace(B, T, Tk, has_cross, slot)
        logits = torch.empty((B, self.C), dtype=torch.float32, device=x.device)
        check(lib.vmc_tfam_forward(ptr(x), ptr(motion) if has_cross else None, ptr(mask), ptr(mask_kv) if has_cross else None,
                                   ptr(self.wpack), ptr(self.ppack), ptr(logits), ptr(ws), ws.numel(), B, T, Tk, D, self.H, self.ff,
                                   self.L, self.C, int(has_cross), dt(self.dtype16), stream()), "tfam_forward")
        return logits


def get_pack(model, dtype16) -> TfamPack:
    packs = model.__dict__.setdefault("_tfam_packs", {})
    p = packs.get(dtype16)
    if p is None or p.wpack.device != model.classifier[4].weight.device:
        p = packs[dtype16] = TfamPack(model, dtype16)
    return p
