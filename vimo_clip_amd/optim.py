"""Flat parameter/gradient arena, fused Adam/AdamW (K15) and the reference's LR schedule.

``GradArena`` lays the parameters that receive gradients out in ONE flat fp32 buffer (plus a flat
gradient buffer, Adam m and v): every ``p.data`` / ``p.grad`` becomes a 64-element-aligned view, the
backward kernels write gradients in place (autograd_ops._grad_out), the optimiser is one vmc_adam_step
launch over the whole arena (28 B/parameter of HBM traffic), and data-parallel training all-reduces the
flat gradient buffer in large buckets (parallel.py).  Mirrors ``torch.optim.Adam(lr)`` (train.py:66) and
``torch.optim.AdamW(lr=1e-4, weight_decay=0.1)`` + ``CosineAnnealingLR(T_max=epochs, eta_min=1e-6)``
(TFAM/train_and_eval.py:53-56) in arithmetic.
"""
from __future__ import annotations

import math
import os

import torch

from ._lib import check, lib, ptr, stream

_ALIGN = 64


class GradArena:
    def __init__(self, params):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("GradArena: no trainable parameters")
        dev = params[0].device
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(params, self.offsets):
            n = p.numel()
            self.flat_param[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + n].view(p.shape)
            g = self.flat_grad[o:o + n].view(p.shape)
            p.grad = g
            p._vmc_grad = g

    def zero_grad(self):
        """Not needed between steps (every used gradient is overwritten by its backward kernel); provided
        for loops that call it anyway."""
        self.flat_grad.zero_()

    def grad_norm(self) -> torch.Tensor:
        from . import autograd_ops
        autograd_ops.wgrad_queue.flush()
        out = torch.zeros(1, dtype=torch.float32, device=self.flat_grad.device)
        check(lib.vmc_sumsq(ptr(self.flat_grad), self.numel, ptr(out), stream()), "sumsq")
        return out.sqrt()

    def buckets(self, bucket_bytes: int = 48 << 20):
        """Contiguous views of the flat gradient buffer of at most ``bucket_bytes`` each (all-reduce units)."""
        n = max(_ALIGN, bucket_bytes // 4 // _ALIGN * _ALIGN)
        return [self.flat_grad[s:min(self.numel, s + n)] for s in range(0, self.numel, n)]


def clipped_grad_scale(sum_grad_norm: float, grad_scale: float, max_grad_norm: float) -> float:
    """Scale to apply to the (rank-summed) arena gradient so that it equals ``clip_grad_norm_`` applied to the AVERAGED
    gradient (train.py:105-106 runs on one process; under data parallelism the arena holds the SUM until ``grad_scale`` =
    1/world is applied inside the Adam kernel): norm of the average = ||sum|| * grad_scale, torch's coefficient
    min(1, max_norm / (norm + 1e-6)), final factor = grad_scale * coefficient."""
    total = sum_grad_norm * abs(grad_scale)
    return grad_scale * min(1.0, max_grad_norm / (total + 1e-6))


class FusedAdam:
    """Adam / AdamW over a GradArena: one kernel launch per step."""

    def __init__(self, arena: GradArena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        self.arena = arena
        self.lr, self.betas, self.eps, self.weight_decay, self.decoupled = lr, betas, eps, weight_decay, decoupled
        self.m = torch.zeros_like(arena.flat_param)
        self.v = torch.zeros_like(arena.flat_param)
        self.step_count = 0
        self.param_groups = [{"lr": lr}]          # what LR schedulers / loggers poke at

    def zero_grad(self, set_to_none: bool = False):
        pass                                       # gradients are overwritten in place every backward

    # ---- device-resident step state: what a hipGraph-captured training step needs ----------------------------
    N_SEEDS = 256                                  # dropout call sites per step a model may draw

    def enable_device_state(self, base_seed: int = 0x5EED):
        """Move the step count / bias corrections / learning rate / dropout seeds into device memory (vmc_train_tick,
        vmc_adam_step_dev): ``tick()`` + ``step()`` then enqueue only launches whose arguments never change, so a captured
        step replays correctly.  ``sync_hyper()`` must be called (outside a capture) whenever lr or grad_scale change."""
        dev = self.arena.flat_param.device
        self.dev_state = torch.zeros(2 + self.N_SEEDS, dtype=torch.int64, device=dev)
        self.dev_state[0] = self.step_count
        self.dev_state[1] = int(base_seed) & 0x7FFFFFFFFFFFFFFF
        self.dev_hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self.sync_hyper()
        return self

    def sync_hyper(self, grad_scale: float = 1.0):
        want = (float(self.param_groups[0]["lr"]), float(grad_scale))
        if self._hyper_host != want:
            self.dev_hyper[0] = want[0]
            self.dev_hyper[3] = want[1]
            self._hyper_host = want

    def tick(self):
        """Start of a step in device-state mode: t += 1, bias corrections and this step's dropout seeds (one launch)."""
        check(lib.vmc_train_tick(ptr(self.dev_state), ptr(self.dev_hyper), float(self.betas[0]), float(self.betas[1]), self.N_SEEDS,
                                 stream()), "train_tick")
        self.step_count += 1

    def seed_address(self, i: int) -> int:
        """Seed argument (pointer form, include/vmc.h) of dropout call site i of the current step."""
        if not 0 <= i < self.N_SEEDS:
            raise IndexError("more dropout call sites per step than FusedAdam.N_SEEDS")
        return (1 << 63) | (self.dev_state.data_ptr() + 8 * (2 + i))

    # ---- optimiser step overlapped with the backward ----------------------------------------------------------------------
    def enable_backward_overlap(self, groups):
        """groups: lists of parameters, each a contiguous run of the arena, whose gradients become final TOGETHER during the
        backward (a TFAM layer; the classifier head).  ``group_ready(i)`` -- called by the backward the moment group i's
        gradients have been enqueued (tfam_train.TfamTrainFn.backward) -- then enqueues that range's AdamW update and the refresh
        of its 16-bit compute copies on a side stream, beside the rest of the backward: the update is HBM-bound (28 B per
        parameter), the backward of the layers below is a chain of latency-bound launches that read none of the updated
        buffers.  ``step()`` afterwards updates whatever is left and joins the side stream.  Same arithmetic, element for
        element, as one whole-arena step.  Only for a single process without gradient clipping (both need the complete
        gradient first); ``step(grad_scale != 1)`` / ``max_grad_norm`` after a group was already updated raise."""
        a = self.arena
        index = {id(p): i for i, p in enumerate(a.params)}
        ranges = []
        for g in groups:
            idx = sorted(index[id(p)] for p in g if id(p) in index)
            if not idx:
                continue
            if idx != list(range(idx[0], idx[-1] + 1)):
                raise ValueError("enable_backward_overlap: a group must be a contiguous run of the arena's parameters")
            lo = a.offsets[idx[0]]
            hi = a.offsets[idx[-1] + 1] if idx[-1] + 1 < len(a.params) else a.numel
            ranges.append((lo, hi, frozenset(id(a.params[i]) for i in idx)))
        covered = sorted((lo, hi) for lo, hi, _ in ranges)
        rest, pos = [], 0
        for lo, hi in covered:
            if lo < pos:
                raise ValueError("enable_backward_overlap: groups overlap")
            if lo > pos:
                rest.append((pos, lo))
            pos = hi
        if pos < a.numel:
            rest.append((pos, a.numel))
        rest_ids = frozenset(id(p) for p in a.params) - frozenset().union(*[r[2] for r in ranges]) if ranges else frozenset(id(p) for p in a.params)
        self._ov = {"ranges": ranges, "rest": rest, "rest_ids": rest_ids, "done": [False] * len(ranges),
                    "stream": torch.cuda.Stream(device=a.flat_param.device, priority=int(os.environ.get("VMC_ADAM_SIDE_PRIO", "0"))),
                    "used": False}
        return self

    def disable_backward_overlap(self):
        self._ov = None

    def _open_step(self):
        if not getattr(self, "_step_open", False):
            if getattr(self, "dev_state", None) is None:
                self.step_count += 1                   # device-state mode: tick() already advanced t
            self._step_open = True

    BG_WORKGROUPS = int(os.environ.get("VMC_ADAM_BG_WGS", "512"))      # side-stream update: 2 workgroups of 256 threads per CU

    def _update_range(self, lo, hi, grad_scale=1.0, background=False):
        a, n = self.arena, hi - lo
        if n <= 0:
            return
        p, g, m, v = a.flat_param[lo:hi], a.flat_grad[lo:hi], self.m[lo:hi], self.v[lo:hi]
        if getattr(self, "dev_state", None) is not None:
            check(lib.vmc_adam_step_dev_bg(ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(self.dev_hyper), float(self.betas[0]), float(self.betas[1]),
                                           float(self.eps), float(self.weight_decay), int(self.decoupled),
                                           self.BG_WORKGROUPS if background else 2048, stream()), "adam_step_dev")
        else:
            check(lib.vmc_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), n, float(self.param_groups[0]["lr"]), float(self.betas[0]),
                                    float(self.betas[1]), float(self.eps), float(self.weight_decay), int(self.decoupled), self.step_count,
                                    float(grad_scale), stream()), "adam_step")

    def group_ready(self, i: int):
        """The gradients of overlap group i are complete (enqueued on the current stream): update it now, beside the backward."""
        ov = getattr(self, "_ov", None)
        if ov is None or i >= len(ov["ranges"]) or ov["done"][i]:
            return
        lo, hi, ids = ov["ranges"][i]
        self._open_step()
        side = ov["stream"]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._update_range(lo, hi, background=True)
            from . import autograd_ops
            autograd_ops.weights.refresh(owner=(id(self), i), param_ids=ids)
        ov["done"][i] = ov["used"] = True

    def step(self, grad_scale: float = 1.0, max_grad_norm=None):
        from . import autograd_ops
        autograd_ops.wgrad_queue.flush()               # normally empty: the backward pass flushes its grouped weight gradients itself
        a = self.arena
        dev_mode = getattr(self, "dev_state", None) is not None
        ov = getattr(self, "_ov", None)
        if dev_mode and max_grad_norm is not None:
            raise ValueError("gradient clipping needs a host read of the norm: not available in device-state mode")
        if ov is not None and ov["used"]:
            # some ranges were updated during the backward: finish the others on this stream, then join
            if max_grad_norm is not None or grad_scale != 1.0:
                raise ValueError("backward-overlapped optimiser steps cannot be combined with grad_scale / max_grad_norm")
            from . import autograd_ops
            for i, (lo, hi, ids) in enumerate(ov["ranges"]):
                if not ov["done"][i]:
                    self._update_range(lo, hi)
                    autograd_ops.weights.refresh(owner=(id(self), i), param_ids=ids)
            for lo, hi in ov["rest"]:
                self._update_range(lo, hi)
            if ov["rest"] and ov["rest_ids"]:
                autograd_ops.weights.refresh(owner=(id(self), "rest"), param_ids=ov["rest_ids"])
            torch.cuda.current_stream().wait_stream(ov["stream"])
            ov["done"] = [False] * len(ov["ranges"])
            ov["used"] = False
            self._step_open = False
            return
        if max_grad_norm is not None:              # torch.nn.utils.clip_grad_norm_ (train.py:105-106)
            grad_scale = clipped_grad_scale(float(a.grad_norm().item()), grad_scale, max_grad_norm)
        self._open_step()
        if dev_mode and self._fused_update():          # AdamW + refresh of the 16-bit copies in one pass over the masters
            self._step_open = False
            return
        self._update_range(0, a.numel, grad_scale)
        self._step_open = False
        invalidate_weight_copies(self)

    FUSED_CAST = os.environ.get("VMC_ADAM_FUSED_CAST", "1") != "0"      # builder A/B switch

    def _fused_update(self) -> bool:
        """Device-state mode: vmc_adam_cast_multi updates every matrix tile by tile and writes its 16-bit compute copies from the
        registers (one read of the masters instead of two), a second small launch covers the parameters without copies.  Needs all
        cached copies of this arena's parameters in ONE compute dtype; otherwise the two-kernel path runs."""
        if not self.FUSED_CAST:
            return False
        import numpy as np
        from . import autograd_ops
        from ._lib import dt
        a = self.arena
        ids = getattr(self, "_param_ids", None)
        if ids is None:
            ids = self._param_ids = frozenset(id(p) for p in a.params)
        tabs = autograd_ops.weights.tables(owner=id(self), param_ids=ids)
        if len(tabs) != 1:
            return False
        (dtype16, tab), = tabs.items()
        covered = tab[4]
        plan = getattr(self, "_fused_plan", None)
        if plan is None or plan[0] != covered:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("optimiser plan changed while a graph is being captured: run one eager step first")
            rest = [(o, p.numel()) for p, o in zip(a.params, a.offsets) if id(p) not in covered]
            rec = np.zeros((max(1, len(rest)), 2), dtype=np.int64)
            block0 = 0
            for i, (o, n) in enumerate(rest):
                rec[i, 0] = o
                rec[i, 1] = n | (block0 << 32)
                block0 += (n + 1023) // 1024
            plan = self._fused_plan = (covered, torch.from_numpy(rec).to(a.flat_param.device), len(rest), block0)
            self._old_plans = getattr(self, "_old_plans", []) + [plan[1]]      # a captured graph may still read an older table
        check(lib.vmc_adam_cast_multi(ptr(tab[1]), tab[2], tab[3], ptr(plan[1]) if plan[2] else None, plan[2], plan[3], ptr(a.flat_param),
                                      ptr(a.flat_grad), ptr(self.m), ptr(self.v), ptr(self.dev_hyper), float(self.betas[0]), float(self.betas[1]),
                                      float(self.eps), float(self.weight_decay), int(self.decoupled), dt(dtype16), stream()), "adam_cast_multi")
        autograd_ops.weights.epoch += 1
        return True

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        """Layout {step, m, v, lr} over the flat arena (NOT torch.optim's per-parameter state: checkpoint.py says so)."""
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups[0]["lr"] = sd["lr"]
        if getattr(self, "dev_state", None) is not None:      # device-state mode: the step count and lr the kernels read
            self.dev_state[0] = self.step_count
            self._hyper_host = None
            self.sync_hyper()


def invalidate_weight_copies(optimizer=None):
    """The optimiser kernel wrote the fp32 masters behind autograd's version counters: the cached 16-bit compute copies of
    the parameters it owns are re-cast into their buffers in one launch (autograd_ops._WeightCache.refresh)."""
    from . import autograd_ops
    if optimizer is None:
        autograd_ops.weights.refresh()
        return
    ids = getattr(optimizer, "_param_ids", None)
    if ids is None:
        ids = optimizer._param_ids = frozenset(id(p) for p in optimizer.arena.params)
    autograd_ops.weights.refresh(owner=id(optimizer), param_ids=ids)


class CosineAnnealingLR:
    """Closed form of torch's CosineAnnealingLR(T_max, eta_min), stepped once per epoch (train_and_eval.py:162)."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.opt, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.opt.param_groups[0]["lr"] = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr}
