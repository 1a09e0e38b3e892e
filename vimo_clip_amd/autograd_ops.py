"""torch.autograd.Function wrappers over the libvmc kernels (K8: autograd of the hot path, train.py:104,
TFAM/train_and_eval.py:82).

Every forward and backward body is libvmc kernels only.  Tensors that are used twice go through an
explicit fork (``LayerNormFn(passthrough=True)`` fuses the sum of the two gradients into the LayerNorm
backward kernel; ``fork`` sums with vmc_add), so autograd itself never has to add gradients with ATen ops.
Parameter gradients are written straight into ``param._vmc_grad`` (a view into a flat gradient arena,
see optim.GradArena) when that attribute exists; otherwise they are returned to autograd.

Activations are 2-D ``[rows, features]`` contiguous tensors in the compute dtype (bf16/f16) unless a
function says otherwise; parameters are fp32 masters whose 16-bit copies are cached per version.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import ops
from ._lib import check, dt, lib, ptr, stream

ACT_NONE = ops.ACT_NONE


# ------------------------------------------------------------------------------------------------
# 16-bit weight copies
# ------------------------------------------------------------------------------------------------
class _WeightCache:
    def __init__(self):
        self._c = {}
        self._tables = {}       # dtype16 -> (key, device descriptor table, n, tiles) of refresh()
        self._old_tables = []
        self.epoch = 0          # bumped whenever the masters may have changed behind autograd's version counters
        self.generation = 0     # bumped when copies are DROPPED (their buffers may be freed): pointer tables built from them are void

    def get(self, p: torch.Tensor, dtype16, transposed=False, pad_k=False, both=False):
        if not isinstance(p, torch.nn.Parameter):      # temporaries (e.g. row slices of in_proj_weight) are not cached
            return ops.cast_weight(p, dtype16, transposed=transposed, pad_k=pad_k)
        key = (id(p), dtype16, transposed, pad_k)
        hit = self._c.get(key)
        # id() values are recycled once a parameter dies: the weak reference proves it is still the same object
        if hit is not None and hit[0]() is p and hit[1] == p._version and hit[2] == p.data_ptr():
            return hit[3]
        # a parameter that is being trained needs both copies every step (forward + dgrad): one launch makes both.
        # (pad_k is by construction "contraction length % 64 != 0" at every call site, which is what cast_weight_both pads.)
        if both and p.requires_grad and p.dim() == 2 and pad_k == ((p.shape[0] if transposed else p.shape[1]) % 64 != 0):
            w, wt = ops.cast_weight_both(p, dtype16)
            ident = (weakref.ref(p), p._version, p.data_ptr())
            self._c[(id(p), dtype16, False, p.shape[1] % 64 != 0)] = ident + (w,)
            self._c[(id(p), dtype16, True, p.shape[0] % 64 != 0)] = ident + (wt,)
            return wt if transposed else w
        w = ops.cast_weight(p, dtype16, transposed=transposed, pad_k=pad_k)
        self._c[key] = (weakref.ref(p), p._version, p.data_ptr(), w)
        return w

    def clear(self):
        self._c.clear()
        self._tables.clear()
        self.epoch += 1
        self.generation += 1

    def refresh(self, owner=None, param_ids=None):
        """The optimiser kernel rewrote the fp32 masters in place: re-cast every cached copy of the trained parameters into its
        existing buffer with ONE launch per compute dtype (vmc_cast_weights_multi) instead of dropping the copies and re-casting
        them one launch per parameter in the next forward.  Buffers keep their addresses (what a captured step needs).
        param_ids: ids of the parameters that were updated (an optimiser's arena); None = every cached trained parameter.
        owner: key under which the descriptor table of that parameter set is kept (the optimiser)."""
        for dtype16, tab in self.tables(owner, param_ids).items():
            check(lib.vmc_cast_weights_multi(ptr(tab[1]), tab[2], tab[3], dt(dtype16), stream()), "cast_weights_multi")
        self.epoch += 1

    def tables(self, owner=None, param_ids=None):
        """dtype16 -> (key, device descriptor table, n, tiles, ids of the parameters it covers): the records vmc_cast_weights_multi and
        vmc_adam_cast_multi take for every cached copy of the trained parameters in `param_ids` (stale copies are dropped)."""
        import numpy as np
        groups = {}                                   # dtype16 -> {id(p): [p, w16, w16t]}
        for (pid, dtype16, transposed, _pad), hit in list(self._c.items()):
            if param_ids is not None and pid not in param_ids:
                continue                              # another model's parameters: untouched by this step, copies stay valid
            p = hit[0]()
            if p is None or hit[2] != p.data_ptr() or hit[1] != p._version or not p.requires_grad:
                if p is None or hit[2] != p.data_ptr() or hit[1] != p._version:
                    del self._c[(pid, dtype16, transposed, _pad)]      # stale: the next get() re-casts
                continue                              # frozen parameters keep their copies as they are
            if p.dtype != torch.float32 or not p.is_contiguous():
                del self._c[(pid, dtype16, transposed, _pad)]
                continue
            slot = groups.setdefault(dtype16, {}).setdefault(pid, [p, None, None])
            slot[2 if transposed else 1] = hit[3]
        out = {}
        for dtype16, params in groups.items():
            key = (dtype16, tuple((pid, 0 if w is None else w.data_ptr(), 0 if wt is None else wt.data_ptr(), p.data_ptr())
                                  for pid, (p, w, wt) in params.items()))
            tab = self._tables.get((owner, dtype16))
            if tab is None or tab[0] != key:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("weight-copy table changed while a graph is being captured: run one eager step first "
                                       "(GraphedTrainStep's warm-up does) so that every compute copy exists before the capture")
                rec = np.zeros((len(params), 6), dtype=np.int64)
                tile0 = 0
                for i, (p, w, wt) in enumerate(params.values()):
                    rows = p.shape[0]
                    cols = p.numel() // rows
                    tx, ty = (cols + 63) // 64, (rows + 63) // 64
                    rec[i, 0] = p.data_ptr()
                    rec[i, 1] = 0 if w is None else w.data_ptr()
                    rec[i, 2] = 0 if wt is None else wt.data_ptr()
                    rec[i, 3] = rows | (cols << 32)
                    rec[i, 4] = (0 if w is None else w.stride(0)) | ((0 if wt is None else wt.stride(0)) << 32)
                    rec[i, 5] = tile0 | (tx << 32)
                    tile0 += tx * ty
                dev = next(iter(params.values()))[0].device
                tab = (key, torch.from_numpy(rec).to(dev), len(params), tile0, frozenset(params.keys()))
                self._tables[(owner, dtype16)] = tab
                self._old_tables.append(tab[1])       # a captured graph may still read an older table
            out[dtype16] = tab
        return out


weights = _WeightCache()


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


grad_ready_hooks = []      # callables(param): a backward kernel has just written (part of) this parameter's arena gradient


def _deliver(param, grad):
    """Write a parameter gradient into its arena slot (if any) and tell autograd nothing is left to do."""
    if param is None or not param.requires_grad:
        return None
    slot = getattr(param, "_vmc_grad", None)
    if slot is None:
        return grad.view(param.shape) if grad.shape != param.shape else grad
    if grad.data_ptr() != slot.data_ptr():
        slot.view(-1).copy_(grad.reshape(-1))      # device-to-device copy (plumbing); GEMMs write in place below
    for hook in grad_ready_hooks:                  # data-parallel reducer: launch a bucket as soon as it is complete
        hook(param)
    return None


class _WgradQueue:
    """Weight gradients of one backward pass, batched.  A training step leaves one dW = dY^T X per linear, each too small to fill
    the chip on its own (TFAM at B = 512: 9 .. 27 output tiles over 8192 tokens; launched alone it is cut into token slices plus a
    reduce launch).  LinearFn.backward parks eligible problems here -- operands, the arena slots they write, the parameters whose
    gradient-ready hooks must fire -- and they leave as grouped launches (ops.wgrad_tn_group: every tile over all tokens, no slabs):
    when the queue is full, at the end of the backward pass (autograd engine callback) and, as a safety net, before the optimiser
    or a gradient norm reads the arena.  With gradient-ready hooks registered in a job of more than one rank (data-parallel bucket
    exchange) a group leaves as soon as it fills one round of the chip, so that buckets still go out during the backward."""
    enabled = os.environ.get("VMC_WGRAD_GROUP", "1") != "0"       # builder A/B switch
    MAX_PROBLEMS = ops.WGRAD_GROUP_MAX      # 32: the 30 linears of a 4-layer TFAM step leave as one launch (480 tiles = 1.9 rounds)
    MAX_TILES = 1024
    MAX_TILES_OVERLAPPED = 256          # one round of 256 x 256 tiles

    def __init__(self):
        self.items = []
        self.tiles = 0
        self._task = -1                 # autograd graph task (one per backward pass) whose end-of-pass callback is registered

    def push(self, dz, x, out, db, params):
        task = torch._C._current_graph_task_id()
        if task != self._task:
            # first problem of a new backward pass.  Anything still parked belongs to a pass that raised before its callback ran:
            # its slots are written again by this pass (two writers in one launch would race), so it is dropped
            self.items, self.tiles, self._task = [], 0, task
            if task >= 0:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        self.items.append((dz, x, out, db, params))
        self.tiles += ((dz.shape[1] + 255) // 256) * ((x.shape[1] + 255) // 256)
        if task < 0:                                   # not inside an engine-driven backward pass: nothing will call back
            self.flush()
            return
        if len(self.items) >= self.MAX_PROBLEMS or self.tiles >= (self.MAX_TILES_OVERLAPPED if self._overlapping() else self.MAX_TILES):
            self.flush()

    @staticmethod
    def _overlapping():
        if not grad_ready_hooks:
            return False
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _end_of_backward(self):
        self._task = -1
        self.flush()

    def flush(self):
        items, self.items, self.tiles = self.items, [], 0
        if not items:
            return
        by_dtype = {}
        for it in items:
            by_dtype.setdefault(it[0].dtype, []).append(it)
        for dtype16, group in by_dtype.items():
            for i in range(0, len(group), ops.WGRAD_GROUP_MAX):
                ops.wgrad_tn_group([(dz, x, out, db) for dz, x, out, db, _ in group[i:i + ops.WGRAD_GROUP_MAX]], dtype16)
        for it in items:
            for param in it[4]:
                for hook in grad_ready_hooks:          # data-parallel reducer: the gradient is enqueued now
                    hook(param)


wgrad_queue = _WgradQueue()


def _grad_out(param, shape):
    """Destination for a parameter gradient: the arena slot when present (written in place)."""
    slot = getattr(param, "_vmc_grad", None)
    if slot is not None and tuple(slot.shape) == tuple(shape) and slot.is_contiguous():
        return slot
    return torch.empty(shape, dtype=torch.float32, device=param.device)


# ------------------------------------------------------------------------------------------------
# Linear
# ------------------------------------------------------------------------------------------------
FUSE_POSTNORM_BWD = True      # False: vmc_layernorm_bwd2 + vmc_cast_dropout2 (the two-pass path the fused launch is tested against)


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) (+ res).  x [M,Kx] 16-bit (Kx = K or K padded to 64), W [N,K...] f32 parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act, out_f32, rows=None, inference=False):
        """rows = (lo, hi): use only parameter rows [lo, hi) (the q / kv thirds of a packed in_proj, as
        F.multi_head_attention_forward's _in_projection_packed does) -- sliced from the cached 16-bit copy."""
        dt16 = x.dtype
        K = weight[0].numel()
        ctx.rows = rows
        w16 = weights.get(weight, dt16, pad_k=(K % 64 != 0), both=not inference)   # training: the dgrad copy in the same launch
        if rows is not None:
            w16 = w16[rows[0]:rows[1]]
        ctx.x_cols = x.shape[1]
        if w16.shape[1] != x.shape[1]:
            if x.shape[1] != K:
                raise ValueError(f"LinearFn: x has {x.shape[1]} columns, weight needs {K} (or {w16.shape[1]} padded)")
            xp = torch.zeros((x.shape[0], w16.shape[1]), dtype=dt16, device=x.device)   # K not a multiple of 64: zero-pad
            xp[:, :K].copy_(x)
            x = xp
        b = bias.detach() if bias is not None else None
        if b is not None and rows is not None:
            b = b[rows[0]:rows[1]]
        Nw = w16.shape[0]
        if Nw % 4:                                 # odd output width (e.g. a 10-class head): zero rows up to 4n (vmc_linear: N % 4 == 0;
            if res is not None:                    # the 140 Animal-Kingdom classes go through as they are)
                raise ValueError("LinearFn: a residual needs an output width that is a multiple of 4")
            Np = (Nw + 3) // 4 * 4
            wp = torch.zeros((Np, w16.shape[1]), dtype=dt16, device=x.device)
            wp[:Nw].copy_(w16)
            w16 = wp
            if b is not None:
                bp = torch.zeros(Np, dtype=torch.float32, device=x.device)
                bp[:Nw].copy_(b)
                b = bp
        crop = (lambda t: t[:, :Nw].contiguous()) if w16.shape[0] != Nw else (lambda t: t)
        z = None
        if act != ACT_NONE:
            if res is not None:
                raise ValueError("LinearFn: activation and residual are not combined on this path")
            if inference:                          # no graph is being built: activation fused into the GEMM epilogue
                return crop(ops.linear(x, w16, bias=b, act=act, out_dtype=torch.float32 if out_f32 else dt16))
            if act == ops.ACT_RELU:                # relu'(z) = [z > 0] = [y > 0]: the fused output doubles as the saved "z"
                z = y = crop(ops.linear(x, w16, bias=b, act=act))
            elif w16.shape[0] == Nw:               # one launch: y = act(z) and the pre-activation z as a side output
                z = torch.empty((x.shape[0], Nw), dtype=dt16, device=x.device)
                y = ops.linear(x, w16, bias=b, act=act, z_out=z)
            else:
                z = crop(ops.linear(x, w16, bias=b))
                y = ops.act_fwd(z, act)
            if out_f32:
                y = ops.cast32(y)
        else:
            y = crop(ops.linear(x, w16, bias=b, res=res, out_dtype=torch.float32 if out_f32 else dt16))
        ctx.save_for_backward(x, z)
        ctx.weight, ctx.bias, ctx.act = weight, bias, act
        ctx.has_res = res is not None
        ctx.res_dtype = res.dtype if res is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z = ctx.saved_tensors
        weight, bias, act = ctx.weight, ctx.bias, ctx.act
        dt16 = x.dtype
        M = x.shape[0]
        rows = ctx.rows
        lo, hi = rows if rows is not None else (0, weight.shape[0])
        N = hi - lo
        K = weight[0].numel()
        dy = dy.contiguous()
        dy16 = ops.cast16(dy, dt16)
        dz = ops.act_bwd(z, dy16, act) if act != ACT_NONE else dy16
        dres = None
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy if dy.dtype == ctx.res_dtype else (ops.cast32(dy) if ctx.res_dtype == torch.float32 else dy16)
        # ---- dgrad: dx = dz @ W  (contraction over N; pad N to a multiple of 64 with zeros) ----
        dx = None
        if ctx.needs_input_grad[0]:
            wt = weights.get(weight, dt16, transposed=True, pad_k=(weight.shape[0] % 64 != 0))      # [K, Npad]
            if rows is not None:
                wt = wt[:, lo:hi]                                                      # K-contiguous column window
            dzp = dz
            if wt.shape[1] != N:
                dzp = torch.zeros((M, wt.shape[1]), dtype=dt16, device=dz.device)
                dzp[:, :N].copy_(dz)
            dx = ops.linear(dzp, wt)                                                    # [M, K]
            if ctx.x_cols != K:                                                         # caller passed K-padded x (patch GEMM)
                pad = torch.zeros((M, x.shape[1]), dtype=dt16, device=dz.device)
                pad[:, :K].copy_(dx)
                dx = pad
        # ---- wgrad: dW = dz^T @ x (contraction over the M tokens) and, from the same launch, db = column sums of dz ----
        dw = db = None
        want_db = bias is not None and bias.requires_grad
        db_done = False
        if weight.requires_grad:
            Kx = x.shape[1]
            slot = getattr(weight, "_vmc_grad", None)
            if N % 8 == 0 and Kx % 8 == 0:
                # TN kernel on the token-major operands as they are
                db_view = None
                if want_db:
                    bslot = getattr(bias, "_vmc_grad", None)
                    if rows is not None:
                        db = bslot if bslot is not None else torch.zeros_like(bias, dtype=torch.float32)
                        db_view = db[lo:hi]
                    else:
                        db = db_view = _grad_out(bias, (N,))
                    db_done = True
                dst = None
                if Kx == K and slot is not None and slot.is_contiguous() and wgrad_queue.enabled:
                    dst = slot.view(weight.shape[0], K)[lo:hi]
                    bias_in_arena = db_view is None or getattr(bias, "_vmc_grad", None) is not None
                    if not (bias_in_arena and ops.wgrad_group_ok(dz, x, dst) and (db_view is None or db_view.data_ptr() % 16 == 0)):
                        dst = None
                if dst is not None:                    # grouped with the other weight gradients of this backward pass
                    wgrad_queue.push(dz, x, dst, db_view, [weight] + ([bias] if db_view is not None else []))
                    return dx, None, None, dres, None, None, None, None
                if Kx == K:
                    if rows is not None:
                        if slot is not None:
                            ops.wgrad_tn(dz, x, slot.view(weight.shape[0], K)[lo:hi], db_view)
                            dw = slot
                        else:
                            dw = torch.zeros((weight.shape[0], K), dtype=torch.float32, device=dz.device)
                            ops.wgrad_tn(dz, x, dw[lo:hi], db_view)
                    else:
                        dw = ops.wgrad_tn(dz, x, _grad_out(weight, (N, K)).view(N, K), db_view)
                else:                                                                   # K-padded x (patch GEMM)
                    full = ops.wgrad_tn(dz, x, torch.empty((N, Kx), dtype=torch.float32, device=dz.device), db_view)
                    dw = full[:, :K].contiguous()
            else:
                # odd widths: transposed operands (M zero-padded to 64) through the NT kernel
                Mp = _pad64(M)
                dzt = torch.zeros((N, Mp), dtype=dt16, device=dz.device) if Mp != M else torch.empty((N, Mp), dtype=dt16, device=dz.device)
                check(lib.vmc_transpose16(ptr(dz), ptr(dzt), M, N, dz.stride(0), Mp, stream()), "transpose16")
                xt = torch.zeros((Kx, Mp), dtype=dt16, device=dz.device) if Mp != M else torch.empty((Kx, Mp), dtype=dt16, device=dz.device)
                check(lib.vmc_transpose16(ptr(x), ptr(xt), M, Kx, x.stride(0), Mp, stream()), "transpose16")
                full = ops.linear_wgrad(dzt, xt, torch.empty((N, Kx), dtype=torch.float32, device=dz.device))
                dwp = full[:, :K] if Kx != K else full
                if rows is not None:
                    dw = slot if slot is not None else torch.zeros((weight.shape[0], K), dtype=torch.float32, device=dz.device)
                    dw.view(weight.shape[0], K)[lo:hi].copy_(dwp)
                else:
                    dw = dwp.contiguous()
        if want_db and not db_done:
            if N % 8:                                                                   # odd width: column sums of a padded copy
                dzc = torch.zeros((M, (N + 7) // 8 * 8), dtype=dt16, device=dz.device)
                dzc[:, :N].copy_(dz)
                db = ops.colsum(dzc)[:N].contiguous()
            else:
                db = ops.colsum(dz)
            if rows is not None:
                slot = getattr(bias, "_vmc_grad", None)
                if slot is not None:
                    slot[lo:hi].copy_(db)
                    db = slot
                else:
                    full = torch.zeros_like(bias)
                    full[lo:hi].copy_(db)
                    db = full
        return dx, _deliver(weight, dw), _deliver(bias, db), dres, None, None, None, None


def linear(x, weight, bias=None, res=None, act=ACT_NONE, out_f32=False, rows=None):
    # (grad mode is always off INSIDE Function.forward, so it is sampled here)
    return LinearFn.apply(x, weight, bias, res, act, out_f32, rows, not torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# LayerNorm (optionally forwarding its input so the residual fork is summed inside the backward kernel)
# ------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, dt16, passthrough, out_f32):
        D = gamma.shape[0]
        rows = x.numel() // D
        y16, y32, mean, rstd = ops.layernorm(x, gamma.detach(), beta.detach(), dt16, out16=not out_f32, out32=out_f32,
                                             rows=rows, save_stats=True)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta, ctx.dt16 = gamma, beta, dt16
        y = y32 if out_f32 else y16
        if passthrough:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta, dt16 = ctx.gamma, ctx.beta, ctx.dt16
        D = gamma.shape[0]
        rows = x.numel() // D
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        add = None
        if dpass is not None:
            add = dpass.contiguous()
            if add.dtype != x.dtype:
                add = ops.cast32(add) if x.dtype == torch.float32 else ops.cast16(add, x.dtype)
        dg = _grad_out(gamma, (D,))
        db = _grad_out(beta, (D,))
        nbytes = lib.vmc_layernorm_bwd_workspace_bytes(rows, D)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        check(lib.vmc_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma.detach()), ptr(mean), ptr(rstd), ptr(add), ptr(dx), ptr(dg), ptr(db),
                                    rows, D, D, dt(dy), dt(x), dt(dx), dt(dt16), ptr(ws), nbytes, stream()), "layernorm_bwd")
        return dx, _deliver(gamma, dg), _deliver(beta, db), None, None, None


class PostNormFn(torch.autograd.Function):
    """(y32, y16) = LN(x32 + drop2(drop1(branch16))): the tail of a post-norm block in one kernel (vmc_postnorm_dropout_fwd).
    ``drops`` = ((p1, seed1), (p2, seed2)) with p = 0 for "no dropout"."""

    @staticmethod
    def forward(ctx, x32, branch, gamma, beta, need, drops=((0.0, 0), (0.0, 0))):
        D = gamma.shape[0]
        rows = x32.numel() // D
        dev = x32.device
        y32 = torch.empty((rows, D), dtype=torch.float32, device=dev)
        y16 = torch.empty((rows, D), dtype=branch.dtype, device=dev)
        ssum = torch.empty((rows, D), dtype=torch.float32, device=dev) if need else None
        mean = torch.empty(rows, dtype=torch.float32, device=dev) if need else None
        rstd = torch.empty(rows, dtype=torch.float32, device=dev) if need else None
        (p1, s1), (p2, s2) = drops
        check(lib.vmc_postnorm_dropout_fwd(ptr(x32), ptr(branch), ptr(gamma.detach()), ptr(beta.detach()), ptr(ssum), ptr(y32), ptr(y16),
                                           ptr(mean), ptr(rstd), rows, D, 1e-5, float(p1), int(s1), float(p2), int(s2), dt(branch),
                                           stream()), "postnorm_dropout_fwd")
        ctx.save_for_backward(ssum, mean, rstd)
        ctx.gamma, ctx.beta, ctx.dt16, ctx.drops = gamma, beta, branch.dtype, drops
        return y32, y16

    @staticmethod
    def backward(ctx, dy32, dy16):
        ssum, mean, rstd = ctx.saved_tensors
        gamma, beta, dt16 = ctx.gamma, ctx.beta, ctx.dt16
        D = gamma.shape[0]
        rows = ssum.numel() // D
        dy2 = None
        if dy32 is None:
            dy = dy16.contiguous()
        elif dy16 is None:
            dy = dy32.contiguous()
        else:                                   # both consumers sent a gradient: summed in fp32 inside the LayerNorm backward
            dy, dy2 = dy32.contiguous(), dy16.contiguous()
            if dy2.dtype != dt16:
                dy, dy2 = _add(dy, dy2, torch.float32, dt16), None
        dsum = torch.empty_like(ssum)
        dg, db = _grad_out(gamma, (D,)), _grad_out(beta, (D,))
        nbytes = lib.vmc_layernorm_bwd_workspace_bytes(rows, D)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=ssum.device)
        (p1, s1), (p2, s2) = ctx.drops
        dbr = torch.empty(dsum.shape, dtype=dt16, device=dsum.device)
        if FUSE_POSTNORM_BWD:
            # d branch = d sum * mask1 (* mask2), masks regenerated from the seeds, cast to the branch's type -- from the LayerNorm
            # backward's own store loop (vmc_postnorm_bwd) instead of a second pass over d sum
            check(lib.vmc_postnorm_bwd(ptr(dy), ptr(dy2), ptr(ssum), ptr(gamma.detach()), ptr(mean), ptr(rstd), ptr(dsum), ptr(dbr), ptr(dg),
                                       ptr(db), rows, D, dt(dy), float(p1), int(s1), float(p2), int(s2), dt(dt16), ptr(ws), nbytes,
                                       stream()), "postnorm_bwd")
            return dsum, dbr, _deliver(gamma, dg), _deliver(beta, db), None, None
        check(lib.vmc_layernorm_bwd2(ptr(dy), ptr(dy2), ptr(ssum), ptr(gamma.detach()), ptr(mean), ptr(rstd), None, ptr(dsum), ptr(dg), ptr(db),
                                     rows, D, D, dt(dy), 0, 0, dt(dt16), ptr(ws), nbytes, stream()), "layernorm_bwd2")
        if p1 > 0.0:
            check(lib.vmc_cast_dropout2(ptr(dsum), ptr(dbr), dsum.numel(), float(p1), int(s1), float(p2), int(s2), dt(dt16), stream()),
                  "cast_dropout2")
        else:
            dbr = ops.cast16(dsum, dt16)
        return dsum, dbr, _deliver(gamma, dg), _deliver(beta, db), None, None


def postnorm(x32, branch, gamma, beta, drops=((0.0, 0), (0.0, 0))):
    return PostNormFn.apply(x32, branch, gamma, beta, torch.is_grad_enabled(), drops)


def layernorm(x, gamma, beta, dt16, passthrough=False, out_f32=False):
    return LayerNormFn.apply(x, gamma, beta, dt16, passthrough, out_f32)


# ------------------------------------------------------------------------------------------------
# Attention
# ------------------------------------------------------------------------------------------------
def _attn_bwd(q, k, v, mask, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, dh, p=0.0, seed=0):
    nbytes = lib.vmc_attention_bwd_workspace_bytes(B, H, Tq)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=q.device)
    check(lib.vmc_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(mask), ptr(out), ptr(dout), ptr(lse), ptr(dq), ptr(dk), ptr(dv),
                                B, H, Tq, Tk, dh, q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                dq.stride(0), dk.stride(0), dv.stride(0), float(p), int(seed), ptr(ws), nbytes, dt(q),
                                stream()), "attention_bwd")


class SelfAttnPackedFn(torch.autograd.Function):
    """qkv [B*T, 3D] packed in_proj output -> o [B*T, D].  mask u8 [B,T] (1 = real token) or None."""

    @staticmethod
    def forward(ctx, qkv, mask, B, T, H):
        D = qkv.shape[1] // 3
        dh = D // H
        if mask is None and dh == 64 and T <= 288:
            out, lse = ops.attention_vit(qkv, B, T, H, want_lse=True)
        else:
            out, lse = ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], mask, B, H, T, T, dh, want_lse=True)
        ctx.save_for_backward(qkv, out, lse, mask)
        ctx.dims = (B, T, H, D, dh)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, mask = ctx.saved_tensors
        B, T, H, D, dh = ctx.dims
        dout = ops.cast16(dout.contiguous(), qkv.dtype)
        dqkv = torch.empty_like(qkv)
        _attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], mask, out, dout, lse,
                  dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, T, T, dh)
        return dqkv, None, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """q [B*Tq, D], kv [B*Tk, 2D] (K | V) -> o [B*Tq, D]; mask u8 [B,Tk] over the keys."""

    @staticmethod
    def forward(ctx, q, kv, mask, B, Tq, Tk, H):
        D = q.shape[1]
        dh = D // H
        out, lse = ops.attention(q, kv[:, :D], kv[:, D:], mask, B, H, Tq, Tk, dh, want_lse=True)
        ctx.save_for_backward(q, kv, out, lse, mask)
        ctx.dims = (B, Tq, Tk, H, D, dh)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse, mask = ctx.saved_tensors
        B, Tq, Tk, H, D, dh = ctx.dims
        dout = ops.cast16(dout.contiguous(), q.dtype)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        _attn_bwd(q, kv[:, :D], kv[:, D:], mask, out, dout, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Tq, Tk, dh)
        return dq, dkv, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# small pieces
# ------------------------------------------------------------------------------------------------
class CastFn(torch.autograd.Function):
    """dtype change with the gradient cast back (f32 <-> 16-bit)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast32(x) if dtype == torch.float32 else ops.cast16(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (ops.cast32(dy) if ctx.src == torch.float32 else ops.cast16(dy, ctx.src)), None


def cast(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


def _add(a, b, out_dtype, dt16):
    y = torch.empty(a.shape, dtype=out_dtype, device=a.device)
    check(lib.vmc_add(ptr(a), ptr(b), ptr(y), a.numel(), dt(a), dt(b), dt(y), dt(dt16), stream()), "add")
    return y


class ForkFn(torch.autograd.Function):
    """x -> (x, x); the backward sums the two incoming gradients with vmc_add."""

    @staticmethod
    def forward(ctx, x, dt16):
        ctx.dt16, ctx.dtype = dt16, x.dtype
        return x.view(x.shape), x.view(x.shape)

    @staticmethod
    def backward(ctx, d1, d2):
        if d1 is None:
            return d2, None
        if d2 is None:
            return d1, None
        return _add(d1.contiguous(), d2.contiguous(), ctx.dtype, ctx.dt16), None


def fork(x, dt16):
    return ForkFn.apply(x, dt16)


class AddFn(torch.autograd.Function):
    """y = a + alpha * b (f32), ResidualMLP's skip (models/student_model.py:35)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = alpha
        y = torch.empty_like(a)
        check(lib.vmc_axpby_f32(ptr(a), ptr(b), ptr(y), a.numel(), 1.0, float(alpha), stream()), "axpby")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        db = torch.empty_like(dy)
        check(lib.vmc_axpby_f32(ptr(dy), ptr(dy), ptr(db), dy.numel(), float(ctx.alpha), 0.0, stream()), "axpby")
        return dy, db, None


class MeanPoolFn(torch.autograd.Function):
    """[B*T, D] -> [B, D] mean over T (all rows, padded ones included: TFAM/models/AMO_CLIP.py:170)."""

    @staticmethod
    def forward(ctx, x, B, T, dt16, out_f32):
        D = x.shape[-1]
        o16, o32 = ops.mean_pool(x, B, T, D, dt16, out16=not out_f32, out32=out_f32)
        ctx.meta = (B, T, D, x.dtype, dt16)
        return o32 if out_f32 else o16

    @staticmethod
    def backward(ctx, dout):
        B, T, D, xdtype, dt16 = ctx.meta
        dout = dout.contiguous()
        dx = torch.empty((B * T, D), dtype=xdtype, device=dout.device)
        check(lib.vmc_mean_pool_bwd(ptr(dout), ptr(dx), B, T, D, dt(dout), dt(dx), dt(dt16), stream()), "mean_pool_bwd")
        return dx, None, None, None, None


class AssembleTokensFn(torch.autograd.Function):
    """Patch rows + class token + positional embedding -> token matrix [F*N, D] (training path of K1)."""

    @staticmethod
    def forward(ctx, xp, cls, pos, F, N, out_dtype):
        D = xp.shape[1]
        x = torch.empty((F * N, D), dtype=out_dtype, device=xp.device)
        check(lib.vmc_assemble_tokens(ptr(xp), ptr(cls.detach()), ptr(pos.detach()), ptr(x), F, N, D, dt(x), dt(xp), stream()),
              "assemble_tokens")
        ctx.cls, ctx.pos, ctx.meta = cls, pos, (F, N, D, xp.dtype)
        return x

    @staticmethod
    def backward(ctx, dx):
        F, N, D, dt16 = ctx.meta
        dx = dx.contiguous()
        dpos = ops.colsum(dx.view(F, N * D)).view(N, D)                 # sum over frames
        dcls = dpos[0].contiguous()
        dxp = ops.cast16(dx.view(F, N, D)[:, 1:].contiguous().view(F * (N - 1), D), dt16)
        return dxp, _deliver(ctx.cls, dcls), _deliver(ctx.pos, dpos), None, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        y = torch.empty_like(x)
        dt16 = x.dtype if x.dtype != torch.float32 else torch.bfloat16
        check(lib.vmc_dropout(ptr(x), ptr(y), x.numel(), float(p), int(seed), dt(x), dt(dt16), stream()), "dropout")
        ctx.meta = (p, seed, dt16)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, dt16 = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.vmc_dropout(ptr(dy), ptr(dx), dy.numel(), float(p), int(seed), dt(dy), dt(dt16), stream()), "dropout")
        return dx, None, None


def dropout(x, p, training, seed_fn):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x.contiguous(), p, seed_fn())


def scale_by_device_scalar(x: torch.Tensor, scalar: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    s = scalar.reshape(1).float().contiguous()
    check(lib.vmc_scale_by_device_scalar(ptr(x), ptr(y), x.numel(), ptr(s), stream()), "scale_by_device_scalar")
    return y
