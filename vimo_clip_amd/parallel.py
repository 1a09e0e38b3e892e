"""Clip-level data parallelism: one process per GPU, gradients all-reduced over RCCL/xGMI.

Replaces the reference's single-process ``torch.nn.DataParallel`` (train.py:64, TFAM/train_and_eval.py:392:
per-step parameter broadcast + input scatter + output gather + reduce_add of gradients onto GPU 0) by the
one exchange the algorithm needs: a SUM all-reduce of the flat gradient arena per step, in ~48 MB buckets
issued asynchronously (each bucket is a contiguous slice of GradArena.flat_grad; NCCL == RCCL on ROCm), the
mean folded into the optimiser's ``grad_scale``.  Parameters that get no gradient in the active fusion mode
are not in the arena, so nothing is reduced for them (SURVEY.md §7 "DDP with unused parameters").

Works with the ``gloo`` backend on CPU tensors too (tests/test_parallel_cpu.py, world_size 2).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(n_items: int, rank: int, world: int, drop_last: bool = True):
    """Contiguous shard of ``n_items`` clips for ``rank`` (DistributedSampler-style, drop_last mirrors the
    reference loaders, TFAM/train_and_eval.py:374,398)."""
    if drop_last:
        per = n_items // world
        return rank * per, (rank + 1) * per
    per = (n_items + world - 1) // world
    return min(n_items, rank * per), min(n_items, (rank + 1) * per)


class GradientAllReducer:
    """Bucketed asynchronous SUM all-reduce of a flat gradient buffer.

    Plain use: ``all_reduce()`` after the backward issues every bucket and waits.  With ``attach(arena)`` the buckets are
    issued *during* the backward: backward kernels write gradients straight into the arena and report each
    parameter (``autograd_ops.grad_ready_hooks``); a bucket goes out on the RCCL stream the moment the last parameter that
    overlaps it is complete, i.e. in reverse layer order while earlier layers are still back-propagating.  How many reports
    a parameter gets per backward (a packed ``in_proj`` is written in two row slices) is learned from the first step, which
    runs un-overlapped; parameters that never report (unused in the fusion mode) are flushed by ``all_reduce()``.  Every rank
    runs the same graph, so the launch order of the collectives is the same on every rank."""

    EXCHANGES = ("all_reduce", "rs_ag")

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 48 << 20, exchange: str | None = None):
        """exchange: "all_reduce" (one collective per bucket; RCCL picks ring or tree) or "rs_ag" (reduce_scatter_tensor into a
        1/world shard, then all_gather_into_tensor back into the bucket: on the fully connected 8-GPU xGMI mesh both halves are
        direct exchanges that keep all seven links of a GPU busy, where a ring is bound by one link -- SURVEY.md 8e).  Default:
        the VMC_GRAD_EXCHANGE environment variable, else "all_reduce".  The sums are the same numbers in both; their
        association order may differ for world > 2."""
        exchange = exchange or os.environ.get("VMC_GRAD_EXCHANGE", "all_reduce")
        if exchange not in self.EXCHANGES:
            raise ValueError(f"GradientAllReducer: exchange must be one of {self.EXCHANGES}, got {exchange!r}")
        self.exchange = exchange
        self._shards = {}
        self.flat = flat_grad
        n = max(64, bucket_bytes // flat_grad.element_size())
        self.bucket_elems = n
        self.buckets = [flat_grad[s:min(flat_grad.numel(), s + n)] for s in range(0, flat_grad.numel(), n)]
        self._pending = []
        self._attached = False

    # ---- overlap with the backward ------------------------------------------------------------------
    def attach(self, arena, register: bool = True):
        """arena: optim.GradArena (``params`` + ``offsets`` into the flat gradient buffer)."""
        import weakref
        self._pbuckets, self._prefs = {}, {}
        for p, o in zip(arena.params, arena.offsets):
            b0, b1 = o // self.bucket_elems, (o + max(1, p.numel()) - 1) // self.bucket_elems
            self._pbuckets[id(p)] = list(range(b0, b1 + 1))
            self._prefs[id(p)] = weakref.ref(p)     # id() values are recycled once a parameter dies
        self._expected = None                      # id(param) -> reports per backward, learned in the first step
        self._counts = {}
        self._launched = [False] * len(self.buckets)
        self._remaining = None
        self._attached = True
        if register:
            from . import autograd_ops
            autograd_ops.grad_ready_hooks.append(self.on_grad_ready)
        return self

    def detach(self):
        from . import autograd_ops
        if self.on_grad_ready in autograd_ops.grad_ready_hooks:
            autograd_ops.grad_ready_hooks.remove(self.on_grad_ready)
        self._attached = False

    def _issue(self, b):
        """Start the exchange of bucket b; returns what finish() must complete."""
        buf, world = self.buckets[b], world_size()
        if self.exchange == "all_reduce" or buf.numel() % world:      # a ragged bucket cannot be cut into equal shards
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        shard = self._shards.get(b)
        if shard is None or shard.numel() != buf.numel() // world:
            shard = self._shards[b] = torch.empty(buf.numel() // world, dtype=buf.dtype, device=buf.device)
        rs = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, async_op=True)
        if dist.get_backend() == "nccl":            # RCCL runs a group's collectives in issue order on its own stream
            return dist.all_gather_into_tensor(buf, shard, async_op=True)
        return (rs, buf, shard)                     # gloo's worker threads give no such order: the gather is issued in finish()

    def _launch(self, b):
        self._launched[b] = True
        self._pending.append(self._issue(b))

    def on_grad_ready(self, param):
        k = id(param)
        if not self._attached or world_size() == 1 or k not in self._pbuckets or self._prefs[k]() is not param:
            return
        self._counts[k] = self._counts.get(k, 0) + 1
        if self._expected is None or self._counts[k] != self._expected.get(k, 0):
            return                                  # calibration step, or more slices of this parameter to come
        for b in self._pbuckets[k]:
            self._remaining[b] -= 1
            if self._remaining[b] == 0 and not self._launched[b]:
                self._launch(b)

    def _arm(self):
        self._counts = {}
        self._launched = [False] * len(self.buckets)
        self._remaining = [0] * len(self.buckets)
        for k, n in self._expected.items():
            if n > 0:
                for b in self._pbuckets[k]:
                    self._remaining[b] += 1

    # ---- after the backward ---------------------------------------------------------------------------
    def start(self):
        if world_size() == 1:
            return
        if not self._attached:
            self._pending = [self._issue(b) for b in range(len(self.buckets))]
            return
        self._n_early = sum(self._launched)
        if self._expected is None:                  # first step: learn the report counts, reduce everything now
            self._expected = dict(self._counts)
            self._launched = [False] * len(self.buckets)
        for b in range(len(self.buckets)):          # whatever the backward did not complete (unused parameters, padding)
            if not self._launched[b]:
                self._launch(b)

    def finish(self) -> float:
        """Wait for the buckets; returns the factor the optimiser must scale gradients by (1/world)."""
        gathers = []
        for w in self._pending:
            if isinstance(w, tuple):                # (reduce-scatter work, bucket, shard): second half of an rs_ag exchange
                w[0].wait()
                gathers.append(dist.all_gather_into_tensor(w[1], w[2], async_op=True))
            else:
                w.wait()
        for w in gathers:
            w.wait()
        self._pending = []
        if self._attached and self._expected is not None:
            self._arm()
        return 1.0 / world_size()

    def all_reduce(self) -> float:
        self.start()
        return self.finish()

    @property
    def overlapped_last_step(self) -> int:
        """Number of buckets the most recent backward issued before ``all_reduce()`` was called (diagnostics / tests)."""
        return getattr(self, "_n_early", 0)


def broadcast_parameters(flat_param: torch.Tensor, src: int = 0):
    """Make every replica start from rank ``src``'s weights (one broadcast of the flat arena)."""
    if world_size() > 1:
        dist.broadcast(flat_param, src=src)


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """Concatenate per-rank [n_local, C] score/target matrices (exact micro-AP needs all of them: a mean of
    per-rank APs is not the global AP)."""
    if world_size() == 1:
        return x
    counts = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(world_size())]
    dist.all_gather(counts, torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device))
    nmax = int(max(c.item() for c in counts))
    pad = torch.zeros((nmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    outs = [torch.empty_like(pad) for _ in range(world_size())]
    dist.all_gather(outs, pad)
    return torch.cat([o[: int(c.item())] for o, c in zip(outs, counts)], dim=0)


def all_reduce_scalars(values: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values
