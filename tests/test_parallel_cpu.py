"""CPU, world_size 2, gloo: the data-parallel pieces (gradient bucket all-reduce over the flat arena,
clip sharding, score all-gather for the exact micro-AP)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from vimo_clip_amd import parallel
    from vimo_clip_amd.metrics import micro_average_precision
    parallel.init_from_env("gloo")
    try:
        n = 100_003
        flat = torch.full((n,), float(rank + 1))
        flat[rank::7] += 0.5
        red = parallel.GradientAllReducer(flat, bucket_bytes=64 * 1024)
        assert len(red.buckets) > 5
        scale = red.all_reduce()
        want = torch.full((n,), 3.0)
        want[0::7] += 0.5
        want[1::7] += 0.5
        assert scale == 0.5 and torch.equal(flat, want)
        # ---- the two exchange algorithms: reduce-scatter + all-gather == all-reduce, bit for bit ----
        gen = torch.Generator().manual_seed(1234 + rank)
        base = torch.randn(70_000, generator=gen)               # 70 000 = 4 buckets of 16 384 + a ragged bucket of 4 464 (even: shards)
        odd = torch.randn(16_385, generator=gen)                # last bucket of 1 element: not divisible by world -> all-reduce fallback
        for src in (base, odd):
            outs = {}
            for ex in parallel.GradientAllReducer.EXCHANGES:
                buf = src.clone()
                r = parallel.GradientAllReducer(buf, bucket_bytes=64 * 1024, exchange=ex)
                assert r.exchange == ex and r.all_reduce() == 0.5
                outs[ex] = buf
            assert torch.equal(outs["all_reduce"], outs["rs_ag"])
            both = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(both, src)
            assert torch.equal(outs["rs_ag"], both[0] + both[1])
        try:
            parallel.GradientAllReducer(base.clone(), exchange="ring")
            raise AssertionError("bad exchange name accepted")
        except ValueError:
            pass
        # ---- bucket launches overlapped with the backward (attach): reports arrive in reverse parameter order ----
        class _P:                                            # stand-in for a parameter: only numel() and identity matter
            def __init__(self, n):
                self.n = n

            def numel(self):
                return self.n

        class _Arena:
            pass
        sizes = [300, 5000, 64, 9000, 128, 4096, 700]        # 64-aligned offsets; several parameters straddle 4096-element buckets
        ar = _Arena()
        ar.params = [_P(nn) for nn in sizes]
        ar.offsets, off = [], 0
        for nn in sizes:
            ar.offsets.append(off)
            off += (nn + 63) // 64 * 64
        g = torch.zeros(off)
        red2 = parallel.GradientAllReducer(g, bucket_bytes=4096 * 4).attach(ar, register=False)
        unused = {2}                                         # a parameter the fusion mode never touches
        twice = {3}                                          # a packed in_proj: two row-slice reports per backward
        for step in range(3):
            g.fill_(float(rank + 1) * (step + 1))
            for i in reversed(range(len(sizes))):
                if i in unused:
                    continue
                for _ in range(2 if i in twice else 1):
                    red2.on_grad_ready(ar.params[i])
            scale2 = red2.all_reduce()
            assert scale2 == 0.5 and torch.all(g == 3.0 * (step + 1)), (step, g.unique())
            early = red2.overlapped_last_step
            assert (early == 0) if step == 0 else (early >= len(red2.buckets) - 2), (step, early, len(red2.buckets))
        p = torch.full((10,), float(rank))
        parallel.broadcast_parameters(p, src=0)
        assert torch.all(p == 0)
        lo, hi = parallel.shard_range(11, rank, world)
        rows = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1).repeat(1, 3)
        if rank == 1:
            rows = rows[:-1]                                  # ragged per-rank row counts
        allr = parallel.all_gather_rows(rows)
        g = torch.Generator().manual_seed(0)
        s_all, y_all = torch.rand(40, 5, generator=g), (torch.rand(40, 5, generator=g) > 0.6).long()
        s_loc, y_loc = s_all[rank * 20:(rank + 1) * 20], y_all[rank * 20:(rank + 1) * 20]
        ap = micro_average_precision(parallel.all_gather_rows(s_loc), parallel.all_gather_rows(y_loc))
        q.put((rank, allr[:, 0].tolist(), float(ap), float(micro_average_precision(s_all, y_all))))
    finally:
        dist.destroy_process_group()


class _EagerGraph:
    """Stand-in for graphs.GraphedCallable on a machine without a GPU: same interface (warm-up run, then call = run)."""

    def __init__(self, fn, *example_inputs, warmup=1):
        self.fn = fn
        for _ in range(warmup):
            fn(*example_inputs)

    def __call__(self, *inputs):
        return self.fn(*inputs)


def _two_graph_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from vimo_clip_amd import autograd_ops, parallel
    from vimo_clip_amd.graphs import GraphedTrainStep
    parallel.init_from_env("gloo")
    try:
        class _Arena:
            pass

        class _Opt:           # the attribute surface GraphedTrainStep and the trainer use of optim.FusedAdam (device-state mode)
            def __init__(self, n):
                self.arena = _Arena()
                g = torch.Generator().manual_seed(7)
                self.arena.flat_param = torch.randn(n, generator=g)
                self.arena.flat_grad = torch.zeros(n)
                self.m, self.v = torch.zeros(n), torch.zeros(n)
                self.dev_state = torch.zeros(4, dtype=torch.int64)
                self.dev_hyper = torch.tensor([0.1, 0.0, 0.0, 1.0])
                self.step_count = 0

            def sync_hyper(self, grad_scale=1.0):
                self.dev_hyper[3] = grad_scale

            def tick(self):
                self.dev_state[0] += 1

            def step(self):   # momentum SGD on the exchanged gradient, scaled by the device-resident factor
                self.m.mul_(0.9).add_(self.arena.flat_grad * self.dev_hyper[3])
                self.arena.flat_param.sub_(self.dev_hyper[0] * self.m)

        def run(two_graph, exchange):
            opt = _Opt(50_000)
            red = parallel.GradientAllReducer(opt.arena.flat_grad, bucket_bytes=64 * 1024, exchange=exchange)
            fired = []
            hook = lambda p: fired.append(p)                      # noqa: E731
            autograd_ops.grad_ready_hooks.append(hook)

            def fwd_bwd(x):
                opt.tick()
                opt.arena.flat_grad.copy_(torch.sin(opt.arena.flat_param * (rank + 1)) * x.sum())     # rank-dependent "gradient"
                for h in list(autograd_ops.grad_ready_hooks):
                    h("p")
                return opt.arena.flat_grad[:4].clone()
            try:
                stepper = GraphedTrainStep(fwd_bwd, opt, exchange=red.all_reduce, opt_fn=opt.step, graph_factory=_EagerGraph) if two_graph else None
                for i in range(4):
                    x = torch.full((3 if i % 2 else 5,), float(i + 1))     # two batch shapes -> two forward/backward "graphs"
                    if two_graph:
                        stepper(x)
                    else:
                        fwd_bwd(x)
                        opt.sync_hyper(grad_scale=red.all_reduce())
                        opt.step()
                        opt.step_count += 1
            finally:
                autograd_ops.grad_ready_hooks.remove(hook)
            return opt, len(fired), stepper
        ref, fired_ref, _ = run(False, "all_reduce")
        assert fired_ref == 4
        for ex in parallel.GradientAllReducer.EXCHANGES:
            o, fired, st = run(True, ex)
            assert len(st._graphs) == 2 and st._opt_graph is not None
            assert fired == 4, fired                               # silenced during the 3 warm-up / capture runs, live in the 4 steps
            assert o.step_count == ref.step_count == 4 and int(o.dev_state[0]) == 4          # capture runs were undone
            assert torch.equal(o.arena.flat_param, ref.arena.flat_param) and torch.equal(o.m, ref.m), ex
        both = [torch.empty_like(ref.arena.flat_param) for _ in range(world)]
        dist.all_gather(both, ref.arena.flat_param)
        assert torch.equal(both[0], both[1])                       # replicas stay identical
        q.put((rank, True))
    finally:
        dist.destroy_process_group()


def test_two_graph_data_parallel_step_equals_the_plain_step_gloo_world2():
    """GraphedTrainStep(exchange=..., opt_fn=...) -- forward/backward graph, gradient exchange, optimiser graph --
    must give the parameters of the plain step (forward/backward, exchange, optimiser) bit for bit, with both exchange algorithms,
    over several batch shapes; the warm-up / capture runs must leave no trace (graph stand-in: eager callables, no GPU here)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rows, ap, ap_ref in res:
        assert rows == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]      # 5 + 4 rows, rank order, drop_last shards of 11
        assert abs(ap - ap_ref) < 1e-7


def test_shard_range():
    from vimo_clip_amd.parallel import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert [shard_range(10, r, 4, drop_last=False) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]


def test_grad_clip_uses_the_norm_of_the_averaged_gradient():
    """clipping after a SUM all-reduce must see ||sum|| / world, or N GPUs clip N times too hard."""
    import math

    from vimo_clip_amd.optim import clipped_grad_scale
    g = torch.tensor([3.0, 4.0])                       # per-rank gradient, identical on both ranks: average norm 5
    for world in (1, 2, 8):
        s = clipped_grad_scale(float((g * world).norm()), 1.0 / world, max_grad_norm=1.0)
        eff = g * world * s                             # what the optimiser applies
        ref = g.clone()
        torch.nn.utils.clip_grad_norm_([torch.nn.Parameter(ref)], 1.0)      # torch on the averaged gradient
        p = torch.nn.Parameter(torch.zeros(2))
        p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p], 1.0)
        assert torch.allclose(eff, p.grad, rtol=1e-6), (world, eff, p.grad)
        assert math.isclose(float(eff.norm()), 1.0, rel_tol=1e-5)
    assert clipped_grad_scale(0.5, 0.25, max_grad_norm=10.0) == 0.25   # below the threshold: only the averaging factor
