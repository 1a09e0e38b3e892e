"""hipGraph capture of launch-bound steps.

The libvmc entry points only enqueue kernels on the stream they are given (no allocation, no synchronisation;
include/vmc.h), so a whole TFAM forward — or a forward + backward + fused AdamW step — over fixed shapes can be
captured once and replayed as one graph launch.  At the reference's TFAM batch size (8 clips of 16x768 tokens) a
forward is ~90 kernels of a few microseconds each: replay removes the per-launch host cost that otherwise dominates.
"""
from __future__ import annotations

import torch


class GraphedCallable:
    """Captures ``fn(*static_inputs)`` once; ``__call__(*inputs)`` copies the inputs into the static buffers, replays
    the graph and returns the static outputs (valid until the next call)."""

    def __init__(self, fn, *example_inputs, warmup: int = 3):
        self.static_inputs = [x.clone() if torch.is_tensor(x) else x for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up outside capture: lazy kernel attributes, caches, allocator
            for _ in range(warmup):
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_outputs = fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if torch.is_tensor(dst) and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_outputs

    def replay(self):
        self.graph.replay()
        return self.static_outputs


class GraphedTrainStep:
    """One training step (vmc_train_tick + forward + backward + fused Adam) per batch shape as a hipGraph replay.

    The optimiser must be in device-state mode (``FusedAdam.enable_device_state``): the step count, the bias corrections, the
    learning rate and the dropout seeds then live in device memory, so every captured launch has constant arguments and a
    replay IS the next step (host scalars would be frozen into the graph).  The warm-up run that precedes a
    capture is undone (parameters, Adam moments and the step state are restored), so capturing a new batch shape does not
    train on its example batch.

    Data parallel (``exchange`` given; replaces nn.DataParallel's per-step reduce_add, TFAM/train_and_eval.py:392): the step is
    TWO graphs with the gradient exchange between them -- ``step_fn`` = tick + forward + loss + backward (one graph per batch
    shape), then ``exchange()`` (the bucketed RCCL all-reduce or reduce-scatter + all-gather of parallel.GradientAllReducer, eager:
    collectives are not captured) whose return value (1 / world) goes to the optimiser's device-resident grad_scale, then
    ``opt_fn`` = the fused AdamW + the refresh of the 16-bit copies (one graph, shape independent).  Gradient-ready hooks are
    silenced while capturing (a replay runs no Python, so the buckets go out after the backward graph, not during it)."""

    def __init__(self, step_fn, optimizer, max_graphs: int = 16, exchange=None, opt_fn=None, graph_factory=None):
        if getattr(optimizer, "dev_state", None) is None:
            raise ValueError("GraphedTrainStep needs FusedAdam.enable_device_state()")
        if (exchange is None) != (opt_fn is None):
            raise ValueError("GraphedTrainStep: exchange and opt_fn come together (two-graph data-parallel step)")
        self.step_fn, self.opt, self.max_graphs = step_fn, optimizer, max_graphs
        self.exchange, self.opt_fn = exchange, opt_fn
        self._factory = graph_factory or GraphedCallable
        self._graphs = {}
        self._opt_graph = None

    def _live(self):
        o, a = self.opt, self.opt.arena
        return (a.flat_param, a.flat_grad, o.m, o.v, o.dev_state, o.dev_hyper)

    def _capture(self, fn, inputs):
        from . import autograd_ops
        o = self.opt
        live = self._live()
        saved = [t.clone() for t in live]
        count = o.step_count
        hooks = list(autograd_ops.grad_ready_hooks)
        autograd_ops.grad_ready_hooks[:] = []          # no collective inside a warm-up or a capture
        try:
            g = self._factory(fn, *inputs, warmup=1)
        finally:
            autograd_ops.grad_ready_hooks[:] = hooks
        for t, s in zip(live, saved):
            t.copy_(s)
        o.step_count = count
        autograd_ops.weights.refresh()          # the 16-bit compute copies follow the restored masters
        return g

    def __call__(self, *inputs):
        from . import autograd_ops
        key = tuple((tuple(x.shape), x.dtype) if torch.is_tensor(x) else x for x in inputs)
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= self.max_graphs:           # too many shapes: eager step (same device-state arithmetic)
                out = self.step_fn(*inputs)
                if self.exchange is not None:
                    self.opt.sync_hyper(grad_scale=self.exchange())
                    self.opt_fn()
                return out
            g = self._graphs[key] = self._capture(self.step_fn, inputs)
        if self.exchange is not None and self._opt_graph is None:
            self._opt_graph = self._capture(self.opt_fn, ())
        out = g(*inputs)
        if self.exchange is not None:
            self.opt.sync_hyper(grad_scale=self.exchange())    # host -> device only when the factor changes (1 / world: once)
            self._opt_graph()
        self.opt.step_count += 1                               # host mirror of the device step count
        # host mirror of what the replayed Adam + vmc_cast_weights_multi launches did: the masters and their 16-bit copies
        # changed without any Python running, so everything keyed on the weight epoch (TfamPack.pack_is_current, captured
        # evaluation forwards) must see a new epoch (the fused eval path scored stale packs after pure replays)
        autograd_ops.weights.epoch += 1
        return out
