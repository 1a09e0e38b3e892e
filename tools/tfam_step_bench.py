#!/usr/bin/env python3
"""TFAM train step at batch B (configs[3] geometry, dropout 0.1 / 0.1): eager and as ONE hipGraph replay, with and without the
backward-overlapped AdamW.    python tools/tfam_step_bench.py [B] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.graphs import GraphedTrainStep  # noqa: E402
from vimo_clip_amd.losses import bce_with_logits_loss, loss_and_grad  # noqa: E402
from vimo_clip_amd.optim import FusedAdam, GradArena  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for overlap in (False,):
    for fused in (True, False):
        m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.1, mlp_dropout=0.1, device=dev).to(dev).train()
        m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
        m.fused_training = fused
        opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-4, weight_decay=0.1, decoupled=True)
        if overlap:
            opt.enable_backward_overlap(m.parameter_groups_by_layer())
            m.grad_group_callback = opt.group_ready
        rgb, mot = synth.normal(30, "rgb", (B, 16, 768)).to(dev), synth.normal(30, "mot", (B, 16, 768)).to(dev)
        mk = torch.ones(B, 16, dtype=torch.bool, device=dev)
        y = synth.multi_hot_labels(30, "lab", B, 140).to(dev)
        opt.enable_device_state(base_seed=0)
        m.use_device_seeds(opt)

        def dev_step(rgb, mot, mk, y):
            opt.tick()
            out = m(rgb, mot, mask_rgb=mk, mask_flow=mk)
            loss, dl = loss_and_grad(bce_with_logits_loss, out, y)
            out.backward(dl)
            opt.step()
            return loss, out.detach()

        te = timeit(lambda: dev_step(rgb, mot, mk, y), max(20, iters // 4))
        g = GraphedTrainStep(dev_step, opt)
        tg = timeit(lambda: g(rgb, mot, mk, y), iters)
        gr = next(iter(g._graphs.values()))
        tr = timeit(lambda: gr.replay(), iters)
        print(f"B={B} fused={fused} overlap={overlap}: eager {te*1e6:8.1f} us   captured {tg*1e6:8.1f} us   bare replay {tr*1e6:8.1f} us", flush=True)
