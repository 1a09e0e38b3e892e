#!/usr/bin/env python3
"""TFAM forward micro-benchmark: python tools/tfam_bench.py [B] [train]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.losses import bce_with_logits_loss  # noqa: E402
from vimo_clip_amd.optim import FusedAdam, GradArena  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
train = len(sys.argv) > 2
dev = "cuda"
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0, device=dev).to(dev)
m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
rgb = synth.normal(10, "rgb", (B, 16, 768)).to(dev)
mot = synth.normal(10, "mot", (B, 16, 768)).to(dev)
mk = torch.ones(B, 16, dtype=torch.bool, device=dev)
y = synth.multi_hot_labels(20, "lab", B, 140).to(dev)
if train:
    m.train()
    opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-4, weight_decay=0.1, decoupled=True)

    def f():
        bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mk), y).backward()
        opt.step()
else:
    m.eval()

    def f():
        with torch.no_grad():
            m(rgb, mot, mask_rgb=mk, mask_flow=mk)
for _ in range(3):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"TFAM {'train' if train else 'fwd'} B={B}: {dt*1e3:.3f} ms, {B/dt:.0f} clips/s")
