#!/usr/bin/env python3
"""Host-side (Python) cost of a small-batch TFAM train step: cProfile over 100 eager steps at B = 8."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.losses import bce_with_logits_loss  # noqa: E402
from vimo_clip_amd.optim import FusedAdam, GradArena  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.1, mlp_dropout=0.1, device=dev).to(dev).train()
opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-4, weight_decay=0.1, decoupled=True)
rgb = synth.normal(10, "rgb", (B, 16, 768)).to(dev)
mot = synth.normal(10, "mot", (B, 15, 768)).to(dev)
mk, mf = torch.ones(B, 16, dtype=torch.bool, device=dev), torch.ones(B, 15, dtype=torch.bool, device=dev)
y = synth.multi_hot_labels(20, "lab", B, 140).to(dev)


def step():
    bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mf), y).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step eager")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
