#!/usr/bin/env python3
"""N eager TFAM training steps at batch B (configs[3] geometry) -- run under rocprofv3 --kernel-trace --stats to count the launches
of a step by kernel.    python tools/tfam_train_census.py [B] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.losses import bce_with_logits_loss  # noqa: E402
from vimo_clip_amd.optim import FusedAdam, GradArena  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.1, mlp_dropout=0.1, device=dev).to(dev).train()
m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-4, weight_decay=0.1, decoupled=True).enable_device_state(base_seed=11)
m.use_device_seeds(opt)      # the launches of the captured trainer's step (device step state, fused AdamW + copy refresh), issued eagerly
rgb, mot = synth.normal(30, "rgb", (B, 16, 768)).to(dev), synth.normal(30, "mot", (B, 16, 768)).to(dev)
mk = torch.ones(B, 16, dtype=torch.bool, device=dev)
y = synth.multi_hot_labels(30, "lab", B, 140).to(dev)
for _ in range(steps):
    opt.tick()
    loss = bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mk), y)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("steps", steps, "loss", float(loss))
