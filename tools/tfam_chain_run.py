#!/usr/bin/env python3
"""Runs the fused TFAM eval forward N times (for rocprofv3 --kernel-trace): python3 tools/tfam_chain_run.py [B] [iters] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cdt = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "f16") else torch.bfloat16
dev = "cuda"
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0, device=dev,
             compute_dtype=cdt).to(dev).eval()
m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
rgb = synth.normal(10, "rgb", (B, 16, 768)).to(dev)
mot = synth.normal(10, "mot", (B, 16, 768)).to(dev)
mk = torch.ones(B, 16, dtype=torch.bool, device=dev)
with torch.no_grad():
    for _ in range(iters):
        y = m(rgb, mot, mask_rgb=mk, mask_flow=mk)
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
