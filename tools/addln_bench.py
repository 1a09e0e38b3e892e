#!/usr/bin/env python3
"""Micro-benchmark of vmc_add_layernorm_fwd on the encoder's residual stream (ViT-L/14, 256 frames by default):
python tools/addln_bench.py [rows D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import ops  # noqa: E402

rows, D = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (65792, 1024)
x = torch.randn(rows, D, device="cuda")
br = (torch.randn(rows, D, device="cuda") * 0.01).bfloat16()
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
for _ in range(3):
    ops.add_layernorm_(x, br, g, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.add_layernorm_(x, br, g, b)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"add+LayerNorm rows={rows} D={D}: {us:.1f} us, {12.0 * rows * D / us / 1e3:.0f} GB/s (algorithmic 12 B per element)")
