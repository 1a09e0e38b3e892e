#!/usr/bin/env python3
"""Micro-benchmark of vmc_attention_vit_fwd (ViT-L/14 shape by default): python tools/attn_bench.py [F N H]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import ops  # noqa: E402

F, N, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 257, 16)
D = H * 64
qkv = torch.randn(F * N, 3 * D, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attention_vit(qkv, F, N, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention_vit(qkv, F, N, H)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"attention F={F} N={N} H={H}: {ms*1e3:.1f} us, {4.0*N*N*64*F*H/ms/1e9:.1f} TFLOP/s (algorithmic 4*N^2*dh)")
