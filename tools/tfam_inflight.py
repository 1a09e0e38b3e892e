#!/usr/bin/env python3
"""Fused TFAM eval chain: G independent batches of B clips in flight on G streams (one hipGraph + one scratch slot each).
    python tools/tfam_inflight.py [B] [G ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.graphs import GraphedCallable  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Gs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
dev = torch.device("cuda", 0)
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0, device=dev).to(dev).eval()
m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
rgb, mot = synth.normal(10, "rgb", (B, 16, 768)).to(dev), synth.normal(10, "mot", (B, 16, 768)).to(dev)
mk = torch.ones(B, 16, dtype=torch.bool, device=dev)


def fwd(r, f, a, b):
    with torch.no_grad():
        return m(r, f, mask_rgb=a, mask_flow=b)


for G in Gs:
    streams = [torch.cuda.Stream() for _ in range(G)]
    graphs = []
    for i, st in enumerate(streams):
        m.fused_slot = i
        with torch.cuda.stream(st):
            graphs.append(GraphedCallable(fwd, rgb, mot, mk, mk))
    m.fused_slot = 0
    torch.cuda.synchronize()
    reps = max(20, 400 // G)
    for _ in range(3):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / (reps * G)
    print(f"B={B} G={G}: {t*1e6:8.1f} us per forward, {B/t:10.0f} clips/s", flush=True)
    del graphs
