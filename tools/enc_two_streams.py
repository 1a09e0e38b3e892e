#!/usr/bin/env python3
"""Encoder step (ViT-L/14, 256 frames) on one stream vs two half-batches on two streams (VMC_GEMM_CUS sets the GEMM walk's workgroups)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth
from vimo_clip_amd.clip_vit import VisionTransformer
m = VisionTransformer.from_name("ViT-L/14", compute_dtype=torch.bfloat16).to("cuda").eval()
m.load_state_dict(synth.vit_state_dict("ViT-L/14", 1), strict=True)
u8 = synth.randint_u8(1, "frames", (256, 3, 224, 224)).cuda()
def timeit(n=6):
    for _ in range(2): y = m.encode_frames_u8(u8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = m.encode_frames_u8(u8)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, y
m.slice_streams = 1
t1, y1 = timeit()
print(f"one stream {t1*1e3:.2f} ms ({256/t1:.0f} frames/s)", flush=True)
for n in (2, 3, 4, 2, 1):
    m.slice_streams = n
    t2, y2 = timeit()
    print(f"{n} streams {t2*1e3:.2f} ms ({256/t2:.0f} frames/s)   identical {torch.equal(y1, y2)}", flush=True)
