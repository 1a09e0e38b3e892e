#!/usr/bin/env python3
"""Student train-step micro-benchmark (BASELINE.json configs[2]): python tools/student_bench.py [model] [B] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth  # noqa: E402
from vimo_clip_amd.losses import classification_loss, distillation_loss  # noqa: E402
from vimo_clip_amd.models import FlowStudentModel  # noqa: E402
from vimo_clip_amd.optim import FusedAdam, GradArena  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
E = synth.VIT_GEOMETRY[name][5]
m = FlowStudentModel(name, device="cuda", num_classes=140).train()
m.load_state_dict(synth.student_state_dict(name, 3, zero_fc2=True), strict=True)
arena = GradArena(m.parameters())
opt = FusedAdam(arena, lr=1e-3)
vids = synth.randint_u8(3, "vids", (B, T, 3, 224, 224)).cuda()
teacher = synth.normal(3, "teacher", (B, T + 1, E)).cuda()
labels = synth.multi_hot_labels(3, "labels", B, 140).cuda()


def step():
    emb, emb_d, logits = m(vids)
    loss = distillation_loss(emb_d, teacher[:, :-1, :], mode="cosine") + classification_loss(logits, labels, positive_weight=9)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"student {name} train step B={B} T={T}: {dt*1e3:.1f} ms, {B*T/dt:.0f} frames/s, loss {loss.item():.4f}")
