import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimo_clip_amd import synth
from vimo_clip_amd.graphs import GraphedCallable
from vimo_clip_amd.losses import bce_with_logits_loss
from vimo_clip_amd.optim import FusedAdam, GradArena
from vimo_clip_amd.TFAM.models import AMO_CLIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0, device=dev).to(dev).train()
m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
rgb = synth.normal(10, "rgb", (B, 16, 768)).to(dev); mot = synth.normal(10, "mot", (B, 16, 768)).to(dev)
mk = torch.ones(B, 16, dtype=torch.bool, device=dev); y = synth.multi_hot_labels(20, "lab", B, 140).to(dev)
opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-4, weight_decay=0.1, decoupled=True)
def step(r, f, a, b, yy):
    loss = bce_with_logits_loss(m(r, f, mask_rgb=a, mask_flow=b), yy)
    loss.backward()
    opt.step()
    return loss.detach()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print(f"B={B} eager train step: {timeit(lambda: step(rgb, mot, mk, mk, y))*1e3:.3f} ms")
try:
    g = GraphedCallable(step, rgb, mot, mk, mk, y)
    print(f"B={B} graphed train step (frozen host scalars, probe only): {timeit(g.replay)*1e3:.3f} ms")
except Exception as e:
    print("capture failed:", type(e).__name__, e)
