import sys, time, torch
sys.path.insert(0, '/root/repo')
from vimo_clip_amd import synth
from vimo_clip_amd.TFAM.models import AMO_CLIP
from vimo_clip_amd.graphs import GraphedCallable
from vimo_clip_amd.losses import bce_with_logits_loss
from vimo_clip_amd.optim import FusedAdam, GradArena
dev='cuda'
m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0, device=dev).to(dev)
m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
for B in (8, 64):
    rgb = synth.normal(10, f"rgb{B}", (B, 16, 768)).to(dev); mot = synth.normal(10, f"mot{B}", (B, 16, 768)).to(dev)
    mk = torch.ones(B, 16, dtype=torch.bool, device=dev)
    m.eval()
    def fwd(r, f, a, b):
        with torch.no_grad():
            return m(r, f, mask_rgb=a, mask_flow=b)
    ref = fwd(rgb, mot, mk, mk).clone()
    g = GraphedCallable(fwd, rgb, mot, mk, mk)
    out = g(rgb, mot, mk, mk)
    torch.cuda.synchronize()
    print("B", B, "graph == eager:", torch.equal(out, ref))
    for name, f in (("eager", lambda: fwd(rgb, mot, mk, mk)), ("graph", lambda: g.replay())):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
        print(f"  {name}: {dt*1e6:.1f} us/forward, {B/dt:.0f} clips/s")
# train step capture
B=8
m.train()
arena = GradArena(m.used_parameters()); opt = FusedAdam(arena, lr=1e-4, weight_decay=0.1, decoupled=True)
rgb = synth.normal(20, "rgb_t", (B, 16, 768)).to(dev); mot = synth.normal(20, "mot_t", (B, 16, 768)).to(dev)
mk = torch.ones(B, 16, dtype=torch.bool, device=dev); y = synth.multi_hot_labels(20, "lab_t", B, 140).to(dev)
def step(r, f, a, b, yy):
    loss = bce_with_logits_loss(m(r, f, mask_rgb=a, mask_flow=b), yy)
    loss.backward()
    opt.step()
    return loss.detach()
try:
    for _ in range(3): step(rgb, mot, mk, mk, y)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): step(rgb, mot, mk, mk, y)
    torch.cuda.synchronize(); print(f"train eager: {(time.perf_counter()-t0)/20*1e6:.1f} us/step")
    gs = GraphedCallable(step, rgb, mot, mk, mk, y)
    l1 = gs.replay().item(); l2 = gs.replay().item()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): gs.replay()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
    print(f"train graph: {dt*1e6:.1f} us/step, {B/dt:.0f} clips/s, losses {l1:.5f} -> {l2:.5f} (decreasing: {l2 < l1})")
except Exception as e:
    print("train-step capture failed:", repr(e)[:300])
