"""GPU: BASELINE.json configs[4] — end-to-end TFAM training on Animal Kingdom annotations (label subset fixture
recorded from the reference's dataset/annotations) with synthetic class-dependent embeddings: the HIP path (bf16)
and the fp32 CPU oracle (torch autograd through oracle/tfam.py + oracle AdamW) start from the same weights, see
the same batches, and must end with the same validation logits (within bf16 training drift) and micro-AP."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics as ometrics
from oracle import student as ostudent
from oracle import tfam as otfam
from vimo_clip_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _labels(split, n):
    z = np.load(os.path.join(ROOT, "tests", "golden", "ak_labels.npz"))
    return torch.from_numpy(np.unpackbits(z[f"{split}/labels"], axis=1)[:n, :140].astype(np.float32))


def test_tfam_training_trajectory_and_map_parity():
    from vimo_clip_amd.TFAM.data.dataset import SyntheticEmbeddingDataset
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    from vimo_clip_amd.TFAM.train_and_eval import Config, ModelTester, ModelTrainer, batches
    D, H, L, FF, C, BS, STEPS = 256, 8, 2, 512, 140, 8, 24
    ytr, yva = _labels("train", BS * STEPS), _labels("val", 64)
    assert ytr.sum() > 0 and ytr.sum(1).max() >= 2            # real multi-label rows
    tr = SyntheticEmbeddingDataset(ytr, D, tmin=9, tmax=20, seed=5, signal=0.6)
    va = SyntheticEmbeddingDataset(yva, D, tmin=9, tmax=20, seed=6, signal=0.6)
    sd0 = synth.tfam_state_dict(D, H, L, FF, C, 77)
    cfg = Config(epochs=1, batch_size=BS, d_model=D, nhead=H, num_layers=L, dim_feedforward=FF, dropout=0.0, mlp_dropout=0.0, device="cuda")
    model = AMO_CLIP(d_model=D, nhead=H, num_layers=L, dim_feedforward=FF, num_classes=C, dropout=0.0, mlp_dropout=0.0, device="cuda").cuda()
    model.load_state_dict(sd0, strict=True)
    trainer = ModelTrainer(model, tr, va, cfg)
    order = list(range(len(tr)))
    # ---- HIP: one epoch of STEPS steps, fixed order ----
    model.train()
    hip_losses = []
    for batch in batches(tr, BS, order=order):
        out, lab = trainer._forward(batch)
        loss = trainer.criterion(out, lab)
        loss.backward()
        trainer.optimizer.step()
        hip_losses.append(loss.item())
    mAP_hip, _ = ModelTester(model, va, cfg).evaluate()
    model.eval()
    with torch.no_grad():
        hip_val = torch.cat([model(b["embeddings"].cuda(), b["flow_embeddings"].cuda(), mask_rgb=b["mask_rgb"].cuda(),
                                   mask_flow=b["mask_flow"].cuda()).cpu() for b in batches(va, BS)])
    # ---- oracle: same batches, fp32 autograd + AdamW(1e-4, wd 0.1) ----
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    used = {n for n, p in model.named_parameters() if any(p is q for q in model.used_parameters())}
    sd = {k: v.clone() for k, v in sd0.items()}
    mstate = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}
    ora_losses = []
    for step, batch in enumerate(batches(tr, BS, order=order), 1):
        p = {k: v.clone().requires_grad_(k in used) for k, v in sd.items()}
        out = otfam.amo_clip_forward(p, batch["embeddings"], batch["flow_embeddings"], batch["mask_rgb"], batch["mask_flow"], nhead=H)
        loss = otfam.bce_with_logits_mean(out, batch["labels"])
        loss.backward()
        ora_losses.append(loss.item())
        for k in used:
            newp, m, v = ostudent.adam_step(sd[k], p[k].grad, mstate[k][0], mstate[k][1], step, 1e-4, weight_decay=0.1, decoupled=True)
            sd[k], mstate[k] = newp.detach(), (m.detach(), v.detach())
    with torch.no_grad():
        ora_val = torch.cat([otfam.amo_clip_forward(sd, b["embeddings"], b["flow_embeddings"], b["mask_rgb"], b["mask_flow"], nhead=H)
                             for b in batches(va, BS)])
    mAP_ora = ometrics.micro_average_precision(ometrics.maybe_sigmoid(ora_val.numpy()), torch.cat([b["labels"] for b in batches(va, BS)]).numpy())
    print(f"losses hip first/last {hip_losses[0]:.5f}/{hip_losses[-1]:.5f}  oracle {ora_losses[0]:.5f}/{ora_losses[-1]:.5f}")
    print(f"val logits max abs diff {(hip_val - ora_val).abs().max():.3e}; micro-AP hip {mAP_hip:.5f} oracle {mAP_ora:.5f}")
    assert abs(hip_losses[0] - ora_losses[0]) <= 5e-3 * ora_losses[0]
    assert abs(hip_losses[-1] - ora_losses[-1]) <= 2e-2 * ora_losses[-1]
    assert hip_losses[-1] < hip_losses[0]                       # it trains
    assert (hip_val - ora_val).abs().max().item() <= 3e-2       # bf16 training drift over 24 AdamW steps
    assert abs(mAP_hip - mAP_ora) <= 1e-2
