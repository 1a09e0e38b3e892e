"""MoCLIP student training loop on the MI355X engine — the hot loop of the reference's train.py (train :52-175,
evaluate :14-49) with the same argument names, one process per GPU under torchrun instead of nn.DataParallel (:64).
Real data needs h5py + a video decoder (absent offline): ``--synthetic N`` trains on N synthetic segments instead.
"""
from __future__ import annotations

import argparse
import json
import time

import torch

from . import parallel
from .dataset import SyntheticSegmentDataset, collate_fn
from .losses import classification_loss, distillation_loss
from .models.student_model import FlowStudentModel
from .optim import FusedAdam, GradArena


def _batches(ds, batch_size, rank, world):
    lo, hi = parallel.shard_range(len(ds), rank, world)
    for s in range(lo, hi - batch_size + 1, batch_size):
        yield collate_fn([ds[i] for i in range(s, s + batch_size)])


def evaluate(model, val_set, device, distillation_loss_mode, class_positive_weight, batch_size, rank=0, world=1):
    model.eval()
    tot = torch.zeros(4, device=device)
    with torch.no_grad():
        for batch in _batches(val_set, batch_size, rank, world):
            _, emb_d, logits = model(batch["flow_frames"].to(device))
            dl = distillation_loss(emb_d, batch["rgb_emb"].to(device)[:, :-1, :], mode=distillation_loss_mode)
            cl = classification_loss(logits, batch["labels"].to(device), positive_weight=class_positive_weight)
            tot += torch.stack([dl, cl, dl + cl, torch.ones((), device=device)])
    tot = parallel.all_reduce_scalars(tot)
    n = max(1.0, float(tot[3]))
    return float(tot[0]) / n, float(tot[1]) / n, float(tot[2]) / n


def train(args):
    rank, world, local = parallel.init_from_env()
    device = f"cuda:{local}"
    E = {"ViT-B/32": 512, "ViT-B/16": 512, "ViT-L/14": 768}.get(args.clip_model_name, 512)
    train_set = SyntheticSegmentDataset(args.synthetic, args.sequence_length, E, args.num_classes, seed=3)
    val_set = SyntheticSegmentDataset(max(args.batch_size * world, args.synthetic // 8), args.sequence_length, E, args.num_classes, seed=4)
    model = FlowStudentModel(clip_model_name=args.clip_model_name, device=device, num_classes=args.num_classes, alpha=args.residual_alpha)
    arena = GradArena(model.parameters())
    parallel.broadcast_parameters(arena.flat_param)
    optimizer = FusedAdam(arena, lr=args.learning_rate)
    reducer = parallel.GradientAllReducer(arena.flat_grad)
    best = float("inf")
    for epoch in range(args.epochs):
        model.train()
        t0, nframes = time.time(), 0
        for batch in _batches(train_set, args.batch_size, rank, world):
            emb, emb_d, logits = model(batch["flow_frames"].to(device))
            dl = distillation_loss(emb_d, batch["rgb_emb"].to(device)[:, :-1, :], mode=args.distillation_loss_mode)
            cl = classification_loss(logits, batch["labels"].to(device), positive_weight=args.class_positive_weight)
            (dl + cl).backward()
            optimizer.step(grad_scale=reducer.all_reduce(), max_grad_norm=args.grad_clip_norm)
            nframes += batch["flow_frames"].shape[0] * batch["flow_frames"].shape[1]
        torch.cuda.synchronize()
        vd, vc, vt = evaluate(model, val_set, device, args.distillation_loss_mode, args.class_positive_weight, args.batch_size, rank, world)
        best = min(best, vt)
        if rank == 0:
            print(json.dumps({"epoch": epoch + 1, "val_distill": vd, "val_class": vc, "val_total": vt,
                              "train_frames_per_s_all_gpus": round(nframes * world / (time.time() - t0), 1)}), flush=True)
    return best


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Train flow-only student model (MI355X engine)")
    p.add_argument("--synthetic", type=int, default=256)
    p.add_argument("--clip_model_name", default="ViT-B/32")
    p.add_argument("--num_classes", type=int, default=140)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--sequence_length", type=int, default=17)
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=1e-3)
    p.add_argument("--distillation_loss_mode", default="cosine")
    p.add_argument("--class_positive_weight", type=int, default=9)
    p.add_argument("--residual_alpha", type=float, default=0.1)
    p.add_argument("--grad_clip_norm", type=float, default=None)
    train(p.parse_args())
