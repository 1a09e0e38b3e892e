"""TFAM training / evaluation harness on the MI355X engine — the hot loop of the reference's
TFAM/train_and_eval.py (ModelTrainer.train_epoch :66-101, validate :103-131, save_checkpoint :133-148, cosine
schedule :54-56,162, ModelTester.evaluate :193-248) with the same method names and checkpoint dict layout, run as one
process per GPU (``torchrun --nproc-per-node N -m vimo_clip_amd.TFAM.train_and_eval --synthetic ...``) instead of
``nn.DataParallel`` (:392).  TensorBoard / tabulate reporting of the reference is host-side tooling and not reproduced;
scalars are printed.  Kept quirks: AdamW lr is 1e-4 whatever the config says (:53), loaders drop the last batch (:374,398).
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from .. import parallel, synth
from ..losses import bce_with_logits_loss
from ..metrics import MultilabelAveragePrecision
from ..optim import CosineAnnealingLR, FusedAdam, GradArena
from .data.dataset import SyntheticEmbeddingDataset, collate_fn_pad
from .models import AMO_CLIP


class Config:
    def __init__(self, **kw):
        self.mode, self.seed, self.lr, self.epochs, self.batch_size = "both", 49, 1e-4, 30, 8
        self.num_classes, self.d_model, self.nhead, self.num_layers, self.dim_feedforward = 140, 512, 8, 4, 2048
        self.use_cross_attention, self.use_only_rgb, self.use_only_flow, self.use_pe, self.concat_dim = True, False, False, False, 1
        self.dropout, self.mlp_dropout, self.device, self.checkpoint_dir = 0.1, 0.1, "cuda", "checkpoints"
        self.__dict__.update(kw)


def set_seed(seed: int = 0):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def batches(dataset, batch_size, rank=0, world=1, drop_last=True, order=None):
    """Contiguous per-rank shard, then fixed-size batches through collate_fn_pad (the DataLoader of :374,398)."""
    idx = list(range(len(dataset))) if order is None else list(order)
    lo, hi = parallel.shard_range(len(idx), rank, world, drop_last=True)
    idx = idx[lo:hi]
    for s in range(0, len(idx) - (batch_size - 1 if drop_last else 0), batch_size):
        yield collate_fn_pad([dataset[i] for i in idx[s:s + batch_size]])


class ModelTrainer:
    def __init__(self, model, train_set, val_set, config, rank=0, world=1):
        self.model, self.train_set, self.val_set, self.config, self.rank, self.world = model, train_set, val_set, config, rank, world
        self.mAP_metric = MultilabelAveragePrecision(num_labels=config.num_classes, average="micro")
        self.best_val_mAP, self.best_val_loss = 0.0, float("inf")
        self.arena = GradArena(model.used_parameters())
        parallel.broadcast_parameters(self.arena.flat_param)
        self.optimizer = FusedAdam(self.arena, lr=1e-4, weight_decay=0.1, decoupled=True)       # lr hard-coded as :53
        self.scheduler = CosineAnnealingLR(self.optimizer, T_max=config.epochs, eta_min=1e-6)
        self.reducer = parallel.GradientAllReducer(self.arena.flat_grad)
        self.criterion = bce_with_logits_loss

    def _forward(self, batch):
        dev = self.config.device
        out = self.model(batch["embeddings"].to(dev), batch["flow_embeddings"].to(dev), mask_rgb=batch["mask_rgb"].to(dev),
                         mask_flow=batch["mask_flow"].to(dev))
        return out, batch["labels"].to(dev)

    def train_epoch(self, epoch):
        self.model.train()
        self.mAP_metric.reset()
        total, n = torch.zeros((), device=self.config.device), 0
        g = torch.Generator().manual_seed(self.config.seed + epoch)
        order = torch.randperm(len(self.train_set), generator=g).tolist()
        for batch in batches(self.train_set, self.config.batch_size, self.rank, self.world, order=order):
            output, labels = self._forward(batch)
            loss = self.criterion(output, labels)
            loss.backward()
            self.optimizer.step(grad_scale=self.reducer.all_reduce())
            total += loss.detach()
            n += 1
            self.mAP_metric.update(output, labels.to(dtype=torch.int))
        stats = parallel.all_reduce_scalars(torch.stack([total, torch.tensor(float(n), device=total.device)]))
        return float(stats[0] / stats[1].clamp(min=1)), float(self.mAP_metric.compute(distributed=self.world > 1))

    def validate(self, epoch):
        self.model.eval()
        self.mAP_metric.reset()
        total, n = torch.zeros((), device=self.config.device), 0
        with torch.no_grad():
            for batch in batches(self.val_set, self.config.batch_size, self.rank, self.world):
                output, labels = self._forward(batch)
                total += self.criterion(output, labels)
                n += 1
                self.mAP_metric.update(output, labels.to(dtype=torch.int))
        stats = parallel.all_reduce_scalars(torch.stack([total, torch.tensor(float(n), device=total.device)]))
        return float(stats[0] / stats[1].clamp(min=1)), float(self.mAP_metric.compute(distributed=self.world > 1))

    def save_checkpoint(self, val_loss, val_mAP, epoch, best=False, path=None):
        state = {"epoch": epoch, "state_dict": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
                 "scheduler": self.scheduler.state_dict(), "best_val_loss": self.best_val_loss, "best_val_mAP": self.best_val_mAP}
        if path and self.rank == 0:
            torch.save(state, path)
        return state

    def train(self):
        t0 = time.time()
        for epoch in range(self.config.epochs):
            tl, tm = self.train_epoch(epoch)
            vl, vm = self.validate(epoch)
            if vm > self.best_val_mAP:
                self.best_val_mAP, self.best_val_loss = vm, vl
            self.scheduler.step()
            if self.rank == 0:
                print(json.dumps({"epoch": epoch + 1, "train_loss": tl, "train_mAP": tm, "val_loss": vl, "val_mAP": vm,
                                  "lr": self.scheduler.get_last_lr()[0], "elapsed_s": round(time.time() - t0, 1)}), flush=True)
        return self.best_val_mAP


class ModelTester:
    def __init__(self, model, test_set, config, rank=0, world=1):
        self.model, self.test_set, self.config, self.rank, self.world = model, test_set, config, rank, world
        self.mAP_metric = MultilabelAveragePrecision(num_labels=config.num_classes, average="micro")

    def evaluate(self, k=5):
        """sigmoid top-k predictions + micro mAP (:193-248)."""
        self.model.eval()
        results, dev = {}, self.config.device
        with torch.no_grad():
            for batch in batches(self.test_set, self.config.batch_size, self.rank, self.world):
                out = self.model(batch["embeddings"].to(dev), batch["flow_embeddings"].to(dev), mask_rgb=batch["mask_rgb"].to(dev),
                                 mask_flow=batch["mask_flow"].to(dev))
                self.mAP_metric.update(out, batch["labels"].to(dev).to(torch.int))
                probs = torch.sigmoid(out)
                top = torch.topk(probs, k, dim=1)
                for vid, idx, pr in zip(batch["video_id"], top.indices.tolist(), top.values.tolist()):
                    results[vid] = {"top_classes": idx, "top_probs": pr}
        return float(self.mAP_metric.compute(distributed=self.world > 1)), results


def _labels_from_annotations(path, num_classes, limit=None):
    with open(path, "r", encoding="utf-8") as f:
        ann = [line.strip().split() for line in f if line.strip()]
    ann = ann[:limit] if limit else ann
    lab = torch.zeros(len(ann), num_classes)
    for i, a in enumerate(ann):
        lab[i, [int(c) for c in a[1:]]] = 1.0
    return lab


def main():
    ap = argparse.ArgumentParser(description="TFAM train/eval on MI355X (synthetic embeddings over real or synthetic labels)")
    ap.add_argument("--train-annotations", default=None, help="train_multi.txt (video_id class ids...); synthetic labels if omitted")
    ap.add_argument("--val-annotations", default=None)
    ap.add_argument("--limit", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--d-model", type=int, default=512)
    ap.add_argument("--dropout", type=float, default=0.1)
    args = ap.parse_args()
    rank, world, local = parallel.init_from_env()
    cfg = Config(epochs=args.epochs, batch_size=args.batch_size, d_model=args.d_model, dropout=args.dropout, mlp_dropout=args.dropout,
                 device=f"cuda:{local}")
    set_seed(cfg.seed)
    tl = _labels_from_annotations(args.train_annotations, 140, args.limit) if args.train_annotations else synth.multi_hot_labels(1, "tr", args.limit, 140)
    vl = _labels_from_annotations(args.val_annotations, 140, args.limit // 4) if args.val_annotations else synth.multi_hot_labels(2, "va", args.limit // 4, 140)
    train_set, val_set = SyntheticEmbeddingDataset(tl, cfg.d_model, seed=5), SyntheticEmbeddingDataset(vl, cfg.d_model, seed=6)
    model = AMO_CLIP(d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers, dim_feedforward=cfg.dim_feedforward,
                     num_classes=cfg.num_classes, dropout=cfg.dropout, mlp_dropout=cfg.mlp_dropout, device=cfg.device).to(cfg.device)
    model.set_dropout_seed(cfg.seed * 1000 + rank)
    best = ModelTrainer(model, train_set, val_set, cfg, rank, world).train()
    mAP, _ = ModelTester(model, val_set, cfg, rank, world).evaluate()
    if rank == 0:
        print(json.dumps({"best_val_mAP": best, "test_mAP": mAP, "world": world}))


if __name__ == "__main__":
    main()
