"""MoCLIP student training loop on the MI355X engine — the hot loop of the reference's train.py (train :52-175,
evaluate :14-49) with the same argument names, one process per GPU under torchrun instead of nn.DataParallel (:64).
``--clip_embeddings_dir`` / ``--flow_videos_dir`` read the reference's HDF5 + video layout (HDF5 natively through
h5lite; decoded ``.npy`` frame stacks when no video decoder is installed); ``--synthetic N`` trains on N synthetic
segments.  ``--single_label`` is the MammalNet variant (train_frame_diff_mn.py:82,102: CrossEntropy on
``labels.argmax(dim=1)``, no gradient clipping); ``--motion_key frame_diff`` the frame-difference twin
(train_frame_diff.py: batch key ``frame_diff`` instead of ``flow_frames``).
"""
from __future__ import annotations

import argparse
import json
import time

import torch

from . import parallel
from .dataset import SyntheticSegmentDataset, collate_fn
from .losses import classification_loss, cross_entropy_loss, distillation_loss
from .models.student_model import FlowStudentModel
from .optim import FusedAdam, GradArena


def _batches(ds, batch_size, rank, world):
    lo, hi = parallel.shard_range(len(ds), rank, world)
    for s in range(lo, hi - batch_size + 1, batch_size):
        yield collate_fn([ds[i] for i in range(s, s + batch_size)])


def _frames(batch):
    return batch["flow_frames"] if "flow_frames" in batch else batch["frame_diff"]


def _class_loss(logits, labels, class_positive_weight, single_label):
    if single_label:                                   # train_frame_diff_mn.py:102
        return cross_entropy_loss(logits, labels.argmax(dim=1))
    return classification_loss(logits, labels, positive_weight=class_positive_weight)


def evaluate(model, val_set, device, distillation_loss_mode, class_positive_weight, batch_size, rank=0, world=1, single_label=False):
    model.eval()
    tot = torch.zeros(4, device=device)
    with torch.no_grad():
        for batch in _batches(val_set, batch_size, rank, world):
            _, emb_d, logits = model(_frames(batch).to(device))
            dl = distillation_loss(emb_d, batch["rgb_emb"].to(device)[:, :-1, :], mode=distillation_loss_mode)
            cl = _class_loss(logits, batch["labels"].to(device), class_positive_weight, single_label)
            tot += torch.stack([dl, cl, dl + cl, torch.ones((), device=device)])
    tot = parallel.all_reduce_scalars(tot)
    n = max(1.0, float(tot[3]))
    return float(tot[0]) / n, float(tot[1]) / n, float(tot[2]) / n


def train(args):
    rank, world, local = parallel.init_from_env()
    device = f"cuda:{local}"
    E = {"ViT-B/32": 512, "ViT-B/16": 512, "ViT-L/14": 768}.get(args.clip_model_name, 512)
    single = bool(getattr(args, "single_label", False))
    if getattr(args, "clip_embeddings_dir", None):
        from .dataset import HDF5VideoDataset
        train_set = HDF5VideoDataset(args.clip_embeddings_dir, args.flow_videos_dir, sequence_length=args.sequence_length)
        val_set = HDF5VideoDataset(args.val_clip_embeddings_dir or args.clip_embeddings_dir, args.val_flow_videos_dir or args.flow_videos_dir,
                                   sequence_length=args.sequence_length)
    else:
        train_set = SyntheticSegmentDataset(args.synthetic, args.sequence_length, E, args.num_classes, seed=3)
        val_set = SyntheticSegmentDataset(max(args.batch_size * world, args.synthetic // 8), args.sequence_length, E, args.num_classes, seed=4)
    model = FlowStudentModel(clip_model_name=args.clip_model_name, device=device, num_classes=args.num_classes, alpha=args.residual_alpha)
    arena = GradArena(model.parameters())
    parallel.broadcast_parameters(arena.flat_param)
    optimizer = FusedAdam(arena, lr=args.learning_rate)
    reducer = parallel.GradientAllReducer(arena.flat_grad).attach(arena)     # buckets go out during the backward
    best = float("inf")
    ckpt_dir = getattr(args, "checkpoint_dir", None)       # reference: checkpoints/<run timestamp> (train.py:69-74)
    for epoch in range(args.epochs):
        model.train()
        t0, nframes = time.time(), 0
        for batch in _batches(train_set, args.batch_size, rank, world):
            frames = _frames(batch)
            emb, emb_d, logits = model(frames.to(device))
            dl = distillation_loss(emb_d, batch["rgb_emb"].to(device)[:, :-1, :], mode=args.distillation_loss_mode)
            cl = _class_loss(logits, batch["labels"].to(device), args.class_positive_weight, single)
            (dl + cl).backward()
            optimizer.step(grad_scale=reducer.all_reduce(), max_grad_norm=None if single else args.grad_clip_norm)
            nframes += frames.shape[0] * frames.shape[1]
        torch.cuda.synchronize()
        vd, vc, vt = evaluate(model, val_set, device, args.distillation_loss_mode, args.class_positive_weight, args.batch_size, rank, world, single)
        if rank == 0 and ckpt_dir:                          # train.py:164-172: every epoch + the best-so-far copy
            from .checkpoint import save_state_dict
            save_state_dict(model, f"{ckpt_dir}/student_epoch_{epoch + 1}.pth")
            if vt < best:
                save_state_dict(model, f"{ckpt_dir} - best/student_best.pth")
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
    p.add_argument("--checkpoint_dir", default=None, help="write student_epoch_N.pth here and student_best.pth into '<dir> - best' "
                                                           "(the reference uses checkpoints/<timestamp>); omitted = no files")
    p.add_argument("--single_label", action="store_true", help="MammalNet variant: CrossEntropy on labels.argmax(1)")
    p.add_argument("--clip_embeddings_dir", default=None, help="HDF5 file of teacher embeddings (reference layout)")
    p.add_argument("--flow_videos_dir", default=None)
    p.add_argument("--val_clip_embeddings_dir", default=None)
    p.add_argument("--val_flow_videos_dir", default=None)
    train(p.parse_args())
