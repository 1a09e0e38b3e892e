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

import os

import numpy as np
import torch

from .. import parallel, synth
from ..losses import bce_with_logits_loss, cross_entropy_loss
from ..metrics import Accuracy, MultilabelAveragePrecision
from ..optim import CosineAnnealingLR, FusedAdam, GradArena
from .data.dataset import SyntheticEmbeddingDataset, collate_fn_pad
from .models import AMO_CLIP


class Config:
    """The attribute bag the reference fills from its YAML (train_and_eval_frame_diff_AK.py:311-365), plus two switches
    for the variants that are separate scripts upstream: ``task`` ("multilabel": BCE + micro-mAP, Animal Kingdom;
    "singlelabel": CrossEntropy + Accuracy, MammalNet, train_and_eval_frame_diff_MN.py:49,59) and ``motion_key``
    ("flow" / "frame_diff": the batch keys ``{motion_key}_embeddings`` and ``mask_{motion_key}``)."""

    def __init__(self, **kw):
        self.mode, self.seed, self.lr, self.epochs, self.batch_size, self.num_workers = "both", 49, 1e-4, 30, 8, 4
        self.num_classes, self.d_model, self.nhead, self.num_layers, self.dim_feedforward = 140, 512, 8, 4, 2048
        self.use_cross_attention, self.use_only_rgb, self.use_only_flow, self.use_pe, self.concat_dim = True, False, False, False, 1
        self.dropout, self.mlp_dropout, self.device, self.checkpoint_dir, self.log_dir = 0.1, 0.1, "cuda", "checkpoints", "logs"
        self.task, self.motion_key, self.use_graphs = "multilabel", "flow", False    # use_graphs: hipGraph replay of the eval forward
        self.class_names_dir = self.train_dataset_path = self.val_dataset_path = self.frame_diff_dataset_path = None
        self.__dict__.update(kw)

    @property
    def use_cross_attn(self):      # the reference stores model.use_cross_attention under this name (:359)
        return self.use_cross_attention

    @classmethod
    def from_yaml(cls, path, **overrides):
        """Sections ``training / logging / data / model`` with the reference's keys (:320-365).  Keys the reference
        requires raise KeyError when missing, as ``cfg[...]`` does there."""
        import yaml
        with open(path, "r") as f:
            cfg = yaml.safe_load(f)
        t, lg, d, m = cfg["training"], cfg["logging"], cfg["data"], cfg["model"]
        kw = dict(mode=t["mode"], seed=t["seed"], lr=float(t["lr"]), epochs=t["epochs"], batch_size=t["batch_size"],
                  num_workers=t["num_workers"], device=t["device"], log_dir=lg["log_dir"], checkpoint_dir=lg["checkpoint_dir"],
                  num_classes=d["num_classes"], class_names_dir=d["class_names_dir"], train_dataset_path=d["train_dataset_path"],
                  val_dataset_path=d["val_dataset_path"],
                  frame_diff_dataset_path=d.get("frame_diff_dataset_path", d.get("flow_dataset_path")),
                  d_model=m["d_model"], nhead=m["nhead"], num_layers=m["num_layers"], dim_feedforward=m["dim_feedforward"],
                  use_cross_attention=m["use_cross_attention"], concat_dim=m["concat_dim"], dropout=m["dropout"],
                  mlp_dropout=m["mlp_dropout"], use_pe=m["use_pe"], use_only_rgb=m["use_only_rgb"], use_only_flow=m["use_only_flow"])
        if "frame_diff_dataset_path" in d:
            kw["motion_key"] = "frame_diff"
        kw.update(overrides)
        return cls(**kw)


def task_objects(config):
    """(criterion, metric) of the task: BCE-with-logits + micro mAP (:49,58) or CrossEntropy + Accuracy (MN :49,59)."""
    if config.task == "multilabel":
        return bce_with_logits_loss, MultilabelAveragePrecision(num_labels=config.num_classes, average="micro")
    if config.task == "singlelabel":
        return cross_entropy_loss, Accuracy(num_classes=config.num_classes)
    raise ValueError(f"Unsupported task '{config.task}'. Choose 'multilabel' or 'singlelabel'.")


def set_seed(seed: int = 0):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def batches(dataset, batch_size, rank=0, world=1, drop_last=True, order=None, motion_key="flow"):
    """Contiguous per-rank shard, then fixed-size batches through collate_fn_pad (the DataLoader of :374,398)."""
    idx = list(range(len(dataset))) if order is None else list(order)
    lo, hi = parallel.shard_range(len(idx), rank, world, drop_last=True)
    idx = idx[lo:hi]
    for s in range(0, len(idx) - (batch_size - 1 if drop_last else 0), batch_size):
        yield collate_fn_pad([dataset[i] for i in idx[s:s + batch_size]], motion_key=motion_key)


def _model_forward(model, batch, config):
    dev, mk = config.device, config.motion_key
    return model(batch["embeddings"].to(dev), batch[f"{mk}_embeddings"].to(dev), mask_rgb=batch["mask_rgb"].to(dev),
                 mask_flow=batch[f"mask_{mk}"].to(dev))


class GraphedEvalForward:
    """Evaluation forward as hipGraph replays.  At the reference's batch size (8 clips) a forward is ~90 kernels of a few
    microseconds: launch bound.  One graph per distinct (B, T_rgb, T_motion) shape, captured on first sight; the reference
    pools with ``x.mean(dim=1)`` over the padded length (AMO_CLIP.py:169), so a batch cannot be padded further to share a
    graph without changing its logits -- ``bucket`` therefore defaults to 1 (exact shapes: fixed-length loaders, e.g.
    ``num_frames=16``, replay one graph; ragged loaders fill ``max_graphs`` and fall back to eager launches).  Only for
    ``model.eval()`` under ``no_grad`` (no dropout, no optimiser state inside the graph)."""

    def __init__(self, model, config, bucket=1, max_graphs=32, streams=2):
        self.model, self.config, self.bucket, self.max_graphs = model, config, bucket, max_graphs
        self._graphs = {}
        # Two batches in flight: an evaluation loop's batches are independent, and at the reference batch size a forward is a
        # chain of ~27 dependent launches of ~4 us fixed cost each, so a second stream's forward fills the first one's launch
        # gaps (measured: 205 -> 136 us per forward of 8 clips).  Each slot has its own stream, graphs, static buffers and
        # scratch (AMO_CLIP.fused_slot).
        self._streams = [torch.cuda.Stream() for _ in range(max(1, streams))] if torch.cuda.is_available() else [None]
        self._free = [None] * len(self._streams)       # event: the consumer has copied this slot's previous output
        self._rr = -1

    def _pad(self, x, T):
        if x.shape[1] == T:
            return x
        out = torch.zeros((x.shape[0], T) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        out[:, :x.shape[1]] = x
        return out

    def _run(self, batch, slot):
        from ..graphs import GraphedCallable
        dev, mk, bk = self.config.device, self.config.motion_key, self.bucket
        rgb, mot = batch["embeddings"].to(dev), batch[f"{mk}_embeddings"].to(dev)
        mr, mf = batch["mask_rgb"].to(dev), batch[f"mask_{mk}"].to(dev)
        Tr, Tf = -(-rgb.shape[1] // bk) * bk, -(-mot.shape[1] // bk) * bk
        key = (slot, rgb.shape[0], Tr, Tf, rgb.shape[2])
        self.model.fused_slot = slot
        if key not in self._graphs and len(self._graphs) >= self.max_graphs * len(self._streams):
            return self.model(rgb, mot, mask_rgb=mr, mask_flow=mf)              # too many shapes: eager
        rgb, mot, mr, mf = self._pad(rgb, Tr), self._pad(mot, Tf), self._pad(mr, Tr), self._pad(mf, Tf)
        g = self._graphs.get(key)
        if g is None:
            def fwd(a, b, c, d):
                with torch.no_grad():
                    return self.model(a, b, mask_rgb=c, mask_flow=d)
            g = self._graphs[key] = GraphedCallable(fwd, rgb, mot, mr, mf)
        return g(rgb, mot, mr, mf)

    def launch(self, batch):
        """Start the forward of `batch` on the next slot's stream; returns a handle for ``result``.  Launch batch k+1 before
        consuming batch k to keep two forwards in flight."""
        # a captured per-op forward bakes the pointers of 16-bit weight copies that an optimiser step invalidates ->
        # graphs die with the weight epoch they were captured in.  (The fused chain reads persistent packs that are rewritten
        # in place and refreshed during the re-capture's warm-up; its graphs are dropped with the others for simplicity.)
        from .. import autograd_ops as ag
        if getattr(self, "_epoch", None) != ag.weights.epoch:
            torch.cuda.synchronize()
            self._graphs.clear()
            self._epoch = ag.weights.epoch
        self._rr = slot = (self._rr + 1) % len(self._streams)
        st = self._streams[slot]
        st.wait_stream(torch.cuda.current_stream())     # weights / packs written on the caller's stream
        if self._free[slot] is not None:
            st.wait_event(self._free[slot])             # the previous output of this slot has been copied out
        with torch.cuda.stream(st):
            out = self._run(batch, slot)
            done = torch.cuda.Event()
            done.record(st)
        return out, done, slot

    def result(self, handle):
        out, done, slot = handle
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        res = out.clone()
        self._free[slot] = torch.cuda.Event()
        self._free[slot].record(cur)
        return res

    def __call__(self, batch):
        return self.result(self.launch(batch))

    def pipelined(self, batch_iter):
        """Yield (batch, logits) with one batch of look-ahead (two forwards in flight)."""
        pending = None
        for batch in batch_iter:
            h = self.launch(batch)
            if pending is not None:
                yield pending[0], self.result(pending[1])
            pending = (batch, h)
        if pending is not None:
            yield pending[0], self.result(pending[1])


class ModelTrainer:
    def __init__(self, model, train_set, val_set, config, rank=0, world=1):
        self.model, self.train_set, self.val_set, self.config, self.rank, self.world = model, train_set, val_set, config, rank, world
        self.criterion, self.mAP_metric = task_objects(config)       # the reference keeps the name mAP_metric / metric for both tasks
        self.metric = self.mAP_metric
        self.best_val_mAP, self.best_val_loss = 0.0, float("inf")
        self.arena = GradArena(model.used_parameters())
        parallel.broadcast_parameters(self.arena.flat_param)
        self.optimizer = FusedAdam(self.arena, lr=1e-4, weight_decay=0.1, decoupled=True)       # lr hard-coded as :53
        self.scheduler = CosineAnnealingLR(self.optimizer, T_max=config.epochs, eta_min=1e-6)
        self.reducer = parallel.GradientAllReducer(self.arena.flat_grad).attach(self.arena)   # buckets go out during the backward
        self._graphed_eval = GraphedEvalForward(model, config) if getattr(config, "use_graphs", False) else None
        self._graphed_train = None
        if getattr(config, "use_graphs", False):
            # captured training steps: step count / lr / dropout seeds in device memory (optim.FusedAdam.enable_device_state).
            # One process: the whole step is one graph.  Data parallel: forward + backward graph, the gradient exchange, optimiser graph.
            from ..graphs import GraphedTrainStep
            self.optimizer.enable_device_state(base_seed=config.seed if world == 1 else config.seed * 1000 + rank)
            model.use_device_seeds(self.optimizer)
            if world == 1:
                self._graphed_train = GraphedTrainStep(self._device_state_step, self.optimizer)
            else:
                self._graphed_train = GraphedTrainStep(self._device_state_fwd_bwd, self.optimizer, exchange=self.reducer.all_reduce,
                                                       opt_fn=self.optimizer.step)
        if world == 1 and os.environ.get("VMC_ADAM_OVERLAP", "0") == "1":
            # AdamW of a finished layer on a side stream beside the backward of the layers below it.  Bit-identical, but measured
            # SLOWER on MI355X (captured B = 8 step 0.88 -> 0.95-1.01 ms: the fork / join edges of a multi-stream hipGraph cost more
            # than the overlap returns; profiles/README.md), so it is opt-in
            self.optimizer.enable_backward_overlap(model.parameter_groups_by_layer())
            model.grad_group_callback = self.optimizer.group_ready

    def _forward(self, batch):
        return _model_forward(self.model, batch, self.config), batch["labels"].to(self.config.device)

    def _device_state_step(self, rgb, mot, mr, mf, labels):
        """tick + forward + loss + backward + AdamW with every step-dependent scalar read from device memory."""
        from ..losses import loss_and_grad
        self.optimizer.tick()
        output = self.model(rgb, mot, mask_rgb=mr, mask_flow=mf)
        loss, dlogits = loss_and_grad(self.criterion, output, labels)      # criterion(output, labels); loss.backward() (:81-83)
        output.backward(dlogits)
        self.optimizer.step()
        return loss, output.detach()

    def _device_state_fwd_bwd(self, rgb, mot, mr, mf, labels):
        """The data-parallel step's first graph: tick + forward + loss + backward (the exchange and AdamW follow outside it)."""
        from ..losses import loss_and_grad
        self.optimizer.tick()
        output = self.model(rgb, mot, mask_rgb=mr, mask_flow=mf)
        loss, dlogits = loss_and_grad(self.criterion, output, labels)
        output.backward(dlogits)
        return loss, output.detach()

    def train_epoch(self, epoch):
        self.model.train()
        self.mAP_metric.reset()
        total, n = torch.zeros((), device=self.config.device), 0
        g = torch.Generator().manual_seed(self.config.seed + epoch)
        order = torch.randperm(len(self.train_set), generator=g).tolist()
        if self._graphed_train is not None:
            self.optimizer.sync_hyper()                     # the epoch's learning rate -> device memory
        for batch in batches(self.train_set, self.config.batch_size, self.rank, self.world, order=order, motion_key=self.config.motion_key):
            if self._graphed_train is not None:
                dev, mk = self.config.device, self.config.motion_key
                labels = batch["labels"].to(dev)
                loss, output = self._graphed_train(batch["embeddings"].to(dev), batch[f"{mk}_embeddings"].to(dev),
                                                   batch["mask_rgb"].to(dev), batch[f"mask_{mk}"].to(dev), labels)
                loss, output = loss.clone(), output.clone()
            else:
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
            it = batches(self.val_set, self.config.batch_size, self.rank, self.world, motion_key=self.config.motion_key)
            pairs = self._graphed_eval.pipelined(it) if self._graphed_eval is not None else ((b, self._forward(b)[0]) for b in it)
            for batch, output in pairs:
                labels = batch["labels"].to(self.config.device)
                total += self.criterion(output, labels)
                n += 1
                self.mAP_metric.update(output, labels.to(dtype=torch.int))
        stats = parallel.all_reduce_scalars(torch.stack([total, torch.tensor(float(n), device=total.device)]))
        return float(stats[0] / stats[1].clamp(min=1)), float(self.mAP_metric.compute(distributed=self.world > 1))

    def save_checkpoint(self, val_loss, val_mAP, epoch, best=False, path=None):
        """:133-148: the checkpoint dict is written as ``<checkpoint_dir>/best_model.pth`` only when the validation mAP improves
        (rank 0; ``checkpoint_dir = None`` keeps it in memory only).  Tensors are cloned: ``state_dict()`` returns views into
        the gradient arena, and optimiser moments are live buffers."""
        from ..checkpoint import snapshot
        improved = val_mAP > self.best_val_mAP
        if improved:
            self.best_val_mAP, self.best_val_loss = val_mAP, val_loss
        opt = {k: (v.detach().to("cpu", copy=True) if torch.is_tensor(v) else v) for k, v in self.optimizer.state_dict().items()}
        state = {"epoch": epoch, "state_dict": snapshot(self.model), "optimizer": opt,
                 "scheduler": self.scheduler.state_dict(), "best_val_loss": self.best_val_loss, "best_val_mAP": self.best_val_mAP}
        ckpt_dir = getattr(self.config, "checkpoint_dir", None)
        if self.rank == 0:
            if improved and ckpt_dir:
                os.makedirs(ckpt_dir, exist_ok=True)
                torch.save(state, os.path.join(ckpt_dir, "best_model.pth"))
            if path:
                torch.save(state, path)
        return state

    def train(self):
        t0 = time.time()
        for epoch in range(self.config.epochs):
            tl, tm = self.train_epoch(epoch)
            vl, vm = self.validate(epoch)
            self.save_checkpoint(vl, vm, epoch)          # best_model.pth when the mAP improved (:157-160)
            self.scheduler.step()
            if self.rank == 0:
                print(json.dumps({"epoch": epoch + 1, "train_loss": tl, "train_mAP": tm, "val_loss": vl, "val_mAP": vm,
                                  "lr": self.scheduler.get_last_lr()[0], "elapsed_s": round(time.time() - t0, 1)}), flush=True)
        return self.best_val_mAP


class ModelTester:
    def __init__(self, model, test_set, config, rank=0, world=1):
        self.model, self.test_set, self.config, self.rank, self.world = model, test_set, config, rank, world
        _, self.mAP_metric = task_objects(config)
        self._graphed_eval = GraphedEvalForward(model, config) if getattr(config, "use_graphs", False) else None

    def load_best_model(self, checkpoint_dir):
        """:186-191 — weights-only load of ``best_model.pth`` (keys with or without the DataParallel ``module.`` prefix)."""
        from ..checkpoint import load_state_dict
        load_state_dict(self.model, os.path.join(checkpoint_dir, "best_model.pth"))
        from .. import autograd_ops
        autograd_ops.weights.clear()      # new masters: cached 16-bit copies, weight packs and captured forwards are stale (bumps the epoch)
        self.model.eval()
        return self

    def evaluate(self, k=5):
        """sigmoid top-k predictions + micro mAP (:193-248)."""
        self.model.eval()
        results, dev = {}, self.config.device
        with torch.no_grad():
            it = batches(self.test_set, self.config.batch_size, self.rank, self.world, motion_key=self.config.motion_key)
            pairs = (self._graphed_eval.pipelined(it) if self._graphed_eval is not None
                     else ((b, _model_forward(self.model, b, self.config)) for b in it))
            for batch, out in pairs:
                self.mAP_metric.update(out, batch["labels"].to(dev).to(torch.int))
                probs = torch.sigmoid(out) if self.config.task == "multilabel" else torch.softmax(out, dim=1)
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


def build_model(cfg):
    """AMO_CLIP(...) exactly as :379-393 builds it from the config."""
    return AMO_CLIP(d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers, dim_feedforward=cfg.dim_feedforward,
                    num_classes=cfg.num_classes, use_only_rgb=cfg.use_only_rgb, use_only_flow=cfg.use_only_flow, use_pe=cfg.use_pe,
                    use_cross_attention=cfg.use_cross_attention, concat_dim=cfg.concat_dim, dropout=cfg.dropout,
                    mlp_dropout=cfg.mlp_dropout, device=cfg.device).to(cfg.device)


def build_datasets(cfg, limit=2048, train_annotations=None, val_annotations=None):
    """HDF5 datasets when the config names existing files (:375,396), otherwise synthetic embeddings over real or
    synthetic label rows (there are no AK / MammalNet embeddings offline)."""
    import os
    paths = (cfg.train_dataset_path, cfg.val_dataset_path, cfg.frame_diff_dataset_path)
    if all(p and os.path.exists(p) for p in paths):
        from .data import dataset as dflow
        from .data import dataset_frame_diff as ddiff
        D = ddiff.HDF5VideoDataset if cfg.motion_key == "frame_diff" else dflow.HDF5VideoDataset
        return D(cfg.train_dataset_path, cfg.frame_diff_dataset_path), D(cfg.val_dataset_path, cfg.frame_diff_dataset_path)
    C = cfg.num_classes
    if cfg.task == "singlelabel":
        one_hot = lambda seed, n: torch.nn.functional.one_hot(synth.randint(seed, "cls", (n,), 0, C), C).float()
        tl, vl = one_hot(1, limit), one_hot(2, limit // 4)
    else:
        tl = _labels_from_annotations(train_annotations, C, limit) if train_annotations else synth.multi_hot_labels(1, "tr", limit, C)
        vl = _labels_from_annotations(val_annotations, C, limit // 4) if val_annotations else synth.multi_hot_labels(2, "va", limit // 4, C)
    return (SyntheticEmbeddingDataset(tl, cfg.d_model, seed=5, motion_key=cfg.motion_key, class_seed=5),
            SyntheticEmbeddingDataset(vl, cfg.d_model, seed=6, motion_key=cfg.motion_key, class_seed=5))


def run(cfg, rank=0, world=1, limit=2048, train_annotations=None, val_annotations=None):
    """The ``__main__`` block of the reference (:367-408): train and / or test according to ``cfg.mode``."""
    set_seed(cfg.seed)
    train_set, val_set = build_datasets(cfg, limit, train_annotations, val_annotations)
    model = build_model(cfg)
    model.set_dropout_seed(cfg.seed * 1000 + rank)
    out = {"world": world, "task": cfg.task}
    wrote_best = False
    if cfg.mode in ("train", "both"):
        trainer = ModelTrainer(model, train_set, val_set, cfg, rank, world)
        out["best_val_metric"] = trainer.train()
        # rank 0 writes best_model.pth exactly when the validation metric improved during THIS run (save_checkpoint)
        wrote_best = bool(getattr(cfg, "checkpoint_dir", None)) and trainer.best_val_mAP > 0.0
    if cfg.mode in ("test", "both"):
        tester = ModelTester(model, val_set, cfg, rank, world)
        best = os.path.join(cfg.checkpoint_dir, "best_model.pth") if getattr(cfg, "checkpoint_dir", None) else None
        # every rank must take the same branch.  Rank 0 decides (after training: "did this run write the file";
        # a best_model.pth left behind by an earlier run is NOT picked up silently; test-only mode: "does the file exist"),
        # the decision is broadcast, and the barrier that orders rank 0's write before the other ranks' reads is unconditional.
        have = wrote_best if cfg.mode == "both" else bool(best and os.path.exists(best))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()                              # rank 0 has finished writing it
            flag = [have]
            dist.broadcast_object_list(flag, src=0)
            have = bool(flag[0])
        if have:                                        # :404-406: the tester evaluates the best checkpoint, not the last epoch
            tester.load_best_model(cfg.checkpoint_dir)
        elif cfg.mode == "test":
            raise FileNotFoundError(f"mode='test' needs {best or '<checkpoint_dir>/best_model.pth'} (TFAM/train_and_eval.py:404-406)")
        out["test_metric"], _ = tester.evaluate()
    return out


def main(default_task="multilabel", default_motion_key="flow"):
    ap = argparse.ArgumentParser(description="TFAM train/eval on MI355X (HDF5 embeddings, or synthetic ones over real / synthetic labels)")
    ap.add_argument("--config", default=None, help="YAML in the reference's schema (TFAM/cfg_AK/*.yaml)")
    ap.add_argument("--task", default=default_task, choices=["multilabel", "singlelabel"])
    ap.add_argument("--train-annotations", default=None, help="train_multi.txt (video_id class ids...); synthetic labels if omitted")
    ap.add_argument("--val-annotations", default=None)
    ap.add_argument("--limit", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--d-model", type=int, default=None)
    ap.add_argument("--dropout", type=float, default=None)
    args = ap.parse_args()
    rank, world, local = parallel.init_from_env()
    over = {k: v for k, v in dict(epochs=args.epochs, batch_size=args.batch_size, d_model=args.d_model, dropout=args.dropout,
                                  mlp_dropout=args.dropout).items() if v is not None}
    over.update(task=args.task, device=f"cuda:{local}")
    if args.config:
        cfg = Config.from_yaml(args.config, **over)
    else:
        cfg = Config(**{**dict(epochs=3, motion_key=default_motion_key), **over})
    res = run(cfg, rank, world, args.limit, args.train_annotations, args.val_annotations)
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
