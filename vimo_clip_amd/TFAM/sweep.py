"""The hyper-parameter sweep of the reference (TFAM/cfg_AK/config_*.yaml driven by TFAM/run_experiments.sh): fusion mode x
positional encoding x dropout pair, one YAML per run in the reference's schema, executed one after the other.

    python -m vimo_clip_amd.TFAM.sweep --out cfg --write            # emit the YAML files
    python -m vimo_clip_amd.TFAM.sweep --out cfg --run --epochs 3   # run them in sequence (one JSON line each)
"""
from __future__ import annotations

import argparse
import json
import os

FUSION_MODES = {                       # the five ways AMO_CLIP combines the streams (AMO_CLIP.py:104-150)
    "cross": dict(use_cross_attention=True, use_only_rgb=False, use_only_flow=False, concat_dim=1),
    "concat_time": dict(use_cross_attention=False, use_only_rgb=False, use_only_flow=False, concat_dim=1),
    "concat_feature": dict(use_cross_attention=False, use_only_rgb=False, use_only_flow=False, concat_dim=-1),
    "rgb_only": dict(use_cross_attention=False, use_only_rgb=True, use_only_flow=False, concat_dim=1),
    "motion_only": dict(use_cross_attention=False, use_only_rgb=False, use_only_flow=True, concat_dim=1),
}
DROPOUTS = ((0.1, 0.1), (0.2, 0.3))


def sweep_configs(base=None):
    """name -> nested dict (training / logging / data / model) for every point of the grid."""
    base = base or {}
    out = {}
    for fname, fusion in FUSION_MODES.items():
        for pe in (False, True):
            for dp, mdp in DROPOUTS:
                name = f"{fname}{'_pe' if pe else ''}_do{int(dp * 10)}{int(mdp * 10)}"
                out[name] = {
                    "training": {"mode": "both", "seed": 49, "lr": 1e-4, "epochs": 30, "batch_size": 8, "num_workers": 4, "device": "cuda:0",
                                 **base.get("training", {})},
                    "logging": {"log_dir": "logs", "checkpoint_dir": "checkpoints", **base.get("logging", {})},
                    "data": {"num_classes": 140, "class_names_dir": "", "train_dataset_path": "", "val_dataset_path": "",
                             "frame_diff_dataset_path": "", **base.get("data", {})},
                    "model": {"d_model": 512, "nhead": 8, "num_layers": 4, "dim_feedforward": 2048, "use_pe": pe, "dropout": dp,
                              "mlp_dropout": mdp, **fusion, **base.get("model", {})},
                }
    return out


def write_sweep(out_dir, base=None):
    import yaml
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for name, cfg in sweep_configs(base).items():
        p = os.path.join(out_dir, f"config_{name}.yaml")
        with open(p, "w") as f:
            yaml.safe_dump(cfg, f, sort_keys=False)
        paths.append(p)
    return paths


def run_experiments(config_paths, rank=0, world=1, limit=2048, **overrides):
    """run_experiments.sh: every configuration in sequence; one result dict per configuration."""
    from .train_and_eval import Config, run
    results = {}
    for p in config_paths:
        cfg = Config.from_yaml(p, **overrides)
        results[os.path.basename(p)] = run(cfg, rank, world, limit)
        if rank == 0:
            print(json.dumps({"config": os.path.basename(p), **results[os.path.basename(p)]}), flush=True)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="cfg_sweep")
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--limit", type=int, default=2048)
    args = ap.parse_args()
    paths = write_sweep(args.out) if args.write or not os.path.isdir(args.out) else sorted(
        os.path.join(args.out, f) for f in os.listdir(args.out) if f.endswith(".yaml"))
    if args.run:
        from .. import parallel
        rank, world, local = parallel.init_from_env()
        over = {"device": f"cuda:{local}"}
        if args.epochs:
            over["epochs"] = args.epochs
        run_experiments(paths, rank, world, args.limit, **over)


if __name__ == "__main__":
    main()
