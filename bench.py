#!/usr/bin/env python3
"""bench.py — headline benchmark of the ViMoCLIP hot path on MI355X (contract: see the task prompt).

A "step" = one pass of the hot path over one batch of synthetic input on every rank: the CLIP ViT-L/14
frame encoder (BASELINE.json configs[1]) over B*T = 256 synthetic u8 224x224 frames, bf16 MFMA compute,
fp32 accumulate; inputs are resident in HBM before the timed region.  Frames shard across ranks with no
data-path collective (weak scaling: 256 frames per rank per step).

    python bench.py [--gpus N --steps K --warmup W] [--model ViT-L/14 --frames 256 --dtype bf16]

``--gpus N`` with no WORLD_SIZE in the environment starts the N ranks itself: the parent process (which never touches a
GPU) spawns N children of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays
rank 0's JSON line and exits non-zero if any rank fails.  Under ``python -m torch.distributed.run`` (WORLD_SIZE set) it is one
of the ranks.  One rank per GPU, RCCL backend.

Prints ONE JSON line on rank 0.  `roofline` is the MFMA roofline of the dominant kernel family (the
vmc_linear GEMMs): algorithmic FLOPs of every GEMM launch of K instrumented steps divided by the HIP-event
duration of exactly those launches.  `cpu_baseline` times the CPU oracle (oracle/vit.py, fp32 PyTorch) on a
bounded sample of the same workload on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/f16, MI355X_MICROARCH.md "Chip-level parameters"


def vit_flops_per_frame(name: str) -> float:
    """2(N-1)*3p^2*D + L(6ND^2 + 4N^2 D + 2ND^2 + 4NDM) + 2DE."""
    from vimo_clip_amd.synth import VIT_GEOMETRY
    R, p, D, L, H, E = VIT_GEOMETRY[name]
    N = (R // p) ** 2 + 1
    M = 4 * D
    return 2 * (N - 1) * 3 * p * p * D + L * (6 * N * D * D + 4 * N * N * D + 2 * N * D * D + 4 * N * D * M) + 2 * D * E


def _time_cuda(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def spawn_ranks(n: int, argv) -> int:
    """Parent of an N-rank run: no torch.cuda / HIP call happens in this process.  Children get the torchrun environment."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = list(procs)
    while live:                                      # a rank that dies must not leave the others waiting in a rendezvous
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                deadline = time.time() + 20          # the rest get a moment to fail by themselves, then are ended
                while live and time.time() < deadline:
                    live = [q for q in live if q.poll() is None]
                    time.sleep(0.2)
                for q in live:
                    q.kill()                         # exact PIDs of children this process started
                for q in live:
                    q.wait()
                live = []
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = chunks[0] if chunks else ""
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    return int(rc != 0)


def all_ranks_ok(ok: bool, dev, world: int) -> bool:
    """Agree across ranks before entering a leg's collectives: one rank's failed set-up must not leave the others hanging."""
    if world == 1:
        return ok
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def max_over_ranks(x: float, dev, world: int) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _time_step_dist(step, iters, dev, world, warmup=3):
    """barrier + synchronize on both sides of `iters` steps, MAX over ranks (the bench contract, for the secondary legs too)."""
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return max_over_ranks((time.perf_counter() - t0) / iters, dev, world)


def student_train_leg(dev, rank, world, cdt, clips=32, T=16, name="ViT-B/32", iters=8):
    """MoCLIP student distillation step (BASELINE.json configs[2]; train.py:89-107): per GPU 32 clips x 16 flow frames 224^2 u8,
    forward + cosine distillation + BCE + backward with the gradient all-reduce (RCCL, 48 MB buckets issued during the
    backward) + fused Adam; weak scaling (per-GPU batch fixed)."""
    from vimo_clip_amd import synth
    from vimo_clip_amd.losses import classification_loss, distillation_loss
    from vimo_clip_amd.models import FlowStudentModel
    from vimo_clip_amd.optim import FusedAdam, GradArena
    from vimo_clip_amd.parallel import GradientAllReducer, broadcast_parameters

    ok, err = True, ""
    try:
        E = synth.VIT_GEOMETRY[name][5]
        m = FlowStudentModel(name, device=str(dev), num_classes=140, compute_dtype=cdt).train()
        m.load_state_dict(synth.student_state_dict(name, 3, zero_fc2=True), strict=True)
        arena = GradArena(m.parameters())
        vids = synth.randint_u8(3 + rank, "vids", (clips, T, 3, 224, 224)).to(dev)
        teacher = synth.normal(3 + rank, "teacher", (clips, T + 1, E)).to(dev)
        labels = synth.multi_hot_labels(3 + rank, "labels", clips, 140).to(dev)
    except Exception as e:      # noqa: BLE001
        ok, err = False, f"{type(e).__name__}: {e}"
    if not all_ranks_ok(ok, dev, world):
        return {"error": err or "set-up failed on another rank"}
    broadcast_parameters(arena.flat_param)
    opt = FusedAdam(arena, lr=1e-3)
    red = GradientAllReducer(arena.flat_grad).attach(arena)

    def step():
        emb, emb_d, logits = m(vids)
        loss = distillation_loss(emb_d, teacher[:, :-1, :], mode="cosine") + classification_loss(logits, labels, positive_weight=9)
        loss.backward()
        opt.step(grad_scale=red.all_reduce())

    t = _time_step_dist(step, iters, dev, world, warmup=2)
    red.detach()                                     # the hook list is global: this leg's reducer must not see the next leg's backward
    frames = clips * T
    fwd = vit_flops_per_frame(name)
    from vimo_clip_amd.synth import VIT_GEOMETRY
    R, p, D = VIT_GEOMETRY[name][0], VIT_GEOMETRY[name][1], VIT_GEOMETRY[name][2]
    step_flops = 3 * fwd - 2 * ((R // p) ** 2) * 3 * p * p * D        # no input dgrad of the patch GEMM
    return {"model": name, "clips_per_gpu": clips, "frames_per_clip": T, "ms_per_step": round(1e3 * t, 3),
            "frames_per_s": round(frames * world / t, 1), "grad_allreduce_bytes": arena.numel * 4,
            "mfma_frac": round(frames / t * step_flops / (MFMA_PEAK_TFLOPS * 1e12), 4),
            "note": "fwd + bwd + overlapped gradient all-reduce + fused Adam; whole job frames/s over all ranks"}


def tfam_hbm_bytes(B, T=16, Tk=16, D=768, ff=2048, L=4, C=140, e_w=2):
    """BYTES_fwd(B) = P_used*e_w + B*(T_r+T_f)*D*e_a + B*(T_r+T_f) + B*C*4 (cross mode: every parameter but
    the unused projection_layer; fp32 tokens in, u8 masks in, fp32 logits out)."""
    p_layer = 2 * (3 * D * D + 3 * D) + 2 * (D * D + D) + (ff * D + ff) + (D * ff + D) + 6 * D
    p_used = L * p_layer + 2 * D + (D // 2) * D + D // 2 + C * (D // 2) + C
    return p_used * e_w + B * (T + Tk) * D * 4 + B * (T + Tk) + B * C * 4, p_used


def tfam_forward_block(dev, rank, cdt, batches=(8, 16, 64), iters=200):
    """TFAM eval forward at the reference's small batches (BASELINE.json configs[3]; north_star: 'TFAM fusion step on 16x768
    tokens at >= 70 % HBM roofline'): the fused launch chain (vmc_tfam_forward) captured in ONE hipGraph, timed with HIP
    events on the replay stream, (a) back to back = weights MALL-resident (63.7 MB < 256 MiB Infinity Cache) and (b) with
    512 MiB streamed through another buffer between replays = weights evicted to HBM (SURVEY.md 8d protocol).  Below B ~ 20
    the bound is HBM (weight stream); above it the MFMA fraction is the one to read."""
    from vimo_clip_amd import synth
    from vimo_clip_amd.graphs import GraphedCallable
    from vimo_clip_amd.TFAM.models import AMO_CLIP

    m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0,
                 device=dev, compute_dtype=cdt).to(dev).eval()
    m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
    evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    flops_clip = 1.0136e9
    out = {}
    for B in batches:
        rgb = synth.normal(10 + rank, f"rgb{B}", (B, 16, 768)).to(dev)
        mot = synth.normal(10 + rank, f"mot{B}", (B, 16, 768)).to(dev)
        mk = torch.ones(B, 16, dtype=torch.bool, device=dev)

        def fwd(r, f, a, b):
            with torch.no_grad():
                return m(r, f, mask_rgb=a, mask_flow=b)
        row = {}
        for label, fused in (("fused_chain", True), ("per_op", False)):
            m.fused_inference = fused
            g = GraphedCallable(fwd, rgb, mot, mk, mk)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            t_res = e0.elapsed_time(e1) * 1e-3 / iters
            ts = []
            for _ in range(max(10, iters // 10)):
                evict.add_(1)                                  # 512 MiB read + 512 MiB written: the weights leave L2 and MALL
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                g.replay()
                e.record()
                ts.append((s, e))
            torch.cuda.synchronize()
            cold = sorted(s.elapsed_time(e) * 1e-3 for s, e in ts)
            t_cold = cold[len(cold) // 2]
            nbytes, _ = tfam_hbm_bytes(B)
            row[label] = {"us_mall_resident": round(t_res * 1e6, 2), "us_evicted": round(t_cold * 1e6, 2),
                          "clips_per_s_mall_resident": round(B / t_res, 1), "clips_per_s_evicted": round(B / t_cold, 1),
                          "hbm_frac_mall_resident": round(nbytes / t_res / 8e12, 4), "hbm_frac_evicted": round(nbytes / t_cold / 8e12, 4),
                          "mfma_frac": round(B / t_res * flops_clip / (MFMA_PEAK_TFLOPS * 1e12), 4)}
            del g
        m.fused_inference = True
        if B == 8:
            # throughput with several independent batches in flight (an evaluation loop: batches do not depend on each other):
            # one graph + one scratch slot per stream, replayed round-robin; a launch's fixed ~4 us overlaps the other streams' work
            for nfl in (2, 4, 8):
                streams = [torch.cuda.Stream() for _ in range(nfl)]
                graphs = []
                for i, st in enumerate(streams):
                    m.fused_slot = i
                    with torch.cuda.stream(st):
                        graphs.append(GraphedCallable(fwd, rgb, mot, mk, mk))
                m.fused_slot = 0
                torch.cuda.synchronize()
                reps = max(20, iters // nfl)
                t0 = time.perf_counter()
                for _ in range(reps):
                    for st, g in zip(streams, graphs):
                        with torch.cuda.stream(st):
                            g.replay()
                torch.cuda.synchronize()
                t_in = (time.perf_counter() - t0) / (reps * nfl)
                row[f"fused_chain_{nfl}_in_flight"] = {"us_per_forward": round(t_in * 1e6, 2), "clips_per_s": round(B / t_in, 1),
                                                       "hbm_frac_per_forward": round(tfam_hbm_bytes(B)[0] / t_in / 8e12, 4),
                                                       "note": "weights served by L2 / Infinity Cache across the concurrent forwards; hipGraph replays on "
                                                               "different streams do not scale past two on this runtime (four with GPU_MAX_HW_QUEUES=8, "
                                                               "collapse at eight): profiles/README.md round 3"}
                del graphs
        row["bytes_fwd"] = tfam_hbm_bytes(B)[0]
        out[f"B{B}"] = row
    return {"bound": "hbm (B <~ 20) / mfma", "peak": 8000.0, "unit": "GB/s", "launches_per_forward": 1 + 6 * 4 + 3,
            "note": "one hipGraph replay per forward, HIP events on the replay stream; evicted = 512 MiB streamed between replays",
            **out}


def tfam_extras(dev, rank, world, cdt):
    """TFAM (BASELINE.json configs[3]): d_model 768, 8 heads, 4 layers, ff 2048, 16x768 RGB + motion tokens,
    cross-attention.  Large-batch forward clips/s (MFMA regime; single GPU only) and the full train step (fwd + bwd +
    gradient all-reduce over RCCL when world > 1 + fused AdamW) with the HBM roofline of the AdamW kernel (28 B/parameter)."""
    from vimo_clip_amd import synth
    from vimo_clip_amd.losses import bce_with_logits_loss
    from vimo_clip_amd.optim import FusedAdam, GradArena
    from vimo_clip_amd.parallel import GradientAllReducer, broadcast_parameters
    from vimo_clip_amd.TFAM.models import AMO_CLIP

    out, ok, err = {}, True, ""
    flops_clip = 1.0136e9
    B = 512
    try:
        # dropout 0.1 / 0.1 as the reference's cfg_AK/config_default.yaml:31-32 (active in the train legs only)
        m = AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, num_classes=140, dropout=0.1, mlp_dropout=0.1,
                     device=dev, compute_dtype=cdt).to(dev)
        m.load_state_dict(synth.tfam_state_dict(768, 8, 4, 2048, 140, 4), strict=True)
        if world == 1:
            for Bf in (512, 4096):
                rgb = synth.normal(10 + rank, f"rgb{Bf}", (Bf, 16, 768)).to(dev)
                mot = synth.normal(10 + rank, f"mot{Bf}", (Bf, 16, 768)).to(dev)
                mk = torch.ones(Bf, 16, dtype=torch.bool, device=dev)
                m.eval()
                with torch.no_grad():
                    t = _time_cuda(lambda: m(rgb, mot, mask_rgb=mk, mask_flow=mk), 20 if Bf <= 512 else 5)
                out[f"tfam_fwd_clips_per_s_B{Bf}"] = round(Bf / t, 1)
                out[f"tfam_fwd_mfma_frac_B{Bf}"] = round(Bf / t * flops_clip / (MFMA_PEAK_TFLOPS * 1e12), 4)
        # ---- train step, per-GPU batch 512 (weak scaling), AdamW lr 1e-4 wd 0.1 as TFAM/train_and_eval.py:53 ----
        m.train()
        arena = GradArena(m.used_parameters())
        rgb = synth.normal(20 + rank, "rgb_t", (B, 16, 768)).to(dev)
        mot = synth.normal(20 + rank, "mot_t", (B, 16, 768)).to(dev)
        mk = torch.ones(B, 16, dtype=torch.bool, device=dev)
        y = synth.multi_hot_labels(20 + rank, "lab_t", B, 140).to(dev)
    except Exception as e:      # noqa: BLE001
        ok, err = False, f"{type(e).__name__}: {e}"
    if not all_ranks_ok(ok, dev, world):
        return {"error": err or "set-up failed on another rank"}
    broadcast_parameters(arena.flat_param)
    opt = FusedAdam(arena, lr=1e-4, weight_decay=0.1, decoupled=True)
    red = GradientAllReducer(arena.flat_grad).attach(arena)      # bucket all-reduces overlap the backward (N > 1)

    def train_step():
        loss = bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mk), y)
        loss.backward()
        opt.step(grad_scale=red.all_reduce())

    t = _time_step_dist(train_step, 10, dev, world)
    red.detach()
    out["tfam_train_clips_per_s"] = round(B * world / t, 1)          # whole job, per-GPU batch 512 (weak scaling)
    out["tfam_train_per_gpu_batch"] = B
    out["tfam_train_ms_per_step"] = round(1e3 * t, 3)
    out["tfam_grad_allreduce_bytes"] = arena.numel * 4
    # AdamW kernel alone: HIP events on the launch stream, HBM roofline (16 B read + 12 B write per parameter)
    from vimo_clip_amd import optim as _optim
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    refresh, _optim.invalidate_weight_copies = _optim.invalidate_weight_copies, (lambda *a: None)   # the AdamW kernel alone, without
    try:                                                                                         # the 16-bit copy refresh after it
        opt.step()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            opt.step()
        e1.record()
        torch.cuda.synchronize()
    finally:
        _optim.invalidate_weight_copies = refresh
    refresh(opt)
    t_adam = e0.elapsed_time(e1) * 1e-3 / 20
    bytes_adam = arena.numel * 28.0
    out["adamw_roofline"] = {"bound": "hbm", "achieved": round(bytes_adam / t_adam / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                             "frac": round(bytes_adam / t_adam / 8e12, 4), "params": arena.numel,
                             "note": "133 MB of state fits the 256 MiB Infinity Cache: MALL-resident, not an HBM-only figure"}
    # the same step as ModelTrainer(use_graphs=True) runs it under data parallelism: forward + backward as one hipGraph, the bucket
    # exchange (eager), AdamW + copy refresh as a second graph (graphs.GraphedTrainStep(exchange=, opt_fn=)); the eager figure above is
    # bound by the host's launch rate
    try:
        from vimo_clip_amd.graphs import GraphedTrainStep
        from vimo_clip_amd.losses import loss_and_grad
        opt.enable_device_state(base_seed=rank)
        m.use_device_seeds(opt)
        red2 = GradientAllReducer(arena.flat_grad).attach(arena, register=False)

        def fwd_bwd(a, b_, c, d):
            opt.tick()
            o = m(a, b_, mask_rgb=c, mask_flow=c)
            loss, dl = loss_and_grad(bce_with_logits_loss, o, d)
            o.backward(dl)
            return loss

        g2 = GraphedTrainStep(fwd_bwd, opt, exchange=red2.all_reduce, opt_fn=opt.step)
        t2 = _time_step_dist(lambda: g2(rgb, mot, mk, y), 10, dev, world)
        out["tfam_train_ms_per_step_two_graph"] = round(1e3 * t2, 3)
        out["tfam_train_clips_per_s_two_graph"] = round(B * world / t2, 1)
    except Exception as e:      # noqa: BLE001
        out["tfam_train_ms_per_step_two_graph"] = f"error: {type(e).__name__}: {e}"
    if world == 1:
        try:
            out["tfam_train_small_batch"] = tfam_small_batch_train(dev, cdt, m)
        except Exception as e:      # noqa: BLE001
            out["tfam_train_small_batch"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def tfam_small_batch_train(dev, cdt, m, batches=(8, 64, 512), iters=30):
    """The reference's own batch size (TFAM/train_and_eval.py:34 batch_size = 8 per GPU) is launch-bound: ~240 launches per step (and at
    B = 512 the eager step is still bound by the host's launch rate on a slow host: the captured figure is the GPU's).
    Eager step vs ONE hipGraph replay per step (graphs.GraphedTrainStep: step count, lr, bias corrections and dropout seeds in
    device memory, advanced by vmc_train_tick inside the graph).  Single process; wall clock around synchronised loops."""
    from vimo_clip_amd import synth
    from vimo_clip_amd.graphs import GraphedTrainStep
    from vimo_clip_amd.losses import bce_with_logits_loss, loss_and_grad
    from vimo_clip_amd.optim import FusedAdam, GradArena
    m.train()
    opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-4, weight_decay=0.1, decoupled=True)
    res = {}
    data = {}
    for B in batches:
        data[B] = (synth.normal(30, f"rgb{B}", (B, 16, 768)).to(dev), synth.normal(30, f"mot{B}", (B, 16, 768)).to(dev),
                   torch.ones(B, 16, dtype=torch.bool, device=dev), synth.multi_hot_labels(30, f"lab{B}", B, 140).to(dev))

    def eager_step(rgb, mot, mk, y):
        loss = bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mk), y)
        loss.backward()
        opt.step()

    for B in batches:
        res[f"B{B}_eager_ms"] = round(1e3 * _time_cuda(lambda: eager_step(*data[B]), iters), 4)
    opt.enable_device_state(base_seed=0)
    m.use_device_seeds(opt)

    def dev_step(rgb, mot, mk, y):
        opt.tick()                                                     # the step of TFAM/train_and_eval.ModelTrainer._device_state_step
        out = m(rgb, mot, mask_rgb=mk, mask_flow=mk)
        loss, dlogits = loss_and_grad(bce_with_logits_loss, out, y)
        out.backward(dlogits)
        opt.step()
        return loss, out.detach()

    g = GraphedTrainStep(dev_step, opt)
    p_used = tfam_hbm_bytes(8)[1]
    evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for B in batches:
        rgb, mot, mk, y = data[B]
        t = _time_cuda(lambda: g(rgb, mot, mk, y), iters)
        res[f"B{B}_captured_ms"] = round(1e3 * t, 4)
        res[f"B{B}_captured_clips_per_s"] = round(B / t, 1)
        if B > 64:
            continue
        # BYTES_train(B) = P_used * (2 e_w + 4 + 28) + activations: the 16-bit weights read by the forward and by the
        # backward, the fp32 gradient written, AdamW's 16 B read + 12 B written per parameter; tokens in / logits out as BYTES_fwd.
        # HIP events around ONE replay on its stream; "evicted" = 512 MiB streamed through another buffer before each replay.
        nbytes = p_used * (2 * 2 + 4 + 28) + B * 32 * 768 * 4 + B * 32 + B * 140 * 4
        gr = g._graphs[next(iter(k for k in g._graphs if k[0][0][0] == B))]
        ts = []
        for cold in (False, True):
            evs = []
            for _ in range(12):
                if cold:
                    evict.add_(1)
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                gr.replay()
                s1.record()
                evs.append((s0, s1))
                opt.step_count += 1
            torch.cuda.synchronize()
            el = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
            ts.append(el[len(el) // 2])
        res[f"B{B}_roofline"] = {"bound": "hbm", "bytes_train": nbytes, "peak": 8000.0, "unit": "GB/s",
                                 "us_mall_resident": round(ts[0] * 1e6, 1), "us_evicted": round(ts[1] * 1e6, 1),
                                 "achieved_mall_resident": round(nbytes / ts[0] / 1e9, 1), "achieved_evicted": round(nbytes / ts[1] / 1e9, 1),
                                 "hbm_frac_mall_resident": round(nbytes / ts[0] / 8e12, 4), "hbm_frac_evicted": round(nbytes / ts[1] / 8e12, 4)}
    res["launches_per_step_B8"] = "27 forward + 38 backward (fused chains, tfam_train.py) + tick, loss, AdamW, 16-bit copy refresh"
    del evict
    m.use_device_seeds(None)
    return res


def tfam_cpu_baseline(ncpu: int, batches=(8, 64), reps: int = 3) -> dict:
    """the CPU restatement of the TFAM forward (oracle/tfam.py, fp32 PyTorch) timed on the host cores at B = 8 and B = 64,
    the geometry of the `tfam_forward` block (768 / 8 heads / 4 layers / ff 2048 / 140 classes, 16 + 16 tokens).  Checker code, timed
    beside the GPU figures as a reported baseline only."""
    from oracle import tfam as otfam
    from vimo_clip_amd import synth
    torch.set_num_threads(ncpu)
    sd = synth.tfam_state_dict(768, 8, 4, 2048, 140, 4)
    out = {"cores": ncpu, "kind": "port", "unit": "clips/s"}
    with torch.no_grad():
        for B in batches:
            rgb, mot = synth.normal(10, "rgb", (B, 16, 768)), synth.normal(10, "mot", (B, 16, 768))
            mk = torch.ones(B, 16, dtype=torch.bool)
            otfam.amo_clip_forward(sd, rgb, mot, mk, mk, nhead=8)      # warm-up
            t0 = time.perf_counter()
            for _ in range(reps):
                otfam.amo_clip_forward(sd, rgb, mot, mk, mk, nhead=8)
            out[f"B{B}"] = round(B * reps / (time.perf_counter() - t0), 1)
    out["sample"] = f"oracle/tfam.py fp32 PyTorch CPU, {reps} forwards per batch size"
    return out


def selftest_spawn(rank: int, world: int) -> None:
    """CPU rehearsal of the N-rank control flow (tests/test_bench_spawn.py): gloo rendezvous from the spawned environment,
    one all-reduce, rank 0 prints a line in the bench format.  No GPU is touched."""
    import torch.distributed as dist
    if os.environ.get("VMC_SELFTEST_DIE_EARLY_RANK") == str(rank):
        sys.exit(3)                                  # dies before the rendezvous: the other ranks would wait for it
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "selftest", "value": float(t.item()), "n_gpus": world, "ranks_sum": float(t.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if os.environ.get("VMC_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(3)                                  # the parent must report a failing rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--frames", type=int, default=256, help="frames per rank per step (B*T)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--no-extras", action="store_true", help="skip the TFAM measurements")
    ap.add_argument("--no-fuse-add-ln", action="store_true", help="A/B: residual add in the GEMM epilogue + plain LayerNorm")
    ap.add_argument("--selftest-spawn", action="store_true", help="CPU/gloo rehearsal of the --gpus N spawn path (tests)")
    ap.add_argument("--only", default="", choices=["", "tfam"], help="builder shortcut: run one secondary leg alone and print it")
    ap.add_argument("--chunk", type=int, default=0, help="frames per encoder pass inside a step (0 = all frames of the step at once)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's command form `python bench.py --gpus N`: start the N ranks here, before anything touches a GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.selftest_spawn:
        selftest_spawn(rank, world)
        return
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") in production; VMC_BENCH_BACKEND=gloo rehearses the N>1 control flow with several ranks
        # sharing one GPU (the 1-GPU development boxes), where RCCL refuses duplicate devices.
        backend = os.environ.get("VMC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            import datetime
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=10))
        else:
            local_rank %= max(1, torch.cuda.device_count())
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vimo_clip_amd import ops, synth
    from vimo_clip_amd.clip_vit import VisionTransformer

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    if args.only == "tfam":
        print(json.dumps(tfam_forward_block(dev, rank, cdt)), flush=True)
        print(json.dumps(tfam_extras(dev, rank, world, cdt)), flush=True)
        return
    model = VisionTransformer.from_name(args.model, compute_dtype=cdt).to(dev).eval()
    sd = synth.vit_state_dict(args.model, seed=2)
    model.load_state_dict(sd, strict=True)
    model.frame_chunk = args.chunk or args.frames
    if args.no_fuse_add_ln:
        model.fuse_add_ln = False
    R = model.input_resolution
    frames = synth.randint_u8(1 + rank, "frames", (args.frames, 3, R, R)).to(dev)   # random, never zeros (DVFS)

    def step():
        return model.encode_frames_u8(frames)

    # one-time costs outside the warm-up contract: code-object load, 16-bit weight copies, allocator growth, clock ramp
    for _ in range(2):
        out = step()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    total_frames = args.frames * args.steps * world
    fps = total_frames / elapsed
    flops_frame = vit_flops_per_frame(args.model)

    result = {
        "metric": "CLIP frame-embeddings/sec (whole job; TFAM fused-clips/sec and mAP parity reported in DESIGN.md/tests)",
        "value": round(fps, 2), "unit": "frame-embeddings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"CLIP {args.model} frame encoder, B*T={args.frames} u8 224x224 frames per GPU per step "
                               f"(BASELINE.json configs[1]), random-init weights, fp32 residual stream",
                   "frames_per_gpu_per_step": args.frames, "parallelism": f"frame-sharded x{world}, no collective"},
        "per_gpu_value": round(fps / world, 2),
        "end_to_end_mfma_frac": round(fps / world * flops_frame / (MFMA_PEAK_TFLOPS * 1e12), 4),
    }
    # end_to_end_mfma_frac is ALGORITHMIC: frames/s x the full forward of SURVEY.md 8d.  The last residual block runs on the class
    # rows only (bit-identical output: only x[:, 0] reaches ln_post), so its q projection, attention for N-1 queries, out_proj and
    # MLP are not executed; K|V of all tokens still are.  roofline.frac below counts executed launches only.
    try:
        R_, p_, D_, L_, H_, E_ = synth.VIT_GEOMETRY[args.model]
        N_ = (R_ // p_) ** 2 + 1
        layer = 6 * N_ * D_ * D_ + 4 * N_ * N_ * D_ + 2 * N_ * D_ * D_ + 16 * N_ * D_ * D_
        last_exec = 4 * N_ * D_ * D_ + 2 * D_ * D_ + 4 * N_ * D_ + 2 * D_ * D_ + 16 * D_ * D_      # K|V of all tokens + the class row's share
        result["end_to_end_mfma_frac_note"] = "algorithmic FLOPs (full forward, SURVEY 8d); executed_flop_frac = share of them the launches perform"
        result["executed_flop_frac"] = round(1.0 - (layer - last_exec) / flops_frame, 4)
    except Exception:      # noqa: BLE001
        pass

    # The parity claim "within 1e-3 fp16" of BASELINE.json's north_star belongs to the f16 compute type; the headline
    # runs in bf16 as configs[1] names it.  The same workload in f16, a few steps, beside it (one GPU only).
    y16_sample = None
    if world == 1 and args.dtype == "bf16" and not args.no_extras:
        try:
            m16 = VisionTransformer.from_name(args.model, compute_dtype=torch.float16).to(dev).eval()
            m16.load_state_dict(sd, strict=True)
            m16.frame_chunk = model.frame_chunk
            t16 = _time_cuda(lambda: m16.encode_frames_u8(frames), 3, warmup=2)
            y16_sample = m16.encode_frames_u8(frames)[:8].float().cpu()
            result["f16_same_workload"] = {"frame_embeddings_per_s": round(args.frames / t16, 1), "ms_per_step": round(1e3 * t16, 3),
                                           "note": "compute dtype f16: the type the 1e-3 parity bound is asserted for (tests/test_gpu_encoder.py)"}
            del m16
        except Exception as e:      # noqa: BLE001
            result["f16_same_workload"] = {"error": f"{type(e).__name__}: {e}"}

    y_sample = out[:8].float().cpu()       # the timed dtype's embeddings of the first frames (parity block below)
    # secondary measurements never take the headline down with them: a failure is reported inside the JSON line, and every
    # leg with collectives agrees on its set-up across ranks before it enters them (all_ranks_ok)
    if not args.no_extras:
        def guarded(leg):
            if world > 1:
                return leg()                        # a rank that dies takes the job down (the parent reports it); no silent hang
            try:
                return leg()
            except Exception as e:      # noqa: BLE001
                return {"error": f"{type(e).__name__}: {e}"}
        result["student_train"] = guarded(lambda: student_train_leg(dev, rank, world, cdt))        # BASELINE.json configs[2]
        result["extras"] = guarded(lambda: tfam_extras(dev, rank, world, cdt))                     # configs[3], train step
        if world == 1:
            try:
                result["tfam_forward"] = tfam_forward_block(dev, rank, cdt)        # configs[3], small-batch forward vs HBM roofline
            except Exception as e:      # noqa: BLE001
                result["tfam_forward"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        try:
            # ---- roofline of the GEMM family: instrument every vmc_linear launch with HIP events --------
            events = []
            orig_linear = ops.linear

            def timed_linear(a, w16, *pa, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig_linear(a, w16, *pa, **kw)
                e.record()
                events.append((s, e, 2.0 * a.shape[0] * w16.shape[0] * w16.shape[1], a.shape[0] * w16.shape[0] >= 192 * 65536))
                return r

            # the two other kernels of a ViT layer, against their own bounds: add+LayerNorm (HBM: x fp32 read + write, 16-bit
            # branch in, 16-bit h out = 12 B / element) and attention (HBM: q/k/v read + o write = 8 B per token-channel of D)
            sec_events = {"add_ln": [], "attention": []}
            orig_addln, orig_attn = ops.add_layernorm_, ops.attention_vit

            def timed_addln(x32, b16, gm, bt, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig_addln(x32, b16, gm, bt, **kw)
                e.record()
                rows = kw.get("rows") or x32.numel() // gm.shape[0]
                sec_events["add_ln"].append((s, e, 12.0 * rows * gm.shape[0], rows * gm.shape[0] >= (1 << 24)))
                return r

            def timed_attn(qkv, F_, N_, H_, *pa, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig_attn(qkv, F_, N_, H_, *pa, **kw)
                e.record()
                sec_events["attention"].append((s, e, 8.0 * F_ * N_ * H_ * 64, True))
                return r

            ops.linear, ops.add_layernorm_, ops.attention_vit = timed_linear, timed_addln, timed_attn
            try:
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
            finally:
                ops.linear, ops.add_layernorm_, ops.attention_vit = orig_linear, orig_addln, orig_attn
            sec = {}
            for name, evs in sec_events.items():
                sel = [(s.elapsed_time(e) * 1e-3, b) for s, e, b, big_ in evs if big_]
                if sel:
                    t_, b_ = sum(t for t, _ in sel), sum(b for _, b in sel)
                    sec[name] = {"bound": "hbm", "achieved": round(b_ / t_ / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(b_ / t_ / 8e12, 4), "launches": len(sel), "avg_launch_ms": round(1e3 * t_ / len(sel), 4)}
            try:    # HBM bytes per launch of the two kernels from the committed rocprofv3 --pmc passes (tools/_prof_r02b.sh)
                with open(os.path.join(ROOT, "profiles", "r02_secondary_traffic.json")) as f:
                    sec_traffic = json.load(f)
                for name in sec:
                    sec[name]["traffic"] = sec_traffic.get(name, {}).get("hbm_bytes_per_launch")
            except Exception:
                pass
            result["secondary_rooflines"] = sec
            try:    # the clock THIS box holds under an MFMA load: kernels that are latency / issue bound (ViT attention: 160 us on one
                    # box, 212 us on another in round 2) move with it, HBM-bound ones (add+LayerNorm) do not
                from vimo_clip_amd._lib import check as _check, lib as _lib, ptr as _ptr, stream as _stream
                stamps = torch.zeros(2 * 256, dtype=torch.int64, device=dev)
                for _ in range(3):      # ~3 x 0.4 ms of back-to-back MFMAs; the last launch is read
                    _check(_lib.vmc_clock_probe(_ptr(stamps), 256, 3000, 0, _stream()), "clock_probe")
                torch.cuda.synchronize()
                st = stamps.view(256, 2).cpu().double()
                mhz = (st[:, 0] / st[:, 1].clamp(min=1) * 100.0).sort().values
                result["clock_probe"] = {"mfma_loop_mhz_median": round(float(mhz[128]), 1), "mhz_min": round(float(mhz[0]), 1),
                                         "mhz_max": round(float(mhz[-1]), 1),
                                         "note": "s_memtime / s_memrealtime around 192k bf16 MFMAs per wave, one workgroup per CU (vmc_clock_probe)"}
            except Exception as e:      # noqa: BLE001
                result["clock_probe"] = {"error": f"{type(e).__name__}: {e}"}
            big = [(s.elapsed_time(e) * 1e-3, f) for s, e, f, is_big in events if is_big]
            t_big = sum(t for t, _ in big)
            f_big = sum(f for _, f in big)
            achieved = f_big / t_big / 1e12 if big else 0.0
            traffic = None
            try:    # HBM bytes per launch of the c_fc shape from the committed rocprofv3 --pmc passes (cannot be collected in-process)
                with open(os.path.join(ROOT, "profiles", "r03_gemm8_traffic.json")) as f:
                    traffic = json.load(f)["hbm_bytes_per_launch"]
            except Exception:
                pass
            result["roofline"] = {
                "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_note": "PMC FETCH_SIZE x2 + WRITE_SIZE of one [65792,1024]x[1024,4096] launch (profiles/r03_gemm8_traffic.json); algorithmic bytes 6.8e8",
                "kernel": "gemm8p_kernel<BF16,*> (vmc_linear: persistent walk of 8-phase 256x256x64 tiles; gemm8_kernel for the few non-eligible launches), all large-GEMM launches of the step", "launches": len(big),
                "avg_launch_ms": round(1e3 * t_big / max(1, len(big)), 4),
                "note": "algorithmic FLOPs = 2*M*N*K (K incl. zero padding 588->640 of the patch GEMM) per launch",
            }

        except Exception as e:      # noqa: BLE001
            result["roofline"] = {"error": f"{type(e).__name__}: {e}"}

        try:
            if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only
                from oracle import vit as ovit
                # the GPU box gives one GPU a 16-core CPU share; more threads than that only oversubscribes
                ncpu = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
                torch.set_num_threads(ncpu)
                nf = args.cpu_frames
                pix = ovit.normalize_u8(frames[:nf].cpu())
                H = model.heads
                with torch.no_grad():
                    ovit.vit_forward(sd, pix[:1], H)                      # warm-up
                    t1 = time.perf_counter()
                    ref = ovit.vit_forward(sd, pix, H)
                    dt_cpu = time.perf_counter() - t1
                result["cpu_baseline"] = {"value": round(nf / dt_cpu, 3), "unit": "frame-embeddings/s", "cores": ncpu, "kind": "port",
                                          "sample": f"oracle/vit.py fp32 PyTorch CPU, {nf} frames of the same workload, 1 pass"}
                # parity of THIS run against the oracle's embeddings of the same frames (checker use of the oracle): the figures
                # behind "CLIP embeddings within 1e-3 fp16" (BASELINE.json north_star; tolerance discussion in DESIGN.md 5)
                np_ = min(nf, 8)
                scale = float(ref[:np_].abs().max())

                def _par(y):
                    d = float((y[:np_] - ref[:np_]).abs().max())
                    return {"max_abs_err": round(d, 6), "rel_to_max_ref": round(d / max(scale, 1e-12), 6)}
                par = {"frames": np_, "ref_abs_max": round(scale, 4), args.dtype: _par(y_sample),
                       "note": "max |y - oracle| over the first frames of the timed workload; rel = / max|oracle|; "
                               "tests assert f16 <= 1e-3 * max(1, |ref|max), bf16 <= 8e-3 * max(1, |ref|max)"}
                if y16_sample is not None:
                    par["f16"] = _par(y16_sample)
                result["parity"] = par
                if not args.no_extras:
                    try:
                        result["cpu_baseline"]["tfam_forward"] = tfam_cpu_baseline(ncpu)
                    except Exception as e:      # noqa: BLE001
                        result["cpu_baseline"]["tfam_forward"] = {"error": f"{type(e).__name__}: {e}"}
        except Exception as e:      # noqa: BLE001
            result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
