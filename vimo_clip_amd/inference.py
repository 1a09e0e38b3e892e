"""Student embedding export — drop-in for the reference's inference.py (whole video per group) and
inference_frame_diff.py:183-410 (chunked streaming into extendable datasets, resumable).

The reference decodes a video on the host, runs ``to_pil_image -> CLIP preprocess`` per frame and feeds chunks of 32
frames to the student (inference_frame_diff.py:183-232, :270-299).  Here the decoded u8 frames go straight to the GPU:
resize + crop + normalise + ViT + residual MLP are HIP kernels inside ``model.forward``, so the chunk is sized for the
GPU (default 256 frames) and the only host work left is decode and the HDF5 append.

File layout and control flow are the reference's: one group per ``video_id`` (basename without extension), an
``embeddings`` dataset created on the first chunk as ``(0,E)`` / ``maxshape (None,E)`` / chunks ``(min(chunk,1024), E)``
float32 and grown by ``resize`` + slice assignment, ``(0,0)`` dataset when no frame could be decoded, ``error`` /
``skipped_low_ram`` attributes on failures, ``--resume`` skipping groups that already exist.  The reference flushes after
every chunk; h5lite commits are whole-tree rewrites of the dirty path, so the flush is rate limited (``flush_interval_s``,
0 = every chunk as the reference) — a crash loses at most that interval, which ``resume`` recomputes.
"""
from __future__ import annotations

import os
import time
import warnings

import numpy as np
import torch

from . import h5lite as h5py
from .extract_embeddings import open_video


class LowMemoryError(RuntimeError):
    """inference_frame_diff.py:60: raised by the host-RAM guard."""


def available_gb() -> float:
    """inference_frame_diff.py:33-57: psutil when present, else MemAvailable of /proc/meminfo, else "plenty"."""
    try:
        import psutil
        return psutil.virtual_memory().available / (1024 ** 3)
    except ImportError:
        pass
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return float(line.split()[1]) / (1024 ** 2)
    except OSError:
        pass
    return 1e9


def memory_guard(min_free_gb: float):
    """inference_frame_diff.py:60-62."""
    if available_gb() < min_free_gb:
        raise LowMemoryError(f"Low RAM: {available_gb():.2f} GB available")


def iter_frame_chunks(video_path, chunk_size=256, min_free_gb=0.0, frame_source=None):
    """Yield ``[n<=chunk_size, 3, H, W]`` u8 tensors in decode order (inference_frame_diff.py:183-232 without the
    per-frame PIL transform: that arithmetic runs on the GPU inside the model)."""
    vr = (frame_source or open_video)(video_path)
    total = len(vr)
    for lo in range(0, total, chunk_size):
        memory_guard(min_free_gb)
        hi = min(total, lo + chunk_size)
        frames = vr.get_batch(np.arange(lo, hi))          # [n,H,W,3] u8
        if frames.ndim != 4 or frames.shape[-1] != 3:
            raise ValueError(f"unexpected frame block shape {tuple(frames.shape)}")
        yield frames.permute(0, 3, 1, 2)


class _Committer:
    def __init__(self, h5f, interval_s):
        self.h5f, self.interval, self.last = h5f, interval_s, time.monotonic()

    def __call__(self, force=False):
        now = time.monotonic()
        if force or self.interval <= 0 or now - self.last >= self.interval:
            self.h5f.flush()
            self.last = now


@torch.no_grad()
def process_and_write_video_incremental(video_path, model, h5f, chunk_size=256, min_free_gb=0.0, compression="lzf",
                                        frame_source=None, commit=None):
    """inference_frame_diff.py:235-312.  Returns the final ``(T, D)`` shape."""
    video_id = os.path.splitext(os.path.basename(video_path))[0]
    group = h5f.require_group(video_id)
    if "embeddings" in group:
        return tuple(group["embeddings"].shape)
    commit = commit or (lambda force=False: h5f.flush())
    dset, embed_dim, total = None, None, 0
    for frames in iter_frame_chunks(video_path, chunk_size, min_free_gb, frame_source):
        out, _, _ = model(frames.unsqueeze(0))                                  # (1, n, D)
        emb = out.squeeze(0).float().cpu().numpy().astype("float32")
        if embed_dim is None:
            embed_dim = emb.shape[1]
            dset = group.create_dataset("embeddings", shape=(0, embed_dim), maxshape=(None, embed_dim),
                                        chunks=(max(1, min(chunk_size, 1024)), embed_dim), dtype="float32",
                                        compression=compression if compression else None)
        memory_guard(min_free_gb)
        old_n = dset.shape[0]
        new_n = old_n + emb.shape[0]
        dset.resize((new_n, embed_dim))
        dset[old_n:new_n, :] = emb
        total = new_n
        commit()
    if dset is None:
        group.create_dataset("embeddings", shape=(0, 0), maxshape=(None, 0), dtype="float32")
        return (0, 0)
    return (total, embed_dim)


@torch.no_grad()
def export_embeddings(video_paths, model, output_h5_path, resume=False, overwrite=False, chunk_size=256, min_free_gb=0.0,
                      compression="lzf", frame_source=None, flush_interval_s=5.0, streaming=True):
    """Main loop of inference_frame_diff.py:318-410 (``streaming=True``) or inference.py:94-114 (``streaming=False``: the
    whole video in one forward, plain contiguous ``embeddings`` dataset, file rewritten from scratch).
    Returns ``{"processed", "skipped_existing", "skipped_low_ram", "errors"}``."""
    model.eval()
    out_dir = os.path.dirname(output_h5_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    stats = {"processed": 0, "skipped_existing": 0, "skipped_low_ram": 0, "errors": 0}
    if not streaming:
        if os.path.exists(output_h5_path):
            os.remove(output_h5_path)                                           # inference.py:73-74
        with h5py.File(output_h5_path, "w") as h5f:
            for video_path in video_paths:
                video_path = video_path.strip()
                video_id = os.path.splitext(os.path.basename(video_path))[0]
                vr = (frame_source or open_video)(video_path)
                frames = vr.get_batch(np.arange(len(vr))).permute(0, 3, 1, 2)
                emb, _, _ = model(frames.unsqueeze(0))
                h5f.create_group(video_id).create_dataset("embeddings", data=emb.squeeze(0).float().cpu().numpy())
                stats["processed"] += 1
        return stats
    if os.path.exists(output_h5_path) and not resume and overwrite:
        os.remove(output_h5_path)
    with h5py.File(output_h5_path, "a") as h5f:
        commit = _Committer(h5f, flush_interval_s)
        for video_path in video_paths:
            video_path = video_path.strip()
            video_id = os.path.splitext(os.path.basename(video_path))[0]
            if resume and video_id in h5f:
                stats["skipped_existing"] += 1
                continue
            try:
                process_and_write_video_incremental(video_path, model, h5f, chunk_size, min_free_gb, compression, frame_source, commit)
                stats["processed"] += 1
            except LowMemoryError as e:
                warnings.warn(f"Skipping {video_id} due to low RAM: {e}")
                h5f.require_group(video_id).attrs["skipped_low_ram"] = True
                commit(force=True)
                stats["skipped_low_ram"] += 1
            except Exception as e:   # the reference records the message and moves on (:399-407)
                warnings.warn(f"Error on {video_id}: {e}. Moving on.")
                h5f.require_group(video_id).attrs["error"] = str(e)
                commit(force=True)
                stats["errors"] += 1
    return stats


class FrameDiffVideoDataset:
    """inference_frame_diff.py:78-93 / inference.py:14-29: every file under the directory (recursive ``**/*.*``), sorted."""

    def __init__(self, frame_diff_videos_dir):
        import glob
        pattern = os.path.join(frame_diff_videos_dir, "**", "*.*")
        self.video_paths = sorted(p for p in glob.iglob(pattern, recursive=True) if os.path.isfile(p))

    def __len__(self):
        return len(self.video_paths)

    def __getitem__(self, idx):
        return self.video_paths[idx]


FlowVideoDataset = FrameDiffVideoDataset
