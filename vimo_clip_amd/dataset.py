"""Student dataset — drop-in for the reference's dataset.py (HDF5VideoDataset + collate_fn).

The integer contracts (segment table, pad-by-repeat, flow-window clamping; dataset.py:35-57,77-91,103-127) are
pure functions here so they can be tested bit-exactly without HDF5 or a video codec.  Video decode and HDF5
I/O stay host-side (h5py / torchvision.io are imported lazily and are absent offline: SURVEY.md §8c); that
wall-clock is outside the accelerated path.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset


def build_segments(lengths: dict, sequence_length: int) -> list:
    """dataset.py:35-57.  lengths: {video_id: T}.  Non-overlapping windows; a shorter tail becomes one
    padded segment; T == 0 videos are skipped."""
    segments = []
    for k, T in lengths.items():
        if T == 0:
            continue
        full, tail = divmod(T, sequence_length)
        segments.extend((k, i * sequence_length, sequence_length) for i in range(full))
        if tail:
            segments.append((k, full * sequence_length, tail))
    return segments


def slice_rgb_segment(embeddings: torch.Tensor, start_idx: int, seg_len: int, sequence_length: int) -> torch.Tensor:
    """dataset.py:77-91: rows [start, start+seg_len) padded to sequence_length by repeating the last row."""
    rgb = embeddings[start_idx:start_idx + seg_len]
    leftover = sequence_length - seg_len
    if leftover > 0:
        pad = rgb[-1:].repeat(leftover, 1) if seg_len > 0 else torch.zeros((leftover, embeddings.shape[1]))
        rgb = torch.cat([rgb, pad], dim=0)
    return rgb


def flow_window(start_idx: int, seg_len: int, sequence_length: int, t_flow: int):
    """dataset.py:101-113 -> (flow_start, flow_end, flow_seg_len)."""
    flow_seg_len = seg_len - 1
    if sequence_length - seg_len > 0:
        flow_seg_len = sequence_length - 1
    flow_start = min(start_idx, max(t_flow - 1, 0))
    flow_end = min(start_idx + flow_seg_len, t_flow)
    return flow_start, flow_end, flow_seg_len


def slice_flow_segment(flow_video: torch.Tensor, start_idx: int, seg_len: int, sequence_length: int) -> torch.Tensor:
    """dataset.py:101-127: flow frames [T_flow,C,H,W] -> [flow_seg_len,C,H,W], padded by repeating the last
    frame (zeros if the window is empty)."""
    t_flow = flow_video.shape[0]
    fs, fe, n = flow_window(start_idx, seg_len, sequence_length, t_flow)
    seq = flow_video[fs:fe]
    needed = n - seq.shape[0]
    if needed > 0:
        if seq.shape[0] > 0:
            pad = seq[-1:].repeat(needed, 1, 1, 1)
        else:
            pad = torch.zeros((needed,) + tuple(flow_video.shape[1:]))
        seq = torch.cat([seq, pad], dim=0)
    return seq


def read_video_frames(path: str) -> torch.Tensor:
    """[T,H,W,3] u8 frames of one flow / frame-diff video (dataset.py:95 ``io.read_video(..., pts_unit="sec")``).
    Decoding is host-side I/O outside the hot path: torchvision.io when it is installed; offline, a ``.npy`` file of the
    decoded frames (same stem, or the path itself) is read instead.  Neither present -> the ImportError is raised."""
    stem = os.path.splitext(path)[0]
    for cand in (path if path.endswith(".npy") else None, path + ".npy", stem + ".npy"):
        if cand and os.path.exists(cand):
            arr = np.load(cand, mmap_mode="r")
            if arr.ndim != 4 or arr.shape[-1] != 3 or arr.dtype != np.uint8:
                raise ValueError(f"{cand}: expected [T,H,W,3] uint8 frames, got {arr.shape} {arr.dtype}")
            return torch.from_numpy(np.array(arr))
    import torchvision.io as io
    return io.read_video(path, pts_unit="sec")[0]


class HDF5VideoDataset(Dataset):
    """Same constructor, item keys and semantics as the reference class (dataset.py:8-134)."""

    def __init__(self, clip_embeddings_dir, flow_videos_dir, sequence_length=2, transform=None):
        super().__init__()
        from . import h5lite as h5py          # native reader of the reference's HDF5 layout (h5py itself is not needed)
        self.hdf5_path, self.flow_videos_dir = clip_embeddings_dir, flow_videos_dir
        self.sequence_length, self.transform = sequence_length, transform
        with h5py.File(self.hdf5_path, "r") as f:
            lengths = {k: f[k]["embeddings"].shape[0] for k in f.keys()}
        self.segments = build_segments(lengths, sequence_length)

    def __len__(self):
        return len(self.segments)

    def __getitem__(self, idx):
        from . import h5lite as h5py
        video_id, start_idx, seg_len = self.segments[idx]
        with h5py.File(self.hdf5_path, "r") as f:
            group = f[video_id]
            embeddings = torch.from_numpy(group["embeddings"][:])
            labels = torch.from_numpy(group["labels"][:])
        rgb_seq = slice_rgb_segment(embeddings, start_idx, seg_len, self.sequence_length)
        if self.transform:
            rgb_seq = self.transform(rgb_seq)
        flow_video = read_video_frames(os.path.join(self.flow_videos_dir, video_id)).permute(0, 3, 1, 2)
        flow_seq = slice_flow_segment(flow_video, start_idx, seg_len, self.sequence_length)
        return {"video_id": video_id, "rgb_emb": rgb_seq, "flow_frames": flow_seq, "labels": labels}


class SyntheticSegmentDataset(Dataset):
    """Synthetic stand-in with the same item layout (BASELINE.json config 3: 16 flow frames, 224x224)."""

    def __init__(self, n_items, sequence_length=17, embed_dim=512, num_classes=140, resolution=224, seed=3):
        from . import synth
        self.synth, self.n, self.T, self.E, self.C, self.R, self.seed = synth, n_items, sequence_length, embed_dim, num_classes, resolution, seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        s = self.synth
        return {"video_id": f"synthetic_{idx:06d}",
                "rgb_emb": s.normal(self.seed, f"rgb/{idx}", (self.T, self.E)),
                "flow_frames": s.randint_u8(self.seed, f"flow/{idx}", (self.T - 1, 3, self.R, self.R)),
                "labels": s.multi_hot_labels(self.seed, f"lab/{idx}", 1, self.C)[0]}


def collate_fn(samples):
    """dataset.py:137-148."""
    return {"video_id": [s["video_id"] for s in samples],
            "rgb_emb": torch.stack([s["rgb_emb"] for s in samples], dim=0),
            "flow_frames": torch.stack([s["flow_frames"] for s in samples], dim=0),
            "labels": torch.stack([s["labels"] for s in samples], dim=0)}
