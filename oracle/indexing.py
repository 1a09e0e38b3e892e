"""Oracle: integer index / sampling / padding contracts, bit-exact (test infrastructure only).

Deliberately written with plain Python loops and numpy so that it shares no code path with the
product's vectorised implementations.
"""
from __future__ import annotations

import numpy as np
import torch


def sample_frame_indices(total_frames: int, max_frames=None) -> np.ndarray:
    """extract_embeddings.py:77-81."""
    if max_frames is None or total_frames <= max_frames:
        return np.array(list(range(total_frames)), dtype=np.int64)
    step = total_frames // max_frames
    out = []
    i = 0
    while i < total_frames and len(out) < max_frames:
        out.append(i)
        i += step
    return np.array(out, dtype=np.int64)


def build_segments(lengths: dict, sequence_length: int) -> list:
    """dataset.py:35-57: non-overlapping windows (video_id, start, seg_len); T == 0 skipped."""
    segs = []
    for k, T in lengths.items():
        if T == 0:
            continue
        start = 0
        while start < T:
            seg_len = sequence_length if T - start >= sequence_length else T - start
            segs.append((k, start, seg_len))
            start += seg_len
    return segs


def rgb_segment_indices(start: int, seg_len: int, sequence_length: int) -> list:
    """dataset.py:77-91 as a gather index list: slice then pad by repeating the last row."""
    idx = [start + i for i in range(seg_len)]
    while len(idx) < sequence_length:
        idx.append(start + seg_len - 1)
    return idx


def flow_segment_indices(start: int, seg_len: int, sequence_length: int, t_flow: int) -> list:
    """dataset.py:101-127 as a gather index list; -1 means an all-zero frame (:124-126)."""
    leftover = sequence_length - seg_len
    flow_seg_len = seg_len - 1
    if leftover > 0:
        flow_seg_len = sequence_length - 1
    flow_start = min(start, max(t_flow - 1, 0))
    flow_end = min(start + flow_seg_len, t_flow)
    idx = [i for i in range(flow_start, flow_end)]
    if len(idx) > 0:
        while len(idx) < flow_seg_len:
            idx.append(idx[-1])
    else:
        idx = [-1] * max(flow_seg_len, 0)
    return idx


def sparse_sampling_indices(total_frames: int, num_frames: int) -> torch.Tensor:
    """TFAM/data/dataset.py:7-12.  Uses torch.linspace itself: its float32 two-sided evaluation is
    part of the contract."""
    if total_frames > num_frames:
        return torch.linspace(0, total_frames - 1, num_frames).long()
    return torch.arange(total_frames)


def pad_and_mask(lengths: list, t_max=None):
    """TFAM/data/dataset.py:86-102: mask[b,t] = t < len[b]."""
    t_max = max(lengths) if t_max is None else t_max
    m = np.zeros((len(lengths), t_max), dtype=bool)
    for b, n in enumerate(lengths):
        for t in range(t_max):
            m[b, t] = t < n
    return m
