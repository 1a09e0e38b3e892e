"""Teacher embedding extraction — the accelerated slice of the reference's extract_embeddings.py.

``sample_frame_indices`` (:77-81), ``frames_to_nchw`` (:84), ``multi_hot`` (:97-103) and ``encode_video_frames``
(:89-94: per-frame PIL + CLIPImageProcessor + ``get_image_features`` -> one fused HIP pass) are the pieces on
the hot path; ``create_hdf5_dataset`` keeps the reference's signature and writes the reference's file layout through
the native HDF5 writer (``h5lite``; no h5py).  Video decode is host I/O outside the path: decord when it is installed;
offline a ``frame_source`` callable (or ``.npy`` frame stacks next to the video names) supplies the decoded frames.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def sample_frame_indices(total_frames: int, max_frames=None) -> np.ndarray:
    """extract_embeddings.py:77-81, bit-exact."""
    if (max_frames is None) or (total_frames <= max_frames):
        return np.arange(total_frames)
    step = total_frames // max_frames
    return np.arange(0, total_frames, step)[:max_frames]


def frames_to_nchw(frames_nhwc: torch.Tensor) -> torch.Tensor:
    """decord ``get_batch`` gives [T,H,W,3] u8; the reference permutes to [T,3,H,W] (:84)."""
    return frames_nhwc.permute(0, 3, 1, 2)


def multi_hot(labels, num_classes: int) -> np.ndarray:
    """:97-103 (labels outside [0, num_classes) are skipped with a warning in the reference)."""
    out = np.zeros(num_classes, dtype=np.float32)
    for lab in labels:
        if 0 <= int(lab) < num_classes:
            out[int(lab)] = 1.0
    return out


@torch.no_grad()
def encode_video_frames(encoder, frames_u8_nchw: torch.Tensor) -> np.ndarray:
    """[T,3,R,R] u8 (already at the model resolution) -> [T,E] float32 numpy, like
    ``clip_model.get_image_features(pixel_values).cpu().numpy()`` (:94)."""
    vis = getattr(encoder, "visual", encoder)
    dev = next(vis.parameters()).device
    return vis.encode_frames_u8(frames_u8_nchw.to(dev), crop_mode="hf").cpu().numpy()      # HF centre-crop rounding


class NpyVideoReader:
    """Offline stand-in for ``decord.VideoReader``: a ``[T,H,W,3]`` uint8 ``.npy`` stack, ``len()`` and ``get_batch()``."""

    def __init__(self, path):
        self.arr = np.load(path, mmap_mode="r")
        if self.arr.ndim != 4 or self.arr.shape[-1] != 3 or self.arr.dtype != np.uint8:
            raise ValueError(f"{path}: expected [T,H,W,3] uint8 frames, got {self.arr.shape} {self.arr.dtype}")

    def __len__(self):
        return self.arr.shape[0]

    def get_batch(self, indices):
        return torch.from_numpy(np.array(self.arr[np.asarray(indices)]))


def open_video(video_path):
    """``VideoReader(video_path, ctx=cpu(0))`` (:72) — or the ``.npy`` frame stack of the same name when decord is absent."""
    for cand in (video_path if video_path.endswith(".npy") else None, video_path + ".npy", os.path.splitext(video_path)[0] + ".npy"):
        if cand and os.path.exists(cand):
            return NpyVideoReader(cand)
    import decord
    decord.bridge.set_bridge("torch")
    return decord.VideoReader(video_path, ctx=decord.cpu(0))


def read_num_classes(class_file) -> int:
    """``len(pd.read_csv(class_file))`` (:33): data rows of the class CSV (header excluded)."""
    import csv
    with open(class_file, "r", encoding="utf-8", newline="") as f:
        return max(0, sum(1 for _ in csv.reader(f)) - 1)


def create_hdf5_dataset(data_root, annotation_file, class_file, output_hdf5, max_frames=None, encoder=None,
                        clip_model_name="ViT-B/16", frame_source=None, compression="gzip"):
    """Reference signature (:23) + an optional prebuilt encoder / frame source.  Output layout (:106-119): group per
    video with ``embeddings`` [T,E] f32 gzip chunks (1,E), ``labels`` [C] f32, attrs total_frames/original_frames; root
    attrs and a ``video_ids`` dataset.  A video that is missing or fails to decode is reported and skipped (:62-64,
    :113-115).  Returns the number of videos written."""
    from . import h5lite as h5py
    from .clip_vit import CLIPImageEncoder
    frame_source = frame_source or open_video
    encoder = encoder or CLIPImageEncoder(clip_model_name).cuda().eval()
    out_dir = os.path.dirname(output_hdf5)
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir)
    num_classes = read_num_classes(class_file)
    with open(annotation_file, "r", encoding="utf-8") as f:
        annotations = [line.strip().split() for line in f if line.strip()]
    written = 0
    with h5py.File(output_hdf5, "w") as hf:
        hf.attrs["num_classes"], hf.attrs["dataset_name"], hf.attrs["type"], hf.attrs["clip_model"] = num_classes, "AnimalKingdom", "val", clip_model_name
        for info in annotations:
            video_id, video_path = info[0], os.path.join(data_root, info[0])
            if frame_source is open_video and not any(os.path.exists(c) for c in (video_path, video_path + ".npy", os.path.splitext(video_path)[0] + ".npy")):
                print(f"Video no encontrado: {video_path}")
                continue
            try:
                vr = frame_source(video_path)
                total = len(vr)
                idx = sample_frame_indices(total, max_frames)
                frames = frames_to_nchw(vr.get_batch(idx))
                emb = encode_video_frames(encoder, frames)      # resize + centre crop + normalise + encode on the GPU
                grp = hf.create_group(video_id)
                grp.create_dataset("embeddings", data=emb, compression=compression, chunks=(1, emb.shape[1]))
                grp.create_dataset("labels", data=multi_hot([int(x) for x in info[1:]], num_classes))
                grp.attrs["total_frames"], grp.attrs["original_frames"] = len(idx), total
                written += 1
            except Exception as e:  # the reference logs and continues (:113-115)
                print(f"\nError procesando {video_id}: {str(e)}")
                continue
        hf.create_dataset("video_ids", data=np.array([a[0] for a in annotations], dtype=h5py.string_dtype()))
    return written
