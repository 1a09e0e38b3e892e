"""TFAM dataset — drop-in for TFAM/data/dataset.py (sparse_sampling, HDF5VideoDataset, collate_fn_pad)."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


def sparse_sampling(embeddings, num_frames):
    """TFAM/data/dataset.py:7-12.  torch.linspace itself is part of the contract (float32 two-sided
    evaluation, then .long() truncation)."""
    total_frames = embeddings.shape[0]
    if total_frames > num_frames:
        embeddings = embeddings[torch.linspace(0, total_frames - 1, num_frames).long()]
    return embeddings


class HDF5VideoDataset(Dataset):
    """TFAM/data/dataset.py:15-73: RGB group by key, flow group by ``key.split('.')[0]``."""
    motion_key = "flow"            # item key ``{motion_key}_embeddings`` (dataset_frame_diff.py renames it, nothing else)

    def __init__(self, hdf5_path, flow_path, transform=None, num_frames=None, max_frames=None):
        from ... import h5lite as h5py        # native reader of the reference's HDF5 layout
        self.hdf5_path, self.flow_path, self.transform = hdf5_path, flow_path, transform
        self.num_frames, self.max_frames = num_frames, max_frames

        def keys_of(path):
            with h5py.File(path, "r") as f:
                ks = list(f.keys())
                if self.max_frames:
                    ks = [k for k in ks if f[k]["embeddings"].shape[0] < self.max_frames]
            return ks

        self.keys, self.flow_keys = keys_of(hdf5_path), keys_of(flow_path)

    def __len__(self):
        return len(self.keys)

    def __getitem__(self, idx):
        from ... import h5lite as h5py
        video_id = self.keys[idx]
        with h5py.File(self.hdf5_path, "r") as f:
            embeddings = torch.from_numpy(f[video_id]["embeddings"][:])
            labels = torch.from_numpy(f[video_id]["labels"][:])
        with h5py.File(self.flow_path, "r") as f:
            flow_embeddings = torch.from_numpy(f[video_id.split(".")[0]]["embeddings"][:])
        if self.num_frames:
            embeddings, flow_embeddings = sparse_sampling(embeddings, self.num_frames), sparse_sampling(flow_embeddings, self.num_frames)
        if self.transform:
            embeddings, flow_embeddings = self.transform(embeddings), self.transform(flow_embeddings)
        return {"video_id": video_id, "embeddings": embeddings.float(), f"{self.motion_key}_embeddings": flow_embeddings.float(),
                "labels": labels, "total_frames": embeddings.shape[0]}


def collate_fn_pad(batch, motion_key="flow"):
    """TFAM/data/dataset.py:76-112: zero-pad both streams to the batch maxima; masks True = real token."""
    embeddings = [item["embeddings"] for item in batch]
    flow_embeddings = [item[f"{motion_key}_embeddings"] for item in batch]
    lens_rgb = torch.tensor([x.shape[0] for x in embeddings])
    lens_flow = torch.tensor([x.shape[0] for x in flow_embeddings])
    padded_rgb = torch.nn.utils.rnn.pad_sequence(embeddings, batch_first=True)
    padded_flow = torch.nn.utils.rnn.pad_sequence(flow_embeddings, batch_first=True)
    mask_rgb = torch.arange(padded_rgb.size(1)).unsqueeze(0) < lens_rgb.unsqueeze(1)
    mask_flow = torch.arange(padded_flow.size(1)).unsqueeze(0) < lens_flow.unsqueeze(1)
    return {"video_id": [item["video_id"] for item in batch], "embeddings": padded_rgb, f"{motion_key}_embeddings": padded_flow,
            "labels": torch.stack([item["labels"] for item in batch]), "mask_rgb": mask_rgb, f"mask_{motion_key}": mask_flow}


class SyntheticEmbeddingDataset(Dataset):
    """Synthetic variable-length RGB / motion token stacks with class-dependent means (BASELINE.json configs 4-5):
    labels from an annotation list, T_rgb ~ U{tmin..tmax}, T_flow = T_rgb - 1."""

    def __init__(self, labels: torch.Tensor, d_model=768, tmin=17, tmax=64, seed=5, signal=0.5, motion_key="flow", class_seed=None):
        from ... import synth
        self.motion_key = motion_key
        self.labels, self.D, self.seed, self.signal, self.synth = labels.float(), d_model, seed, signal, synth
        n, C = labels.shape
        self.lengths = synth.randint(seed, "lengths", (n,), tmin, tmax + 1)
        # class_seed: train and validation sets that share it share the class directions (so a model can generalise)
        self.class_dirs = synth.normal(seed if class_seed is None else class_seed, "class_dirs", (C, d_model))

    def __len__(self):
        return self.labels.shape[0]

    def __getitem__(self, idx):
        T = int(self.lengths[idx])
        mean = self.signal * (self.labels[idx] @ self.class_dirs)
        rgb = self.synth.normal(self.seed, f"rgb/{idx}", (T, self.D)) + mean
        flow = self.synth.normal(self.seed, f"flow/{idx}", (T - 1, self.D)) + mean
        return {"video_id": f"v{idx:06d}", "embeddings": rgb, f"{self.motion_key}_embeddings": flow, "labels": self.labels[idx], "total_frames": T}
