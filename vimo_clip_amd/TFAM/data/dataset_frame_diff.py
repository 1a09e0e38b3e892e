"""Frame-difference twin of the TFAM dataset — drop-in for TFAM/data/dataset_frame_diff.py: identical arithmetic, the
motion stream is called ``frame_diff`` (item key ``frame_diff_embeddings``, mask ``mask_frame_diff``; constructor argument
``frame_diff_path``)."""
from __future__ import annotations

from . import dataset as _flow

sparse_sampling = _flow.sparse_sampling


class HDF5VideoDataset(_flow.HDF5VideoDataset):
    motion_key = "frame_diff"

    def __init__(self, hdf5_path, frame_diff_path, transform=None, num_frames=None, max_frames=None):
        super().__init__(hdf5_path, frame_diff_path, transform=transform, num_frames=num_frames, max_frames=max_frames)
        self.frame_diff_path, self.frame_diff_keys = self.flow_path, self.flow_keys


def collate_fn_pad(batch):
    return _flow.collate_fn_pad(batch, motion_key="frame_diff")
