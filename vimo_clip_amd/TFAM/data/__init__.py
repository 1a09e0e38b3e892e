from .dataset import HDF5VideoDataset, collate_fn_pad, sparse_sampling  # noqa: F401
