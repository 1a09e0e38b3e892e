#!/usr/bin/env python3
"""Generate tests/golden/ak_like_libhdf5.h5 with the stock libhdf5 (tests/h5ref.py) using the call sequence h5py
performs for extract_embeddings.py:50-119 (root attributes, per-video group with gzip ``embeddings`` chunked (1,E),
``labels``, frame-count attributes, ``video_ids``) plus an extendable dataset grown as in
extract_embeddings_mammalNet.py:113-142.  Data are seeded random numbers (a fixture, no reference source).
    python tests/make_h5_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import h5ref  # noqa: E402


def content():
    rng = np.random.default_rng(7)
    vids = {}
    for i in range(11):
        T = 3 + 2 * i
        lab = np.zeros(140, dtype=np.float32)
        lab[rng.integers(0, 140, 3)] = 1.0
        vids[f"AAACXZTV_{i:02d}.mp4"] = (rng.standard_normal((T, 32)).astype(np.float32), lab, T, 10 * T + 1)
    return vids


def main(path):
    R = h5ref.H5Ref()
    f = R.create(path)
    for k, v in (("num_classes", 140), ("dataset_name", "AnimalKingdom"), ("type", "val"), ("clip_model", "ViT-B/16")):
        R.set_attr(f, k, v)
    vids = content()
    for vid, (emb, lab, tf, of) in vids.items():
        g = R.group(f, vid)
        R.lib.H5Dclose(R.dataset(g, "embeddings", emb, chunks=(1, emb.shape[1]), gzip=4))
        R.lib.H5Dclose(R.dataset(g, "labels", lab))
        R.set_attr(g, "total_frames", tf)
        R.set_attr(g, "original_frames", of)
        R.lib.H5Gclose(g)
    g = R.group(f, "grown")
    d = R.dataset(g, "embeddings", np.zeros((0, 32), np.float32), chunks=(4, 32), gzip=4, maxshape=(None, 32))
    for j in range(3):
        R.append_rows(d, np.full((4 if j < 2 else 3, 32), j + 1, np.float32))
    R.lib.H5Dclose(d)
    R.set_attr(g, "skipped_low_ram", True)
    R.lib.H5Gclose(g)
    R.string_dataset(f, "video_ids", list(vids))
    R.close(f)


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ak_like_libhdf5.h5"))
