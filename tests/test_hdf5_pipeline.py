"""The HDF5 data path end to end on the host: TFAM / student datasets reading the reference's file layout through
h5lite (TFAM/data/dataset.py:15-73, dataset.py:24-134) and the streaming exporter's control flow
(inference_frame_diff.py:235-410) with a stub model.  GPU leg: the real extractor / student in test_gpu_e2e.py."""
import os

import numpy as np
import pytest
import torch

from vimo_clip_amd import h5lite as h5
from vimo_clip_amd import inference as inf
from vimo_clip_amd.dataset import HDF5VideoDataset as StudentDataset
from vimo_clip_amd.dataset import collate_fn
from vimo_clip_amd.TFAM.data.dataset import HDF5VideoDataset, collate_fn_pad, sparse_sampling


def _make_rgb_and_flow(tmp_path, lengths, E=24, C=140):
    rng = np.random.default_rng(1)
    rgb, flow = {}, {}
    with h5.File(str(tmp_path / "rgb.h5"), "w") as f, h5.File(str(tmp_path / "flow.h5"), "w") as g:
        for i, T in enumerate(lengths):
            vid = f"clip_{i:02d}.mp4"
            rgb[vid] = rng.standard_normal((T, E)).astype(np.float32)
            flow[vid] = rng.standard_normal((T - 1, E)).astype(np.float32)
            lab = np.zeros(C, np.float32)
            lab[i % C] = 1.0
            grp = f.create_group(vid)
            grp.create_dataset("embeddings", data=rgb[vid], compression="gzip", chunks=(1, E))
            grp.create_dataset("labels", data=lab)
            g.create_group(vid.split(".")[0]).create_dataset("embeddings", data=flow[vid])      # inference.py:111-112 layout
    return rgb, flow


def test_tfam_dataset_over_hdf5(tmp_path):
    lengths = [5, 40, 17, 64, 9]
    rgb, flow = _make_rgb_and_flow(tmp_path, lengths)
    ds = HDF5VideoDataset(str(tmp_path / "rgb.h5"), str(tmp_path / "flow.h5"), num_frames=None, max_frames=None)
    assert len(ds) == 5 and ds.keys == sorted(rgb)
    it = ds[3]
    assert it["video_id"] == "clip_03.mp4" and it["total_frames"] == 64
    assert torch.equal(it["embeddings"], torch.from_numpy(rgb["clip_03.mp4"]))
    assert torch.equal(it["flow_embeddings"], torch.from_numpy(flow["clip_03.mp4"]))        # key.split(".")[0] mapping (:65)
    assert it["labels"].shape == (140,) and it["labels"][3] == 1.0
    ds16 = HDF5VideoDataset(str(tmp_path / "rgb.h5"), str(tmp_path / "flow.h5"), num_frames=16, max_frames=60)
    assert ds16.keys == ["clip_00.mp4", "clip_01.mp4", "clip_02.mp4", "clip_04.mp4"]          # T < max_frames only (:28-31)
    it = ds16[1]
    assert torch.equal(it["embeddings"], sparse_sampling(torch.from_numpy(rgb["clip_01.mp4"]), 16)) and it["embeddings"].shape[0] == 16
    batch = collate_fn_pad([ds[0], ds[1], ds[2]])
    assert batch["embeddings"].shape == (3, 40, 24) and batch["flow_embeddings"].shape == (3, 39, 24)
    assert batch["mask_rgb"].sum(1).tolist() == [5, 40, 17] and batch["mask_flow"].sum(1).tolist() == [4, 39, 16]


def test_student_dataset_over_hdf5_and_npy_videos(tmp_path):
    rgb, _ = _make_rgb_and_flow(tmp_path, [5, 9], E=16)
    flow_dir = tmp_path / "flow_videos"
    flow_dir.mkdir()
    rng = np.random.default_rng(2)
    vids = {}
    for vid, e in rgb.items():
        vids[vid] = rng.integers(0, 256, (e.shape[0] - 1, 20, 28, 3), dtype=np.uint8)
        np.save(str(flow_dir / (vid + ".npy")), vids[vid])
    ds = StudentDataset(str(tmp_path / "rgb.h5"), str(flow_dir), sequence_length=4)
    assert [s[1:] for s in ds.segments] == [(0, 4), (4, 1), (0, 4), (4, 4), (8, 1)]
    item = ds[3]
    assert item["video_id"] == "clip_01.mp4" and item["rgb_emb"].shape == (4, 16) and item["flow_frames"].shape == (3, 3, 20, 28)
    assert torch.equal(item["rgb_emb"], torch.from_numpy(rgb["clip_01.mp4"][4:8]))
    assert torch.equal(item["flow_frames"], torch.from_numpy(vids["clip_01.mp4"][4:7]).permute(0, 3, 1, 2))
    b = collate_fn([ds[0], ds[2]])
    assert b["rgb_emb"].shape == (2, 4, 16) and b["flow_frames"].shape == (2, 3, 3, 20, 28) and b["labels"].shape == (2, 140)


class _StubStudent(torch.nn.Module):
    """(1,n,3,H,W) u8 -> per-frame [mean R, mean G, mean B, n-th frame index marker]; counts the forward calls."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, x):
        self.calls.append(x.shape[1])
        m = x.float().mean(dim=(3, 4))                                   # (1,n,3)
        return torch.cat([m, m.sum(-1, keepdim=True)], -1), None, None


def test_streaming_export_resume_errors_and_layout(tmp_path):
    vdir = tmp_path / "videos" / "sub"
    vdir.mkdir(parents=True)
    rng = np.random.default_rng(4)
    frames = {"a": rng.integers(0, 256, (70, 8, 10, 3), dtype=np.uint8), "b": rng.integers(0, 256, (5, 8, 10, 3), dtype=np.uint8),
              "empty": np.zeros((0, 8, 10, 3), np.uint8)}
    for k, v in frames.items():
        np.save(str(vdir / f"{k}.npy"), v)
    np.save(str(vdir / "bad.npy"), np.zeros((4, 4), np.float32))         # not a frame stack -> error attribute
    paths = inf.FrameDiffVideoDataset(str(tmp_path / "videos")).video_paths
    assert [os.path.basename(p) for p in paths] == ["a.npy", "b.npy", "bad.npy", "empty.npy"]
    out = str(tmp_path / "out" / "emb.h5")
    model = _StubStudent()
    with pytest.warns(UserWarning, match="Error on bad"):
        stats = inf.export_embeddings(paths[:3], model, out, chunk_size=32, flush_interval_s=0, compression="gzip")
    assert stats == {"processed": 2, "skipped_existing": 0, "skipped_low_ram": 0, "errors": 1}
    assert model.calls == [32, 32, 6, 5]
    with h5.File(out, "r") as f:
        assert f.keys() == ["a", "b", "bad"]
        d = f["a/embeddings"]
        assert d.shape == (70, 4) and d.maxshape == (None, 4) and d.chunks == (32, 4) and d.dtype == np.float32 and d.compression == "gzip"
        want = frames["a"].astype(np.float32).mean(axis=(1, 2))
        assert np.allclose(d[:, :3], want, atol=1e-4)
        assert "expected [T,H,W,3]" in f["bad"].attrs["error"] and "embeddings" not in f["bad"]
    stats = inf.export_embeddings(paths, model, out, resume=True, chunk_size=32)            # resumed run: only "empty" is new
    assert stats == {"processed": 1, "skipped_existing": 3, "skipped_low_ram": 0, "errors": 0}
    with h5.File(out, "r") as f:
        assert f["empty/embeddings"].shape == (0, 0)
    stats = inf.export_embeddings(paths[:1], model, out, min_free_gb=1e6, resume=False, overwrite=True)
    assert stats["skipped_low_ram"] == 1
    with h5.File(out, "r") as f:
        assert f.keys() == ["a"] and f["a"].attrs["skipped_low_ram"] == True  # noqa: E712
    stats = inf.export_embeddings(paths[:2], model, out, streaming=False)                     # inference.py: whole video, contiguous
    with h5.File(out, "r") as f:
        assert f["b/embeddings"].shape == (5, 4) and f["b/embeddings"].chunks is None
