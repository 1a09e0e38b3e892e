"""CPU: host-side logic of the product package (index/sampling/padding contracts, module surfaces, metric,
LR schedule) against the oracle loops and the golden fixtures recorded from the reference."""
import math

import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import indexing, metrics as ometrics, tfam as otfam
from vimo_clip_amd import synth
from vimo_clip_amd import dataset as ds
from vimo_clip_amd import extract_embeddings as ee
from vimo_clip_amd.TFAM.data import collate_fn_pad, sparse_sampling


def test_frame_index_sampling_golden(golden):
    g = golden["indexing"]
    for k in g.files:
        p = k.split("/")
        if p[0] == "frames":
            mx = None if p[2] == "None" else int(p[2])
            got = ee.sample_frame_indices(int(p[1]), mx)
            assert got.dtype == g[k].dtype and np.array_equal(got, g[k]), k


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 3000), st.one_of(st.none(), st.integers(1, 400)))
def test_frame_index_sampling_property(total, mx):
    assert np.array_equal(ee.sample_frame_indices(total, mx), indexing.sample_frame_indices(total, mx))


def test_sparse_sampling_golden(golden):
    g = golden["indexing"]
    for k in g.files:
        p = k.split("/")
        if p[0] == "sparse":
            T, n = int(p[1]), int(p[2])
            emb = torch.arange(T, dtype=torch.float32).unsqueeze(1)
            assert np.array_equal(sparse_sampling(emb, n)[:, 0].long().numpy(), g[k]), k


def test_collate_fn_pad_golden(golden):
    g = golden["indexing"]
    lens_r, lens_f = list(g["collate/lens_rgb"]), list(g["collate/lens_flow"])
    batch = [dict(video_id=f"v{i}", embeddings=torch.ones(int(a), 4) * (i + 1), flow_embeddings=torch.ones(int(b), 4) * (i + 1),
                  labels=torch.zeros(3)) for i, (a, b) in enumerate(zip(lens_r, lens_f))]
    col = collate_fn_pad(batch)
    assert np.array_equal(col["mask_rgb"].numpy(), g["collate/mask_rgb"])
    assert np.array_equal(col["mask_flow"].numpy(), g["collate/mask_flow"])
    assert np.array_equal(col["embeddings"].numpy(), g["collate/embeddings"])
    assert col["mask_rgb"].dtype == torch.bool


@settings(max_examples=200, deadline=None)
@given(st.dictionaries(st.text("abc", min_size=1, max_size=3), st.integers(0, 200), max_size=6), st.integers(1, 40))
def test_segment_table_property(lengths, seq_len):
    assert ds.build_segments(lengths, seq_len) == indexing.build_segments(lengths, seq_len)


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 120), st.integers(2, 33), st.integers(0, 130), st.data())
def test_segment_slicing_property(T, seq_len, t_flow, data):
    segs = ds.build_segments({"v": T}, seq_len)
    _, start, seg_len = segs[data.draw(st.integers(0, len(segs) - 1))]
    emb = torch.arange(T, dtype=torch.float32).unsqueeze(1).repeat(1, 3)
    rgb = ds.slice_rgb_segment(emb, start, seg_len, seq_len)
    assert rgb.shape == (seq_len, 3)
    assert rgb[:, 0].long().tolist() == indexing.rgb_segment_indices(start, seg_len, seq_len)
    flow = torch.arange(t_flow, dtype=torch.float32).view(t_flow, 1, 1, 1).repeat(1, 1, 2, 2) + 1.0   # frame i holds i+1; 0 = zero pad
    seq = ds.slice_flow_segment(flow, start, seg_len, seq_len)
    want = indexing.flow_segment_indices(start, seg_len, seq_len, t_flow)
    assert seq.shape[0] == len(want)
    assert [int(v) - 1 for v in seq[:, 0, 0, 0].tolist()] == want


def test_collate_fn_shapes():
    d = ds.SyntheticSegmentDataset(3, sequence_length=5, embed_dim=8, resolution=32)
    b = ds.collate_fn([d[0], d[1], d[2]])
    assert b["rgb_emb"].shape == (3, 5, 8) and b["flow_frames"].shape == (3, 4, 3, 32, 32) and b["flow_frames"].dtype == torch.uint8
    assert b["labels"].shape == (3, 140)


def test_multi_hot_and_layout():
    assert ee.multi_hot([0, 139, 7, 500], 140).sum() == 3
    x = torch.zeros(2, 5, 7, 3, dtype=torch.uint8)
    assert ee.frames_to_nchw(x).shape == (2, 3, 5, 7)


def test_micro_ap_matches_oracle_and_golden(golden):
    from vimo_clip_amd.metrics import MultilabelAveragePrecision
    g = golden["metrics"]
    for i in range(3):
        N, C, quant = int(g[f"ap{i}/N"]), int(g[f"ap{i}/C"]), bool(g[f"ap{i}/quant"])
        logits = synth.normal(50 + i, "ap_logits", (N, C), std=2.0)
        if quant:
            logits = torch.round(logits * 2) / 2
        y = synth.multi_hot_labels(50 + i, "ap_labels", N, C)
        m = MultilabelAveragePrecision(num_labels=C, average="micro")
        for s in range(0, N, 16):                       # batched updates, like the training loop
            m.update(logits[s:s + 16], y[s:s + 16].int())
        assert abs(float(m.compute()) - float(g[f"ap{i}/value"])) < 1e-6
    # the per-update sigmoid rule: a batch already inside [0,1] is NOT squashed again
    m = MultilabelAveragePrecision(num_labels=4)
    m.update(torch.tensor([[0.2, 0.9, 0.1, 0.6]]), torch.tensor([[0, 1, 0, 1]]))
    m.update(torch.tensor([[2.0, -1.0, 0.5, 0.3]]), torch.tensor([[1, 0, 0, 1]]))
    s = np.concatenate([[0.2, 0.9, 0.1, 0.6], 1 / (1 + np.exp(-np.array([2.0, -1.0, 0.5, 0.3])))])
    want = ometrics.micro_average_precision(s.astype(np.float32), np.array([0, 1, 0, 1, 1, 0, 0, 1]))
    assert abs(float(m.compute()) - want) < 1e-6


def test_state_dict_surface_matches_reference_keys():
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    from vimo_clip_amd.models import FlowStudentModel
    m = AMO_CLIP(d_model=64, nhead=8, num_layers=2, dim_feedforward=128, num_classes=140, device="cpu")
    ref = synth.tfam_state_dict(64, 8, 2, 128, 140, 1)      # these keys loaded strict=True into the reference module
    assert set(m.state_dict().keys()) == set(ref.keys())
    assert all(m.state_dict()[k].shape == v.shape for k, v in ref.items())
    m.load_state_dict(ref, strict=True)
    assert sum(p.numel() for p in AMO_CLIP(d_model=768, nhead=8, num_layers=4, dim_feedforward=2048, device="cpu").parameters()) == 33042700
    s = FlowStudentModel("ViT-tiny/32", device="cpu", num_classes=140)
    ref = synth.student_state_dict("ViT-tiny/32", 1)
    assert set(s.state_dict().keys()) == set(ref.keys())
    s.load_state_dict({k: v for k, v in ref.items()}, strict=True)
    # DataParallel-prefixed checkpoints (train.py:167) load after stripping "module."
    s.load_state_dict({k[len("module."):]: v for k, v in {"module." + k: v for k, v in ref.items()}.items()}, strict=True)
    assert s.visual_encoder.output_dim == 64 and hasattr(s.preprocess, "transforms") and s.residual_mlp.alpha == 0.1
    assert torch.all(FlowStudentModel("ViT-tiny/32", device="cpu").residual_mlp.fc2.weight == 0)      # zero-init (:24-25)


def test_used_parameters_per_mode():
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    kw = dict(d_model=64, nhead=8, num_layers=1, dim_feedforward=128, device="cpu")
    names = lambda m: {n for n, p in m.named_parameters() if any(p is q for q in m.used_parameters())}
    cross = names(AMO_CLIP(**kw))
    assert not any(n.startswith("projection_layer") for n in cross) and any("cross_attn" in n for n in cross)
    rgb = names(AMO_CLIP(use_only_rgb=True, **kw))
    assert not any("cross" in n or n.startswith("projection_layer") for n in rgb)
    cat = names(AMO_CLIP(use_cross_attention=False, concat_dim=-1, **kw))
    assert any(n.startswith("projection_layer") for n in cat) and not any("cross" in n for n in cat)


def test_cosine_schedule_closed_form():
    from vimo_clip_amd.optim import CosineAnnealingLR

    class O:
        param_groups = [{"lr": 1e-4}]
    sch = CosineAnnealingLR(O(), T_max=30, eta_min=1e-6)
    ref = torch.optim.lr_scheduler.CosineAnnealingLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4), T_max=30, eta_min=1e-6)
    for e in range(1, 31):
        sch.step()
        ref.optimizer.step()
        ref.step()
        assert math.isclose(sch.get_last_lr()[0], ref.get_last_lr()[0], rel_tol=1e-9)
        assert math.isclose(sch.get_last_lr()[0], otfam.cosine_lr(e, 30), rel_tol=1e-12)


def test_accuracy_metric_and_task_objects():
    import torch
    from vimo_clip_amd.metrics import Accuracy
    from vimo_clip_amd.TFAM.train_and_eval import Config, task_objects
    m = Accuracy(num_classes=4)
    logits = torch.tensor([[0.1, 2.0, 0.0, -1.0], [3.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.5, 0.4], [0.0, 0.0, 0.0, 9.0]])
    onehot = torch.eye(4)[[1, 2, 2, 3]].to(torch.int)
    m.update(logits[:2], onehot[:2])
    m.update(logits[2:], onehot[2:])
    assert abs(float(m.compute()) - 0.75) < 1e-7
    m.reset()
    m.update(logits, torch.tensor([1, 0, 2, 3]))                      # index targets are accepted too
    assert float(m.compute()) == 1.0
    crit, metric = task_objects(Config(task="singlelabel", num_classes=4))
    assert isinstance(metric, Accuracy) and crit.__name__ == "cross_entropy_loss"
    import pytest
    with pytest.raises(ValueError):
        task_objects(Config(task="ranking"))


def test_yaml_config_and_sweep(tmp_path):
    from vimo_clip_amd.TFAM import sweep
    from vimo_clip_amd.TFAM.train_and_eval import Config
    paths = sweep.write_sweep(str(tmp_path / "cfg"), base={"training": {"epochs": 2}, "data": {"num_classes": 12}})
    assert len(paths) == 20                                             # 5 fusion modes x PE x 2 dropout pairs
    cfgs = [Config.from_yaml(p) for p in paths]
    assert {(c.use_cross_attn, c.use_only_rgb, c.use_only_flow, c.concat_dim) for c in cfgs} == {
        (True, False, False, 1), (False, False, False, 1), (False, False, False, -1), (False, True, False, 1), (False, False, True, 1)}
    assert all(c.epochs == 2 and c.num_classes == 12 and c.motion_key == "frame_diff" and c.lr == 1e-4 for c in cfgs)
    (tmp_path / "bad.yaml").write_text("training: {mode: both}\nlogging: {}\ndata: {}\nmodel: {}\n")
    import pytest
    with pytest.raises(KeyError):
        Config.from_yaml(str(tmp_path / "bad.yaml"))
