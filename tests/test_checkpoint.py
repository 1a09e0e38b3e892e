"""CPU: checkpoint plumbing with the reference's layouts: DataParallel ``module.`` prefix on write, both forms
on read, weights-only loads, TFAM checkpoint dict, HF CLIPModel -> OpenAI-clip key remap (round trip against the oracle's
OpenAI -> HF map, which tests/golden/vit.npz pins to transformers.CLIPModel)."""
import torch

from oracle import vit as ovit
from vimo_clip_amd import checkpoint as ck
from vimo_clip_amd import synth


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(3, 2)
        self.b = torch.nn.LayerNorm(2)


def test_prefix_round_trip_and_weights_only_load(tmp_path):
    m = _Tiny()
    path = tmp_path / "run - best" / "student_best.pth"
    ck.save_state_dict(m, str(path))
    raw = torch.load(str(path), weights_only=True)
    assert set(raw) == {"module." + k for k in m.state_dict()}          # what nn.DataParallel(model).state_dict() holds (train.py:167)
    m2 = _Tiny()
    ck.load_state_dict(m2, str(path))                                    # strict, prefix stripped
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    ck.load_state_dict(m2, {k: v for k, v in m.state_dict().items()})    # un-prefixed dicts load too
    assert ck.strip_prefix(ck.add_prefix({"x": 1})) == {"x": 1}


def test_snapshot_is_a_copy_not_a_view():
    m = _Tiny()
    snap = ck.snapshot(m)
    with torch.no_grad():
        m.a.weight.add_(1.0)
    assert not torch.equal(snap["module.a.weight"], m.a.weight)


def test_tfam_checkpoint_dict_layout_loads(tmp_path):
    m = _Tiny()
    state = {"epoch": 3, "state_dict": ck.snapshot(m), "optimizer": {"step": 5}, "scheduler": {"last_epoch": 3},
             "best_val_loss": 0.5, "best_val_mAP": 0.25}                # TFAM/train_and_eval.py:134-141
    torch.save(state, str(tmp_path / "best_model.pth"))
    m2 = _Tiny()
    ck.load_state_dict(m2, str(tmp_path / "best_model.pth"))
    assert torch.equal(m2.a.weight, m.a.weight)


def test_hf_to_openai_key_remap_inverts_the_oracle_map():
    sd = synth.vit_state_dict("ViT-tiny/32", 5)
    H = synth.VIT_GEOMETRY["ViT-tiny/32"][4]
    back = ck.hf_clip_to_openai_visual(ovit.openai_to_hf_vision(sd, H))
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
