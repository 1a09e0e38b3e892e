"""GPU: the fused TFAM TRAINING chains (vmc_tfam_train_fwd / vmc_tfam_head_bwd / vmc_tfam_layer_bwd, vimo_clip_amd/tfam_train.py)
against (a) the reference's own autograd (tests/golden/tfam.npz: TFAM/models/AMO_CLIP.py imported and run by oracle/make_golden.py),
(b) torch autograd through the CPU oracle for EVERY parameter, (c) the per-op training path with dropout on (identical seeds ->
identical masks), over the fusion modes, batch sizes, clip lengths and widths the chain covers.

Tolerances (bf16 compute against fp32 references, as tests/test_gpu_models.py): loss 5e-3 relative; gradients 4e-2 relative L2
(measured 0.5-1.5e-2); f16 is not used for gradients (a mean-reduced loss puts them in its subnormal range without loss scaling).
"""
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from oracle import tfam as otfam
from vimo_clip_amd import synth

pytestmark = pytest.mark.gpu


def _model(c, dtype=torch.bfloat16, dropout=0.0, mlp_dropout=0.0):
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    kw = mg.tfam_mode_kwargs(c["mode"])
    m = AMO_CLIP(d_model=c["D"], nhead=c["H"], num_layers=c["L"], dim_feedforward=c["ff"], num_classes=c["C"], use_pe=c["pe"],
                 dropout=dropout, mlp_dropout=mlp_dropout, device="cuda", compute_dtype=dtype, **kw).cuda()
    m.load_state_dict(synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"]), strict=True)
    return m.train()


def _count_fused(monkeypatch):
    from vimo_clip_amd import tfam_train
    calls = []
    orig = tfam_train.forward_train

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(out is not None)
        return out
    monkeypatch.setattr(tfam_train, "forward_train", spy)
    return calls


def _step(m, c, fused):
    from vimo_clip_amd.losses import bce_with_logits_loss
    m.fused_training = fused
    arena = any(getattr(p, "_vmc_grad", None) is not None for p in m.parameters())
    if not arena:
        m.zero_grad(set_to_none=True)
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"]).cuda()
    logits = m(rgb.cuda(), mot.cuda(), mask_rgb=mr.cuda(), mask_flow=mf.cuda())
    loss = bce_with_logits_loss(logits, y)
    loss.backward()
    used = {id(q) for q in m.used_parameters()}
    grads = {}
    for n, p in m.named_parameters():
        g = getattr(p, "_vmc_grad", None) if id(p) in used else None          # arena slot (written in place), else autograd's .grad
        g = p.grad if g is None else g
        if g is not None:
            grads[n] = g.detach().float().cpu().clone()
    return loss.item(), logits.detach().float().cpu(), grads


FUSED_GOLDEN = ["cross_d512", "cross_d768", "cross_d768_ragged_pe", "rgb_only", "flow_only", "concat_time"]


@pytest.mark.parametrize("name", FUSED_GOLDEN)
def test_fused_training_vs_reference_autograd(golden, name, monkeypatch):
    """Loss and the five recorded gradients of the reference's train-mode step (dropout 0) through the fused chains."""
    calls = _count_fused(monkeypatch)
    c = next(x for x in mg.TFAM_CASES if x["name"] == name)
    m = _model(c)
    loss, _, grads = _step(m, c, True)
    assert calls == [True], "the fused training chain was not taken"
    g = golden["tfam"]
    ref_loss = float(g[f"{name}/train_loss"])
    assert abs(loss - ref_loss) <= 5e-3 * abs(ref_loss), (loss, ref_loss)
    for k in ("classifier.4.weight", "classifier.1.bias", "layers.0.ffn.0.bias", "layers.0.self_attn.in_proj_bias", "layers.0.norm_self.weight"):
        ref = torch.from_numpy(g[f"{name}/grad/{k}"])
        got = grads[k]
        rel = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        rel_l2 = ((got - ref).norm() / (ref.norm() + 1e-20)).item()
        print(f"{name} grad {k}: rel-to-max {rel:.3e}, rel L2 {rel_l2:.3e}")
        assert rel_l2 <= 4e-2, (k, rel_l2)
        assert rel <= (2e-1 if "ffn.0" in k else 6e-2), (k, rel)
    used = {n for n, p in m.named_parameters() if id(p) in {id(q) for q in m.used_parameters()}}
    assert set(grads) == used                                        # exactly the statically-known used set gets gradients


def _oracle_grads(c):
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"]).items()}
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"])
    logits = otfam.amo_clip_forward(sd, rgb, mot, mr, mf, nhead=c["H"], use_pe=c["pe"], **mg.tfam_mode_kwargs(c["mode"]))
    loss = otfam.bce_with_logits_mean(logits, y)
    loss.backward()
    return loss.item(), logits.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}


CASES_ALL = [
    dict(name="b8_d768", D=768, H=8, L=4, ff=2048, C=140, B=8, Tr=16, Tf=15, mode="cross", pe=False, ragged=True, seed=51),
    dict(name="b16_d768", D=768, H=8, L=2, ff=2048, C=140, B=16, Tr=16, Tf=16, mode="cross", pe=False, ragged=True, seed=52),
    dict(name="b1_d512", D=512, H=8, L=2, ff=1024, C=20, B=1, Tr=7, Tf=6, mode="cross", pe=False, ragged=False, seed=53),
    dict(name="t32_d512", D=512, H=8, L=2, ff=2048, C=140, B=5, Tr=32, Tf=31, mode="cross", pe=False, ragged=True, seed=54),
    dict(name="t20_k12_d768", D=768, H=8, L=1, ff=512, C=140, B=3, Tr=20, Tf=12, mode="cross", pe=True, ragged=True, seed=55),
    dict(name="rgb_d768", D=768, H=12, L=2, ff=2048, C=140, B=6, Tr=16, Tf=15, mode="rgb", pe=False, ragged=True, seed=56),
    dict(name="concat_d512", D=512, H=8, L=3, ff=2048, C=140, B=4, Tr=16, Tf=15, mode="concat1", pe=False, ragged=True, seed=57),
    # clips longer than 32 tokens (round 3): query parts of 32 rows, four key tiles
    dict(name="t40_k39_d512", D=512, H=8, L=2, ff=1024, C=140, B=4, Tr=40, Tf=39, mode="cross", pe=False, ragged=True, seed=58),
    dict(name="t64_k63_d768", D=768, H=8, L=2, ff=1024, C=140, B=3, Tr=64, Tf=63, mode="cross", pe=False, ragged=True, seed=59),
    dict(name="t20_k50_d768", D=768, H=12, L=1, ff=512, C=140, B=3, Tr=20, Tf=50, mode="cross", pe=False, ragged=True, seed=60),
]


@pytest.mark.parametrize("c", CASES_ALL, ids=lambda c: c["name"])
def test_fused_training_every_gradient_vs_oracle_autograd(c, monkeypatch):
    """Every parameter gradient of one train-mode step (dropout 0) against torch autograd through the fp32 CPU oracle."""
    calls = _count_fused(monkeypatch)
    m = _model(c)
    loss, logits, grads = _step(m, c, True)
    assert calls == [True]
    ref_loss, ref_logits, ref = _oracle_grads(c)
    err = (logits - ref_logits).abs().max().item()
    print(f"{c['name']}: loss {loss:.6f} vs {ref_loss:.6f}, logits max abs err {err:.3e} (|ref|max {ref_logits.abs().max():.2f})")
    assert abs(loss - ref_loss) <= 5e-3 * abs(ref_loss)
    assert err <= 8e-3 * max(1.0, ref_logits.abs().max().item())
    assert set(grads) == set(ref), set(grads) ^ set(ref)
    worst = ("", 0.0)
    for k, r in ref.items():
        rel_l2 = ((grads[k] - r).norm() / (r.norm() + 1e-20)).item()
        worst = max(worst, (k, rel_l2), key=lambda t: t[1])
        # ffn.0 sits behind the ReLU: pre-activations within bf16 rounding distance of 0 flip relu'(z) for single elements; four
        # layers deep the per-op path measures the same 4.3-4.7e-2 on layers.0.ffn.0.weight, and 5.8e-2 (fused 7.4-8.0e-2) on the
        # long ragged clips, whose zero-padded rows sit at the ReLU boundary in bulk (tools/tfam_train_check.py)
        assert rel_l2 <= (1e-1 if ".ffn.0." in k else 4e-2), (k, rel_l2)
    print(f"{c['name']}: worst gradient rel L2 {worst[1]:.3e} ({worst[0]})")


@pytest.mark.parametrize("c", [CASES_ALL[0], CASES_ALL[3], CASES_ALL[5], CASES_ALL[7]], ids=lambda c: c["name"])
def test_fused_training_equals_per_op_path_with_dropout(c):
    """dropout 0.1 / mlp_dropout 0.3: the fused chains and the per-op path draw the same seeds in the same order and index their
    masks identically, so one step from the same weights gives the same loss and gradients up to 16-bit rounding points."""
    m = _model(c, dropout=0.1, mlp_dropout=0.3)
    out = []
    for fused in (False, True):
        m.set_dropout_seed(1234)
        out.append(_step(m, c, fused))
    (l0, y0, g0), (l1, y1, g1) = out
    err = (y0 - y1).abs().max().item()
    print(f"{c['name']}: per-op loss {l0:.6f}, fused {l1:.6f}, logits diff {err:.3e}")
    assert abs(l0 - l1) <= 5e-3 * abs(l0)
    assert err <= 1.6e-2 * max(1.0, y0.abs().max().item())
    assert set(g0) == set(g1)
    for k in g0:
        rel_l2 = ((g0[k] - g1[k]).norm() / (g0[k].norm() + 1e-20)).item()
        assert rel_l2 <= 5e-2, (k, rel_l2)
    # and dropout is really on: a different seed changes the result
    m.set_dropout_seed(99)
    l2, _, _ = _step(m, c, True)
    assert l2 != l1


def test_fused_training_is_deterministic_and_writes_into_the_arena():
    """Two runs of the same step are bit-identical (no atomics anywhere in the chains), and with a GradArena the gradients land in
    the arena slots (what FusedAdam and the data-parallel reducer read)."""
    from vimo_clip_amd.optim import GradArena
    c = CASES_ALL[0]
    m = _model(c, dropout=0.1, mlp_dropout=0.1)
    m.set_dropout_seed(5)
    l0, y0, g0 = _step(m, c, True)
    arena = GradArena(m.used_parameters())
    arena.flat_grad.fill_(float("nan"))
    m.set_dropout_seed(5)
    l1, y1, g1 = _step(m, c, True)
    assert l0 == l1 and torch.equal(y0, y1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    used = torch.zeros(arena.numel, dtype=torch.bool)
    for p, o in zip(arena.params, arena.offsets):
        used[o:o + p.numel()] = True
    assert torch.isfinite(arena.flat_grad.cpu()[used]).all()             # every used gradient element was written


def test_unsupported_shapes_take_the_per_op_path(monkeypatch):
    calls = _count_fused(monkeypatch)
    c = dict(next(x for x in mg.TFAM_CASES if x["name"] == "cross_long"), Tr=70, Tf=69)      # T = 70 > 64
    m = _model(c)
    loss, _, grads = _step(m, c, True)
    assert calls in ([], [False]) and np.isfinite(loss) and grads       # declined by AMO_CLIP._fused_inputs or by forward_train


@pytest.mark.parametrize("device_state", [False, True], ids=["host-scalars", "device-state"])
def test_backward_overlapped_adamw_equals_the_plain_step(device_state):
    """FusedAdam.enable_backward_overlap: AdamW (+ the 16-bit copy refresh) of a layer whose gradients are complete runs on a side
    stream beside the backward of the layers below it.  Five steps with dropout on must leave exactly the parameters, moments and
    16-bit copies of five plain steps (same kernels over sub-ranges of the same flat buffers: bit for bit)."""
    from vimo_clip_amd import autograd_ops as ag
    from vimo_clip_amd.losses import bce_with_logits_loss
    from vimo_clip_amd.optim import FusedAdam, GradArena
    c = dict(CASES_ALL[0], L=3)
    rgb, mot, mr, mf = (t.cuda() for t in mg.tfam_inputs(c))
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"]).cuda()
    runs = []
    for overlap in (False, True):
        m = _model(c, dropout=0.1, mlp_dropout=0.1)
        opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-3, weight_decay=0.1, decoupled=True)
        if device_state:
            opt.enable_device_state(base_seed=11)
            m.use_device_seeds(opt)
        else:
            m.set_dropout_seed(11)
        if overlap:
            opt.enable_backward_overlap(m.parameter_groups_by_layer())
            m.grad_group_callback = opt.group_ready
            assert len(opt._ov["ranges"]) == c["L"] + 1 and not opt._ov["rest"]
        losses = []
        for _ in range(5):
            if device_state:
                opt.tick()
            loss = bce_with_logits_loss(m(rgb, mot, mask_rgb=mr, mask_flow=mf), y)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        w16 = ag.weights.get(m.layers[1].ffn[0].weight, torch.bfloat16).clone()
        runs.append((losses, opt.arena.flat_param.clone(), opt.m.clone(), opt.v.clone(), w16, opt.step_count))
    a, b = runs
    assert a[0] == b[0] and a[5] == b[5] == 5
    for i in range(1, 5):
        assert torch.equal(a[i], b[i]), i
    assert a[0][0] != a[0][-1]
