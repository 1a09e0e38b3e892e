"""GPU: TFAM (AMO_CLIP), losses, student model and the training steps (HIP path through the C ABI) against
the golden fixtures recorded from the reference (losses.py, TFAM/models/AMO_CLIP.py) and the CPU oracle.

Tolerances: BASELINE.json asks for "TFAM logits within 1e-3 fp16": f16 compute -> |d| <= 1e-3 * max(1, max|ref|);
bf16 (3 fewer mantissa bits) -> 8e-3.  Losses are fp32 kernels: 1e-5 relative.
"""
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from oracle import student as ostudent
from oracle import tfam as otfam
from oracle import vit as ovit
from vimo_clip_amd import synth

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def _tfam(c, dtype, **extra):
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    kw = mg.tfam_mode_kwargs(c["mode"])
    m = AMO_CLIP(d_model=c["D"], nhead=c["H"], num_layers=c["L"], dim_feedforward=c["ff"], num_classes=c["C"], use_pe=c["pe"],
                 dropout=extra.pop("dropout", 0.0), mlp_dropout=extra.pop("mlp_dropout", 0.0), device="cuda", compute_dtype=dtype, **kw).cuda()
    m.load_state_dict(synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"]), strict=True)
    return m


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("c", mg.TFAM_CASES, ids=lambda c: c["name"])
def test_tfam_eval_logits_vs_reference(golden, c, dtype):
    m = _tfam(c, dtype).eval()
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    rgb_d, mot_d = rgb.cuda(), mot.cuda()
    with torch.no_grad():
        y = m(rgb_d, mot_d, mask_rgb=mr.cuda(), mask_flow=mf.cuda()).cpu()
    ref = torch.from_numpy(golden["tfam"][f"{c['name']}/logits"])
    err = (y - ref).abs().max().item()
    print(f"tfam {c['name']} {dtype}: max abs err {err:.3e} (|ref|max {ref.abs().max():.2f})")
    assert err <= TOL[dtype] * max(1.0, ref.abs().max().item())
    if c["pe"]:     # the reference adds the positional encoding in place to the caller's tensors (AMO_CLIP.py:133-134)
        torch.testing.assert_close(rgb_d.cpu(), rgb + otfam.positional_encoding(rgb.shape[1], c["D"]), atol=1e-5, rtol=0)


@pytest.mark.parametrize("name", ["cross_d512", "rgb_only", "concat_embed", "cross_d768_ragged_pe"])
def test_tfam_train_loss_and_grads_vs_reference(golden, name):
    from vimo_clip_amd.losses import bce_with_logits_loss
    c = next(x for x in mg.TFAM_CASES if x["name"] == name)
    # bf16: f16 gradients of a mean-reduced loss sit in the subnormal range without loss scaling
    m = _tfam(c, torch.bfloat16).train()
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"]).cuda()
    logits = m(rgb.cuda(), mot.cuda(), mask_rgb=mr.cuda(), mask_flow=mf.cuda())
    loss = bce_with_logits_loss(logits, y)
    loss.backward()
    g = golden["tfam"]
    assert abs(loss.item() - float(g[f"{name}/train_loss"])) <= 5e-3 * abs(float(g[f"{name}/train_loss"]))
    params = dict(m.named_parameters())
    for k in ("classifier.4.weight", "classifier.1.bias", "layers.0.ffn.0.bias", "layers.0.self_attn.in_proj_bias", "layers.0.norm_self.weight"):
        ref = torch.from_numpy(g[f"{name}/grad/{k}"])
        got = params[k].grad.cpu()
        denom = ref.abs().max().item() + 1e-12
        rel = (got - ref).abs().max().item() / denom
        rel_l2 = ((got - ref).norm() / (ref.norm() + 1e-20)).item()
        print(f"{name} grad {k}: rel-to-max err {rel:.3e}, rel L2 err {rel_l2:.3e}")
        # bf16 activations/gradients end to end vs the fp32 reference.  ffn.0 sits behind a ReLU: pre-activations
        # within rounding distance of 0 flip relu'(z) for single elements, so only the L2 bound is meaningful there
        assert rel_l2 <= 4e-2, (k, rel_l2)
        assert rel <= (2e-1 if "ffn.0" in k else 6e-2), (k, rel)
    used = {id(p) for p in m.used_parameters()}
    for n, p in m.named_parameters():
        assert (p.grad is not None) == (id(p) in used), n      # exactly the statically-known used set gets gradients


def test_tfam_dropout_train_mode_statistics():
    # dropout > 0: counter-based masks in the kernels; same seed -> same logits; mean logits close to eval
    c = mg.TFAM_CASES[0]
    m = _tfam(c, torch.float16, dropout=0.1, mlp_dropout=0.3).train()
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    args = (rgb.cuda(), mot.cuda())
    kw = dict(mask_rgb=mr.cuda(), mask_flow=mf.cuda())
    m.set_dropout_seed(7)
    a = m(*args, **kw)
    m.set_dropout_seed(7)
    b = m(*args, **kw)
    print("dropout repeat max diff", (a - b).abs().max().item())
    assert torch.equal(a, b)
    c2 = m(*args, **kw)
    assert not torch.equal(a, c2)
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.used_parameters())


@pytest.mark.parametrize("c", mg.LOSS_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("mode", ["cosine", "mse"])
def test_distillation_loss_vs_reference(golden, c, mode):
    from vimo_clip_amd.losses import distillation_loss
    s, t = mg.loss_inputs(c)
    s = s.cuda().requires_grad_(True)
    # the teacher slice of train.py:98: [B, T+1, E][:, :-1]
    t_full = torch.cat([t, torch.zeros(c["B"], 1, c["E"])], dim=1).cuda()
    loss = distillation_loss(s, t_full[:, :-1, :], mode=mode)
    (3.0 * loss).backward()
    g = golden["losses"]
    np.testing.assert_allclose(loss.item(), g[f"{c['name']}/{mode}/loss"], rtol=1e-5, atol=1e-7)
    ref = 3.0 * g[f"{c['name']}/{mode}/grad"]
    np.testing.assert_allclose(s.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
    with pytest.raises(ValueError, match="Unsupported mode"):
        distillation_loss(s, t_full[:, :-1, :], mode="l1")


@pytest.mark.parametrize("c", mg.BCE_CASES, ids=lambda c: c["name"])
def test_classification_loss_vs_reference(golden, c):
    from vimo_clip_amd.losses import classification_loss
    x, y = mg.bce_inputs(c)
    x = x.cuda().requires_grad_(True)
    loss = classification_loss(x, y.cuda(), positive_weight=c["pw"])
    loss.backward()
    g = golden["losses"]
    np.testing.assert_allclose(loss.item(), g[f"{c['name']}/loss"], rtol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{c['name']}/grad"], rtol=1e-4, atol=1e-8)


def _student(name, seed, dtype):
    from vimo_clip_amd.models import FlowStudentModel
    m = FlowStudentModel(name, device="cuda", num_classes=140, alpha=0.1, compute_dtype=dtype)
    sd = synth.student_state_dict(name, seed)
    m.load_state_dict(sd, strict=True)
    return m, sd


def test_student_forward_vs_reference_class_run(golden):
    """a5: FlowStudentModel.forward on the GPU against tests/golden/student.npz fwd/* -- outputs of the reference's own class
    (models/student_model.py:38-98, compiled from its AST, clip.load / Compose / to_pil_image bound to stand-ins:
    oracle/make_golden_student.py).  f16: 1e-3 * max(1, |ref|max) as the encoder tests."""
    from oracle import make_golden_student as mgs
    from vimo_clip_amd.models import FlowStudentModel
    for c in mgs.FWD_CASES:
        m = FlowStudentModel(c["model"], device="cuda", num_classes=c["C"], alpha=c["alpha"], compute_dtype=torch.float16)
        m.load_state_dict(synth.student_state_dict(c["model"], c["seed"], num_classes=c["C"]), strict=True)
        R = synth.VIT_GEOMETRY[c["model"]][0]
        vids = synth.randint_u8(c["seed"], "vids", (c["B"], c["T"], 3, R, R))
        with torch.no_grad():
            outs = m.eval()(vids.cuda())
        for got, key in zip(outs, ("emb", "emb_distill", "logits")):
            ref = torch.from_numpy(golden["student"][f"fwd/{c['name']}/{key}"])
            err = (got.float().cpu() - ref).abs().max().item()
            print(f"student fwd {c['name']} {key}: max abs err {err:.3e} (|ref|max {ref.abs().max():.2f})")
            assert got.shape == ref.shape and err <= 1e-3 * max(1.0, ref.abs().max().item()), (c["name"], key, err)


@pytest.mark.parametrize("name,B,T", [("ViT-tiny/32", 3, 5), ("ViT-B/32", 2, 4)])
def test_student_forward_vs_oracle(name, B, T):
    R, H = synth.VIT_GEOMETRY[name][0], synth.VIT_GEOMETRY[name][4]
    m, sd = _student(name, 61, torch.float16)
    vids = synth.randint_u8(61, "vids", (B, T, 3, R, R))
    with torch.no_grad():
        emb, emb_d, logits = m.eval()(vids.cuda())
    r_emb, r_emb_d, r_logits = ostudent.student_forward(sd, vids, H, alpha=0.1, wrap_quirk=True)
    for got, ref, nm in ((emb, r_emb, "emb"), (emb_d, r_emb_d, "emb_distill"), (logits, r_logits, "logits")):
        err = (got.cpu() - ref).abs().max().item()
        print(f"student {name} {nm}: max abs err {err:.3e} scale {ref.abs().max():.2f}")
        assert got.shape == ref.shape and err <= 2e-3 * max(1.0, ref.abs().max().item())
    # float inputs holding 0..255 values take the same path as u8 (student_model.py:74)
    with torch.no_grad():
        emb2, _, _ = m(vids.float().cuda())
    assert torch.equal(emb, emb2)


def test_student_train_step_vs_oracle_autograd():
    """One full student step on ViT-tiny/32: loss = cosine distill + BCE(pos_weight 9) (train.py:95-107);
    gradients of every parameter vs torch autograd through the fp32 oracle; then Adam vs the oracle update."""
    from vimo_clip_amd.losses import classification_loss, distillation_loss
    from vimo_clip_amd.optim import FusedAdam, GradArena
    name, B, T = "ViT-tiny/32", 4, 5
    R, H = synth.VIT_GEOMETRY[name][0], synth.VIT_GEOMETRY[name][4]
    m, sd = _student(name, 71, torch.bfloat16)
    m.train()
    vids = synth.randint_u8(71, "vids", (B, T, 3, R, R))
    teacher = synth.normal(71, "teacher", (B, T + 1, 64))
    labels = synth.multi_hot_labels(71, "labels", B, 140)
    arena = GradArena(m.parameters())
    opt = FusedAdam(arena, lr=1e-3)
    emb, emb_d, logits = m(vids.cuda())
    loss = distillation_loss(emb_d, teacher.cuda()[:, :-1, :], mode="cosine") + classification_loss(logits, labels.cuda(), positive_weight=9)
    loss.backward()
    # ---- oracle ----
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    _, oe_d, ol = ostudent.student_forward(sdo, vids, H, alpha=0.1, wrap_quirk=True)
    oloss = ostudent.distillation_loss(oe_d, teacher[:, :-1, :], "cosine") + ostudent.classification_loss(ol, labels, 9)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) <= 1e-2 * abs(oloss.item())
    worst = 0.0
    for k, p in m.named_parameters():
        ref = sdo[k].grad
        got = p.grad.cpu()
        rel = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        worst = max(worst, rel)
        assert rel <= 8e-2, (k, rel)
    print(f"student step: loss {loss.item():.5f} vs {oloss.item():.5f}; worst grad rel-to-max err {worst:.3e}")
    # ---- Adam step (train.py:66,107) with the HIP gradients on both sides ----
    before = {k: p.detach().cpu().clone() for k, p in m.named_parameters()}
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    opt.step()
    for k, p in m.named_parameters():
        want, _, _ = ostudent.adam_step(before[k], grads[k], torch.zeros_like(before[k]), torch.zeros_like(before[k]), 1, 1e-3)
        torch.testing.assert_close(p.detach().cpu(), want, atol=1e-7, rtol=1e-5)
    # the 16-bit weight copies were invalidated: a second forward sees the updated weights
    emb2, _, _ = m(vids.cuda())
    assert not torch.equal(emb2, emb)


def test_adamw_matches_torch():
    from vimo_clip_amd.optim import FusedAdam, GradArena
    ps = [torch.nn.Parameter(synth.normal(5, f"p{i}", s).cuda()) for i, s in enumerate([(33, 7), (129,), (64, 64)])]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    arena = GradArena(ps)
    opt = FusedAdam(arena, lr=1e-4, weight_decay=0.1, decoupled=True)
    topt = torch.optim.AdamW(ref, lr=1e-4, weight_decay=0.1)
    for step in range(3):
        for i, (p, r) in enumerate(zip(ps, ref)):
            g = synth.normal(step, f"g{i}", tuple(p.shape))
            p.grad.copy_(g.cuda())
            r.grad = g.clone()
        opt.step()
        topt.step()
    for p, r in zip(ps, ref):
        torch.testing.assert_close(p.detach().cpu(), r.detach(), atol=1e-7, rtol=2e-6)
    assert abs(arena.grad_norm().item() - torch.cat([r.grad.reshape(-1) for r in ref]).norm().item()) < 1e-3


@pytest.mark.parametrize("B,H,Tq,Tk,dh,masked", [(3, 8, 16, 15, 96, True), (2, 8, 16, 16, 64, False), (2, 2, 50, 50, 64, False),
                                                  (2, 3, 197, 197, 64, False), (1, 2, 257, 257, 64, False), (2, 8, 40, 39, 96, True),
                                                  (1, 1, 300, 300, 64, False)])     # last: does not fit LDS -> generic fp32 kernels
def test_attention_backward_vs_torch(B, H, Tq, Tk, dh, masked):
    from vimo_clip_amd import autograd_ops as ag
    D = H * dh
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B * Tq, D, generator=g).to(torch.float16)
    kv = torch.randn(B * Tk, 2 * D, generator=g).to(torch.float16)
    do = torch.randn(B * Tq, D, generator=g).to(torch.float16)
    mask = None
    if masked:
        lens = torch.randint(1, Tk + 1, (B,), generator=g)
        mask = (torch.arange(Tk)[None] < lens[:, None])
    qd, kvd = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
    out = ag.CrossAttnFn.apply(qd, kvd, mask.to(torch.uint8).cuda() if masked else None, B, Tq, Tk, H)
    out.backward(do.cuda())
    qf = q.float().requires_grad_(True)
    kvf = kv.float().requires_grad_(True)
    qh = qf.view(B, Tq, H, dh).transpose(1, 2)
    kh = kvf[:, :D].reshape(B, Tk, H, dh).transpose(1, 2)
    vh = kvf[:, D:].reshape(B, Tk, H, dh).transpose(1, 2)
    s = (qh * dh ** -0.5) @ kh.transpose(-1, -2)
    if masked:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Tq, D)
    ref.backward(do.float())
    torch.testing.assert_close(out.float().cpu(), ref.detach(), atol=3e-3, rtol=3e-3)
    torch.testing.assert_close(qd.grad.float().cpu(), qf.grad, atol=5e-3, rtol=2e-2)
    torch.testing.assert_close(kvd.grad.float().cpu(), kvf.grad, atol=5e-3, rtol=2e-2)


@pytest.mark.parametrize("rows,C", [(8, 12), (37, 173), (512, 140), (3, 1000)])
def test_cross_entropy_loss_vs_oracle(rows, C):
    # MammalNet variants: index targets (train_frame_diff_mn.py:102) and float one-hot / soft rows (TFAM ..._MN.py:83)
    from vimo_clip_amd.losses import cross_entropy_loss
    x = synth.normal(61, f"ce_x{rows}", (rows, C)) * 5.0
    x[0, 0] = 80.0                                                      # large-logit row: max-subtraction path
    t = synth.randint(61, f"ce_t{rows}", (rows,), 0, C)
    y = torch.nn.functional.one_hot(t, C).float()
    y[-1] = torch.softmax(synth.normal(61, "soft", (C,)), 0)
    for tgt in (t, y):
        xr = x.clone().requires_grad_(True)
        ref = ostudent.cross_entropy_loss(xr, tgt)
        ref.backward()
        xg = x.cuda().requires_grad_(True)
        got = cross_entropy_loss(xg, tgt.cuda())
        (got * 2.0).backward()                                          # upstream gradient is applied
        assert abs(got.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
        assert (xg.grad.cpu() - 2.0 * xr.grad).abs().max().item() <= 2e-7
    with pytest.raises(ValueError):
        cross_entropy_loss(x.cuda(), torch.zeros(rows + 1, dtype=torch.int64, device="cuda"))


def test_tfam_single_label_training_learns():
    # TFAM/train_and_eval_frame_diff_MN.py:41-131: CrossEntropy + Accuracy, frame_diff batch keys; the loop must learn
    # class-dependent synthetic embeddings (accuracy far above chance) and the YAML/run() plumbing must hold together
    from vimo_clip_amd.TFAM.train_and_eval import Config, run
    cfg = Config(task="singlelabel", motion_key="frame_diff", num_classes=10, d_model=128, nhead=4, num_layers=1, dim_feedforward=256,
                 epochs=3, batch_size=16, dropout=0.0, mlp_dropout=0.0, device="cuda:0", mode="both", checkpoint_dir=None)
    res = run(cfg, limit=512)
    assert res["task"] == "singlelabel" and res["best_val_metric"] > 0.5 and res["test_metric"] > 0.5, res


def test_full_size_tfam_batch_properties():
    """BASELINE.json TFAM configuration at full size (D = 768, 4 layers, 16 + 15 tokens, B = 4096 clips, ragged masks):
    clips are independent -- a 4096-clip pass, its 512-clip slices and a shuffled pass give the same logits bit for bit --
    and a 4-clip sample agrees with the CPU oracle within the bf16 tolerance."""
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    D, H, L, FF, C, seed, B = 768, 8, 4, 2048, 140, 4, 4096
    m = AMO_CLIP(d_model=D, nhead=H, num_layers=L, dim_feedforward=FF, num_classes=C, dropout=0.0, mlp_dropout=0.0, device="cuda").cuda().eval()
    sd = synth.tfam_state_dict(D, H, L, FF, C, seed)
    m.load_state_dict(sd, strict=True)
    rgb, mot = synth.normal(71, "rgb", (B, 16, D)), synth.normal(71, "mot", (B, 15, D))
    lr, lf = synth.randint(71, "lr", (B,), 3, 17), synth.randint(71, "lf", (B,), 2, 16)
    mr, mf = torch.arange(16)[None] < lr[:, None], torch.arange(15)[None] < lf[:, None]
    d = lambda *t: [x.cuda() for x in t]
    with torch.no_grad():
        R, M, MR, MF = d(rgb, mot, mr, mf)
        full = m(R, M, mask_rgb=MR, mask_flow=MF)
        assert full.shape == (B, C) and torch.isfinite(full).all()
        for s in (0, 1536, 3584):
            part = m(R[s:s + 512], M[s:s + 512], mask_rgb=MR[s:s + 512], mask_flow=MF[s:s + 512])
            assert torch.equal(part, full[s:s + 512])
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
        assert torch.equal(m(R[perm], M[perm], mask_rgb=MR[perm], mask_flow=MF[perm]), full[perm])
    pick = [0, 1000, 2047, 4095]
    ref = otfam.amo_clip_forward(sd, rgb[pick], mot[pick], mr[pick], mf[pick], nhead=H)
    err = (full[pick].cpu() - ref).abs().max().item()
    assert err <= TOL[torch.bfloat16] * max(1.0, ref.abs().max().item()), err


def test_tfam_graphed_eval_matches_eager_and_is_reused():
    # hipGraph replay of the evaluation forward: same kernels, same bits as the eager forward; one graph per batch shape
    from vimo_clip_amd.TFAM.data.dataset import SyntheticEmbeddingDataset, collate_fn_pad
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    from vimo_clip_amd.TFAM.train_and_eval import Config, GraphedEvalForward, ModelTester
    cfg = Config(num_classes=24, d_model=256, nhead=4, num_layers=2, dim_feedforward=512, batch_size=8, dropout=0.0, mlp_dropout=0.0,
                 device="cuda:0", use_graphs=True)
    m = AMO_CLIP(d_model=256, nhead=4, num_layers=2, dim_feedforward=512, num_classes=24, dropout=0.0, mlp_dropout=0.0, device="cuda").cuda().eval()
    m.load_state_dict(synth.tfam_state_dict(256, 4, 2, 512, 24, 3), strict=True)
    fixed = SyntheticEmbeddingDataset(synth.multi_hot_labels(3, "y", 64, 24), 256, tmin=16, tmax=16, seed=9)      # num_frames=16 loaders
    ragged = SyntheticEmbeddingDataset(synth.multi_hot_labels(3, "y", 64, 24), 256, tmin=5, tmax=40, seed=9)
    for ds, max_graphs in ((fixed, 1), (ragged, 8)):
        gf = GraphedEvalForward(m, cfg)
        for s in range(0, 64, 8):
            batch = collate_fn_pad([ds[i] for i in range(s, s + 8)])
            with torch.no_grad():
                ref = m(batch["embeddings"].cuda(), batch["flow_embeddings"].cuda(), mask_rgb=batch["mask_rgb"].cuda(), mask_flow=batch["mask_flow"].cuda())
            assert torch.equal(gf(batch), ref)
        assert 1 <= len(gf._graphs) <= max_graphs * len(gf._streams)      # one graph per (slot, shape)
        # two batches in flight (launch k+1 before consuming k): same logits, same order
        bs = [collate_fn_pad([ds[i] for i in range(s, s + 8)]) for s in range(0, 64, 8)]
        with torch.no_grad():
            refs = [m(b["embeddings"].cuda(), b["flow_embeddings"].cuda(), mask_rgb=b["mask_rgb"].cuda(), mask_flow=b["mask_flow"].cuda()) for b in bs]
        got = list(GraphedEvalForward(m, cfg).pipelined(iter(bs)))
        assert len(got) == len(bs) and all(g_[0] is b for g_, b in zip(got, bs))
        assert all(torch.equal(g_[1], r) for g_, r in zip(got, refs))
    mAP_g, _ = ModelTester(m, fixed, cfg).evaluate()
    cfg.use_graphs = False
    mAP_e, _ = ModelTester(m, fixed, cfg).evaluate()
    assert mAP_g == mAP_e


def test_tfam_fused_dropout_tail_matches_unfused_path():
    # dropout masks applied inside the add + LayerNorm kernel (forward) and regenerated in its backward must give the loss and
    # the gradients of the separate dropout -> add -> LayerNorm path (same seeds => same masks); differences are roundings only
    from vimo_clip_amd.losses import bce_with_logits_loss
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    rgb, mot = synth.normal(81, "r", (6, 9, 256)).cuda(), synth.normal(81, "m", (6, 8, 256)).cuda()
    y = synth.multi_hot_labels(81, "y", 6, 24).cuda()
    res = []
    for fuse in (True, False):
        m = AMO_CLIP(d_model=256, nhead=4, num_layers=2, dim_feedforward=512, num_classes=24, dropout=0.3, mlp_dropout=0.2, device="cuda").cuda().train()
        m.load_state_dict(synth.tfam_state_dict(256, 4, 2, 512, 24, 6), strict=True)
        for layer in m.layers:
            layer.fuse_tail = fuse
        m.set_dropout_seed(1234)
        loss = bce_with_logits_loss(m(rgb, mot), y)
        loss.backward()
        res.append((loss.item(), {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}))
    (l1, g1), (l0, g0) = res
    assert abs(l1 - l0) <= 2e-3 * abs(l0), (l1, l0)
    assert set(g1) == set(g0)
    for k in g0:
        num, den = (g1[k] - g0[k]).norm().item(), g0[k].norm().item()
        tol = 6e-2 if ".ffn.0." in k else 3e-2       # a few ReLU gates flip when an intermediate is rounded differently
        assert num <= tol * den + 1e-6, (k, num, den)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("drops", [((0.0, 0), (0.0, 0)), ((0.3, 77), (0.0, 0)), ((0.1, 5), (0.2, 9))])
@pytest.mark.parametrize("rows,D,two", [(128, 768, True), (8192, 768, False), (77, 512, True), (1000, 256, True)])
def test_postnorm_backward_in_one_launch_equals_the_two_pass_backward(rows, D, two, drops, dtype):
    """vmc_postnorm_bwd (LayerNorm backward + the branch gradient through the regenerated dropout masks, cast to 16 bits, from one store
    loop) == vmc_layernorm_bwd2 followed by vmc_cast_dropout2 / the plain cast, bit for bit: d sum, d branch, d gamma, d beta."""
    from vimo_clip_amd import autograd_ops as ag
    g = torch.Generator().manual_seed(rows + D)
    x32 = torch.randn(rows, D, generator=g).cuda()
    br = torch.randn(rows, D, generator=g).to(dtype).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    dy32 = torch.randn(rows, D, generator=g).cuda()
    dy16 = torch.randn(rows, D, generator=g).to(dtype).cuda()
    out = {}
    try:
        for fuse in (True, False):
            ag.FUSE_POSTNORM_BWD = fuse
            xx, bb = x32.clone().requires_grad_(True), br.clone().requires_grad_(True)
            gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            y32, y16 = ag.PostNormFn.apply(xx, bb, gm, bt, True, drops)
            torch.autograd.backward([y32, y16] if two else [y32], [dy32, dy16] if two else [dy32])
            out[fuse] = (xx.grad.clone(), bb.grad.clone(), gm.grad.clone(), bt.grad.clone())
    finally:
        ag.FUSE_POSTNORM_BWD = True
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
    assert out[True][1].dtype == dtype and bool(torch.isfinite(out[True][1].float()).all())
    if drops[0][0] > 0:      # the masks really are applied: dropped elements carry no gradient
        assert float((out[True][1] == 0).float().mean()) >= 0.5 * drops[0][0]


@pytest.mark.parametrize("c", __import__("oracle.make_golden_student", fromlist=["MLP_CASES"]).MLP_CASES, ids=lambda c: c["name"])
def test_residual_mlp_vs_reference_class(golden, c):
    """HIP ResidualMLP against outputs + gradients of the reference class itself (models/student_model.py:8-35, compiled from
    its AST by oracle/make_golden_student.py): forward f16 1e-3, bf16 8e-3 (x scale); fresh module = exact identity (zero fc2)."""
    from oracle import make_golden_student as mgs
    from vimo_clip_amd.models.student_model import ResidualMLP
    g = golden["student"]
    x, w1, b1, w2, b2 = mgs.mlp_inputs(c)
    ref = torch.from_numpy(g[f"mlp/{c['name']}/y"])
    for dtype, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        m = ResidualMLP(c["E"], alpha=c["alpha"], compute_dtype=dtype).cuda()
        assert float(m.fc2.weight.detach().abs().max()) == 0.0 and float(m.fc2.bias.detach().abs().max()) == 0.0
        with torch.no_grad():
            y0 = m(x.cuda())
            assert torch.equal(y0.cpu().float(), x) or (y0.cpu().float() - x).abs().max() <= tol * x.abs().max()
            m.fc1.weight.copy_(w1); m.fc1.bias.copy_(b1); m.fc2.weight.copy_(w2); m.fc2.bias.copy_(b2)
        xr = x.cuda().requires_grad_(True)
        y = m(xr)
        err = (y.detach().cpu().float() - ref).abs().max().item()
        print(f"residual_mlp {c['name']} {dtype}: max abs err {err:.3e} (|ref|max {ref.abs().max():.2f})")
        assert err <= tol * max(1.0, ref.abs().max().item())
        if dtype == torch.bfloat16:
            gup = synth.normal(c["seed"], "g", tuple(ref.shape)).cuda()
            (y.float() * gup).sum().backward()
            dx_ref = torch.from_numpy(g[f"mlp/{c['name']}/dx"])
            rel = ((xr.grad.cpu().float() - dx_ref).norm() / dx_ref.norm()).item()
            dw_ref = torch.from_numpy(g[f"mlp/{c['name']}/dfc1w"])
            relw = ((m.fc1.weight.grad.cpu()[:8] - dw_ref).norm() / dw_ref.norm()).item()
            print(f"   backward: dx rel L2 {rel:.3e}, dfc1.weight rel L2 {relw:.3e}")
            assert rel <= 2e-2 and relw <= 3e-2


def test_distillation_loss_strided_teacher_in_other_dtypes():
    """teacher = rgb_emb[:, :-1] stored as f16 / f64 -- strides must be taken from the tensor the kernel gets."""
    from vimo_clip_amd.losses import distillation_loss
    B, T, E = 3, 5, 64
    s = synth.normal(5, "s", (B, T, E)).cuda().requires_grad_(True)
    t_full = synth.normal(5, "t", (B, T + 1, E))
    ref = ostudent.distillation_loss(s.detach().cpu(), t_full[:, :-1], "cosine")
    for dt_ in (torch.float32, torch.float64, torch.float16):
        tt = t_full.to(dt_).cuda()
        loss = distillation_loss(s, tt[:, :-1, :], mode="cosine")
        tol = 2e-3 if dt_ == torch.float16 else 1e-5
        assert abs(loss.item() - float(ref)) <= tol * abs(float(ref)), (dt_, loss.item(), float(ref))


def test_fused_adam_grad_clip_matches_torch():
    from vimo_clip_amd.optim import FusedAdam, GradArena
    ps = [torch.nn.Parameter(synth.normal(8, f"p{i}", sh).cuda()) for i, sh in enumerate([(64, 32), (32,), (8, 8)])]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    arena = GradArena(ps)
    opt = FusedAdam(arena, lr=1e-2)
    topt = torch.optim.Adam(ref, lr=1e-2)
    for step in range(3):
        for i, (p, r) in enumerate(zip(ps, ref)):
            g = synth.normal(9 + step, f"g{i}", tuple(p.shape)).cuda() * 3.0
            p._vmc_grad.copy_(g)
            r.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref, 0.7)
        topt.step()
        opt.step(max_grad_norm=0.7)
    for p, r in zip(ps, ref):
        assert (p.detach() - r.detach()).abs().max().item() <= 2e-6 * max(1.0, r.abs().max().item())


def test_tfam_best_checkpoint_round_trip(tmp_path):
    """train() writes best_model.pth (reference dict layout, module.-prefixed keys), the tester reloads it."""
    from vimo_clip_amd.TFAM import train_and_eval as te
    cfg = te.Config(epochs=2, batch_size=8, d_model=512, num_layers=1, dropout=0.0, mlp_dropout=0.0, checkpoint_dir=str(tmp_path), mode="both")
    res = te.run(cfg, limit=64)
    ck = torch.load(str(tmp_path / "best_model.pth"), weights_only=True)
    assert set(ck) == {"epoch", "state_dict", "optimizer", "scheduler", "best_val_loss", "best_val_mAP"}
    assert all(k.startswith("module.") for k in ck["state_dict"])
    assert abs(ck["best_val_mAP"] - res["best_val_metric"]) < 1e-9
    m2 = te.build_model(cfg)
    te.ModelTester(m2, None, cfg).load_best_model(str(tmp_path))
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), ck["state_dict"]["module." + k]), k


def test_student_full_size_step_properties_vit_b32():
    """BASELINE.json configs[2] at full size (ViT-B/32, 32 clips x 16 flow frames 224^2 per GPU: the bench's student leg), through
    size-independent properties since the CPU oracle needs minutes for 512 frames:
      (a) the loss is finite and every parameter gets a finite gradient;
      (b) batch-split invariance: loss and gradients of the 32-clip batch equal the mean of those of its two 16-clip halves
          (the losses are means over rows: train.py:98-100) although the GEMMs then take different tile paths;
      (c) a 3-clip sample of the same batch: embeddings, loss and gradients against torch autograd through the fp32 oracle."""
    from vimo_clip_amd.losses import classification_loss, distillation_loss
    name, B, T, seed = "ViT-B/32", 32, 16, 83
    R, H, E = synth.VIT_GEOMETRY[name][0], synth.VIT_GEOMETRY[name][4], synth.VIT_GEOMETRY[name][5]
    m, sd = _student(name, seed, torch.bfloat16)
    m.train()
    vids = synth.randint_u8(seed, "vids", (B, T, 3, R, R)).cuda()
    teacher = synth.normal(seed, "teacher", (B, T + 1, E)).cuda()
    labels = synth.multi_hot_labels(seed, "labels", B, 140).cuda()
    watch = ["classification_head.2.weight", "residual_mlp.fc1.weight", "visual_encoder.proj", "visual_encoder.transformer.resblocks.5.attn.in_proj_weight",
             "visual_encoder.transformer.resblocks.0.mlp.c_fc.bias", "visual_encoder.conv1.weight", "visual_encoder.ln_pre.weight"]

    def step(sl):
        for p in m.parameters():
            p.grad = None
        emb, emb_d, logits = m(vids[sl])
        loss = distillation_loss(emb_d, teacher[sl][:, :-1, :], mode="cosine") + classification_loss(logits, labels[sl], positive_weight=9)
        loss.backward()
        params = dict(m.named_parameters())
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params.values())
        return loss.item(), {k: params[k].grad.detach().float().clone() for k in watch}, emb.detach()

    full_loss, full_g, full_emb = step(slice(0, B))
    l1, g1, _ = step(slice(0, B // 2))
    l2, g2, _ = step(slice(B // 2, B))
    assert np.isfinite(full_loss)
    assert abs(full_loss - 0.5 * (l1 + l2)) <= 2e-4 * abs(full_loss), (full_loss, l1, l2)
    for k in watch:
        avg = 0.5 * (g1[k] + g2[k])
        rel = ((full_g[k] - avg).norm() / (avg.norm() + 1e-20)).item()
        print(f"split invariance {k}: rel L2 {rel:.2e}")
        assert rel <= 1e-5, (k, rel)          # same arithmetic per clip whatever the batch split: only fp32 summation order differs (measured <= 3.5e-7)
    # (c) three clips of the batch against the oracle (forward embeddings of the FULL pass, then a 3-clip step)
    pick = [0, 15, 31]
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oe, oe_d, ol = ostudent.student_forward(sdo, vids[pick].cpu(), H, alpha=0.1, wrap_quirk=True)
    err = (full_emb[pick].cpu().float() - oe.detach()).abs().max().item()
    print(f"full-batch embeddings of clips {pick} vs oracle: max abs err {err:.3e} (|ref|max {oe.abs().max().item():.2f})")
    assert err <= 8e-3 * max(1.0, oe.abs().max().item())
    idx = torch.tensor(pick).cuda()
    for p in m.parameters():
        p.grad = None
    emb, emb_d, logits = m(vids[idx])
    loss = distillation_loss(emb_d, teacher[idx][:, :-1, :], mode="cosine") + classification_loss(logits, labels[idx], positive_weight=9)
    loss.backward()
    oloss = ostudent.distillation_loss(oe_d, teacher[idx].cpu()[:, :-1, :], "cosine") + ostudent.classification_loss(ol, labels[idx].cpu(), 9)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) <= 1e-2 * abs(oloss.item()), (loss.item(), oloss.item())
    params = dict(m.named_parameters())
    for k in watch:
        ref, got = sdo[k].grad, params[k].grad.cpu().float()
        rel = ((got - ref).norm() / (ref.norm() + 1e-20)).item()
        print(f"3-clip step grad {k}: rel L2 {rel:.2e}")
        assert rel <= 2e-2, (k, rel)          # measured 3.5e-3 .. 6.8e-3 (bf16 activations end to end)


def test_student_last_block_on_class_rows_trains_the_same_function():
    """Forward + backward with the last block's out_proj / MLP on the class rows only == the full-width last block (same loss,
    same gradients up to fp32 summation order in the smaller wgrad GEMMs)."""
    from vimo_clip_amd.losses import classification_loss, distillation_loss
    name, B, T, seed = "ViT-tiny/32", 4, 6, 91
    R, E = synth.VIT_GEOMETRY[name][0], synth.VIT_GEOMETRY[name][5]
    vids = synth.randint_u8(seed, "vids", (B, T, 3, R, R)).cuda()
    teacher = synth.normal(seed, "teacher", (B, T + 1, E)).cuda()
    labels = synth.multi_hot_labels(seed, "labels", B, 140).cuda()
    res = []
    for cls_only in (True, False):
        m, _ = _student(name, seed, torch.bfloat16)
        m.train()
        m.visual_encoder.cls_only_last_block = cls_only
        emb, emb_d, logits = m(vids)
        loss = distillation_loss(emb_d, teacher[:, :-1, :], mode="cosine") + classification_loss(logits, labels, positive_weight=9)
        loss.backward()
        res.append((loss.item(), {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0]), (res[0][0], res[1][0])
    worst = 0.0
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
        worst = max(worst, rel)
        assert rel <= 1e-3, (k, rel)
    print(f"class-rows-only last block: loss {res[0][0]:.6f} vs {res[1][0]:.6f}, worst gradient rel L2 diff {worst:.2e}")


def test_device_state_adam_and_tick_match_host_adam():
    """vmc_train_tick + vmc_adam_step_dev (step count, bias corrections, lr from device memory) == vmc_adam_step over 5 steps."""
    from vimo_clip_amd.optim import FusedAdam, GradArena
    mk = lambda: [torch.nn.Parameter(synth.normal(8, f"p{i}", sh).cuda()) for i, sh in enumerate([(64, 32), (33,), (8, 8)])]
    pa, pb = mk(), mk()
    oa = FusedAdam(GradArena(pa), lr=3e-3, weight_decay=0.1, decoupled=True)
    ob = FusedAdam(GradArena(pb), lr=3e-3, weight_decay=0.1, decoupled=True).enable_device_state(base_seed=7)
    for step in range(5):
        if step == 3:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 1e-3
            ob.sync_hyper()
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = synth.normal(9 + step, f"g{i}", tuple(a.shape)).cuda()
            a._vmc_grad.copy_(g)
            b._vmc_grad.copy_(g)
        oa.step()
        ob.tick()
        ob.step()
    assert int(ob.dev_state[0].item()) == 5 == ob.step_count
    for a, b in zip(pa, pb):
        assert (a.detach() - b.detach()).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())
    seeds = ob.dev_state[2:10].tolist()
    assert len(set(seeds)) == 8 and all(0 <= s_ < (1 << 63) for s_ in seeds)
    ob.tick()
    assert ob.dev_state[2:10].tolist() != seeds                # every step draws new dropout seeds


def test_grouped_weight_gradients_equal_the_per_linear_launches():
    """autograd_ops.wgrad_queue: the weight gradients of a backward pass parked and launched as groups (vmc_linear_wgrad_tn_group,
    every tile over all tokens) against one sliced launch + reduce per linear -- same arena, every parameter; also with a
    gradient-ready hook registered (each parameter reported as often as without grouping)."""
    from vimo_clip_amd import autograd_ops as ag
    from vimo_clip_amd.losses import bce_with_logits_loss
    from vimo_clip_amd.optim import GradArena
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    B, T, D = 64, 16, 768
    m = AMO_CLIP(d_model=D, nhead=8, num_layers=2, dim_feedforward=2048, num_classes=140, dropout=0.0, mlp_dropout=0.0, device="cuda").cuda().train()
    m.load_state_dict(synth.tfam_state_dict(D, 8, 2, 2048, 140, 5), strict=True)
    arena = GradArena(m.used_parameters())
    rgb, mot = synth.normal(41, "rgb", (B, T, D)).cuda(), synth.normal(41, "mot", (B, T, D)).cuda()
    mk = torch.ones(B, T, dtype=torch.bool, device="cuda")
    y = synth.multi_hot_labels(41, "lab", B, 140).cuda()
    grads, reports = {}, {}
    was = ag.wgrad_queue.enabled
    try:
        for mode in ("single", "grouped", "grouped+hook"):
            ag.wgrad_queue.enabled = mode != "single"
            seen = {}
            hook = lambda p: seen.__setitem__(id(p), seen.get(id(p), 0) + 1)      # noqa: E731
            if mode != "single":
                ag.grad_ready_hooks.append(hook)
            if mode == "grouped":
                ag.grad_ready_hooks.remove(hook)
            arena.flat_grad.fill_(123.0)
            bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mk), y).backward()
            assert not ag.wgrad_queue.items                     # flushed by the end-of-backward callback
            torch.cuda.synchronize()
            grads[mode] = arena.flat_grad.clone()
            reports[mode] = seen
            if hook in ag.grad_ready_hooks:
                ag.grad_ready_hooks.remove(hook)
    finally:
        ag.wgrad_queue.enabled = was
    ref = grads["single"]
    assert not (ref == 123.0).all()
    for mode in ("grouped", "grouped+hook"):
        for p, o in zip(arena.params, arena.offsets):
            a, b = ref[o:o + p.numel()], grads[mode][o:o + p.numel()]
            assert (a - b).abs().max().item() <= 2e-4 * max(1e-6, a.abs().max().item()), (mode, tuple(p.shape))
    ag.wgrad_queue.enabled = False
    seen = {}
    hook = lambda p: seen.__setitem__(id(p), seen.get(id(p), 0) + 1)              # noqa: E731
    ag.grad_ready_hooks.append(hook)
    try:
        bce_with_logits_loss(m(rgb, mot, mask_rgb=mk, mask_flow=mk), y).backward()
    finally:
        ag.grad_ready_hooks.remove(hook)
        ag.wgrad_queue.enabled = was
    assert seen == reports["grouped+hook"]                      # the reducer's per-parameter report counts are unchanged


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_fused_adam_and_copy_refresh_equals_the_two_kernel_step(dtype):
    """vmc_adam_cast_multi (AdamW + refresh of both 16-bit copies from the registers, one pass over the masters) against
    vmc_adam_step_dev + vmc_cast_weights_multi: masters, moments and every compute copy bit for bit over 4 steps -- matrices with
    both / one / no cached copy, ragged and non-multiple-of-4 shapes, 1-D parameters."""
    from vimo_clip_amd import autograd_ops as ag
    from vimo_clip_amd.optim import FusedAdam, GradArena
    shapes = [(192, 128), (100, 72), (130, 50), (33,), (64, 64), (7, 9), (256,), (96, 3, 4, 4)]

    def build(fused):
        ps = [torch.nn.Parameter(synth.normal(8, f"p{i}", sh).cuda()) for i, sh in enumerate(shapes)]
        opt = FusedAdam(GradArena(ps), lr=3e-3, weight_decay=0.1, decoupled=True).enable_device_state(base_seed=7)
        opt.FUSED_CAST = fused
        copies = {}
        for i, p in enumerate(ps):
            if p.dim() < 2 or i == 4:
                continue                                      # 1-D parameters and one matrix have no compute copy
            if i == 2:
                copies[i] = (ag.weights.get(p, dtype, transposed=False, pad_k=p.shape[1] % 64 != 0),)
            elif p.dim() == 2:
                copies[i] = (ag.weights.get(p, dtype, transposed=False, pad_k=p.shape[1] % 64 != 0, both=True),
                             ag.weights.get(p, dtype, transposed=True, pad_k=p.shape[0] % 64 != 0, both=True))
        return ps, opt, copies

    ag.weights.clear()
    (pa, oa, ca), (pb, ob, cb) = build(False), build(True)
    for step in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = synth.normal(20 + step, f"g{i}", tuple(a.shape)).cuda()
            a._vmc_grad.copy_(g)
            b._vmc_grad.copy_(g)
        for o in (oa, ob):
            o.tick()
            o.step()
    assert getattr(ob, "_fused_plan", None) is not None and getattr(oa, "_fused_plan", None) is None
    assert torch.equal(oa.arena.flat_param, ob.arena.flat_param) and torch.equal(oa.m, ob.m) and torch.equal(oa.v, ob.v)
    assert not torch.equal(oa.arena.flat_param, torch.zeros_like(oa.arena.flat_param))
    for i in ca:
        for x, y in zip(ca[i], cb[i]):
            assert torch.equal(x, y), i
        w = ca[i][0]
        ref = pa[i].detach().to(dtype)
        assert torch.equal(w[:, :pa[i].shape[1]], ref)        # the copies are the rounded masters
    ag.weights.clear()


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_captured_train_steps_equal_eager_device_state_steps(p_drop):
    """hipGraph-captured TFAM training steps (tick + fwd + bwd + AdamW): five replays follow the same trajectory as five eager
    steps in the same device-state mode -- also with dropout, whose masks must change from step to step inside ONE graph."""
    from vimo_clip_amd.graphs import GraphedTrainStep
    from vimo_clip_amd.losses import bce_with_logits_loss
    from vimo_clip_amd.optim import FusedAdam, GradArena
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    c = mg.TFAM_CASES[0]
    rgb, mot, mr, mf = (t.cuda() for t in mg.tfam_inputs(c))
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"]).cuda()
    traj = []
    for captured in (False, True):
        m = AMO_CLIP(d_model=c["D"], nhead=c["H"], num_layers=c["L"], dim_feedforward=c["ff"], num_classes=c["C"], dropout=p_drop,
                     mlp_dropout=p_drop, device="cuda", **mg.tfam_mode_kwargs(c["mode"])).cuda().train()
        m.load_state_dict(synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"]), strict=True)
        opt = FusedAdam(GradArena(m.used_parameters()), lr=1e-3, weight_decay=0.1, decoupled=True).enable_device_state(base_seed=11)
        m.use_device_seeds(opt)

        def step(a, b, cm, d, yy):
            opt.tick()
            out = m(a, b, mask_rgb=cm, mask_flow=d)
            loss = bce_with_logits_loss(out, yy)
            loss.backward()
            opt.step()
            return loss.detach(), out.detach()

        run = GraphedTrainStep(step, opt) if captured else step
        losses = [float(run(rgb, mot, mr, mf, y)[0].clone()) for _ in range(5)]
        traj.append((losses, {k: p.detach().clone() for k, p in m.named_parameters()}, opt.step_count, int(opt.dev_state[0].item())))
    (le, pe, ce, de), (lc, pc, cc, dc) = traj
    print(f"dropout {p_drop}: eager {le}  captured {lc}")
    assert ce == de == cc == dc == 5                          # the capture's warm-up did not count as a step
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(le, lc)), (le, lc)
    assert le[-1] < le[0]
    if p_drop > 0:
        assert len({round(x, 6) for x in lc}) == 5            # masks differ from replay to replay
    for k in pe:
        d = (pe[k] - pc[k]).abs().max().item()
        assert d <= 5e-3 * max(1e-3, pe[k].abs().max().item()), (k, d)
