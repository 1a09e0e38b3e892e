"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container (test infrastructure).

    python -m oracle.make_golden            # needs /root/reference (not present on the GPU box)

What is executed from the reference / third-party code it calls, and what each fixture pins:
  * losses.py (imported as a module)                     -> golden/losses.npz   (values + autograd grads)
  * TFAM/models/AMO_CLIP.py (imported as a module)       -> golden/tfam.npz     (eval logits, 4 fusion modes,
                                                            masks, PE; train-mode grads with dropout 0)
  * TFAM/data/dataset.py: sparse_sampling, collate_fn_pad — the module's top-level ``import h5py`` is an
    ordinary ModuleNotFoundError offline, so the two pure-torch functions are compiled from the file's
    AST without executing the rest                         -> golden/indexing.npz
  * transformers.CLIPModel built from a CLIPConfig (no download) with our seeded weights mapped in
    (the ViT arithmetic lives in third-party code: OpenAI clip @dcba3cb / transformers 4.53.2; the
    container has transformers 5.15.0)                     -> golden/vit.npz
  * sklearn.metrics.average_precision_score (same estimator as torchmetrics micro AP, which is not
    installed)                                             -> golden/metrics.npz
Only inputs' seeds and the reference OUTPUTS are stored (weights/inputs are regenerated from
vimo_clip_amd.synth).  While generating, every oracle restatement is asserted against the reference
output, so a committed fixture implies oracle == reference on that case at generation time.
"""
from __future__ import annotations

import ast
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("VIMOCLIP_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from vimo_clip_amd import synth  # noqa: E402
from oracle import indexing, metrics, student, tfam, vit  # noqa: E402

# ---- shared case tables (tests import these so the seeds/configs live in one place) -----------------
LOSS_CASES = [
    dict(name="cos_b4", B=4, T=16, E=512, seed=11),
    dict(name="cos_b2_e768", B=2, T=15, E=768, seed=12),
    dict(name="cos_degenerate", B=2, T=4, E=64, seed=13),   # zero rows + identical rows: clamps active
]
BCE_CASES = [dict(name="bce_pw9", B=8, C=140, pw=9, seed=21), dict(name="bce_none", B=5, C=140, pw=None, seed=22)]

TFAM_CASES = [
    dict(name="cross_d512", D=512, H=8, L=4, ff=2048, C=140, B=4, Tr=16, Tf=15, mode="cross", pe=False, ragged=True, seed=31),
    dict(name="cross_d768", D=768, H=8, L=4, ff=2048, C=140, B=3, Tr=16, Tf=16, mode="cross", pe=False, ragged=False, seed=32),
    dict(name="cross_d768_ragged_pe", D=768, H=8, L=2, ff=2048, C=140, B=5, Tr=16, Tf=16, mode="cross", pe=True, ragged=True, seed=33),
    dict(name="rgb_only", D=512, H=8, L=2, ff=2048, C=140, B=4, Tr=16, Tf=15, mode="rgb", pe=False, ragged=True, seed=34),
    dict(name="flow_only", D=512, H=8, L=2, ff=2048, C=140, B=4, Tr=16, Tf=15, mode="flow", pe=True, ragged=True, seed=35),
    dict(name="concat_time", D=512, H=8, L=2, ff=2048, C=140, B=4, Tr=16, Tf=15, mode="concat1", pe=False, ragged=True, seed=36),
    dict(name="concat_embed", D=512, H=8, L=2, ff=2048, C=140, B=4, Tr=16, Tf=15, mode="concat-1", pe=False, ragged=True, seed=37),
    dict(name="cross_long", D=512, H=8, L=1, ff=2048, C=140, B=2, Tr=40, Tf=39, mode="cross", pe=False, ragged=True, seed=38),
]

VIT_CASES = [
    dict(name="tiny32", model="ViT-tiny/32", F=3, seed=41, stress=1.0),
    dict(name="tiny16", model="ViT-tiny/16", F=2, seed=42, stress=1.0),
    dict(name="tiny14", model="ViT-tiny/14", F=2, seed=43, stress=2.0),
    dict(name="b32", model="ViT-B/32", F=4, seed=44, stress=1.0),
    dict(name="b16", model="ViT-B/16", F=2, seed=45, stress=1.0),
    dict(name="l14", model="ViT-L/14", F=4, seed=46, stress=1.0),          # BASELINE config 1 shape
]


def tfam_inputs(c):
    rgb = synth.normal(c["seed"], "rgb", (c["B"], c["Tr"], c["D"]))
    mot = synth.normal(c["seed"], "motion", (c["B"], c["Tf"], c["D"]))
    if c["ragged"]:
        lens = synth.randint(c["seed"], "lens", (c["B"],), 5, c["Tr"] + 1)
        lens[0] = c["Tr"]
    else:
        lens = torch.full((c["B"],), c["Tr"], dtype=torch.int64)
    # real data has Tf = Tr - 1 per video; keep >= 1 valid key
    lens_f = torch.clamp(lens - (c["Tr"] - c["Tf"]), min=1)
    mask_rgb = torch.arange(c["Tr"]).unsqueeze(0) < lens.unsqueeze(1)
    mask_flow = torch.arange(c["Tf"]).unsqueeze(0) < lens_f.unsqueeze(1)
    rgb = rgb * mask_rgb.unsqueeze(-1)     # collate_fn_pad zero-pads (TFAM/data/dataset.py:86-87)
    mot = mot * mask_flow.unsqueeze(-1)
    return rgb, mot, mask_rgb, mask_flow


def tfam_mode_kwargs(mode):
    return dict(
        use_cross_attention=(mode == "cross"), use_only_rgb=(mode == "rgb"), use_only_flow=(mode == "flow"),
        concat_dim=(-1 if mode == "concat-1" else 1))


def loss_inputs(c):
    s = synth.normal(c["seed"], "student", (c["B"], c["T"], c["E"]))
    t = synth.normal(c["seed"], "teacher", (c["B"], c["T"], c["E"]))
    if c["name"] == "cos_degenerate":
        s[0, 0] = 0.0                      # zero-norm student row -> norm clamp
        t[0, 1] = 0.0
        s[1, 2] = t[1, 2]                  # identical rows -> cos clamp at 1-eps (zero gradient)
        s[1, 3] = -t[1, 3]
    return s, t


def bce_inputs(c):
    x = synth.normal(c["seed"], "logits", (c["B"], c["C"]), std=3.0)
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"])
    return x, y


def vit_pixels(c):
    R = synth.VIT_GEOMETRY[c["model"]][0]
    u8 = synth.randint_u8(c["seed"], "frames", (c["F"], 3, R, R))
    return u8


def _load_ref_functions(path, names):
    """Compile selected top-level pure functions of a reference file from its AST (the module itself
    cannot be imported: ModuleNotFoundError on h5py).  Nothing else in the file is executed."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__, "reference": REF}

    # ---------------- losses.py ----------------
    sys.path.insert(0, REF)
    import losses as ref_losses
    out = {}
    for c in LOSS_CASES:
        s, t = loss_inputs(c)
        for mode in ("cosine", "mse"):
            s1 = s.clone().requires_grad_(True)
            l = ref_losses.distillation_loss(s1, t, mode=mode)
            l.backward()
            out[f"{c['name']}/{mode}/loss"] = l.detach().numpy()
            out[f"{c['name']}/{mode}/grad"] = s1.grad.numpy()
            lo = student.distillation_loss(s, t, mode)
            assert abs(float(lo) - float(l)) <= 1e-6 * max(1.0, abs(float(l))), (c, mode, float(lo), float(l))
    for c in BCE_CASES:
        x, y = bce_inputs(c)
        x1 = x.clone().requires_grad_(True)
        l = ref_losses.classification_loss(x1, y, positive_weight=c["pw"])
        l.backward()
        out[f"{c['name']}/loss"] = l.detach().numpy()
        out[f"{c['name']}/grad"] = x1.grad.numpy()
        lo = student.classification_loss(x, y, c["pw"])
        assert abs(float(lo) - float(l)) <= 2e-6 * max(1.0, abs(float(l))), (c, float(lo), float(l))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("losses.npz", len(out))

    # ---------------- TFAM/models/AMO_CLIP.py ----------------
    sys.path.insert(0, os.path.join(REF, "TFAM"))
    from models.AMO_CLIP import AMO_CLIP as RefAMO
    out = {}
    for c in TFAM_CASES:
        sd = synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"])
        kw = tfam_mode_kwargs(c["mode"])
        m = RefAMO(d_model=c["D"], nhead=c["H"], num_layers=c["L"], dim_feedforward=c["ff"], num_classes=c["C"],
                   use_pe=c["pe"], dropout=0.0, mlp_dropout=0.0, device="cpu", **kw)
        m.load_state_dict(sd, strict=True)
        m.eval()
        rgb, mot, mr, mf = tfam_inputs(c)
        with torch.no_grad():
            logits = m(rgb.clone(), mot.clone(), mask_rgb=mr, mask_flow=mf)
        out[f"{c['name']}/logits"] = logits.numpy()
        lo = tfam.amo_clip_forward(sd, rgb, mot, mr, mf, nhead=c["H"], use_pe=c["pe"], **kw)
        err = (lo - logits).abs().max().item()
        assert err < 2e-5, (c["name"], err)
        # train-mode (dropout 0) BCE loss + a few gradients, for the backward parity tests
        m.train()
        y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"])
        lt = m(rgb.clone(), mot.clone(), mask_rgb=mr, mask_flow=mf)
        loss = torch.nn.BCEWithLogitsLoss()(lt, y)
        loss.backward()
        out[f"{c['name']}/train_loss"] = loss.detach().numpy()
        for k in ("classifier.4.weight", "classifier.1.bias", "layers.0.ffn.0.bias", "layers.0.self_attn.in_proj_bias",
                  "layers.0.norm_self.weight"):
            gk = dict(m.named_parameters())[k].grad
            out[f"{c['name']}/grad/{k}"] = gk.numpy()
        print("  tfam", c["name"], "oracle-vs-ref max abs", err)
    np.savez_compressed(os.path.join(OUT, "tfam.npz"), **out)
    print("tfam.npz", len(out))

    # ---------------- TFAM/data/dataset.py: sparse_sampling, collate_fn_pad ----------------
    ref_sparse, ref_collate = _load_ref_functions(os.path.join(REF, "TFAM", "data", "dataset.py"),
                                                  ["sparse_sampling", "collate_fn_pad"])
    out = {}
    pairs = [(T, n) for T in (1, 2, 5, 16, 17, 31, 64, 100, 257, 450, 999, 1800) for n in (1, 2, 8, 16, 30, 64)]
    for T, n in pairs:
        emb = torch.arange(T, dtype=torch.float32).unsqueeze(1)
        got = ref_sparse(emb, n)[:, 0].long()
        out[f"sparse/{T}/{n}"] = got.numpy()
        assert torch.equal(got, indexing.sparse_sampling_indices(T, n)), (T, n)
    lens_r = [16, 5, 9, 12]
    lens_f = [15, 4, 8, 11]
    batch = [dict(video_id=f"v{i}", embeddings=torch.ones(a, 4) * (i + 1), flow_embeddings=torch.ones(b, 4) * (i + 1),
                  labels=torch.zeros(3)) for i, (a, b) in enumerate(zip(lens_r, lens_f))]
    col = ref_collate(batch)
    out["collate/lens_rgb"] = np.array(lens_r)
    out["collate/lens_flow"] = np.array(lens_f)
    out["collate/mask_rgb"] = col["mask_rgb"].numpy()
    out["collate/mask_flow"] = col["mask_flow"].numpy()
    out["collate/embeddings"] = col["embeddings"].numpy()
    assert np.array_equal(indexing.pad_and_mask(lens_r), col["mask_rgb"].numpy())
    assert np.array_equal(indexing.pad_and_mask(lens_f), col["mask_flow"].numpy())
    # extract_embeddings.py:77-81 — the expression is numpy only; evaluate the reference expression directly
    for total in (1, 4, 16, 17, 99, 100, 450, 1801):
        for mx in (None, 1, 4, 16, 64, 100):
            if (mx is None) or (total <= mx):
                ref_idx = np.arange(total)
            else:
                ref_idx = np.arange(0, total, total // mx)[:mx]
            out[f"frames/{total}/{mx}"] = ref_idx
            assert np.array_equal(ref_idx, indexing.sample_frame_indices(total, mx)), (total, mx)
    np.savez_compressed(os.path.join(OUT, "indexing.npz"), **out)
    print("indexing.npz", len(out))

    # ---------------- ViT arithmetic vs transformers.CLIPModel from config ----------------
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig
    out = {}
    for c in VIT_CASES:
        R, p, D, L, H, E = synth.VIT_GEOMETRY[c["model"]]
        sd = synth.vit_state_dict(c["model"], c["seed"], c["stress"])
        vcfg = CLIPVisionConfig(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=L, num_attention_heads=H,
                                image_size=R, patch_size=p, projection_dim=E, hidden_act="quick_gelu",
                                layer_norm_eps=1e-5, attention_dropout=0.0)
        tcfg = CLIPTextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                              projection_dim=E, vocab_size=100, max_position_embeddings=8)
        cfg = CLIPConfig(text_config=tcfg.to_dict(), vision_config=vcfg.to_dict(), projection_dim=E)
        hf = CLIPModel(cfg).eval()
        missing, unexpected = hf.load_state_dict(vit.openai_to_hf_vision(sd, H), strict=False)
        assert not unexpected and all(not k.startswith(("vision_model", "visual_projection")) for k in missing), \
            (missing[:5], unexpected[:5])
        pix = vit.normalize_u8(vit_pixels(c))
        with torch.no_grad():
            ref = hf.get_image_features(pixel_values=pix)
            if not isinstance(ref, torch.Tensor):      # transformers >= 5
                ref = ref.pooler_output
            mine = vit.vit_forward(sd, pix, H)
        err = (mine - ref).abs().max().item()
        scale = ref.abs().max().item()
        print(f"  vit {c['name']}: oracle-vs-HF max abs {err:.3e} (|y|max {scale:.3f})")
        assert err <= 2e-5 * max(1.0, scale), (c["name"], err, scale)
        out[f"{c['name']}/emb"] = ref.numpy()
    np.savez_compressed(os.path.join(OUT, "vit.npz"), **out)
    print("vit.npz", len(out))

    # ---------------- micro AP vs sklearn ----------------
    from sklearn.metrics import average_precision_score
    out = {}
    for i, (N, C, quant) in enumerate([(64, 140, False), (200, 140, True), (33, 7, False)]):
        logits = synth.normal(50 + i, "ap_logits", (N, C), std=2.0)
        if quant:
            logits = torch.round(logits * 2) / 2          # many ties
        y = synth.multi_hot_labels(50 + i, "ap_labels", N, C).numpy().astype(np.int64)
        s = metrics.maybe_sigmoid(logits.numpy())
        ref = average_precision_score(y.ravel(), s.ravel())
        mine = metrics.micro_average_precision(s, y)
        assert abs(ref - mine) < 1e-9, (ref, mine)
        out[f"ap{i}/N"], out[f"ap{i}/C"], out[f"ap{i}/quant"], out[f"ap{i}/value"] = N, C, quant, ref
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics.npz", len(out))

    # ---------------- Animal Kingdom annotations (data shipped with the reference): label subset ----------------
    # BASELINE.json configs[4] trains on the AK annotations with synthetic embeddings; the first 768 train and 256 val
    # lines (video id + class ids, parsed as extract_embeddings.py:46-47,97-103 does) are kept as a data fixture.
    out = {}
    for split, n in (("train", 768), ("val", 256)):
        with open(os.path.join(REF, "dataset", "annotations", f"{split}_multi.txt"), "r", encoding="utf-8") as f:
            ann = [line.strip().split() for line in f if line.strip()][:n]
        lab = np.zeros((len(ann), 140), dtype=np.uint8)
        for i, a in enumerate(ann):
            for c in a[1:]:
                lab[i, int(c)] = 1
        out[f"{split}/labels"] = np.packbits(lab, axis=1)
        out[f"{split}/ids"] = np.array([a[0] for a in ann])
    np.savez_compressed(os.path.join(OUT, "ak_labels.npz"), **out)
    print("ak_labels.npz", {k: v.shape for k, v in out.items()})

    with open(os.path.join(OUT, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
