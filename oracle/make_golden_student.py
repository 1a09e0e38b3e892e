"""Golden fixtures for the student side, produced by running the REFERENCE's own classes in the build container
(test infrastructure; needs /root/reference, which does not travel -> the .npz it writes is committed).

The modules themselves cannot be imported here (ordinary ModuleNotFoundError: ``clip``, ``torchvision`` for
models/student_model.py; ``h5py``, ``torchvision.io`` for dataset.py), so, as make_golden.py already does for
TFAM/data/dataset.py, the classes are compiled from the file's AST -- nothing else in the file runs:

* ``ResidualMLP`` (models/student_model.py:8-35): pure torch.nn, runs as is.  Pins the zero-initialised fc2, the exact-GELU
  and ``x + alpha * mlp(x)``.
* ``HDF5VideoDataset`` + ``collate_fn`` (dataset.py:8-148): run with DATA-SOURCE stand-ins only -- ``h5py`` is this repo's
  own HDF5 implementation (vimo_clip_amd/h5lite.py, itself pinned against libhdf5 in tests/test_h5lite.py) and
  ``io.read_video`` returns stored frame arrays.  Every index / pad / clamp decision is the reference's code.  Embedding rows
  and flow frames carry their own index as value, so the outputs ARE the gather indices (-1 = all-zero padding frame).

* ``FlowStudentModel`` (models/student_model.py:38-98), round 3: ``__init__`` and ``forward`` run as written, with stand-ins for the
  three things that do not exist offline: ``clip.load(name, device)`` returns (a model whose ``.visual`` is the oracle ViT as an
  ``nn.Module`` with seeded weights and ``.output_dim``, a preprocess object whose ``.transforms`` is [u8 frame -> normalised
  tensor]); ``transforms.Compose`` chains callables; ``to_pil_image`` is bound to the float -> u8 wrap the oracle ascribes to it
  (``vit.to_pil_wrap_u8``).  That pins the COMPOSITION -- view(B*T) -> per-frame preprocess -> stack -> visual encoder ->
  view(B, T, -1) -> ResidualMLP -> mean(dim=1) -> classification_head(.float()) and the three returned tensors -- against
  oracle/student.py:student_forward.  What remains unpinnable: torchvision's ``to_pil_image`` itself (its float handling is
  restated, not run) and the weights ``clip.load`` would fetch by name.

    python -m oracle.make_golden_student        # writes tests/golden/student.npz
"""
from __future__ import annotations

import ast
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import indexing, student  # noqa: E402
from vimo_clip_amd import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

MLP_CASES = [dict(name="e32", E=32, B=3, T=5, alpha=0.1, seed=71), dict(name="e512", E=512, B=2, T=16, alpha=0.1, seed=72),
             dict(name="e768_a05", E=768, B=1, T=4, alpha=0.5, seed=73)]
# FlowStudentModel.forward composition (models/student_model.py:61-98) on tiny geometries; alpha != default in one case
FWD_CASES = [dict(name="tiny32", model="ViT-tiny/32", B=3, T=5, C=140, alpha=0.1, seed=81),
             dict(name="tiny14_a03", model="ViT-tiny/14", B=2, T=4, C=20, alpha=0.3, seed=82)]
# video lengths (embedding rows) and flow lengths; real data has T_flow = T - 1, the others exercise the clamps (:108-113)
VIDEOS = [("a.mp4", 12, 11), ("b.mp4", 5, 4), ("c.mp4", 1, 0), ("d.mp4", 0, 0), ("e.mp4", 7, 3), ("f.mp4", 30, 29), ("g.mp4", 9, 12)]
SEQ_LENS = [2, 4, 5, 17, 30]


def _ast_classes(path, names, ns):
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def mlp_inputs(c):
    x = synth.normal(c["seed"], "x", (c["B"], c["T"], c["E"]))
    w1 = synth.normal(c["seed"], "fc1.w", (c["E"], c["E"]), std=c["E"] ** -0.5)
    b1 = synth.normal(c["seed"], "fc1.b", (c["E"],), std=0.1)
    w2 = synth.normal(c["seed"], "fc2.w", (c["E"], c["E"]), std=c["E"] ** -0.5)
    b2 = synth.normal(c["seed"], "fc2.b", (c["E"],), std=0.1)
    return x, w1, b1, w2, b2


def main():
    out = {}
    # ---------------- ResidualMLP ----------------
    (RefMLP,) = _ast_classes(os.path.join(REF, "models", "student_model.py"), ["ResidualMLP"], {"torch": torch, "nn": torch.nn})
    for c in MLP_CASES:
        x, w1, b1, w2, b2 = mlp_inputs(c)
        m = RefMLP(c["E"], alpha=c["alpha"])
        assert float(m.fc2.weight.detach().abs().max()) == 0.0 and float(m.fc2.bias.detach().abs().max()) == 0.0      # :25-26
        with torch.no_grad():
            assert torch.equal(m(x), x)                                                              # fresh module = identity
            m.fc1.weight.copy_(w1); m.fc1.bias.copy_(b1); m.fc2.weight.copy_(w2); m.fc2.bias.copy_(b2)
            y = m(x)
        out[f"mlp/{c['name']}/y"] = y.numpy()
        sd = {"residual_mlp.fc1.weight": w1, "residual_mlp.fc1.bias": b1, "residual_mlp.fc2.weight": w2, "residual_mlp.fc2.bias": b2}
        err = (student.residual_mlp(sd, x, c["alpha"]) - y).abs().max().item()
        assert err < 1e-6, (c["name"], err)
        # gradient of sum(y * g) w.r.t. x and fc1.weight, for the backward parity test
        g = synth.normal(c["seed"], "g", tuple(y.shape))
        xr = x.clone().requires_grad_(True)
        (m(xr) * g).sum().backward()
        out[f"mlp/{c['name']}/dx"] = xr.grad.numpy()
        out[f"mlp/{c['name']}/dfc1w"] = m.fc1.weight.grad[:8].numpy()          # first 8 rows: keeps the fixture small

    # ---------------- FlowStudentModel: __init__ + forward as written, stand-ins for clip.load / Compose / to_pil_image -------------
    from oracle import vit as ovit
    for c in FWD_CASES:
        sd = synth.student_state_dict(c["model"], c["seed"], num_classes=c["C"])
        R, heads, E = synth.VIT_GEOMETRY[c["model"]][0], synth.VIT_GEOMETRY[c["model"]][4], synth.VIT_GEOMETRY[c["model"]][5]

        class _Visual(torch.nn.Module):
            output_dim = E

            def forward(self, pix):
                return ovit.vit_forward(sd, pix, heads, prefix="visual_encoder.")

        class _ClipModel:
            visual = _Visual()

            def float(self):
                return self

        class _Preprocess:
            transforms = [lambda fr_u8: ovit.normalize_u8(fr_u8.unsqueeze(0))[0]]

        class _Clip:
            @staticmethod
            def load(name, device=None):
                assert name == c["model"]
                return _ClipModel(), _Preprocess()

        class _Transforms:
            @staticmethod
            def Compose(fns):
                def run(x):
                    for f in fns:
                        x = f(x)
                    return x
                return run

        ns = {"torch": torch, "nn": torch.nn, "clip": _Clip, "transforms": _Transforms, "to_pil_image": ovit.to_pil_wrap_u8}
        RefMLP2, RefStudent = _ast_classes(os.path.join(REF, "models", "student_model.py"), ["ResidualMLP", "FlowStudentModel"], ns)
        m = RefStudent(clip_model_name=c["model"], device="cpu", num_classes=c["C"], alpha=c["alpha"])
        assert isinstance(m.residual_mlp, RefMLP2) and m.classification_head[0].weight.shape == (E // 2, E)
        own = {k: v for k, v in sd.items() if not k.startswith("visual_encoder.")}
        missing = m.load_state_dict(own, strict=False)            # the stand-in encoder holds its weights outside the module tree
        assert not missing.unexpected_keys and not missing.missing_keys, missing
        vids = synth.randint_u8(c["seed"], "vids", (c["B"], c["T"], 3, R, R))
        with torch.no_grad():
            emb, emb_d, logits = m(vids)
        o_emb, o_emb_d, o_logits = student.student_forward(sd, vids, heads, alpha=c["alpha"], wrap_quirk=True)
        for a, b, nm in ((emb, o_emb, "emb"), (emb_d, o_emb_d, "emb_distill"), (logits, o_logits, "logits")):
            err = (a - b).abs().max().item()
            assert a.shape == b.shape and err < 1e-5, (c["name"], nm, err)
        out[f"fwd/{c['name']}/emb"], out[f"fwd/{c['name']}/emb_distill"], out[f"fwd/{c['name']}/logits"] = emb.numpy(), emb_d.numpy(), logits.numpy()
        print("  student fwd", c["name"], "oracle-vs-reference-class max abs", max((a - b).abs().max().item() for a, b in
                                                                                 ((emb, o_emb), (emb_d, o_emb_d), (logits, o_logits))))

    # ---------------- dataset.py ----------------
    from vimo_clip_amd import h5lite

    class _IO:      # data-source stand-in for torchvision.io: frames of video v are [T_flow, 2, 2, 3] u8 filled with index + 1
        @staticmethod
        def read_video(path, pts_unit="sec"):
            tf = dict((v, t) for v, _, t in VIDEOS)[os.path.basename(path)]
            fr = torch.zeros((tf, 2, 2, 3), dtype=torch.uint8)
            for t in range(tf):
                fr[t] = t + 1
            return fr, None, None

    ns = {"os": os, "h5py": h5lite, "torch": torch, "io": _IO, "Dataset": torch.utils.data.Dataset}
    RefDS, ref_collate = _ast_classes(os.path.join(REF, "dataset.py"), ["HDF5VideoDataset", "collate_fn"], ns)
    with tempfile.TemporaryDirectory() as td:
        h5p = os.path.join(td, "emb.h5")
        with h5lite.File(h5p, "w") as f:
            for vi, (v, T, _) in enumerate(VIDEOS):
                g = f.create_group(v)
                emb = np.zeros((T, 4), dtype=np.float32)
                emb[:, 0] = np.arange(T)            # column 0 = row index, column 1 = video index
                emb[:, 1] = vi
                g.create_dataset("embeddings", data=emb)
                lab = np.zeros(6, dtype=np.float32)
                lab[vi % 6] = 1.0
                g.create_dataset("labels", data=lab)
        vidx = {v: i for i, (v, _, _) in enumerate(VIDEOS)}
        for sl in SEQ_LENS:
            ds = RefDS(h5p, td, sequence_length=sl)
            segs = np.array([(vidx[k], s, n) for k, s, n in ds.segments], dtype=np.int64).reshape(-1, 3)
            rgb_idx = np.zeros((len(ds), sl), dtype=np.int64)
            flow_idx = np.zeros((len(ds), max(sl - 1, 0)), dtype=np.int64)
            for i in range(len(ds)):
                it = ds[i]
                assert it["video_id"] == ds.segments[i][0]
                assert it["rgb_emb"].shape == (sl, 4) and it["flow_frames"].shape[0] == sl - 1
                assert float(it["rgb_emb"][:, 1].min()) == float(it["rgb_emb"][:, 1].max()) == vidx[it["video_id"]]
                rgb_idx[i] = it["rgb_emb"][:, 0].long().numpy()
                ff = it["flow_frames"].reshape(sl - 1, -1)
                assert bool((ff == ff[:, :1]).all())
                flow_idx[i] = ff[:, 0].long().numpy() - 1                 # 0 (zero frame) -> -1
            out[f"ds/L{sl}/segments"], out[f"ds/L{sl}/rgb_idx"], out[f"ds/L{sl}/flow_idx"] = segs, rgb_idx, flow_idx
            # the oracle restatement against the reference's own output
            lengths = {v: T for v, T, _ in VIDEOS}
            osegs = indexing.build_segments(lengths, sl)
            assert [(vidx[k], s, n) for k, s, n in osegs] == [tuple(r) for r in segs.tolist()], sl
            tfl = {v: t for v, _, t in VIDEOS}
            for i, (k, s, n) in enumerate(osegs):
                assert indexing.rgb_segment_indices(s, n, sl) == rgb_idx[i].tolist(), (sl, i)
                assert indexing.flow_segment_indices(s, n, sl, tfl[k]) == flow_idx[i].tolist(), (sl, i, k)
            if sl == 4:     # collate_fn (:137-148) on the first three items
                b = ref_collate([ds[i] for i in range(3)])
                out["ds/collate/rgb"], out["ds/collate/flow"] = b["rgb_emb"].numpy(), b["flow_frames"].numpy()
                out["ds/collate/labels"] = b["labels"].numpy()
                assert b["video_id"] == [ds.segments[i][0] for i in range(3)]
    out["ds/videos"] = np.array([(T, tf) for _, T, tf in VIDEOS], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "student.npz"), **out)
    print("student.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
