"""GPU: CLIP ViT encoder (HIP path through the C ABI) vs the CPU oracle and the golden fixtures that
oracle/make_golden.py recorded from transformers.CLIPModel.

Tolerances (BASELINE.json north_star: "CLIP embeddings ... within 1e-3 fp16"):
  f16 compute, fp32 residual stream : |y - y_ref| <= 1e-3 * max(1, max|y_ref|)   (the stated 1e-3)
  bf16 compute (the bench dtype)    : 8x looser (bf16 has 3 fewer mantissa bits than f16): 8e-3
  "stress" fixtures (weight scales x2, i.e. attention logits x4; tiny14) get the bound x stress^2: they exist
  to show the error stays proportional under large-norm activations, not to meet the nominal bound.
Measured on MI355X (gpurun, round 1): f16 1.0e-3..1.5e-3 absolute on |y|max 3.1..3.9 (3e-4..5e-4 relative);
bf16 0.9e-2..1.2e-2 absolute (3e-3 relative).
"""
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from oracle import vit as ovit
from vimo_clip_amd import synth

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def _encoder(c, dtype):
    from vimo_clip_amd.clip_vit import VisionTransformer
    m = VisionTransformer.from_name(c["model"], compute_dtype=dtype).to("cuda")
    m.load_state_dict(synth.vit_state_dict(c["model"], c["seed"], c["stress"]), strict=True)
    return m.eval()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("c", mg.VIT_CASES, ids=lambda c: c["name"])
def test_encoder_vs_golden(golden, c, dtype):
    m = _encoder(c, dtype)
    u8 = mg.vit_pixels(c)
    ref = torch.from_numpy(golden["vit"][f"{c['name']}/emb"])
    y = m.encode_frames_u8(u8.cuda()).cpu()
    y2 = m.encode_pixel_values(ovit.normalize_u8(u8).cuda()).cpu()
    scale = max(1.0, ref.abs().max().item())
    err, err2 = (y - ref).abs().max().item(), (y2 - ref).abs().max().item()
    print(f"{c['name']} {dtype}: max abs err {err:.3e} / {err2:.3e}, scale {scale:.2f}")
    tol = TOL[dtype] * c["stress"] ** 2
    assert err <= tol * scale and err2 <= tol * scale


def test_encoder_vs_oracle_fresh_seed():
    # not a stored fixture: oracle run live on the host CPU (ViT-B/32, 8 frames, stress weights, wrap quirk)
    c = dict(model="ViT-B/32", seed=97, stress=1.5)
    m = _encoder(c, torch.float16)
    u8 = synth.randint_u8(97, "frames", (8, 3, 224, 224))
    sd = synth.vit_state_dict(c["model"], c["seed"], c["stress"])
    ref = ovit.vit_forward(sd, ovit.normalize_u8(ovit.to_pil_wrap_u8(u8)), 12)
    y = m.encode_frames_u8(u8.cuda(), wrap_quirk=True).cpu()
    assert (y - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


def test_encoder_chunking_and_batch_independence():
    c = mg.VIT_CASES[0]
    m = _encoder(c, torch.bfloat16)
    u8 = synth.randint_u8(5, "frames", (7, 3, 64, 64)).cuda()
    full = m.encode_frames_u8(u8)
    m.frame_chunk = 3
    chunked = m.encode_frames_u8(u8)
    assert torch.equal(full, chunked)          # frames are independent: identical bits whatever the batch split


@pytest.mark.parametrize("H,W", [(360, 640), (240, 320), (300, 300), (224, 400), (500, 375), (112, 150)])
@pytest.mark.parametrize("mode", ["torchvision", "hf"])
def test_gpu_resize_center_crop_is_pil_exact(H, W, mode):
    """Resize(224, BICUBIC) + CenterCrop(224) on the GPU is bit-identical to PIL (run live here)."""
    from PIL import Image

    from oracle import pil_resize as opr
    from vimo_clip_amd.preprocess import resize_center_crop_u8
    fr = synth.randint_u8(9, f"fr{H}x{W}", (2, 3, H, W))
    out, pending = resize_center_crop_u8(fr.cuda(), 224, mode, wrap_quirk=True)
    out = out.cpu()
    if pending:      # pure crop (no resampling pass ran): the wrap is left to the normalisation kernel
        assert (H, W) == (224, 400)
        out = ovit.to_pil_wrap_u8(out)
    src = ovit.to_pil_wrap_u8(fr).numpy()
    nh, nw = opr.shortest_edge_size(H, W, 224)
    top, left = opr.center_crop_offsets(nh, nw, 224, mode)
    for f in range(2):
        ref = np.asarray(Image.fromarray(np.transpose(src[f], (1, 2, 0))).resize((nw, nh), Image.BICUBIC))[top:top + 224, left:left + 224]
        assert np.array_equal(np.transpose(out[f].numpy(), (1, 2, 0)), ref)


def test_encoder_on_non_square_frames_vs_oracle():
    # 360x640 frames (the Animal Kingdom flow videos) through the student's preprocessing incl. the wrap quirk
    from oracle import pil_resize as opr
    c = dict(model="ViT-B/32", seed=98, stress=1.0)
    m = _encoder(c, torch.float16)
    u8 = synth.randint_u8(98, "frames", (3, 3, 360, 640))
    sd = synth.vit_state_dict(c["model"], c["seed"], c["stress"])
    pre = torch.from_numpy(opr.clip_resize_crop(ovit.to_pil_wrap_u8(u8).numpy(), 224, "torchvision").copy())
    ref = ovit.vit_forward(sd, ovit.normalize_u8(pre), 12)
    y = m.encode_frames_u8(u8.cuda(), wrap_quirk=True).cpu()
    assert (y - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


def test_full_size_workload_properties_vit_l14():
    """BASELINE.json configs[1] at full size (ViT-L/14, B*T = 256 frames = 65 792 token rows: the bench workload), checked
    through size-independent properties since the CPU oracle needs minutes for it: (a) frames are independent -- one pass
    over 256 frames, four passes over 64 and a shuffled pass give the same bits per frame although the GEMMs then take
    different kernels (8-phase tiles + 256-row tail vs other tile counts); (b) a 3-frame sample of the same batch agrees
    with the CPU oracle within the bf16 tolerance."""
    name, seed = "ViT-L/14", 2
    from vimo_clip_amd.clip_vit import VisionTransformer
    m = VisionTransformer.from_name(name, compute_dtype=torch.bfloat16).to("cuda").eval()
    sd = synth.vit_state_dict(name, seed)
    m.load_state_dict(sd, strict=True)
    u8 = synth.randint_u8(1, "frames", (256, 3, 224, 224)).cuda()
    m.frame_chunk = 256
    full = m.encode_frames_u8(u8)
    assert full.shape == (256, 768) and torch.isfinite(full).all()
    m.frame_chunk = 64
    assert torch.equal(m.encode_frames_u8(u8), full)
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).cuda()
    m.frame_chunk = 256
    assert torch.equal(m.encode_frames_u8(u8[perm]), full[perm])
    pick = [0, 131, 255]
    ref = ovit.vit_forward(sd, ovit.normalize_u8(u8[pick].cpu()), 16)
    err = (full[pick].cpu() - ref).abs().max().item()
    assert err <= TOL[torch.bfloat16] * max(1.0, ref.abs().max().item()), err
