"""GPU: the fused TFAM launch chain (vmc_tfam_kv_fwd / vmc_tfam_layer_fwd / vmc_tfam_head_fwd / vmc_tfam_forward) through the C ABI.

Pins: the golden logits recorded from the imported reference module (tests/golden/tfam.npz, TFAM/models/AMO_CLIP.py) for
every fusion mode the chain covers; bit-exact integer GEMM through the hoisted K|V entry point; the per-op path (the same
model with ``fused_inference = False``) on batch sizes and ragged masks the fixtures do not hold; hipGraph replay after an
optimiser step (the packs are rewritten in place, so a captured forward must see the new weights).

Tolerance: north_star's "TFAM logits within 1e-3 fp16": f16 |d| <= 1e-3 * max(1, max|ref|); bf16 8e-3 (3 fewer mantissa bits).
"""
import pytest
import torch

from oracle import make_golden as mg
from oracle import tfam as otfam
from vimo_clip_amd import synth

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
FUSED_CASES = list(mg.TFAM_CASES)      # incl. cross_long (T = 40, Tk = 39): clips of up to 64 tokens run through the chain (round 3)


def _tfam(c, dtype):
    from vimo_clip_amd.TFAM.models import AMO_CLIP
    kw = mg.tfam_mode_kwargs(c["mode"])
    m = AMO_CLIP(d_model=c["D"], nhead=c["H"], num_layers=c["L"], dim_feedforward=c["ff"], num_classes=c["C"], use_pe=c["pe"],
                 dropout=0.0, mlp_dropout=0.0, device="cuda", compute_dtype=dtype, **kw).cuda()
    m.load_state_dict(synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"]), strict=True)
    return m.eval()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("c", FUSED_CASES, ids=lambda c: c["name"])
def test_fused_chain_is_taken_and_matches_reference(golden, c, dtype, monkeypatch):
    from vimo_clip_amd import tfam_fused as tf
    m = _tfam(c, dtype)
    calls = []
    orig = tf.TfamPack.forward
    monkeypatch.setattr(tf.TfamPack, "forward", lambda self, *a, **k: calls.append(1) or orig(self, *a, **k))
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    with torch.no_grad():
        y = m(rgb.cuda(), mot.cuda(), mask_rgb=mr.cuda(), mask_flow=mf.cuda()).cpu()
    assert calls, "the fused chain was not used for a shape it supports"
    ref = torch.from_numpy(golden["tfam"][f"{c['name']}/logits"])
    err = (y - ref).abs().max().item()
    print(f"fused tfam {c['name']} {dtype}: max abs err {err:.3e} (|ref|max {ref.abs().max():.2f})")
    assert err <= TOL[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("c", [mg.TFAM_CASES[0], mg.TFAM_CASES[3], mg.TFAM_CASES[7]], ids=lambda c: c["name"])
def test_per_op_eval_path_still_matches_reference(golden, c):
    m = _tfam(c, torch.float16)
    m.fused_inference = False
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    with torch.no_grad():
        y = m(rgb.cuda(), mot.cuda(), mask_rgb=mr.cuda(), mask_flow=mf.cuda()).cpu()
    ref = torch.from_numpy(golden["tfam"][f"{c['name']}/logits"])
    assert (y - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


def _lib():
    from vimo_clip_amd import _lib
    return _lib


@pytest.mark.parametrize("B,Tk", [(1, 16), (8, 16), (5, 15), (33, 9)])
def test_hoisted_kv_gemm_bit_exact_on_integers(B, Tk):
    """ws.kv = motion x kv_all^T + bias with operands in {-2..2}: every partial sum is an integer < 2048, exact in f16, so the
    LDS-DMA image, the XOR swizzle, the MFMA operand maps, the block -> tile map and the epilogue must agree bit for bit
    with an integer matmul."""
    L_ = _lib()
    lib, ptr, stream, check = L_.lib, L_.ptr, L_.stream, L_.check
    D, H, ff, L, C, T = 768, 8, 2048, 4, 140, 16
    g = torch.Generator().manual_seed(B * 100 + Tk)
    motion = torch.randint(-2, 3, (B * Tk, D), generator=g).float()
    wkv = torch.randint(-2, 3, (L * 2 * D, D), generator=g).float()
    bias = torch.randint(-3, 4, (L * 2 * D,), generator=g).float()
    off = lambda slot, layer=0: int(lib.vmc_tfam_pack_offset(slot, layer, D, ff, L, C))
    wpack = torch.zeros(off(9), dtype=torch.float16, device="cuda")
    ppack = torch.zeros(off(29), dtype=torch.float32, device="cuda")
    wpack[off(6):off(6) + wkv.numel()] = wkv.reshape(-1).half().cuda()
    ppack[off(25):off(25) + bias.numel()] = bias.cuda()
    n = lib.vmc_tfam_workspace_bytes(B, T, Tk, D, ff, L, C, 1)
    ws = torch.zeros(n, dtype=torch.uint8, device="cuda")
    md = motion.cuda()
    check(lib.vmc_tfam_kv_fwd(ptr(md), ptr(wpack), ptr(ppack), ptr(ws), n, B, T, Tk, D, H, ff, L, C, 2, stream()), "tfam_kv_fwd")
    torch.cuda.synchronize()
    # workspace layout (tfam_fused.hip tf_ws): y, xa, xb f32 [M,D]; qkv [M,3D]; q [M,D]; h [M,ff]; kv [Mk, L*2D] (V columns);
    # pool [B,D]; g [B,D/2]; then the fragment-major buffers q, self k, cross k x L (tf_frag_off)
    al = lambda x: (x + 255) // 256 * 256
    M, dh, KK = B * T, D // H, D // H // 32
    o = 3 * al(M * D * 4) + al(M * 3 * D * 2) + al(M * D * 2) + al(M * ff * 2)
    kv = ws[o:o + B * Tk * L * 2 * D * 2].view(torch.float16).view(B * Tk, L * 2 * D).float().cpu()
    fe = B * D * 32
    o += al(B * Tk * L * 2 * D * 2) + al(B * D * 2) + al(B * (D // 2) * 2) + 2 * al(fe * 2)
    ref = motion.double() @ wkv.double().t() + bias.double()
    assert ref.abs().max() < 2048
    # index map of the fragment-major layout: element (clip, head, t, d)
    clip, head, t, d = torch.meshgrid(torch.arange(B), torch.arange(H), torch.arange(Tk), torch.arange(dh), indexing="ij")
    off = ((((clip * H + head) * 2 + (t // 16)) * KK + (d // 32)) * 64 + ((d // 8) % 4) * 16 + (t % 16)) * 8 + (d % 8)
    for l in range(L):
        kx = ws[o + l * al(fe * 2):o + l * al(fe * 2) + fe * 2].view(torch.float16).float().cpu()
        k_got = kx[off.reshape(-1)].view(B, H, Tk, dh).permute(0, 2, 1, 3).reshape(B * Tk, D)
        assert torch.equal(k_got.double(), ref[:, l * 2 * D:l * 2 * D + D]), f"K of layer {l} (fragment-major)"
        assert torch.equal(kv[:, l * 2 * D + D:(l + 1) * 2 * D].double(), ref[:, l * 2 * D + D:(l + 1) * 2 * D]), f"V of layer {l}"


@pytest.mark.parametrize("B", [1, 2, 8, 9, 16, 40])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_fused_chain_vs_oracle_over_batch_sizes(B, dtype):
    """Cross mode at the BASELINE geometry (D 768, H 8, ff 2048, L 4, T 16 / Tk 15) with ragged masks, batches that do and
    do not fill the 2-clip row blocks, against the CPU oracle (pinned to the reference by tests/golden/tfam.npz)."""
    c = dict(name=f"b{B}", D=768, H=8, L=4, ff=2048, C=140, B=B, Tr=16, Tf=15, mode="cross", pe=False, ragged=True, seed=900 + B)
    from vimo_clip_amd import tfam_fused as tf
    m = _tfam(c, dtype)
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    # straight through the C ABI (AMO_CLIP.forward itself hands batches above tfam_fused.MAX_ROWS rows to the per-op path)
    pack = tf.get_pack(m, dtype).refresh()
    y = pack.forward(rgb.cuda().contiguous(), mot.cuda().contiguous(), mr.cuda().to(torch.uint8).contiguous(),
                     mf.cuda().to(torch.uint8).contiguous(), True).cpu()
    sd = synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"])
    ref = otfam.amo_clip_forward(sd, rgb, mot, mr, mf, nhead=8)
    err = (y - ref).abs().max().item()
    print(f"fused B={B} {dtype}: max abs err {err:.3e} (|ref|max {ref.abs().max():.2f})")
    assert tf.supported(m, B, 16, 15, True) == (B * 16 <= tf.MAX_ROWS)
    assert err <= TOL[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("D,H,Tr,Tf,ragged", [(512, 8, 40, 39, True), (768, 8, 64, 63, True), (768, 8, 33, 20, False), (768, 12, 48, 64, True),
                                              (512, 8, 20, 50, True)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_clips_longer_than_32_tokens_run_through_the_chain(D, H, Tr, Tf, ragged, dtype, monkeypatch):
    """real Animal-Kingdom clips are longer than 32 tokens (the reference consumes whole videos).  Queries in parts of 32
    per row block, up to four key tiles of 16 (the V image shares its LDS region with the W tile), fragment records of ceil(T / 16)
    token tiles: T and Tk up to 64 against the CPU oracle, ragged masks, d_model 512 and 768, head_dim 64 and 96."""
    from vimo_clip_amd import tfam_fused as tf
    c = dict(name="long", D=D, H=H, L=2, ff=1024, C=140, B=3, Tr=Tr, Tf=Tf, mode="cross", pe=False, ragged=ragged, seed=970 + Tr)
    m = _tfam(c, dtype)
    calls = []
    orig = tf.TfamPack.forward
    monkeypatch.setattr(tf.TfamPack, "forward", lambda self, *a, **k: calls.append(1) or orig(self, *a, **k))
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    with torch.no_grad():
        y = m(rgb.cuda(), mot.cuda(), mask_rgb=mr.cuda(), mask_flow=mf.cuda()).cpu()
    assert calls, "the fused chain was not used"
    sd = synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"])
    ref = otfam.amo_clip_forward(sd, rgb, mot, mr, mf, nhead=H)
    err = (y - ref).abs().max().item()
    print(f"fused T={Tr} Tk={Tf} D={D} H={H} {dtype}: max abs err {err:.3e} (|ref|max {ref.abs().max():.2f})")
    assert err <= TOL[dtype] * max(1.0, ref.abs().max().item())


def test_long_clips_T32_and_no_masks():
    c = dict(name="t32", D=512, H=8, L=2, ff=2048, C=140, B=3, Tr=32, Tf=31, mode="cross", pe=False, ragged=False, seed=950)
    m = _tfam(c, torch.float16)
    rgb, mot, _, _ = mg.tfam_inputs(c)
    with torch.no_grad():
        y = m(rgb.cuda(), mot.cuda()).cpu()
    sd = synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"])
    ref = otfam.amo_clip_forward(sd, rgb, mot, None, None, nhead=8)
    err = (y - ref).abs().max().item()
    print(f"fused T=32 no masks: max abs err {err:.3e}")
    assert err <= 1e-3 * max(1.0, ref.abs().max().item())


def test_layerwise_entry_points_equal_the_single_call():
    L_ = _lib()
    lib, ptr, stream, check, dt = L_.lib, L_.ptr, L_.stream, L_.check, L_.dt
    from vimo_clip_amd import tfam_fused as tf
    c = mg.TFAM_CASES[1]
    m = _tfam(c, torch.bfloat16)
    rgb, mot, mr, mf = mg.tfam_inputs(c)
    x, mo = rgb.cuda().contiguous(), mot.cuda().contiguous()
    m8, f8 = mr.cuda().to(torch.uint8).contiguous(), mf.cuda().to(torch.uint8).contiguous()
    pack = tf.get_pack(m, torch.bfloat16).refresh()
    one = pack.forward(x, mo, m8, f8, True)
    B, T, D = x.shape
    Tk = mo.shape[1]
    ws = pack.workspace(B, T, Tk, True)
    dims = (B, T, Tk, D, pack.H, pack.ff, pack.L, pack.C)
    check(lib.vmc_tfam_kv_fwd(ptr(mo), ptr(pack.wpack), ptr(pack.ppack), ptr(ws), ws.numel(), *dims, dt(torch.bfloat16), stream()), "kv")
    for layer in range(pack.L):
        check(lib.vmc_tfam_layer_fwd(ptr(x) if layer == 0 else None, ptr(m8), ptr(f8), ptr(pack.wpack), ptr(pack.ppack), layer,
                                     ptr(ws), ws.numel(), *dims, 1, dt(torch.bfloat16), stream()), "layer")
    out = torch.empty_like(one)
    check(lib.vmc_tfam_head_fwd(ptr(pack.wpack), ptr(pack.ppack), ptr(out), ptr(ws), ws.numel(), *dims, 1, dt(torch.bfloat16), stream()), "head")
    assert torch.equal(out, one)
    # argument checking: unsupported shapes are refused, not mis-executed
    assert lib.vmc_tfam_layer_fwd(None, None, None, ptr(pack.wpack), ptr(pack.ppack), 0, ptr(ws), ws.numel(), *dims, 1, 1, stream()) == -1
    assert lib.vmc_tfam_kv_fwd(ptr(mo), ptr(pack.wpack), ptr(pack.ppack), ptr(ws), 16, *dims, 1, stream()) == -1
    bad = (B, 70, Tk, D, pack.H, pack.ff, pack.L, pack.C)      # more than TF_MAX_T = 64 tokens per clip
    assert lib.vmc_tfam_head_fwd(ptr(pack.wpack), ptr(pack.ppack), ptr(out), ptr(ws), ws.numel(), *bad, 1, 1, stream()) == -3


def test_graph_replay_sees_optimizer_updates():
    """a captured eval graph must not keep reading weight copies an optimiser step has invalidated.  The packs
    are persistent buffers rewritten in place; after FusedAdam.step + refresh, a replay equals a fresh eager forward."""
    from vimo_clip_amd import tfam_fused as tf
    from vimo_clip_amd.graphs import GraphedCallable
    from vimo_clip_amd.losses import bce_with_logits_loss
    from vimo_clip_amd.optim import FusedAdam, GradArena
    c = mg.TFAM_CASES[0]
    m = _tfam(c, torch.bfloat16)
    rgb, mot, mr, mf = (t.cuda() for t in mg.tfam_inputs(c))

    def fwd(a, b, cm, d):
        with torch.no_grad():
            return m(a, b, mask_rgb=cm, mask_flow=d)

    g = GraphedCallable(fwd, rgb, mot, mr, mf)
    before = g(rgb, mot, mr, mf).clone()
    assert torch.equal(before, fwd(rgb, mot, mr, mf))
    arena = GradArena(m.used_parameters())
    opt = FusedAdam(arena, lr=1e-2, weight_decay=0.1, decoupled=True)
    m.train()
    y = synth.multi_hot_labels(c["seed"], "labels", c["B"], c["C"]).cuda()
    bce_with_logits_loss(m(rgb, mot, mask_rgb=mr, mask_flow=mf), y).backward()
    opt.step()
    m.eval()
    pack = tf.get_pack(m, torch.bfloat16)
    assert not pack.pack_is_current()
    pack.refresh()
    after = g(rgb, mot, mr, mf).clone()
    eager = fwd(rgb, mot, mr, mf)
    assert not torch.equal(after, before)
    assert torch.equal(after, eager)
