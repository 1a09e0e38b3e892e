"""GPU: every libvmc kernel against a plain PyTorch fp32 CPU reference of the same op (called through
the C ABI via vimo_clip_amd.ops).  Integer-valued operands make the MFMA GEMM checks bit-exact."""
import math

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
DT16 = [torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from vimo_clip_amd import ops as _ops
    return _ops


def _ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


# ---------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [
    (256, 256, 64), (512, 1024, 128),          # whole 256x256 tiles (identity-free, asymmetric W)
    (65792 // 16, 1024, 1024),                 # 4112 rows: 128x128 config with a ragged last row tile
    (3333, 4096, 128), (3100, 3072, 64),       # 256x256 tiles (>= 192): 8-phase kernel (K % 128 == 0) / two-stage kernel
    (16448, 1024, 1024), (3333, 4100, 256), (65792, 1024, 128),   # 8-phase: many K tiles, ragged N, the ViT-L/14 row count
    (4100, 2304, 256), (8224, 1024, 128),      # 129 .. 191 tiles of 256x256 with more than 512 tiles of 128x128: one 8-phase round
    (300, 140, 64), (50, 768, 192), (8, 384, 768), (129, 2304, 768), (1000, 64, 3072), (16448, 256, 640),
])
def test_linear_exact_integers(ops, dtype, M, N, K):
    a = _ints((M, K), -3, 3, 1)
    w = _ints((N, K), -2, 2, 2)
    w[:, 0] += torch.arange(N).float() % 5      # asymmetric in n
    a[:, 1] += torch.arange(M).float() % 3      # asymmetric in m
    ref = a @ w.t()
    assert ref.abs().max() < 2 ** 24
    out = ops.linear(a.to(DEV, dtype), w.to(DEV, dtype), out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref), f"max diff {(out.cpu() - ref).abs().max()}"


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,act", [
    (65536, 256, 256, 0),        # 256 tiles, one per workgroup: head + roll iterations only (K = 4 K tiles)
    (16384, 4096, 384, 1),       # 1024 tiles = 4 per workgroup, head / steady / roll, QuickGELU
    (65792, 1024, 512, 3),       # ViT-L/14 row count: 1024 persistent tiles + the 256-row tail launch, ReLU
    (8192, 8192, 1024, 0),       # 1024 tiles, 16 K tiles
    (21760, 768, 768, 0),        # 255 tiles (the main part of the student's 300-tile GEMMs): fewer tiles than workgroups, one walk each
])
def test_linear_persistent_walk(ops, dtype, M, N, K, act):
    """Persistent 8-phase GEMM (whole 256x256 tiles, bias, 16-bit output: what vmc_linear takes for the encoder's big linears).
    (1) integer operands: exact against the fp32 CPU product for every tile a workgroup walks (a wrong prefetch / a quadrant
    stored before its last MFMA shows as wrong values); (2) random operands: bit-identical to the one-tile-per-workgroup kernel
    (VMC_GEMM_ONE_TILE) on repeated launches -- the vmcnt accounting differs per iteration kind, a race shows as rare tiles."""
    if act != 1:
        a = _ints((M, K), -1, 1, 11)
        w = _ints((N, K), -1, 1, 12)
        w[:, 0] += torch.arange(N).float() % 3
        a[:, 1] += torch.arange(M).float() % 2
        bias = (torch.arange(N).float() % 7) - 3
        z = (a.to(DEV) @ w.to(DEV).t() + bias.to(DEV))
        z = torch.relu(z) if act == 3 else z
        assert z.abs().max().item() <= 256             # integers up to 256 are exact in bf16 and f16
        out = ops.linear(a.to(DEV, dtype), w.to(DEV, dtype), bias=bias.to(DEV), act=act, out_dtype=dtype)
        assert torch.equal(out.float(), z), f"max diff {(out.float() - z).abs().max()}"
    g = torch.Generator(device=DEV).manual_seed(M + K)
    ar = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    wr = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dtype)
    br = torch.randn(N, device=DEV, generator=g)
    ref = ops.linear(ar, wr, bias=br, act=act, alpha=0.75, out_dtype=dtype, variant=4)      # VMC_GEMM_ONE_TILE
    for _ in range(3):
        got = ops.linear(ar, wr, bias=br, act=act, alpha=0.75, out_dtype=dtype)              # default: persistent walk
        assert torch.equal(got, ref)
    if act == 0:        # bias-free instantiation (input-gradient GEMMs of training)
        ref = ops.linear(ar, wr, out_dtype=dtype, variant=4)
        for _ in range(3):
            assert torch.equal(ops.linear(ar, wr, out_dtype=dtype), ref)
    if act == 1:        # pre-activation side output (training forward of QuickGELU linears): 8 stores per quadrant
        z_ref = ops.linear(ar, wr, bias=br, out_dtype=dtype, variant=4)              # the value before the activation
        y_ref = ops.linear(ar, wr, bias=br, act=1, out_dtype=dtype, variant=4)
        for _ in range(3):
            z = torch.zeros(M, N, device=DEV, dtype=dtype)
            y = ops.linear(ar, wr, bias=br, act=1, out_dtype=dtype, z_out=z)
            assert torch.equal(y, y_ref) and torch.equal(z, z_ref)


def test_linear_k_slice_groups_in_a_subprocess():
    """gemm_kernel's K-slice groups (VMC_GEMM_KS=1: measured, not routed by default -- DESIGN 3.1 / profiles/README.md) stay correct:
    the switch is read once per process, so the check runs in a child process.  Integer operands: exact whatever the summation order;
    the 256-row tail shapes of the encoder (32-row tiles with four groups, 64^2 tiles with two) with bias, QuickGELU and fp32 residual."""
    import subprocess
    import sys
    code = r'''
import torch
from vimo_clip_amd import ops
g = torch.Generator().manual_seed(3)
for (M, N, K, act, res32) in [(256, 1024, 1024, 0, True), (256, 1024, 4096, 0, True), (256, 3072, 1024, 0, False), (256, 4096, 1024, 3, False), (128, 768, 2048, 0, False)]:
    for dt in (torch.bfloat16, torch.float16):
        a = torch.randint(-1, 2, (M, K), generator=g).float()
        w = torch.randint(-1, 2, (N, K), generator=g).float()
        b = (torch.arange(N).float() % 7) - 3
        z = a.cuda() @ w.cuda().t() + b.cuda()
        if act == 3:
            z = torch.relu(z)
        assert z.abs().max().item() <= 256
        if res32:
            x0 = torch.randint(-8, 9, (M, N), generator=g).float().cuda()
            out = ops.linear(a.cuda().to(dt), w.cuda().to(dt), bias=b.cuda(), out=x0.clone(), res=x0)
            assert torch.equal(out, x0 + z), (M, N, K, dt)
        else:
            out = ops.linear(a.cuda().to(dt), w.cuda().to(dt), bias=b.cuda(), act=act, out_dtype=dt)
            assert torch.equal(out.float(), z), (M, N, K, dt)
print("k-slice groups ok")
'''
    env = dict(os.environ, VMC_GEMM_KS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "k-slice groups ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,act", [(16384, 4096, 384, 1), (8192, 8192, 1024, 0), (21760, 768, 768, 0)])
def test_linear_persistent_walk_mfma32(ops, dtype, M, N, K, act):
    """The persistent walk on 32x32x16 MFMA fragments (VMC_GEMM_MFMA32; kept as the measured, slower alternative of DESIGN 3.1):
    integer operands exact, random operands equal to the 16x16x32 form (they agree bit for bit on this hardware) on repeats."""
    if act == 0:
        a = _ints((M, K), -1, 1, 21)
        w = _ints((N, K), -1, 1, 22)
        w[:, 0] += torch.arange(N).float() % 3
        bias = (torch.arange(N).float() % 7) - 3
        z = a.to(DEV) @ w.to(DEV).t() + bias.to(DEV)
        assert z.abs().max().item() <= 256
        out = ops.linear(a.to(DEV, dtype), w.to(DEV, dtype), bias=bias.to(DEV), out_dtype=dtype, variant=5)
        assert torch.equal(out.float(), z), f"max diff {(out.float() - z).abs().max()}"
    g = torch.Generator(device=DEV).manual_seed(M + K + 5)
    ar = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    wr = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dtype)
    br = torch.randn(N, device=DEV, generator=g)
    ref = ops.linear(ar, wr, bias=br, act=act, alpha=0.75, out_dtype=dtype)
    for _ in range(3):
        assert torch.equal(ops.linear(ar, wr, bias=br, act=act, alpha=0.75, out_dtype=dtype, variant=5), ref)
    if act == 0:
        ref = ops.linear(ar, wr, out_dtype=dtype)
        assert torch.equal(ops.linear(ar, wr, out_dtype=dtype, variant=5), ref)


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("res_f32,out_f32", [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize("shape", [(333, 200, 128), (3900, 4096, 256)], ids=["small", "g8"])
def test_linear_epilogue(ops, dtype, act, res_f32, out_f32, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.2).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(torch.float32 if res_f32 else dtype)
    z = a.float() @ w.float().t() + bias
    if act == 1:
        z = z * torch.sigmoid(1.702 * z)
    elif act == 2:
        z = torch.nn.functional.gelu(z)
    elif act == 3:
        z = torch.relu(z)
    ref = 0.5 * z + res.float()
    out = ops.linear(a.to(DEV), w.to(DEV), bias=bias.to(DEV), res=res.to(DEV), act=act, alpha=0.5,
                     out_dtype=torch.float32 if out_f32 else dtype)
    tol = 2e-5 if out_f32 else (1e-2 if dtype == torch.bfloat16 else 2e-3)
    torch.testing.assert_close(out.float().cpu(), ref, atol=tol * max(1.0, ref.abs().max().item()), rtol=tol)


def test_linear_patch_rows_and_posemb(ops):
    # out_row_group / res_row_mod: patch rows -> token rows with positional-embedding broadcast (K1)
    F, g2, D, K = 3, 16, 128, 64
    dtype = torch.bfloat16
    a = _ints((F * g2, K), -2, 2, 7)
    w = _ints((D, K), -2, 2, 8)
    pos = _ints((g2 + 1, D), -4, 4, 9)
    x = torch.full((F * (g2 + 1), D), -77.0, device=DEV)
    ops.linear(a.to(DEV, dtype), w.to(DEV, dtype), res=pos.to(DEV)[1:], out=x, out_row_group=g2, res_row_mod=g2)
    ref = torch.full((F, g2 + 1, D), -77.0)
    ref[:, 1:] = (a @ w.t()).view(F, g2, D) + pos[1:]
    assert torch.equal(x.cpu().view(F, g2 + 1, D), ref)


def test_linear_bad_args(ops):
    a = torch.zeros(8, 60, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(16, 60, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="libvmc linear failed"):
        ops.linear(a, w)


# ---------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,D,xf32", [(1000, 1024, True), (77, 768, False), (5, 128, True), (4099, 512, False)])
def test_layernorm_fwd(ops, dtype, rows, D, xf32):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(rows, D, generator=g) * 3 + 1).to(torch.float32 if xf32 else dtype)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), gamma, beta, 1e-5)
    y16, y32, mean, rstd = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), dtype, out16=True, out32=True, save_stats=True)
    torch.testing.assert_close(y32.cpu(), ref, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(y16.float().cpu(), ref.to(dtype).float(), atol=2e-2 if dtype == torch.bfloat16 else 2e-3, rtol=1e-2)
    torch.testing.assert_close(mean.cpu(), x.float().mean(-1), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(rstd.cpu(), (x.float().var(-1, unbiased=False) + 1e-5).rsqrt(), atol=1e-5, rtol=1e-4)


def test_layernorm_strided_rows_inplace(ops):
    # CLS-row gather (ldx = N*D) and in-place fp32 (ln_pre)
    F, N, D = 6, 5, 256
    g = torch.Generator().manual_seed(4)
    x = torch.randn(F * N, D, generator=g)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    xd = x.to(DEV)
    y16, *_ = ops.layernorm(xd, gamma.to(DEV), beta.to(DEV), torch.float16, rows=F, ldx=N * D)
    ref = torch.nn.functional.layer_norm(x.view(F, N, D)[:, 0], (D,), gamma, beta, 1e-5)
    torch.testing.assert_close(y16.float().cpu(), ref, atol=3e-3, rtol=3e-3)
    ops.layernorm(xd, gamma.to(DEV), beta.to(DEV), torch.float16, out16=False, out32=True, y32=xd)
    torch.testing.assert_close(xd.cpu(), torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5), atol=2e-5, rtol=2e-5)


# ---------------------------------------------------------------- attention
def _attn_ref(q, k, v, mask=None):
    dh = q.shape[-1]
    s = (q * dh ** -0.5) @ k.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("F,N,H", [(3, 257, 2), (2, 50, 12), (2, 197, 3), (5, 17, 2), (1, 5, 2)])
def test_attention_vit(ops, dtype, F, N, H):
    D = H * 64
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(F * N, 3 * D, generator=g).to(dtype)
    qkv[3 % (F * N), D:D + 64] *= 6.0                      # a spiky key: softmax far from uniform
    q, k, v = [t.float().view(F, N, H, 64).transpose(1, 2) for t in qkv.split(D, dim=1)]
    ref = _attn_ref(q, k, v).transpose(1, 2).reshape(F * N, D)
    out, lse = ops.attention_vit(qkv.to(DEV), F, N, H, want_lse=True)
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    torch.testing.assert_close(out.float().cpu(), ref, atol=tol, rtol=tol)
    lse_ref = torch.logsumexp((q * 0.125) @ k.transpose(-1, -2), -1)
    torch.testing.assert_close(lse.cpu(), lse_ref, atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("B,H,Tq,Tk,dh", [(4, 8, 16, 15, 96), (3, 8, 16, 16, 64), (2, 8, 40, 39, 64), (2, 2, 100, 257, 64)])
def test_attention_generic_masked(ops, dtype, B, H, Tq, Tk, dh):
    D = H * dh
    g = torch.Generator().manual_seed(12)
    q = torch.randn(B * Tq, D, generator=g).to(dtype)
    kv = torch.randn(B * Tk, 2 * D, generator=g).to(dtype)
    lens = torch.randint(1, Tk + 1, (B,), generator=g)
    lens[0] = Tk
    mask = torch.arange(Tk)[None, :] < lens[:, None]
    qd, kvd = q.to(DEV), kv.to(DEV)
    out, lse = ops.attention(qd, kvd[:, :D], kvd[:, D:], mask.to(torch.uint8).to(DEV), B, H, Tq, Tk, dh, want_lse=True)
    qf = q.float().view(B, Tq, H, dh).transpose(1, 2)
    kf = kv[:, :D].float().view(B, Tk, H, dh).transpose(1, 2)
    vf = kv[:, D:].float().view(B, Tk, H, dh).transpose(1, 2)
    ref = _attn_ref(qf, kf, vf, mask).transpose(1, 2).reshape(B * Tq, D)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    torch.testing.assert_close(out.float().cpu(), ref, atol=tol, rtol=tol)


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("B,H,Tq,Tk,dh", [(3, 8, 16, 16, 96), (2, 4, 20, 33, 64), (2, 8, 16, 64, 96), (2, 2, 40, 200, 64), (1, 2, 150, 197, 64)])
def test_attention_with_probability_dropout(ops, dtype, B, H, Tq, Tk, dh):
    """Dropout on the attention probabilities (nn.MultiheadAttention(dropout=p), training) in the MFMA kernels (short sequences;
    the long ones are past what the backward's prologue prefetches in registers: its chunk-by-chunk remainder loops for Q | dO,
    K | V and the delta rows): forward and the three gradients against torch autograd with the SAME
    mask, which is vmc_dropout's counter-based mask on the flat (batch, head, query, key) index -- obtained by running vmc_dropout
    on a tensor of ones."""
    from vimo_clip_amd import autograd_ops as ag
    from vimo_clip_amd._lib import check, dt, lib, ptr, stream
    D, p, seed = H * dh, 0.25, 0x1234567
    g = torch.Generator().manual_seed(21)
    q = torch.randn(B * Tq, D, generator=g).to(dtype)
    kv = torch.randn(B * Tk, 2 * D, generator=g).to(dtype)
    dout = torch.randn(B * Tq, D, generator=g).to(dtype)
    lens = torch.randint(1, Tk + 1, (B,), generator=g)
    lens[0] = Tk
    mask = torch.arange(Tk)[None, :] < lens[:, None]
    ones = torch.ones(B * H * Tq * Tk, device=DEV)
    fac = torch.empty_like(ones)
    check(lib.vmc_dropout(ptr(ones), ptr(fac), ones.numel(), p, seed, dt(ones), dt(dtype), stream()), "dropout")
    fac = fac.view(B, H, Tq, Tk).cpu().double()
    assert 0.1 < (fac == 0).double().mean().item() < 0.4
    # reference in float64 with autograd
    qf = q.double().view(B, Tq, H, dh).transpose(1, 2).requires_grad_(True)
    kf = kv[:, :D].double().view(B, Tk, H, dh).transpose(1, 2).requires_grad_(True)
    vf = kv[:, D:].double().view(B, Tk, H, dh).transpose(1, 2).requires_grad_(True)
    sc = (qf @ kf.transpose(-1, -2)) / math.sqrt(dh)
    sc = sc.masked_fill(~mask[:, None, None, :], float("-inf"))
    o_ref = ((torch.softmax(sc, dim=-1) * fac) @ vf).transpose(1, 2).reshape(B * Tq, D)
    o_ref.backward(dout.double())
    # device
    qd, kvd, dd, md = q.to(DEV), kv.to(DEV), dout.to(DEV), mask.to(torch.uint8).to(DEV)
    out, lse = ops.attention(qd, kvd[:, :D], kvd[:, D:], md, B, H, Tq, Tk, dh, want_lse=True, dropout_p=p, dropout_seed=seed)
    dq, dkv = torch.empty_like(qd), torch.empty_like(kvd)
    ag._attn_bwd(qd, kvd[:, :D], kvd[:, D:], md, out, dd, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Tq, Tk, dh, p=p, seed=seed)
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    flat = lambda t, T_: t.detach().transpose(1, 2).reshape(B * T_, D)
    for name, got, ref in (("out", out, o_ref.detach()), ("dq", dq, flat(qf.grad, Tq)), ("dk", dkv[:, :D], flat(kf.grad, Tk)), ("dv", dkv[:, D:], flat(vf.grad, Tk))):
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (name, err)


def test_attention_vit_matches_generic(ops):
    F, N, H = 2, 257, 4
    D = H * 64
    qkv = torch.randn(F * N, 3 * D, generator=torch.Generator().manual_seed(13)).to(torch.float16).to(DEV)
    a, _ = ops.attention_vit(qkv, F, N, H)
    b, _ = ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], None, F, H, N, N, 64)
    torch.testing.assert_close(a.float(), b.float(), atol=3e-3, rtol=3e-3)


# ---------------------------------------------------------------- preprocess / misc
@pytest.mark.parametrize("R,p", [(224, 14), (224, 32), (224, 16), (64, 16)])
@pytest.mark.parametrize("wrap", [False, True])
def test_preprocess_patches(ops, R, p, wrap):
    from oracle import vit as ovit
    F = 2
    g = torch.Generator().manual_seed(21)
    fr = torch.randint(0, 256, (F, 3, R, R), generator=g, dtype=torch.uint8)
    src = ovit.to_pil_wrap_u8(fr) if wrap else fr
    pix = ovit.normalize_u8(src)
    gg = R // p
    ref = pix.view(F, 3, gg, p, gg, p).permute(0, 2, 4, 1, 3, 5).reshape(F * gg * gg, 3 * p * p)
    out = ops.preprocess_patches_u8(fr.to(DEV), p, torch.float16, wrap).float().cpu()
    k = 3 * p * p
    assert out.shape[1] % 64 == 0 and torch.all(out[:, k:] == 0)
    torch.testing.assert_close(out[:, :k], ref.to(torch.float16).float(), atol=1e-3, rtol=1e-3)
    out2 = ops.patches_f32(pix.to(DEV), p, torch.float16).float().cpu()
    assert torch.equal(out2[:, :k], ref.to(torch.float16).float())


def test_transpose_colsum_cast_pool_pe(ops):
    g = torch.Generator().manual_seed(22)
    x = torch.randn(203, 77, generator=g).to(torch.bfloat16)
    assert torch.equal(ops.transpose16(x.to(DEV)).cpu(), x.t().contiguous())
    y = torch.randn(1030, 140, generator=g)
    torch.testing.assert_close(ops.colsum(y.to(DEV)).cpu(), y.sum(0), atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(ops.colsum(y.to(torch.float16).to(DEV)).cpu(), y.to(torch.float16).float().sum(0), atol=1e-3, rtol=1e-4)
    w = torch.randn(70, 588, generator=g)
    c = ops.cast_weight(w.to(DEV), torch.bfloat16, pad_k=True)
    assert c.shape == (70, 640) and torch.equal(c[:, :588].cpu(), w.to(torch.bfloat16)) and torch.all(c[:, 588:] == 0)
    ct = ops.cast_weight(w.to(DEV), torch.float16, transposed=True)
    assert torch.equal(ct.cpu(), w.t().contiguous().to(torch.float16))
    z = torch.randn(3, 16, 768, generator=g)
    _, p32 = ops.mean_pool(z.to(DEV), 3, 16, 768, torch.bfloat16, out16=False, out32=True)
    torch.testing.assert_close(p32.cpu(), z.mean(1), atol=1e-6, rtol=1e-5)
    from oracle import tfam
    e = torch.zeros(2, 40, 512, device=DEV)
    ops.add_sinusoidal_pe_(e)
    torch.testing.assert_close(e.cpu()[0], tfam.positional_encoding(40, 512), atol=2e-5, rtol=0)
    for act in (1, 2, 3):
        v = torch.randn(1000, generator=g).to(torch.float16)
        ref = {1: v.float() * torch.sigmoid(1.702 * v.float()), 2: torch.nn.functional.gelu(v.float()), 3: torch.relu(v.float())}[act]
        torch.testing.assert_close(ops.act_fwd(v.to(DEV), act).float().cpu(), ref, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(25600, 768, 768), (1000, 2304, 768), (77, 144, 3072), (8192, 768, 2048), (130, 64, 40), (4096, 512, 2048), (65536, 512, 512),
                                   (25600, 3072, 768), (25600, 2304, 768), (6400, 1000, 2056), (256, 1024, 1024)])   # the last four: 256 x 256 tile (gemm_tn256.hip), ragged N / K, one stage pair
def test_wgrad_tn_exact_integers(ops, dtype, M, N, K):
    # dW = dY^T X straight from token-major operands (ds_read_b64_tr_b16 fragments, split over the token range)
    dy = _ints((M, N), -2, 2, 31)
    x = _ints((M, K), -3, 3, 32)
    dy[:, 1] += (torch.arange(M) % 3).float()
    x[:, 2] += (torch.arange(M) % 2).float()
    ref = dy.t() @ x
    assert ref.abs().max() < 2 ** 24
    out = ops.wgrad_tn(dy.to(DEV, dtype), x.to(DEV, dtype), torch.empty(N, K, device=DEV))
    assert torch.equal(out.cpu(), ref), f"max diff {(out.cpu() - ref).abs().max()}"


@pytest.mark.parametrize("M,N,K", [(512, 2048, 4096), (512, 512, 65536), (768, 768, 25600), (64, 144, 1088), (768, 3072, 1024)])
def test_linear_wgrad_splitk_exact_integers(ops, M, N, K):
    # NT split-K path (vmc_linear_splitk_f32): every slab written (no empty trailing slice), deterministic reduce
    a = _ints((M, K), -2, 2, 41)
    w = _ints((N, K), -3, 3, 42)
    ref = a @ w.t()
    assert ref.abs().max() < 2 ** 24
    out = ops.linear_wgrad(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), torch.empty(M, N, device=DEV))
    assert torch.equal(out.cpu(), ref), f"max diff {(out.cpu() - ref).abs().max()}"
    again = ops.linear_wgrad(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), torch.full((M, N), 7.0, device=DEV))
    assert torch.equal(again, out)


@pytest.mark.parametrize("M,N,K", [(25600, 768, 768), (1000, 2304, 768), (77, 144, 3072), (8192, 512, 2048), (25600, 3072, 768), (6400, 1000, 2056)])
def test_wgrad_tn_with_bias_gradient(ops, M, N, K):
    # the same launch also returns db = column sums of dY (ones-MFMA on the fragments already in registers)
    dy = _ints((M, N), -2, 2, 51)
    x = _ints((M, K), -3, 3, 52)
    out = torch.empty(N, K, device=DEV)
    db = torch.full((N,), 123.0, device=DEV)
    ops.wgrad_tn(dy.to(DEV, torch.bfloat16), x.to(DEV, torch.bfloat16), out, db)
    assert torch.equal(out.cpu(), dy.t() @ x)
    assert torch.equal(db.cpu(), dy.sum(0))


@pytest.mark.parametrize("dtype", DT16, ids=["bf16", "f16"])
def test_wgrad_tn_group_exact_integers(ops, dtype):
    """vmc_linear_wgrad_tn_group: several weight gradients (and their bias gradients) from one launch, every tile over all tokens of its
    problem -- different M, ragged N / K, strided operands (a column window of a wider tensor), with and without bias gradient."""
    shapes = [(8192, 768, 768, True), (8192, 2304, 768, True), (1024, 1000, 2056, False), (256, 64, 40, True), (4096, 768, 2048, True),
              (8192, 768, 768, False), (2560, 520, 264, True)]
    probs, refs = [], []
    for i, (M, N, K, with_b) in enumerate(shapes):
        dy = _ints((M, N), -2, 2, 71 + i)
        x = _ints((M, K), -3, 3, 91 + i)
        dy[:, 1] += (torch.arange(M) % 3).float()
        x[:, 2] += (torch.arange(M) % 2).float()
        ref = dy.t() @ x
        assert ref.abs().max() < 2 ** 24
        if i == 1:                                     # dY as a column window of a wider row-major tensor (lddy > N)
            wide = torch.zeros(M, N + 512)
            wide[:, 256:256 + N] = dy
            dyd = wide.to(DEV, dtype)[:, 256:256 + N]
        else:
            dyd = dy.to(DEV, dtype)
        out = torch.full((N, K), 7.0, device=DEV)
        db = torch.full((N,), 5.0, device=DEV) if with_b else None
        assert ops.wgrad_group_ok(dyd, x.to(DEV, dtype), out)
        probs.append((dyd, x.to(DEV, dtype), out, db))
        refs.append((ref, dy.sum(0) if with_b else None))
    ops.wgrad_tn_group(probs, dtype)
    for (dyd, xd, out, db), (ref, rb) in zip(probs, refs):
        assert torch.equal(out.cpu(), ref), f"max diff {(out.cpu() - ref).abs().max()}"
        if rb is not None:
            assert torch.equal(db.cpu(), rb)
    assert not ops.wgrad_group_ok(torch.zeros(200, 64, device=DEV, dtype=dtype), torch.zeros(200, 64, device=DEV, dtype=dtype),
                                  torch.zeros(64, 64, device=DEV))        # M not a multiple of 128: the sliced kernel's case


@pytest.mark.parametrize("M,N,K,act", [(65792, 1024, 128, 1), (25600, 768, 256, 2), (300, 96, 64, 1), (4096, 512, 128, 3), (777, 264, 192, 1)])
def test_linear_preact_side_output(ops, M, N, K, act):
    # vmc_linear_preact: C = act(A W^T + b) and Z = A W^T + b from one epilogue (8-phase interior tiles, its 256-row tail,
    # the small-tile kernels and ragged edges) == the two separate launches
    a = (_ints((M, K), -2, 2, 61) * 0.25).to(DEV, torch.bfloat16)
    w = (_ints((N, K), -2, 2, 62) * 0.25).to(DEV, torch.bfloat16)
    bias = (_ints((N,), -3, 3, 63) * 0.5).to(DEV)
    z_ref = ops.linear(a, w, bias=bias)
    y_ref = ops.linear(a, w, bias=bias, act=act)
    z = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    y = ops.linear(a, w, bias=bias, act=act, z_out=z)
    assert torch.equal(z, z_ref) and torch.equal(y, y_ref)


def test_cast_weights_multi_refresh_matches_single_casts(ops):
    """vmc_cast_weights_multi (all cached 16-bit copies of the trained parameters re-cast in place after an optimiser step) ==
    vmc_cast_weight per parameter, for aligned and unaligned masters, ragged shapes (vector and scalar paths), padded copies."""
    from vimo_clip_amd import autograd_ops as ag
    ag.weights.clear()
    flat = torch.randn(4 * 1024 * 1024, device=DEV)
    shapes = [(768, 768), (2304, 768), (140, 768), (768, 140), (77, 130), (64, 3), (1000, 36)]
    params, off = [], 0
    for i, (r, c) in enumerate(shapes):
        off += (i % 2)                               # every other master starts at an odd float offset (unaligned)
        params.append(torch.nn.Parameter(flat[off:off + r * c].view(r, c)))
        off += r * c

    def copies(dtype, p):
        r, c = p.shape
        return (ag.weights.get(p, dtype, pad_k=(c % 64 != 0), both=True), ag.weights.get(p, dtype, transposed=True, pad_k=(r % 64 != 0)))

    before = {(dtype, i): copies(dtype, p) for dtype in DT16 for i, p in enumerate(params)}
    # rewrite the masters the way the optimiser kernels do: through the raw storage, behind autograd's version counters
    flat.untyped_storage().copy_(torch.randn_like(flat).untyped_storage())
    epoch = ag.weights.epoch
    ag.weights.refresh()
    assert ag.weights.epoch == epoch + 1
    for dtype in DT16:
        for i, p in enumerate(params):
            w, wt = copies(dtype, p)
            assert w.data_ptr() == before[(dtype, i)][0].data_ptr() and wt.data_ptr() == before[(dtype, i)][1].data_ptr()   # refreshed in place
            ref, ref_t = ops.cast_weight_both(p, dtype)
            assert torch.equal(w, ref) and torch.equal(wt, ref_t), (dtype, tuple(p.shape))
    ag.weights.clear()
