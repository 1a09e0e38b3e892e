"""Checked Python entry points over the libvmc C ABI (include/vmc.h).

PyTorch is used for device memory and streams only: every function here enqueues HIP kernels on the
current stream and returns tensors allocated by the caching allocator.  No function has a PyTorch
compute fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import ACT_GELU_ERF, ACT_NONE, ACT_QUICKGELU, ACT_RELU, F32, check, dt, lib, ptr, stream  # noqa: F401


def _kpad(k: int) -> int:
    return (k + 63) // 64 * 64


def cast_weight(w: torch.Tensor, dtype16: torch.dtype, transposed: bool = False, pad_k: bool = False):
    """fp32 [rows, cols] parameter -> 16-bit compute copy (optionally the transposed copy instead).
    pad_k pads the K (= last) dimension of the copy with zeros up to a multiple of 64."""
    w2 = w.detach().reshape(w.shape[0], -1)
    if w2.dtype != torch.float32 or not w2.is_contiguous():
        w2 = w2.float().contiguous()
    rows, cols = w2.shape
    if transposed:
        ld = _kpad(rows) if pad_k else rows
        out = (torch.zeros if ld != rows else torch.empty)((cols, ld), dtype=dtype16, device=w.device)
        check(lib.vmc_cast_weight(ptr(w2), None, ptr(out), rows, cols, 0, ld, dt(dtype16), stream()), "cast_weight")
    else:
        ld = _kpad(cols) if pad_k else cols
        out = (torch.zeros if ld != cols else torch.empty)((rows, ld), dtype=dtype16, device=w.device)
        check(lib.vmc_cast_weight(ptr(w2), ptr(out), None, rows, cols, ld, 0, dt(dtype16), stream()), "cast_weight")
    return out


def cast_weight_both(w: torch.Tensor, dtype16: torch.dtype):
    """One launch for both 16-bit copies a training step needs: [rows, cols(+pad to 64)] for the forward and the
    transposed [cols, rows(+pad to 64)] for the dgrad (K padding only where the contraction length is not a multiple of 64)."""
    w2 = w.detach().reshape(w.shape[0], -1)
    if w2.dtype != torch.float32 or not w2.is_contiguous():
        w2 = w2.float().contiguous()
    rows, cols = w2.shape
    ld = _kpad(cols) if cols % 64 else cols
    ldt = _kpad(rows) if rows % 64 else rows
    out = (torch.zeros if ld != cols else torch.empty)((rows, ld), dtype=dtype16, device=w.device)
    out_t = (torch.zeros if ldt != rows else torch.empty)((cols, ldt), dtype=dtype16, device=w.device)
    check(lib.vmc_cast_weight(ptr(w2), ptr(out), ptr(out_t), rows, cols, ld, ldt, dt(dtype16), stream()), "cast_weight")
    return out, out_t


def cast16(x: torch.Tensor, dtype16: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype16:
        return x
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=dtype16, device=x.device)
    check(lib.vmc_cast_f32_to_16(ptr(x), ptr(y), x.numel(), dt(dtype16), stream()), "cast_f32_to_16")
    return y


def cast32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.float32:
        return x
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.vmc_cast_16_to_f32(ptr(x), ptr(y), x.numel(), dt(x), stream()), "cast_16_to_f32")
    return y


def linear(a: torch.Tensor, w16: torch.Tensor, bias=None, res=None, act: int = ACT_NONE, alpha: float = 1.0,
           out_dtype=None, out=None, out_row_group: int = 0, res_row_mod: int = 0, out_rows=None,
           lda=None, z_out=None, variant=None) -> torch.Tensor:
    """C = alpha * act(A @ W^T + bias) + res   (vmc_linear).  a: [M,K] 16-bit (row stride lda), w16: [N,K].
    z_out: optional 16-bit [M,N] tensor that receives A @ W^T + bias before the activation (vmc_linear_preact)."""
    M = a.shape[0]
    K = w16.shape[1]
    N = w16.shape[0]
    lda = a.stride(0) if lda is None else lda
    if a.stride(-1) != 1 or w16.stride(-1) != 1:
        raise ValueError("linear: operands must be contiguous in K")
    out_dtype = out_dtype or (out.dtype if out is not None else a.dtype)
    if out is None:
        out = torch.empty((out_rows or M, N), dtype=out_dtype, device=a.device)
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("linear: bias must be float32")
    if z_out is not None:
        check(lib.vmc_linear_preact(ptr(a), ptr(w16), ptr(bias), ptr(res), ptr(out), ptr(z_out), M, N, K, lda, w16.stride(0), out.stride(0),
                                    res.stride(0) if res is not None else 0, z_out.stride(0), act, float(alpha), dt(out),
                                    dt(res) if res is not None else 0, out_row_group, res_row_mod, dt(a), stream()), "linear_preact")
        return out
    if variant is not None:         # A/B measurements: large-problem kernel / epilogue options chosen per call (vmc_linear_variant)
        check(lib.vmc_linear_variant(ptr(a), ptr(w16), ptr(bias), ptr(res), ptr(out), M, N, K, lda, w16.stride(0), out.stride(0),
                                     res.stride(0) if res is not None else 0, act, float(alpha), dt(out), dt(res) if res is not None else 0,
                                     out_row_group, res_row_mod, dt(a), int(variant), stream()), "linear_variant")
        return out
    check(lib.vmc_linear(ptr(a), ptr(w16), ptr(bias), ptr(res), ptr(out), M, N, K, lda, w16.stride(0), out.stride(0),
                         res.stride(0) if res is not None else 0, act, float(alpha), dt(out), dt(res) if res is not None else 0,
                         out_row_group, res_row_mod, dt(a), stream()), "linear")
    return out


def linear_wgrad(a: torch.Tensor, w16: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[M,N] (f32, contiguous, overwritten) = a[M,K] @ w16[N,K]^T for weight gradients: split-K when the output is
    small and the contraction long (vmc_linear_splitk_f32), the plain tiled GEMM otherwise."""
    M, K = a.shape
    N = w16.shape[0]
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if out.is_contiguous() and tiles < 384 and K >= 1024:
        nbytes = lib.vmc_linear_splitk_workspace_bytes(M, N, K)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device) if nbytes else None
        check(lib.vmc_linear_splitk_f32(ptr(a), ptr(w16), ptr(out), M, N, K, a.stride(0), w16.stride(0), ptr(ws), nbytes, dt(a), stream()),
              "linear_splitk_f32")
        return out
    return linear(a, w16, out=out)


def wgrad_tn(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor, dbias: torch.Tensor = None) -> torch.Tensor:
    """out[N,K] (f32, contiguous, overwritten) = dy[M,N]^T @ x[M,K] (vmc_linear_wgrad_bias_tn): no transposed copies.
    dbias[N] (f32, optional, overwritten) = column sums of dy, from the same launch."""
    M, N = dy.shape
    K = x.shape[1]
    nbytes = lib.vmc_linear_wgrad_tn_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device) if nbytes else None
    check(lib.vmc_linear_wgrad_bias_tn(ptr(dy), ptr(x), ptr(out), ptr(dbias), M, N, K, dy.stride(0), x.stride(0), ptr(ws), nbytes, dt(dy),
                                       stream()), "linear_wgrad_bias_tn")
    return out


class WgradProblem(ctypes.Structure):          # vmc_wgrad_tn_problem (include/vmc.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("C", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("lddy", ctypes.c_int), ("ldx", ctypes.c_int),
                ("reserved", ctypes.c_int)]


WGRAD_GROUP_MAX = 32


def wgrad_group_ok(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor) -> bool:
    """Shapes vmc_linear_wgrad_tn_group takes (whole 128-token pairs, 8-column granularity, < 2 GiB operands, contiguous f32 output)."""
    M, N = dy.shape
    K = x.shape[1]
    return (M >= 256 and M % 128 == 0 and N % 8 == 0 and K % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
            and dy.stride(1) == 1 and x.stride(1) == 1 and out.is_contiguous() and out.dtype == torch.float32
            and M * dy.stride(0) * 2 < (1 << 31) and M * x.stride(0) * 2 < (1 << 31)
            and (dy.data_ptr() | x.data_ptr() | out.data_ptr()) % 16 == 0)


def wgrad_tn_group(problems, dtype16) -> None:
    """problems: list of (dy [M,N], x [M,K], out [N,K] f32, dbias [N] f32 or None), at most WGRAD_GROUP_MAX: every out = dy^T @ x and
    dbias = column sums of dy from ONE launch (vmc_linear_wgrad_tn_group: no token slices, no workspace, no reduce)."""
    n = len(problems)
    arr = (WgradProblem * n)()
    for i, (dy, x, out, db) in enumerate(problems):
        arr[i] = WgradProblem(dy.data_ptr(), x.data_ptr(), out.data_ptr(), db.data_ptr() if db is not None else None,
                              dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), 0)
    check(lib.vmc_linear_wgrad_tn_group(ctypes.cast(arr, ctypes.c_void_p), n, dt(dtype16), stream()), "linear_wgrad_tn_group")


def layernorm(x: torch.Tensor, gamma, beta, dtype16, *, out16=True, out32=False, y32=None, rows=None, ldx=None,
              eps: float = 1e-5, save_stats: bool = False):
    """LayerNorm over the last dim of x viewed as [rows, D] with row stride ldx.  Returns (y16, y32, mean, rstd)."""
    D = gamma.shape[0]
    rows = x.numel() // D if rows is None else rows
    ldx = D if ldx is None else ldx
    y16 = torch.empty((rows, D), dtype=dtype16, device=x.device) if out16 else None
    if out32 and y32 is None:
        y32 = torch.empty((rows, D), dtype=torch.float32, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    check(lib.vmc_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y16), ptr(y32), ptr(mean), ptr(rstd), rows, D, ldx,
                                float(eps), dt(x), dt(dtype16), stream()), "layernorm_fwd")
    return y16, y32, mean, rstd


def add_layernorm_(x32: torch.Tensor, branch16: torch.Tensor, gamma, beta, *, rows=None, ldx=None, ldb=None, write_x=True,
                   eps: float = 1e-5, branch0=None) -> torch.Tensor:
    """x32 += branch16 (in place, fp32 residual stream) and return LN(x32) in branch16's dtype (vmc_add_layernorm_fwd).
    branch0: a second 16-bit branch added FIRST, x32 = (x32 + branch0) + branch16 (vmc_add2_layernorm_fwd): the add an earlier
    call with write_x=False left out of the stored stream."""
    D = gamma.shape[0]
    rows = x32.numel() // D if rows is None else rows
    y = torch.empty((rows, D), dtype=branch16.dtype, device=x32.device)
    if branch0 is not None:
        check(lib.vmc_add2_layernorm_fwd(ptr(x32), ptr(branch0), ptr(branch16), ptr(gamma), ptr(beta), ptr(y), rows, D, ldx or D, D, ldb or D,
                                         float(eps), int(write_x), dt(branch16), stream()), "add2_layernorm_fwd")
        return y
    check(lib.vmc_add_layernorm_fwd(ptr(x32), ptr(branch16), ptr(gamma), ptr(beta), ptr(y), rows, D, ldx or D, ldb or D, float(eps),
                                    int(write_x), dt(branch16), stream()), "add_layernorm_fwd")
    return y


def attention_vit(qkv: torch.Tensor, F: int, N: int, H: int, want_lse: bool = False):
    D = H * 64
    out = torch.empty((F * N, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((F, H, N), dtype=torch.float32, device=qkv.device) if want_lse else None
    check(lib.vmc_attention_vit_fwd(ptr(qkv), ptr(out), ptr(lse), F, N, H, dt(qkv), stream()), "attention_vit_fwd")
    return out, lse


def attention_vit_cls(q_cls: torch.Tensor, kv: torch.Tensor, F: int, N: int, H: int) -> torch.Tensor:
    """Attention of the class-token queries only: q_cls [F, D], kv [F*N, 2D] (K | V of every token) -> [F, D]."""
    out = torch.empty((F, H * 64), dtype=kv.dtype, device=kv.device)
    check(lib.vmc_attention_vit_cls_fwd(ptr(q_cls), ptr(kv), ptr(out), F, N, H, dt(kv), stream()), "attention_vit_cls_fwd")
    return out


def attention(q, k, v, key_mask_u8, B, H, Tq, Tk, dh, want_lse=False, dropout_p=0.0, dropout_seed=0):
    """Generic masked attention.  q/k/v are 2-D 16-bit views [B*T, ld] whose first H*dh columns are used."""
    out = torch.empty((B * Tq, H * dh), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device) if want_lse else None
    check(lib.vmc_attention_fwd(ptr(q), ptr(k), ptr(v), ptr(key_mask_u8), ptr(out), ptr(lse), B, H, Tq, Tk, dh,
                                q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(dropout_p), int(dropout_seed),
                                dt(q), stream()), "attention_fwd")
    return out, lse


def preprocess_patches_u8(frames_u8: torch.Tensor, patch: int, dtype16, wrap_quirk: bool) -> torch.Tensor:
    F, C, R, R2 = frames_u8.shape
    if C != 3 or R != R2 or frames_u8.dtype != torch.uint8:
        raise ValueError("frames must be u8 [F,3,R,R]")
    frames_u8 = frames_u8.contiguous()
    g = R // patch
    kpad = _kpad(3 * patch * patch)
    out = torch.empty((F * g * g, kpad), dtype=dtype16, device=frames_u8.device)
    check(lib.vmc_preprocess_patches_u8(ptr(frames_u8), ptr(out), F, R, patch, kpad, int(wrap_quirk), dt(dtype16), stream()),
          "preprocess_patches_u8")
    return out


def patches_f32(pixel_values: torch.Tensor, patch: int, dtype16) -> torch.Tensor:
    F, C, R, R2 = pixel_values.shape
    if C != 3 or R != R2:
        raise ValueError("pixel_values must be [F,3,R,R]")
    pixel_values = pixel_values.float().contiguous()
    g = R // patch
    kpad = _kpad(3 * patch * patch)
    out = torch.empty((F * g * g, kpad), dtype=dtype16, device=pixel_values.device)
    check(lib.vmc_patches_f32(ptr(pixel_values), ptr(out), F, R, patch, kpad, dt(dtype16), stream()), "patches_f32")
    return out


def patches_u8_exact(frames_u8: torch.Tensor, patch: int, dtype16, wrap_quirk: bool) -> torch.Tensor:
    """[F*g*g, 2*kpad] = [v | v]: raw pixels as exact 16-bit integers (vmc_patches_u8_exact)."""
    F, C, R, R2 = frames_u8.shape
    if C != 3 or R != R2 or frames_u8.dtype != torch.uint8:
        raise ValueError("frames must be u8 [F,3,R,R]")
    frames_u8 = frames_u8.contiguous()
    g = R // patch
    kpad = _kpad(3 * patch * patch)
    out = torch.empty((F * g * g, 2 * kpad), dtype=dtype16, device=frames_u8.device)
    check(lib.vmc_patches_u8_exact(ptr(frames_u8), ptr(out), F, R, patch, kpad, int(wrap_quirk), dt(dtype16), stream()), "patches_u8_exact")
    return out


def patches_f32_split(pixel_values: torch.Tensor, patch: int, dtype16) -> torch.Tensor:
    """[F*g*g, 3*kpad] = [x_hi | x_lo | x_hi] (vmc_patches_f32_split)."""
    F, C, R, R2 = pixel_values.shape
    if C != 3 or R != R2:
        raise ValueError("pixel_values must be [F,3,R,R]")
    pixel_values = pixel_values.float().contiguous()
    g = R // patch
    kpad = _kpad(3 * patch * patch)
    out = torch.empty((F * g * g, 3 * kpad), dtype=dtype16, device=pixel_values.device)
    check(lib.vmc_patches_f32_split(ptr(pixel_values), ptr(out), F, R, patch, kpad, dt(dtype16), stream()), "patches_f32_split")
    return out


def set_class_rows(x: torch.Tensor, a, b, F: int, D: int, row_stride: int, dtype16):
    check(lib.vmc_set_class_rows(ptr(x), ptr(a), ptr(b), F, D, row_stride, dt(x), dt(dtype16), stream()), "set_class_rows")


def mean_pool(x: torch.Tensor, B: int, T: int, D: int, dtype16, out16=True, out32=False):
    o16 = torch.empty((B, D), dtype=dtype16, device=x.device) if out16 else None
    o32 = torch.empty((B, D), dtype=torch.float32, device=x.device) if out32 else None
    check(lib.vmc_mean_pool(ptr(x), ptr(o16), ptr(o32), B, T, D, dt(x), dt(dtype16), stream()), "mean_pool")
    return o16, o32


def add_sinusoidal_pe_(x: torch.Tensor):
    B, T, D = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("add_sinusoidal_pe_: need contiguous float32 [B,T,D]")
    check(lib.vmc_add_sinusoidal_pe(ptr(x), B, T, D, stream()), "add_sinusoidal_pe")
    return x


def transpose16(x: torch.Tensor) -> torch.Tensor:
    rows, cols = x.shape
    out = torch.empty((cols, rows), dtype=x.dtype, device=x.device)
    check(lib.vmc_transpose16(ptr(x), ptr(out), rows, cols, x.stride(0), rows, stream()), "transpose16")
    return out


def colsum(x: torch.Tensor) -> torch.Tensor:
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    nbytes = lib.vmc_colsum_workspace_bytes(M, N)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    check(lib.vmc_colsum(ptr(x), ptr(out), M, N, x.stride(0), dt(x), ptr(ws), nbytes, stream()), "colsum")
    return out


def act_fwd(x: torch.Tensor, act: int) -> torch.Tensor:
    y = torch.empty_like(x)
    check(lib.vmc_act_fwd(ptr(x), ptr(y), x.numel(), act, dt(x), stream()), "act_fwd")
    return y


def act_bwd(x: torch.Tensor, dy: torch.Tensor, act: int) -> torch.Tensor:
    dx = torch.empty_like(x)
    check(lib.vmc_act_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), act, dt(x), stream()), "act_bwd")
    return dx
