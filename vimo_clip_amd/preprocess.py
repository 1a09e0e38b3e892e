"""CLIP image preprocessing geometry on the GPU: ``Resize(n_px, BICUBIC)`` + ``CenterCrop(n_px)`` with Pillow's exact
arithmetic — the CPU PIL loop of models/student_model.py:77-78 and extract_embeddings.py:89-91.

Pillow's resize is a separable two-pass filter whose per-output-pixel coefficients are computed in float64, normalised
and rounded to 22-bit fixed point; the pixel arithmetic is integer.  The tables are built here with vectorised numpy in
float64 using Pillow's formulas (``precompute_coeffs`` / ``normalize_coeffs_8bpc`` of src/libImaging/Resample.c) and
cached on the device; the integer passes run in vmc_resample_u8, so results are bit-identical to PIL (tests compare
against PIL itself on the GPU box).  The centre crop is fused: only the kept columns / rows are computed.
"""
from __future__ import annotations

import functools

import numpy as np
import torch

from ._lib import check, lib, ptr, stream

_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=64)
def _tables_np(in_size: int, out_size: int):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((centers - support + 0.5).astype(np.int64), 0)          # C (int) cast truncates; values are >= -0.x here
    xmin = np.where(centers - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    t = np.arange(ksize, dtype=np.float64)[None, :]
    w = _bicubic((t + xmin[:, None] - centers[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(t < n[:, None], w, 0.0)
    # Pillow sums left to right in double; np.cumsum reproduces that order
    ww = np.cumsum(w, axis=1)[:, -1:]
    w = np.where(ww != 0.0, w / ww, w)
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << _PRECISION_BITS)), np.trunc(0.5 + w * (1 << _PRECISION_BITS))).astype(np.int32)
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return bounds, fixed, ksize


_device_tables = {}


def _tables(in_size: int, out_size: int, device):
    key = (in_size, out_size, str(device))
    if key not in _device_tables:
        b, k, ksize = _tables_np(in_size, out_size)
        _device_tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ksize)
    return _device_tables[key]


def shortest_edge_size(h: int, w: int, n_px: int):
    """torchvision ``Resize(int)`` and HF ``get_resize_output_image_size``: long side = int(n_px * long / short)."""
    return (int(n_px * h / w), n_px) if w <= h else (n_px, int(n_px * w / h))


def center_crop_offsets(h: int, w: int, n_px: int, mode: str = "torchvision"):
    if mode == "torchvision":
        return int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))
    if mode == "hf":
        return (h - n_px) // 2, (w - n_px) // 2
    raise ValueError("crop mode must be 'torchvision' or 'hf'")


def resize_center_crop_u8(frames: torch.Tensor, n_px: int, crop_mode: str = "torchvision", wrap_quirk: bool = False) -> torch.Tensor:
    """[F,3,H,W] u8 -> [F,3,n_px,n_px] u8 = CenterCrop(n_px)(Resize(n_px, BICUBIC)(frame)), PIL-exact.
    ``wrap_quirk`` first maps every input pixel v -> (256 - v) mod 256."""
    if frames.dtype != torch.uint8 or frames.dim() != 4:
        raise ValueError("frames must be u8 [F,C,H,W]")
    F, C, H, W = frames.shape
    nh, nw = shortest_edge_size(H, W, n_px)
    if nh < n_px or nw < n_px:
        raise ValueError("resized image smaller than the crop")
    top, left = center_crop_offsets(nh, nw, n_px, crop_mode)
    x = frames.contiguous()
    planes = F * C
    dev = x.device
    wrapped = False
    if nw != W or left != 0 or nw != n_px:
        if nw != W:
            b, k, ksize = _tables(W, nw, dev)
            y = torch.empty((F, C, H, n_px), dtype=torch.uint8, device=dev)
            check(lib.vmc_resample_u8(ptr(x), ptr(y), ptr(b), ptr(k), planes, H, W, left, n_px, ksize, 1, int(wrap_quirk), stream()), "resample_u8")
            x, wrapped = y, True
        else:
            x = x[..., left:left + n_px].contiguous()      # width already right: the crop is a strided copy (plumbing)
    W2 = x.shape[-1]
    if nh != H:
        b, k, ksize = _tables(H, nh, dev)
        y = torch.empty((F, C, n_px, W2), dtype=torch.uint8, device=dev)
        check(lib.vmc_resample_u8(ptr(x), ptr(y), ptr(b), ptr(k), planes, H, W2, top, n_px, ksize, 0, int(wrap_quirk and not wrapped), stream()),
              "resample_u8")
        x, wrapped = y, True
    elif top != 0 or H != n_px:
        x = x[..., top:top + n_px, :].contiguous()
    return x, (wrap_quirk and not wrapped)
