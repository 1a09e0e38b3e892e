"""Oracle: Pillow's antialiased bicubic resize for 8-bit images + the shortest-edge / centre-crop geometry of the CLIP
preprocessing (test infrastructure only).

Follows Pillow's ``ImagingResample`` (src/libImaging/Resample.c, the library behind ``PIL.Image.resize`` that
torchvision ``Resize(n_px, BICUBIC)`` — clip._transform, used at models/student_model.py:77-78 — and HF
``CLIPImageProcessor`` — extract_embeddings.py:91 — both call): separable two-pass filter, horizontal pass first,
float64 coefficients normalised per output pixel then rounded to 22-bit fixed point, accumulation in integers with a
rounding bias of 1 << 21, result clipped to [0, 255] after each pass.  Pinned against PIL itself (tests/test_pil_resize.py).
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int, support: float = 2.0):
    """-> (bounds [out,2] int32 (xmin, count), coeffs [out, ksize] int32 fixed point)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """img u8 [..., H, W]; resample along axis (-1 horizontal, -2 vertical)."""
    img = np.moveaxis(img, axis, -1)
    bounds, kk = precompute_coeffs(img.shape[-1], out_size)
    out = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = (src[..., xmin:xmin + n] * kk[xx, :n].astype(np.int64)).sum(-1) + (1 << (PRECISION_BITS - 1))
        out[..., xx] = _clip8(acc)
    return np.moveaxis(out, -1, axis)


def resize_bicubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) on planar u8 [..., H, W]: horizontal pass, then vertical."""
    H, W = img.shape[-2:]
    if out_w != W:
        img = resample_axis(img, out_w, -1)
    if out_h != H:
        img = resample_axis(img, out_h, -2)
    return img


def shortest_edge_size(h: int, w: int, n_px: int):
    """torchvision Resize(int) / HF get_resize_output_image_size(shortest_edge): new_long = int(n_px * long / short)."""
    if w <= h:
        return int(n_px * h / w), n_px
    return n_px, int(n_px * w / h)


def center_crop_offsets(h: int, w: int, n_px: int, mode: str = "torchvision"):
    if mode == "torchvision":      # transforms.functional.center_crop: int(round((size - crop) / 2.0))
        return int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))
    return (h - n_px) // 2, (w - n_px) // 2      # HF image_transforms.center_crop


def clip_resize_crop(frames_u8: np.ndarray, n_px: int, mode: str = "torchvision") -> np.ndarray:
    """[F,3,H,W] u8 -> [F,3,n_px,n_px] u8: Resize(n_px, BICUBIC) + CenterCrop(n_px)."""
    H, W = frames_u8.shape[-2:]
    nh, nw = shortest_edge_size(H, W, n_px)
    r = resize_bicubic(frames_u8, nh, nw)
    top, left = center_crop_offsets(nh, nw, n_px, mode)
    return r[..., top:top + n_px, left:left + n_px]
