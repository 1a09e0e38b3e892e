"""ctypes binding of libvmc.so (include/vmc.h).  There is NO fallback: if the library is missing the
import fails loudly, and every compute entry point raises on a non-zero return code."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvmc.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF, ACT_RELU = 0, 1, 2, 3

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

P, I, F, Z = c_void_p, c_int, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/vmc.h one to one (tests/test_abi.py checks both against
# the header and the built library).
SIGNATURES = {
    "vmc_abi_version": (I, []),
    "vmc_error_string": (c_char_p, [I]),
    "vmc_clock_probe": (I, [P, I, I, I, P]),
    "vmc_preprocess_patches_u8": (I, [P, P, I, I, I, I, I, I, P]),
    "vmc_patches_f32": (I, [P, P, I, I, I, I, I, P]),
    "vmc_patches_u8_exact": (I, [P, P, I, I, I, I, I, I, P]),
    "vmc_patches_f32_split": (I, [P, P, I, I, I, I, I, P]),
    "vmc_resample_u8": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "vmc_linear": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, I, I, I, I, P]),
    "vmc_linear_preact": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, I, I, I, P]),
    "vmc_linear_splitk_workspace_bytes": (Z, [I, I, I]),
    "vmc_linear_splitk_f32": (I, [P, P, P, I, I, I, I, I, P, Z, I, P]),
    "vmc_linear_wgrad_tn_workspace_bytes": (Z, [I, I, I]),
    "vmc_linear_wgrad_tn": (I, [P, P, P, I, I, I, I, I, P, Z, I, P]),
    "vmc_linear_wgrad_bias_tn": (I, [P, P, P, P, I, I, I, I, I, P, Z, I, P]),
    "vmc_linear_wgrad_tn_group": (I, [P, I, I, P]),
    "vmc_linear_variant": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, I, I, I, I, I, P]),
    "vmc_transpose16": (I, [P, P, I, I, I, I, P]),
    "vmc_cast_weight": (I, [P, P, P, I, I, I, I, I, P]),
    "vmc_cast_weights_multi": (I, [P, I, I, I, P]),
    "vmc_colsum_workspace_bytes": (Z, [I, I]),
    "vmc_colsum": (I, [P, P, I, I, I, I, P, Z, P]),
    "vmc_layernorm_fwd": (I, [P, P, P, P, P, P, P, I, I, I, F, I, I, P]),
    "vmc_add_layernorm_fwd": (I, [P, P, P, P, P, I, I, I, I, F, I, I, P]),
    "vmc_add2_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, F, I, I, P]),
    "vmc_postnorm_fwd": (I, [P, P, P, P, P, P, P, P, P, I, I, F, I, P]),
    "vmc_postnorm_dropout_fwd": (I, [P, P, P, P, P, P, P, P, P, I, I, F, F, ctypes.c_uint64, F, ctypes.c_uint64, I, P]),
    "vmc_layernorm_bwd_workspace_bytes": (Z, [I, I]),
    "vmc_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "vmc_layernorm_bwd2": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "vmc_postnorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, F, ctypes.c_uint64, F, ctypes.c_uint64, I, P, Z, P]),
    "vmc_scale_by_device_scalar": (I, [P, P, Z, P, P]),
    "vmc_add": (I, [P, P, P, Z, I, I, I, I, P]),
    "vmc_mean_pool_bwd": (I, [P, P, I, I, I, I, I, I, P]),
    "vmc_assemble_tokens": (I, [P, P, P, P, I, I, I, I, I, P]),
    "vmc_dropout": (I, [P, P, Z, F, ctypes.c_uint64, I, I, P]),
    "vmc_cast_dropout2": (I, [P, P, ctypes.c_size_t, F, ctypes.c_uint64, F, ctypes.c_uint64, I, P]),
    "vmc_attention_vit_fwd": (I, [P, P, P, I, I, I, I, P]),
    "vmc_attention_vit_cls_fwd": (I, [P, P, P, I, I, I, I, P]),
    "vmc_attention_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, ctypes.c_uint64, I, P]),
    "vmc_attention_bwd_workspace_bytes": (Z, [I, I, I]),
    "vmc_attention_bwd": (I, [P] * 10 + [I] * 12 + [F, ctypes.c_uint64, P, Z, I, P]),
    "vmc_set_class_rows": (I, [P, P, P, I, I, Z, I, I, P]),
    "vmc_act_fwd": (I, [P, P, Z, I, I, P]),
    "vmc_act_bwd": (I, [P, P, P, Z, I, I, P]),
    "vmc_mean_pool": (I, [P, P, P, I, I, I, I, I, P]),
    "vmc_add_sinusoidal_pe": (I, [P, I, I, I, P]),
    "vmc_axpby_f32": (I, [P, P, P, Z, F, F, P]),
    "vmc_cast_f32_to_16": (I, [P, P, Z, I, P]),
    "vmc_cast_16_to_f32": (I, [P, P, Z, I, P]),
    "vmc_loss_workspace_bytes": (Z, [I]),
    "vmc_distill_loss": (I, [P, P, P, P, I, I, I, Z, I, P, Z, P]),
    "vmc_bce_loss": (I, [P, P, P, P, I, F, P, Z, P]),
    "vmc_cross_entropy_loss": (I, [P, P, P, P, P, I, I, P, Z, P]),
    "vmc_tfam_pack_offset": (ctypes.c_longlong, [I, I, I, I, I, I]),
    "vmc_tfam_fold_layernorm": (I, [P, P, P, P, P, P, I, I, I, P]),
    "vmc_tfam_workspace_bytes": (Z, [I, I, I, I, I, I, I, I]),
    "vmc_tfam_kv_fwd": (I, [P, P, P, P, Z, I, I, I, I, I, I, I, I, I, P]),
    "vmc_tfam_layer_fwd": (I, [P, P, P, P, P, I, P, Z, I, I, I, I, I, I, I, I, I, I, P]),
    "vmc_tfam_head_fwd": (I, [P, P, P, P, Z, I, I, I, I, I, I, I, I, I, I, P]),
    "vmc_tfam_forward": (I, [P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, I, I, I, P]),
    "vmc_tfam_train_workspace_bytes": (Z, [I] * 9),
    "vmc_tfam_layer_train_fwd": (I, [P, P, P, P, P, I, P, Z] + [I] * 9 + [F, P, I, P]),
    "vmc_tfam_head_train_fwd": (I, [P, P, P, P, Z] + [I] * 9 + [F, ctypes.c_uint64, I, P]),
    "vmc_tfam_head_bwd": (I, [P, P, P, P, Z] + [I] * 9 + [F, ctypes.c_uint64, I, P]),
    "vmc_tfam_layer_bwd": (I, [P, P, P, I, P, Z] + [I] * 9 + [F, P, I, P]),
    "vmc_tfam_train_fwd": (I, [P, P, P, P, P, P, P, P, Z] + [I] * 9 + [F, F, P, I, P]),
    "vmc_tfam_train_bwd": (I, [P, P, P, P, P, P, Z] + [I] * 9 + [F, F, P, I, P]),
    "vmc_adam_step": (I, [P, P, P, P, Z, F, F, F, F, F, I, I, F, P]),
    "vmc_train_tick": (I, [P, P, F, F, I, P]),
    "vmc_adam_step_dev": (I, [P, P, P, P, Z, P, F, F, F, F, I, P]),
    "vmc_adam_step_dev_bg": (I, [P, P, P, P, Z, P, F, F, F, F, I, I, P]),
    "vmc_adam_cast_multi": (I, [P, I, I, P, I, I, P, P, P, P, P, F, F, F, F, I, I, P]),
    "vmc_sumsq": (I, [P, Z, P, P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C vimo_clip_amd/csrc`). There is no CPU/PyTorch fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.vmc_abi_version() != 1:
        raise ImportError("libvmc.so ABI version mismatch; rebuild it")
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.vmc_error_string(int(rc))
        raise RuntimeError(f"libvmc {what} failed with code {rc}: {msg.decode() if msg else '?'}")


def dt(t) -> int:
    try:
        return _DT[t if isinstance(t, torch.dtype) else t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype for libvmc: {t}") from None


def ptr(t) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libvmc kernels need device (HIP) tensors; there is no CPU path")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """hipStream_t of torch's current stream on the current device.  The raw C getters cost ~1 us per call;
    ``torch.cuda.current_stream().cuda_stream`` builds a Stream object (~9 us), which was 14 % of a launch-bound step."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream
