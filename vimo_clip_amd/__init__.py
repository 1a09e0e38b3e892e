"""vimo_clip_amd — MI355X-native (gfx950) engine for the ViMoCLIP hot path.

Python mirror of the reference's module/function interface over the libvmc C ABI (include/vmc.h).
Importing the package loads libvmc.so; there is no CPU or PyTorch-op fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is not built)

__all__ = ["_lib"]
