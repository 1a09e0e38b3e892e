import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    # libvmc.so is git-ignored; build it (hipcc cross-compiles gfx950 without a GPU) when missing.
    if not os.path.exists(os.path.join(ROOT, "vimo_clip_amd", "libvmc.so")):
        import __graft_entry__
        __graft_entry__.build()


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = os.path.join(ROOT, "tests", "golden")
    return {n: np.load(os.path.join(d, n + ".npz")) for n in ("losses", "tfam", "indexing", "vit", "metrics", "student")}
