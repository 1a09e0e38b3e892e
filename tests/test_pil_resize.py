"""CPU: the numpy restatement of Pillow's antialiased bicubic resize (oracle/pil_resize.py) is bit-exact against PIL
itself, and the shortest-edge / centre-crop geometry matches the two callers of the reference (torchvision-style and
HF-style rounding)."""
import numpy as np
import pytest
from PIL import Image

from oracle import pil_resize as pr
from vimo_clip_amd import synth


@pytest.mark.parametrize("H,W,n", [(360, 640, 224), (240, 320, 224), (300, 300, 224), (224, 400, 224), (97, 181, 64), (500, 375, 224), (112, 112, 224)])
def test_resize_matches_pil_bit_exact(H, W, n):
    img = synth.randint_u8(7, f"img{H}x{W}", (3, H, W)).numpy()
    nh, nw = pr.shortest_edge_size(H, W, n)
    ref = np.asarray(Image.fromarray(np.transpose(img, (1, 2, 0))).resize((nw, nh), Image.BICUBIC))
    got = np.transpose(pr.resize_bicubic(img, nh, nw), (1, 2, 0))
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_geometry():
    assert pr.shortest_edge_size(360, 640, 224) == (224, 398)
    assert pr.shortest_edge_size(640, 360, 224) == (398, 224)
    assert pr.center_crop_offsets(224, 398, 224) == (0, 87)
    assert pr.center_crop_offsets(224, 399, 224, "torchvision") == (0, 88) and pr.center_crop_offsets(224, 399, 224, "hf") == (0, 87)
