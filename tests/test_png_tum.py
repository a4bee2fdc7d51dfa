"""The PNG reader and association parser behind tools/run_tum.py (the harness for the TUM sequences BASELINE configs[0,1,4] name; no image
library ships with the image, so the decoder is in-tree): round trips through every scan-line filter, the formats the TUM sequences use
(8-bit RGB colour, 16-bit depth), and the association file of the reference."""
import os
import zlib

import numpy as np
import pytest

from rgbd_pl_slam_amd import png
from rgbd_pl_slam_amd.synth import synth_frame


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4, -1])
def test_png_round_trip_all_filters(tmp_path, filter_type):
    rng = np.random.default_rng(3 + filter_type)
    gray, d16 = synth_frame(11, 96, 64, with_depth=True)
    rgb = np.stack([gray, np.roll(gray, 3, 1), rng.integers(0, 256, gray.shape, dtype=np.uint8)], -1)
    rgba = np.concatenate([rgb, np.full(gray.shape + (1,), 255, np.uint8)], -1)
    ga16 = np.stack([d16, d16[::-1]], -1)
    for name, img in (("gray", gray), ("rgb", rgb), ("rgba", rgba), ("depth16", d16), ("ga16", ga16)):
        p = str(tmp_path / (name + ".png"))
        png.write_png(p, img, filter_type)
        back = png.read_png(p)
        assert back.dtype == img.dtype and back.shape == img.shape and np.array_equal(back, img), (name, filter_type)


def test_png_rejects_what_it_cannot_decode(tmp_path):
    p = str(tmp_path / "x.png")
    with open(p, "wb") as fh:
        fh.write(b"not a png at all")
    with pytest.raises(ValueError):
        png.read_png(p)
    # interlaced header
    import struct
    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    with open(p, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(b"\0" * 20)) + chunk(b"IEND", b""))
    with pytest.raises(ValueError):
        png.read_png(p)


def test_association_file_format(tmp_path):
    p = str(tmp_path / "assoc.txt")
    with open(p, "w") as fh:
        fh.write("1305031453.359684 rgb/1305031453.359684.png 1305031453.374112 depth/1305031453.374112.png\n\n"
                 "1305031453.391690 rgb/1305031453.391690.png 1305031453.404816 depth/1305031453.404816.png\n")
    a = png.read_associations(p)
    assert a == [(1305031453.359684, "rgb/1305031453.359684.png", "depth/1305031453.374112.png"),
                 (1305031453.391690, "rgb/1305031453.391690.png", "depth/1305031453.404816.png")]
    with open(p, "w") as fh:
        fh.write("1.0 rgb/a.png\n")
    with pytest.raises(ValueError):
        png.read_associations(p)


def test_reference_association_file_parses_if_present():
    """573 rows in the reference's own file (only in the build container: /root/reference does not exist on the GPU box)"""
    ref = "/root/reference/Examples/RGB-D/associations/fr1_desk.txt"
    if not os.path.exists(ref):
        pytest.skip("reference snapshot not present")
    a = png.read_associations(ref)
    assert len(a) == 573 and a[0][1].startswith("rgb/") and a[0][2].startswith("depth/") and all(a[i][0] < a[i + 1][0] for i in range(len(a) - 1))
