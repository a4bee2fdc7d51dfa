"""tools/run_tum.py end to end on a tiny synthetic "sequence" written as PNG files in the TUM layout (rgb/*.png 8-bit colour, depth/*.png 16-bit,
an association file): decode, the drop-in loop (one frame in flight), the batch driver, and bit-exact parity of every Frame member against the oracle."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import gpu_available, ROOT

pytestmark = pytest.mark.gpu


def test_run_tum_on_a_synthetic_sequence(tmp_path):
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")
    from rgbd_pl_slam_amd import png
    from rgbd_pl_slam_amd.synth import synth_frame
    seq = tmp_path / "seq"
    os.makedirs(seq / "rgb"); os.makedirs(seq / "depth")
    rng = np.random.default_rng(5)
    lines = []
    for i in range(5):
        gray, d16 = synth_frame(4000 + i, with_depth=True)
        rgb = np.stack([np.roll(gray, 2, 0), gray, rng.integers(0, 256, gray.shape, dtype=np.uint8)], -1)    # file order R, G, B
        t = 1305031453.0 + 0.033 * i
        png.write_png(str(seq / "rgb" / ("%.6f.png" % t)), rgb, filter_type=-1)
        png.write_png(str(seq / "depth" / ("%.6f.png" % (t + 0.01))), d16, filter_type=4)
        lines.append("%.6f rgb/%.6f.png %.6f depth/%.6f.png" % (t, t, t + 0.01, t + 0.01))
    assoc = seq / "assoc.txt"
    assoc.write_text("\n".join(lines) + "\n")
    spec = importlib.util.spec_from_file_location("run_tum", os.path.join(ROOT, "tools", "run_tum.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for camera_rgb in (1, 0):
        out = mod.run(str(seq), str(assoc), in_flight=2, parity_stride=1, camera="TUM1", camera_rgb=camera_rgb)
        assert out["frames"] == 5 and out["size"] == [640, 480]
        assert out["parity"]["frames_checked_against_oracle"] == 5 and out["parity"]["mismatches"] == [], out["parity"]
        assert out["single_frame_ms"]["median"] > 0 and out["batch"]["frames_per_s"] > 0


def test_run_tum_on_a_pan_over_real_photographs(tmp_path):
    """the same harness on REAL colour PNG content: a camera pan over the astronaut and the coffee-cup photographs (tests/golden/real; 6-pixel steps, VGA windows),
    written in the TUM layout with a synthetic 16-bit depth plane -- decode, colour -> gray with the yaml's Camera.RGB quirk, both extractors, Frame tail, every
    frame against the oracle"""
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")
    from rgbd_pl_slam_amd import png
    from rgbd_pl_slam_amd.synth import photo_pan_rgb
    seq = tmp_path / "seq"
    os.makedirs(seq / "rgb"); os.makedirs(seq / "depth")
    rng = np.random.default_rng(6)
    frames = np.concatenate([photo_pan_rgb("astronaut.png", 3), photo_pan_rgb("coffee.png", 3)])
    yy, xx = np.mgrid[0:480, 0:640]
    lines = []
    for i, rgb in enumerate(frames):
        d16 = np.clip(5000 * (1.2 + 0.002 * xx + 0.001 * yy) + rng.normal(0, 20, xx.shape), 0, 65535).astype(np.uint16)
        d16[rng.uniform(0, 1, d16.shape) < 0.04] = 0
        t = 1311868164.0 + 0.033 * i
        png.write_png(str(seq / "rgb" / ("%.6f.png" % t)), rgb, filter_type=-1)
        png.write_png(str(seq / "depth" / ("%.6f.png" % (t + 0.01))), d16, filter_type=4)
        lines.append("%.6f rgb/%.6f.png %.6f depth/%.6f.png" % (t, t, t + 0.01, t + 0.01))
    assoc = seq / "assoc.txt"
    assoc.write_text("\n".join(lines) + "\n")
    spec = importlib.util.spec_from_file_location("run_tum", os.path.join(ROOT, "tools", "run_tum.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = mod.run(str(seq), str(assoc), in_flight=3, parity_stride=1, camera="TUM3", camera_rgb=1)
    assert out["frames"] == 6 and out["size"] == [640, 480]
    assert out["parity"]["frames_checked_against_oracle"] == 6 and out["parity"]["mismatches"] == [], out["parity"]
