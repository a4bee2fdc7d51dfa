"""Properties of the CPU oracle that no golden vector covers (run without a GPU)."""
import ctypes as C

import numpy as np

import orc


def test_sincosf_restatement_equals_host_libm():
    """The BRIEF steering angle goes through glibc's sincosf in the reference (so@0x77803); the HIP kernel
    evaluates orc_sincosf_glibc's operation sequence.  It must be bit-identical to the libm the oracle calls."""
    L = orc.lib()
    L.orc_sincosf_selftest.restype = C.c_long
    assert L.orc_sincosf_selftest(C.c_long(4_000_003)) == 0


def test_fast_atan2_quadrants():
    L = orc.lib()
    for y, x in ((0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (1.0, 1.0), (-3.0, 2.0), (0.0, 0.0)):
        a = L.orc_fast_atan2(C.c_float(y), C.c_float(x))
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        assert abs(a - ref) < 0.02 or (y == 0 and x == 0)
        assert 0.0 <= a < 360.0 or a == 360.0


def test_orb_extract_basic_invariants():
    from rgbd_pl_slam_amd.synth import synth_frame
    img = synth_frame(0)
    r = orc.orb_extract(img)
    k = r["kps"]
    assert 900 <= len(k) <= 1000 + 3 * 8
    assert np.all(np.diff(k["octave"]) >= 0)                     # level-major output order
    assert np.all((k["angle"] >= 0) & (k["angle"] < 360))
    assert set(np.unique(k["size"]).tolist()) <= {31.0, 37.0, 44.0, 53.0, 64.0, 77.0, 92.0, 111.0}
    # keypoints keep EDGE_THRESHOLD distance from the border of their level
    sc = orc.orb_tables(1000, 1.2, 8)["scale"]
    for l in range(8):
        m = k["octave"] == l
        assert np.all(k["x"][m] >= 19 * sc[l] - 1e-3) and np.all(k["y"][m] >= 19 * sc[l] - 1e-3)
    assert r["ncells"] == [280, 192, 130, 88, 54, 35, 24, 12]
    # empty / flat image: no keypoints
    assert len(orc.orb_extract(np.full((480, 640), 77, np.uint8))["kps"]) == 0
