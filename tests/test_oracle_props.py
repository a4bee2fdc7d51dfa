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


def test_lsd_seed_without_aligned_neighbour_grows_alone(tmp_path):
    """The argument behind the GPU's static singles (lsd_kernels.hip, singles_run): a seed none of whose 8 neighbours has a level-line angle within
    the tolerance of its own grows a ONE-pixel region, whatever is marked around it -- counted inside the oracle's own region loop
    (oracle/lsd_oracle.c under ORC_LSD_STATS, built here the way tools/singleton_stats.c says)."""
    import os
    import subprocess
    from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libsingle.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-fopenmp", "-ffp-contract=off", "-w", "-DORC_LSD_STATS", "-I", os.path.join(root, "oracle"),
                           os.path.join(root, "tools", "singleton_stats.c"), os.path.join(root, "oracle", "orb_oracle.c"), os.path.join(root, "oracle", "lbd_oracle.c"),
                           os.path.join(root, "oracle", "timing.c"), "-o", so, "-lm"])
    L = C.CDLL(so)
    L.orc_stats.restype = C.POINTER(C.c_long)
    st = L.orc_stats()
    rng = np.random.default_rng(3)
    imgs = [synth_frame(11, 320, 240), natural_frame(12, 320, 240), rng.integers(0, 256, (120, 160)).astype(np.uint8)]
    for g in imgs:
        for i in range(16):
            st[i] = 0
        g = np.ascontiguousarray(g, np.uint8)
        h, w = g.shape
        lines = np.zeros((1 << 14, 4), np.float32)
        L.orc_lsd_detect(g.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h), C.c_ssize_t(w), C.c_int(0), lines.ctypes.data_as(C.c_void_p), C.c_int(1 << 14), None)
        assert st[0] > 0 and st[2] > 0        # regions were grown, some of them foreseen singles
        assert st[2] <= st[1] <= st[0]
        assert st[6] == 0                      # no foreseen single ever grew beyond itself
