"""The C-ABI library loads, exports every symbol include/plf.h declares, refuses to run without a GPU
(no CPU fallback), and the C++ host mirror (include/plf.hpp) compiles against it.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rgbd_pl_slam_amd", "libplf_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    return C.CDLL(LIB)


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "plf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(plf_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_sizes_match_opencv_layouts():
    from rgbd_pl_slam_amd import _lib as L
    assert L.KP_DTYPE.itemsize == 28      # cv::KeyPoint
    assert L.KL_DTYPE.itemsize == 68      # cv::line_descriptor::KeyLine
    assert L.DMATCH_DTYPE.itemsize == 16  # cv::DMatch


def test_hamming_host_utility(lib):
    import numpy as np
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
    assert lib.plf_hamming256(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == int(np.unpackbits(a ^ b).sum())


def test_no_cpu_fallback_without_gpu(lib):
    """Without a HIP device every create() fails loudly with PLF_E_HIP; nothing routes to a CPU path."""
    from conftest import gpu_available
    if gpu_available():
        pytest.skip("a GPU is visible")
    from rgbd_pl_slam_amd import _lib as L
    h = C.c_void_p()
    p = L.OrbParams(1000, 1.2, 8, 20, 7, 0, 640, 480, 1)
    assert lib.plf_orb_create(C.byref(p), C.byref(h)) == L.PLF_E_HIP and not h.value
    lp = L.line_params(100, 0, 0, 640, 480, 1)
    assert lib.plf_line_create(C.byref(lp), C.byref(h)) == L.PLF_E_HIP and not h.value
    assert lib.plf_matcher_create(0, 1000, 5000, 100, 1, C.byref(h)) == L.PLF_E_HIP and not h.value
    assert lib.plf_device_count() == 0


def test_bad_arguments_are_rejected_before_touching_the_device(lib):
    from rgbd_pl_slam_amd import _lib as L
    h = C.c_void_p()
    bad = L.OrbParams(0, 1.2, 8, 20, 7, 0, 640, 480, 1)          # nfeatures < 1
    assert lib.plf_orb_create(C.byref(bad), C.byref(h)) == L.PLF_E_BADARG
    bad = L.OrbParams(1000, 1.0, 8, 20, 7, 0, 640, 480, 1)       # scale factor must exceed 1
    assert lib.plf_orb_create(C.byref(bad), C.byref(h)) == L.PLF_E_BADARG
    bad = L.OrbParams(1000, 1.2, 13, 20, 7, 0, 640, 480, 1)      # too many levels
    assert lib.plf_orb_create(C.byref(bad), C.byref(h)) == L.PLF_E_BADARG
    assert lib.plf_orb_create(None, C.byref(h)) == L.PLF_E_BADARG
    assert lib.plf_line_create(None, C.byref(h)) == L.PLF_E_BADARG


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    src = tmp_path / "mirror.cpp"
    src.write_text('''
#include "plf.hpp"
#include <cstdio>
int main() {
    // no GPU in the build container: construction must throw plf::Error(PLF_E_HIP), never fall back
    try { plf::ORBextractor e(1000, 1.2f, 8, 20, 7); std::printf("created\\n"); }
    catch (const plf::Error &e) { std::printf("error %d\\n", e.status); }
    try { plf::LineSegment l(100); std::printf("created\\n"); }
    catch (const plf::Error &e) { std::printf("error %d\\n", e.status); }
    uint8_t a[32] = {0}, b[32] = {0}; b[3] = 0x81;
    std::printf("hamming %d\\n", plf::ORBmatcher::DescriptorDistance(a, b));
    // the Frame mirror: a null pointer must be rejected by argument checking, before any device work
    plf_camera cam = {517.3f, 516.5f, 318.6f, 255.3f, 0.f, 0.f, 0.f, 0.f, 0.f, 40.f};
    try { plf::Frame::UndistortKeyPoints(nullptr, 10, cam, nullptr); std::printf("accepted\\n"); }
    catch (const plf::Error &e) { std::printf("frame error %d\\n", e.status); }
    try { plf::Frame::UndistortKeyLines(nullptr, 10, nullptr, 0, 0, cam, nullptr, nullptr, nullptr, nullptr, nullptr); std::printf("accepted\\n"); }
    catch (const plf::Error &e) { std::printf("frame error %d\\n", e.status); }
    return 0;
}
''')
    exe = tmp_path / "mirror"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), LIB,
                           "-Wl,-rpath," + os.path.dirname(LIB), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True, stderr=subprocess.DEVNULL)
    assert "hamming 2" in out and out.count("frame error -2") == 2
    from conftest import gpu_available
    if not gpu_available():
        assert out.count("error -4") == 2


def test_exact_signature_adapters_compile_against_the_mock_headers(tmp_path):
    """include/plf.hpp under PLF_WITH_OPENCV: ORBextractor::operator()(InputArray, ...), LineSegment::ExtractLineSegment(Mat, ...),
    ORBmatcher::SearchByProjection x2, LSDmatcher::SearchByProjection x3 / SearchForTriangulation / Fuse with the reference's signatures, instantiated
    over mock Frame / KeyFrame / MapPoint / MapLine classes that carry the reference's member names (tests/mock/) -- compiled and linked here (no GPU:
    the driver is only built, tests/test_gpu_cpp_mirror.py runs it)."""
    exe = tmp_path / "mirror_driver"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror=return-type", "-DPLF_WITH_OPENCV", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "mock"), os.path.join(ROOT, "tests", "cpp", "mirror_driver.cpp"), "-o", str(exe), LIB,
                           "-Wl,-rpath," + os.path.dirname(LIB), "-Wl,-rpath,/opt/rocm/lib"])
    assert exe.exists()
