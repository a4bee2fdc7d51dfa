"""ctypes loader of libplf_hip.so (the C-ABI of include/plf.h).  There is NO CPU fallback: if the
HIP library is missing or cannot be loaded this module raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLF_LIB_PATH") or os.path.join(_HERE, "libplf_hip.so")   # (PLF_LIB_PATH: A/B measurements of scratch builds)

PLF_OK, PLF_E_EMPTY, PLF_E_BADARG, PLF_E_CAPACITY, PLF_E_HIP, PLF_E_NOMEM, PLF_E_RECTS = 0, -1, -2, -3, -4, -5, -6
PLF_W_TRUNCATED, PLF_W_SLOW = 1, 2
last_warning = 0   # the last positive (warning) status a call returned through check(): PLF_W_SLOW
MEM_HOST, MEM_DEVICE = 0, 1

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                     ("response", "<f4"), ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"),
                     ("endPointX", "<f4"), ("endPointY", "<f4"), ("sPointInOctaveX", "<f4"), ("sPointInOctaveY", "<f4"),
                     ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"), ("lineLength", "<f4"), ("numOfPixels", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("ini_th_fast", C.c_int32),
                ("min_th_fast", C.c_int32), ("device", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32),
                ("max_batch", C.c_int32)]


class LineParams(C.Structure):
    _fields_ = [("nlines", C.c_int32), ("seed_order", C.c_int32), ("device", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("lbd_sobel_input", C.c_int32), ("max_ms", C.c_float)]


LBD_BLURRED, LBD_RAW = 0, 1


def line_params(nlines, seed_order, device, max_width, max_height, max_batch, lbd_sobel_input=LBD_BLURRED, max_ms=0.0):
    return LineParams(nlines, seed_order, device, max_width, max_height, max_batch, lbd_sobel_input, max_ms)


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_device", C.c_void_p), ("keys_un", C.c_void_p), ("uright", C.c_void_p), ("desc", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("scale_factors", C.c_void_p), ("nlevels", C.c_int32)]


class MapPointView(C.Structure):
    _fields_ = [("m", C.c_int32), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p),
                ("level", C.c_void_p), ("view_cos", C.c_void_p), ("in_view", C.c_void_p), ("desc", C.c_void_p),
                ("obs_positive", C.c_void_p)]


class LastFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("has_mappoint", C.c_void_p), ("outlier", C.c_void_p), ("world_pos", C.c_void_p),
                ("keys", C.c_void_p), ("mp_desc", C.c_void_p), ("obs_positive", C.c_void_p)]


class PosePair(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Rlw", C.c_float * 9), ("tlw", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("b", C.c_float)]


class LineFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_device", C.c_void_p), ("lines_un", C.c_void_p), ("desc", C.c_void_p), ("scale_factors", C.c_void_p)]


class MapLineView(C.Structure):
    _fields_ = [("m", C.c_int32), ("x1", C.c_void_p), ("y1", C.c_void_p), ("x2", C.c_void_p), ("y2", C.c_void_p),
                ("level", C.c_void_p), ("view_cos", C.c_void_p), ("in_view", C.c_void_p), ("desc", C.c_void_p)]


class BowView(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_f", C.c_int32), ("kf_desc", C.c_void_p), ("f_desc", C.c_void_p), ("kf_angle", C.c_void_p),
                ("f_angle", C.c_void_p), ("kf_has_mp", C.c_void_p), ("f_has_mp", C.c_void_p), ("kf_nodes", C.c_int32), ("f_nodes", C.c_int32),
                ("kf_node_id", C.c_void_p), ("f_node_id", C.c_void_p), ("kf_node_start", C.c_void_p), ("f_node_start", C.c_void_p),
                ("kf_feat", C.c_void_p), ("f_feat", C.c_void_p)]


class Points3DView(C.Structure):
    _fields_ = [("m", C.c_int32), ("world_pos", C.c_void_p), ("normal", C.c_void_p), ("min_distance", C.c_void_p), ("max_distance", C.c_void_p),
                ("desc", C.c_void_p), ("valid", C.c_void_p)]


class KfPose(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float), ("log_scale_factor", C.c_float), ("inv_level_sigma2", C.c_void_p)]


class TriView(C.Structure):
    _fields_ = [("n1", C.c_int32), ("n2", C.c_int32), ("keys1", C.c_void_p), ("keys2", C.c_void_p), ("uright1", C.c_void_p), ("uright2", C.c_void_p),
                ("desc1", C.c_void_p), ("desc2", C.c_void_p), ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p), ("nodes1", C.c_int32), ("nodes2", C.c_int32),
                ("node_id1", C.c_void_p), ("node_id2", C.c_void_p), ("node_start1", C.c_void_p), ("node_start2", C.c_void_p), ("feat1", C.c_void_p),
                ("feat2", C.c_void_p), ("scale_factors2", C.c_void_p), ("level_sigma2_2", C.c_void_p)]


class PlfError(RuntimeError):
    def __init__(self, status, what):
        super().__init__("%s failed: %s (%d)" % (what, lib().plf_status_string(status).decode(), status))
        self.status = status


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libplf_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback" % LIB_PATH)
        # torch (the Python mirror's device-memory plumbing) ships its own copy of the HIP runtime: it has to be in the process BEFORE this library's
        # dependency on libamdhip64 is resolved, otherwise a later `import torch` brings a second runtime and neither sees the GPU any more
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.plf_version.restype = C.c_char_p
        _lib.plf_status_string.restype = C.c_char_p
        _lib.plf_status_string.argtypes = [C.c_int]
    return _lib


def check(status, what):
    """errors (< 0) raise; warnings (> 0: outputs complete) are kept in `last_warning`"""
    global last_warning
    last_warning = status if status > 0 else 0
    if status < 0:
        raise PlfError(status, what)


def vp(a):
    """pointer of a numpy array, a torch tensor (device or host) or an int address"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)
