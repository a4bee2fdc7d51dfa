"""Host-side mirror of the multi-GPU batch driver (include/plf.h: plf_batch_*): the reference's caller loop
Examples/RGB-D/rgbd_tum.cc:84-128 for a batch of independent host frames, sharded by contiguous blocks over the GPUs
of the node.  Everything -- sharding, pinned staging, async copies, per-GPU worker threads -- lives in
libplf_hip.so; this file only marshals numpy arrays."""
import ctypes as C

import numpy as np

from . import _lib as L
from .frame import Camera

FMT_GRAY8, FMT_RGB8, FMT_BGR8 = 0, 1, 2


class BatchParams(C.Structure):
    _fields_ = [("orb", L.OrbParams), ("line", L.LineParams), ("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32)),
                ("frames_in_flight", C.c_int32), ("input_format", C.c_int32), ("max_mappoints", C.c_int32), ("max_maplines", C.c_int32), ("rgbd", C.c_int32)]


class BatchOutputs(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("n_kps", C.c_void_p), ("kp_capacity", C.c_int32),
                ("lines", C.c_void_p), ("ldesc", C.c_void_p), ("line_eq", C.c_void_p), ("n_lines", C.c_void_p), ("line_capacity", C.c_int32),
                ("match_of_kp", C.c_void_p), ("n_kp_matches", C.c_void_p), ("match_of_line", C.c_void_p), ("n_line_matches", C.c_void_p)]


class BatchRgbd(C.Structure):
    _fields_ = [("cam", Camera), ("depth_factor", C.c_float), ("depth", C.c_void_p), ("depth_pitch_elems", C.c_ssize_t),
                ("depth_frame_stride_elems", C.c_ssize_t), ("kps_un", C.c_void_p), ("uright", C.c_void_p), ("kp_depth", C.c_void_p),
                ("lines_un", C.c_void_p), ("uright_start", C.c_void_p), ("uright_end", C.c_void_p), ("depth_start", C.c_void_p),
                ("depth_end", C.c_void_p)]


def shard(n_frames, parts, part):
    """contiguous block of frames [lo, hi) owned by `part` of `parts` -- plf_batch_shard, the partition the driver itself uses
    for its GPUs and bench.py uses for its ranks"""
    first, count = C.c_int64(0), C.c_int64(0)
    L.check(L.lib().plf_batch_shard(C.c_int64(n_frames), int(parts), int(part), C.byref(first), C.byref(count)), "plf_batch_shard")
    return first.value, first.value + count.value


_PINNED = {}


def pinned_array(shape, dtype=np.uint8):
    """numpy array backed by page-locked host memory (plf_host_alloc): frames decoded into it are uploaded without a staging copy"""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    L.check(L.lib().plf_host_alloc(C.c_size_t(n), C.byref(p)), "plf_host_alloc")
    buf = (C.c_uint8 * n).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    _PINNED[arr.ctypes.data] = p.value
    return arr


def free_pinned(arr):
    p = _PINNED.pop(arr.ctypes.data, None)
    if p:
        L.lib().plf_host_free(C.c_void_p(p))


class BatchExtractor:
    """One object = the per-GPU extractors + matchers of a node.  extract(images) returns per-frame features (and matches against
    the local map set with set_local_map)."""

    def __init__(self, nfeatures=1000, nlines=100, width=640, height=480, frames_in_flight=8, devices=None, scaleFactor=1.2, nlevels=8,
                 iniThFAST=20, minThFAST=7, input_format=FMT_GRAY8, max_mappoints=0, max_maplines=0, seed_order=0, lbd_sobel_input=L.LBD_BLURRED, rgbd=False, max_ms=0.0):
        p = BatchParams()
        p.orb = L.OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, 0, width, height, 1)
        p.line = L.line_params(nlines, seed_order, 0, width, height, 1, lbd_sobel_input, max_ms)
        self._devs = None
        if devices is not None:
            self._devs = (C.c_int32 * len(devices))(*devices)
            p.n_devices = len(devices); p.devices = self._devs
        p.frames_in_flight = frames_in_flight; p.input_format = input_format
        p.max_mappoints = max_mappoints; p.max_maplines = max_maplines; p.rgbd = int(bool(rgbd))
        self.rgbd = bool(rgbd)
        self._h = C.c_void_p()
        lib = L.lib()
        lib.plf_batch_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_ssize_t, C.c_ssize_t, C.c_void_p]
        lib.plf_batch_extract_rgbd.argtypes = lib.plf_batch_extract.argtypes + [C.c_void_p]
        L.check(lib.plf_batch_create(C.byref(p), C.byref(self._h)), "plf_batch_create")
        self.n_devices = lib.plf_batch_device_count(self._h)
        self.devices = [lib.plf_batch_device(self._h, i) for i in range(self.n_devices)]
        self.nfeatures, self.nlines, self.nlevels = nfeatures, nlines, nlevels
        self.kp_capacity = nfeatures + 4 * nlevels if nfeatures > 0 else 0
        self.bpp = 1 if input_format == FMT_GRAY8 else 3
        self._has_map = False

    def close(self):
        if self._h:
            L.lib().plf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_local_map(self, points=None, lines=None, th=3.0, nnratio=0.8, bounds=(0.0, 0.0, 640.0, 480.0)):
        """points / lines: dicts of HOST numpy arrays with the fields of plf_mappoint_view / plf_mapline_view"""
        keep = []
        pv = lv = None
        if points is not None:
            pv = L.MapPointView(); pv.m = int(points["desc"].shape[0])
            for k, dt in (("proj_x", np.float32), ("proj_y", np.float32), ("proj_xr", np.float32), ("level", np.int32), ("view_cos", np.float32),
                          ("in_view", np.uint8), ("desc", np.uint8)):
                a = np.ascontiguousarray(points[k], dt); keep.append(a); setattr(pv, k, a.ctypes.data)
            if points.get("obs_positive") is not None:
                a = np.ascontiguousarray(points["obs_positive"], np.uint8); keep.append(a); pv.obs_positive = a.ctypes.data
        if lines is not None:
            lv = L.MapLineView(); lv.m = int(lines["desc"].shape[0])
            for k, dt in (("x1", np.float32), ("y1", np.float32), ("x2", np.float32), ("y2", np.float32), ("level", np.int32),
                          ("view_cos", np.float32), ("in_view", np.uint8), ("desc", np.uint8)):
                a = np.ascontiguousarray(lines[k], dt); keep.append(a); setattr(lv, k, a.ctypes.data)
        L.check(L.lib().plf_batch_set_local_map(self._h, C.byref(pv) if pv is not None else None, C.byref(lv) if lv is not None else None,
                                                C.c_float(th), C.c_float(nnratio), *[C.c_float(b) for b in bounds]), "plf_batch_set_local_map")
        self._has_map = points is not None or lines is not None

    def alloc_outputs(self, n):
        o = {}
        if self.nfeatures > 0:
            o["kps"] = np.zeros((n, self.kp_capacity), L.KP_DTYPE); o["desc"] = np.zeros((n, self.kp_capacity, 32), np.uint8)
            o["n_kps"] = np.zeros(n, np.int32)
            o["match_of_kp"] = np.full((n, self.kp_capacity), -1, np.int32); o["n_kp_matches"] = np.zeros(n, np.int32)
        if self.nlines > 0:
            o["lines"] = np.zeros((n, self.nlines), L.KL_DTYPE); o["ldesc"] = np.zeros((n, self.nlines, 32), np.uint8)
            o["line_eq"] = np.zeros((n, self.nlines, 3), np.float64); o["n_lines"] = np.zeros(n, np.int32)
            o["match_of_line"] = np.full((n, self.nlines), -1, np.int32); o["n_line_matches"] = np.zeros(n, np.int32)
        return o

    def alloc_rgbd_outputs(self, n):
        o = {}
        if self.nfeatures > 0:
            o["kps_un"] = np.zeros((n, self.kp_capacity), L.KP_DTYPE)
            o["uright"] = np.zeros((n, self.kp_capacity), np.float32); o["kp_depth"] = np.zeros((n, self.kp_capacity), np.float32)
        if self.nlines > 0:
            o["lines_un"] = np.zeros((n, self.nlines), L.KL_DTYPE)
            for k in ("uright_start", "uright_end", "depth_start", "depth_end"):
                o[k] = np.zeros((n, self.nlines), np.float32)
        return o

    def extract_into(self, images, out, depth=None, cam=None, depth_factor=1.0 / 5000.0, rgbd_out=None):
        """images: (n, H, W) uint8 (gray) or (n, H, W, 3); row / frame strides may be padded.  out: alloc_outputs(n).  Returns the status
        (0 or PLF_E_CAPACITY).  With cam (a frame.Camera) the RGB-D Frame tail runs too (plf_batch_extract_rgbd): depth = (n, H, W) uint16 or
        None, rgbd_out = alloc_rgbd_outputs(n)."""
        n, h, w = images.shape[:3]
        if images.dtype != np.uint8 or images.strides[2] != self.bpp or (self.bpp == 3 and (images.ndim != 4 or images.strides[3] != 1)):
            raise ValueError("uint8 frames with %d byte(s) per pixel expected" % self.bpp)
        O = BatchOutputs()
        for k in ("kps", "desc", "n_kps", "lines", "ldesc", "line_eq", "n_lines", "match_of_kp", "n_kp_matches", "match_of_line", "n_line_matches"):
            if k in out:
                setattr(O, k, out[k].ctypes.data)
        O.kp_capacity = self.kp_capacity; O.line_capacity = self.nlines
        if cam is None:
            st = L.lib().plf_batch_extract(self._h, images.ctypes.data, n, w, h, images.strides[1], images.strides[0], C.byref(O))
        else:
            R = BatchRgbd()
            R.cam = cam; R.depth_factor = depth_factor
            if depth is not None:
                if depth.dtype != np.uint16 or depth.shape[:3] != (n, h, w) or depth.strides[2] != 2:
                    raise ValueError("depth: (n, H, W) uint16 expected")
                R.depth = depth.ctypes.data; R.depth_pitch_elems = depth.strides[1] // 2; R.depth_frame_stride_elems = depth.strides[0] // 2
            for k in ("kps_un", "uright", "kp_depth", "lines_un", "uright_start", "uright_end", "depth_start", "depth_end"):
                if rgbd_out is not None and k in rgbd_out:
                    setattr(R, k, rgbd_out[k].ctypes.data)
            st = L.lib().plf_batch_extract_rgbd(self._h, images.ctypes.data, n, w, h, images.strides[1], images.strides[0], C.byref(O), C.byref(R))
        if st not in (L.PLF_OK, L.PLF_E_CAPACITY):
            L.check(st, "plf_batch_extract")
        return st

    def extract(self, images, depth=None, cam=None, depth_factor=1.0 / 5000.0):
        """-> list of per-frame dicts (kps, desc, lines, ldesc, line_eq[, match_of_kp, n_kp_matches, match_of_line, n_line_matches]); with cam:
        the RGB-D Frame constructor -- also kps_un, uright, kp_depth, lines_un, uright_start / _end, depth_start / _end"""
        images = np.asarray(images)
        out = self.alloc_outputs(images.shape[0])
        ro = self.alloc_rgbd_outputs(images.shape[0]) if cam is not None else None
        self.extract_into(images, out, depth, cam, depth_factor, ro)
        res = []
        for f in range(images.shape[0]):
            d = {}
            if self.nfeatures > 0:
                k = int(out["n_kps"][f]); d["kps"] = out["kps"][f, :k].copy(); d["desc"] = out["desc"][f, :k].copy()
                if self._has_map:
                    d["match_of_kp"] = out["match_of_kp"][f, :k].copy(); d["n_kp_matches"] = int(out["n_kp_matches"][f])
                if ro is not None:
                    for q in ("kps_un", "uright", "kp_depth"):
                        d[q] = ro[q][f, :k].copy()
            if self.nlines > 0:
                k = int(out["n_lines"][f]); d["lines"] = out["lines"][f, :k].copy(); d["ldesc"] = out["ldesc"][f, :k].copy()
                d["line_eq"] = out["line_eq"][f, :k].copy()
                if self._has_map:
                    d["match_of_line"] = out["match_of_line"][f, :k].copy(); d["n_line_matches"] = int(out["n_line_matches"][f])
                if ro is not None:
                    for q in ("lines_un", "uright_start", "uright_end", "depth_start", "depth_end"):
                        d[q] = ro[q][f, :k].copy()
            res.append(d)
        return res

    def truncated_frames(self):
        """frames of the last extract whose line extraction ran out of max_ms"""
        f = L.lib().plf_batch_truncated_frames
        f.restype = C.c_int64
        return int(f(self._h))

    def worker_affinity(self, worker):
        """(NUMA node of the worker's GPU or -1, CPUs its thread was bound to or 0) -- plf_batch_worker_affinity"""
        node, n = C.c_int32(-2), C.c_int32(-1)
        L.check(L.lib().plf_batch_worker_affinity(self._h, int(worker), C.byref(node), C.byref(n)), "plf_batch_worker_affinity")
        return node.value, n.value

    def last_timing(self):
        t = (C.c_double * 4)()
        L.check(L.lib().plf_batch_last_timing(self._h, t), "plf_batch_last_timing")
        return dict(total=t[0], staging=t[1], gpu_wait=t[2], unpack=t[3])

    def worker_timing(self, worker):
        """seconds worker `worker` (= GPU devices[worker]) spent in the last extract: total, staging copies, waiting for its GPU, unpacking outputs"""
        t = (C.c_double * 4)()
        L.check(L.lib().plf_batch_worker_timing(self._h, int(worker), t), "plf_batch_worker_timing")
        return dict(total=t[0], staging=t[1], gpu_wait=t[2], unpack=t[3])
