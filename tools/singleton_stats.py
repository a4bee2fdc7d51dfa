"""python tools/singleton_stats.py: region-size statistics of the oracle's LSD loop on the two synthetic families (tools/singleton_stats.c)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "libsingle.so"))
L.orc_stats.restype = C.POINTER(C.c_long)
st = L.orc_stats()
for fam, gen in (("polygons", synth_frame), ("natural", natural_frame)):
    for i in range(16): st[i] = 0
    N = 4
    for s in range(N):
        g = np.ascontiguousarray(gen(100 + s), np.uint8); h, w = g.shape
        lines = np.zeros((1 << 15, 4), np.float32)
        L.orc_lsd_detect(g.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h), C.c_ssize_t(w), C.c_int(0), lines.ctypes.data_as(C.c_void_p), C.c_int(1 << 15), None)
    print("%-9s per frame: regions %7.0f  accepted pixels %8.0f | single-pixel regions %7.0f (%.0f %%), foreseen by the static neighbour test %7.0f (%.0f %% of them) | regions of <= 3 pixels %7.0f holding %7.0f pixels"
          % (fam, st[0] / N, st[3] / N, st[1] / N, 100.0 * st[1] / max(st[0], 1), st[2] / N, 100.0 * st[2] / max(st[1], 1), st[4] / N, st[5] / N))
