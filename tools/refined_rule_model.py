"""python tools/refined_rule_model.py: the CPU model of the validation rounds (oracle/lsd_oracle.c, orc_lsd_band_rounds: 48 bands, the GPU's warm-up = mode 3
with 4 rows) with the validity rule as shipped and REFINED -- a dirty neighbour that the speculation saw free and that is truly taken invalidates a record only
if the record accepted it.  Prints the accepts redone (total and along the critical path), regions redone, rounds, exactness."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
L = orc.lib()
L.orc_lsd_band_rounds_mode(3); L.orc_lsd_band_speculation_halo(4)
for fam, gen in (("polygons", synth_frame), ("natural", natural_frame)):
    for seed in (100, 101, 102):
        img = np.ascontiguousarray(gen(seed), np.uint8)
        row = []
        for refined in (0, 1):
            L.orc_lsd_band_rounds_refined(refined)
            st = (C.c_long * 8)()
            ok = L.orc_lsd_band_rounds(img.ctypes.data_as(C.c_void_p), 640, 480, C.c_ssize_t(640), 48, st)
            L.orc_lsd_band_rounds_needless.restype = C.c_long
            row.append((ok, st[0], st[1], st[2], st[3], st[4], st[7], L.orc_lsd_band_rounds_needless()))
        a, b = row
        print("%-9s seed %d: serial accepts %6d, slowest band %5d | shipped rule: exact %d redo total %6d critical %5d regions %5d rounds %2d | refined: exact %d redo total %6d critical %5d regions %5d rounds %2d, of the redone accepts %d reproduced the old record"
              % (fam, seed, a[1], a[2], a[0], a[4], a[3], a[6], a[5], b[0], b[4], b[3], b[6], b[5], b[7]))
L.orc_lsd_band_rounds_refined(0); L.orc_lsd_band_rounds_mode(0); L.orc_lsd_band_speculation_halo(0)
