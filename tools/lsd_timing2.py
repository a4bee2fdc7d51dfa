"""per-section cycle counts of one band wave / of frame 0 (tools/lsd_timing.sh builds the -DPLF_LSD_TIMING library): python tools/lsd_timing2.py [polygons|natural] [seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rgbd_pl_slam_amd._lib as L
L.LIB_PATH = os.environ.get("PLF_TIMING_LIB", "/tmp/plft/libplf_hip.so")
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
fam = sys.argv[1] if len(sys.argv) > 1 else "polygons"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
img = natural_frame(seed) if fam == "natural" else synth_frame(seed)
ls = LineSegment(nlines=100)
for _ in range(3):
    ls.ExtractLineSegment(img)
print("family", fam, "seed", seed, "(3 calls accumulated)")
L.lib().plf_lsd_timing_dump()
