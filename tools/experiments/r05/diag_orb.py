import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import orc
from rgbd_pl_slam_amd import ORBextractor
from rgbd_pl_slam_amd.synth import synth_frame
libm = ctypes.CDLL("libm.so.6")
libm.sincosf.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
tot = 0
for (seed, w, h, nf) in ((4, 752, 480, 1200), (0, 640, 480, 1000), (3, 640, 480, 2000), (5, 1280, 960, 4000), (6, 640, 480, 1000), (7, 640, 480, 1000)):
    img = synth_frame(seed, w, h)
    ext = ORBextractor(nfeatures=nf, max_width=w, max_height=h)
    kps, desc = ext(img)
    ref = orc.orb_extract(img, nfeatures=nf)
    same_k = len(kps) == len(ref["kps"]) and all(np.array_equal(kps[f].view(np.uint32), ref["kps"][f].view(np.uint32)) for f in ("x", "y", "angle", "response"))
    bad = np.nonzero((desc != ref["desc"]).any(1))[0] if len(desc) == len(ref["desc"]) else []
    print(seed, w, h, nf, "n", len(kps), "kps_equal", same_k, "bad desc rows", len(bad))
    for i in bad[:10]:
        ang = np.float32(kps["angle"][i]) * np.float32(0.01745329238)
        s, c = ctypes.c_float(), ctypes.c_float()
        libm.sincosf(ctypes.c_float(ang), ctypes.byref(s), ctypes.byref(c))
        cd, sd = np.float32(np.cos(np.float64(ang))), np.float32(np.sin(np.float64(ang)))
        nb = int(np.unpackbits(desc[i] ^ ref["desc"][i]).sum())
        print("   row", i, "bits", nb, "angle", kps["angle"][i], "libm", s.value.hex(), c.value.hex(), "dbl", float(sd).hex(), float(cd).hex())
    ext.close()
