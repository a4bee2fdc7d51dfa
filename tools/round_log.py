"""Per-round anatomy of the validation rounds (few frames in flight): python tools/round_log.py <polygons|natural|photo> [B=1] [calls=4]   (photo: PHOTO=k picks the photograph, 6 = gravel)
Needs a -DPLF_ROUND_LOG library:  bash tools/variant_build.sh rl lsd_kernels.hip=-DPLF_ROUND_LOG line_host.hip=-DPLF_ROUND_LOG
                                  PLF_LIB_PATH=tools/scratch/libplf_rl.so PLF_LSD_ROUND_LOG=1 python tools/round_log.py natural 1"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rgbd_pl_slam_amd._lib as L
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_frame, natural_frame, photo_frame
fam = sys.argv[1] if len(sys.argv) > 1 else "polygons"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gen = natural_frame if fam == "natural" else (lambda s: photo_frame(51000 + int(os.environ.get("PHOTO", "6")) + 7 * (s % 2))) if fam == "photo" else synth_frame
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
lib = L.lib()
lib.plf_line_debug_round_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
for c in range(N):
    imgs = np.stack([gen(200 + c * B + i) for i in range(B)])
    ls.extract_batch(imgs); ls.extract_batch(imgs)
    buf = np.zeros(B * 64 * 64, np.int32); bt = np.zeros(B * 64 * 2, np.int32)
    nb = lib.plf_line_debug_round_log(ls._h, L.vp(buf), B, L.vp(bt))
    if nb <= 0:
        print("no round log (library without -DPLF_ROUND_LOG or PLF_LSD_ROUND_LOG unset):", nb); break
    rl = buf[:B * nb * 64].reshape(B, nb, 16, 4).copy(); bt = bt[:B * nb * 2].reshape(B, nb, 2)
    same = (rl[:, :, :, 1] >> 16) & 0xFFFF; same_px = (rl[:, :, :, 2] >> 16) & 0xFFFF      # regrown records that came out with the accepted list they had
    b_only = (rl[:, :, :, 3] >> 16) & 0xFF; b_small = (rl[:, :, :, 3] >> 24) & 0xFF  # invalid records none of whose accepted pixels is truly taken (only "seen taken, truly free" neighbours); of those, regions below the minimum size
    rl[:, :, :, 1] &= 0xFFFF; rl[:, :, :, 2] &= 0xFFFF; rl[:, :, :, 3] &= 0xFFFF
    g = bt[:, :, 0] / 100.0
    print("%s call %d, %d frame(s) x %d bands: band waves us max %.0f mean %.0f (slowest band of each frame: %s), logged px %d" %
          (fam, c, B, nb, g.max(), g.mean(), " ".join("%.0f" % x for x in g.max(axis=1)), int(bt[:, :, 1].sum())))
    if os.environ.get("ROUND_LOG_BANDS"):
        by = np.zeros(nb + 1, np.int32)
        print("   frame 0 per band (us of the band wave, logged px | us of rounds 1..4):")
        for b in range(nb):
            print("     band %2d: %5.0f us %5d px | %s" % (b, g[0, b], bt[0, b, 1], " ".join("%4.0f(%d,%d)" % (rl[0, b, r, 0] / 100.0, rl[0, b, r, 1], rl[0, b, r, 2]) for r in range(5))))
    for r in range(16):
        t = rl[:, :, r, 0] / 100.0
        act = (rl[:, :, r, 0] > 0)
        if not act.any(): continue
        redo = rl[:, :, r, 1] > 0
        print("   round %2d: bands that ran %3d, that regrew %3d | us max %.0f, mean of active %.0f | seeds regrown %d (max per band %d), pixels regrown %d (max per band %d), records standing %d | regrown to the SAME list: %d seeds, %d pixels | invalid without a taken accepted pixel: %d (small regions: %d)" %
              (r + 1, int(act.sum()), int(redo.sum()), t.max(), t[act].mean(), int(rl[:, :, r, 1].sum()), int(rl[:, :, r, 1].max()), int(rl[:, :, r, 2].sum()),
               int(rl[:, :, r, 2].max()), int(rl[:, :, r, 3].sum()), int(same[:, :, r].sum()), int(same_px[:, :, r].sum()), int(b_only[:, :, r].sum()), int(b_small[:, :, r].sum())))
    # what the round barriers cost: Jacobi rounds as launched (every round lasts as long as its slowest band of ANY frame) against the same passes under
    # dataflow synchronisation (band b starts round r when the bands b' < b of ITS frame are through round r - 1: that is all pre[b] depends on)
    t = rl[:, :, :, 0] / 100.0
    R = max([r for r in range(16) if (rl[:, :, r, 0] > 0).any()] + [0]) + 1
    jac = g.max() + sum(t[:, :, r].max() for r in range(R))
    per_frame = []
    for f in range(B):
        Cprev = g[f].copy()                      # completion of the band waves
        for r in range(R):
            pm = np.maximum.accumulate(np.concatenate([[0.0], Cprev[:-1]]))   # max over b' < b of C[b'][r-1]
            Cprev = np.maximum(Cprev, pm) + t[f, :, r]
        per_frame.append(Cprev.max())
    fj = [g[f].max() + sum(t[f, :, r].max() for r in range(R)) for f in range(B)]
    print("   band waves + rounds, us: as launched (global barriers) %.0f | barriers per frame %.0f | dataflow per frame %.0f | mean band %.0f" %
          (jac, max(fj), max(per_frame), g.mean() + t.sum(axis=2).mean()))
ls.close()
