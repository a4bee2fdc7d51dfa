"""Time-bounded randomised parity soak: ORB and LSD+LBD extraction of many differently textured frames (synthetic scenes, their noisy /
low-contrast / quantised / rotated variants, pure noise, stripes, checkerboards, flat images; random sizes and extractor parameters) on the GPU
against the oracle, byte for byte.  The oracle runs on a thread pool (ctypes releases the GIL); the GPU paths exercised are the single-frame
entry points (banded speculation), the batched ones (speculative and serial schedules) and the published seed order.

    python tools/soak.py [seconds=300] [threads=16] [first_seed=0]

Prints one summary line; every mismatch is listed with the seed that reproduces it.  Exit code 1 on any mismatch."""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import orc
from rgbd_pl_slam_amd import ORBextractor, LineSegment
from rgbd_pl_slam_amd.synth import synth_frame, texture_frame

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
THREADS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
SEED0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def make_image(seed):
    if seed % 5 == 4:      # a window of one of the real photographs of tests/golden/real: VGA, or a random size (reflection where the photograph is smaller)
        from rgbd_pl_slam_amd.synth import photo_frame
        rng = np.random.default_rng(55 + seed)
        if rng.random() < 0.5:
            return photo_frame(seed), "photo"
        w = int(rng.integers(240, 900)); h = int(rng.integers(max(200, w // 2 + 40), min(700, w) + 1))
        return photo_frame(seed, w, h), "photo"
    return texture_frame(seed)


def eq_orb(got, ref):
    kps, desc = got
    if len(kps) != len(ref["kps"]):
        return False
    return kps.tobytes() == ref["kps"].tobytes() and np.array_equal(desc, ref["desc"])


def eq_lines(got, ref):
    kl, desc, eq = got
    if len(kl) != len(ref["kl"]):
        return False
    return kl.tobytes() == ref["kl"].tobytes() and np.array_equal(desc, ref["desc"]) and np.allclose(eq, ref["eq"], rtol=0, atol=1e-9)


def main():
    pool = ThreadPoolExecutor(THREADS)
    t_end = time.time() + SECONDS
    seed = SEED0
    n_orb = n_line = n_batchline = 0
    bad = []
    vga_orb = ORBextractor(nfeatures=1000, max_width=640, max_height=480, max_batch=8)
    vga_line = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=8)
    while time.time() < t_end:
        group = [make_image(seed + i) for i in range(THREADS)]
        rng = np.random.default_rng(991 + seed)
        # (the octree kernel holds <= ~1900 nodes per level in LDS: plf_orb_create rejects larger per-level quotas)
        nf = int(rng.choice([300, 1000, 2000, 3000])); nl = int(rng.choice([50, 100, 200, 1000]))
        # frames that are not VGA also draw the other extractor parameters (every level keeps at least one 30-px FAST cell) and the LSD seed order
        sf = float(np.float32(rng.choice([1.1, 1.2, 1.3, 1.5]))); nlev = int(rng.integers(3, 9)); ini = int(rng.integers(12, 40)); mn = int(rng.integers(3, 12))
        so = int(rng.integers(0, 2))
        def orb_par(im):
            if im.shape == (480, 640):
                return dict(nfeatures=1000)
            k = nlev
            while min(im.shape) / (sf ** (k - 1)) < 70:
                k -= 1
            return dict(nfeatures=nf, scale_factor=sf, nlevels=k, ini_th=ini, min_th=mn)
        futs_o = [pool.submit(orc.orb_extract, im, **orb_par(im)) for im, _ in group]
        futs_l = [pool.submit(orc.line_extract, im, (100 if im.shape == (480, 640) else nl), (0 if im.shape == (480, 640) else so)) for im, _ in group]
        got_o, got_l = [], []
        vga = [i for i, (im, _) in enumerate(group) if im.shape == (480, 640)]
        for i, (im, kind) in enumerate(group):
            if im.shape == (480, 640):
                got_o.append(vga_orb(im))
                try:
                    got_l.append(vga_line.ExtractLineSegment(im))
                except Exception as ex:
                    print("EXCEPTION", seed + i, kind, im.shape, ex, flush=True)
                    raise
            else:
                h, w = im.shape
                pr = orb_par(im)
                try:
                    e = ORBextractor(nfeatures=nf, scaleFactor=pr["scale_factor"], nlevels=pr["nlevels"], iniThFAST=ini, minThFAST=mn, max_width=w, max_height=h)
                except Exception as ex:
                    print("EXCEPTION", seed + i, kind, im.shape, pr, ini, mn, ex, flush=True)
                    raise
                got_o.append(e(im)); e.close()
                ls = LineSegment(nlines=nl, max_width=w, max_height=h, seed_order=so); got_l.append(ls.ExtractLineSegment(im)); ls.close()
        # batches of the VGA frames: speculative schedule (8 frames -> 16 bands), then the serial schedule of the same frames
        batch_res = []
        for b0 in range(0, len(vga) - 7, 8):
            ids = vga[b0:b0 + 8]
            stack = np.stack([group[i][0] for i in ids])
            r_spec = vga_line.extract_batch(stack)
            # (the schedule knobs are read once per handle: plf_line_tune, not the environment -- the environment toggles here were silent no-ops in round 4)
            vga_line.tune("spec_bands", 0)
            r_ser = vga_line.extract_batch(stack)
            vga_line.tune("spec_bands", 4)
            r_4 = vga_line.extract_batch(stack)
            vga_line.tune("spec_bands", -2147483647)
            r_orb = vga_orb.extract_batch(stack)
            batch_res.append((ids, r_spec, r_ser, r_4, r_orb))
        refs_o = [f.result() for f in futs_o]; refs_l = [f.result() for f in futs_l]
        for i, (im, kind) in enumerate(group):
            n_orb += 1; n_line += 1
            if not eq_orb(got_o[i], refs_o[i]):
                bad.append(("orb", seed + i, kind, im.shape))
            if not eq_lines(got_l[i], refs_l[i]):
                bad.append(("lines", seed + i, kind, im.shape))
        for ids, r_spec, r_ser, r_4, r_orb in batch_res:
            for k, i in enumerate(ids):
                n_batchline += 3
                for name, r in (("lines-batch-spec16", r_spec), ("lines-batch-serial", r_ser), ("lines-batch-spec4", r_4)):
                    if not eq_lines(r[k], refs_l[i]):
                        bad.append((name, seed + i, group[i][1], group[i][0].shape))
                if not eq_orb(r_orb[k], refs_o[i]):
                    bad.append(("orb-batch", seed + i, group[i][1], group[i][0].shape))
        seed += THREADS
    print("soak: %d frames (seeds %d..%d): %d ORB + %d line single-frame checks, %d batched line checks, %d mismatches" %
          (seed - SEED0, SEED0, seed - 1, n_orb, n_line, n_batchline, len(bad)))
    for b in bad[:40]:
        print("MISMATCH", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
