#!/usr/bin/env python3
"""Run a TUM RGB-D sequence through the MI355X front-end the way the reference's driver does (Examples/RGB-D/rgbd_tum.cc:84-128: imread colour + depth,
SLAM.TrackRGBD -> Frame constructor = colour->gray, depth->float, ExtractORB || ExtractLSD, UndistortKeyPoints, ComputeStereoFromRGBD), compare every
frame (or every --parity-stride-th) with the CPU oracle bit for bit, and report frame rates.  BASELINE.json configs[0, 1, 4] name such sequences; the
snapshot holds only the association files (Examples/RGB-D/associations/*.txt), so this is the harness for the day a user supplies the images.

    python tools/run_tum.py <sequence_dir> <associations.txt> [--max-frames N] [--in-flight 8] [--parity-stride 10] [--camera TUM1]
                            [--camera-rgb 1] [--nfeatures 1000] [--nlines 100] [--json out.json]

<sequence_dir>/rgb/*.png (8-bit colour or gray) and <sequence_dir>/depth/*.png (16-bit, 5000 units per metre) are decoded with the in-tree reader
(rgbd_pl_slam_amd/png.py).  Colour order: cv::imread hands the reference B, G, R in memory and the yaml key Camera.RGB (1 in TUM1.yaml:29) selects
cvtColor(RGB2GRAY) for it, i.e. the reference weights the BLUE channel with 0.299 on these files; --camera-rgb takes the yaml value and the harness
feeds the front-end the same memory layout and the same flag, quirk included.
Two passes: (a) one frame at a time (the drop-in loop of configs[1]): per-frame latency; (b) --in-flight frames per chunk through the batch driver:
throughput.  Exit code 1 on any parity mismatch."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # tests/orc.py: ctypes wrapper of the CPU oracle (the checker)

import numpy as np

CAMERAS = {   # Examples/RGB-D/TUM{1,2,3}.yaml
    "TUM1": dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, k1=0.262383, k2=-0.953104, p1=-0.005358, p2=0.002628, k3=1.163314, bf=40.0),
    "TUM2": dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, k1=0.231222, k2=-0.784899, p1=-0.003257, p2=-0.000105, k3=0.917205, bf=40.0),
    "TUM3": dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0, bf=40.0),
}


def load_frames(seq_dir, assoc, first, count):
    """-> (colour (n, H, W, 3) uint8 in cv::imread memory order B, G, R -- or (n, H, W) for gray files --, depth (n, H, W) uint16, timestamps)"""
    from rgbd_pl_slam_amd import png
    cols, deps, ts = [], [], []
    for t, rgb, dep in assoc[first:first + count]:
        c = png.read_png(os.path.join(seq_dir, rgb))
        if c.ndim == 3:
            c = np.ascontiguousarray(c[..., 2::-1])          # R, G, B[, A] in the file -> B, G, R as cv::imread delivers
        d = png.read_png(os.path.join(seq_dir, dep))
        if d.dtype != np.uint16 or d.ndim != 2:
            raise ValueError("%s: 16-bit single-channel depth expected" % dep)
        cols.append(c); deps.append(d); ts.append(t)
    return np.stack(cols), np.stack(deps), ts


def oracle_frame(orc, col, dep, camera_rgb, cam9, bf, nfeatures, nlines, factor):
    gray = col if col.ndim == 2 else orc.rgb_to_gray(np.ascontiguousarray(col), 0 if camera_rgb else 1)
    ro = orc.orb_extract(gray, nfeatures=nfeatures)
    rl = orc.line_extract(gray, nlines)
    df = orc.depth_to_float(np.ascontiguousarray(dep), np.float32(factor))
    un, ur, kd = orc.frame_tail(ro["kps"], df, cam9, bf)
    lun, urs, ure, ds, de = orc.line_tail(rl["kl"], df, cam9, bf)
    return dict(kps=ro["kps"], desc=ro["desc"], lines=rl["kl"], ldesc=rl["desc"], kps_un=un, uright=ur, kp_depth=kd, lines_un=lun, uright_start=urs,
                uright_end=ure, depth_start=ds, depth_end=de)


def same_frame(got, ref):
    """bit-exact comparison of every Frame member the front-end fills; -> list of the members that differ"""
    bad = []
    for k in ("kps", "lines", "kps_un", "lines_un"):
        if len(got[k]) != len(ref[k]) or got[k].tobytes() != ref[k].tobytes():
            bad.append(k)
    for k in ("desc", "ldesc"):
        if got[k].shape != ref[k].shape or not np.array_equal(got[k], ref[k]):
            bad.append(k)
    for k in ("uright", "kp_depth", "uright_start", "uright_end", "depth_start", "depth_end"):
        if got[k].shape != ref[k].shape or not np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)):
            bad.append(k)
    return bad


def run(seq_dir, assoc_path, max_frames=0, in_flight=8, parity_stride=10, camera="TUM1", camera_rgb=1, nfeatures=1000, nlines=100, chunk=64,
        depth_factor=5000.0):
    import orc
    from rgbd_pl_slam_amd import png
    from rgbd_pl_slam_amd.batch import BatchExtractor, FMT_GRAY8, FMT_RGB8, FMT_BGR8
    from rgbd_pl_slam_amd.frame import camera as make_camera
    assoc = png.read_associations(assoc_path)
    if max_frames > 0:
        assoc = assoc[:max_frames]
    if not assoc:
        raise ValueError("no frames in %s" % assoc_path)
    cp = CAMERAS[camera]
    cam = make_camera(**cp)
    cam9 = np.array([cp[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")], np.float32)
    factor = 1.0 / depth_factor        # Tracking ctor: mDepthMapFactor = 1 / DepthMapFactor (TUM1.yaml:35)
    col0, dep0, _ = load_frames(seq_dir, assoc, 0, 1)
    h, w = dep0.shape[1:]
    fmt = FMT_GRAY8 if col0.ndim == 3 else (FMT_RGB8 if camera_rgb else FMT_BGR8)   # (n, H, W) = gray files
    single = BatchExtractor(nfeatures=nfeatures, nlines=nlines, width=w, height=h, frames_in_flight=1, devices=[0], input_format=fmt, rgbd=True)
    batch = BatchExtractor(nfeatures=nfeatures, nlines=nlines, width=w, height=h, frames_in_flight=in_flight, devices=[0], input_format=fmt, rgbd=True)
    lat, checked, mismatches, t_batch, n_done, t_decode = [], 0, [], 0.0, 0, 0.0
    for first in range(0, len(assoc), chunk):
        t0 = time.perf_counter()
        col, dep, ts = load_frames(seq_dir, assoc, first, chunk)
        t_decode += time.perf_counter() - t0
        n = len(ts)
        # (a) the drop-in loop: one frame in flight
        res1 = []
        for f in range(n):
            t0 = time.perf_counter()
            res1.append(single.extract(col[f:f + 1], depth=dep[f:f + 1], cam=cam, depth_factor=factor)[0])
            lat.append(time.perf_counter() - t0)
        # (b) the batch driver
        t0 = time.perf_counter()
        resb = batch.extract(col, depth=dep, cam=cam, depth_factor=factor)
        t_batch += time.perf_counter() - t0
        n_done += n
        for f in range(n):
            g = first + f
            if same_frame(resb[f], res1[f]):
                mismatches.append((g, "batch vs single", same_frame(resb[f], res1[f])))
            if parity_stride > 0 and g % parity_stride == 0:
                ref = oracle_frame(orc, col[f], dep[f], camera_rgb, cam9, cp["bf"], nfeatures, nlines, factor)
                bad = same_frame(res1[f], ref)
                checked += 1
                if bad:
                    mismatches.append((g, "GPU vs oracle", bad))
    single.close(); batch.close()
    lat = np.array(lat)
    skip = min(3, len(lat) - 1)        # the first calls include allocations
    out = {"sequence": os.path.abspath(seq_dir), "frames": n_done, "size": [w, h], "camera": camera, "camera_rgb": camera_rgb,
           "nfeatures": nfeatures, "nlines": nlines,
           "single_frame_ms": {"median": round(1e3 * float(np.median(lat[skip:])), 3), "p95": round(1e3 * float(np.percentile(lat[skip:], 95)), 3),
                               "frames_per_s": round(1.0 / float(np.median(lat[skip:])), 1)},
           "batch": {"frames_in_flight": in_flight, "frames_per_s": round(n_done / t_batch, 1)},
           "png_decode_s_per_frame": round(t_decode / n_done, 3),
           "parity": {"frames_checked_against_oracle": checked, "mismatches": [{"frame": g, "what": wht, "members": bad} for g, wht, bad in mismatches]}}
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sequence_dir")
    ap.add_argument("associations")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--in-flight", type=int, default=8)
    ap.add_argument("--parity-stride", type=int, default=10, help="compare every n-th frame with the CPU oracle (0: no parity pass)")
    ap.add_argument("--camera", default="TUM1", choices=sorted(CAMERAS))
    ap.add_argument("--camera-rgb", type=int, default=1, help="the yaml's Camera.RGB")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--nlines", type=int, default=100)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    out = run(a.sequence_dir, a.associations, a.max_frames, a.in_flight, a.parity_stride, a.camera, a.camera_rgb, a.nfeatures, a.nlines)
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.json:
        with open(a.json, "w") as fh:
            fh.write(txt + "\n")
    sys.exit(1 if out["parity"]["mismatches"] else 0)


if __name__ == "__main__":
    main()
