"""Time-bounded randomised parity soak of the matchers: the random-scene GPU parity tests of tests/test_gpu_match.py are re-run with fresh seeds and
sizes (map / last-frame / BoW projection searches, Fuse x2, the Scw SearchByProjection, SearchBySim3, SearchForTriangulation, and every LSDmatcher
overload: kNN, the MAD rule, SearchForTriangulation, Fuse, the map-line projection search), each comparing the HIP
result with the oracle element for element.

    python tools/soak_match.py [seconds=240] [first_seed=1000]

Prints one summary line; failures are listed with the parameters that reproduce them.  Exit code 1 on any failure."""
import sys, os, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import orc
import test_gpu_match as T
from rgbd_pl_slam_amd import Matcher

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000


def bow_case(seed, rng):
    nkf = int(rng.integers(1, 2000)); nf = int(rng.integers(1, 2000)); nn = int(rng.integers(1, 600))
    ratio = float(rng.choice([0.6, 0.7, 0.75, 0.9])); chk = int(rng.integers(0, 2)); shared = bool(rng.integers(0, 2))
    c = T._bow_random_case(seed, nkf, nf, nn, shared)
    exp = orc.search_by_bow(c["kf_desc"], c["f_desc"], c["kf_angle"], c["f_angle"], c["kf_has_mp"], c["kf_nodes"], c["f_nodes"], ratio, chk)
    m = Matcher(max_keypoints=2048, max_mappoints=16, max_batch=2)
    t = [T._dev(c[k]) for k in ("kf_desc", "f_desc", "kf_angle", "f_angle", "kf_has_mp")]
    kn = tuple(T._dev(x) for x in c["kf_nodes"]); fn = tuple(T._dev(x) for x in c["f_nodes"])
    view = Matcher.bow_view(t[0], t[1], t[2], t[3], t[4], kn, fn)
    match = torch.full((1, 2048), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    m.SearchByBoW([view], ratio, chk, match, 2048, nm)
    torch.cuda.synchronize()
    ok = int(nm[0]) == exp[1] and np.array_equal(match[0, :nf].cpu().numpy(), exp[0])
    m.close()
    assert ok, ("bow", seed, nkf, nf, nn, ratio, chk, shared)


_LINE_POOL = {}


def line_case(seed, rng):
    """rows 13 / 14 and the two keyframe overloads of LSDmatcher with random sizes, noise, occupancy and thresholds: kNN (k = 2), the MAD rule in
    both directions, SearchForTriangulation, Fuse and the projection search against random map lines"""
    import matchgen
    from rgbd_pl_slam_amd.synth import texture_frame, synth_frame
    def lines_of(k):
        if k not in _LINE_POOL:   # LSD on the CPU oracle is the slow part: a small pool of frames is reused with fresh perturbations
            img = synth_frame(3000 + k) if k % 3 else texture_frame(3000 + k, kind=(0, 1, 3, 11)[(k // 3) % 4], size=(640, 480))[0]
            _LINE_POOL[k] = orc.line_extract(img, 400)
        return _LINE_POOL[k]
    a = lines_of(int(rng.integers(0, 12))); b = lines_of(int(rng.integers(0, 12)))
    na = int(rng.integers(1, len(a["desc"]) + 1)); keep = int(rng.integers(0, na + 1)); flips = int(rng.choice([0, 3, 10, 25, 60]))
    d1 = np.ascontiguousarray(a["desc"][:na])
    nb = int(rng.integers(0, len(b["desc"]) + 1))
    d2 = np.concatenate([matchgen.flip_bits(d1[:keep], rng, flips), b["desc"][:nb]]) if keep + nb > 0 else np.zeros((0, 32), np.uint8)
    d2 = np.ascontiguousarray(d2[rng.permutation(len(d2))])
    has1 = (rng.uniform(0, 1, len(d1)) < rng.uniform(0, 1)).astype(np.uint8); has2 = (rng.uniform(0, 1, max(len(d2), 1)) < rng.uniform(0, 1)).astype(np.uint8)
    st1 = (rng.uniform(0, 1, len(d1)) < 0.7).astype(np.uint8); st2 = (rng.uniform(0, 1, max(len(d2), 1)) < 0.7).astype(np.uint8)
    m = Matcher(max_lines=1024, max_mappoints=4096)
    dev = T._dev
    t1, t2 = dev(d1), dev(d2 if len(d2) else np.zeros((1, 32), np.uint8))
    n2 = len(d2)
    nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    ok = True
    if n2 >= 1:
        idx, dist = orc.knn2(d1, d2)
        dm = m.knnMatch(t1, t2[:n2])
        ok &= np.array_equal(dm["trainIdx"], idx) and np.array_equal(dm["distance"], dist.astype(np.float32))
    rm, rn = orc.match_lines_knn(d1, d2, has1) if n2 >= 2 else (np.full(n2, -1, np.int32), 0)
    match = torch.full((max(n2, 1),), -1, dtype=torch.int32, device="cuda")
    m.SearchLinesLastFrame(t1, t2[:n2], dev(has1), match, nm)
    torch.cuda.synchronize()
    ok &= int(nm[0]) == rn and np.array_equal(match.cpu().numpy()[:n2], rm)
    only = int(rng.integers(0, 2)); fac = float(rng.choice([0.0, 0.1, 0.5, 2.0]))
    rm, rn = orc.lines_search_for_triangulation(d1, d2, has1, has2[:n2] if n2 else has2[:0], st1, st2[:n2] if n2 else st2[:0], only, fac)
    match12 = torch.zeros(len(d1), dtype=torch.int32, device="cuda")
    m.SearchLinesForTriangulation(t1, t2[:n2], dev(has1), dev(has2), dev(st1), dev(st2), only, match12, nm, mad_factor=fac)
    torch.cuda.synchronize()
    ok &= int(nm[0]) == rn and np.array_equal(match12.cpu().numpy(), rm)
    valid = (rng.uniform(0, 1, len(d1)) < 0.8).astype(np.uint8)
    fb, fn = orc.lines_fuse(d2, d1, valid)
    best = torch.zeros(len(d1), dtype=torch.int32, device="cuda")
    m.FuseLines(t2[:n2], t1, dev(valid), best, nm)
    torch.cuda.synchronize()
    ok &= int(nm[0]) == fn and np.array_equal(best.cpu().numpy(), fb)
    # projection search of random map lines into the frame of `a`
    kl = a["kl"][:na]
    M = int(rng.integers(1, 3000)); th = float(rng.choice([1.0, 3.0, 5.0])); nnr = float(rng.choice([0.6, 0.8, 0.9]))
    ml = matchgen.make_map_lines(kl, d1, M, seed)
    init = np.full(na, -1, np.int32); init[rng.uniform(0, 1, na) < 0.1] = -2
    scale = orc.orb_tables(1000, 1.2, 8)["scale"]
    rm, rn = orc.search_lines_by_projection(kl, d1, scale, ml, th, nnr, init)
    dkl = torch.from_numpy(np.frombuffer(np.ascontiguousarray(kl).tobytes(), np.uint8).copy()).cuda(); dsc = dev(scale)
    view = Matcher.lineframe_view(na, dkl, t1, dsc)
    dml = {k: dev(v) for k, v in ml.items()}
    pm = dev(init)
    m.SearchLinesByProjection([view], dml, th, nnr, pm, na, nm)
    torch.cuda.synchronize()
    ok &= int(nm[0]) == rn and np.array_equal(pm.cpu().numpy(), rm)
    m.close()
    assert ok, ("lines", seed, na, keep, flips, nb, only, fac, M, th, nnr)


def main():
    t_end = time.time() + SECONDS
    seed = SEED0
    counts = {}
    bad = []
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        jobs = [
            ("map", lambda: T.test_search_by_projection_map(int(rng.integers(1, 16000)), float(rng.choice([1.0, 3.0, 5.0, 9.0])), bool(rng.integers(0, 2)))),
            ("lastframe", lambda: T.test_search_by_projection_lastframe(seed, int(rng.integers(0, 2)), int(rng.integers(0, 2)))),
            ("kf-family", lambda: T.test_keyframe_projection_family_random_scenes(seed, int(rng.integers(1, 4000)), int(rng.integers(1, 16000)))),
            ("two-kf", lambda: T.test_two_keyframe_overloads_random_scenes(seed, int(rng.integers(200, 4000)), float(rng.uniform(0.03, 0.5)), float(rng.uniform(0.9, 1.1)))),
            ("bow", lambda: bow_case(seed, rng)),
            ("lines", lambda: line_case(seed, rng)),
        ]
        for name, job in jobs:
            try:
                job()
                counts[name] = counts.get(name, 0) + 1
            except AssertionError:
                tb = traceback.extract_tb(sys.exc_info()[2])[-1]
                bad.append((name, seed, (tb.line or "").strip()[:200]))
            except Exception as ex:                                 # API errors count as failures
                bad.append((name, seed, repr(ex)[:200]))
        seed += 1
    print("soak_match: seeds %d..%d, passed %s, %d failures (an assert that only says a random scene is uninteresting -- no matches at all -- also lands here)" % (SEED0, seed - 1, counts, len(bad)))
    for b in bad[:40]:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
