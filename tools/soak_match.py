"""Time-bounded randomised parity soak of the matchers: the random-scene GPU parity tests of tests/test_gpu_match.py are re-run with fresh seeds and
sizes (map / last-frame / BoW projection searches, Fuse x2, the Scw SearchByProjection, SearchBySim3, SearchForTriangulation), each comparing the HIP
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
