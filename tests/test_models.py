"""numpy MODELS of the data-parallel reformulations used by the HIP kernels, checked against the
sequential oracle (and through it against the reference-binary fixtures)."""
import json
import os

import numpy as np
import pytest

import orc
import refgen
from models.octree_parallel import distribute_octree as octree_model

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_parallel_octree_model_matches_reference_fixture():
    cases = json.load(open(os.path.join(GOLD, "ref_octree_cases.json")))["cases"]
    for c in cases:
        W, H, N, k = refgen.octree_case(c["seed"], c["cluster"], c["resp_levels"])
        out = octree_model(k[:, 0].copy(), k[:, 1].copy(), k[:, 2].copy(), 16, 16 + W, 16, 16 + H, N)
        assert out.tolist() == c["out"], "seed %d" % c["seed"]


@pytest.mark.parametrize("seed", range(12))
def test_parallel_octree_model_matches_oracle_random(seed):
    rng = np.random.default_rng(seed)
    W = int(rng.integers(40, 1300)); H = int(rng.integers(30, min(W, 1000) + 1))
    nk = int(rng.integers(1, 5000)); N = int(rng.integers(1, 900))
    k = np.zeros((nk, 3), np.float32)
    if seed % 3 == 0:  # duplicates and clustered points
        k[:, 0] = rng.integers(0, max(2, W // 8), nk); k[:, 1] = rng.integers(0, max(2, H // 8), nk)
    else:
        k[:, 0] = rng.integers(0, W, nk); k[:, 1] = rng.integers(0, H, nk)
    k[:, 2] = rng.integers(7, 12 if seed % 2 else 250, nk)
    ref = orc.distribute_octree(k, 16, 16 + W, 16, 16 + H, N)["class_id"]
    out = octree_model(k[:, 0].copy(), k[:, 1].copy(), k[:, 2].copy(), 16, 16 + W, 16, 16 + H, N)
    assert out.tolist() == ref.tolist()
