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


def test_banded_speculative_region_growing_model_is_exact():
    """CPU model of the scheme behind k_lsd_spec_grow / k_lsd_spec_commit (oracle/lsd_oracle.c: orc_lsd_band_speculation): for every band
    count the committed USED map and the rectangles equal the serial seed loop's bit for bit, and only a small share of the work is redone."""
    import ctypes as C
    import orc
    from rgbd_pl_slam_amd.synth import synth_frame
    L = orc.lib()
    rng = np.random.default_rng(5)
    imgs = [synth_frame(11), (rng.integers(0, 256, (480, 640)) // 64 * 64).astype(np.uint8)]
    for img in imgs:
        img = np.ascontiguousarray(img)
        for nb in (1, 3, 8, 20):
            st = (C.c_long * 8)()
            ok = L.orc_lsd_band_speculation(img.ctypes.data_as(C.c_void_p), 640, 480, C.c_ssize_t(640), nb, st)
            assert ok == 1 and st[6] == 1, (nb, list(st))
            if nb == 1:
                assert st[2] == 0 and st[4] == 0          # a single band is the serial loop: nothing to redo
            if nb == 8:
                assert st[2] < 0.25 * st[0]               # redone accept steps stay a small share
                redo_plain = st[2]
                L.orc_lsd_band_speculation_halo(16)       # warm-up rows above every band: still exact, far less to redo
                st2 = (C.c_long * 8)()
                ok2 = L.orc_lsd_band_speculation(img.ctypes.data_as(C.c_void_p), 640, 480, C.c_ssize_t(640), nb, st2)
                L.orc_lsd_band_speculation_halo(0)
                assert ok2 == 1 and st2[6] == 1 and st2[2] < 0.8 * redo_plain


def test_validation_rounds_model_is_exact_with_both_validity_rules():
    """CPU model of the parallel validation rounds (k_lsd_spec_prefix / k_lsd_spec_validate; oracle/lsd_oracle.c: orc_lsd_band_rounds, the GPU's warm-up =
    mode 3 with 4 rows): the fixpoint equals the serial seed loop bit for bit with the rule 'any differing flag in the 3x3 dilation of the accepted set
    invalidates a record' and with the refined rule the kernels ship (lsd_kernels.hip, spec_flag_matters), which redoes less.  (The rule needs every accepted
    pixel in the record: a GPU soak found the one list that missed some, reduce_region_radius -- fixed there.)"""
    import ctypes as C
    from rgbd_pl_slam_amd.synth import synth_frame, natural_frame
    L = orc.lib()
    rng = np.random.default_rng(9)
    imgs = [synth_frame(21, 320, 240), natural_frame(22, 320, 240), (rng.integers(0, 256, (240, 320)) // 32 * 32).astype(np.uint8)]
    L.orc_lsd_band_rounds_mode(3); L.orc_lsd_band_speculation_halo(4)
    try:
        for img in imgs:
            img = np.ascontiguousarray(img, np.uint8)
            redo = []
            for refined in (0, 1):
                L.orc_lsd_band_rounds_refined(refined)
                for nb in (6, 24):
                    st = (C.c_long * 8)()
                    ok = L.orc_lsd_band_rounds(img.ctypes.data_as(C.c_void_p), 320, 240, C.c_ssize_t(320), nb, st)
                    assert ok == 1 and st[6] == 1, (refined, nb, list(st))
                    if nb == 24:
                        redo.append(st[3])
            assert redo[1] <= redo[0]
    finally:
        L.orc_lsd_band_rounds_refined(0); L.orc_lsd_band_rounds_mode(0); L.orc_lsd_band_speculation_halo(0)
