"""The hot path driven through the C++ host mirror (include/plf.hpp) and its EXACT-signature adapters -- ORBextractor::operator()(InputArray, InputArray,
vector<KeyPoint>&, OutputArray), LineSegment::ExtractLineSegment(Mat, vector<KeyLine>&, Mat&, vector<Vector3d>&), ORBmatcher::SearchByProjection x2,
LSDmatcher::SearchByProjection x3, LSDmatcher::SearchForTriangulation / Fuse -- not through ctypes: tests/cpp/mirror_driver.cpp is compiled with g++
against tests/mock/ (a stand-in for the OpenCV / ORB_SLAM2 headers this image lacks), run on the GPU, and its outputs compared with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import matchgen
import orc
from conftest import gpu_available

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_full_path_through_the_cpp_adapters(tmp_path):
    if not gpu_available():
        pytest.fail("no GPU visible: the -m gpu tests need a real MI355X")
    from rgbd_pl_slam_amd._lib import KP_DTYPE, KL_DTYPE
    from rgbd_pl_slam_amd.synth import synth_frame
    lib = os.path.join(ROOT, "rgbd_pl_slam_amd", "libplf_hip.so")
    exe = tmp_path / "mirror_driver"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-rdynamic", "-DPLF_WITH_OPENCV", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "mock"),
                           os.path.join(ROOT, "tests", "cpp", "mirror_driver.cpp"), "-o", str(exe), lib, "-Wl,-rpath," + os.path.dirname(lib),
                           "-Wl,-rpath,/opt/rocm/lib"])
    w, h, nfeat, nlines = 640, 480, 1000, 100
    img = synth_frame(77, w, h)
    ro = orc.orb_extract(img, nfeatures=nfeat); rl = orc.line_extract(img, nlines)
    kps, desc, kl, ldesc = ro["kps"], ro["desc"], rl["kl"], rl["desc"]
    d = str(tmp_path)
    put = lambda name, arr: np.ascontiguousarray(arr).tofile(os.path.join(d, name))
    put("dims.i32", np.array([w, h, nfeat, nlines], np.int32)); put("image.u8", img)
    mp = matchgen.make_local_map(kps, desc, 2500, 11)
    for k, ext in (("proj_x", "f32"), ("proj_y", "f32"), ("proj_xr", "f32"), ("view_cos", "f32"), ("level", "i32"), ("in_view", "u8"), ("obs_positive", "u8"), ("desc", "u8")):
        put("mp_%s.%s" % (k, ext), mp[k])
    last, pose = matchgen.make_last_frame(kps, desc, 12, cx=w / 2 - 0.5, cy=h / 2 - 0.5)
    put("last_world_pos.f32", last["world_pos"]); put("last_has_mappoint.u8", last["has_mappoint"]); put("last_outlier.u8", last["outlier"]); put("last_mp_desc.u8", last["mp_desc"])
    T = np.eye(4, dtype=np.float32); T[:3, :3] = pose["Rcw"]; T[:3, 3] = pose["tcw"]
    put("cur_Tcw.f32", T)
    ml = matchgen.make_map_lines(kl, ldesc, 300, 13)
    for k, ext in (("x1", "f32"), ("y1", "f32"), ("x2", "f32"), ("y2", "f32"), ("view_cos", "f32"), ("level", "i32"), ("in_view", "u8"), ("desc", "u8")):
        put("ml_%s.%s" % (k, ext), ml[k])
    rng = np.random.default_rng(14)
    other = orc.line_extract(synth_frame(78, w, h), nlines)["desc"]
    lastl = np.concatenate([matchgen.flip_bits(ldesc[:70], rng, 15), other[:40]])
    lastl = np.ascontiguousarray(lastl[rng.permutation(len(lastl))])
    lhas = (rng.uniform(0, 1, len(lastl)) < 0.8).astype(np.uint8)
    put("lastl_desc.u8", lastl); put("lastl_has.u8", lhas)
    st1 = (rng.uniform(0, 1, len(lastl)) < 0.7).astype(np.uint8); st2 = (rng.uniform(0, 1, len(kl)) < 0.7).astype(np.uint8)
    put("tri_stereo1.u8", st1); put("tri_stereo2.u8", st2)
    kfh = (rng.uniform(0, 1, len(kl)) < 0.5).astype(np.uint8)
    put("fuse_kf_has.u8", kfh)
    # plf::BatchExtractor::extract_rgbd: 3 frames of the same image with different depth maps
    from rgbd_pl_slam_amd.frame import TUM1
    nb = 3
    d16 = rng.integers(2000, 30000, (nb, h, w), dtype=np.uint16); d16[rng.uniform(0, 1, (nb, h, w)) < 0.1] = 0
    camv = np.array([TUM1[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")], np.float32)
    put("batch_depth.u16", d16); put("batch_cam.f32", camv)
    run = subprocess.run([str(exe), d], text=True, capture_output=True)
    assert run.returncode == 0 and "mirror driver ok" in run.stdout, "driver failed (rc %d)\n%s\n%s" % (run.returncode, run.stdout, run.stderr[-2000:])
    get = lambda name, dt: np.fromfile(os.path.join(d, name), dt)
    # ---- extraction: bit-equal to the oracle
    gk = get("out_kps.bin", KP_DTYPE)
    assert len(gk) == len(kps) and all(np.array_equal(gk[n].view(np.uint32), kps[n].view(np.uint32)) for n in KP_DTYPE.names)
    assert np.array_equal(get("out_desc.u8", np.uint8).reshape(-1, 32), desc)
    gl = get("out_kl.bin", KL_DTYPE)
    assert len(gl) == len(kl) and all(np.array_equal(gl[n].view(np.uint32), kl[n].view(np.uint32)) for n in KL_DTYPE.names)
    assert np.array_equal(get("out_ldesc.u8", np.uint8).reshape(-1, 32), ldesc)
    assert np.allclose(get("out_eq.f64", np.float64).reshape(-1, 3), rl["eq"], rtol=0, atol=1e-4)
    # ---- point matchers
    scale = orc.orb_tables(nfeat, 1.2, 8)["scale"]; bounds = (0.0, 0.0, float(w), float(h))
    rm, rn = orc.search_by_projection_map(kps, desc, None, scale, bounds, mp, 3.0, 0.8, np.full(len(kps), -1, np.int32))
    g = get("out_match_map.i32", np.int32)
    assert g[-1] == rn and np.array_equal(g[:-1], rm) and rn > 100
    rm, rn = orc.search_by_projection_last(kps, desc, None, scale, bounds, last, pose, 7.0, 0, 1, np.full(len(kps), -1, np.int32))
    g = get("out_match_last.i32", np.int32)
    assert g[-1] == rn and np.array_equal(g[:-1], rm) and rn > 100
    # ---- line matchers
    rm, rn = orc.search_lines_by_projection(kl, ldesc, scale, ml, 3.0, 0.8, np.full(len(kl), -1, np.int32))
    g = get("out_lmatch_map.i32", np.int32)
    assert g[-1] == rn and np.array_equal(g[:-1], rm) and rn > 5
    rm, rn = orc.match_lines_knn(lastl, ldesc, lhas)
    for name in ("out_lmatch_last.i32", "out_lmatch_kf.i32"):     # the Frame / Frame and the KeyFrame / Frame overloads are the same rule
        g = get(name, np.int32)
        assert g[-1] == rn and np.array_equal(g[:-1], rm) and rn > 10
    # ---- the remaining LineSegment members (include/ExtractLineSegment.h:41-47)
    kidx, kdist = orc.knn2(lastl, ldesc)
    g = get("out_lsmatch.i32", np.int32).reshape(-1, 2, 2)
    assert np.array_equal(g[:, :, 0], kidx) and np.array_equal(g[:, :, 1], kdist)
    gm = get("out_lsmad.f64", np.float64)
    assert (gm[0], gm[1]) == orc.line_mad(kdist) and gm[2] == orc.line_segment_overlap(2.0, 10.0, 4.0, 7.0) == 0.5
    hml1 = lhas                                                     # keyframe 1 = the "last" lines with their MapLines; keyframe 2 holds none
    rm, rn = orc.lines_search_for_triangulation(lastl, ldesc, hml1, np.zeros(len(kl), np.uint8), st1, st2, 1, 0.1)
    g = get("out_ltri.i32", np.int32)
    assert g[-1] == rn and np.array_equal(g[:-1], rm)
    # Fuse: the search half against the oracle, then the reference's map mutation as the adapter applied it
    M = len(ml["desc"])
    valid = np.array([not (i % 7 == 0) and not (i % 11 == 0) for i in range(M)], np.uint8)
    best, nf = orc.lines_fuse(ldesc, ml["desc"], valid)
    g = get("out_lfuse.i32", np.int32); held_rep = get("out_lfuse_held.i32", np.int32)
    assert g[-1] == nf and nf > 3
    # the reference loop's mutation, replayed: slot k of the keyframe holds a MapLine from the start (kfh) or receives one (AddMapLine)
    slot = {k: ("held", k) for k in range(len(kl)) if kfh[k]}
    nobs = {("held", k): 1 + k % 4 for k in range(len(kl))}; nobs.update({("ml", i): 1 + i % 3 for i in range(M)})
    bad = set(); exp_added = np.full(M, -1); exp_rep = np.full(M, -1); exp_held = np.full(len(kl), -1)
    for i in range(M):
        if best[i] < 0:
            continue
        k = int(best[i]); cur = slot.get(k)
        if cur is not None:
            if cur not in bad:
                if nobs[cur] > nobs[("ml", i)]:
                    exp_rep[i] = cur[1] if cur[0] == "held" else 100000 + cur[1]; bad.add(("ml", i))
                else:
                    bad.add(cur)
                    if cur[0] == "held":
                        exp_held[cur[1]] = i
                    else:
                        exp_rep[cur[1]] = 100000 + i      # a map line added by an earlier iteration is replaced by this one
        else:
            nobs[("ml", i)] += 1; exp_added[i] = k; slot[k] = ("ml", i)
    assert np.array_equal(g[0:-1:2], exp_added) and np.array_equal(g[1:-1:2], exp_rep) and np.array_equal(held_rep, exp_held)
    # ---- plf::BatchExtractor (RGB-D Frame constructor per frame)
    caps = get("out_b_caps.i32", np.int32); kc, lc = int(caps[0]), int(caps[1])
    bn = get("out_b_n.i32", np.int32); bnl = get("out_b_nl.i32", np.int32)
    assert list(bn) == [len(kps)] * nb and list(bnl) == [len(kl)] * nb
    bk = get("out_b_kps.bin", KP_DTYPE).reshape(nb, kc); bun = get("out_b_kun.bin", KP_DTYPE).reshape(nb, kc)
    bur = get("out_b_ur.f32", np.float32).reshape(nb, kc); bkd = get("out_b_kd.f32", np.float32).reshape(nb, kc)
    blun = get("out_b_lun.bin", KL_DTYPE).reshape(nb, lc); blds = get("out_b_lds.f32", np.float32).reshape(nb, lc)
    for f in range(nb):
        df = orc.depth_to_float(np.ascontiguousarray(d16[f]), np.float32(1.0 / 5000.0))
        un, ur, kd = orc.frame_tail(kps, df, camv[:9], float(camv[9]))
        assert all(np.array_equal(bk[f, :len(kps)][n].view(np.uint32), kps[n].view(np.uint32)) for n in KP_DTYPE.names)
        assert all(np.array_equal(bun[f, :len(kps)][n].view(np.uint32), un[n].view(np.uint32)) for n in KP_DTYPE.names)
        assert np.array_equal(bur[f, :len(kps)].view(np.uint32), ur.view(np.uint32)) and np.array_equal(bkd[f, :len(kps)].view(np.uint32), kd.view(np.uint32))
        lun, urs, ure, ds, de = orc.line_tail(kl, df, camv[:9], float(camv[9]))
        assert all(np.array_equal(blun[f, :len(kl)][n].view(np.uint32), lun[n].view(np.uint32)) for n in KL_DTYPE.names)
        assert np.array_equal(blds[f, :len(kl)].view(np.uint32), ds.view(np.uint32))
