"""Host-side mirror of the tracking matchers (ORB_SLAM2::ORBmatcher include/ORBmatcher.h:44,61,78 and
ORB_SLAM2::LSDmatcher include/LSDmatcher.h:32,40,43) over the C ABI.  All feature arrays are device
tensors (torch, cuda); nothing is computed on the host."""
import ctypes as C

import numpy as np

from . import _lib as L

TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30


def DescriptorDistance(a, b):
    """static int ORBmatcher::DescriptorDistance(const Mat&, const Mat&) -- two 32-byte host descriptors"""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return L.lib().plf_hamming256(L.vp(a), L.vp(b))


class Matcher:
    def __init__(self, max_keypoints=4096, max_mappoints=16384, max_lines=1024, max_batch=1, device=0):
        self._h = C.c_void_p()
        L.check(L.lib().plf_matcher_create(device, max_keypoints, max_mappoints, max_lines, max_batch, C.byref(self._h)), "plf_matcher_create")

    def close(self):
        if self._h:
            L.lib().plf_matcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def frame_view(n, keys_un, desc, scale_factors, bounds, uright=None, n_device=None):
        v = L.FrameView()
        v.n = int(n); v.n_device = L.vp(n_device).value if n_device is not None else None
        v.keys_un = L.vp(keys_un).value; v.uright = L.vp(uright).value if uright is not None else None
        v.desc = L.vp(desc).value
        v.min_x, v.min_y, v.max_x, v.max_y = bounds
        v.scale_factors = L.vp(scale_factors).value; v.nlevels = int(scale_factors.shape[0])
        return v

    @staticmethod
    def view_array(views, ctype=None):
        """a list of views as one ctypes array (build it once and pass it to the batched calls instead of the list)"""
        if isinstance(views, C.Array):
            return views
        ctype = ctype or type(views[0])
        return (ctype * len(views))(*views)

    @staticmethod
    def last_view(last):
        lv = L.LastFrameView()
        lv.n = int(last["mp_desc"].shape[0])
        lv.has_mappoint = L.vp(last["has_mappoint"]).value; lv.outlier = L.vp(last["outlier"]).value
        lv.world_pos = L.vp(last["world_pos"]).value; lv.keys = L.vp(last["keys"]).value; lv.mp_desc = L.vp(last["mp_desc"]).value
        lv.obs_positive = L.vp(last["obs_positive"]).value if last.get("obs_positive") is not None else None
        return lv

    @staticmethod
    def pose_pair(pose):
        pp = L.PosePair()
        for name in ("Rcw", "tcw", "Rlw", "tlw"):
            getattr(pp, name)[:] = np.asarray(pose[name], np.float32).ravel().tolist()
        for name in ("fx", "fy", "cx", "cy", "bf", "b"):
            setattr(pp, name, float(pose[name]))
        return pp

    def SearchByProjectionLastFrameBatch(self, frames, last, poses, th, mono, check_ori, match_of_kp, kp_stride, nmatches, stream=None):
        """ORBmatcher::SearchByProjection(Cur, Last, th, bMono) for a batch of current frames (list / array of FrameView) against one last frame;
        poses: list of pose dicts, or a prebuilt (PosePair * n) array"""
        arr = self.view_array(frames, L.FrameView)
        pa = poses if isinstance(poses, C.Array) else (L.PosePair * len(poses))(*[self.pose_pair(p) for p in poses])
        lv = last if isinstance(last, L.LastFrameView) else self.last_view(last)
        L.check(L.lib().plf_match_project_lastframe_batch(self._h, arr, len(arr), C.byref(lv), pa, C.c_float(th), int(mono), int(check_ori),
                                                          L.vp(match_of_kp), int(kp_stride), L.vp(nmatches), C.c_void_p(stream) if stream else None),
                "plf_match_project_lastframe_batch")

    def SearchLinesLastFrameBatch(self, last_desc, last_has_mapline, frames, match_of_line, line_stride, nmatches, stream=None):
        """LSDmatcher::SearchByProjection(Cur, Last) (BF kNN + MAD rule) for a batch of current frames (LineFrameView list / array)"""
        arr = self.view_array(frames, L.LineFrameView)
        L.check(L.lib().plf_match_lines_lastframe_batch(self._h, L.vp(last_desc), int(last_desc.shape[0]), L.vp(last_has_mapline), arr, len(arr),
                                                        L.vp(match_of_line), int(line_stride), L.vp(nmatches), C.c_void_p(stream) if stream else None),
                "plf_match_lines_lastframe_batch")

    def SearchByProjection(self, frames, mp, th, nnratio, match_of_kp, kp_stride, nmatches, stream=None):
        """frames: list of FrameView; mp: dict of device tensors (proj_x, proj_y, proj_xr, level, view_cos, in_view, desc[, obs_positive])"""
        arr = self.view_array(frames, L.FrameView)
        m = L.MapPointView()
        m.m = int(mp["desc"].shape[0])
        for k in ("proj_x", "proj_y", "proj_xr", "level", "view_cos", "in_view", "desc"):
            setattr(m, k, L.vp(mp[k]).value)
        m.obs_positive = L.vp(mp["obs_positive"]).value if mp.get("obs_positive") is not None else None
        L.check(L.lib().plf_match_project_points(self._h, arr, len(arr), C.byref(m), C.c_float(th), C.c_float(nnratio), L.vp(match_of_kp),
                                                 int(kp_stride), L.vp(nmatches), C.c_void_p(stream) if stream else None),
                "plf_match_project_points")

    def SearchByProjectionLastFrame(self, cur, last, pose, th, mono, check_ori, match_of_kp, nmatches, stream=None):
        lv = L.LastFrameView()
        lv.n = int(last["mp_desc"].shape[0])
        lv.has_mappoint = L.vp(last["has_mappoint"]).value; lv.outlier = L.vp(last["outlier"]).value
        lv.world_pos = L.vp(last["world_pos"]).value; lv.keys = L.vp(last["keys"]).value; lv.mp_desc = L.vp(last["mp_desc"]).value
        lv.obs_positive = L.vp(last["obs_positive"]).value if last.get("obs_positive") is not None else None
        pp = L.PosePair()
        for name in ("Rcw", "tcw", "Rlw", "tlw"):
            a = np.asarray(pose[name], np.float32).ravel()
            getattr(pp, name)[:] = a.tolist()
        for name in ("fx", "fy", "cx", "cy", "bf", "b"):
            setattr(pp, name, float(pose[name]))
        L.check(L.lib().plf_match_project_lastframe(self._h, C.byref(cur), C.byref(lv), C.byref(pp), C.c_float(th), int(mono), int(check_ori),
                                                    L.vp(match_of_kp), L.vp(nmatches), C.c_void_p(stream) if stream else None),
                "plf_match_project_lastframe")

    def SearchByProjectionKeyFrame(self, cur, kf, pose, log_scale_factor, th, orb_dist, check_ori, match_of_kp, nmatches, stream=None):
        """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (relocalisation); kf: device tensors keys, valid,
        world_pos, min_dist, max_dist, mp_desc"""
        lv = L.LastFrameView()
        lv.n = int(kf["mp_desc"].shape[0])
        lv.has_mappoint = L.vp(kf["valid"]).value; lv.outlier = None
        lv.world_pos = L.vp(kf["world_pos"]).value; lv.keys = L.vp(kf["keys"]).value; lv.mp_desc = L.vp(kf["mp_desc"]).value
        pp = L.PosePair()
        for name in ("Rcw", "tcw"):
            getattr(pp, name)[:] = np.asarray(pose[name], np.float32).ravel().tolist()
        for name in ("fx", "fy", "cx", "cy"):
            setattr(pp, name, float(pose[name]))
        L.check(L.lib().plf_match_project_keyframe(self._h, C.byref(cur), C.byref(lv), L.vp(kf["min_dist"]), L.vp(kf["max_dist"]), C.byref(pp),
                                                   C.c_float(log_scale_factor), C.c_float(th), int(orb_dist), int(check_ori), L.vp(match_of_kp),
                                                   L.vp(nmatches), C.c_void_p(stream) if stream else None), "plf_match_project_keyframe")

    @staticmethod
    def points3d_view(pts):
        """pts: dict of device tensors world_pos (m,3) f32, normal (m,3) f32, min_dist, max_dist f32, desc (m,32) u8, valid u8"""
        v = L.Points3DView()
        v.m = int(pts["desc"].shape[0])
        v.world_pos = L.vp(pts["world_pos"]).value; v.normal = L.vp(pts["normal"]).value if pts.get("normal") is not None else None
        v.min_distance = L.vp(pts["min_dist"]).value; v.max_distance = L.vp(pts["max_dist"]).value
        v.desc = L.vp(pts["desc"]).value; v.valid = L.vp(pts["valid"]).value
        return v

    @staticmethod
    def kf_pose(pose):
        """pose: Rcw (3,3), tcw, Ow, fx, fy, cx, cy, bf, log_scale_factor, inv_sigma2 (host values); returns (struct, keep-alive)"""
        kp = L.KfPose()
        for name in ("Rcw", "tcw", "Ow"):
            getattr(kp, name)[:] = np.asarray(pose[name], np.float32).ravel().tolist()
        for name in ("fx", "fy", "cx", "cy", "bf", "log_scale_factor"):
            setattr(kp, name, float(pose[name]))
        isg = np.ascontiguousarray(pose["inv_sigma2"], np.float32)
        kp.inv_level_sigma2 = isg.ctypes.data
        return kp, isg

    def Fuse(self, kf, pose, pts, th, best_idx, nfused, stream=None):
        """ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th), search half: best_idx[i] = key point of kf fused with map point i"""
        kp, keep = self.kf_pose(pose)
        pv = self.points3d_view(pts)
        L.check(L.lib().plf_match_fuse(self._h, C.byref(kf), C.byref(kp), C.byref(pv), C.c_float(th), L.vp(best_idx), L.vp(nfused),
                                       C.c_void_p(stream) if stream else None), "plf_match_fuse")

    def _intr(self, intr):
        pose = dict(Rcw=np.eye(3, dtype=np.float32), tcw=np.zeros(3, np.float32), Ow=np.zeros(3, np.float32), inv_sigma2=np.ones(1, np.float32),
                    bf=intr.get("bf", 0.0), **{k: intr[k] for k in ("fx", "fy", "cx", "cy", "log_scale_factor")})
        return self.kf_pose(pose)

    def FuseSim3(self, kf, Scw, intr, pts, th, best_idx, nfused, stream=None):
        """ORBmatcher::Fuse(KeyFrame*, Scw, points, th, vpReplacePoint), search half"""
        kp, keep = self._intr(intr)
        S = np.ascontiguousarray(Scw, np.float32)
        pv = self.points3d_view(pts)
        L.check(L.lib().plf_match_fuse_sim3(self._h, C.byref(kf), L.vp(S), C.byref(kp), C.byref(pv), C.c_float(th), L.vp(best_idx), L.vp(nfused),
                                            C.c_void_p(stream) if stream else None), "plf_match_fuse_sim3")

    def SearchByProjectionSim3(self, kf, Scw, intr, pts, th, match_of_kp, nmatches, stream=None):
        """ORBmatcher::SearchByProjection(KeyFrame*, Scw, points, vpMatched, int th)"""
        kp, keep = self._intr(intr)
        S = np.ascontiguousarray(Scw, np.float32)
        pv = self.points3d_view(pts)
        L.check(L.lib().plf_match_project_sim3(self._h, C.byref(kf), L.vp(S), C.byref(kp), C.byref(pv), int(th), L.vp(match_of_kp), L.vp(nmatches),
                                               C.c_void_p(stream) if stream else None), "plf_match_project_sim3")

    def SearchBySim3(self, kf1, kf2, pose1, pose2, s12, R12, t12, th, pts1, pts2, match12, nfound, stream=None):
        """ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)"""
        p1, k1 = self.kf_pose(pose1); p2, k2 = self.kf_pose(pose2)
        R = np.ascontiguousarray(R12, np.float32); t = np.ascontiguousarray(t12, np.float32)
        v1 = self.points3d_view(pts1); v2 = self.points3d_view(pts2)
        L.check(L.lib().plf_match_sim3(self._h, C.byref(kf1), C.byref(kf2), C.byref(p1), C.byref(p2), C.c_float(s12), L.vp(R), L.vp(t), C.c_float(th),
                                       C.byref(v1), C.byref(v2), L.vp(match12), L.vp(nfound), C.c_void_p(stream) if stream else None), "plf_match_sim3")

    def SearchForTriangulation(self, kf1, kf2, F12, Cw1, pose2, only_stereo, check_ori, match12, nmatches, stream=None):
        """ORBmatcher::SearchForTriangulation; kf1 / kf2: dicts of device tensors keys, uright, desc, has_mp, nodes=(node_id, node_start, feat);
        kf2 also scale_factors, level_sigma2"""
        v = L.TriView()
        v.n1 = int(kf1["desc"].shape[0]); v.n2 = int(kf2["desc"].shape[0])
        for sfx, kf in (("1", kf1), ("2", kf2)):
            setattr(v, "keys" + sfx, L.vp(kf["keys"]).value); setattr(v, "uright" + sfx, L.vp(kf["uright"]).value)
            setattr(v, "desc" + sfx, L.vp(kf["desc"]).value); setattr(v, "has_mp" + sfx, L.vp(kf["has_mp"]).value)
            setattr(v, "nodes" + sfx, int(kf["nodes"][0].shape[0])); setattr(v, "node_id" + sfx, L.vp(kf["nodes"][0]).value)
            setattr(v, "node_start" + sfx, L.vp(kf["nodes"][1]).value); setattr(v, "feat" + sfx, L.vp(kf["nodes"][2]).value)
        v.scale_factors2 = L.vp(kf2["scale_factors"]).value; v.level_sigma2_2 = L.vp(kf2["level_sigma2"]).value
        kp, keep = self.kf_pose(pose2)
        F = np.ascontiguousarray(F12, np.float32); Cw = np.ascontiguousarray(Cw1, np.float32)
        L.check(L.lib().plf_match_triangulation(self._h, C.byref(v), L.vp(F), L.vp(Cw), C.byref(kp), int(only_stereo), int(check_ori), L.vp(match12),
                                                L.vp(nmatches), C.c_void_p(stream) if stream else None), "plf_match_triangulation")

    def AssignFeaturesToGrid(self, frame):
        """Frame::AssignFeaturesToGrid: (cell_start[64*48+1], cell_idx[n]) host arrays, cell = ix*48 + iy"""
        cs = np.zeros(64 * 48 + 1, np.int32); ci = np.zeros(max(1, int(frame.n)), np.int32)
        L.check(L.lib().plf_match_assign_grid(self._h, C.byref(frame), L.vp(cs), L.vp(ci), None), "plf_match_assign_grid")
        return cs, ci[:cs[-1]]

    def knnMatch(self, query, train):
        """cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2); device tensors in, host DMATCH array (nq,2) out"""
        nq, nt = int(query.shape[0]), int(train.shape[0])
        out = np.zeros((nq, 2), L.DMATCH_DTYPE)
        L.check(L.lib().plf_match_lines_knn(self._h, L.vp(query), nq, L.vp(train), nt, L.vp(out), L.MEM_HOST, None), "plf_match_lines_knn")
        return out

    def LineDescriptorMAD(self, ldesc1, ldesc2, want_knn=True):
        """LineSegment::LineSegmentMathch + LineDescriptorMAD (include/ExtractLineSegment.h:41,44; Frame::lineDescriptorMAD include/Frame.h:75):
        device descriptor tensors in; (knn (n1,2) DMATCH array or None, nn_mad, nn12_mad) on the host"""
        n1, n2 = int(ldesc1.shape[0]), int(ldesc2.shape[0])
        knn = np.zeros((n1, 2), L.DMATCH_DTYPE) if want_knn else None
        mad = np.zeros(2, np.float64)
        L.check(L.lib().plf_line_descriptor_mad(self._h, L.vp(ldesc1), n1, L.vp(ldesc2), n2, L.vp(knn) if want_knn else None, L.vp(mad), L.MEM_HOST, None),
                "plf_line_descriptor_mad")
        return knn, float(mad[0]), float(mad[1])

    def SearchLinesLastFrame(self, last_desc, cur_desc, last_has_mapline, match_of_line, nmatches, stream=None):
        L.check(L.lib().plf_match_lines_lastframe(self._h, L.vp(last_desc), int(last_desc.shape[0]), L.vp(cur_desc), int(cur_desc.shape[0]),
                                                  L.vp(last_has_mapline), L.vp(match_of_line), L.vp(nmatches),
                                                  C.c_void_p(stream) if stream else None), "plf_match_lines_lastframe")

    def SearchLinesForTriangulation(self, desc1, desc2, has_ml1, has_ml2, stereo1, stereo2, only_stereo, match12, nmatches, mad_factor=0.1, stream=None):
        """LSDmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo); device tensors; match12[q] = keyframe-2 line or -1"""
        L.check(L.lib().plf_match_lines_triangulation(self._h, L.vp(desc1), int(desc1.shape[0]), L.vp(desc2), int(desc2.shape[0]), L.vp(has_ml1), L.vp(has_ml2),
                                                      L.vp(stereo1), L.vp(stereo2), int(bool(only_stereo)), C.c_float(mad_factor), L.vp(match12), L.vp(nmatches),
                                                      C.c_void_p(stream) if stream else None), "plf_match_lines_triangulation")

    def FuseLines(self, kf_desc, ml_desc, valid, best_idx, nfused, stream=None):
        """LSDmatcher::Fuse(pKF, vpMapLines), search half: best_idx[i] = keyframe line fused with map line i, -1 none"""
        L.check(L.lib().plf_match_lines_fuse(self._h, L.vp(kf_desc), int(kf_desc.shape[0]), L.vp(ml_desc), L.vp(valid), int(ml_desc.shape[0]), L.vp(best_idx),
                                             L.vp(nfused), C.c_void_p(stream) if stream else None), "plf_match_lines_fuse")

    @staticmethod
    def lineframe_view(n, lines_un, desc, scale_factors, n_device=None):
        v = L.LineFrameView()
        v.n = int(n); v.n_device = L.vp(n_device).value if n_device is not None else None
        v.lines_un = L.vp(lines_un).value; v.desc = L.vp(desc).value; v.scale_factors = L.vp(scale_factors).value
        return v

    @staticmethod
    def bow_view(kf_desc, f_desc, kf_angle, f_angle, kf_has_mp, kf_nodes, f_nodes, f_has_mp=None):
        """one (keyframe, frame) pair of SearchByBoW; *_nodes = (node_id uint32[K], node_start int32[K+1], feat int32[*]) device tensors"""
        v = L.BowView()
        v.n_kf = int(kf_desc.shape[0]); v.n_f = int(f_desc.shape[0])
        v.kf_desc = L.vp(kf_desc).value; v.f_desc = L.vp(f_desc).value; v.kf_angle = L.vp(kf_angle).value; v.f_angle = L.vp(f_angle).value
        v.kf_has_mp = L.vp(kf_has_mp).value
        v.f_has_mp = L.vp(f_has_mp).value if f_has_mp is not None else None
        v.kf_nodes = int(kf_nodes[0].shape[0]); v.f_nodes = int(f_nodes[0].shape[0])
        v.kf_node_id = L.vp(kf_nodes[0]).value; v.kf_node_start = L.vp(kf_nodes[1]).value; v.kf_feat = L.vp(kf_nodes[2]).value
        v.f_node_id = L.vp(f_nodes[0]).value; v.f_node_start = L.vp(f_nodes[1]).value; v.f_feat = L.vp(f_nodes[2]).value
        return v

    def SearchByBoW(self, pairs, nnratio, check_orientation, match_of_f, stride, nmatches, stream=None):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) for a batch of (keyframe, frame) pairs"""
        arr = (L.BowView * len(pairs))(*pairs)
        L.check(L.lib().plf_match_bow(self._h, arr, len(pairs), C.c_float(nnratio), int(bool(check_orientation)), L.vp(match_of_f), int(stride),
                                      L.vp(nmatches), C.c_void_p(stream) if stream else None), "plf_match_bow")

    def SearchByBoWKeyFrames(self, pairs, nnratio, check_orientation, match12, stride, nmatches, stream=None):
        """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) for a batch of keyframe pairs"""
        arr = (L.BowView * len(pairs))(*pairs)
        L.check(L.lib().plf_match_bow_kf(self._h, arr, len(pairs), C.c_float(nnratio), int(bool(check_orientation)), L.vp(match12), int(stride),
                                         L.vp(nmatches), C.c_void_p(stream) if stream else None), "plf_match_bow_kf")

    def SearchLinesByProjection(self, frames, ml, th, nnratio, match_of_line, line_stride, nmatches, stream=None):
        arr = self.view_array(frames, L.LineFrameView)
        m = L.MapLineView()
        m.m = int(ml["desc"].shape[0])
        for k in ("x1", "y1", "x2", "y2", "level", "view_cos", "in_view", "desc"):
            setattr(m, k, L.vp(ml[k]).value)
        L.check(L.lib().plf_match_project_lines(self._h, arr, len(arr), C.byref(m), C.c_float(th), C.c_float(nnratio), L.vp(match_of_line),
                                                int(line_stride), L.vp(nmatches), C.c_void_p(stream) if stream else None),
                "plf_match_project_lines")
