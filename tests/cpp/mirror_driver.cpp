// Drives the C++ host mirror (include/plf.hpp) INCLUDING its exact-signature adapters (PLF_WITH_OPENCV, against tests/mock/) on the GPU:
// ORBextractor::operator()(InputArray, InputArray, vector<KeyPoint>&, OutputArray), LineSegment::ExtractLineSegment(Mat, vector<KeyLine>&, Mat&,
// vector<Vector3d>&), ORBmatcher::SearchByProjection x2, LSDmatcher::SearchByProjection x3, LSDmatcher::SearchForTriangulation / Fuse.
// Inputs and outputs are raw little-endian arrays in a scratch directory (argv[1]); tests/test_gpu_cpp_mirror.py writes the inputs and compares
// the outputs with the CPU oracle.  Test infrastructure only.
#include <cstdio>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstring>
#include "plf.hpp"
#include <ORB_SLAM2/mock_slam.h>

float ORB_SLAM2::Frame::fx, ORB_SLAM2::Frame::fy, ORB_SLAM2::Frame::cx, ORB_SLAM2::Frame::cy, ORB_SLAM2::Frame::mnMinX, ORB_SLAM2::Frame::mnMaxX,
    ORB_SLAM2::Frame::mnMinY, ORB_SLAM2::Frame::mnMaxY;

static std::string dir;
#define STAGE(msg) do { fprintf(stderr, "[mirror] %s\n", msg); fflush(stderr); } while (0)
template <class T> static std::vector<T> rd(const char *name)
{
    FILE *f = fopen((dir + "/" + name).c_str(), "rb");
    if (!f) { fprintf(stderr, "missing input %s\n", name); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}
template <class T> static void wr(const char *name, const T *p, size_t n)
{
    FILE *f = fopen((dir + "/" + name).c_str(), "wb");
    fwrite(p, sizeof(T), n, f);
    fclose(f);
}
struct V3 { double v[3]; double &operator[](int i) { return v[i]; } };

static void on_segv(int) { void *bt[64]; const int n = backtrace(bt, 64); backtrace_symbols_fd(bt, n, 2); _exit(139); }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    signal(SIGSEGV, on_segv);
    dir = argv[1];
    using namespace ORB_SLAM2;
    const std::vector<int32_t> dims = rd<int32_t>("dims.i32");   // w, h, nfeatures, nlines
    const int w = dims[0], h = dims[1];
    std::vector<uint8_t> img = rd<uint8_t>("image.u8");
    cv::Mat im(h, w, CV_8UC1, img.data());
    // ---- extraction through the reference signatures
    STAGE("ORBextractor");
    ORB_SLAM2_PLF::ORBextractor orb(dims[2], 1.2f, 8, 20, 7, w, h);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc, mask;
    orb(im, mask, kps, desc);
    wr("out_kps.bin", kps.data(), kps.size());
    wr("out_desc.u8", desc.data, (size_t)desc.rows * 32);
    STAGE("LineSegment");
    ORB_SLAM2_PLF::LineSegment ls(dims[3], w, h);
    std::vector<cv::line_descriptor::KeyLine> kl;
    cv::Mat ldesc;
    std::vector<V3> eq;
    ls.ExtractLineSegment(im, kl, ldesc, eq);
    wr("out_kl.bin", kl.data(), kl.size());
    wr("out_ldesc.u8", ldesc.data, (size_t)ldesc.rows * 32);
    wr("out_eq.f64", (const double *)eq.data(), eq.size() * 3);
    // ---- Frame built from the extraction
    Frame::fx = 525.f; Frame::fy = 525.f; Frame::cx = w / 2 - 0.5f; Frame::cy = h / 2 - 0.5f;
    Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = (float)w; Frame::mnMaxY = (float)h;
    Frame F;
    F.N = (int)kps.size(); F.mvKeys = kps; F.mvKeysUn = kps; F.mvuRight.assign(kps.size(), -1.f); F.mDescriptors = desc;
    F.mvpMapPoints.assign(kps.size(), nullptr); F.mvbOutlier.assign(kps.size(), false);
    F.mvScaleFactors = orb.GetScaleFactors();
    F.NL = (int)kl.size(); F.mvKeylines = kl; F.mvKeylinesUn = kl; F.mLdesc = ldesc; F.mvpMapLines.assign(kl.size(), nullptr);
    // ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
    {
        const std::vector<float> px = rd<float>("mp_proj_x.f32"), py = rd<float>("mp_proj_y.f32"), pxr = rd<float>("mp_proj_xr.f32"), vc = rd<float>("mp_view_cos.f32");
        const std::vector<int32_t> lv = rd<int32_t>("mp_level.i32");
        const std::vector<uint8_t> iv = rd<uint8_t>("mp_in_view.u8"), ob = rd<uint8_t>("mp_obs_positive.u8");
        std::vector<uint8_t> md = rd<uint8_t>("mp_desc.u8");
        const int M = (int)px.size();
        std::vector<MapPoint> pts(M);
        std::vector<MapPoint *> vp(M);
        for (int i = 0; i < M; i++) {
            pts[i].mTrackProjX = px[i]; pts[i].mTrackProjY = py[i]; pts[i].mTrackProjXR = pxr[i]; pts[i].mnTrackScaleLevel = lv[i]; pts[i].mTrackViewCos = vc[i];
            pts[i].mbTrackInView = iv[i] != 0; pts[i].nObs = ob[i] ? 2 : 0; pts[i].desc = cv::Mat(1, 32, CV_8U, &md[(size_t)i * 32]);
            vp[i] = &pts[i];
        }
    STAGE("ORBmatcher map");
        ORB_SLAM2_PLF::ORBmatcher m(0.8f, true);
        const int n = m.SearchByProjection(F, vp, 3.0f);
        std::vector<int32_t> out(kps.size() + 1, -1);
        for (size_t k = 0; k < kps.size(); k++) out[k] = F.mvpMapPoints[k] ? (int32_t)(F.mvpMapPoints[k] - pts.data()) : -1;
        out[kps.size()] = n;
        wr("out_match_map.i32", out.data(), out.size());
        // ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono): the last frame = this frame's key points with map points at the positions of "last_*"
        const std::vector<float> xw = rd<float>("last_world_pos.f32"), Tc = rd<float>("cur_Tcw.f32");
        const std::vector<uint8_t> lhas = rd<uint8_t>("last_has_mappoint.u8"), lout = rd<uint8_t>("last_outlier.u8");
        std::vector<uint8_t> ldm = rd<uint8_t>("last_mp_desc.u8");
        const int NLst = (int)lhas.size();
        Frame L;
        L.N = NLst; L.mvKeys.assign(kps.begin(), kps.begin() + NLst); L.mvKeysUn = L.mvKeys; L.mvuRight.assign(NLst, -1.f); L.mDescriptors = desc;
        L.mvbOutlier.resize(NLst); L.mvScaleFactors = F.mvScaleFactors;
        std::vector<MapPoint> lpts(NLst);
        std::vector<float> xwv = xw;
        L.mvpMapPoints.assign(NLst, nullptr);
        for (int i = 0; i < NLst; i++) {
            L.mvbOutlier[i] = lout[i] != 0;
            lpts[i].desc = cv::Mat(1, 32, CV_8U, &ldm[(size_t)i * 32]); lpts[i].pos = cv::Mat(3, 1, CV_32F, &xwv[(size_t)i * 3]); lpts[i].nObs = 1;
            if (lhas[i]) L.mvpMapPoints[i] = &lpts[i];
        }
        float Tl[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::vector<float> Tcv = Tc;
        L.mTcw = cv::Mat(4, 4, CV_32F, Tl);
        Frame Cur = F;
        Cur.mvpMapPoints.assign(kps.size(), nullptr);
        Cur.mTcw = cv::Mat(4, 4, CV_32F, Tcv.data());
    STAGE("ORBmatcher last");
        const int n2 = m.SearchByProjection(Cur, L, 7.0f, false);
        std::vector<int32_t> out2(kps.size() + 1, -1);
        for (size_t k = 0; k < kps.size(); k++) out2[k] = Cur.mvpMapPoints[k] ? (int32_t)(Cur.mvpMapPoints[k] - lpts.data()) : -1;
        out2[kps.size()] = n2;
        wr("out_match_last.i32", out2.data(), out2.size());
    }
    // ---- LSDmatcher
    {
    STAGE("LSDmatcher");
        ORB_SLAM2_PLF::LSDmatcher lm(0.8f, true);
        // (Frame&, const vector<MapLine*>&, th)
        const std::vector<float> x1 = rd<float>("ml_x1.f32"), y1 = rd<float>("ml_y1.f32"), x2 = rd<float>("ml_x2.f32"), y2 = rd<float>("ml_y2.f32"), vc = rd<float>("ml_view_cos.f32");
        const std::vector<int32_t> lv = rd<int32_t>("ml_level.i32");
        const std::vector<uint8_t> iv = rd<uint8_t>("ml_in_view.u8");
        std::vector<uint8_t> md = rd<uint8_t>("ml_desc.u8");
        const int M = (int)x1.size();
        std::vector<MapLine> mls(M);
        std::vector<MapLine *> vp(M);
        for (int i = 0; i < M; i++) {
            mls[i].mTrackProjX1 = x1[i]; mls[i].mTrackProjY1 = y1[i]; mls[i].mTrackProjX2 = x2[i]; mls[i].mTrackProjY2 = y2[i]; mls[i].mnTrackScaleLevel = lv[i];
            mls[i].mTrackViewCos = vc[i]; mls[i].mbTrackInView = iv[i] != 0; mls[i].mLDescriptor = cv::Mat(1, 32, CV_8U, &md[(size_t)i * 32]);
            vp[i] = &mls[i];
        }
        Frame Fl = F;
        const int n = lm.SearchByProjection(Fl, vp, 3.0f);
        std::vector<int32_t> out(kl.size() + 1, -1);
        for (size_t k = 0; k < kl.size(); k++) out[k] = Fl.mvpMapLines[k] ? (int32_t)(Fl.mvpMapLines[k] - mls.data()) : -1;
        out[kl.size()] = n;
        wr("out_lmatch_map.i32", out.data(), out.size());
        // (Frame &Cur, const Frame &Last): last frame's lines = "lastl_desc", a MapLine on the lines flagged in lastl_has
        std::vector<uint8_t> ld = rd<uint8_t>("lastl_desc.u8");
        const std::vector<uint8_t> lh = rd<uint8_t>("lastl_has.u8");
        const int nl = (int)lh.size();
        {   // LineSegment::LineSegmentMathch / LineDescriptorMAD / LineSegmentOverlap (include/ExtractLineSegment.h:41-47)
            ls.LineSegmentMathch(ld.data(), nl, ldesc.data, ldesc.rows);
            std::vector<int32_t> mm((size_t)nl * 4);
            for (int q = 0; q < nl; q++) for (int k = 0; k < 2; k++) { mm[4 * q + 2 * k] = ls.mvlineMatches[q][k].trainIdx; mm[4 * q + 2 * k + 1] = (int32_t)ls.mvlineMatches[q][k].distance; }
            wr("out_lsmatch.i32", mm.data(), mm.size());
            double mad[3];
            ls.LineDescriptorMAD(mad[0], mad[1]);
            mad[2] = ORB_SLAM2_PLF::LineSegment::LineSegmentOverlap(2.0, 10.0, 4.0, 7.0);
            wr("out_lsmad.f64", mad, 3);
        }
        Frame Last;
        Last.mLdesc = cv::Mat(nl, 32, CV_8U, ld.data());
        std::vector<MapLine> lml(nl);
        Last.mvpMapLines.assign(nl, nullptr);
        for (int q = 0; q < nl; q++) if (lh[q]) Last.mvpMapLines[q] = &lml[q];
        Frame Cur = F;
        Cur.mvpMapLines.assign(kl.size(), nullptr);
    STAGE("LSD last");
        const int n2 = lm.SearchByProjection(Cur, Last, 3.0f, false);
        std::vector<int32_t> out2(kl.size() + 1, -1);
        for (size_t k = 0; k < kl.size(); k++) out2[k] = Cur.mvpMapLines[k] ? (int32_t)(Cur.mvpMapLines[k] - lml.data()) : -1;
        out2[kl.size()] = n2;
        wr("out_lmatch_last.i32", out2.data(), out2.size());
        // (KeyFrame*, Frame&, vector<MapLine*>&): the same lines as a keyframe
        KeyFrame KF;
        KF.mLineDescriptors = Last.mLdesc; KF.lines = Last.mvpMapLines;
        std::vector<MapLine *> vm;
    STAGE("LSD kf");
        const int n3 = lm.SearchByProjection(&KF, F, vm);
        std::vector<int32_t> out3(kl.size() + 1, -1);
        for (size_t k = 0; k < kl.size(); k++) out3[k] = vm[k] ? (int32_t)(vm[k] - lml.data()) : -1;
        out3[kl.size()] = n3;
        wr("out_lmatch_kf.i32", out3.data(), out3.size());
        // SearchForTriangulation(pKF1 = the "last" lines, pKF2 = this frame's lines, pairs, bOnlyStereo)
        KeyFrame K2;
        K2.mLineDescriptors = ldesc; K2.lines.assign(kl.size(), nullptr);
        const std::vector<uint8_t> st1 = rd<uint8_t>("tri_stereo1.u8"), st2 = rd<uint8_t>("tri_stereo2.u8");
        KF.mvuRightLineStart.resize(nl); KF.mvuRightLineEnd.resize(nl);
        for (int q = 0; q < nl; q++) { KF.mvuRightLineStart[q] = st1[q] ? 10.f : -1.f; KF.mvuRightLineEnd[q] = 5.f; }
        K2.mvuRightLineStart.resize(kl.size()); K2.mvuRightLineEnd.assign(kl.size(), 7.f);
        for (size_t k = 0; k < kl.size(); k++) K2.mvuRightLineStart[k] = st2[k] ? 3.f : -1.f;
        std::vector<std::pair<size_t, size_t>> pairs;
    STAGE("LSD tri");
        const int n4 = lm.SearchForTriangulation(&KF, &K2, pairs, true);
        std::vector<int32_t> out4(nl + 1, -1);
        for (const auto &pr : pairs) out4[pr.first] = (int32_t)pr.second;
        out4[nl] = n4;
        wr("out_ltri.i32", out4.data(), out4.size());
        // Fuse(pKF = this frame's lines as a keyframe holding MapLines on some of them, vpMapLines = the map lines above)
        std::vector<MapLine> held(kl.size());
        const std::vector<uint8_t> kfh = rd<uint8_t>("fuse_kf_has.u8");
        for (size_t k = 0; k < kl.size(); k++) if (kfh[k]) { K2.lines[k] = &held[k]; held[k].nObs = 1 + (int)(k % 4); }
        for (int i = 0; i < M; i++) { mls[i].nObs = 1 + i % 3; if (i % 7 == 0) mls[i].bad = true; if (i % 11 == 0) mls[i].inKF.insert(&K2); }
    STAGE("LSD fuse");
        const int n5 = lm.Fuse(&K2, vp);
        STAGE("fuse returned");
        std::vector<int32_t> out5(2 * M + 1, -1);   // per map line: (keyframe line it was added to or -1, what replaced it or -1)
        for (int i = 0; i < M; i++) {
            out5[2 * i] = mls[i].lastObsIdx;
            const MapLine *rb = mls[i].replacedBy;   // a MapLine the keyframe held from the start (index k) or one an earlier iteration added (100000 + j)
            out5[2 * i + 1] = !rb ? -1 : (rb >= held.data() && rb < held.data() + held.size()) ? (int32_t)(rb - held.data()) : 100000 + (int32_t)(rb - mls.data());
        }
        out5[2 * M] = n5;
        wr("out_lfuse.i32", out5.data(), out5.size());
        std::vector<int32_t> heldrep(kl.size(), -1);
        for (size_t k = 0; k < kl.size(); k++) heldrep[k] = held[k].replacedBy ? (int32_t)(held[k].replacedBy - mls.data()) : -1;
        wr("out_lfuse_held.i32", heldrep.data(), heldrep.size());
        STAGE("outputs written");
    }
    STAGE("end of LSD block");
    // ---- plf::BatchExtractor: three RGB-D frames (the same image, shifted depth) through the product-side batch driver, 2 in flight
    {
        STAGE("BatchExtractor");
        const int nb = 3;
        std::vector<uint8_t> imgs((size_t)nb * w * h);
        for (int f = 0; f < nb; f++) std::memcpy(imgs.data() + (size_t)f * w * h, img.data(), (size_t)w * h);
        const std::vector<uint16_t> depth = rd<uint16_t>("batch_depth.u16");   // nb x h x w
        const std::vector<float> camv = rd<float>("batch_cam.f32");             // fx fy cx cy k1 k2 p1 p2 k3 bf
        plf_camera cam = {camv[0], camv[1], camv[2], camv[3], camv[4], camv[5], camv[6], camv[7], camv[8], camv[9]};
        plf::BatchExtractor bx(dims[2], dims[3], w, h, 2, {0}, PLF_FMT_GRAY8, 0, 0, true);
        plf::BatchFrames F;
        const int st = bx.extract_rgbd(imgs.data(), depth.data(), nb, w, h, w, (ptrdiff_t)w * h, w, (ptrdiff_t)w * h, cam, 1.0f / 5000.0f, F);
        if (st != PLF_OK || F.n != nb) { fprintf(stderr, "batch status %d\n", st); return 3; }
        wr("out_b_n.i32", F.N.data(), F.N.size()); wr("out_b_nl.i32", F.NL.data(), F.NL.size());
        wr("out_b_kps.bin", F.mvKeys.data(), F.mvKeys.size()); wr("out_b_kun.bin", F.mvKeysUn.data(), F.mvKeysUn.size());
        wr("out_b_ur.f32", F.mvuRight.data(), F.mvuRight.size()); wr("out_b_kd.f32", F.mvDepth.data(), F.mvDepth.size());
        wr("out_b_lun.bin", F.mvKeylinesUn.data(), F.mvKeylinesUn.size()); wr("out_b_lds.f32", F.mvDepthLineStart.data(), F.mvDepthLineStart.size());
        const int32_t caps[2] = {F.kp_capacity, F.line_capacity};
        wr("out_b_caps.i32", caps, 2);
    }
    printf("mirror driver ok: %zu key points, %zu lines\n", kps.size(), kl.size());
    return 0;
}
