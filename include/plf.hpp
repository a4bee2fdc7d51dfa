// plf.hpp -- header-only C++ host mirror of the reference classes over the C ABI (plf.h).
//
// The reference is compiled C++ and its hot path sits behind member calls, so the host side above the C ABI is
// C++ with the SAME class / method names and argument meaning:
//   ORB_SLAM2::ORBextractor   include/ORBextractor.h:44-112      -> plf::ORBextractor
//   ORB_SLAM2::LineSegment    include/ExtractLineSegment.h:29-57 -> plf::LineSegment
//   ORB_SLAM2::ORBmatcher     include/ORBmatcher.h:36-140        -> plf::ORBmatcher   (tracking overloads)
//   ORB_SLAM2::LSDmatcher     include/LSDmatcher.h:27-78         -> plf::LSDmatcher   (tracking overloads)
// OpenCV-free by default (plain structs that are bit-compatible with cv::KeyPoint / KeyLine / DMatch).  When
// OpenCV headers are available (`__has_include(<opencv2/core.hpp>)`), PLF_WITH_OPENCV adapters with the exact
// reference signatures (cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray ...) are provided as well --
// see INTEGRATION.md.  Error behaviour: the reference returns silently on an empty image and asserts on a wrong
// type; here empty -> outputs cleared, everything else -> plf::Error (never a silent fallback).
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>
#include "plf.h"

namespace plf {

struct Error : std::runtime_error {
    int status;
    Error(int st, const char *what) : std::runtime_error(std::string(what) + ": " + plf_status_string(st)), status(st) {}
};
// errors (< 0) throw; warnings (> 0: the outputs are complete -- PLF_W_SLOW) are kept for the caller to look at
inline int &last_warning() { static thread_local int w = 0; return w; }
inline void check(int st, const char *what) { last_warning() = st > 0 ? st : 0; if (st < 0) throw Error(st, what); }

// ------------------------------------------------------------------ ORBextractor
class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxWidth = 640, int maxHeight = 480,
                 int maxBatch = 1, int device = 0)
    {
        plf_orb_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, maxWidth, maxHeight, maxBatch};
        check(plf_orb_create(&p, &h_), "plf_orb_create");
        nlevels_ = nlevels; scaleFactor_ = scaleFactor;
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        check(plf_orb_get_tables(h_, nullptr, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                                 mnFeaturesPerLevel.data()), "plf_orb_get_tables");
    }
    ~ORBextractor() { plf_orb_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // void operator()(InputArray image, InputArray mask, vector<KeyPoint>& keypoints, OutputArray descriptors)
    // image: 8-bit single channel; mask is ignored (as in the reference); descriptors: keypoints.size() x 32 bytes
    void operator()(const uint8_t *image, int width, int height, ptrdiff_t pitch, std::vector<plf_keypoint> &keypoints,
                    std::vector<uint8_t> &descriptors)
    {
        const int cap = plf_orb_capacity(h_);
        keypoints.resize(cap); descriptors.resize((size_t)cap * 32);
        int32_t n = 0;
        const int st = plf_orb_extract(h_, image, width, height, pitch, keypoints.data(), descriptors.data(), cap, &n);
        if (st == PLF_E_EMPTY) { keypoints.clear(); descriptors.clear(); return; }
        check(st, "plf_orb_extract");
        keypoints.resize(n); descriptors.resize((size_t)n * 32);
    }

    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return scaleFactor_; }
    std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() const { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() const { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }

    // mvImagePyramid[level] of the last call (with its 19-px border), pitch = width + 38
    std::vector<uint8_t> ImagePyramidLevel(int level, int *w = nullptr, int *h = nullptr)
    {
        int32_t lw = 0, lh = 0;
        check(plf_orb_get_pyramid_level(h_, 0, level, nullptr, &lw, &lh), "plf_orb_get_pyramid_level");
        std::vector<uint8_t> out((size_t)(lw + 38) * (lh + 38));
        check(plf_orb_get_pyramid_level(h_, 0, level, out.data(), nullptr, nullptr), "plf_orb_get_pyramid_level");
        if (w) *w = lw;
        if (h) *h = lh;
        return out;
    }
    plf_orb *handle() { return h_; }

    std::vector<int32_t> mnFeaturesPerLevel;

private:
    plf_orb *h_ = nullptr;
    int nlevels_ = 0;
    float scaleFactor_ = 0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// ------------------------------------------------------------------ LineSegment
struct Vector3d { double v[3]; double operator()(int i) const { return v[i]; } };

class LineSegment {
public:
    // maxMs > 0: opt-in time budget of LSD region growing per call (plf_line_params.max_ms; the reference has none)
    explicit LineSegment(int nlines = 100, int maxWidth = 640, int maxHeight = 480, int maxBatch = 1, int device = 0, float maxMs = 0.f) : nlines_(nlines)
    {
        plf_line_params p = {nlines, 0, device, maxWidth, maxHeight, maxBatch, PLF_LBD_BLURRED, maxMs};
        check(plf_line_create(&p, &h_), "plf_line_create");
    }
    ~LineSegment() { plf_line_destroy(h_); }
    LineSegment(const LineSegment &) = delete;
    LineSegment &operator=(const LineSegment &) = delete;

    // void ExtractLineSegment(const Mat& img, vector<KeyLine>& keylines, Mat& ldesc, vector<Vector3d>& lineFunctions,
    //                         int scale = 1.2, int numOctaves = 1)     (scale truncates to 1, one octave)
    void ExtractLineSegment(const uint8_t *img, int width, int height, ptrdiff_t pitch, std::vector<plf_keyline> &vkeyLines,
                            std::vector<uint8_t> &ldesc, std::vector<Vector3d> &vkeylineFunctions, int scale = 1, int numOctaves = 1)
    {
        if (scale != 1 || numOctaves != 1) throw Error(PLF_E_BADARG, "ExtractLineSegment: only scale=1, numOctaves=1 (the reference's call)");
        vkeyLines.resize(nlines_); ldesc.resize((size_t)nlines_ * 32); vkeylineFunctions.resize(nlines_);
        int32_t n = 0;
        const int st = plf_line_extract(h_, img, width, height, pitch, vkeyLines.data(), ldesc.data(), &vkeylineFunctions[0].v[0], nlines_, &n);
        if (st == PLF_E_EMPTY) n = 0;
        else check(st, "plf_line_extract");
        vkeyLines.resize(n); ldesc.resize((size_t)n * 32); vkeylineFunctions.resize(n);
    }
    // void LineSegmentMathch(Mat &ldesc1, Mat &ldesc2)   include/ExtractLineSegment.h:41   (host descriptor rows, n x 32; knnMatch k = 2 into mvlineMatches)
    // void LineDescriptorMAD()                             include/ExtractLineSegment.h:44   (mnnMad, mnn12Mad of those matches; computed in the same device pass)
    void LineSegmentMathch(const uint8_t *ldesc1, int n1, const uint8_t *ldesc2, int n2, int device = 0)
    {
        mvlineMatches.clear(); mnnMad = mnn12Mad = 0.0;
        if (n1 < 1 || n2 < 2) return;
        plf_matcher *m = nullptr;
        check(plf_matcher_create(device, 64, 64, n1 > n2 ? n1 : n2, 1, &m), "plf_matcher_create");
        void *d1 = nullptr, *d2 = nullptr;
        int st = plf_device_alloc(device, (size_t)n1 * 32, &d1);
        if (st == PLF_OK) st = plf_device_alloc(device, (size_t)n2 * 32, &d2);
        if (st == PLF_OK) st = plf_upload(d1, ldesc1, (size_t)n1 * 32, nullptr);
        if (st == PLF_OK) st = plf_upload(d2, ldesc2, (size_t)n2 * 32, nullptr);
        std::vector<plf_dmatch> knn((size_t)n1 * 2);
        double mad[2] = {0.0, 0.0};
        if (st == PLF_OK) st = plf_line_descriptor_mad(m, (const uint8_t *)d1, n1, (const uint8_t *)d2, n2, knn.data(), mad, PLF_MEM_HOST, nullptr);
        plf_device_free(d1); plf_device_free(d2); plf_matcher_destroy(m);
        check(st, "plf_line_descriptor_mad");
        mvlineMatches.resize(n1);
        for (int q = 0; q < n1; q++) mvlineMatches[q] = {knn[2 * q], knn[2 * q + 1]};
        mnnMad = mad[0]; mnn12Mad = mad[1];
    }
    void LineDescriptorMAD(double &nn_mad, double &nn12_mad) const { nn_mad = mnnMad; nn12_mad = mnn12Mad; }
    // double LineSegmentOverlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj)   include/ExtractLineSegment.h:47
    static double LineSegmentOverlap(double spl_obs, double epl_obs, double spl_proj, double epl_proj)
    {
        return plf_line_segment_overlap(spl_obs, epl_obs, spl_proj, epl_proj);
    }
    plf_line *handle() { return h_; }

    std::vector<std::vector<plf_dmatch>> mvlineMatches;   // include/ExtractLineSegment.h:51
    double mnnMad = 0.0, mnn12Mad = 0.0;                    // include/ExtractLineSegment.h:52

private:
    plf_line *h_ = nullptr;
    int nlines_;
};

// ------------------------------------------------------------------ matchers (device-resident views, see plf.h)
class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    ORBmatcher(plf_matcher *m, float nnratio = 0.6f, bool checkOri = true) : m_(m), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b) { return plf_hamming256(a, b); }
    // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th = 3)
    void SearchByProjection(const plf_frame_view &F, const plf_mappoint_view &vpMapPoints, float th, int32_t *match_of_kp_dev,
                            int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_project_points(m_, &F, 1, &vpMapPoints, th, mfNNratio, match_of_kp_dev, F.n, nmatches_dev, stream), "SearchByProjection");
    }
    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    void SearchByProjection(const plf_frame_view &CurrentFrame, const plf_lastframe_view &LastFrame, const plf_pose_pair &pose, float th, bool bMono,
                            int32_t *match_of_kp_dev, int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_project_lastframe(m_, &CurrentFrame, &LastFrame, &pose, th, bMono, mbCheckOrientation, match_of_kp_dev, nmatches_dev, stream),
              "SearchByProjection(last frame)");
    }
    // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
    void SearchByProjection(const plf_frame_view &CurrentFrame, const plf_lastframe_view &pKF, const float *min_distance_dev, const float *max_distance_dev,
                            const plf_pose_pair &pose, float log_scale_factor, float th, int ORBdist, int32_t *match_of_kp_dev, int32_t *nmatches_dev,
                            void *stream = nullptr)
    {
        check(plf_match_project_keyframe(m_, &CurrentFrame, &pKF, min_distance_dev, max_distance_dev, &pose, log_scale_factor, th, ORBdist,
                                         mbCheckOrientation, match_of_kp_dev, nmatches_dev, stream), "SearchByProjection(keyframe)");
    }
    // int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)
    void SearchByBoW(const plf_bow_view &pKF_and_F, int32_t *match_of_f_dev, int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_bow(m_, &pKF_and_F, 1, mfNNratio, mbCheckOrientation, match_of_f_dev, pKF_and_F.n_f, nmatches_dev, stream), "SearchByBoW");
    }
    // int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12)
    void SearchByBoW(const plf_bow_view &pKF1_and_pKF2, int32_t *match12_dev, int32_t *nmatches_dev, bool /*keyframes*/, void *stream = nullptr)
    {
        check(plf_match_bow_kf(m_, &pKF1_and_pKF2, 1, mfNNratio, mbCheckOrientation, match12_dev, std::max(pKF1_and_pKF2.n_kf, pKF1_and_pKF2.n_f), nmatches_dev, stream),
              "SearchByBoW(keyframes)");
    }
    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs, const bool bOnlyStereo)
    void SearchForTriangulation(const plf_tri_view &pKF1_and_pKF2, const float *F12, const float *Cw1, const plf_kf_pose &pose2, bool bOnlyStereo,
                                int32_t *match12_dev, int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_triangulation(m_, &pKF1_and_pKF2, F12, Cw1, &pose2, bOnlyStereo, mbCheckOrientation, match12_dev, nmatches_dev, stream),
              "SearchForTriangulation");
    }
    // int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th = 3.0)   (search half; the caller applies Replace / AddObservation)
    void Fuse(const plf_frame_view &pKF, const plf_kf_pose &pose, const plf_points3d_view &vpMapPoints, float th, int32_t *best_idx_dev,
              int32_t *nfused_dev, void *stream = nullptr)
    {
        check(plf_match_fuse(m_, &pKF, &pose, &vpMapPoints, th, best_idx_dev, nfused_dev, stream), "Fuse");
    }
    // int Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint)
    void Fuse(const plf_frame_view &pKF, const float *Scw, const plf_kf_pose &intrinsics, const plf_points3d_view &vpPoints, float th,
              int32_t *best_idx_dev, int32_t *nfused_dev, void *stream = nullptr)
    {
        check(plf_match_fuse_sim3(m_, &pKF, Scw, &intrinsics, &vpPoints, th, best_idx_dev, nfused_dev, stream), "Fuse(Scw)");
    }
    // int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
    void SearchByProjection(const plf_frame_view &pKF, const float *Scw, const plf_kf_pose &intrinsics, const plf_points3d_view &vpPoints, int th,
                            int32_t *match_of_kp_dev, int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_project_sim3(m_, &pKF, Scw, &intrinsics, &vpPoints, th, match_of_kp_dev, nmatches_dev, stream), "SearchByProjection(Scw)");
    }
    // int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)
    void SearchBySim3(const plf_frame_view &pKF1, const plf_frame_view &pKF2, const plf_kf_pose &pose1, const plf_kf_pose &pose2, float s12, const float *R12,
                      const float *t12, float th, const plf_points3d_view &vpMapPoints1, const plf_points3d_view &vpMapPoints2, int32_t *match12_dev,
                      int32_t *nfound_dev, void *stream = nullptr)
    {
        check(plf_match_sim3(m_, &pKF1, &pKF2, &pose1, &pose2, s12, R12, t12, th, &vpMapPoints1, &vpMapPoints2, match12_dev, nfound_dev, stream), "SearchBySim3");
    }

private:
    plf_matcher *m_;
    float mfNNratio;
    bool mbCheckOrientation;
};

class LSDmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    LSDmatcher(plf_matcher *m, float nnratio = 0.6f, bool checkOri = true) : m_(m), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b) { return plf_hamming256(a, b); }
    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)   (BF kNN + MAD rule)
    void SearchByProjection(const uint8_t *last_desc_dev, int nlast, const uint8_t *cur_desc_dev, int ncur, const uint8_t *last_has_mapline_dev,
                            int32_t *match_of_line_dev, int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_lines_lastframe(m_, last_desc_dev, nlast, cur_desc_dev, ncur, last_has_mapline_dev, match_of_line_dev, nmatches_dev, stream),
              "LSDmatcher::SearchByProjection(last frame)");
    }
    // int SearchByProjection(Frame &F, const vector<MapLine*> &vpMapLines, const float th = 3)
    void SearchByProjection(const plf_lineframe_view &F, const plf_mapline_view &vpMapLines, float th, int32_t *match_of_line_dev,
                            int32_t *nmatches_dev, void *stream = nullptr)
    {
        check(plf_match_project_lines(m_, &F, 1, &vpMapLines, th, mfNNratio, match_of_line_dev, F.n, nmatches_dev, stream), "LSDmatcher::SearchByProjection");
    }

    // int SearchByProjection(KeyFrame *pKF, Frame &F, vector<MapLine*> &vpMapLineMatches): the same rule, the keyframe's lines on the query side
    void SearchByProjection(const uint8_t *kf_desc_dev, int nkf, const uint8_t *f_desc_dev, int nf, const uint8_t *kf_has_mapline_dev, int32_t *match_of_line_dev,
                            int32_t *nmatches_dev, bool /*keyframe*/, void *stream = nullptr)
    {
        check(plf_match_lines_lastframe(m_, kf_desc_dev, nkf, f_desc_dev, nf, kf_has_mapline_dev, match_of_line_dev, nmatches_dev, stream),
              "LSDmatcher::SearchByProjection(keyframe)");
    }
    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo)
    void SearchForTriangulation(const uint8_t *desc1_dev, int n1, const uint8_t *desc2_dev, int n2, const uint8_t *has_ml1_dev, const uint8_t *has_ml2_dev,
                                const uint8_t *stereo1_dev, const uint8_t *stereo2_dev, bool bOnlyStereo, int32_t *match12_dev, int32_t *nmatches_dev,
                                float mad_factor = 0.1f, void *stream = nullptr)
    {
        check(plf_match_lines_triangulation(m_, desc1_dev, n1, desc2_dev, n2, has_ml1_dev, has_ml2_dev, stereo1_dev, stereo2_dev, bOnlyStereo, mad_factor,
                                            match12_dev, nmatches_dev, stream), "LSDmatcher::SearchForTriangulation");
    }
    // int Fuse(KeyFrame *pKF, const vector<MapLine*> &vpMapLines)   (search half; the caller applies Replace / AddObservation / AddMapLine)
    void Fuse(const uint8_t *kf_desc_dev, int nkf, const uint8_t *ml_desc_dev, const uint8_t *valid_dev, int m, int32_t *best_idx_dev, int32_t *nfused_dev,
              void *stream = nullptr)
    {
        check(plf_match_lines_fuse(m_, kf_desc_dev, nkf, ml_desc_dev, valid_dev, m, best_idx_dev, nfused_dev, stream), "LSDmatcher::Fuse");
    }

private:
    plf_matcher *m_;
    float mfNNratio;
    bool mbCheckOrientation;
};

// RAII device array for host code that stages data for the matchers (plf_device_alloc / plf_upload / plf_download)
template <class T> class DeviceArray {
public:
    DeviceArray() = default;
    DeviceArray(size_t n, int device = 0) { reset(n, device); }
    template <class U> DeviceArray(const std::vector<U> &host, int device = 0) { static_assert(sizeof(U) == sizeof(T), "element size"); reset(host.size(), device); upload(host.data(), host.size()); }
    ~DeviceArray() { plf_device_free(p_); }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    void reset(size_t n, int device = 0) { plf_device_free(p_); p_ = nullptr; n_ = n; void *q = nullptr; check(plf_device_alloc(device, n * sizeof(T), &q), "plf_device_alloc"); p_ = (T *)q; }
    void upload(const void *host, size_t n) { check(plf_upload(p_, host, n * sizeof(T), nullptr), "plf_upload"); }
    void fill(int byte) { check(plf_fill(p_, byte, n_ * sizeof(T), nullptr), "plf_fill"); }
    std::vector<T> download() const { std::vector<T> v(n_); check(plf_download(v.data(), p_, n_ * sizeof(T), nullptr), "plf_download"); return v; }
    T *get() const { return p_; }
    size_t size() const { return n_; }
private:
    T *p_ = nullptr;
    size_t n_ = 0;
};

// ORB_SLAM2::Frame members either side of the extractor / matcher path (include/Frame.h), stateless: device pointers in and out, one frame.
// Names and roles follow the reference; the reference works on its own member vectors, these take the same data as arrays.
struct Frame {
    // void Frame::UndistortKeyPoints()  include/Frame.h (so@0xf8630): mvKeys -> mvKeysUn
    static void UndistortKeyPoints(const plf_keypoint *mvKeys_dev, int N, const plf_camera &cam, plf_keypoint *mvKeysUn_dev, int device = 0,
                                   void *stream = nullptr)
    {
        check(plf_frame_tail(mvKeys_dev, nullptr, N, 1, N, nullptr, 0, 0, &cam, mvKeysUn_dev, nullptr, nullptr, device, stream), "Frame::UndistortKeyPoints");
    }
    // void Frame::ComputeStereoFromRGBD(const cv::Mat &imDepth) (so@0xf6860): mvuRight, mvDepth (also rewrites mvKeysUn: same values)
    static void ComputeStereoFromRGBD(const plf_keypoint *mvKeys_dev, int N, const float *imDepth_dev, int width, int height, const plf_camera &cam,
                                      plf_keypoint *mvKeysUn_dev, float *mvuRight_dev, float *mvDepth_dev, int device = 0, void *stream = nullptr)
    {
        check(plf_frame_tail(mvKeys_dev, nullptr, N, 1, N, imDepth_dev, width, height, &cam, mvKeysUn_dev, mvuRight_dev, mvDepth_dev, device, stream),
              "Frame::ComputeStereoFromRGBD");
    }
    // void Frame::UndistortKeyLines() include/Frame.h:267 + mvuRightLineStart/End, mvDepthLineStart/End (:208-211)
    static void UndistortKeyLines(const plf_keyline *mvKeylines_dev, int NL, const float *imDepth_dev, int width, int height, const plf_camera &cam,
                                  plf_keyline *mvKeylinesUn_dev, float *mvuRightLineStart_dev, float *mvuRightLineEnd_dev, float *mvDepthLineStart_dev,
                                  float *mvDepthLineEnd_dev, int device = 0, void *stream = nullptr)
    {
        check(plf_frame_line_tail(mvKeylines_dev, nullptr, NL, 1, NL, imDepth_dev, width, height, &cam, mvKeylinesUn_dev, mvuRightLineStart_dev,
                                  mvuRightLineEnd_dev, mvDepthLineStart_dev, mvDepthLineEnd_dev, device, stream), "Frame::UndistortKeyLines");
    }
    // bool Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit) include/Frame.h:104, for M map points at once -> the plf_mappoint_view fields
    static void isInFrustum(const float *world_pos_dev, const float *normal_dev, const float *min_distance_dev, const float *max_distance_dev, int M,
                            const plf_frustum_pose &pose, const plf_camera &cam, const float bounds[4], float mfLogScaleFactor, int mnScaleLevels,
                            float viewingCosLimit, float *mTrackProjX_dev, float *mTrackProjY_dev, float *mTrackProjXR_dev, int32_t *mnTrackScaleLevel_dev,
                            float *mTrackViewCos_dev, uint8_t *mbTrackInView_dev, int device = 0, void *stream = nullptr)
    {
        check(plf_frustum_points(world_pos_dev, normal_dev, min_distance_dev, max_distance_dev, M, &pose, &cam, bounds[0], bounds[1], bounds[2], bounds[3],
                                 mfLogScaleFactor, mnScaleLevels, viewingCosLimit, mTrackProjX_dev, mTrackProjY_dev, mTrackProjXR_dev,
                                 mnTrackScaleLevel_dev, mTrackViewCos_dev, mbTrackInView_dev, device, stream), "Frame::isInFrustum(MapPoint)");
    }
    // bool Frame::isInFrustum(MapLine *pML, float viewingCosLimit) include/Frame.h:107, for M map lines (world_pos6: start xyz, end xyz)
    static void isInFrustum(const float *world_pos6_dev, const float *normal_dev, const float *min_distance_dev, const float *max_distance_dev, int M,
                            const plf_frustum_pose &pose, const plf_camera &cam, const float bounds[4], float mfLogScaleFactor, int mnScaleLevels,
                            float viewingCosLimit, float *mTrackProjX1_dev, float *mTrackProjY1_dev, float *mTrackProjX1R_dev, float *mTrackProjX2_dev,
                            float *mTrackProjY2_dev, float *mTrackProjX2R_dev, int32_t *mnTrackScaleLevel_dev, float *mTrackViewCos_dev,
                            uint8_t *mbTrackInView_dev, int device = 0, void *stream = nullptr)
    {
        check(plf_frustum_lines(world_pos6_dev, normal_dev, min_distance_dev, max_distance_dev, M, &pose, &cam, bounds[0], bounds[1], bounds[2], bounds[3],
                                mfLogScaleFactor, mnScaleLevels, viewingCosLimit, mTrackProjX1_dev, mTrackProjY1_dev, mTrackProjX1R_dev, mTrackProjX2_dev,
                                mTrackProjY2_dev, mTrackProjX2R_dev, mnTrackScaleLevel_dev, mTrackViewCos_dev, mbTrackInView_dev, device, stream),
              "Frame::isInFrustum(MapLine)");
    }
};

// ------------------------------------------------------------------ batches of independent host frames on all GPUs of the node
// The caller loop of the reference (Examples/RGB-D/rgbd_tum.cc:84-128 -> System::TrackRGBD -> Tracking::GrabImageRGBD -> RGB-D Frame::Frame,
// include/Frame.h:60) for N frames at once: plf_batch_* shards the frames over the GPUs in contiguous blocks (no collective), stages them through
// pinned double buffers and returns every Frame member the front-end produces.  Frames are independent: this is the offline / dataset-replay /
// multi-camera mode (BASELINE configs 3-4); the live loop uses ORBextractor / LineSegment above.
struct BatchFrames {   // outputs of one call: frame f at [f * kp_capacity, ...) / [f * line_capacity, ...)
    int n = 0, kp_capacity = 0, line_capacity = 0;
    std::vector<plf_keypoint> mvKeys, mvKeysUn; std::vector<uint8_t> mDescriptors; std::vector<int32_t> N;
    std::vector<float> mvuRight, mvDepth;
    std::vector<plf_keyline> mvKeylines, mvKeylinesUn; std::vector<uint8_t> mLdesc; std::vector<double> mvKeyLineFunctions; std::vector<int32_t> NL;
    std::vector<float> mvuRightLineStart, mvuRightLineEnd, mvDepthLineStart, mvDepthLineEnd;
    std::vector<int32_t> match_of_kp, n_kp_matches, match_of_line, n_line_matches;   // against the local map of set_local_map (-1 = none)
};

class BatchExtractor {
public:
    // devices empty = every visible GPU.  rgbd = the Frame-tail buffers of extract_rgbd.
    BatchExtractor(int nfeatures, int nlines, int width, int height, int frames_in_flight = 8, const std::vector<int> &devices = {},
                   int input_format = PLF_FMT_GRAY8, int max_mappoints = 0, int max_maplines = 0, bool rgbd = false, float scaleFactor = 1.2f, int nlevels = 8,
                   int iniThFAST = 20, int minThFAST = 7)
        : nfeatures_(nfeatures), nlines_(nlines), kp_cap_(nfeatures > 0 ? nfeatures + 4 * nlevels : 0)
    {
        plf_batch_params p;
        std::memset(&p, 0, sizeof(p));
        p.orb.nfeatures = nfeatures; p.orb.scale_factor = scaleFactor; p.orb.nlevels = nlevels; p.orb.ini_th_fast = iniThFAST; p.orb.min_th_fast = minThFAST;
        p.orb.max_width = width; p.orb.max_height = height; p.orb.max_batch = 1;
        p.line.nlines = nlines; p.line.max_width = width; p.line.max_height = height; p.line.max_batch = 1; p.line.lbd_sobel_input = PLF_LBD_BLURRED;
        std::vector<int32_t> dv(devices.begin(), devices.end());
        p.n_devices = (int32_t)dv.size(); p.devices = dv.empty() ? nullptr : dv.data();
        p.frames_in_flight = frames_in_flight; p.input_format = input_format; p.max_mappoints = max_mappoints; p.max_maplines = max_maplines; p.rgbd = rgbd ? 1 : 0;
        check(plf_batch_create(&p, &b_), "plf_batch_create");
    }
    ~BatchExtractor() { plf_batch_destroy(b_); }
    BatchExtractor(const BatchExtractor &) = delete;
    BatchExtractor &operator=(const BatchExtractor &) = delete;

    int device_count() const { return plf_batch_device_count(b_); }
    // host arrays of the tracking fields of the local map; replicated on every GPU (Tracking::SearchLocalPoints / SearchLocalLines feed)
    void set_local_map(const plf_mappoint_view *points, const plf_mapline_view *lines, float th, float nnratio, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY)
    {
        check(plf_batch_set_local_map(b_, points, lines, th, nnratio, mnMinX, mnMinY, mnMaxX, mnMaxY), "plf_batch_set_local_map");
        has_map_ = (points && points->m > 0) || (lines && lines->m > 0);
    }
    // gray / RGB / BGR frames (the format given to the constructor); returns PLF_OK or PLF_E_CAPACITY (outputs truncated), throws otherwise
    int extract(const uint8_t *images, int64_t n_frames, int width, int height, ptrdiff_t pitch, ptrdiff_t frame_stride, BatchFrames &out)
    {
        return run(images, n_frames, width, height, pitch, frame_stride, nullptr, nullptr, 0, 0, 0.f, out);
    }
    // RGB-D Frame constructor per frame: depth = uint16 images (NULL: none), mDepthMapFactor = 1 / DepthMapFactor of the settings file
    int extract_rgbd(const uint8_t *images, const uint16_t *depth, int64_t n_frames, int width, int height, ptrdiff_t pitch, ptrdiff_t frame_stride,
                     ptrdiff_t depth_pitch_elems, ptrdiff_t depth_frame_stride_elems, const plf_camera &cam, float mDepthMapFactor, BatchFrames &out)
    {
        return run(images, n_frames, width, height, pitch, frame_stride, &cam, depth, depth_pitch_elems, depth_frame_stride_elems, mDepthMapFactor, out);
    }
    plf_batch *handle() { return b_; }

private:
    int run(const uint8_t *images, int64_t n, int width, int height, ptrdiff_t pitch, ptrdiff_t frame_stride, const plf_camera *cam, const uint16_t *depth,
            ptrdiff_t dpitch, ptrdiff_t dstride, float factor, BatchFrames &o)
    {
        const size_t K = (size_t)n * kp_cap_, Lc = (size_t)n * nlines_;
        o.n = (int)n; o.kp_capacity = kp_cap_; o.line_capacity = nlines_;
        plf_batch_outputs O;
        std::memset(&O, 0, sizeof(O));
        if (nfeatures_ > 0) {
            o.mvKeys.assign(K, plf_keypoint()); o.mDescriptors.assign(K * 32, 0); o.N.assign(n, 0);
            O.kps = o.mvKeys.data(); O.desc = o.mDescriptors.data(); O.n_kps = o.N.data(); O.kp_capacity = kp_cap_;
            if (has_map_) { o.match_of_kp.assign(K, -1); o.n_kp_matches.assign(n, 0); O.match_of_kp = o.match_of_kp.data(); O.n_kp_matches = o.n_kp_matches.data(); }
        }
        if (nlines_ > 0) {
            o.mvKeylines.assign(Lc, plf_keyline()); o.mLdesc.assign(Lc * 32, 0); o.mvKeyLineFunctions.assign(Lc * 3, 0.0); o.NL.assign(n, 0);
            O.lines = o.mvKeylines.data(); O.ldesc = o.mLdesc.data(); O.line_eq = o.mvKeyLineFunctions.data(); O.n_lines = o.NL.data(); O.line_capacity = nlines_;
            if (has_map_) { o.match_of_line.assign(Lc, -1); o.n_line_matches.assign(n, 0); O.match_of_line = o.match_of_line.data(); O.n_line_matches = o.n_line_matches.data(); }
        }
        int st;
        if (!cam) st = plf_batch_extract(b_, images, n, width, height, pitch, frame_stride, &O);
        else {
            plf_batch_rgbd R;
            std::memset(&R, 0, sizeof(R));
            R.cam = *cam; R.depth_factor = factor; R.depth = depth; R.depth_pitch_elems = dpitch; R.depth_frame_stride_elems = dstride;
            if (nfeatures_ > 0) {
                o.mvKeysUn.assign(K, plf_keypoint()); o.mvuRight.assign(K, -1.f); o.mvDepth.assign(K, -1.f);
                R.kps_un = o.mvKeysUn.data(); R.uright = o.mvuRight.data(); R.kp_depth = o.mvDepth.data();
            }
            if (nlines_ > 0) {
                o.mvKeylinesUn.assign(Lc, plf_keyline());
                o.mvuRightLineStart.assign(Lc, -1.f); o.mvuRightLineEnd.assign(Lc, -1.f); o.mvDepthLineStart.assign(Lc, -1.f); o.mvDepthLineEnd.assign(Lc, -1.f);
                R.lines_un = o.mvKeylinesUn.data(); R.uright_start = o.mvuRightLineStart.data(); R.uright_end = o.mvuRightLineEnd.data();
                R.depth_start = o.mvDepthLineStart.data(); R.depth_end = o.mvDepthLineEnd.data();
            }
            st = plf_batch_extract_rgbd(b_, images, n, width, height, pitch, frame_stride, &O, &R);
        }
        if (st != PLF_OK && st != PLF_E_CAPACITY && st != PLF_E_EMPTY) check(st, "plf_batch_extract");
        return st;
    }
    plf_batch *b_ = nullptr;
    int nfeatures_, nlines_, kp_cap_;
    bool has_map_ = false;
};

}  // namespace plf

// ---------------------------------------------------------------------------------------------------------------
// Drop-in adapters with the EXACT reference signatures; compiled only where OpenCV (+ contrib line_descriptor) headers exist
// (tests/mock/ holds a minimal stand-in so that tests/test_abi.py can at least compile them in this image).
// The matcher adapters are templates over the reference's own Frame / KeyFrame / MapPoint / MapLine classes (include/Frame.h,
// KeyFrame.h, MapPoint.h, MapLine.h): they read exactly the members the reference bodies read, flatten them into the plf_*_view
// structs, run the device search and write the result back into mvpMapPoints / mvpMapLines -- the state the reference loop leaves.
// ---------------------------------------------------------------------------------------------------------------
#if defined(PLF_WITH_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <cstring>
#include <utility>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
namespace ORB_SLAM2_PLF {
static_assert(sizeof(cv::KeyPoint) == sizeof(plf_keypoint), "cv::KeyPoint must be 28 bytes");
static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(plf_keyline), "cv::line_descriptor::KeyLine must be 68 bytes");

class ORBextractor : public plf::ORBextractor {
public:
    using plf::ORBextractor::ORBextractor;
    using plf::ORBextractor::operator();
    // include/ORBextractor.h:59-61
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors)
    {
        if (image.empty()) return;  // so@0x76dda
        cv::Mat im = image.getMat();
        CV_Assert(im.type() == CV_8UC1);
        std::vector<plf_keypoint> k;
        std::vector<uint8_t> d;
        plf::ORBextractor::operator()(im.data, im.cols, im.rows, (ptrdiff_t)im.step, k, d);
        keypoints.resize(k.size());
        if (!k.empty()) memcpy((void *)keypoints.data(), k.data(), k.size() * sizeof(plf_keypoint));
        if (k.empty()) { descriptors.release(); return; }
        descriptors.create((int)k.size(), 32, CV_8U);
        memcpy(descriptors.getMat().data, d.data(), d.size());
    }
};

// include/ExtractLineSegment.h:29-57.  Vector3dT = Eigen::Vector3d in the reference (any type with operator()(int) or [] assignable from double works).
class LineSegment : public plf::LineSegment {
public:
    using plf::LineSegment::LineSegment;
    using plf::LineSegment::ExtractLineSegment;
    // void ExtractLineSegment(const Mat &img, vector<KeyLine> &keylines, Mat &ldesc, vector<Vector3d> &keylineFunctions, int scale = 1.2, int numOctaves = 1)
    template <class Vector3dT>
    void ExtractLineSegment(const cv::Mat &img, std::vector<cv::line_descriptor::KeyLine> &keylines, cv::Mat &ldesc, std::vector<Vector3dT> &keylineFunctions,
                            int scale = 1.2, int numOctaves = 1)
    {
        keylines.clear(); keylineFunctions.clear();
        if (img.empty()) { ldesc.release(); return; }
        CV_Assert(img.type() == CV_8UC1);
        std::vector<plf_keyline> kl;
        std::vector<uint8_t> d;
        std::vector<plf::Vector3d> eq;
        plf::LineSegment::ExtractLineSegment(img.data, img.cols, img.rows, (ptrdiff_t)img.step, kl, d, eq, scale, numOctaves);
        keylines.resize(kl.size());
        if (!kl.empty()) memcpy((void *)keylines.data(), kl.data(), kl.size() * sizeof(plf_keyline));
        if (kl.empty()) { ldesc.release(); return; }
        ldesc.create((int)kl.size(), 32, CV_8U);
        memcpy(ldesc.data, d.data(), d.size());
        keylineFunctions.resize(eq.size());
        for (size_t i = 0; i < eq.size(); i++) { keylineFunctions[i][0] = eq[i].v[0]; keylineFunctions[i][1] = eq[i].v[1]; keylineFunctions[i][2] = eq[i].v[2]; }
    }
};

namespace detail {
inline std::vector<uint8_t> rows32(const cv::Mat &m)   // n x 32 CV_8U, rows possibly strided -> packed
{
    std::vector<uint8_t> out((size_t)m.rows * 32);
    for (int r = 0; r < m.rows; r++) memcpy(&out[(size_t)r * 32], m.data + (size_t)r * m.step, 32);
    return out;
}
inline void pose34(const cv::Mat &Tcw, float *R, float *t)   // 4x4 CV_32F
{
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = Tcw.at<float>(r, c); t[r] = Tcw.at<float>(r, 3); }
}
}  // namespace detail

// include/ORBmatcher.h:36-140, the two tracking overloads with their reference signatures
class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0, int maxKeypoints = 8192, int maxMapPoints = 65535)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device)
    {
        plf::check(plf_matcher_create(device, maxKeypoints, maxMapPoints, 1024, 1, &m_), "plf_matcher_create");
    }
    ~ORBmatcher() { plf_matcher_destroy(m_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return plf_hamming256(a.data, b.data); }   // include/ORBmatcher.h:44

    // int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3)   include/ORBmatcher.h:61
    template <class FrameT, class MapPointT> int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3)
    {
        const int N = (int)F.mvKeysUn.size(), M = (int)vpMapPoints.size();
        if (N == 0 || M == 0) return 0;
        std::vector<float> px(M), py(M), pxr(M), vc(M);
        std::vector<int32_t> lvl(M);
        std::vector<uint8_t> inview(M), obs(M), desc((size_t)M * 32);
        for (int i = 0; i < M; i++) {
            MapPointT *p = vpMapPoints[i];
            inview[i] = p && p->mbTrackInView && !p->isBad();
            if (!inview[i]) continue;
            px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; pxr[i] = p->mTrackProjXR; lvl[i] = p->mnTrackScaleLevel; vc[i] = p->mTrackViewCos;
            obs[i] = p->Observations() > 0;
            const cv::Mat d = p->GetDescriptor();
            memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        std::vector<int32_t> init(N, -1);
        for (int k = 0; k < N; k++)
            if (F.mvpMapPoints[k] && F.mvpMapPoints[k]->Observations() > 0) init[k] = -2;
        plf::DeviceArray<plf_keypoint> dk(N, device_); dk.upload(F.mvKeysUn.data(), N);
        plf::DeviceArray<float> dur(F.mvuRight, device_), dsc(F.mvScaleFactors, device_);
        plf::DeviceArray<uint8_t> dd(detail::rows32(F.mDescriptors), device_), dmd(desc, device_), div(inview, device_), dob(obs, device_);
        plf::DeviceArray<float> dpx(px, device_), dpy(py, device_), dpxr(pxr, device_), dvc(vc, device_);
        plf::DeviceArray<int32_t> dl(lvl, device_), dm(init, device_), dn(1, device_);
        plf_frame_view fv = {N, nullptr, dk.get(), dur.get(), dd.get(), FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY, dsc.get(),
                             (int32_t)F.mvScaleFactors.size()};
        plf_mappoint_view mv = {M, dpx.get(), dpy.get(), dpxr.get(), dl.get(), dvc.get(), div.get(), dmd.get(), dob.get()};
        plf::check(plf_match_project_points(m_, &fv, 1, &mv, th, mfNNratio, dm.get(), N, dn.get(), nullptr), "ORBmatcher::SearchByProjection");
        const std::vector<int32_t> match = dm.download();
        for (int k = 0; k < N; k++)
            if (match[k] >= 0) F.mvpMapPoints[k] = vpMapPoints[match[k]];
        return dn.download()[0];
    }

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)   include/ORBmatcher.h:78
    template <class FrameT> int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)
    {
        const int N = (int)CurrentFrame.mvKeysUn.size(), NLst = (int)LastFrame.mvKeys.size();
        if (N == 0 || NLst == 0) return 0;
        std::vector<uint8_t> has(NLst), outl(NLst), obs(NLst), desc((size_t)NLst * 32);
        std::vector<float> xw((size_t)NLst * 3);
        for (int i = 0; i < NLst; i++) {
            auto *p = LastFrame.mvpMapPoints[i];
            has[i] = p != nullptr; outl[i] = LastFrame.mvbOutlier[i];
            if (!p) continue;
            obs[i] = p->Observations() > 0;
            const cv::Mat w = p->GetWorldPos();
            for (int c = 0; c < 3; c++) xw[(size_t)i * 3 + c] = w.template at<float>(c);
            const cv::Mat d = p->GetDescriptor();
            memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        std::vector<int32_t> init(N, -1);
        for (int k = 0; k < N; k++)
            if (CurrentFrame.mvpMapPoints[k] && CurrentFrame.mvpMapPoints[k]->Observations() > 0) init[k] = -2;
        plf_pose_pair P;
        detail::pose34(CurrentFrame.mTcw, P.Rcw, P.tcw); detail::pose34(LastFrame.mTcw, P.Rlw, P.tlw);
        P.fx = FrameT::fx; P.fy = FrameT::fy; P.cx = FrameT::cx; P.cy = FrameT::cy; P.bf = CurrentFrame.mbf; P.b = CurrentFrame.mb;
        plf::DeviceArray<plf_keypoint> dk(N, device_), dlk(NLst, device_);
        dk.upload(CurrentFrame.mvKeysUn.data(), N); dlk.upload(LastFrame.mvKeysUn.data(), NLst);
        plf::DeviceArray<float> dur(CurrentFrame.mvuRight, device_), dsc(CurrentFrame.mvScaleFactors, device_), dxw(xw, device_);
        plf::DeviceArray<uint8_t> dd(detail::rows32(CurrentFrame.mDescriptors), device_), dh(has, device_), dou(outl, device_), dob(obs, device_), dmd(desc, device_);
        plf::DeviceArray<int32_t> dm(init, device_), dn(1, device_);
        plf_frame_view fv = {N, nullptr, dk.get(), dur.get(), dd.get(), FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY, dsc.get(),
                             (int32_t)CurrentFrame.mvScaleFactors.size()};
        plf_lastframe_view lv = {NLst, dh.get(), dou.get(), dxw.get(), dlk.get(), dmd.get(), dob.get()};
        plf::check(plf_match_project_lastframe(m_, &fv, &lv, &P, th, bMono, mbCheckOrientation ? 2 : 0, dm.get(), dn.get(), nullptr), "ORBmatcher::SearchByProjection(last frame)");
        const std::vector<int32_t> match = dm.download();
        for (int k = 0; k < N; k++) {
            if (match[k] >= 0) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[match[k]];
            else if (match[k] == -3) CurrentFrame.mvpMapPoints[k] = NULL;   // assigned, then removed by the rotation-consistency check: NULL in the reference, whatever the key point held before
        }
        return dn.download()[0];
    }
    plf_matcher *handle() { return m_; }

private:
    plf_matcher *m_ = nullptr;
    float mfNNratio;
    bool mbCheckOrientation;
    int device_;
};

// include/LSDmatcher.h:27-78, the three SearchByProjection overloads with their reference signatures (+ SearchForTriangulation / Fuse)
class LSDmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    LSDmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0, int maxLines = 4096, int maxMapLines = 65535)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device)
    {
        plf::check(plf_matcher_create(device, 1, maxMapLines, maxLines, 1, &m_), "plf_matcher_create");
    }
    ~LSDmatcher() { plf_matcher_destroy(m_); }
    LSDmatcher(const LSDmatcher &) = delete;
    LSDmatcher &operator=(const LSDmatcher &) = delete;
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return plf_hamming256(a.data, b.data); }   // include/LSDmatcher.h:43

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th = 3, const bool bMono = false)   include/LSDmatcher.h:32
    template <class FrameT> int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float /*th*/ = 3, const bool /*bMono*/ = false)
    {
        return knn_assign(LastFrame.mLdesc, LastFrame.mvpMapLines, CurrentFrame.mLdesc, CurrentFrame.mvpMapLines);
    }
    // int SearchByProjection(KeyFrame *pKF, Frame &F, std::vector<MapLine*> &vpMapLineMatches)   include/LSDmatcher.h:35
    template <class KeyFrameT, class FrameT, class MapLineT> int SearchByProjection(KeyFrameT *pKF, FrameT &F, std::vector<MapLineT *> &vpMapLineMatches)
    {
        vpMapLineMatches.assign((size_t)F.mLdesc.rows, nullptr);
        const std::vector<MapLineT *> kfLines = pKF->GetMapLineMatches();
        return knn_assign(pKF->mLineDescriptors, kfLines, F.mLdesc, vpMapLineMatches);
    }
    // int SearchByProjection(Frame &F, const std::vector<MapLine*> &vpMapLines, const float th = 3)   include/LSDmatcher.h:40
    template <class FrameT, class MapLineT> int SearchByProjection(FrameT &F, const std::vector<MapLineT *> &vpMapLines, const float th = 3)
    {
        const int NL = (int)F.mvKeylinesUn.size(), M = (int)vpMapLines.size();
        if (NL == 0 || M == 0) return 0;
        std::vector<float> x1(M), y1(M), x2(M), y2(M), vc(M);
        std::vector<int32_t> lvl(M);
        std::vector<uint8_t> inview(M), desc((size_t)M * 32);
        for (int i = 0; i < M; i++) {
            MapLineT *p = vpMapLines[i];
            inview[i] = p && p->mbTrackInView && !p->isBad();
            if (!inview[i]) continue;
            x1[i] = p->mTrackProjX1; y1[i] = p->mTrackProjY1; x2[i] = p->mTrackProjX2; y2[i] = p->mTrackProjY2; lvl[i] = p->mnTrackScaleLevel; vc[i] = p->mTrackViewCos;
            const cv::Mat d = p->GetDescriptor();
            memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        std::vector<int32_t> init(NL, -1);
        for (int k = 0; k < NL; k++)
            if (F.mvpMapLines[k] && F.mvpMapLines[k]->Observations() > 0) init[k] = -2;
        plf::DeviceArray<plf_keyline> dkl(NL, device_); dkl.upload(F.mvKeylinesUn.data(), NL);
        plf::DeviceArray<float> dsc(F.mvScaleFactors, device_), dx1(x1, device_), dy1(y1, device_), dx2(x2, device_), dy2(y2, device_), dvc(vc, device_);
        plf::DeviceArray<uint8_t> dd(detail::rows32(F.mLdesc), device_), dmd(desc, device_), div(inview, device_);
        plf::DeviceArray<int32_t> dl(lvl, device_), dm(init, device_), dn(1, device_);
        plf_lineframe_view fv = {NL, nullptr, dkl.get(), dd.get(), dsc.get()};
        plf_mapline_view mv = {M, dx1.get(), dy1.get(), dx2.get(), dy2.get(), dl.get(), dvc.get(), div.get(), dmd.get()};
        plf::check(plf_match_project_lines(m_, &fv, 1, &mv, th, mfNNratio, dm.get(), NL, dn.get(), nullptr), "LSDmatcher::SearchByProjection");
        const std::vector<int32_t> match = dm.download();
        for (int k = 0; k < NL; k++)
            if (match[k] >= 0) F.mvpMapLines[k] = vpMapLines[match[k]];
        return dn.download()[0];
    }
    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo)   include/LSDmatcher.h:54
    template <class KeyFrameT> int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo)
    {
        vMatchedPairs.clear();
        const int n1 = pKF1->mLineDescriptors.rows, n2 = pKF2->mLineDescriptors.rows;
        if (n1 == 0 || n2 < 2) return 0;
        std::vector<uint8_t> h1(n1), h2(n2), s1(n1), s2(n2);
        for (int i = 0; i < n1; i++) { h1[i] = pKF1->GetMapLine(i) != nullptr; s1[i] = pKF1->mvuRightLineStart[i] >= 0 && pKF1->mvuRightLineEnd[i] >= 0; }
        for (int i = 0; i < n2; i++) { h2[i] = pKF2->GetMapLine(i) != nullptr; s2[i] = pKF2->mvuRightLineStart[i] >= 0 && pKF2->mvuRightLineEnd[i] >= 0; }
        plf::DeviceArray<uint8_t> d1(detail::rows32(pKF1->mLineDescriptors), device_), d2(detail::rows32(pKF2->mLineDescriptors), device_), dh1(h1, device_),
            dh2(h2, device_), ds1(s1, device_), ds2(s2, device_);
        plf::DeviceArray<int32_t> dm(n1, device_), dn(1, device_);
        plf::check(plf_match_lines_triangulation(m_, d1.get(), n1, d2.get(), n2, dh1.get(), dh2.get(), ds1.get(), ds2.get(), bOnlyStereo, 0.1f, dm.get(), dn.get(), nullptr),
                   "LSDmatcher::SearchForTriangulation");
        const std::vector<int32_t> match = dm.download();
        for (int q = 0; q < n1; q++)
            if (match[q] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)q, (size_t)match[q]));
        return (int)vMatchedPairs.size();
    }
    // int Fuse(KeyFrame *pKF, const vector<MapLine*> &vpMapLines)   include/LSDmatcher.h:58: the device search, then the reference's map mutation in list order
    template <class KeyFrameT, class MapLineT> int Fuse(KeyFrameT *pKF, const std::vector<MapLineT *> &vpMapLines)
    {
        const int nkf = pKF->mLineDescriptors.rows, M = (int)vpMapLines.size();
        if (M == 0) return 0;
        std::vector<uint8_t> valid(M), desc((size_t)M * 32);
        for (int i = 0; i < M; i++) {
            MapLineT *p = vpMapLines[i];
            valid[i] = p && !p->isBad() && !p->IsInKeyFrame(pKF);
            if (!valid[i]) continue;
            const cv::Mat d = p->GetDescriptor();
            memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        plf::DeviceArray<uint8_t> dk(detail::rows32(pKF->mLineDescriptors), device_), dmd(desc, device_), dv(valid, device_);
        plf::DeviceArray<int32_t> db(M, device_), dn(1, device_);
        plf::check(plf_match_lines_fuse(m_, dk.get(), nkf, dmd.get(), dv.get(), M, db.get(), dn.get(), nullptr), "LSDmatcher::Fuse");
        const std::vector<int32_t> best = db.download();
        int nFused = 0;
        for (int i = 0; i < M; i++) {
            if (best[i] < 0) continue;
            MapLineT *pML = vpMapLines[i], *pMLinKF = pKF->GetMapLine((size_t)best[i]);
            if (pMLinKF) {
                if (!pMLinKF->isBad()) { if (pMLinKF->Observations() > pML->Observations()) pML->Replace(pMLinKF); else pMLinKF->Replace(pML); }
            } else { pML->AddObservation(pKF, (size_t)best[i]); pKF->AddMapLine(pML, (size_t)best[i]); }
            nFused++;
        }
        return nFused;
    }
    plf_matcher *handle() { return m_; }

private:
    // the brute-force kNN + MAD rule shared by the two frame / keyframe overloads: query = the side that holds MapLines, train = the frame being filled
    template <class VecQ, class VecT> int knn_assign(const cv::Mat &qdesc, const VecQ &qlines, const cv::Mat &tdesc, VecT &tlines)
    {
        const int nq = qdesc.rows, nt = tdesc.rows;
        if (nq == 0 || nt < 2) return 0;
        std::vector<uint8_t> has(nq);
        for (int q = 0; q < nq; q++) has[q] = qlines[q] != nullptr;
        plf::DeviceArray<uint8_t> dq(detail::rows32(qdesc), device_), dt(detail::rows32(tdesc), device_), dh(has, device_);
        plf::DeviceArray<int32_t> dm(nt, device_), dn(1, device_);
        dm.fill(0xFF);
        plf::check(plf_match_lines_lastframe(m_, dq.get(), nq, dt.get(), nt, dh.get(), dm.get(), dn.get(), nullptr), "LSDmatcher::SearchByProjection(kNN)");
        const std::vector<int32_t> match = dm.download();
        for (int t = 0; t < nt; t++)
            if (match[t] >= 0) tlines[t] = qlines[match[t]];
        return dn.download()[0];
    }
    plf_matcher *m_ = nullptr;
    float mfNNratio;
    bool mbCheckOrientation;
    int device_;
};
}  // namespace ORB_SLAM2_PLF
#endif
#endif
