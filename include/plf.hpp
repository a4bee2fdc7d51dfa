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
inline void check(int st, const char *what) { if (st != PLF_OK) throw Error(st, what); }

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
    explicit LineSegment(int nlines = 100, int maxWidth = 640, int maxHeight = 480, int maxBatch = 1, int device = 0) : nlines_(nlines)
    {
        plf_line_params p = {nlines, 0, device, maxWidth, maxHeight, maxBatch, PLF_LBD_BLURRED};
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
    plf_line *handle() { return h_; }

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

private:
    plf_matcher *m_;
    float mfNNratio;
    bool mbCheckOrientation;
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

}  // namespace plf

// ---------------------------------------------------------------------------------------------------------------
// Drop-in adapters with the EXACT reference signatures; compiled only where OpenCV (+ contrib line_descriptor) exists.
// ---------------------------------------------------------------------------------------------------------------
#if defined(PLF_WITH_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
namespace ORB_SLAM2_PLF {
static_assert(sizeof(cv::KeyPoint) == sizeof(plf_keypoint), "cv::KeyPoint must be 28 bytes");
class ORBextractor : public plf::ORBextractor {
public:
    using plf::ORBextractor::ORBextractor;
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
}  // namespace ORB_SLAM2_PLF
#endif
#endif
