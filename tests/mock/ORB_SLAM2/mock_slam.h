// MOCK of the members of ORB_SLAM2::Frame / KeyFrame / MapPoint / MapLine (include/Frame.h, KeyFrame.h, MapPoint.h, MapLine.h of the reference)
// that the exact-signature adapters of include/plf.hpp read and write -- same names, same types.  Test infrastructure only.
#pragma once
#include <set>
#include <utility>
#include <vector>
#include <opencv2/core.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class MapPoint {
public:
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 1;
    bool mbTrackInView = false;
    bool isBad() const { return bad; }
    int Observations() const { return nObs; }
    cv::Mat GetDescriptor() const { return desc; }
    cv::Mat GetWorldPos() const { return pos; }
    bool bad = false;
    int nObs = 1;
    cv::Mat desc, pos;
};
class MapLine {
public:
    float mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX1R = 0, mTrackProjX2 = 0, mTrackProjY2 = 0, mTrackProjX2R = 0;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 1;
    bool mbTrackInView = false;
    bool isBad() const { return bad; }
    int Observations() const { return nObs; }
    cv::Mat GetDescriptor() const { return mLDescriptor; }
    bool IsInKeyFrame(KeyFrame *kf) const { return inKF.count(kf) != 0; }
    void Replace(MapLine *other) { replacedBy = other; bad = true; }
    void AddObservation(KeyFrame *kf, size_t idx) { inKF.insert(kf); nObs++; lastObsIdx = (int)idx; }
    bool bad = false;
    int nObs = 1, lastObsIdx = -1;
    cv::Mat mLDescriptor;
    std::set<KeyFrame *> inKF;
    MapLine *replacedBy = nullptr;
};
class Frame {
public:
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY;
    float mbf = 40.f, mb = 40.f / 525.f;
    int N = 0, NL = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    std::vector<float> mvScaleFactors;
    std::vector<cv::line_descriptor::KeyLine> mvKeylines, mvKeylinesUn;
    cv::Mat mLdesc;
    std::vector<MapLine *> mvpMapLines;
};
class KeyFrame {
public:
    cv::Mat mLineDescriptors;
    std::vector<float> mvuRightLineStart, mvuRightLineEnd;
    std::vector<MapLine *> GetMapLineMatches() { return lines; }
    MapLine *GetMapLine(const size_t &idx) { return lines[idx]; }
    void AddMapLine(MapLine *p, const size_t &idx) { lines[idx] = p; }
    std::vector<MapLine *> lines;
};
}  // namespace ORB_SLAM2
