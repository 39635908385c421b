// stand-in for the reference's include/KeyFrame.h: the members adapter/ORBmatcher_bow.cc reads (same names and types)
#ifndef CVSTUB_KEYFRAME_H
#define CVSTUB_KEYFRAME_H
#include <vector>
#include <opencv2/core/core.hpp>
#include "MapPoint.h"
#include <Thirdparty/DBoW2/DBoW2/BowVector.h>
#include <Thirdparty/DBoW2/DBoW2/FeatureVector.h>
namespace ORB_SLAM2 {
class KeyFrame
{
public:
    KeyFrame() : N(0), fx(0), fy(0), cx(0), cy(0), mbBad(false) {}
    bool isBad() { return mbBad; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { return Rcw.clone(); }
    cv::Mat GetTranslation() { return tcw.clone(); }

    int N;
    float fx, fy, cx, cy;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    std::vector<float> mvScaleFactors, mvLevelSigma2;

    bool mbBad;                             // protected in the reference
    std::vector<MapPoint *> mvpMapPoints;   // protected in the reference
    cv::Mat Ow, Rcw, tcw;                   // protected in the reference
};
}
#endif
