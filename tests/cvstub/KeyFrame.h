// stand-in for the reference's include/KeyFrame.h: the members adapter/ORBmatcher_bow.cc reads (same names and types)
#ifndef CVSTUB_KEYFRAME_H
#define CVSTUB_KEYFRAME_H
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "MapPoint.h"
#include <Thirdparty/DBoW2/DBoW2/BowVector.h>
#include <Thirdparty/DBoW2/DBoW2/FeatureVector.h>
namespace ORB_SLAM2 {
class KeyFrame
{
public:
    KeyFrame() : N(0), fx(0), fy(0), cx(0), cy(0), mbf(0), mnScaleLevels(0), mfLogScaleFactor(0), mnMinX(0), mnMinY(0), mnMaxX(0), mnMaxY(0), mbBad(false) {}
    bool isBad() { return mbBad; }
    bool IsInImage(const float &x, const float &y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }   // src/KeyFrame.cc:649-652
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    std::set<MapPoint *> GetMapPoints()
    {
        std::set<MapPoint *> s;
        for (size_t i = 0; i < mvpMapPoints.size(); i++)
            if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]);
        return s;
    }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { return Rcw.clone(); }
    cv::Mat GetTranslation() { return tcw.clone(); }

    int N;
    float fx, fy, cx, cy, mbf;
    int mnScaleLevels;
    float mfLogScaleFactor;
    int mnMinX, mnMinY, mnMaxX, mnMaxY;      // const int in the reference
    std::vector<float> mvInvLevelSigma2;
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
