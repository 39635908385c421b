// stand-in for the reference's include/Frame.h: the members adapter/Frame_stereo.cc and adapter/ORBmatcher_bow.cc read
#ifndef CVSTUB_FRAME_H
#define CVSTUB_FRAME_H
#include <vector>
#include <opencv2/core/core.hpp>
#include "MapPoint.h"
#include "ORBextractor.h"
#include <Thirdparty/DBoW2/DBoW2/BowVector.h>
#include <Thirdparty/DBoW2/DBoW2/FeatureVector.h>
namespace ORB_SLAM2 {
class Frame
{
public:
    Frame() : mpORBextractorLeft(NULL), mpORBextractorRight(NULL), mbf(0), mb(0), N(0), mnScaleLevels(0), mfLogScaleFactor(0) {}
    void ComputeStereoMatches();
    void ComputeBoW();              // defined by adapter/Frame_bow.cc
    void UndistortKeyPoints();      // defined by adapter/Frame_bow.cc

    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    static float fx, fy, cx, cy;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    float mbf, mb;
    int N;
    int mnScaleLevels;
    float mfLogScaleFactor;
    std::vector<float> mvScaleFactors;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw, mK, mDistCoef;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints;
};
}
#endif
