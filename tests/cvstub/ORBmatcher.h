// stand-in for the reference's include/ORBmatcher.h (:41-103): the declarations of the six methods the adaptors define
// (tests/test_adapter.py::test_signatures_match_the_reference_header checks them against the reference's header text)
#ifndef CVSTUB_ORBMATCHER_H
#define CVSTUB_ORBMATCHER_H
#include <utility>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class ORBmatcher
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                               const bool bOnlyStereo);
protected:
    float mfNNratio;
    bool mbCheckOrientation;
};
}
#endif
