// stand-in for the reference's include/ORBmatcher.h (:41-103): the declarations of the three methods the adaptor defines
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
