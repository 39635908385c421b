// adapter/ORBextractor.h -- drop-in replacement of the reference's include/ORBextractor.h (:58-139) whose
// implementation (adapter/ORBextractor.cc) forwards to liborbx.  Public interface = the reference's: constructor,
// operator(), the six getters and the public member mvImagePyramid; the CPU-side helpers (ExtractorNode,
// ComputePyramid, ComputeKeyPointsOctTree, DistributeOctTree, the pattern / umax tables) have no counterpart here:
// that work runs on the GPU behind orbx_extract*.
//
// Build: put adapter/ in front of the reference's include/ on the include path, compile adapter/*.cc instead of
// src/ORBextractor.cc, and link -lorbx.  Compile-checked in this repo against tests/cvstub (tests/test_adapter.py).
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>
#include <opencv/cv.h>

#include <orbx.h>

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // keypoints + descriptors of one CV_8UC1 image; the mask is ignored (as in the reference)
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);

    int GetLevels() { return nlevels; }
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Public in the reference and read by Frame::ComputeStereoMatches (src/Frame.cc:584,674,686,691).  With
    // adapter/Frame_stereo.cc nothing on the hot path reads it any more; other user code calls FetchPyramid() first.
    std::vector<cv::Mat> mvImagePyramid;
    void FetchPyramid();                     // device pyramid of the last operator() call -> mvImagePyramid (no 19-px border)

    // the liborbx handle: used by Frame::ComputeStereoMatches (both eyes) and by the one-call stereo front end
    orbx_extractor *Handle() const { return mH; }
    // orbx_extract_stereo on THIS extractor: both eyes + ComputeStereoMatches in one call (src/Frame.cc:82-97)
    void ExtractStereo(const cv::Mat &imLeft, const cv::Mat &imRight, float bf, float b, std::vector<cv::KeyPoint> &keysLeft,
                       cv::Mat &descLeft, std::vector<cv::KeyPoint> &keysRight, cv::Mat &descRight, std::vector<float> &uRight,
                       std::vector<float> &depth);

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    orbx_extractor *mH;

private:
    ORBextractor(const ORBextractor &);            // the handle owns device memory: not copyable
    ORBextractor &operator=(const ORBextractor &);
};

} // namespace ORB_SLAM2

#endif
