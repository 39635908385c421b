// adapter/ORBextractor.cc -- ORB_SLAM2::ORBextractor over liborbx (replaces the reference's src/ORBextractor.cc).
// Call sites stay untouched: Frame::ExtractORB (src/Frame.cc:285-292), the getters read by the Frame constructors
// (src/Frame.cc:73-79,136-142,195-201), construction in src/Tracking.cc:124-130.
#include "ORBextractor.h"
#include "orbx_device.h"

#include <stdexcept>
#include <string.h>

namespace ORB_SLAM2
{

// cv::KeyPoint and orbx_keypoint are the same 28 bytes (pt.x, pt.y, size, angle, response, octave, class_id)
typedef char orbx_keypoint_is_cv_keypoint[sizeof(cv::KeyPoint) == sizeof(orbx_keypoint) ? 1 : -1];

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mH(NULL)
{
    // max_batch 2: one handle serves both eyes of a stereo frame through ExtractStereo()
    if (orbx_extractor_create(&mH, nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST, orbx_adapter::Device(), /*max_w*/ 4096, /*max_h*/ 4096,
                              /*max_batch*/ 2) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    orbx_get_scale_tables(mH, &mvScaleFactor[0], &mvInvScaleFactor[0], &mvLevelSigma2[0], &mvInvLevelSigma2[0]);
    orbx_get_features_per_level(mH, &mnFeaturesPerLevel[0]);
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor()
{
    orbx_extractor_destroy(mH);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
{
    if (_image.empty())                      // reference src/ORBextractor.cc:1264-1265: outputs untouched
        return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);         // :1269
    const int cap = orbx_max_keypoints(mH, image.cols, image.rows);
    if (cap < 0)
        throw std::runtime_error(orbx_last_error());
    _keypoints.resize(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0;
    if (orbx_extract(mH, image.data, image.cols, image.rows, image.step, reinterpret_cast<orbx_keypoint *>(&_keypoints[0]), desc.data, cap, &n) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    _keypoints.resize(n);
    if (n == 0)
        _descriptors.release();              // :1290-1291
    else
        desc.rowRange(0, n).copyTo(_descriptors);
}

void ORBextractor::FetchPyramid()
{
    for (int l = 0; l < nlevels; l++) {
        int w = 0, h = 0;
        if (orbx_pyramid_level(mH, 0, l, NULL, 0, &w, &h) != ORBX_OK)
            throw std::runtime_error(orbx_last_error());
        mvImagePyramid[l].create(h, w, CV_8U);
        if (orbx_pyramid_level(mH, 0, l, mvImagePyramid[l].data, mvImagePyramid[l].step, &w, &h) != ORBX_OK)
            throw std::runtime_error(orbx_last_error());
    }
}

void ORBextractor::ExtractStereo(const cv::Mat &imLeft, const cv::Mat &imRight, float bf, float b, std::vector<cv::KeyPoint> &keysLeft,
                                 cv::Mat &descLeft, std::vector<cv::KeyPoint> &keysRight, cv::Mat &descRight, std::vector<float> &uRight,
                                 std::vector<float> &depth)
{
    assert(imLeft.type() == CV_8UC1 && imRight.type() == CV_8UC1 && imLeft.cols == imRight.cols && imLeft.rows == imRight.rows &&
           imLeft.step == imRight.step);
    const int cap = orbx_max_keypoints(mH, imLeft.cols, imLeft.rows);
    if (cap < 0)
        throw std::runtime_error(orbx_last_error());
    std::vector<cv::KeyPoint> kps(2 * (size_t)cap);
    cv::Mat desc(2 * cap, 32, CV_8U);
    uRight.assign(cap, -1.0f);
    depth.assign(cap, -1.0f);
    int n[2] = { 0, 0 };
    if (orbx_extract_stereo(mH, imLeft.data, imRight.data, imLeft.cols, imLeft.rows, imLeft.step, bf, b, reinterpret_cast<orbx_keypoint *>(&kps[0]),
                            desc.data, cap, n, &uRight[0], &depth[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    keysLeft.assign(kps.begin(), kps.begin() + n[0]);
    keysRight.assign(kps.begin() + cap, kps.begin() + cap + n[1]);
    desc.rowRange(0, n[0]).copyTo(descLeft);
    desc.rowRange(cap, cap + n[1]).copyTo(descRight);
    uRight.resize(n[0]);
    depth.resize(n[0]);
}

} // namespace ORB_SLAM2
