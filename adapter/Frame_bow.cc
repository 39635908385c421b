// adapter/Frame_bow.cc -- Frame::ComputeBoW (reference src/Frame.cc:459-466) and Frame::UndistortKeyPoints (:470-515) over liborbx.
// KeyFrame::ComputeBoW (src/KeyFrame.cc:80-90) is the same five lines on the keyframe's members.
#include "Frame.h"

#include <stdexcept>
#include <vector>

#include "orbx_adapter.h"

namespace ORB_SLAM2
{

void Frame::ComputeBoW()
{
    if (!mBowVec.empty())                                // :461
        return;
    orbx_vocab *voc = orbx_adapter::vocabulary();
    if (!voc)
        throw std::runtime_error("orbx_adapter::LoadVocabulary has not been called");
    const size_t n = (size_t)(N > 0 ? N : 1);
    std::vector<uint32_t> bow_id(n), fv_id(n), fv_feat(n);
    std::vector<double> bow_val(n);
    std::vector<int32_t> fv_off(n + 1);
    int nbow = 0, nnodes = 0;
    // TemplatedVocabulary::transform(features, mBowVec, mFeatVec, 4) on the device: word descent, TF-IDF weights added in feature
    // order, L1 norm, FeatureVector at levelsup = 4 -- the doubles are DBoW2's bit for bit (tests/test_dbow2_ref.py)
    if (orbx_bow_transform(voc, orbx_adapter::dense_descriptors(mDescriptors, N), N, 4, NULL, NULL, NULL, &bow_id[0], &bow_val[0], &nbow, &fv_id[0], &fv_off[0], &fv_feat[0],
                           &nnodes) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    for (int i = 0; i < nbow; i++)                        // ids arrive ascending: every insert is at the end of the map
        mBowVec.insert(mBowVec.end(), std::make_pair((DBoW2::WordId)bow_id[i], (DBoW2::WordValue)bow_val[i]));
    for (int j = 0; j < nnodes; j++)
        mFeatVec.insert(mFeatVec.end(), std::make_pair((DBoW2::NodeId)fv_id[j],
                                                       std::vector<unsigned int>(fv_feat.begin() + fv_off[j], fv_feat.begin() + fv_off[j + 1])));
}

void Frame::UndistortKeyPoints()
{
    if (mDistCoef.at<float>(0) == 0.0) {                  // :472-476
        mvKeysUn = mvKeys;
        return;
    }
    std::vector<float> xy(2 * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { xy[2 * i] = mvKeys[i].pt.x; xy[2 * i + 1] = mvKeys[i].pt.y; }
    // cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) (:490): five fixed-point iterations in double, on the device
    if (orbx_undistort_keypoints(orbx_adapter::Device(), &xy[0], N, mK.at<float>(0, 0), mK.at<float>(1, 1), mK.at<float>(0, 2), mK.at<float>(1, 2),
                                 mDistCoef.ptr<float>(), (int)mDistCoef.total(), &xy[0]) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
    mvKeysUn.resize(N);                                   // :493-500
    for (int i = 0; i < N; i++) {
        cv::KeyPoint kp = mvKeys[i];
        kp.pt.x = xy[2 * i];
        kp.pt.y = xy[2 * i + 1];
        mvKeysUn[i] = kp;
    }
}

} // namespace ORB_SLAM2
