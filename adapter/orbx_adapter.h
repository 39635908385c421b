// adapter/orbx_adapter.h -- helpers shared by the matcher adaptors: DBoW2::FeatureVector -> CSR and the per-feature
// attribute arrays of orbx_featset.
#ifndef ORBX_ADAPTER_H
#define ORBX_ADAPTER_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdexcept>
#include <vector>

#include <opencv2/core/core.hpp>

#include <orbx.h>

#include "orbx_device.h"

#include <Thirdparty/DBoW2/DBoW2/FeatureVector.h>

namespace orbx_adapter
{

// The ABI reads std::vector<cv::KeyPoint> storage as orbx_keypoint records and descriptor matrices as dense N x 32 byte arrays.
// A build against an OpenCV whose cv::KeyPoint is laid out differently must fail here, not match garbage (OpenCV 2.4 - 4.x all
// have pt.x, pt.y, size, angle, response, octave, class_id in 28 bytes).
typedef char keypoint_size_ok[sizeof(cv::KeyPoint) == sizeof(orbx_keypoint) ? 1 : -1];
typedef char keypoint_octave_ok[offsetof(cv::KeyPoint, octave) == offsetof(orbx_keypoint, octave) ? 1 : -1];
typedef char keypoint_class_id_ok[offsetof(cv::KeyPoint, class_id) == offsetof(orbx_keypoint, class_id) ? 1 : -1];
typedef char keypoint_angle_ok[offsetof(cv::KeyPoint, angle) == offsetof(orbx_keypoint, angle) ? 1 : -1];

// descriptor matrices reach the ABI through Mat::data: CV_8U, 32 columns, no row padding
inline const uint8_t *dense_descriptors(const cv::Mat &m, int n)
{
    if (n > 0 && (m.rows < n || m.cols != 32 || m.elemSize() != 1 || !m.isContinuous()))
        throw std::runtime_error("orbx adaptor: descriptor matrix is not a continuous N x 32 CV_8U matrix");
    return m.data;
}

// One ordered walk of the std::map (Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22): node ids ascending, feature indices in
// insertion (= ascending) order -- exactly the order orbx_featset requires.
struct Csr {
    std::vector<uint32_t> id, feat;
    std::vector<int32_t> off;
    Csr() : off(1, 0) {}
};

inline Csr flatten(const DBoW2::FeatureVector &fv)
{
    Csr c;
    c.id.reserve(fv.size());
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        c.id.push_back(it->first);
        c.feat.insert(c.feat.end(), it->second.begin(), it->second.end());
        c.off.push_back((int32_t)c.feat.size());
    }
    return c;
}

// the storage behind one orbx_featset (the struct only borrows pointers)
struct Side {
    Csr csr;
    std::vector<uint8_t> flag;
    std::vector<float> angle, x, y, u_right;
    std::vector<int32_t> octave;
    orbx_featset fs;

    // desc: the N x 32 CV_8U descriptor matrix (continuous), n = N
    void bind(const uint8_t *desc, int n)
    {
        memset(&fs, 0, sizeof fs);
        fs.n = n;
        fs.desc = desc;
        fs.nnodes = (int)csr.id.size();
        fs.node_id = csr.id.empty() ? NULL : &csr.id[0];
        fs.node_off = &csr.off[0];
        fs.feat = csr.feat.empty() ? NULL : &csr.feat[0];
        fs.flag = flag.empty() ? NULL : &flag[0];
        fs.angle = angle.empty() ? NULL : &angle[0];
        fs.x = x.empty() ? NULL : &x[0];
        fs.y = y.empty() ? NULL : &y[0];
        fs.octave = octave.empty() ? NULL : &octave[0];
        fs.u_right = u_right.empty() ? NULL : &u_right[0];
    }
};

// the storage behind one orbx_frame_feats: a Frame's (or KeyFrame's) undistorted keypoints as the projection searches read them
struct FrameSide {
    std::vector<float> x, y, angle;
    std::vector<int32_t> octave;
    std::vector<uint8_t> occupied;
    orbx_frame_feats ff;
};

// the storage behind one orbx_proj_points
struct PointSide {
    std::vector<float> u, v, aux, angle, view_cos;
    std::vector<int32_t> level;
    std::vector<uint8_t> desc, valid, has_obs;
    orbx_proj_points pp;
    explicit PointSide(size_t n) : u(n, 0.f), v(n, 0.f), aux(n, 0.f), angle(n, 0.f), view_cos(n, 0.f), level(n, 0), desc(32 * n, 0), valid(n, 0), has_obs(n, 0)
    {
        memset(&pp, 0, sizeof pp);
        pp.n = (int)n;
        if (n) {
            pp.u = &u[0]; pp.v = &v[0]; pp.aux = &aux[0]; pp.level = &level[0]; pp.angle = &angle[0]; pp.view_cos = &view_cos[0];
            pp.desc = &desc[0]; pp.valid = &valid[0]; pp.has_obs = &has_obs[0];
        }
    }
};

// The device vocabulary behind Frame::ComputeBoW / KeyFrame::ComputeBoW: loaded once where the reference loads its
// ORBVocabulary (src/System.cc:67-77, mpVocabulary->loadFromTextFile), from the same ORBvoc.txt.
inline orbx_vocab *&vocabulary()
{
    static orbx_vocab *v = NULL;
    return v;
}
inline int LoadVocabulary(const char *path, int device = -1) { return orbx_vocab_load_text(device < 0 ? Device() : device, path, &vocabulary()); }

#ifdef ORBX_ADAPTER_CAPTURE
// test hook (tests/adapter_driver.cc): the projection adaptors leave a copy of what they handed to the ABI, so that the test can give
// the CPU oracle the very same inputs.  Not compiled into a production build.
struct Capture {
    std::vector<float> cx, cy, cangle, curight, pu, pv, paux, pangle, pview;
    std::vector<int32_t> coctave, plevel;
    std::vector<uint8_t> coccupied, cdesc, pdesc, pvalid, phas_obs;
    float bounds[4];
};
inline Capture &capture(int slot = 0) { static Capture c[2]; return c[slot]; }
inline void capture_call(const orbx_frame_feats &f, const orbx_proj_points *p, int slot = 0)
{
    Capture &c = capture(slot);
    const size_t n = (size_t)f.n;
    c.cx.assign(f.x, f.x + n); c.cy.assign(f.y, f.y + n); c.cangle.assign(f.angle, f.angle + n); c.curight.assign(f.u_right, f.u_right + n);
    c.coctave.assign(f.octave, f.octave + n); c.coccupied.assign(f.occupied, f.occupied + n); c.cdesc.assign(f.desc, f.desc + 32 * n);
    c.bounds[0] = f.min_x; c.bounds[1] = f.min_y; c.bounds[2] = f.max_x; c.bounds[3] = f.max_y;
    if (p) {
        const size_t m = (size_t)p->n;
        c.pu.assign(p->u, p->u + m); c.pv.assign(p->v, p->v + m); c.paux.assign(p->aux, p->aux + m); c.pangle.assign(p->angle, p->angle + m);
        c.pview.assign(p->view_cos, p->view_cos + m); c.plevel.assign(p->level, p->level + m); c.pdesc.assign(p->desc, p->desc + 32 * m);
        c.pvalid.assign(p->valid, p->valid + m); c.phas_obs.assign(p->has_obs, p->has_obs + m);
    }
}
#define ORBX_CAPTURE(f, p) orbx_adapter::capture_call(f, p)
#define ORBX_CAPTURE2(f, p) orbx_adapter::capture_call(f, p, 1)
#else
#define ORBX_CAPTURE(f, p) do { } while (0)
#define ORBX_CAPTURE2(f, p) do { } while (0)
#endif

} // namespace orbx_adapter

#endif
