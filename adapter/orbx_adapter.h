// adapter/orbx_adapter.h -- helpers shared by the matcher adaptors: DBoW2::FeatureVector -> CSR and the per-feature
// attribute arrays of orbx_featset.
#ifndef ORBX_ADAPTER_H
#define ORBX_ADAPTER_H

#include <stdint.h>
#include <string.h>
#include <vector>

#include <orbx.h>

#include <Thirdparty/DBoW2/DBoW2/FeatureVector.h>

namespace orbx_adapter
{

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

} // namespace orbx_adapter

#endif
