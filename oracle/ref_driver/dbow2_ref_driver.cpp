// dbow2_ref_driver.cpp -- C entry points around the REFERENCE's own DBoW2::BowVector / DBoW2::FeatureVector
// (Thirdparty/DBoW2/DBoW2/BowVector.{h,cpp}, FeatureVector.{h,cpp}: the two source files of the hot path that compile
// without OpenCV).  Test infrastructure: built by oracle/Makefile (target ref) into oracle/_ref/libdbow2_ref.so from
// the sources where they lie under /root/reference; used by tools/gen_dbow2_golden.py to write the golden vectors
// tests/golden/dbow2_ref_*.npz and by tests/test_dbow2_ref.py when the library is present.  Nothing of the reference is
// copied: this file only calls its classes the way TemplatedVocabulary::transform does
// (TemplatedVocabulary.h:1147-1165: addWeight + addFeature per non-stopped feature, then normalize(L1)).
#include <stdint.h>

#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

// returns the number of BowVector entries; bow_id/bow_val need n entries
int ref_bowvector_accumulate(const uint32_t *word_id, const double *word_weight, int n, int normalize_l1, uint32_t *bow_id,
                             double *bow_val)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; i++)
        if (word_weight[i] > 0) v.addWeight(word_id[i], word_weight[i]);   // TemplatedVocabulary.h:1157-1159
    if (normalize_l1) v.normalize(DBoW2::L1);                              // :1188-1192 (m_scoring L1_NORM)
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) {
        bow_id[k] = it->first;
        bow_val[k] = it->second;
    }
    return k;
}

// CSR view of the FeatureVector in std::map iteration order; returns the number of nodes.
// node_id needs n entries, node_off n+1, feat n.
int ref_featurevector_build(const uint32_t *node_id_in, const double *word_weight, int n, uint32_t *node_id, int32_t *node_off,
                            uint32_t *feat)
{
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++)
        if (word_weight[i] > 0) fv.addFeature(node_id_in[i], (unsigned)i);   // TemplatedVocabulary.h:1160
    int k = 0, m = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        node_id[k] = it->first;
        node_off[k] = m;
        for (size_t j = 0; j < it->second.size(); j++) feat[m++] = it->second[j];
    }
    node_off[k] = m;
    return k;
}

// BowVector::addIfNotExist (BowVector.cpp:49-56), used by the BINARY weighting branch: first value wins
int ref_bowvector_add_if_not_exist(const uint32_t *word_id, const double *word_weight, int n, uint32_t *bow_id, double *bow_val)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; i++) v.addIfNotExist(word_id[i], word_weight[i]);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) {
        bow_id[k] = it->first;
        bow_val[k] = it->second;
    }
    return k;
}
}
