#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own DBoW2::BowVector / DBoW2::FeatureVector code.

Runs in the build container only (needs /root/reference to have built oracle/_ref/libdbow2_ref.so:
`make -C oracle ref`).  Writes tests/golden/dbow2_ref_s<seed>.npz: seeded per-feature (word id, word weight,
node id) sequences -> the BowVector (ids, L1-normalised doubles) and the FeatureVector (CSR) the reference classes
produce when driven the way TemplatedVocabulary::transform drives them (TemplatedVocabulary.h:1147-1165,1188-1192).

Cases (each a separate key prefix):
  rand*  : random ids from a small alphabet so that words repeat many times, weights spanning 12 decades so that the
           fp64 summation ORDER matters, a fraction of zero weights ("stopped" words, :1157), n in {0, 1, 7, 500, 3000}
  voc    : the per-feature triplets of a seeded vocabulary descent (tools/synth.vocab_tree + the CPU oracle's descent;
           the descent itself -- TemplatedVocabulary.h needs OpenCV -- stays unpinned), together with the vocabulary and
           the descriptors, so that the GPU transform can be run end to end against the reference's accumulation.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdbow2_ref.so"))
    for f in ("ref_bowvector_accumulate", "ref_featurevector_build", "ref_bowvector_add_if_not_exist"):
        getattr(L, f).restype = C.c_int
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_accumulate(L, wid, ww, nid):
    n = len(wid)
    wid = np.ascontiguousarray(wid, np.uint32); ww = np.ascontiguousarray(ww, np.float64); nid = np.ascontiguousarray(nid, np.uint32)
    bid = np.zeros(max(n, 1), np.uint32); bval = np.zeros(max(n, 1), np.float64)
    nb = L.ref_bowvector_accumulate(_p(wid), _p(ww), n, 1, _p(bid), _p(bval))
    fid = np.zeros(max(n, 1), np.uint32); foff = np.zeros(n + 1, np.int32); ffeat = np.zeros(max(n, 1), np.uint32)
    nn = L.ref_featurevector_build(_p(nid), _p(ww), n, _p(fid), _p(foff), _p(ffeat))
    return dict(bow_id=bid[:nb].copy(), bow_val=bval[:nb].copy(), fv_node_id=fid[:nn].copy(), fv_node_off=foff[:nn + 1].copy(),
                fv_feat=ffeat[:foff[nn]].copy())


def main(seed=301):
    L = ref_lib()
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for ci, n in enumerate((0, 1, 7, 500, 3000)):
        wid = rng.integers(0, max(3, n // 9), n).astype(np.uint32)
        ww = (10.0 ** rng.uniform(-6, 6, n)) * rng.uniform(0.5, 1.5, n)
        ww[rng.random(n) < 0.07] = 0.0
        nid = rng.integers(0, 40, n).astype(np.uint32)
        r = ref_accumulate(L, wid, ww, nid)
        pre = f"rand{ci}_"
        out[pre + "word_id"], out[pre + "word_weight"], out[pre + "node_id"] = wid, ww, nid
        for k, v in r.items():
            out[pre + k] = v
    # vocabulary case: descent by the CPU oracle (unpinned), accumulation by the reference
    from oracle import oracle_py
    oracle_py.build()
    k, Lv, levelsup = 10, 3, 1
    centers = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    data = synth.flip_bits(rng, centers[rng.integers(0, 40, 700)], 0.08)
    par, leaf, nd, w = synth.vocab_tree(seed + 1, k, Lv, stop_frac=0.05, data=data)
    t = oracle_py.Vocabulary(k, Lv, par, leaf, nd, w).transform(data, levelsup)
    r = ref_accumulate(L, t["word_id"], t["word_weight"], t["node_id"])
    out.update(voc_k=np.int32(k), voc_L=np.int32(Lv), voc_levelsup=np.int32(levelsup), voc_parent=par, voc_is_leaf=leaf,
               voc_node_desc=nd, voc_weight=w, voc_features=data, voc_word_id=t["word_id"], voc_word_weight=t["word_weight"],
               voc_node_id=t["node_id"])
    for kk, v in r.items():
        out["voc_" + kk] = v
    path = os.path.join(ROOT, "tests", "golden", f"dbow2_ref_s{seed}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
