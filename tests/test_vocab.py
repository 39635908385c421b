"""DBoW2 vocabulary + transform (Frame::ComputeBoW).  CPU: the oracle against an
independent plain-Python restatement with dict/list standing in for std::map/std::vector.  GPU: the HIP
path against the oracle, bit-exact doubles included."""
import os

import numpy as np
import pytest

from tools import synth

POP = np.array([bin(i).count("1") for i in range(256)])


def _python_transform(k, L, parent, is_leaf, ndesc, weight, feats, levelsup):
    n = len(parent) + 1
    children = [[] for _ in range(n)]
    for i, p in enumerate(parent):
        children[p].append(i + 1)
    word = {}
    for i in range(1, n):
        if is_leaf[i - 1]:
            word[i] = len(word)
    bow, fv = {}, {}
    for fi, f in enumerate(feats):
        node, level, nid = 0, 0, 0
        while True:
            level += 1
            ch = children[node]
            ds = [int(POP[f ^ ndesc[c - 1]].sum()) for c in ch]
            node = ch[int(np.argmin(ds))]          # first minimum
            if level == L - levelsup:
                nid = node
            if not children[node]:
                break
        w = weight[node - 1]
        if w > 0:
            bow[word[node]] = bow.get(word[node], 0.0) + w if word[node] in bow else w
            fv.setdefault(nid, []).append(fi)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    norm = 0.0
    for v in vals:
        norm += abs(v)
    if norm > 0:
        vals = [v / norm for v in vals]
    return ids, vals, {k_: fv[k_] for k_ in sorted(fv)}


def _data(seed, n=600):
    rng = np.random.default_rng(seed)
    centers = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    return synth.flip_bits(np.random.Generator(np.random.PCG64(seed)), centers[rng.integers(0, 40, n)], 0.08)


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 1), (4, 5, 2), (7, 2, 4), (10, 3, 0)])
def test_oracle_transform_vs_python(oracle, k, L, levelsup):
    data = _data(1)
    par, leaf, nd, w = synth.vocab_tree(2, k, L, stop_frac=0.05, data=data)
    v = oracle.Vocabulary(k, L, par, leaf, nd, w)
    assert v.nodes() == len(par) + 1 and v.words() == int(leaf.sum())
    t = v.transform(data, levelsup)
    ids, vals, fv = _python_transform(k, L, par, leaf, nd, w, data, levelsup)
    assert t["bow_id"].tolist() == ids
    assert t["bow_val"].tolist() == vals                      # identical doubles
    assert abs(t["bow_val"].sum() - 1.0) < 1e-12
    assert t["fv_node_id"].tolist() == list(fv)
    for j, nid in enumerate(fv):
        assert t["fv_feat"][t["fv_node_off"][j]:t["fv_node_off"][j + 1]].tolist() == fv[nid]
    if L - levelsup <= 0:
        assert t["fv_node_id"].tolist() == [0]                # everything hangs off the root (:1228)


def test_oracle_text_loader_roundtrip(oracle, tmp_path):
    data = _data(3)
    par, leaf, nd, w = synth.vocab_tree(4, 6, 3, data=data)
    path = os.path.join(tmp_path, "voc.txt")
    synth.write_vocab_text(path, 6, 3, par, leaf, nd, w)
    a = oracle.Vocabulary(6, 3, par, leaf, nd, w).transform(data, 1)
    b = oracle.Vocabulary(path=path).transform(data, 1)
    assert all((a[k_] == b[k_]).all() for k_ in a)
    with pytest.raises(ValueError):
        oracle.Vocabulary(path=os.path.join(tmp_path, "missing.txt"))


# ------------------------------------------------------------------------------------------ GPU

def _cmp(a, b):
    for k_ in a:
        assert a[k_].dtype == b[k_].dtype and a[k_].shape == b[k_].shape, k_
        assert a[k_].tobytes() == b[k_].tobytes(), k_


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,levelsup,n", [(10, 3, 1, 1000), (10, 4, 2, 2003), (4, 5, 4, 300), (10, 3, 0, 64), (3, 2, 5, 17), (10, 4, 2, 5000), (10, 4, 1, 4096)])
def test_hip_transform_parity(pkg, oracle, k, L, levelsup, n):
    data = _data(5, n)
    par, leaf, nd, w = synth.vocab_tree(6, k, L, stop_frac=0.05, data=data)
    got = pkg.ORBVocabulary(k, L, par, leaf, nd, w).transform(data, levelsup)
    exp = oracle.Vocabulary(k, L, par, leaf, nd, w).transform(data, levelsup)
    _cmp(got, exp)


@pytest.mark.gpu
def test_hip_transform_on_extracted_descriptors_and_search(pkg, oracle, tmp_path):
    """extract -> ComputeBoW -> SearchByBoW entirely through liborbx, against the oracle chain"""
    img = synth.image(7, 752, 480)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(752, 480))
    kp, d = ex(img)
    par, leaf, nd, w = synth.vocab_tree(8, 10, 3, data=d)
    path = os.path.join(tmp_path, "voc.txt")
    synth.write_vocab_text(path, 10, 3, par, leaf, nd, w)
    voc = pkg.ORBVocabulary.loadFromTextFile(path)
    assert voc.info() == dict(k=10, L=3, nodes=1111, words=1000)
    ovoc = oracle.Vocabulary(path=path)
    rng = np.random.Generator(np.random.PCG64(9))
    perm = rng.permutation(len(d)); d2 = synth.flip_bits(rng, d, 0.05)[perm]
    sides = []
    for desc_, kp_ in ((d2, kp[perm]), (d, kp)):
        g = voc.transform(desc_, 1); o = ovoc.transform(desc_, 1)
        _cmp(g, o)
        sides.append(dict(desc=desc_, node_id=g["fv_node_id"], node_off=g["fv_node_off"], feat=g["fv_feat"],
                          flag=np.ones(len(desc_), np.uint8), angle=kp_["angle"]))
    got, n = pkg.ORBmatcher(0.75, True).SearchByBoW(sides[0], sides[1])
    exp, en = oracle.search_by_bow_kf_f(sides[0], sides[1], 0.75, True)
    assert n == en and (got == exp).all() and n > 100


@pytest.mark.gpu
def test_hip_vocab_edge_cases(pkg, oracle):
    data = _data(11, 50)
    par, leaf, nd, w = synth.vocab_tree(12, 5, 2, data=data)
    voc = pkg.ORBVocabulary(5, 2, par, leaf, nd, w)
    t = voc.transform(np.zeros((0, 32), np.uint8))
    assert len(t["bow_id"]) == 0 and t["fv_node_off"].tolist() == [0]
    # all words stopped: empty vectors
    vz = pkg.ORBVocabulary(5, 2, par, leaf, nd, np.zeros_like(w))
    t = vz.transform(data, 1)
    assert len(t["bow_id"]) == 0 and len(t["fv_node_id"]) == 0 and (t["word_weight"] == 0).all()
    _cmp(t, oracle.Vocabulary(5, 2, par, leaf, nd, np.zeros_like(w)).transform(data, 1))
    # identical features: one word, weight added n times in sequence, then normalised to exactly 1.0
    same = np.repeat(data[:1], 33, axis=0)
    _cmp(voc.transform(same, 1), oracle.Vocabulary(5, 2, par, leaf, nd, w).transform(same, 1))
    with pytest.raises(pkg.OrbxError):            # child before its parent
        bad = par.copy(); bad[0] = 3
        pkg.ORBVocabulary(5, 2, bad, leaf, nd, w)
    with pytest.raises(pkg.OrbxError):
        pkg.ORBVocabulary(25, 2, par, leaf, nd, w)  # k > 20 (reference loader limit)
    with pytest.raises(pkg.OrbxError):
        pkg.ORBVocabulary.loadFromTextFile("/nonexistent/ORBvoc.txt")


@pytest.mark.gpu
def test_hip_device_resident_bow_chain(pkg, oracle):
    """BASELINE config 3 at throughput: extract_batch_device -> orbx_bow_transform_batch_device ->
    orbx_bowdb_search_batch_device without leaving the device, against the oracle per frame / per pair"""
    import torch
    W, H, B = 752, 480, 5
    imgs = [synth.image(40 + i, W, H) for i in range(B)]
    imgs[3] = np.full((H, W), 90, np.uint8)          # a featureless frame inside the batch
    pitch = 768
    host = np.zeros((B, H, pitch), np.uint8)
    for i in range(B): host[i, :, :W] = imgs[i]
    d_img = torch.from_numpy(host).cuda()
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(W, H), max_batch=B)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), device="cuda"); d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream(); st = stream.cuda_stream
    ex.extract_batch_device(d_img.data_ptr(), H * pitch, pitch, B, W, H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), st)
    stream.synchronize()
    n = d_n.cpu().numpy(); desc = d_desc.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    assert n[3] == 0 and n[0] > 500
    par, leaf, nd, w = synth.vocab_tree(41, 10, 4, stop_frac=0.02, data=desc[0, :n[0]])
    voc = pkg.ORBVocabulary(10, 4, par, leaf, nd, w); ovoc = oracle.Vocabulary(10, 4, par, leaf, nd, w)
    fr = pkg.BowFrames(B, cap)
    fr.transform(voc, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), B, 2, st)
    frames = []
    for i in range(B):
        got = fr.read(i, st)
        exp = ovoc.transform(desc[i, :n[i]], 2)
        for k_ in ("bow_id", "fv_node_id", "fv_node_off", "fv_feat"):
            assert (got[k_] == exp[k_]).all(), (i, k_)
        assert got["bow_val"].tobytes() == exp["bow_val"].tobytes(), i       # doubles, bit-exact
        ang = np.frombuffer(kps[i, :n[i]].tobytes(), dtype=pkg.KP_DTYPE)["angle"].copy()
        frames.append(dict(desc=desc[i, :n[i]], node_id=exp["fv_node_id"], node_off=exp["fv_node_off"], feat=exp["fv_feat"],
                           flag=np.zeros(n[i], np.uint8), angle=ang))
    # a small keyframe set made from frame 0 and frame 1
    rng = np.random.Generator(np.random.PCG64(42))
    kfs = []
    for j in range(7):
        base = frames[j % 2]
        perm = rng.permutation(len(base["desc"])); dk = synth.flip_bits(rng, base["desc"], 0.07)[perm]
        t = ovoc.transform(dk, 2)
        kfs.append(dict(desc=dk, node_id=t["fv_node_id"], node_off=t["fv_node_off"], feat=t["fv_feat"],
                        flag=(rng.random(len(dk)) < 0.7).astype(np.uint8), angle=base["angle"][perm]))
    db = pkg.BowDatabase(kfs)
    d_match = torch.full((B, len(kfs), cap), -7, dtype=torch.int32, device="cuda"); d_nm = torch.zeros((B, len(kfs)), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    expect = [[oracle.search_by_bow_kf_f(kf, frames[i], 0.75, True) for kf in kfs] for i in range(B)]
    import os
    for form in ("table", "wave"):       # the throughput and the latency form of the search kernel (orbx_bow.hip: bow_launch)
        pkg.orbx.debug_set_bow_form(form)
        try:
            d_match.fill_(-7); d_nm.zero_()
            torch.cuda.synchronize()
            fr.search(db, B, d_match.data_ptr(), d_nm.data_ptr(), 0.75, True, st)
            stream.synchronize()
        finally:
            pkg.orbx.debug_set_bow_form("auto")
        m = d_match.cpu().numpy(); nm = d_nm.cpu().numpy()
        total = 0
        for i in range(B):
            for j in range(len(kfs)):
                exp, en = expect[i][j]
                assert nm[i, j] == en, (form, i, j, nm[i, j], en)
                assert (m[i, j, :n[i]] == exp).all(), (form, i, j)
                total += en
        assert total > 500 and (nm[3] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["huge_nodes", "many_nodes"])
def test_bow_midsize_multipass(pkg, oracle, shape):
    """64 keyframes x 1000 features against one frame, both kernel forms (moved in from tools/soak_bow.py): `huge_nodes` = 6
    vocabulary nodes, so a node pair holds ~25 000 distances, larger than the 4096-entry LDS table (wave walk inside the table
    form) next to nodes that fill several table passes; `many_nodes` = 700 node ids, more than one 256-node chunk per pair."""
    rng = np.random.Generator(np.random.PCG64(9090 + len(shape)))
    nn = 6 if shape == "huge_nodes" else 700
    ids = np.sort(rng.choice(100000, nn, replace=False)).astype(np.uint32)
    p = rng.gamma(0.6 if shape == "huge_nodes" else 1.0, 1.0, nn) + 1e-9; p /= p.sum()
    base = rng.integers(0, 256, (200, 32), dtype=np.uint8)      # descriptor prototypes: near-duplicates give ties and steals

    def featset(n, flag_p):
        proto = rng.integers(0, len(base), n)
        node = ids[(proto * 7919 + rng.integers(0, 2, n)) % nn] if shape == "many_nodes" else ids[rng.choice(nn, n, p=p)]
        d = synth.flip_bits(rng, base[proto], 0.06)
        order = np.argsort(node, kind="stable")
        u, counts = np.unique(node, return_counts=True)
        off = np.zeros(len(u) + 1, np.int32); off[1:] = np.cumsum(counts)
        return dict(desc=d, node_id=u.astype(np.uint32), node_off=off, feat=order.astype(np.uint32),
                    flag=(rng.random(n) < flag_p).astype(np.uint8), angle=rng.uniform(0, 360, n).astype(np.float32))
    frame = featset(1000, 0.0)
    kfs = [featset(int(rng.integers(700, 1100)), 0.7) for _ in range(64)]
    expect = [oracle.search_by_bow_kf_f(kf, frame, 0.75, True) for kf in kfs]
    db = pkg.BowDatabase(kfs)
    try:
        for form in ("table", "wave"):
            pkg.orbx.debug_set_bow_form(form)
            m, n = db.search(frame, 0.75, True)
            for j, (exp, en) in enumerate(expect):
                assert n[j] == en, (form, j, int(n[j]), en)
                assert (m[j] == exp).all(), (form, j)
    finally:
        pkg.orbx.debug_set_bow_form("auto")
    assert sum(en for _, en in expect) > 64 * 20
    if shape == "huge_nodes":
        a = np.diff(kfs[0]["node_off"]).max(); b = np.diff(frame["node_off"]).max()
        assert int(a) * int(b) > 4096          # at least one node pair really exceeds the LDS table
