"""Randomized parity soak of the per-call / batched / resident matcher forms (orb-slam2_amd/csrc/orbx_match.hip: k_match<0..2>)
against the CPU oracle: feature counts 1 .. 3000, vocabularies from one node to thousands (nodes beyond one 64 x 64 tile, beyond
the register form, row splits of SearchForTriangulation), skewed node sizes, near-duplicate descriptors (ties), random map-point
flags, stereo / mono mixes, bOnlyStereo, batches of 1 .. 24 second sides of different sizes, empty FeatureVectors, both
orientation settings; host-pointer and resident (orbx_kf_*) entry points on the same inputs.
Run on the GPU box: python tools/soak_match.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from tools import synth
from oracle import oracle_py as O
pkg = ge.load_pkg()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 9001))
sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32); sg = (sf * sf).astype(np.float32)


def featset(n, node_of, desc, flag_p, listed):
    """listed: fraction of the features that appear in the FeatureVector at all (stopped words are in no node)"""
    keep = np.nonzero(rng.random(n) < listed)[0]
    order = keep[np.argsort(node_of[keep], kind="stable")]
    ids, counts = np.unique(node_of[keep], return_counts=True)
    off = np.zeros(len(ids) + 1, np.int32); off[1:] = np.cumsum(counts)
    return dict(desc=desc, node_id=ids.astype(np.uint32), node_off=off, feat=order.astype(np.uint32),
                flag=(rng.random(n) < flag_p).astype(np.uint8), angle=rng.uniform(0, 360, n).astype(np.float32),
                x=rng.uniform(0, 1241, n).astype(np.float32), y=rng.uniform(0, 376, n).astype(np.float32),
                octave=rng.integers(0, 8, n).astype(np.int32), u_right=np.where(rng.random(n) < rng.choice([0.0, 0.3, 1.0]), 5.0, -1.0).astype(np.float32))


t0 = time.time(); trial = 0; npairs = 0; nm = [0, 0, 0]
while time.time() - t0 < budget:
    trial += 1
    shape = trial % 6
    if shape == 0: nn = int(rng.integers(1, 6))                 # a few huge nodes
    elif shape == 1: nn = int(rng.integers(300, 2500))
    elif shape == 2: nn = int(rng.integers(50, 150))            # ORB-SLAM2's regime
    else: nn = int(rng.integers(1, 400))
    skew = rng.choice([0.3, 1.0, 3.0])
    pr = rng.gamma(skew, 1.0, nn) + 1e-9; pr /= pr.sum()
    ids = np.sort(rng.choice(100000, nn, replace=False))
    base = rng.integers(0, 256, (max(nn // 3, 1), 32), dtype=np.uint8)
    nmax = 3000 if shape else 6000

    def side(n):
        node = ids[rng.choice(nn, n, p=pr)]
        d = synth.flip_bits(rng, base[rng.integers(0, len(base), n)], float(rng.choice([0.0, 0.02, 0.08, 0.3])))
        return featset(n, node, d, float(rng.choice([0.0, 0.3, 0.7, 1.0])), float(rng.choice([0.0, 0.5, 0.95, 1.0])))
    A = side(int(rng.integers(1, nmax)))
    nb = int(rng.integers(1, 25)) if trial % 3 else 1
    Bs = [side(int(rng.integers(1, nmax))) for _ in range(nb)]
    ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0])); ori = bool(trial % 2); only_stereo = bool(trial % 5 == 0)
    Fs = [(np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) * np.float32(rng.uniform(0.5, 2)) + np.float32(1e-4) * rng.normal(0, 1, (3, 3)).astype(np.float32)).astype(np.float32)
          for _ in range(nb)]
    eps = [(float(rng.uniform(-500, 1700)), float(rng.uniform(0, 376))) for _ in range(nb)]
    tag = f"trial {trial}: nA {len(A['desc'])} nb {nb} nodes {nn} skew {skew} ratio {ratio} ori {ori} only_stereo {only_stereo}"
    m = pkg.ORBmatcher(ratio, ori)
    exp_ff = [O.search_by_bow_kf_f(B, A, ratio, ori) for B in Bs]
    exp_kk = [O.search_by_bow_kf_kf(A, B, ratio, ori) for B in Bs]
    exp_tr = [O.search_for_triangulation(A, B, Fs[i], eps[i][0], eps[i][1], sf, sg, 0.6, ori, only_stereo) for i, B in enumerate(Bs)]
    # host pointers: batched forms, and one-pair calls for the first second side
    g, n = m.SearchByBoWBatch(Bs, A)
    for i in range(nb):
        assert n[i] == exp_ff[i][1] and (g[i] == exp_ff[i][0]).all(), "kf_f batch " + tag
    g, n = m.SearchByBoWKeyFrames(A, Bs)
    for i in range(nb):
        assert n[i] == exp_kk[i][1] and (g[i] == exp_kk[i][0]).all(), "kf_kf batch " + tag
    got = m.SearchForTriangulationBatch(A, Bs, Fs, eps, sf, sg, bOnlyStereo=only_stereo)
    for i in range(nb):
        assert got[i].shape == exp_tr[i].shape and (got[i] == exp_tr[i]).all(), "triangulation batch " + tag
    g1, n1 = m.SearchByBoW(Bs[0], A)
    assert n1 == exp_ff[0][1] and (g1 == exp_ff[0][0]).all(), "kf_f " + tag
    # resident keyframes (greedy searches need every node <= 4096 second-side features there)
    max_node = max(int(np.diff(S["node_off"]).max()) if len(S["node_id"]) else 0 for S in [A] + Bs)
    dA = pkg.DeviceKeyFrame(A); dBs = [pkg.DeviceKeyFrame(B) for B in Bs]
    got = m.SearchForTriangulationResident(dA, A["flag"], dBs, [B["flag"] for B in Bs], Fs, eps, sf, sg, bOnlyStereo=only_stereo)
    for i in range(nb):
        assert got[i].shape == exp_tr[i].shape and (got[i] == exp_tr[i]).all(), "triangulation resident " + tag
    if max_node <= 4096:
        g, n = m.SearchByBoWKeyFramesResident(dA, A["flag"], dBs, [B["flag"] for B in Bs])
        for i in range(nb):
            assert n[i] == exp_kk[i][1] and (g[i] == exp_kk[i][0]).all(), "kf_kf resident " + tag
        g1, n1 = m.SearchByBoWResident(dBs[0], Bs[0]["flag"], dA)
        assert n1 == exp_ff[0][1] and (g1 == exp_ff[0][0]).all(), "kf_f resident " + tag
    del dA, dBs
    npairs += nb
    nm[0] += sum(e[1] for e in exp_ff); nm[1] += sum(e[1] for e in exp_kk); nm[2] += sum(len(e) for e in exp_tr)
    if trial % 20 == 0:
        print(f"{time.time() - t0:6.1f}s trials {trial} pairs {npairs}", flush=True)
print(f"matcher soak done: {trial} trials, {npairs} pairs x 3 searches x (host-pointer batch + resident), all equal to the oracle; "
      f"matches compared: SearchByBoW(KF,F) {nm[0]}, (KF,KF) {nm[1]}, SearchForTriangulation {nm[2]}")
