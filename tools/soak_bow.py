"""Randomized parity soak of the two SearchByBoW kernel forms (table + row fixpoint / wave walk) on synthetic
FeatureVectors with wild node-size distributions: many nodes (several 256-node chunks), nodes larger than the LDS
table (wave fallback), several passes per chunk, empty nodes, near-duplicate descriptors (ties, steals).
Run on the GPU box: python tools/soak_bow.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from tools import synth
from oracle import oracle_py as O
pkg = ge.load_pkg()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 4242))


def featset(n, node_of, desc, flag_p):
    order = np.argsort(node_of, kind="stable")
    ids, counts = np.unique(node_of, return_counts=True)
    off = np.zeros(len(ids) + 1, np.int32); off[1:] = np.cumsum(counts)
    return dict(desc=desc, node_id=ids.astype(np.uint32), node_off=off, feat=order.astype(np.uint32),
                flag=(rng.random(n) < flag_p).astype(np.uint8), angle=rng.uniform(0, 360, n).astype(np.float32),
                x=np.zeros(n, np.float32), y=np.zeros(n, np.float32), octave=np.zeros(n, np.int32), u_right=np.full(n, -1, np.float32))


t0 = time.time(); trial = 0; npairs = 0
while time.time() - t0 < budget:
    trial += 1
    shape = trial % 6
    nA = int(rng.integers(1, 3000)); nB = int(rng.integers(1, 3000))
    if shape == 0: nn = int(rng.integers(1, 8))                 # few huge nodes: wave fallback / several passes
    elif shape == 1: nn = int(rng.integers(300, 2500))          # more than one 256-node chunk
    elif shape == 2: nn = int(rng.integers(20, 200))
    else: nn = int(rng.integers(1, 600))
    skew = rng.choice([0.3, 1.0, 3.0])
    p = rng.gamma(skew, 1.0, nn) + 1e-9; p /= p.sum()
    ids = np.sort(rng.choice(100000, nn, replace=False))
    base = rng.integers(0, 256, (max(nn // 3, 1), 32), dtype=np.uint8)   # descriptor prototypes: many near-duplicates
    def side(n):
        node = ids[rng.choice(nn, n, p=p)]
        d = synth.flip_bits(rng, base[rng.integers(0, len(base), n)], float(rng.choice([0.0, 0.02, 0.08, 0.3])))
        return featset(n, node, d, float(rng.choice([0.3, 0.7, 1.0])))
    A, B = side(nA), side(nB)
    if trial % 7 == 0:   # drop some nodes from one side: unshared nodes
        keep = rng.random(len(A["node_id"])) < 0.6
        if keep.any():
            sel = np.concatenate([A["feat"][A["node_off"][i]:A["node_off"][i + 1]] for i in np.nonzero(keep)[0]])
            A = featset(len(sel), np.repeat(A["node_id"][keep], np.diff(A["node_off"])[keep]), A["desc"][sel], 0.8)
    ratio = float(rng.choice([0.1, 0.3, 0.6, 0.75, 0.9, 1.0])); ori = bool(trial % 2)
    e1, n1 = O.search_by_bow_kf_f(A, B, ratio, ori)
    e2, n2 = O.search_by_bow_kf_kf(A, B, ratio, ori)
    for form in ("table", "wave"):
        pkg.orbx.debug_set_bow_form(form)
        m = pkg.ORBmatcher(ratio, ori)
        g1, gn1 = m.SearchByBoW(A, B)
        Bk = dict(B); Bk["kind"] = "keyframe"
        g2, gn2 = m.SearchByBoW(A, Bk)
        tag = f"trial {trial} form {form} nA {nA} nB {nB} nodes {nn} skew {skew} ratio {ratio} ori {ori}"
        assert gn1 == n1 and (g1 == e1).all(), "kf_f " + tag
        assert gn2 == n2 and (g2 == e2).all(), "kf_kf " + tag
    npairs += 1
    if trial % 20 == 0: print(f"{time.time() - t0:6.1f}s trials {trial} matches(last) {n1}/{n2}", flush=True)
print(f"bow soak done: {npairs} random pairs x 2 searches x 2 kernel forms, all equal to the oracle")
