"""Randomized parity soak of Frame::ComputeBoW on the device (orbx_vocab.hip: breadth-first descent, register / LDS bitonic sorts, sequential L1 norm)
against the CPU oracle's DBoW2 restatement: random vocabularies (k, L, stop words), feature counts on both sides of every sort-size and
chunk boundary, levelsup 0..L.  python tools/soak_vocab.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from tools import synth
from oracle import oracle_py as O
pkg = ge.load_pkg()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 2024))
t0 = time.time(); trials = feats = 0
edge = [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 2000, 2047, 2048, 2049, 4095, 4096, 4097, 6000]
while time.time() - t0 < budget:
    k = int(rng.integers(2, 13)); L = int(rng.integers(1, 6))
    while k ** L > 200000: L -= 1
    n = int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, 3000))
    data = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if rng.random() < 0.3 and n > 4:          # duplicates: several features per word
        data[rng.integers(0, n, n // 3)] = data[rng.integers(0, n, n // 3)]
    par, leaf, nd, w = synth.vocab_tree(int(rng.integers(0, 1 << 30)), k, L, stop_frac=float(rng.choice([0.0, 0.05, 0.3])), data=data)
    V = pkg.ORBVocabulary(k, L, par, leaf, nd, w); OV = O.Vocabulary(k, L, par, leaf, nd, w)
    for levelsup in {0, int(rng.integers(0, L + 2)), 4}:
        got = V.transform(data, levelsup); exp = OV.transform(data, levelsup)
        for key in exp:
            assert got[key].dtype == exp[key].dtype and got[key].shape == exp[key].shape and got[key].tobytes() == exp[key].tobytes(), (k, L, n, levelsup, key)
    trials += 1; feats += n
print("vocab soak done: %d vocabularies / feature sets (%d features), BowVector and FeatureVector byte-equal to the oracle" % (trials, feats))
