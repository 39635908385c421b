"""MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:266-340)."""
import numpy as np
import pytest

from tools import synth


def _numpy_best(desc):
    n = len(desc)
    if n == 0:
        return -1
    d = synth.hamming_matrix(desc, desc)
    med = np.sort(d, axis=1)[:, int(0.5 * (n - 1))]
    return int(np.argmin(med))          # first minimum


def _cases(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    sets = []
    for n in [0, 1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 65, 100, 256] + list(rng.integers(2, 40, 60)):
        base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        obs = synth.flip_bits(rng, np.repeat(base, n, axis=0), float(rng.uniform(0.01, 0.3))) if n else np.zeros((0, 32), np.uint8)
        if n > 3 and rng.random() < 0.3:
            obs[rng.integers(0, n)] = rng.integers(0, 256, 32, dtype=np.uint8)      # an outlier observation
        if n > 3 and rng.random() < 0.3:
            obs[1] = obs[0]                                                          # duplicates -> ties
        sets.append(obs)
    return sets


def test_oracle_vs_numpy(oracle):
    for obs in _cases(1):
        assert oracle.distinctive_descriptor(obs) == _numpy_best(obs)


@pytest.mark.gpu
def test_hip_parity(pkg, oracle):
    sets = _cases(2)
    got = pkg.ComputeDistinctiveDescriptors(sets)
    exp = np.array([oracle.distinctive_descriptor(s) for s in sets])
    assert (got == exp).all(), np.nonzero(got != exp)[0]
    assert len(pkg.ComputeDistinctiveDescriptors([])) == 0
    with pytest.raises(pkg.OrbxError):
        pkg.ComputeDistinctiveDescriptors([np.zeros((257, 32), np.uint8)])
