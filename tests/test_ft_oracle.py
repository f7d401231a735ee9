"""CPU tests of the matcher oracle (oracle/ft_oracle.py: restated ftlib.py:92-116) and of the host-side
Lowe filter of the product against it."""
import numpy as np

from oracle import ft_oracle as fo
from sea_ice_drift_amd import ftlib


def random_descriptors(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def test_vectorised_oracle_matches_python_loops():
    rng = np.random.default_rng(3)
    d1, d2 = random_descriptors(rng, 41), random_descriptors(rng, 67)
    d2[11] = d2[4]; d1[7] = d2[4]                     # exact duplicates: distance 0 twice, smaller index first
    d1[8] = d2[20]; d1[8, 13] ^= np.uint8(4)        # one differing bit
    i1, k1 = fo.knn2(d1, d2, chunk=16)
    i2, k2 = fo.knn2_loops(d1, d2)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(k1, k2)
    assert i1[7].tolist() == [4, 11] and k1[7].tolist() == [0, 0]
    assert i1[8, 0] == 20 and k1[8, 0] == 1


def test_degenerate_sizes():
    rng = np.random.default_rng(4)
    d1 = random_descriptors(rng, 5)
    i, k = fo.knn2(d1, random_descriptors(rng, 1))
    assert (i[:, 1] == -1).all() and (k[:, 1] == -1).all() and (i[:, 0] == 0).all()
    i, k = fo.knn2(d1, np.zeros((0, 32), np.uint8))
    assert (i == -1).all() and (k == -1).all()
    i, k = fo.knn2(np.zeros((0, 32), np.uint8), d1)
    assert i.shape == (0, 2)


def test_product_filter_equals_reference_filter():
    """sea_ice_drift_amd.ftlib.match (mask form, matcher replaced by the oracle) against the loop form of
    ftlib.py:101-116."""
    rng = np.random.default_rng(5)
    d1, d2 = random_descriptors(rng, 300), random_descriptors(rng, 280)
    d1[:60] = d2[rng.permutation(280)[:60]] ^ rng.integers(0, 2, (60, 32), dtype=np.uint8)   # some close pairs
    idx, dist = fo.knn2(d1, d2)
    p1, p2 = rng.random((300, 2)) * 1000, rng.random((280, 2)) * 1000
    exp = fo.filter_matches(idx, dist, 0.7, p1, p2)
    import pytest
    mp = pytest.MonkeyPatch()
    mp.setattr(ftlib, '_get_matches', lambda a, b, device=0, verbose=False: (idx, dist))
    got = ftlib.match(p1, d1, p2, d2, ratio_test=0.7)
    assert len(exp[0]) >= 40
    for a, b in zip(exp, got):
        np.testing.assert_array_equal(a, b)

    class KP:                                          # cv2.KeyPoint-like
        def __init__(self, xy): self.pt = (float(xy[0]), float(xy[1]))
    got2 = ftlib.get_match_coords([KP(p) for p in p1], d1, [KP(p) for p in p2], d2, ratio_test=0.7)
    mp.undo()
    for a, b in zip(exp, got2):
        np.testing.assert_array_equal(a, b)
