"""GPU parity of the Hamming k=2 matcher (include/sid_ft.h, replaces ftlib.py:92-99) against the oracle."""
import numpy as np
import pytest

from oracle import ft_oracle as fo
from sea_ice_drift_amd import _capi, ftlib

pytestmark = pytest.mark.gpu


def descriptors(rng, n, like=None, flips=0):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if like is not None:                               # noisy copies of existing descriptors: realistic near matches
        src = like[rng.integers(0, len(like), n)]
        noise = np.zeros((n, 256), dtype=np.uint8)
        for r in range(n):
            noise[r, rng.permutation(256)[:rng.integers(0, flips + 1)]] = 1
        d = src ^ np.packbits(noise, axis=1)
    return d


@pytest.mark.parametrize('n1,n2', [(1, 2), (257, 255), (1000, 1300), (3000, 2500)])
def test_knn2_matches_oracle(n1, n2):
    rng = np.random.default_rng(100 + n1)
    d2 = descriptors(rng, n2)
    d1 = descriptors(rng, n1, like=d2, flips=60)
    if n2 > 20:
        d2[17] = d2[5]                                 # duplicates in the train set: ties at equal distance
        d1[0] = d2[5]
    idx, dist = _capi.ft_knn2(d1, d2)
    eidx, edist = fo.knn2(d1, d2)
    np.testing.assert_array_equal(dist, edist)         # bit-exact integer distances
    np.testing.assert_array_equal(idx, eidx)           # and indices (ties: smaller index first)


@pytest.mark.parametrize('n1,n2', [(4097, 4100), (5000, 6701)])
def test_mfma_form_equals_the_valu_form_and_the_oracle(monkeypatch, n1, n2):
    """From 2^24 pairs on the distances come from v_mfma_i32_16x16x64_i8 (|a xor b| = |a| + |b| - 2 a.b on 0/1 bytes);
    sizes that are no multiples of the 256-descriptor tiles, duplicates (ties: smaller index first), all-zero and all-one
    descriptors (the extremes of the key range).  SID_FT_NO_MFMA=1 runs the one-thread-per-query kernel on the same input."""
    rng = np.random.default_rng(n1)
    d2 = descriptors(rng, n2)
    d1 = descriptors(rng, n1, like=d2, flips=40)
    d2[17] = d2[5]; d2[n2 - 1] = d2[5]; d1[0] = d2[5]
    d2[100] = 0; d2[101] = 255; d1[1] = 0; d1[2] = 255; d1[3] = d2[n2 - 2]
    idx, dist = _capi.ft_knn2(d1, d2)
    monkeypatch.setenv('SID_FT_NO_MFMA', '1')
    vidx, vdist = _capi.ft_knn2(d1, d2)
    np.testing.assert_array_equal(dist, vdist)
    np.testing.assert_array_equal(idx, vidx)
    m = 600                                            # (the NumPy oracle is slow: the first and the last queries)
    for sl in (slice(0, m), slice(n1 - m, n1)):
        eidx, edist = fo.knn2(d1[sl], d2)
        np.testing.assert_array_equal(dist[sl], edist)
        np.testing.assert_array_equal(idx[sl], eidx)
    assert dist[1, 0] == 0 and idx[1, 0] == 100 and dist[2, 0] == 0 and idx[2, 0] == 101 and idx[0, 0] == 5 and idx[0, 1] == 17


def test_single_train_descriptor_and_empty_sets():
    rng = np.random.default_rng(7)
    d1 = descriptors(rng, 10)
    idx, dist = _capi.ft_knn2(d1, descriptors(rng, 1))
    assert (idx[:, 1] == -1).all() and (dist[:, 1] == -1).all() and (idx[:, 0] == 0).all()
    idx, dist = _capi.ft_knn2(d1, np.zeros((0, 32), np.uint8))
    assert (idx == -1).all() and (dist == -1).all()
    idx, dist = _capi.ft_knn2(np.zeros((0, 32), np.uint8), d1)
    assert idx.shape == (0, 2)
    with pytest.raises(ValueError):
        ftlib.get_match_coords(np.zeros((10, 2)), d1, np.zeros((1, 2)), descriptors(rng, 1))


def test_get_match_coords_end_to_end():
    """Reference call shape (ftlib.py:64-90): key points + descriptors in, x1, y1, x2, y2 out."""
    rng = np.random.default_rng(11)
    n2 = 4000
    d2 = descriptors(rng, n2)
    p2 = rng.random((n2, 2)) * 4000
    sel = rng.permutation(n2)[:1500]
    d1 = d2[sel].copy()
    flip = np.zeros((len(sel), 256), dtype=np.uint8)
    for r in range(len(sel)):
        flip[r, rng.permutation(256)[:rng.integers(0, 30)]] = 1
    d1 ^= np.packbits(flip, axis=1)
    p1 = p2[sel] + rng.normal(0, 3, (len(sel), 2))
    x1, y1, x2, y2 = ftlib.get_match_coords(p1, d1, p2, d2, ratio_test=0.7)
    eidx, edist = fo.knn2(d1, d2)
    ex1, ey1, ex2, ey2 = fo.filter_matches(eidx, edist, 0.7, p1, p2)
    for a, b in zip((x1, y1, x2, y2), (ex1, ey1, ex2, ey2)):
        np.testing.assert_array_equal(a, b)
    assert len(x1) > 1200                              # nearly every planted pair survives the ratio test
    assert np.abs(x2 - x1).max() < 20 and np.abs(y2 - y1).max() < 20
