"""Feature-tracking pipeline of the product (sea_ice_drift_amd.ftlib.feature_tracking: domain filter, matcher +
Lowe filter, drift filter, least-squares filter) against the outputs of the reference's own
ftlib.feature_tracking (fixture g7: synthetic key points, brute-force matcher injected - ftlib.py:241-285)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from oracle import ft_oracle as fo
from sea_ice_drift_amd import ftlib, lib
from sea_ice_drift_amd.seaicedrift import SeaIceDrift

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_module():
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(HERE, 'golden', 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ['make_golden']
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def run_pipeline(mg, timed, **extra):
    n1, n2, xy1, d1, xy2, d2 = mg.g7_inputs(timed)
    feeds = [(xy1, d1), (xy2, d2)]

    def finder(image, **kwargs):
        return feeds.pop(0)                         # (N, 2) arrays of (x, y) stand for the cv2.KeyPoint lists
    kw = dict(max_speed=0.3) if timed else dict(max_drift=25000.0)
    kw.update(extra)
    return n1, n2, ftlib.feature_tracking(n1, n2, find_key_points=finder, domainMargin=10, ratio_test=0.75, psi=150, **kw)


@pytest.mark.parametrize('timed', [False, True])
def test_host_filters_equal_reference(monkeypatch, timed):
    """CPU: the matcher replaced by the oracle, everything else is the product's host code."""
    monkeypatch.setattr(ftlib, '_get_matches', lambda d1, d2, device=0, verbose=False: fo.knn2(d1, d2))
    mg = golden_module()
    g = np.load(os.path.join(HERE, 'golden', 'g7_feature_tracking.npz'))
    _, _, got = run_pipeline(mg, timed)
    tag = 'timed' if timed else 'untimed'
    for name, v in zip(('x1', 'y1', 'x2', 'y2'), got):
        np.testing.assert_array_equal(v, g['%s_%s' % (name, tag)])


def test_missing_drift_limit_raises(monkeypatch):
    monkeypatch.setattr(ftlib, '_get_matches', lambda d1, d2, device=0, verbose=False: fo.knn2(d1, d2))
    mg = golden_module()
    with pytest.raises(ValueError):
        run_pipeline(mg, False, max_drift=None)


def test_drift_vectors_default_projection():
    mg = golden_module()
    n1, n2, xy1, _, xy2, _ = mg.g7_inputs(False)
    u, v, lon1, lat1, lon2, lat2 = lib.get_drift_vectors(n1, xy1[:50, 0], xy1[:50, 1], n2, xy2[:50, 0], xy2[:50, 1])
    np.testing.assert_allclose(u, lon2 - lon1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(v, lat2 - lat1, rtol=0, atol=1e-12)
    km = lib.get_displacement_km(n1, xy1[:50, 0], xy1[:50, 1], n2, xy2[:50, 0], xy2[:50, 1])
    assert km.shape == (50,) and (km >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('timed', [False, True])
def test_feature_tracking_on_gpu_equals_reference(timed):
    """GPU: the whole driver with the HIP matcher behind it."""
    mg = golden_module()
    g = np.load(os.path.join(HERE, 'golden', 'g7_feature_tracking.npz'))
    n1, n2, got = run_pipeline(mg, timed)
    tag = 'timed' if timed else 'untimed'
    for name, v in zip(('x1', 'y1', 'x2', 'y2'), got):
        np.testing.assert_array_equal(v, g['%s_%s' % (name, tag)])
    # and through the public class (seaicedrift.py:42-60)
    n1, n2, xy1, d1, xy2, d2 = mg.g7_inputs(timed)
    feeds = [(xy1, d1), (xy2, d2)]
    kw = dict(max_speed=0.3) if timed else dict(max_drift=25000.0)
    u, v, lon1, lat1, lon2, lat2 = SeaIceDrift(n1, n2).get_drift_FT(find_key_points=lambda image, **k: feeds.pop(0),
                                                                   domainMargin=10, ratio_test=0.75, psi=150, **kw)
    assert len(u) == len(got[0]) and np.isfinite(u).all()
