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


def test_drift_vectors_projected_nsr_and_displacement_pix():
    """lib.py:375-406 with a projected nsr: without nansat the coordinates in the destination SRS come from the
    images' own transform_points(x, y, 0, nsr) (the call pmlib.py:473-478 makes); lib.py:103-121 for the pixel
    displacement.  Expected values written out by hand from the affine stand-in."""
    from sea_ice_drift_amd.domain import ArrayNansat
    img = np.ones((64, 64), dtype=np.uint8)
    n1 = ArrayNansat(img, origin=(3.0, 4.0), matrix=((0.5, 0.0), (0.0, -0.25)), dst_scale=1000.0)
    n2 = ArrayNansat(img, origin=(3.5, 4.5), matrix=((0.5, 0.0), (0.0, -0.25)), dst_scale=1000.0)
    x1, y1 = np.array([10.0, 20.0]), np.array([8.0, 12.0])
    x2, y2 = x1 + 3.0, y1 - 2.0
    u, v, lon1, lat1, lon2, lat2 = lib.get_drift_vectors(n1, x1, y1, n2, x2, y2, nsr=lib.NSR('+proj=stere +lat_0=90'))
    np.testing.assert_array_equal(lon1, 3.0 + 0.5 * x1)
    np.testing.assert_array_equal(lat2, 4.5 - 0.25 * y2)
    # destination units = lon/lat * 1000: (0.5 * 3 + 0.5) * 1000 = 2000 east, (0.25 * 2 + 0.5) * 1000 = 1000 north
    np.testing.assert_allclose(u, [2000.0, 2000.0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v, [1000.0, 1000.0], rtol=0, atol=1e-9)
    dx, dy = lib.get_displacement_pix(n1, x1, y1, n2, x2, y2)
    np.testing.assert_allclose(dx, [4.0, 4.0], rtol=0, atol=1e-12)      # 3 px + 0.5 deg / (0.5 deg/px)
    np.testing.assert_allclose(dy, [-4.0, -4.0], rtol=0, atol=1e-12)    # -2 px + 0.5 deg / (-0.25 deg/px)
    assert lib.AVG_EARTH_RADIUS == 6371

    class NoDst(object):                       # an image object without the destination-SRS argument
        def transform_points(self, x, y, DstToSrc=0):
            return np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    with pytest.raises(NotImplementedError):
        lib.get_drift_vectors(NoDst(), x1, y1, NoDst(), x2, y2, nsr=lib.NSR('+proj=stere'))


def test_projected_nsr_is_recognised_without_the_placeholders_srs_attribute(monkeypatch):
    """nansat's NSR is an osr.SpatialReference - it has ``wkt`` / ``IsGeographic`` and no ``srs``.  Such
    an object must select projected units too, and with nansat installed EVERY nsr goes through the reference's own
    ``Domain(nsr, '-te -10 -10 10 10 -tr 1 1')`` call (lib.py:394-399)."""
    import sys
    import types
    from sea_ice_drift_amd.domain import ArrayNansat

    class OsrLike(object):                       # no .srs
        def __init__(self, wkt, geographic):
            self.wkt, self._geo = wkt, geographic

        def IsGeographic(self):
            return int(self._geo)

    class WktOnly(object):
        def __init__(self, wkt):
            self.wkt = wkt
    assert lib._is_projected(OsrLike('PROJCS["stere"]', False))
    assert not lib._is_projected(OsrLike('GEOGCS["WGS 84"]', True))
    assert lib._is_projected(WktOnly('PROJCS["stere"]')) and not lib._is_projected(WktOnly('GEOGCS["WGS 84"]'))
    assert not lib._is_projected(WktOnly('+proj=longlat +datum=WGS84')) and not lib._is_projected(None)
    assert lib._is_projected(lib.NSR('+proj=stere')) and not lib._is_projected(lib.NSR())
    img = np.ones((64, 64), dtype=np.uint8)
    n1 = ArrayNansat(img, origin=(3.0, 4.0), matrix=((0.5, 0.0), (0.0, -0.25)), dst_scale=1000.0)
    n2 = ArrayNansat(img, origin=(3.5, 4.5), matrix=((0.5, 0.0), (0.0, -0.25)), dst_scale=1000.0)
    x1, y1 = np.array([10.0, 20.0]), np.array([8.0, 12.0])
    u, v = lib.get_drift_vectors(n1, x1, y1, n2, x1 + 3.0, y1 - 2.0, nsr=OsrLike('PROJCS["stere"]', False))[:2]
    np.testing.assert_allclose(u, [2000.0, 2000.0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v, [1000.0, 1000.0], rtol=0, atol=1e-9)
    u, v = lib.get_drift_vectors(n1, x1, y1, n2, x1 + 3.0, y1 - 2.0, nsr=OsrLike('GEOGCS["WGS 84"]', True))[:2]
    np.testing.assert_allclose(u, [2.0, 2.0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(v, [1.0, 1.0], rtol=0, atol=1e-12)

    # nansat "installed": a Domain that records the call and maps lon/lat to the pixels of a 100 m grid
    calls = []

    class FakeDomain(object):
        def __init__(self, nsr, ext):
            calls.append((nsr, ext))

        def transform_points(self, lon, lat, DstToSrc=0):
            assert DstToSrc == 1
            return np.asarray(lon) * 100.0 + 10.0, 10.0 - np.asarray(lat) * 100.0
    fake = types.ModuleType('nansat')
    fake.Domain = FakeDomain
    monkeypatch.setitem(sys.modules, 'nansat', fake)
    nsr = OsrLike('PROJCS["stere"]', False)
    u, v = lib.get_drift_vectors(n1, x1, y1, n2, x1 + 3.0, y1 - 2.0, nsr=nsr)[:2]
    assert calls == [(nsr, '-te -10 -10 10 10 -tr 1 1')]
    np.testing.assert_allclose(u, [200.0, 200.0], rtol=0, atol=1e-9)      # 2 degrees east on the 100-per-degree grid
    np.testing.assert_allclose(v, [100.0, 100.0], rtol=0, atol=1e-9)


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
    # u, v, lon, lat of the public call are those of the fixture's matched key points (lib.py:375-406: pixel
    # differences on the '-te -10 -10 10 10 -tr 1 1' grid of the default lon/lat projection)
    x1, y1, x2, y2 = [g['%s_%s' % (name, tag)] for name in ('x1', 'y1', 'x2', 'y2')]
    elon1, elat1 = n1.transform_points(x1, y1)
    elon2, elat2 = n2.transform_points(x2, y2)
    for got_v, want in ((lon1, elon1), (lat1, elat1), (lon2, elon2), (lat2, elat2)):
        np.testing.assert_array_equal(got_v, want)
    np.testing.assert_array_equal(u, (elon2 + 10.0) - (elon1 + 10.0))
    np.testing.assert_array_equal(v, (10.0 - elat1) - (10.0 - elat2))
    # a projected nsr goes through the images' own transform_points(x, y, 0, nsr)
    up, vp = SeaIceDrift(n1, n2).get_drift_FT(find_key_points=(lambda f: lambda image, **k: f.pop(0))([(xy1, d1), (xy2, d2)]),
                                              domainMargin=10, ratio_test=0.75, psi=150, nsr=lib.NSR('+proj=stere +lat_0=90'), **kw)[:2]
    X1, Y1 = n1.transform_points(x1, y1, 0, lib.NSR('+proj=stere +lat_0=90'))
    X2, Y2 = n2.transform_points(x2, y2, 0, lib.NSR('+proj=stere +lat_0=90'))
    np.testing.assert_array_equal(up, (X2 + 10.0) - (X1 + 10.0))
    np.testing.assert_array_equal(vp, (10.0 - Y1) - (10.0 - Y2))
