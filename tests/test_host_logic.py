"""CPU tests of the host side of pattern_matching (prelude / postlude / option handling)
against fixture G4 = the reference's own pattern_matching run end to end."""
import os

import numpy as np
import pytest
from scipy import ndimage as nd

from sea_ice_drift_amd import lib as mylib, pmlib as my, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat
from sea_ice_drift_amd.seaicedrift import SeaIceDrift
from tests.golden import make_golden as mg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KW = dict(img_size=34, angles=list(range(-3, 4)))


@pytest.fixture(scope='module')
def g4():
    g = np.load(os.path.join(GOLD, 'g4_pattern_matching.npz'))
    n1, n2, c1, r1, c2, r2, lon_g, lat_g = mg.g4_inputs()
    assert syn.sha256(n1[1], n2[1]) == str(g['pair_sha'])
    for a, b in ((c1, g['c1']), (r1, g['r1']), (c2, g['c2']), (r2, g['r2']), (lon_g, g['lon_g'])):
        np.testing.assert_array_equal(a, b)
    return g, (n1, n2, c1, r1, c2, r2, lon_g, lat_g)


def test_prelude_builds_the_reference_kernel_inputs(g4):
    g, (n1, n2, c1, r1, c2, r2, lon_g, lat_g) = g4
    pre = my.pm_prelude(lon_g, lat_g, n1, c1, r1, n2, c2, r2, **KW)
    np.testing.assert_array_equal(pre['c2fg'], g['fg_c2'])
    np.testing.assert_array_equal(pre['r2fg'], g['fg_r2'])
    np.testing.assert_array_equal(pre['brd2'], g['fg_border'])
    gpi = pre['gpi']
    for mine, ref in (('c1pm1i', 'k_c1'), ('r1pm1i', 'k_r1'), ('c2fg', 'k_c2fg'), ('r2fg', 'k_r2fg'),
                      ('brd2', 'k_border')):
        np.testing.assert_array_equal(pre[mine][gpi], g[ref])
    assert pre['alpha0'] == float(g['alpha0'])
    assert abs(pre['alpha0'] + 2.0) < 1e-9
    assert 20 <= pre['brd2'][gpi].min() and pre['brd2'][gpi].max() <= 50


def test_postlude_reproduces_the_reference_grids(g4):
    g, (n1, n2, c1, r1, c2, r2, lon_g, lat_g) = g4
    pre = my.pm_prelude(lon_g, lat_g, n1, c1, r1, n2, c2, r2, **KW)
    gpi = pre['gpi'].reshape(lon_g.shape)
    # rebuild the (N,5) block the reference's workers returned from its output grids
    dci = (pre['c2pm1'] - pre['c2pm1i'])[pre['gpi']]
    dri = (pre['r2pm1'] - pre['r2pm1i'])[pre['gpi']]
    c2pm2, r2pm2 = n2.transform_points(g['lon2'][gpi], g['lat2'][gpi], 1)
    res = np.stack([c2pm2 - dci, r2pm2 - dri, g['a'][gpi], g['r'][gpi], g['h'][gpi]], axis=1)
    out = my.pm_postlude(pre, res, n2)
    for name, arr in zip(('u', 'v', 'a', 'r', 'h', 'lon2', 'lat2'), out):
        np.testing.assert_allclose(arr, g[name], rtol=0, atol=1e-9, equal_nan=True, err_msg=name)
    for name, arr in zip(('a', 'r', 'h'), out[2:5]):
        np.testing.assert_array_equal(arr, g[name])


def test_empty_result_gives_nan_grids(g4):
    g, (n1, n2, c1, r1, c2, r2, lon_g, lat_g) = g4
    pre = my.pm_prelude(lon_g, lat_g, n1, c1, r1, n2, c2, r2, **KW)
    out = my.pm_postlude(pre, np.zeros((0, 5)), n2)
    assert len(out) == 7 and all(o.shape == lon_g.shape and np.isnan(o).all() for o in out)


def test_border_from_kdtree_equals_full_image_edt():
    rng = np.random.default_rng(4)
    shape = (300, 260)
    x = rng.uniform(0, shape[1] - 1, 40)
    y = rng.uniform(0, shape[0] - 1, 40)
    seed = np.zeros(shape, dtype=bool)
    seed[np.uint16(y), np.uint16(x)] = True
    dist = nd.distance_transform_edt(~seed)                       # what pmlib.py:61-77 builds
    rq = rng.integers(0, shape[0], 500)
    cq = rng.integers(0, shape[1], 500)
    np.testing.assert_array_equal(my.nearest_keypoint_distance(x, y, rq, cq), dist[rq, cq])


def test_interpolators_and_fill():
    rng = np.random.default_rng(6)
    x1, y1 = rng.uniform(0, 100, 50), rng.uniform(0, 100, 50)
    x2, y2 = 3 + 1.01 * x1 - 0.02 * y1, -2 + 0.03 * x1 + 0.99 * y1
    gx, gy = np.meshgrid(np.linspace(10, 90, 5), np.linspace(10, 90, 4))
    px, py = mylib.interpolation_poly(x1, y1, x2, y2, gx.ravel(), gy.ravel())
    np.testing.assert_allclose(px, 3 + 1.01 * gx.ravel() - 0.02 * gy.ravel(), atol=1e-9)
    np.testing.assert_allclose(py, -2 + 0.03 * gx.ravel() + 0.99 * gy.ravel(), atol=1e-9)
    nx, ny = mylib.interpolation_near(x1, y1, x2, y2, np.array([50.0, 500.0]), np.array([50.0, 500.0]))
    assert np.isfinite(nx[0]) and np.isnan(nx[1])                 # NaN outside the convex hull
    filled = mylib._fill_gpi((2, 3), np.array([1, 0, 0, 1, 0, 1], dtype=bool), np.array([1.0, 2.0, 3.0]))
    np.testing.assert_array_equal(filled, [[1, np.nan, np.nan], [2, np.nan, 3]])


def test_rotation_table_matches_reference_formula():
    t = my.rotation_table([-3, 0, 3], -3.85, 35)
    assert t.shape == (3, 4)
    a = np.radians(-3 + 3.85)
    assert t[0, 0] == np.cos(a) and t[0, 1] == np.sin(a)
    tc = 18
    np.testing.assert_array_equal(t[0, 2:], np.array([tc, tc]).dot(np.array([[np.cos(a), -np.sin(a)],
                                                                            [np.sin(a), np.cos(a)]])))


def test_unsupported_options_raise_instead_of_falling_back():
    img = np.ones((200, 200), dtype=np.uint8)
    args = ([100.0], [100.0], [100.0], [100.0], [20.0], 34, 0.0)
    for order in (6, -1, True):                      # scipy's spline orders are 0..5 (all on the device since round 6); scipy raises RuntimeError
        with pytest.raises(RuntimeError):
            my.pm_dispatch(img, img, *args, rot_order=order)
    with pytest.raises(NotImplementedError):
        my.pm_dispatch(img, img, *args, template_matcher=lambda *a: None)
    # mtype (pmlib.py:119,156): the device matcher is cv2.TM_CCOEFF_NORMED (5); any other mode would be answered with the
    # wrong correlation - an error, not silence
    for mtype in (0, 1, 2, 3, 4, 'TM_CCORR_NORMED', True):
        with pytest.raises(NotImplementedError):
            my.pm_dispatch(img, img, *args, mtype=mtype)
    assert my.TM_CCOEFF_NORMED == 5
    for ok in ({}, {'mtype': None}, {'mtype': 5}, {'rot_order': 0}, {'rot_order': 1}, {'mtype': 5, 'rot_order': 1}, {'rot_order': 3}, {'rot_order': 5}):
        angles, flags = my._sweep_options(dict(ok))
        assert angles == [-3, 0, 3] and flags == (1 | (ok.get('rot_order', 0) << 3))


def test_api_surface():
    img = np.ones((64, 64), dtype=np.uint8)
    sid = SeaIceDrift(ArrayNansat(img), ArrayNansat(img))
    assert callable(sid.get_drift_PM) and callable(sid.get_drift_FT)
    with pytest.raises(NotImplementedError):
        SeaIceDrift('a.tif', 'b.tif')
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception):                # no device: the detector and the matcher raise, nothing falls back
            sid.get_drift_FT()
    n = ArrayNansat.rotated(img, angle_deg=5.0, scale=0.01, origin=(3.0, 4.0))
    lon, lat = n.transform_points([10.0, 20.0], [5.0, 7.0])
    c, r = n.transform_points(lon, lat, 1)
    np.testing.assert_allclose(c, [10.0, 20.0], atol=1e-9)
    np.testing.assert_allclose(r, [5.0, 7.0], atol=1e-9)
    assert abs(my.get_initial_rotation(ArrayNansat(img), n) + 5.0) < 1e-9


def test_out_of_image_keypoint_raises_like_the_reference_edt():
    """pmlib.py:72-73: seed[np.uint16(y), np.uint16(x)] = True fails for a key point outside image 2."""
    with pytest.raises(IndexError):
        my.nearest_keypoint_distance(np.array([10.0, 400.0]), np.array([10.0, 20.0]), [5], [5], shape=(300, 260))
    with pytest.raises(IndexError):
        my.nearest_keypoint_distance(np.array([10.0]), np.array([-3.0]), [5], [5], shape=(300, 260))
    d = my.nearest_keypoint_distance(np.array([10.9]), np.array([20.2]), [20], [13], shape=(300, 260))
    assert d[0] == 3.0                                              # truncated to pixel (20, 10)


def test_shared_handle_is_held_by_one_call_at_a_time(monkeypatch):
    """The per-device handle carries one call's state (pair, points, results): concurrent pm_dispatch calls must hold
    it from upload to fetch.  A fake handle records the order of the calls it sees."""
    import threading
    import time
    log = []

    class FakeCtx(object):
        def __init__(self, device=0):
            self.tag = None

        def upload_pair(self, img1, img2, slot=0, select=True):
            self.tag = int(img1[0, 0]); log.append(('upload', self.tag)); time.sleep(0.002)

        def set_points(self, c1, *a, **k):
            log.append(('set', self.tag)); self.n = len(c1); time.sleep(0.002)

        def run(self):
            log.append(('run', self.tag)); time.sleep(0.002)

        def fetch(self, want_ij=True):
            log.append(('fetch', self.tag))
            return np.full((self.n, 5), float(self.tag))

        def close(self):
            pass

    my.release_contexts()
    monkeypatch.setattr(my._capi, 'PMContext', FakeCtx)
    outs = {}

    def worker(tag):
        img = np.full((8, 8), tag, dtype=np.uint8)
        for _ in range(20):
            out = my.pm_dispatch(img, img, [1.0], [1.0], [1.0], [1.0], [20.0], 34, 0.0)
            outs.setdefault(tag, []).append(float(out[0, 0]))

    ts = [threading.Thread(target=worker, args=(t,)) for t in (3, 7)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    my.release_contexts()
    assert all(v == 3.0 for v in outs[3]) and all(v == 7.0 for v in outs[7])
    assert len(log) == 160
    for k in range(0, len(log), 4):                       # every call's four steps are contiguous and of one caller
        steps = log[k:k + 4]
        assert [s[0] for s in steps] == ['upload', 'set', 'run', 'fetch'] and len({s[1] for s in steps}) == 1


def test_flat_simplices_send_the_first_guess_to_scipy():
    """SciPy treats a (nearly) flat simplex as degenerate and widens its tolerance around it; the device's
    doubt flags do not model that, so a triangulation that holds one is evaluated by SciPy alone (lib.interpolation_near)."""
    import types
    from sea_ice_drift_amd import lib
    pts = np.array([[0.0, 0.0], [10.0, 0.0], [0.0, 10.0], [10.0, 10.0], [5.0, 5e-12]])
    good = types.SimpleNamespace(points=pts, simplices=np.array([[0, 1, 2], [1, 3, 2]]))
    flat = types.SimpleNamespace(points=pts, simplices=np.array([[0, 1, 2], [0, 1, 4]]))      # (0,0), (10,0), (5, 5e-12)
    dup = types.SimpleNamespace(points=pts, simplices=np.array([[0, 0, 2]]))                  # zero area: rcond = 0
    assert not lib._has_degenerate_simplex(good)
    assert lib._has_degenerate_simplex(flat) and lib._has_degenerate_simplex(dup)
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(5)
    assert not lib._has_degenerate_simplex(Delaunay(rng.uniform(0, 3000, (4000, 2))))
    assert not lib._has_degenerate_simplex(Delaunay(np.stack(np.meshgrid(np.arange(40.0), np.arange(40.0)), -1).reshape(-1, 2)))


def test_launch_classes_and_costs_the_host_predicts():
    """Host arithmetic of the C ABI (no device): the launch class of a border (sid_pm_estimate_residency: workgroups per CU in the
    low four bits, + 16 sums in global memory, + 32 the launch of the borders beyond the LDS) and its estimated cost
    (sid_pm_estimate_cost) follow the residency classes DESIGN.md section 6.3 describes - the three-wavefront class (four per CU,
    sums in global memory) up to border 23, three per CU with the sums in LDS, then in global memory, two per CU, one per CU,
    tables in global memory from border 69 - for the three kernel families."""
    from sea_ice_drift_amd import _capi
    borders = np.arange(20, 112, dtype=np.float64)

    def runs(s, k):
        out, prev = [], None
        for b, c in zip(borders, _capi.estimate_residency(borders, s, k)):
            if c != prev:
                out.append((int(b), int(c)))
                prev = c
        return out
    assert runs(34, 15) == [(20, 4 + 16), (24, 3), (28, 3 + 16), (37, 2 + 16), (48, 1 + 16), (69, 1 + 16 + 32)]
    assert runs(35, 15) == [(20, 4 + 16), (23, 3), (26, 3 + 16), (38, 2 + 16), (48, 1 + 16), (70, 1 + 16 + 32)]
    for k in (3, 7):                                                                # slot groups: sums in global memory at every border
        assert runs(34, k) == [(20, 4 + 16), (24, 3 + 16), (37, 2 + 16), (48, 1 + 16), (69, 1 + 16 + 32)]
    for s in (34, 35):
        cost = _capi.estimate_cost(borders, s, 15)
        assert (np.diff(cost) > -1e-9).all(), cost                                  # a larger border never costs less
        assert 60.0 < cost[0] < 80.0 and 300.0 < cost[30] < 420.0                   # ns per point at borders 20 and 50 (profiles/r04_border_cost.json)
    # beyond one workgroup's LDS - borders above 111, template sides above 64 - the points run the large-window pipeline: a class
    # of their own (SID_PM_CLASS_LARGE), priced at some ten microseconds each (in batches of 64: the launches of a batch are priced by sid_pm_estimate_run_time)
    big = _capi.estimate_residency([111.0, 112.0, 250.0], 34, 15)
    assert list(big) == [1 + 16 + 32, 1 + _capi.CLASS_LARGE, 1 + _capi.CLASS_LARGE]
    assert (_capi.estimate_residency(borders, 65, 15) == 1 + _capi.CLASS_LARGE).all()
    cost = _capi.estimate_cost([112.0, 250.0], 34, 15)
    assert 5e3 < cost[0] < cost[1] < 1e5                                          # (11 / 46 us measured in batches of 64)
    with pytest.raises(_capi.SidPmError):
        _capi.estimate_cost(borders, 256, 15)                                       # template sides beyond 255: unsupported, not estimated
    # the run-time estimate the shard cuts are made with: more than the sum of the costs (launch tails), the same rule as the launcher
    t = _capi.estimate_run_time(np.repeat(borders, 40), 34, 15)
    c = _capi.estimate_cost(np.repeat(borders, 40), 34, 15).sum()
    assert c < t < 2.0 * c
