"""the first guess evaluated on the GPU (include/sid_fg.h) against SciPy on the host - the reference's
lib.interpolation_near (lib.py:179-201) and pmlib.get_distance_to_nearest_keypoint (pmlib.py:61-77)."""
import numpy as np
import pytest
from scipy.interpolate import LinearNDInterpolator
from scipy.spatial import Delaunay, cKDTree

from sea_ice_drift_amd import _capi, lib, pmlib as my
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu


def test_interpolation_in_scipys_triangulation():
    rng = np.random.default_rng(8)
    src = rng.uniform(0, 5000, (6000, 2))
    vals = np.stack([src[:, 1] * 1.01 + 3 + rng.normal(0, 2, 6000), src[:, 0] * 0.99 - 2 + rng.normal(0, 2, 6000)], axis=1)
    q = np.concatenate([rng.uniform(-200, 5200, (20000, 2)), src[:500], (src[:300] + src[300:600]) / 2])   # incl. vertices, outside points
    tri = Delaunay(src)
    exp = LinearNDInterpolator(tri, vals)(q)
    got = _capi.fg_interp_linear(tri.points, tri.simplices, vals, q)
    nan = np.isnan(exp[:, 0])
    # hull membership may differ for points within rounding of the hull's boundary only
    diff = np.isnan(got[:, 0]) != nan
    assert diff.sum() <= 2
    ok = ~nan & ~np.isnan(got[:, 0])
    np.testing.assert_allclose(got[ok], exp[ok], rtol=1e-11, atol=1e-9)
    assert (np.round(got[ok]) == np.round(exp[ok])).mean() > 0.9999      # what the reference does next (pmlib.py:288)


def test_nearest_keypoint_distance_is_exact():
    rng = np.random.default_rng(9)
    seeds = np.floor(rng.uniform(0, 3000, (5000, 2)))
    q = np.floor(rng.uniform(0, 3000, (30000, 2)))
    exp, _ = cKDTree(seeds).query(q, k=1)
    np.testing.assert_array_equal(_capi.fg_nearest_dist(seeds, q), exp)


def test_prelude_on_the_device_equals_the_reference_fixture():
    """pm_prelude with the first guess on the GPU reproduces fixture G4's kernel inputs (the reference's own prelude)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g4_pattern_matching.npz'))
    n1, n2, c1, r1, c2, r2, lon_g, lat_g = mg.g4_inputs()
    pre = my.pm_prelude(lon_g, lat_g, n1, c1, r1, n2, c2, r2, img_size=34, angles=list(range(-3, 4)), first_guess_on='device')
    np.testing.assert_array_equal(pre['c2fg'], g['fg_c2'])
    np.testing.assert_array_equal(pre['r2fg'], g['fg_r2'])
    np.testing.assert_array_equal(pre['brd2'], g['fg_border'])
    host = my.pm_prelude(lon_g, lat_g, n1, c1, r1, n2, c2, r2, img_size=34, angles=list(range(-3, 4)), first_guess_on='host')
    for k in ('c2fg', 'r2fg', 'brd2', 'gpi'):
        np.testing.assert_array_equal(pre[k], host[k])
