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
    """Located simplices and values against SciPy's own find_simplex / LinearNDInterpolator on a triangulation with queries
    inside, outside, ON vertices and ON edge midpoints.  Every query the device does not flag must have SciPy's hull
    membership and round to the same integer (what the reference does next, pmlib.py:288); wherever the located simplex is
    SciPy's the values are compared bit for bit (the 2x2 LU restates LAPACK's) and the share is asserted; queries on
    vertices / edges are resolved on the device when every containing simplex rounds alike, the rest stays flagged and the
    product evaluates it with SciPy (lib.interpolation_near)."""
    rng = np.random.default_rng(8)
    src = rng.uniform(0, 5000, (6000, 2))
    vals = np.stack([src[:, 1] * 1.01 + 3 + rng.normal(0, 2, 6000), src[:, 0] * 0.99 - 2 + rng.normal(0, 2, 6000)], axis=1)
    q = np.concatenate([rng.uniform(-200, 5200, (20000, 2)), src[:500], (src[:300] + src[300:600]) / 2])   # incl. vertices, outside points
    tri = Delaunay(src)
    exp = LinearNDInterpolator(tri, vals)(q)
    exp_sx = tri.find_simplex(q)
    got, sx, doubt = _capi.fg_interp_linear(tri.points, tri.simplices, vals, q, details=True)
    clear = ~doubt
    assert clear[:20000].mean() > 0.999                                   # random queries are never in doubt
    assert clear[20000:20500].mean() > 0.9                                # queries ON vertices: resolved (every incident simplex rounds alike)
    # unflagged: SciPy's hull membership, the same integer after rounding
    np.testing.assert_array_equal(np.isnan(got[clear, 0]), np.isnan(exp[clear, 0]))
    inside = clear & (exp_sx >= 0)
    np.testing.assert_array_equal(np.round(got[inside]), np.round(exp[inside]))
    np.testing.assert_allclose(got[inside], exp[inside], rtol=1e-9, atol=1e-8)
    # bit equality wherever the simplex is SciPy's: counted, and nearly all of them
    same_sx = inside & (sx == exp_sx)
    assert same_sx[:20000].sum() == inside[:20000].sum()                  # strictly interior queries: always SciPy's simplex
    same_bits = (got[same_sx] == exp[same_sx]).all(axis=1)
    assert same_bits.mean() > 0.99, 'only %.4f of the values are bit-identical to SciPy' % same_bits.mean()
    # the product's interpolation_near (flagged queries through SciPy): the rounded first guess equals SciPy's everywhere
    xg, yg = lib.interpolation_near(src[:, 1], src[:, 0], vals[:, 0], vals[:, 1], q[:, 1], q[:, 0], first_guess_device=0)
    both = np.stack([xg, yg], axis=1)
    np.testing.assert_array_equal(np.isnan(both[:, 0]), np.isnan(exp[:, 0]))
    fin = ~np.isnan(exp[:, 0])
    np.testing.assert_array_equal(np.round(both[fin]), np.round(exp[fin]))


def test_lattice_keypoints_are_resolved_or_flagged():
    """Key points on the integer lattice (what a detector delivers at pyramid level 0) and integer grid queries: queries that
    coincide with key points or lie on axis-parallel edges are resolved on the device (all containing simplices round
    alike) or stay flagged; either way the rounded first guess is SciPy's."""
    rng = np.random.default_rng(18)
    src = np.unique(np.floor(rng.uniform(0, 300, (4000, 2))), axis=0)
    vals = np.stack([src[:, 1] + rng.normal(0, 3, len(src)), src[:, 0] + rng.normal(0, 3, len(src))], axis=1)
    qx, qy = np.meshgrid(np.arange(5.0, 295.0, 3.0), np.arange(5.0, 295.0, 3.0))
    q = np.stack([qy.ravel(), qx.ravel()], axis=1)
    tri = Delaunay(src)
    exp = LinearNDInterpolator(tri, vals)(q)
    got, sx, doubt = _capi.fg_interp_linear(tri.points, tri.simplices, vals, q, details=True)
    on_vertex = (q[:, None, :] == src[None, :, :]).all(axis=2).any(axis=1)
    assert on_vertex.sum() > 50                                           # coincidences do occur on a lattice
    assert doubt.mean() < 0.05                                            # ... and nearly all of them are resolved here
    clear = ~doubt
    np.testing.assert_array_equal(np.isnan(got[clear, 0]), np.isnan(exp[clear, 0]))
    fin = clear & ~np.isnan(exp[:, 0])
    np.testing.assert_array_equal(np.round(got[fin]), np.round(exp[fin]))
    xg, yg = lib.interpolation_near(src[:, 1], src[:, 0], vals[:, 0], vals[:, 1], q[:, 1], q[:, 0], first_guess_device=0)
    fin = ~np.isnan(exp[:, 0])
    np.testing.assert_array_equal(np.isnan(xg), np.isnan(exp[:, 0]))
    np.testing.assert_array_equal(np.round(np.stack([xg, yg], axis=1)[fin]), np.round(exp[fin]))


def test_bucket_walk_equals_the_brute_force_pass(monkeypatch):
    """Round 4: point location walks a uniform grid of buckets instead of testing every simplex.  Values, located simplices
    and doubt flags must be those of the brute-force pass (SID_FG_NO_GRID=1) on scattered and on lattice key points, with
    queries inside, outside, on vertices and far outside the bounding box; a triangulation whose bucket lists would not fit
    (one huge sliver fan) falls back to the brute-force pass by itself."""
    rng = np.random.default_rng(28)
    cases = []
    src = rng.uniform(0, 9000, (30000, 2))
    cases.append((src, np.concatenate([rng.uniform(-500, 9500, (40000, 2)), src[:2000], rng.uniform(-1e5, 1e5, (500, 2))])))
    lat = np.unique(np.floor(rng.uniform(0, 400, (6000, 2))), axis=0)
    qx, qy = np.meshgrid(np.arange(2.0, 398.0, 2.0), np.arange(2.0, 398.0, 2.0))
    cases.append((lat, np.stack([qy.ravel(), qx.ravel()], axis=1)))
    # a few points far away: long thin hull triangles across the whole box
    far = np.concatenate([rng.uniform(0, 100, (3000, 2)), [[1e6, 50.0], [-1e6, 50.0], [50.0, 1e6]]])
    cases.append((far, rng.uniform(-10, 110, (5000, 2))))
    for src, q in cases:
        vals = np.stack([src[:, 1] * 1.01 + rng.normal(0, 2, len(src)), src[:, 0] * 0.99 + rng.normal(0, 2, len(src))], axis=1)
        tri = Delaunay(src)
        monkeypatch.delenv('SID_FG_NO_GRID', raising=False)
        got, sx, doubt = _capi.fg_interp_linear(tri.points, tri.simplices, vals, q, details=True)
        monkeypatch.setenv('SID_FG_NO_GRID', '1')
        exp, esx, edoubt = _capi.fg_interp_linear(tri.points, tri.simplices, vals, q, details=True)
        monkeypatch.delenv('SID_FG_NO_GRID')
        np.testing.assert_array_equal(doubt, edoubt)
        np.testing.assert_array_equal(sx, esx)
        np.testing.assert_array_equal(got, exp)                           # (NaN positions and bits alike)


def test_nearest_keypoint_distance_is_exact(monkeypatch):
    """The distance to the nearest key point through the grid of buckets (round 5: rings of cells around the query, a bound
    that stops the search) against a KD-tree and against the brute-force kernel (SID_FG_NO_GRID=1), bit for bit: uniform
    seeds, seeds clustered in a corner (most cells empty: long ring walks), queries far outside the seeds' box, queries ON
    seeds, seeds on one line, a handful of seeds (brute force)."""
    rng = np.random.default_rng(9)
    cases = []
    cases.append((np.floor(rng.uniform(0, 3000, (5000, 2))), np.floor(rng.uniform(0, 3000, (30000, 2)))))
    cl = np.concatenate([np.floor(rng.normal(400, 60, (4000, 2))), np.floor(rng.uniform(0, 10000, (40, 2)))])
    cases.append((cl, np.floor(rng.uniform(-500, 10500, (40000, 2)))))
    s3 = np.floor(rng.uniform(2000, 2600, (20000, 2)))
    cases.append((s3, np.concatenate([s3[:5000], np.floor(rng.uniform(0, 10000, (20000, 2))), rng.uniform(1900, 2700, (5000, 2))])))
    cases.append((np.stack([np.arange(1000.0), np.full(1000, 77.0)], axis=1) + np.array([[0.0, 0.0]]), np.floor(rng.uniform(-50, 1050, (5000, 2)))))
    cases.append((np.floor(rng.uniform(0, 100, (7, 2))), np.floor(rng.uniform(0, 100, (300, 2)))))
    for seeds, q in cases:
        got = _capi.fg_nearest_dist(seeds, q)
        integral = q == np.floor(q)                                   # (pixel coordinates: sqrt of an exact integer, as the EDT's)
        sel = integral.all(axis=1)
        exp, _ = cKDTree(seeds).query(q[sel], k=1)
        np.testing.assert_array_equal(got[sel], exp)
        monkeypatch.setenv('SID_FG_NO_GRID', '1')
        brute = _capi.fg_nearest_dist(seeds, q)
        monkeypatch.delenv('SID_FG_NO_GRID')
        np.testing.assert_array_equal(got, brute)


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
