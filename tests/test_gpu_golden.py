"""GPU parity against the committed golden fixtures (outputs of the reference's own Python)
and, at the benchmark's full size, against size-independent properties."""
import os

import numpy as np
import pytest

from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat
from sea_ice_drift_amd.seaicedrift import SeaIceDrift
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def check(got, exp, got_ij=None):
    """c2, r2, a, r exact (r is the float32 of the NCC spec); h to 1e-5; NaN rows identical."""
    nan = np.isnan(exp[:, 0])
    np.testing.assert_array_equal(np.isnan(got[:, 0]), nan)
    np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4])
    np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)
    assert np.isnan(got[nan]).all()
    if got_ij is not None:
        assert (got_ij[nan] == -1).all() and (got_ij[~nan] >= 0).all()


@pytest.mark.parametrize('s,alpha0', [(34, 0.0), (35, -3.85)])
def test_g3_reference_use_mcc_outputs(pm_ctx, s, alpha0):
    g = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    assert syn.sha256(img1, img2) == str(g['pair_sha'])
    v = [g[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    pm_ctx.upload_pair(img1, img2)
    for k, angles in enumerate(mg.G3_ANGLE_SETS):
        for mcc in ((0, 1) if k == 1 else (0,)):
            exp = g['out_s%d_k%d_m%d' % (s, k, mcc)]
            pm_ctx.set_points(*v, s, alpha0, angles, rot=my.rotation_table(angles, alpha0, s),
                              flags=1 | (4 if mcc else 0))
            pm_ctx.run()
            got, ij = pm_ctx.fetch()
            if mcc:                                  # r is then (r - median)/std: float32 round-off
                nan = np.isnan(exp[:, 0])
                np.testing.assert_array_equal(got[~nan, :3], exp[~nan, :3])
                np.testing.assert_allclose(got[~nan, 3:], exp[~nan, 3:], rtol=1e-5, atol=1e-5)
            else:
                check(got, exp, ij)


def test_g3_full_ncc_matrix_and_template(pm_ctx):
    g = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    pm_ctx.upload_pair(img1, img2)
    angles, s, alpha0 = mg.G3_ANGLE_SETS[1], 34, 0.0
    for i in g['full_points']:
        d = pm_ctx.debug_point(g['c1'][i], g['r1'][i], g['c2fg'][i], g['r2fg'][i], g['border'][i], s, alpha0,
                               angles, rot=my.rotation_table(angles, alpha0, s))
        np.testing.assert_array_equal(d['ccm'], g['full%d_result' % i])
        np.testing.assert_array_equal(d['templates'][d['ij'][2]], g['full%d_template' % i])


def test_g4_pattern_matching_end_to_end():
    """The public call, same arguments as the reference's, against the reference's output grids."""
    g = np.load(os.path.join(GOLD, 'g4_pattern_matching.npz'))
    n1, n2, c1, r1, c2, r2, lon_g, lat_g = mg.g4_inputs()
    out = my.pattern_matching(lon_g, lat_g, n1, c1, r1, n2, c2, r2, img_size=34, angles=list(range(-3, 4)), threads=5)
    for name, arr in zip(('u', 'v', 'a', 'r', 'lon2', 'lat2'), (out[0], out[1], out[2], out[3], out[5], out[6])):
        np.testing.assert_array_equal(arr, g[name], err_msg=name)
    np.testing.assert_allclose(out[4], g['h'], rtol=1e-5, atol=1e-5, equal_nan=True)
    # and through the class, from lon/lat keypoints (seaicedrift.py:62-88)
    lon1, lat1 = n1.transform_points(c1, r1)
    lon2, lat2 = n2.transform_points(c2, r2)
    out2 = SeaIceDrift(n1, n2).get_drift_PM(lon_g, lat_g, lon1, lat1, lon2, lat2, img_size=34,
                                            angles=list(range(-3, 4)))
    np.testing.assert_allclose(out2[3], g['r'], rtol=0, atol=1e-6, equal_nan=True)


def test_use_mcc_single_point_signature():
    img1, img2 = mg.g3_pair()
    g = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    exp = g['out_s34_k1_m0'][3]
    got = my.use_mcc(g['c1'][3], g['r1'][3], g['c2fg'][3], g['r2fg'][3], g['border'][3], img1, img2, 34, 0.0,
                     angles=mg.G3_ANGLE_SETS[1])
    assert got[:4] == tuple(exp[:4]) and abs(got[4] - exp[4]) < 1e-5


def test_unsupported_sizes_and_flags_are_errors(pm_ctx):
    img1, img2 = syn.make_pair(300, 300, seed=2)
    pm_ctx.upload_pair(img1, img2)
    one = ([150.0], [150.0], [150.0], [150.0], [20.0])
    with pytest.raises(_capi.SidPmError) as e:                     # (sides up to 255 run: tests/test_gpu_large_window.py)
        pm_ctx.set_points(*one, 256, 0.0, [0.0])
    assert e.value.code == -4
    with pytest.raises(_capi.SidPmError):
        pm_ctx.set_points(*one, 34, 0.0, [])
    with pytest.raises(_capi.SidPmError) as e:                     # unknown flag bit (bits 3..5 = SID_PM_ROT_ORDER(n))
        pm_ctx.set_points(*one, 34, 0.0, [0.0], flags=64)
    assert e.value.code == -1
    with pytest.raises(_capi.SidPmError) as e:                     # rot_order 6: scipy has orders 0..5
        pm_ctx.set_points(*one, 34, 0.0, [0.0], flags=1 | (6 << 3))
    assert e.value.code == -4
    # a border too large for the LDS of one workgroup is no longer an error: the large-window pipeline takes the point
    pm_ctx.set_points([150.0], [150.0], [150.0], [150.0], [112.0], 34, 0.0, [0.0])
    pm_ctx.run()
    out, ij = pm_ctx.fetch()
    assert np.isfinite(out).all() and (ij >= 0).all()


def test_edge_cases_empty_and_outside(pm_ctx):
    img1, img2 = syn.make_pair(300, 300, seed=2)
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points([], [], [], [], [], 34, 0.0, [0.0])
    pm_ctx.run()
    out, ij = pm_ctx.fetch()
    assert out.shape == (0, 5) and ij.shape == (0, 3)
    # window sticking out of image 2 -> NaN; template sticking out of image 1 -> zero pixel -> NaN
    pm_ctx.set_points([150.0, 10.0, 150.0], [150.0, 150.0, 150.0], [20.0, 150.0, 150.0], [150.0, 150.0, 150.0],
                      [20.0, 20.0, 20.0], 34, 0.0, [-3.0, 0.0, 3.0])
    pm_ctx.run()
    out, ij = pm_ctx.fetch()
    assert np.isnan(out[0]).all() and np.isnan(out[1]).all() and np.isfinite(out[2]).all()
    assert (ij[:2] == -1).all()


@pytest.fixture(scope='module')
def full_pair():
    img1, img2 = syn.make_pair(10000, 10000)
    return img1, img2, syn.make_grid(10000, 10000, 200)


def test_g5_fullsize_subsample(pm_ctx, full_pair):
    """BASELINE configs[1] inputs: hash of the generated pair, then the 1 % subsample of the
    200x200 grid against the committed oracle results."""
    g = np.load(os.path.join(GOLD, 'g5_fullsize.npz'))
    img1, img2, grid = full_pair
    assert syn.sha256(img1, img2) == str(g['pair_sha'])
    assert syn.sha256(*[grid[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]) == str(g['grid_sha'])
    sel = g['sel']
    angles = list(range(-7, 8))
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(*[grid[k][sel] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')], 34, 0.0, angles,
                      rot=my.rotation_table(angles, 0.0, 34))
    pm_ctx.run()
    got, ij = pm_ctx.fetch()
    np.testing.assert_array_equal(ij, g['ij'])
    check(got, g['out'], ij)


def test_fullsize_properties(pm_ctx, full_pair):
    """All 40 000 points of the benchmark workload: properties that need no oracle."""
    img1, img2, grid = full_pair
    angles = list(range(-7, 8))
    rot = my.rotation_table(angles, 0.0, 34)
    v = [grid[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(*v, 34, 0.0, angles, rot=rot)
    pm_ctx.run()
    out, ij = pm_ctx.fetch()
    assert np.isfinite(out).all()
    # (1) idempotence / determinism: a second run is bit-identical
    pm_ctx.run()
    out2, ij2 = pm_ctx.fetch()
    np.testing.assert_array_equal(out, out2)
    np.testing.assert_array_equal(ij, ij2)
    # (2) peak indices lie inside the NCC matrix and reproduce c2/r2 exactly (pmlib.py:168-169,209-210)
    b = grid['border']
    assert (ij[:, 0] >= 0).all() and (ij[:, 0] < 2 * b + 2).all() and (ij[:, 1] < 2 * b + 2).all()
    np.testing.assert_array_equal(out[:, 0], grid['c2fg'] + ij[:, 1] - (b + 0.5))
    np.testing.assert_array_equal(out[:, 1], grid['r2fg'] + ij[:, 0] - (b + 0.5))
    np.testing.assert_array_equal(out[:, 2], np.array(angles, dtype=np.float64)[ij[:, 2]])
    assert (out[:, 3] <= 1.0).all() and (out[:, 3] > 0.2).all()
    # (3) the known synthetic drift is recovered (minus the reference's -1.5 px template-centre bias)
    dc, dr = syn.true_displacement(grid['c1'], grid['r1'])
    assert np.median(np.abs(out[:, 0] - grid['c1'] - dc + 1.5)) < 0.6
    assert np.median(np.abs(out[:, 1] - grid['r1'] - dr + 1.5)) < 0.6
    # (4) translation invariance: moving the first guess by (+2,-3) and widening the border by 3 keeps the
    #     matched position and r wherever the peak was not on the edge of the original search area
    sel = np.arange(0, 40000, 16)
    inner = ((ij[sel, 0] > 0) & (ij[sel, 1] > 0) & (ij[sel, 0] < 2 * b[sel] + 1) & (ij[sel, 1] < 2 * b[sel] + 1)
             & (b[sel] + 3 <= 50))
    sel = sel[inner]
    pm_ctx.set_points(grid['c1'][sel], grid['r1'][sel], grid['c2fg'][sel] + 2, grid['r2fg'][sel] - 3, b[sel] + 3,
                      34, 0.0, angles, rot=rot)
    pm_ctx.run()
    out3, _ = pm_ctx.fetch()
    same = (out3[:, 3] == out[sel, 3])
    assert same.mean() > 0.97                      # a few points find a better peak in the wider window
    np.testing.assert_array_equal(out3[same, :3], out[sel][same, :3])
    # (5) sharded == unsharded (the multi-GPU partition bench.py ships - contiguous runs of equal estimated time in border
    #     order - run on one device: the first, a middle and the last of eight ranks)
    from sea_ice_drift_amd.dist import shard_indices_by_cost
    for rank in (0, 3, 7):
        idx = shard_indices_by_cost(b, 8, rank, 34, len(angles))[::5]
        pm_ctx.set_points(*[x[idx] for x in v], 34, 0.0, angles, rot=rot)
        pm_ctx.run()
        o, j = pm_ctx.fetch()
        np.testing.assert_array_equal(o, out[idx])
        np.testing.assert_array_equal(j, ij[idx])
