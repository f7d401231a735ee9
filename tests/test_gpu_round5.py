"""GPU parity of what round 5 added to the path: bilinear template sampling (``rot_order=1``, pmlib.py:89,112-113) and the
winner's NCC matrix from the accumulators the sweep kept (``PMArgs::gs_keep_acc``)."""
import os

import numpy as np
import pytest

from oracle import pm_oracle as po
from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rot_for(angles, alpha0, s):
    return np.array([po.rotation_terms(a - alpha0, s) for a in angles])


def assert_parity(got, got_ij, exp, exp_ij):
    np.testing.assert_array_equal(got_ij, exp_ij)
    nan = np.isnan(exp[:, 0])
    np.testing.assert_array_equal(np.isnan(got[:, 0]), nan)
    np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4])
    np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('s,alpha0', [(34, 0.0), (35, -3.85)])
def test_rot_order1_equals_the_reference_fixture(pm_ctx, s, alpha0):
    """Fixture G3b: the reference's own use_mcc(..., rot_order=1) on G3's pair and points (fractional and integral centres,
    points on the zero patch).  c2, r2, a, r bit for bit, h to 1e-5; the rotated templates of a few points bit for bit
    against the restated scipy arithmetic (which G1b pins to the reference)."""
    g = np.load(os.path.join(GOLD, 'g3b_use_mcc_order1.npz'))
    g3 = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    assert syn.sha256(img1, img2) == str(g['pair_sha'])
    v = [g3[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    pm_ctx.upload_pair(img1, img2)
    for k, angles in enumerate(mg.G3_ANGLE_SETS):
        exp = g['out_s%d_k%d' % (s, k)]
        pm_ctx.set_points(*v, s, alpha0, angles, rot=my.rotation_table(angles, alpha0, s), flags=_capi.HES_NORM | _capi.ROT_ORDER1)
        pm_ctx.run()
        got, ij = pm_ctx.fetch()
        nan = np.isnan(exp[:, 0])
        np.testing.assert_array_equal(np.isnan(got[:, 0]), nan)
        np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4])
        np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)
        assert (ij[nan] == -1).all() and nan.any()
    angles = mg.G3_ANGLE_SETS[1]
    for i in (0, 1, 2, 7):                                           # integral (0) and fractional centres
        d = pm_ctx.debug_point(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], s, alpha0, angles,
                               rot=my.rotation_table(angles, alpha0, s), flags=_capi.HES_NORM | _capi.ROT_ORDER1)
        for ka, a in enumerate(angles):
            np.testing.assert_array_equal(d['templates'][ka], po.get_template_order1(img1, v[0][i], v[1][i], a - alpha0, s))


@pytest.mark.parametrize('s,angles', [(21, [-3, 0, 3]), (40, list(range(-3, 4))), (34, [0.5 * k for k in range(-8, 9)])])
def test_rot_order1_other_kernels_against_the_oracle(pm_ctx, c_oracle, s, angles):
    """The classic kernel (other template sides), several groups of angles (the winner is then sampled a second time), templates
    cut by the image border (cval = 0 -> NaN point): all through the C oracle's restatement."""
    img1, img2 = syn.make_pair(700, 700, seed=41)
    rng = np.random.default_rng(42)
    n = 60
    c1 = rng.uniform(100, 600, n); r1 = rng.uniform(100, 600, n)
    c1[::2] = np.rint(c1[::2]); r1[::2] = np.rint(r1[::2])
    c1[:3] = [3.0, 697.5, 350.0]; r1[:3] = [350.0, 350.0, 2.25]      # templates that reach beyond image 1 (at every side tested)
    dc, dr = syn.true_displacement(c1, r1)
    c2 = np.rint(c1 + dc); r2 = np.rint(r1 + dr)
    c2[:3] = 350.0; r2[:3] = 350.0
    border = rng.integers(20, 40, n).astype(np.float64)
    rot = rot_for(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, nthreads=8, flags=1 | 8)
    assert np.isnan(exp[:3, 0]).all() and np.isfinite(exp[3:, 0]).all()
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, flags=1 | 8)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    assert_parity(got, got_ij, exp, exp_ij)
    # and it is not order 0
    exp0, _ = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, nthreads=8, flags=1)
    assert (exp0[3:, 3] != exp[3:, 3]).mean() > 0.5


def test_rot_order1_through_the_python_api(c_oracle):
    img1, img2 = syn.make_pair(500, 500, seed=3)
    out = my.pm_dispatch(img1, img2, [250.0, 200.5], [250.0, 260.25], [252.0, 203.0], [249.0, 262.0], [22.0, 30.0], 35, 1.5,
                         angles=[-3, 0, 3], rot_order=1, mtype=my.TM_CCOEFF_NORMED)
    exp, _ = c_oracle.pm_batch(img1, img2, [250.0, 200.5], [250.0, 260.25], [252.0, 203.0], [249.0, 262.0], [22.0, 30.0], 35, 1.5,
                               [-3, 0, 3], rot=my.rotation_table([-3, 0, 3], 1.5, 35), flags=1 | 8)
    np.testing.assert_array_equal(out[:, :4], exp[:, :4])
    np.testing.assert_allclose(out[:, 4], exp[:, 4], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('s,angles,flags', [(35, [-3, 0, 3], 1), (34, [-3, 0, 3], 1), (34, list(range(-3, 4)), 1), (35, [0.0], 1),
                                            (34, [-2, 0, 2], 7), (35, list(range(-3, 4)), 3)])
def test_kept_accumulators_equal_the_recomputed_winner_and_the_oracle(pm_ctx, c_oracle, monkeypatch, s, angles, flags):
    """Slot-group launches (at most 7 angles) keep the sweep's accumulators in the point's block of global memory and normalise
    the winning angle's matrix from them (rp_winner_kept); SID_PM_KEEP_ACC=0 recomputes it on the matrix cores as round 4 did.
    Every launch class (borders 20 .. 50 and a big one), fractional centres, the zero patch, both Hessian routes, mcc_norm:
    both routes against the oracle, and bit-identical to each other - h included (the same float32 matrix)."""
    size = 1100
    img1, img2 = syn.make_pair(size, size, seed=19)
    img1 = img1.copy()
    img1[500:520, 480:530] = 0
    rng = np.random.default_rng(20)
    n = 330
    c1 = rng.uniform(200, size - 200, n); r1 = rng.uniform(200, size - 200, n)
    c1[::4] = np.rint(c1[::4]); r1[::4] = np.rint(r1[::4]); c1[1::4] = np.rint(c1[1::4]); r1[1::4] = np.rint(r1[1::4])
    dc, dr = syn.true_displacement(c1, r1)
    c2 = np.rint(c1 + dc) + rng.integers(-2, 3, n); r2 = np.rint(r1 + dr) + rng.integers(-2, 3, n)
    border = np.array(list(range(20, 51)) * (n // 31 + 1), dtype=np.float64)[:n]
    border[-2:] = [75.0, 90.0]
    c1[-2:] = r1[-2:] = c2[-2:] = r2[-2:] = 550.0
    rot = rot_for(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    assert np.isnan(exp[:, 0]).any() and np.isfinite(exp[:, 0]).sum() > 0.9 * n
    pm_ctx.upload_pair(img1, img2)
    res = []
    for keep in (None, '0'):
        monkeypatch.delenv('SID_PM_KEEP_ACC', raising=False)
        if keep is not None:
            monkeypatch.setenv('SID_PM_KEEP_ACC', keep)
        pm_ctx.set_points(c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, flags=flags)
        pm_ctx.run()
        got, got_ij = pm_ctx.fetch()
        np.testing.assert_array_equal(got_ij, exp_ij)
        nan = np.isnan(exp[:, 0])
        np.testing.assert_array_equal(got[~nan, :3], exp[~nan, :3])
        if flags & 4:
            np.testing.assert_allclose(got[~nan, 3], exp[~nan, 3], rtol=1e-5, atol=1e-5)
        else:
            np.testing.assert_array_equal(got[~nan, 3], exp[~nan, 3])
        np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)
        res.append((got.copy(), got_ij.copy()))
    monkeypatch.delenv('SID_PM_KEEP_ACC', raising=False)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][0], res[1][0])             # bit for bit, NaN rows included
    # the NCC matrix itself (debug_point runs the kept route for these angle sets): against the oracle's matcher
    d = pm_ctx.debug_point(c1[0], r1[0], c2[0], r2[0], 24.0, s, 0.0, angles, rot=rot, flags=1)
    full = po.use_mcc(c1[0], r1[0], c2[0], r2[0], 24.0, img1, img2, s, 0.0, full=True, angles=angles)
    np.testing.assert_array_equal(d['ccm'], full[1][2])


@pytest.mark.parametrize('s,angles', [(34, list(range(-7, 8))), (35, [-3, 0, 3]), (34, list(range(-3, 4))), (35, [0.5 * k for k in range(-8, 9)]),
                                      (34, [-45.0, -20.0, 0.0, 33.3, 90.0])])
def test_sorted_sampling_table_equals_the_plain_one_and_the_oracle(pm_ctx, c_oracle, monkeypatch, s, angles):
    """Round 5 (measured slower, built on request only: SID_PM_SAMP2=1): the sampling table sorted per angle - quads of four
    CONSECUTIVE patch bytes as run records (two aligned dwords + v_alignbyte), the rest as gathers (PMArgs::samp2).  Small and large angles (at 45 degrees almost
    nothing is a run), several groups of angles, both template sides, a scene rotation that makes every angle fractional, points
    on the zero patch: against the oracle, and bit-identical to the plain table."""
    img1, img2 = syn.make_pair(900, 900, seed=31)
    img1 = img1.copy()
    img1[400:420, 380:430] = 0
    g = syn.make_grid(900, 900, 13, margin=120)
    rng = np.random.default_rng(6)
    border = rng.integers(20, 38, len(g['c1'])).astype(np.float64)
    res = []
    for alpha0 in (0.0, -3.85):
        rot = rot_for(angles, alpha0, s)
        exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], border, s, alpha0, angles, rot=rot, nthreads=8)
        assert np.isnan(exp[:, 0]).any()
        pm_ctx.upload_pair(img1, img2)
        for env in (None, 'SID_PM_SAMP2'):
            monkeypatch.delenv('SID_PM_SAMP2', raising=False)
            if env:
                monkeypatch.setenv(env, '1')
            pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], border, s, alpha0, angles, rot=rot)
            pm_ctx.run()
            got, got_ij = pm_ctx.fetch()
            assert_parity(got, got_ij, exp, exp_ij)
            res.append((got.copy(), got_ij.copy()))
        monkeypatch.delenv('SID_PM_SAMP2', raising=False)
        np.testing.assert_array_equal(res[-1][0], res[-2][0])
        np.testing.assert_array_equal(res[-1][1], res[-2][1])
