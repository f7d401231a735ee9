"""GPU parity of ``rot_order`` 2..5 (pmlib.py:89,112-113: scipy's affine_transform with spline order n - the WHOLE image 1 through
scipy's recursive B-spline prefilter, then n + 1 weights per axis per sample): fixture G1c = the reference's get_template at
orders 2..5, fixture G3d = the reference's use_mcc at orders 2 / 3 / 4 / 5, and the C oracle at other kernels and shapes."""
import os

import numpy as np
import pytest

from oracle import pm_oracle as po
from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('order', [2, 3, 4, 5])
def test_g1c_get_template_of_the_reference(order):
    """get_template(img, c, r, a, s, rot_order=order) with the reference's signature, bit for bit: a smooth image, a noise image
    (the splines overshoot to 0 and 255 there), centres at the image border (taps mirrored, cval = 0 outside), 34 / 35 / 64 px."""
    g = np.load(os.path.join(GOLD, 'g1c_templates_spline.npz'))
    for name, img in (('smooth', mg.g1c_image()), ('noise', mg.g1_image())):
        for k, (c, r, a, s) in enumerate(mg.G1C_CASES):
            got = my.get_template(img, c, r, a, int(s), rot_order=order)
            np.testing.assert_array_equal(got, g['%s_o%d_%d' % (name, order, k)], err_msg='%s order %d case %r' % (name, order, (c, r, a, s)))


def test_spline_coefficients_through_the_kernel_templates(pm_ctx):
    """The kernel's own templates (debug_point: prefilter on the device, pre-sampling, load in the one-point kernel) against the
    NumPy oracle's, which tests/test_oracle_golden.py pins to scipy and to the reference."""
    img = mg.g1c_image()
    pm_ctx.upload_pair(img, img)
    angles = [-9, -3, 0, 1.234, 45, 90]
    for order in (2, 3, 5):
        coeffs = po.spline_coefficients(img, order)
        for s in (34, 35):
            for (c, r) in ((200.0, 150.0), (200.3, 150.7)):
                d = pm_ctx.debug_point(c, r, 200.0, 200.0, 20.0, s, 0.0, angles, flags=_capi.HES_NORM | _capi.rot_order_flag(order))
                for ka, a in enumerate(angles):
                    np.testing.assert_array_equal(d['templates'][ka], po.get_template_spline(img, c, r, a, s, order, coeffs=coeffs),
                                                  err_msg='order %d s %d centre %r angle %r' % (order, s, (c, r), a))


@pytest.mark.parametrize('case', mg.G3D_CASES, ids=['s%d_a%g_o%d_k%d' % c for c in mg.G3D_CASES])
def test_g3d_use_mcc_of_the_reference(pm_ctx, case):
    """The reference's use_mcc(..., rot_order = 2 .. 5) on G3's pair and points: c2, r2, a, r bit for bit, h to 1e-5, NaN rows
    identical (about half of the points: a spline template that dips to 0 is the reference's NaN rule)."""
    s, alpha0, order, k = case
    g = np.load(os.path.join(GOLD, 'g3d_use_mcc_spline.npz'))
    g3 = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    assert syn.sha256(img1, img2) == str(g['pair_sha'])
    v = [g3[x] for x in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    angles = mg.G3_ANGLE_SETS[k]
    exp = g['out_s%d_o%d_k%d' % (s, order, k)]
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(*v, s, alpha0, angles, flags=_capi.HES_NORM | _capi.rot_order_flag(order))
    for _ in range(2):                                              # (the second run reuses coefficients and pre-sampled templates)
        pm_ctx.run()
        got, ij = pm_ctx.fetch()
        nan = np.isnan(exp[:, 0])
        np.testing.assert_array_equal(np.isnan(got[:, 0]), nan)
        assert nan.any() and (~nan).sum() > 20 and (ij[nan] == -1).all()
        np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4])
        np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)
    # and through the public single-point call
    i = int(np.flatnonzero(~nan)[3])
    one = my.use_mcc(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], img1, img2, s, alpha0, angles=angles, rot_order=order)
    assert one[:4] == tuple(exp[i, :4])


@pytest.mark.parametrize('s,angles,order', [(21, [-3, 0, 3], 3), (40, list(range(-3, 4)), 2), (34, [0.5 * k for k in range(-8, 9)], 3),
                                            (100, [-3, 0, 3], 3), (34, list(range(-7, 8)), 5)])
def test_spline_orders_in_the_other_kernels_against_the_oracle(pm_ctx, c_oracle, s, angles, order):
    """The classic kernel (other template sides), several groups of angles (the winner is then loaded a second time), the
    large-window pipeline (100 px; a border of 120 px), templates cut by the image border: all against the C oracle."""
    img1, img2 = syn.make_pair(800, 800, seed=41)
    img1 = (60 + img1 * 0.6).astype(np.uint8)                        # (brighter: a spline template that dips to 0 is a NaN point)
    rng = np.random.default_rng(42)
    n = 40
    c1 = rng.uniform(150, 650, n); r1 = rng.uniform(150, 650, n)
    c1[::2] = np.rint(c1[::2]); r1[::2] = np.rint(r1[::2])
    c1[:3] = [3.0, 797.5, 400.0]; r1[:3] = [400.0, 400.0, 2.25]      # templates that reach beyond image 1
    dc, dr = syn.true_displacement(c1, r1)
    c2 = np.rint(c1 + dc); r2 = np.rint(r1 + dr)
    c2[:3] = 400.0; r2[:3] = 400.0
    border = rng.integers(20, 40, n).astype(np.float64)
    if s == 34:
        border[5] = 120.0; c1[5] = r1[5] = c2[5] = r2[5] = 400.0     # one point beyond the LDS classes
    flags = 1 | po.flag_rot_order(order)
    rot = my.rotation_table(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    assert np.isnan(exp[:3, 0]).all() and np.isfinite(exp[3:, 0]).sum() > 30
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, flags=flags)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    np.testing.assert_array_equal(got_ij, exp_ij)
    nan = np.isnan(exp[:, 0])
    np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4])
    np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)
    # a new pair on the same handle: the coefficients are recomputed
    pm_ctx.upload_pair(img2, img1)
    pm_ctx.run()
    got2, _ = pm_ctx.fetch()
    exp2, _ = c_oracle.pm_batch(img2, img1, c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    np.testing.assert_array_equal(got2[:, :4], exp2[:, :4])


def test_spline_order_through_rotate_and_match():
    img1, img2 = syn.make_pair(400, 420, seed=9)
    img1 = (60 + img1 * 0.6).astype(np.uint8)                        # (brighter: a spline template that dips to 0 is a NaN point)
    out = my.rotate_and_match(img1, 200.5, 190.25, 50, img2[50:300, 60:400], 3.0, angles=[-2, 0, 2], rot_order=3)
    from oracle import c_oracle
    exp = c_oracle.rotate_and_match(img1, 200.5, 190.25, 50, np.ascontiguousarray(img2[50:300, 60:400]), 3.0, [-2, 0, 2],
                                    my.rotation_table([-2, 0, 2], 3.0, 50), flags=1 | po.flag_rot_order(3))
    assert exp['ij'][2] >= 0
    assert (out[0], out[1]) == tuple(exp['out'][:2]) and np.float64(out[3]) == exp['out'][3]
    np.testing.assert_array_equal(out[5], exp['ccm'])
    np.testing.assert_array_equal(out[6], exp['template'])
