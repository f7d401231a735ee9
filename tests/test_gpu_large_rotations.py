"""GPU parity at LARGE effective rotations.  The template is sampled at ``angle - alpha0`` (pmlib.py:151) and the scene rotation
``alpha0`` (pmlib.py:79-87) is tens of degrees for ascending / descending pairs; rounds 1-5 never drove the HIP path beyond 11
degrees.  Here: the reference's own templates at 45 / 90 degrees (fixtures G1, G1b) through ``debug_point``, and the reference's
``use_mcc`` with alpha0 of 30 / 90 / -137.5 degrees (fixture G3c) through set_points / run / fetch - with the offset-table sampler
and with the on-the-fly sampler (SID_PM_NO_SAMP_TABLE)."""
import os

import numpy as np
import pytest

from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('no_table', [False, True])
@pytest.mark.parametrize('order', [0, 1])
def test_g1_templates_of_the_reference_through_the_kernel(pm_ctx, monkeypatch, order, no_table):
    """get_template (pmlib.py:89-115) at -9 .. 9, 1.234, -3.85, 45, 90 and -7.5 degrees, integral and fractional centres: the
    kernel's rotated templates bit for bit against what the reference's scipy call returned."""
    if no_table:
        monkeypatch.setenv('SID_PM_NO_SAMP_TABLE', '1')
    g = np.load(os.path.join(GOLD, 'g1b_templates_order1.npz' if order else 'g1_templates.npz'))
    img = mg.g1_image()
    assert syn.sha256(img) == str(g['img_sha'])
    pm_ctx.upload_pair(img, img)
    angles = mg.G1_ANGLES
    flags = _capi.HES_NORM | (_capi.ROT_ORDER1 if order else 0)
    for s, key in ((34, 't34'), (35, 't35')):
        exp = g[key].reshape(len(angles), len(mg.G1_CENTRES), s, s)
        for ci, (c, r) in enumerate(mg.G1_CENTRES):
            for alpha0 in (0.0, 17.0):                              # the same sampling angles reached through alpha0
                shifted = [a + alpha0 for a in angles]
                rot = my.rotation_table(shifted, alpha0, s)
                if alpha0:                                          # (a + alpha0) - alpha0 is not a in floating point:
                    rot = my.rotation_table(angles, 0.0, s)         # the fixture's angles are what the table must hold
                d = pm_ctx.debug_point(c, r, 200.0, 200.0, 20.0, s, alpha0, shifted, rot=rot, flags=flags)
                np.testing.assert_array_equal(d['templates'], exp[:, ci], err_msg='s=%d centre=%r alpha0=%r' % (s, (c, r), alpha0))


@pytest.mark.parametrize('no_table', [False, True])
@pytest.mark.parametrize('s', [34, 35])
def test_g3c_use_mcc_at_large_scene_rotations(pm_ctx, monkeypatch, s, no_table):
    """Fixture G3c (the reference's use_mcc, alpha0 = 30 / 90 / -137.5 degrees, 3 and 15 angles, rot_order 0 and 1): c2, r2, a, r
    bit for bit, h to 1e-5 (north_star's tolerance), NaN rows identical."""
    if no_table:
        monkeypatch.setenv('SID_PM_NO_SAMP_TABLE', '1')
    g = np.load(os.path.join(GOLD, 'g3c_large_rotations.npz'))
    g3 = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    assert syn.sha256(img1, img2) == str(g['pair_sha'])
    v = [g3[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    pm_ctx.upload_pair(img1, img2)
    n_nan = 0
    for ai, alpha0 in enumerate(mg.G3C_ALPHA0):
        for k, angles in enumerate(mg.G3C_ANGLE_SETS):
            for order in (0, 1):
                exp = g['out_s%d_a%d_k%d_o%d' % (s, ai, k, order)]
                pm_ctx.set_points(*v, s, alpha0, angles, rot=my.rotation_table(angles, alpha0, s),
                                  flags=_capi.HES_NORM | (_capi.ROT_ORDER1 if order else 0))
                pm_ctx.run()
                got, ij = pm_ctx.fetch()
                nan = np.isnan(exp[:, 0])
                tag = 's=%d alpha0=%r K=%d order=%d' % (s, alpha0, len(angles), order)
                np.testing.assert_array_equal(np.isnan(got[:, 0]), nan, err_msg=tag)
                np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4], err_msg=tag)
                np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5, err_msg=tag)
                assert (ij[nan] == -1).all() and (ij[~nan] >= 0).all()
                n_nan += int(nan.sum())
    assert n_nan > 0


def test_large_rotations_through_the_python_api():
    """The public single-point call with a scene rotation of 90 degrees (alpha0 reaches the kernel through pm_dispatch)."""
    g = np.load(os.path.join(GOLD, 'g3c_large_rotations.npz'))
    g3 = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    exp = g['out_s35_a1_k0_o0']
    i = int(np.flatnonzero(~np.isnan(exp[:, 0]))[5])
    got = my.use_mcc(g3['c1'][i], g3['r1'][i], g3['c2fg'][i], g3['r2fg'][i], g3['border'][i], img1, img2, 35, 90.0,
                     angles=mg.G3C_ANGLE_SETS[0])
    assert got[:4] == tuple(exp[i, :4]) and abs(got[4] - exp[i, 4]) < 1e-5
