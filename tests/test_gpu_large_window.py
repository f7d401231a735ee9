"""GPU parity of the large-window pipeline (csrc/pm_large.hip): ``rotate_and_match`` as a call of its own (reference
pmlib.py:117-174; the reference's own test passes the WHOLE image 2 as the window, tests.py:336-337), search borders beyond the
LDS launch classes (> 111 px), template sides above 64, and the per-point adaptors ``get_template`` / ``get_hessian`` /
``get_distance_to_nearest_keypoint``.  Fixture G8 = outputs of the imported reference (tests/golden/make_golden.py); the C oracle
checks whole NCC matrices bit for bit at other shapes."""
import os

import numpy as np
import pytest

from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def flags_of(kw):
    return ((1 if kw.get('hes_norm', True) else 0) | (2 if kw.get('hes_smth') else 0) | (4 if kw.get('mcc_norm') else 0) |
            (8 if kw.get('rot_order') == 1 else 0))


@pytest.mark.parametrize('case', mg.G8_RAM, ids=[c[0] for c in mg.G8_RAM])
def test_g8_rotate_and_match_of_the_reference(case):
    """The public call with the reference's signature against what the reference returned: dc, dr, best_a exact; best_r exact
    (1e-5 under mcc_norm); best_h to 1e-5; best_result bit for bit (sha256 of the float32 matrix); best_template bit for bit."""
    name, c1, r1, s, win, alpha0, kw = case
    g = np.load(os.path.join(GOLD, 'g8_rotate_and_match.npz'))
    img1, img2 = mg.g8_pair()
    assert syn.sha256(img1, img2) == str(g['pair_sha'])
    image2 = img2 if win is None else img2[win[0]:win[1], win[2]:win[3]]
    out = my.rotate_and_match(img1, c1, r1, s, image2, alpha0, **kw)
    exp = g['ram_%s_scalars' % name]
    assert len(out) == 7
    if np.isnan(exp[0]):
        assert all(isinstance(x, float) and np.isnan(x) for x in out)
        return
    dc, dr, a, r, h, ccm, tmpl = out
    assert (dc, dr, a) == tuple(exp[:3]), (out[:5], exp)
    assert a in kw['angles'] and isinstance(r, np.float32) and isinstance(h, np.float32)
    if kw.get('mcc_norm'):
        np.testing.assert_allclose(r, exp[3], rtol=1e-5, atol=1e-5)
    else:
        assert np.float64(r) == exp[3]
    np.testing.assert_allclose(h, exp[4], rtol=1e-5, atol=1e-5)
    assert ccm.dtype == np.float32 and ccm.shape == tuple(g['ram_%s_ccm_shape' % name])
    sha, sub = mg.ccm_digest(ccm)
    np.testing.assert_array_equal(sub, g['ram_%s_ccm_sub' % name])
    assert sha == str(g['ram_%s_ccm_sha' % name])
    assert tmpl.dtype == np.uint8
    np.testing.assert_array_equal(tmpl, g['ram_%s_template' % name])


def test_g8_use_mcc_beyond_the_lds_classes(pm_ctx):
    """Search borders of 112 / 160 / 250 px and template sides of 65 / 100 px: the reference's use_mcc through the public
    single-point call, and all of them in ONE batch together with ordinary points (set_points / run / fetch)."""
    g = np.load(os.path.join(GOLD, 'g8_rotate_and_match.npz'))
    img1, img2 = mg.g8_pair()
    for name, c1, r1, c2fg, r2fg, b, s, alpha0, kw in mg.G8_MCC:
        exp = g['mcc_%s' % name]
        got = my.use_mcc(c1, r1, c2fg, r2fg, b, img1, img2, s, alpha0, **kw)
        assert got[:3] == tuple(exp[:3]), (name, got, exp)
        if kw.get('mcc_norm'):
            np.testing.assert_allclose(got[3], exp[3], rtol=1e-5, atol=1e-5)
        else:
            assert np.float64(got[3]) == exp[3], name
        np.testing.assert_allclose(got[4], exp[4], rtol=1e-5, atol=1e-5, err_msg=name)
    # one batch: borders 20 .. 250 at 34 px, 3 angles (the b112 case's parameters), with a point outside image 2 and a zero-pixel point
    name, c1, r1, c2fg, r2fg, b, s, alpha0, kw = mg.G8_MCC[0]
    borders = np.array([20.0, 112.0, 50.0, 160.0, 111.0, 250.0, 30.0, 200.0, 20.0])
    n = len(borders)
    vc1 = np.full(n, c1); vr1 = np.full(n, r1); vc2 = np.full(n, c2fg); vr2 = np.full(n, r2fg)
    vc2[6] = 10.0                                                   # window outside image 2 -> NaN
    vc1[7], vr1[7] = 30.0, 615.0                                    # template on the zero corner -> NaN (a large point)
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(vc1, vr1, vc2, vr2, borders, s, alpha0, kw['angles'])
    pm_ctx.run()
    got, ij = pm_ctx.fetch()
    from oracle import c_oracle
    exp, exp_ij = c_oracle.pm_batch(img1, img2, vc1, vr1, vc2, vr2, borders, s, alpha0, kw['angles'],
                                    rot=my.rotation_table(kw['angles'], alpha0, s), nthreads=8)
    np.testing.assert_array_equal(ij, exp_ij)
    assert np.isnan(exp[6]).all() and np.isnan(exp[7]).all() and np.isfinite(exp[[0, 1, 2, 3, 4, 5, 8]]).all()
    np.testing.assert_array_equal(got[:, :4], exp[:, :4])
    np.testing.assert_allclose(got[:, 4], exp[:, 4], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(got[1], np.r_[g['mcc_b112'][:4], got[1, 4]])   # (and the reference itself for the b112 point)


SHAPES = [  # s, K, window (rows, cols), flags
    (2, 1, (40, 33), 1), (17, 3, (80, 100), 1), (34, 15, (300, 280), 1), (35, 16, (120, 200), 5), (34, 17, (100, 100), 3),
    (63, 7, (200, 130), 1), (64, 2, (129, 257), 7), (65, 3, (131, 140), 1), (96, 40, (150, 160), 1), (129, 4, (300, 260), 9),
    (200, 2, (330, 420), 1), (255, 1, (400, 300), 1), (50, 7, (51, 52), 1), (34, 3, (36, 700), 15),
]


@pytest.mark.parametrize('s,K,shape,flags', SHAPES)
def test_whole_ncc_matrix_against_the_c_oracle(pm_ctx, c_oracle, s, K, shape, flags):
    """Template sides 2 .. 255, 1 .. 40 angles (one to three groups of matrix-instruction slots), rectangular windows down to
    2 x 2 placements, every flag: out5 / ij / the winning NCC matrix / the winning template against the C oracle."""
    rng = np.random.default_rng(1000 + s + K)
    img1, img2 = syn.make_pair(760, 800, seed=s * 7 + K)
    r0, c0 = int(rng.integers(0, 760 - shape[0])), int(rng.integers(0, 800 - shape[1]))
    image2 = np.ascontiguousarray(img2[r0:r0 + shape[0], c0:c0 + shape[1]])
    angles = list(np.round(rng.uniform(-12, 12, K), 2))
    alpha0 = 7.5
    c1, r1 = 380.0 + float(rng.integers(-3, 4)) * 0.25, 400.0
    rot = my.rotation_table(angles, alpha0, s)
    exp = c_oracle.rotate_and_match(img1, c1, r1, s, image2, alpha0, angles, rot, flags=flags)
    pm_ctx.upload_pair(img1, img2)
    got = pm_ctx.rotate_and_match(c1, r1, s, alpha0, angles, rot=rot, flags=flags, window=(r0, c0, shape[0], shape[1]))
    np.testing.assert_array_equal(got['ij'], exp['ij'])
    assert exp['ij'][2] >= 0
    np.testing.assert_array_equal(got['ccm'], exp['ccm'])
    np.testing.assert_array_equal(got['template'], exp['template'])
    np.testing.assert_array_equal(got['out'][:3], exp['out'][:3])
    if flags & 4:
        np.testing.assert_allclose(got['out'][3], exp['out'][3], rtol=1e-5, atol=1e-5)
    else:
        assert got['out'][3] == exp['out'][3]
    np.testing.assert_allclose(got['out'][4], exp['out'][4], rtol=1e-5, atol=1e-5)


def test_degenerate_windows_and_templates(pm_ctx, c_oracle):
    """A constant template (dT = 0: the matrix is all ones, the first placement wins), a window without variance (all zeros
    in the matrix), exact ties between angles (the first angle keeps them), a perfect match."""
    img1, img2 = syn.make_pair(300, 300, seed=77)
    img1 = img1.copy(); img2 = img2.copy()
    img1[100:180, 100:180] = 93                                      # constant template
    img2[0:120, 0:120] = 50                                          # flat window
    img2[150:250, 150:250] = img1[20:120, 30:130]                    # a perfect match for the template around (80.., 70..)
    pm_ctx.upload_pair(img1, img2)
    for (c1, r1, s, win, angles) in ((140.0, 140.0, 34, (130, 130, 120, 150), [-3, 0, 3]),
                                     (60.0, 250.0, 34, (0, 0, 110, 100), [0, 0, 1]),
                                     (80.0, 70.0, 50, (120, 120, 170, 175), [0.0, 0.0])):
        rot = my.rotation_table(angles, 0.0, s)
        image2 = np.ascontiguousarray(img2[win[0]:win[0] + win[2], win[1]:win[1] + win[3]])
        exp = c_oracle.rotate_and_match(img1, c1, r1, s, image2, 0.0, angles, rot, flags=0)
        got = pm_ctx.rotate_and_match(c1, r1, s, 0.0, angles, rot=rot, flags=0, window=win)
        np.testing.assert_array_equal(got['ij'], exp['ij'])
        np.testing.assert_array_equal(got['ccm'], exp['ccm'])
        np.testing.assert_array_equal(got['out'], exp['out'])
    assert got['out'][3] == 1.0 and got['ij'][2] == 0


def test_every_point_through_the_large_pipeline_equals_fixture_g3(pm_ctx, monkeypatch):
    """SID_PM_ALL_LARGE=1 sends ordinary points (borders 20 .. 50) through the large-window pipeline: the reference's G3 results."""
    monkeypatch.setenv('SID_PM_ALL_LARGE', '1')
    g = np.load(os.path.join(GOLD, 'g3_use_mcc.npz'))
    img1, img2 = mg.g3_pair()
    v = [g[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    pm_ctx.upload_pair(img1, img2)
    for s, alpha0 in ((34, 0.0), (35, -3.85)):
        for k, angles in enumerate(mg.G3_ANGLE_SETS):
            for mcc in ((0, 1) if k == 1 else (0,)):
                exp = g['out_s%d_k%d_m%d' % (s, k, mcc)]
                pm_ctx.set_points(*v, s, alpha0, angles, flags=1 | (4 if mcc else 0))
                pm_ctx.run()
                got, ij = pm_ctx.fetch()
                nan = np.isnan(exp[:, 0])
                np.testing.assert_array_equal(np.isnan(got[:, 0]), nan)
                assert nan.any() and (ij[nan] == -1).all()
                np.testing.assert_array_equal(got[~nan, :3], exp[~nan, :3])
                if mcc:
                    np.testing.assert_allclose(got[~nan, 3:], exp[~nan, 3:], rtol=1e-5, atol=1e-5)
                else:
                    np.testing.assert_array_equal(got[~nan, 3], exp[~nan, 3])
                    np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)


def test_template_sides_above_64_in_a_batch(pm_ctx, c_oracle):
    """img_size = 100: no one-workgroup-per-point kernel exists; valid points, a point outside image 2, a zero-pixel point."""
    img1, img2 = syn.make_pair(900, 900, seed=31)
    img1 = img1.copy()
    img1[0:60, 0:60] = 0
    c1 = np.array([450.0, 300.5, 40.0, 600.0, 450.0]); r1 = np.array([450.0, 500.25, 40.0, 300.0, 450.0])
    c2 = np.array([452.0, 301.0, 450.0, 30.0, 447.0]); r2 = np.array([449.0, 503.0, 450.0, 300.0, 452.0])
    b = np.array([20.0, 35.0, 20.0, 20.0, 60.0])
    angles = [-3, 0, 3]
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, b, 100, 1.0, angles, rot=my.rotation_table(angles, 1.0, 100), nthreads=8)
    assert np.isnan(exp[2]).all() and np.isnan(exp[3]).all() and np.isfinite(exp[[0, 1, 4]]).all()
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(c1, r1, c2, r2, b, 100, 1.0, angles)
    pm_ctx.run()
    got, ij = pm_ctx.fetch()
    np.testing.assert_array_equal(ij, exp_ij)
    np.testing.assert_array_equal(got[:, :4], exp[:, :4])
    np.testing.assert_allclose(got[:, 4], exp[:, 4], rtol=1e-5, atol=1e-5)
    # the public API takes the size as well
    out = my.pm_dispatch(img1, img2, c1, r1, c2, r2, b, 100, 1.0, angles=angles)
    np.testing.assert_array_equal(out[:, :4], exp[:, :4])
    with pytest.raises(NotImplementedError):
        my.pm_dispatch(img1, img2, c1, r1, c2, r2, b, 256, 1.0, angles=angles)


def test_get_template_adaptor_against_the_reference_fixtures():
    """get_template (pmlib.py:89-115) with the reference's signature: fixtures G1 (order 0) and G1b (order 1, incl. templates cut
    by the image border)."""
    img = mg.g1_image()
    for order, name in ((0, 'g1_templates.npz'), (1, 'g1b_templates_order1.npz')):
        g = np.load(os.path.join(GOLD, name))
        for s, key in ((34, 't34'), (35, 't35')):
            k = 0
            for a in mg.G1_ANGLES:
                for (c, r) in mg.G1_CENTRES:
                    got = my.get_template(img, c, r, a, s, rot_order=order)
                    assert got.dtype == np.uint8 and got.shape == (s, s)
                    np.testing.assert_array_equal(got, g[key][k].reshape(s, s), err_msg='order %d s %d a %r centre %r' % (order, s, a, (c, r)))
                    k += 1
        if order == 1:
            for k, (c, r, a, s) in enumerate(g['edge_args']):
                np.testing.assert_array_equal(my.get_template(img, c, r, a, int(s), rot_order=1), g['edge%d' % k])
        else:
            np.testing.assert_array_equal(my.get_template(img, 5, 5, 10, 34), g['edge'])
    # the reference's own test (tests.py:296-309): shapes of a 50 px template at 0 and 30 degrees
    assert my.get_template(img, 100, 300, 0, 50).shape == (50, 50) and my.get_template(img, 100, 300, 30, 50).shape == (50, 50)
    with pytest.raises(RuntimeError):
        my.get_template(img, 100, 300, 0, 50, rot_order=6)


@pytest.mark.parametrize('n', [42, 72, 101, 102])
def test_get_hessian_adaptor_against_the_reference_fixture(n):
    """get_hessian (pmlib.py:36-59): raw magnitudes bit for bit, normalised / smoothed to 1e-5 (np.std from float64 sums)."""
    g = np.load(os.path.join(GOLD, 'g2_hessian.npz'))
    m = g['in%d' % n]
    np.testing.assert_array_equal(my.get_hessian(m, hes_norm=False), g['raw%d' % n])
    np.testing.assert_allclose(my.get_hessian(m), g['norm%d' % n], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(my.get_hessian(m, hes_norm=True, hes_smth=True), g['smth%d' % n], rtol=1e-5, atol=1e-5)
    with pytest.raises(NotImplementedError):
        my.get_hessian(m.astype(np.float64))


def test_get_distance_to_nearest_keypoint_adaptor():
    """The full-resolution distance image (pmlib.py:61-77; tests.py:311-321 asserts its shape) against scipy's EDT."""
    from scipy import ndimage as nd
    rng = np.random.default_rng(5)
    for shape, nk in (((300, 412), 700), ((97, 64), 3), ((50, 50), 1)):
        x1 = rng.uniform(0, shape[1] - 1, nk); y1 = rng.uniform(0, shape[0] - 1, nk)
        seed = np.zeros(shape, dtype=bool)
        seed[np.uint16(y1), np.uint16(x1)] = True
        exp = nd.distance_transform_edt(~seed, return_distances=True, return_indices=False)
        got = my.get_distance_to_nearest_keypoint(x1, y1, shape)
        assert got.shape == shape and got.dtype == np.float64
        np.testing.assert_array_equal(got, exp)
    with pytest.raises(IndexError):
        my.get_distance_to_nearest_keypoint([10.0, 500.0], [10.0, 10.0], (100, 100))


def test_rotate_and_match_argument_errors(pm_ctx):
    img1, img2 = syn.make_pair(200, 200, seed=1)
    with pytest.raises(ValueError):                                 # one placement along an axis: np.gradient raises in the reference
        my.rotate_and_match(img1, 100, 100, 50, img2[:50, :80], 0)
    with pytest.raises(RuntimeError):
        my.rotate_and_match(img1, 100, 100, 50, img2, 0, rot_order=6)
    with pytest.raises(NotImplementedError):
        my.rotate_and_match(img1, 100, 100, 50, img2, 0, template_matcher=lambda *a: None)
    pm_ctx.upload_pair(img1, img2)
    with pytest.raises(_capi.SidPmError) as e:                      # window outside image 2
        pm_ctx.rotate_and_match(100, 100, 34, 0.0, [0.0], window=(150, 150, 60, 60))
    assert e.value.code == -1
    # the C ABI requires the rotation terms (include/sid_pm.h): a NULL `rot` is an argument error, not a libm fallback
    import ctypes as C
    L = _capi.lib()
    one = (C.c_double * 1)(100.0)
    b = (C.c_double * 1)(20.0)
    ang = (C.c_double * 1)(0.0)
    rc = L.sid_pm_set_points(pm_ctx._h, one, one, one, one, b, 1, 34, 0.0, ang, None, 1, 1)
    assert rc == -1 and b'rot is required' in L.sid_pm_last_error()
