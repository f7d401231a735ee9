"""GPU parity of the uint8 staging step (include/sid_stage.h, sea_ice_drift_amd.lib.get_uint8_image; replaces
lib.py:27-59) against the reference's own outputs (g6 fixture), the oracle and NumPy."""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import pytest

from oracle import stage_oracle as so
from sea_ice_drift_amd import lib

pytestmark = pytest.mark.gpu
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


def quiet(fn, *a):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a)


def test_reference_fixture_bit_exact():
    mg = golden_module()
    g = np.load(os.path.join(HERE, 'golden', 'g6_uint8_image.npz'))
    for ci, img in enumerate(mg.g6_cases()):
        for pi, (vmin, vmax, pmin, pmax) in enumerate(mg.G6_PARAMS):
            out = quiet(lib.get_uint8_image, img.copy(), vmin, vmax, pmin, pmax)
            np.testing.assert_array_equal(out, g['out_%d_%d' % (ci, pi)], err_msg='case %d params %d' % (ci, pi))


def test_large_image_against_oracle_and_properties():
    rng = np.random.default_rng(77)
    img = rng.normal(-21.0, 5.0, (3000, 2500)).astype(np.float32)
    img[rng.random(img.shape) < 0.05] = np.nan
    img[1000:1100, 500:900] = np.nan
    out = quiet(lib.get_uint8_image, img, None, None, 10, 99)
    exp, vmin, vmax = so.get_uint8_image(img.copy(), None, None, 10, 99)
    np.testing.assert_array_equal(out, exp)
    assert vmin == np.nanpercentile(img, 10) and vmax == np.nanpercentile(img, 99)
    assert (out[np.isnan(img)] == 0).all() and out[np.isfinite(img)].min() == 1 and out.max() == 255
    # torch tensors stay on the device, strided views work
    import torch
    t = torch.from_numpy(img).cuda()
    view = t[10:2000, 100:2100]
    out_t = quiet(lib.get_uint8_image, view, -30.0, -12.0, 10, 99)
    exp_v, _, _ = so.get_uint8_image(img[10:2000, 100:2100].copy(), -30.0, -12.0, 10, 99)
    assert out_t.is_cuda
    np.testing.assert_array_equal(out_t.cpu().numpy(), exp_v)


def test_degenerate_images():
    allnan = np.full((40, 50), np.nan, dtype=np.float32)
    out = quiet(lib.get_uint8_image, allnan, None, None, 10, 99)
    assert (out == 0).all()
    const = np.full((33, 21), -17.5, dtype=np.float32)              # vmax == vmin: 0/0 -> NaN -> 0 here (undefined in C)
    with np.errstate(all='ignore'):
        out = quiet(lib.get_uint8_image, const, None, None, 10, 99)
    exp, _, _ = so.get_uint8_image(const.copy(), None, None, 10, 99)
    np.testing.assert_array_equal(out, exp)
    with pytest.raises(NotImplementedError):
        lib.get_uint8_image(np.zeros((4, 4)), 0.0, 1.0, 10, 99)       # float64: different arithmetic, not offered


def _images_for_order_statistics():
    rng = np.random.default_rng(91)
    sar = rng.normal(-22.0, 4.0, (1500, 2048)).astype(np.float32)
    sar[rng.random(sar.shape) < 0.05] = np.nan
    ties = np.round(rng.normal(0.0, 3.0, (900, 1000))).astype(np.float32)          # a few dozen distinct values
    gap = np.where(rng.random((1200, 800)) < 0.1, rng.normal(-50.0, 0.1, (1200, 800)), rng.normal(40.0, 0.1, (1200, 800))).astype(np.float32)
    inf = rng.normal(0.0, 1.0, (700, 1300)).astype(np.float32)
    inf[::7, ::5] = np.inf; inf[::11, ::3] = -np.inf; inf[::13, ::2] = np.nan
    lin = (10.0 ** (rng.normal(-2.2, 0.4, (1000, 1024)))).astype(np.float32)       # linear sigma0: several binades
    small = rng.normal(5.0, 1.0, (37, 41)).astype(np.float32)                      # fewer pixels than the sample
    const = np.full((300, 400), 3.25, dtype=np.float32)
    odd = rng.normal(-5.0, 2.0, (513, 1027)).astype(np.float32)                    # no 16-byte rows
    return {'sar': sar, 'ties': ties, 'gap': gap, 'inf': inf, 'lin': lin, 'small': small, 'const': const, 'odd': odd}


def test_hinted_first_pass_gives_the_same_order_statistics(monkeypatch):
    """Round 4 (include/sid_stage.h sid_stage_begin_hint): a sample brackets the hinted fractions by key ranges and the first
    pass counts inside them, which saves the radix select one pass.  Exact whatever the image and however wrong the hint:
    against np.sort, against the plain three-pass route, with hints that hold, hints that miss, ties, gaps, infinities."""
    import torch
    from sea_ice_drift_amd import _capi
    ws = _capi.StageWorkspace(0)
    st = torch.cuda.current_stream().cuda_stream
    for name, img in _images_for_order_statistics().items():
        t = torch.from_numpy(img).cuda()
        rows, cols = img.shape
        srt = np.sort(img[~np.isnan(img)].ravel())
        nv = len(srt)
        for fractions in ([0.10, 0.99], [0.5], [0.0, 1.0], [0.001, 0.25, 0.75, 0.999]):
            ranks = sorted({min(nv - 1, max(0, int(f * (nv - 1)) + d)) for f in fractions for d in (0, 1)})
            far = sorted({nv // 3, (2 * nv) // 3, 0, nv - 1})                              # ranks the hint knows nothing about
            assert ws.begin(t.data_ptr(), rows, cols, cols, st, fractions=fractions) == nv, name
            got = ws.order_stats(ranks)
            got_far = ws.order_stats(far)
            assert ws.begin(t.data_ptr(), rows, cols, cols, st) == nv
            plain = ws.order_stats(ranks)
            np.testing.assert_array_equal(got, srt[ranks], err_msg='%s %s' % (name, fractions))
            np.testing.assert_array_equal(got_far, srt[far], err_msg='%s %s (ranks outside the hint)' % (name, fractions))
            np.testing.assert_array_equal(plain, srt[ranks])
        # the whole staging call, with the hint and without (SID_STAGE_NO_HINT), against the oracle
        if name != 'const':
            exp, _, _ = so.get_uint8_image(img.copy(), None, None, 10, 99)
            np.testing.assert_array_equal(quiet(lib.get_uint8_image, img, None, None, 10, 99), exp, err_msg=name)
            monkeypatch.setenv('SID_STAGE_NO_HINT', '1')
            np.testing.assert_array_equal(quiet(lib.get_uint8_image, img, None, None, 10, 99), exp, err_msg=name)
            monkeypatch.delenv('SID_STAGE_NO_HINT')
    ws.close()
