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
