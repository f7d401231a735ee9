"""CPU: the staging oracle (oracle/stage_oracle.py, restated lib.py:27-59 incl. NumPy's float32 nanpercentile)
against the reference's own outputs (tests/golden/g6_uint8_image.npz) and against np.nanpercentile."""
import importlib.util
import os
import sys

import numpy as np

from oracle import stage_oracle as so

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


def test_oracle_equals_reference_outputs():
    mg = golden_module()
    g = np.load(os.path.join(HERE, 'golden', 'g6_uint8_image.npz'))
    for ci, img in enumerate(mg.g6_cases()):
        for pi, (vmin, vmax, pmin, pmax) in enumerate(mg.G6_PARAMS):
            out, v0, v1 = so.get_uint8_image(img.copy(), vmin, vmax, pmin, pmax)
            np.testing.assert_array_equal(out, g['out_%d_%d' % (ci, pi)], err_msg='case %d params %d' % (ci, pi))
            np.testing.assert_array_equal(np.float64(v0), g['vmin_%d_%d' % (ci, pi)])
            np.testing.assert_array_equal(np.float64(v1), g['vmax_%d_%d' % (ci, pi)])


def test_restated_nanpercentile_equals_numpy():
    rng = np.random.default_rng(12)
    for trial in range(60):
        shape = (int(rng.integers(1, 300)), int(rng.integers(1, 300)))
        a = rng.normal(-20, 5, shape).astype(np.float32)
        a[rng.random(shape) < rng.random() * 0.4] = np.nan
        if trial % 9 == 0:
            a.flat[rng.integers(0, a.size)] = np.inf
        for p in (0, 1, 10, 50, 99, 99.9, 100):
            with np.errstate(all='ignore'):
                ref = np.nanpercentile(a, p) if np.isfinite(a).any() or np.isinf(a).any() else np.float32(np.nan)
                got = so.nanpercentile(a, p)
            assert (np.isnan(ref) and np.isnan(got)) or ref == got, (trial, p, ref, got)
