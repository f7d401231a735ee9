"""CPU tests of the seeded synthetic pair / grid generator used by bench.py and the parity tests."""
import numpy as np

from sea_ice_drift_amd import synthetic as syn


def test_pair_is_deterministic_and_valid():
    a1, b1 = syn.make_pair(300, 320, seed=42)
    a2, b2 = syn.make_pair(300, 320, seed=42)
    assert syn.sha256(a1, b1) == syn.sha256(a2, b2)
    assert a1.dtype == np.uint8 and a1.shape == (300, 320)
    assert a1.min() >= 1 and b1.min() >= 1                       # 0 is the reference's invalid value
    a3, _ = syn.make_pair(300, 320, seed=43)
    assert syn.sha256(a1) != syn.sha256(a3)


def test_grid_vectors():
    g = syn.make_grid(1000, 1000, 40)
    assert all(v.shape == (1600,) and v.dtype == np.float64 for v in g.values())
    assert g['border'].min() >= 20 and g['border'].max() <= 50
    assert (g['border'] == np.floor(g['border'])).all() and (g['c2fg'] == np.rint(g['c2fg'])).all()
    dc, dr = syn.true_displacement(g['c1'], g['r1'])
    assert np.abs(g['c2fg'] - g['c1'] - dc).max() <= 3.5 and np.abs(g['r2fg'] - g['r1'] - dr).max() <= 3.5
    g2 = syn.make_grid(1000, 1000, (80, 40), border=20)
    assert g2['c1'].shape == (3200,) and (g2['border'] == 20).all()
