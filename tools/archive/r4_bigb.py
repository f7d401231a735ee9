"""Probe: search borders 69 .. 100 (big layouts of the row-pair kernel) against the C oracle (one GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sea_ice_drift_amd import _capi, synthetic as syn
from oracle import c_oracle, pm_oracle as po


def rot_for(angles, alpha0, s):
    return np.array([po.rotation_terms(a - alpha0, s) for a in angles])


ctx = _capi.PMContext()
bad = 0
size = 1600
img1, img2 = syn.make_pair(size, size, seed=29)
ctx.upload_pair(img1, img2)
rng = np.random.default_rng(4)
for s, angles, flags, borders in [(34, list(range(-7, 8)), 1, [60, 68, 69, 70, 75, 80, 85, 90, 95, 100, 100]),
                                  (35, [-3, 0, 3], 1, [69, 80, 93, 100]),
                                  (34, list(range(-3, 4)), 7, [72, 88]),
                                  (35, [0.5 * k for k in range(-8, 9)], 1, [70, 99]),
                                  (34, list(range(-7, 8)), 3, [81, 100])]:
    borders = np.array(borders, dtype=np.float64)
    n = len(borders)
    c1 = np.rint(rng.uniform(400, size - 400, n)); r1 = np.rint(rng.uniform(400, size - 400, n))
    dc, dr = syn.true_displacement(c1, r1)
    c2 = c1 + np.rint(dc) + rng.integers(-2, 3, n); r2 = r1 + np.rint(dr) + rng.integers(-2, 3, n)
    rot = rot_for(angles, 0.0, s)
    t0 = time.time()
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    t1 = time.time()
    try:
        ctx.set_points(c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot, flags=flags)
        ctx.run()
        got, got_ij = ctx.fetch()
    except Exception as e:
        print(s, len(angles), flags, 'ERROR', e); bad += 1; continue
    fin = np.isfinite(exp[:, 0])
    same_ij = (got_ij == exp_ij).all(axis=1)
    same4 = np.array([np.array_equal(got[i, :4], exp[i, :4], equal_nan=True) for i in range(n)])
    dh = np.nanmax(np.abs(got[:, 4] - exp[:, 4])) if fin.any() else 0.0
    print('s=%d K=%d flags=%d: n=%d finite=%d ij equal %d, (c2,r2,a,r) equal %d, max|dh| %.2e  (oracle %.1f s)' % (s, len(angles), flags, n, fin.sum(), same_ij.sum(), same4.sum(), dh, t1 - t0))
    if not (same_ij.all() and same4.all() and dh < 1e-5):
        bad += 1
        print(np.c_[borders, got_ij, exp_ij, got[:, 3:5], exp[:, 3:5]])
print('BAD' if bad else 'ALL OK', bad)
