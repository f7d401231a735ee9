"""Probe: template sides 50..64 of the classic kernel against the C oracle (one GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sea_ice_drift_amd import _capi, synthetic as syn
from oracle import c_oracle, pm_oracle as po


def rot_for(angles, alpha0, s):
    return np.array([po.rotation_terms(a - alpha0, s) for a in angles])


ctx = _capi.PMContext()
bad = 0
for s, nang, border in [(50, 3, 'mixed'), (51, 7, 20), (56, 15, 'mixed'), (63, 3, 30), (64, 5, 'mixed'), (64, 15, 40), (57, 17, 24)]:
    img1, img2 = syn.make_pair(700, 700, seed=21 + s)
    g = syn.make_grid(700, 700, 7, margin=140, border=border)
    half = nang // 2
    angles = [0.5 * k for k in range(-half, nang - half)]
    rot = rot_for(angles, 1.25, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 1.25, angles, rot=rot, nthreads=8)
    try:
        ctx.upload_pair(img1, img2)
        ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 1.25, angles, rot=rot)
        ctx.run()
        got, got_ij = ctx.fetch()
    except Exception as e:
        print(s, nang, border, 'ERROR', e); bad += 1; continue
    fin = np.isfinite(exp[:, 0])
    same_ij = (got_ij == exp_ij).all(axis=1)
    same4 = np.array([np.array_equal(got[i, :4], exp[i, :4], equal_nan=True) for i in range(len(exp))])
    dh = np.nanmax(np.abs(got[:, 4] - exp[:, 4])) if fin.any() else 0.0
    print('s=%d K=%d border=%s: n=%d finite=%d ij equal %d, (c2,r2,a,r) equal %d, max|dh| %.2e' % (s, nang, border, len(exp), fin.sum(), same_ij.sum(), same4.sum(), dh))
    if not (same_ij.all() and same4.all() and dh < 1e-5): bad += 1
print('BAD' if bad else 'ALL OK', bad)
