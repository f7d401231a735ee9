#!/usr/bin/env python3
"""End-to-end `pattern_matching` (host prelude + GPU dispatch + postlude; the reference's timer spans
pmlib.py:393-450) on the benchmark pair with synthetic feature-tracking points.  Prints one JSON line."""
import contextlib, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import pmlib, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat

size, grid = 10000, 200
img1, img2 = syn.make_pair(size, size)
n1, n2 = ArrayNansat(img1), ArrayNansat(img2)                      # shared georeference: alpha0 = 0
rng = np.random.default_rng(3)
nkp = 20000                                                        # feature-tracking vectors (first guess)
c1 = rng.uniform(60, size - 60, nkp); r1 = rng.uniform(60, size - 60, nkp)
dc, dr = syn.true_displacement(c1, r1)
c2 = c1 + dc + rng.normal(0, 1.0, nkp); r2 = r1 + dr + rng.normal(0, 1.0, nkp)
cg, rg = np.meshgrid(np.rint(np.linspace(100, size - 101, grid)), np.rint(np.linspace(100, size - 101, grid)))
lon, lat = n1.transform_points(cg.ravel(), rg.ravel(), 0)
lon, lat = lon.reshape(cg.shape), lat.reshape(cg.shape)
angles = list(range(-7, 8))

def timed(label, fn, reps=3):
    """Best of `reps` runs (the boxes' hosts are noisy: single shots of a 50 ms call scatter by 30 %)."""
    best, out = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return out, best

with contextlib.redirect_stdout(io.StringIO()):
    pmlib.pattern_matching(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34, angles=angles)   # warm-up (library load)
    pre, t_pre = timed('prelude', lambda: pmlib.pm_prelude(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34, angles=angles))
    gpi = pre['gpi']
    res, t_disp = timed('dispatch', lambda: pmlib.pm_dispatch(img1, img2, pre['c1pm1i'][gpi], pre['r1pm1i'][gpi], pre['c2fg'][gpi],
                                                              pre['r2fg'][gpi], pre['brd2'][gpi], 34, pre['alpha0'], angles=angles))
    post, t_post = timed('postlude', lambda: pmlib.pm_postlude(pre, res, n2))
    full, t_full = timed('all', lambda: pmlib.pattern_matching(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34, angles=angles))
    _, t_full_host = timed('all_host_fg', lambda: pmlib.pattern_matching(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34, angles=angles, first_guess_on='host'))
    _, t_pre_host = timed('prelude_host', lambda: pmlib.pm_prelude(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34, angles=angles, first_guess_on='host'))
u, v = full[0], full[1]
ok = np.isfinite(u)
tdc, tdr = syn.true_displacement(cg, rg)
err = np.hypot(u[ok] - tdc[ok], v[ok] - tdr[ok])
print(json.dumps({'metric': 'pattern_matching end to end, 200x200 grid on a 10000x10000 pair, K=15, %d FT points (best of 3 runs each)' % nkp,
                  'valid_points': int(ok.sum()), 'total_s': t_full, 'prelude_s': t_pre, 'total_s_first_guess_on_host': t_full_host, 'prelude_s_first_guess_on_host': t_pre_host,
                  'dispatch_s_incl_200MB_upload_and_context': t_disp, 'postlude_s': t_post,
                  'grid_points_per_s_end_to_end': float(ok.sum()) / t_full,
                  'median_abs_drift_error_px': float(np.median(err)), 'borders': [float(pre['brd2'][gpi].min()), float(pre['brd2'][gpi].max())]}))
