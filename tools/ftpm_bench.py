#!/usr/bin/env python3
"""BASELINE config 4 without the ORB detector (OpenCV's): synthetic key points + descriptors -> feature tracking
(GPU matcher, host filters) -> first guess -> pattern matching on the 10000x10000 benchmark pair.  One JSON line."""
import contextlib, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import ftlib, pmlib, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat

size, grid, nkp = 10000, 200, 24000
img1, img2 = syn.make_pair(size, size)
scale = 4e-4
n1 = ArrayNansat(img1, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
n2 = ArrayNansat(img2, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
rng = np.random.default_rng(9)
xy1 = rng.uniform(60, size - 60, (nkp, 2))
d1 = rng.integers(0, 256, (nkp, 32), dtype=np.uint8)
dc, dr = syn.true_displacement(xy1[:, 0], xy1[:, 1])
ncommon = 18000
xy2 = np.concatenate([xy1[:ncommon] + np.stack([dc, dr], 1)[:ncommon] + rng.normal(0, 0.5, (ncommon, 2)),
                      rng.uniform(60, size - 60, (nkp - ncommon - 1000, 2))])
flips = (rng.random((ncommon, 256)) < 0.06).astype(np.uint8)
d2 = np.concatenate([d1[:ncommon] ^ np.packbits(flips, axis=1), rng.integers(0, 256, (len(xy2) - ncommon, 32), dtype=np.uint8)])
perm = rng.permutation(len(xy2)); xy2, d2 = xy2[perm], d2[perm]
feeds = []
def finder(image, **kw):
    return feeds.pop(0)
cg, rg = np.meshgrid(np.rint(np.linspace(100, size - 101, grid)), np.rint(np.linspace(100, size - 101, grid)))
lon, lat = n1.transform_points(cg.ravel(), rg.ravel(), 0)
lon, lat = lon.reshape(cg.shape), lat.reshape(cg.shape)
angles = list(range(-7, 8))
out = {}
with contextlib.redirect_stdout(io.StringIO()):
    for rep in range(2):                                             # first pass warms the library up
        feeds[:] = [(xy1, d1), (xy2, d2)]
        t0 = time.perf_counter()
        x1, y1, x2, y2 = ftlib.feature_tracking(n1, n2, find_key_points=finder, max_drift=2000.0)
        t_ft = time.perf_counter() - t0
        t0 = time.perf_counter()
        u, v, a, r, h, lon2, lat2 = pmlib.pattern_matching(lon, lat, n1, x1, y1, n2, x2, y2, img_size=34, angles=angles)
        t_pm = time.perf_counter() - t0
ok = np.isfinite(u)
tdc, tdr = syn.true_displacement(cg, rg)
uu, vv = u / scale, -v / scale                                       # degrees -> pixels of this georeference
err = np.hypot(uu[ok] - tdc[ok], vv[ok] - tdr[ok])
print(json.dumps({'metric': 'FT (matcher + filters) feeding PM, 10000x10000 pair, 200x200 grid, K=15; ORB excluded (OpenCV)',
                  'key_points': [len(xy1), len(xy2)], 'ft_vectors': int(len(x1)), 'ft_s': t_ft, 'pm_s': t_pm,
                  'valid_grid_points': int(ok.sum()), 'median_abs_drift_error_px': float(np.median(err))}))
