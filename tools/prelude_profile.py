#!/usr/bin/env python3
"""Line-level wall clock of pm_prelude on the e2e benchmark inputs (tools/e2e_bench.py): where the host time goes."""
import contextlib, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import pmlib, lib, synthetic as syn, _capi
from sea_ice_drift_amd.domain import ArrayNansat
from scipy.spatial import Delaunay

size, grid = 10000, 200
img1, img2 = syn.make_pair(size, size)
n1, n2 = ArrayNansat(img1), ArrayNansat(img2)
rng = np.random.default_rng(3)
nkp = 20000
c1 = rng.uniform(60, size - 60, nkp); r1 = rng.uniform(60, size - 60, nkp)
dc, dr = syn.true_displacement(c1, r1)
c2 = c1 + dc + rng.normal(0, 1.0, nkp); r2 = r1 + dr + rng.normal(0, 1.0, nkp)
cg, rg = np.meshgrid(np.rint(np.linspace(100, size - 101, grid)), np.rint(np.linspace(100, size - 101, grid)))
lon, lat = n1.transform_points(cg.ravel(), rg.ravel(), 0)
lon, lat = lon.reshape(cg.shape), lat.reshape(cg.shape)

def best(fn, reps=5):
    out, t = None, 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); t = min(t, time.perf_counter() - t0)
    return out, t

res = {}
with contextlib.redirect_stdout(io.StringIO()):
    pmlib.pm_prelude(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34)
    (c2pm1, r2pm1), res['transform_grid_to_image2'] = best(lambda: n2.transform_points(lon.flatten(), lat.flatten(), 1))
    c2i, r2i = np.round([c2pm1, r2pm1])
    _, res['transform_back_and_to_image1'] = best(lambda: n1.transform_points(*n2.transform_points(c2i, r2i), 1))
    (lon1, lat1), res['transform_keypoints'] = best(lambda: n1.transform_points(c1, r1))
    c1n2, r1n2 = n2.transform_points(lon1, lat1, 1)
    _, res['interpolation_poly'] = best(lambda: lib.interpolation_poly(c1n2, r1n2, c2, r2, c2i, r2i))
    src = np.array([r1n2, c1n2]).T
    tri, res['delaunay_%d_points' % nkp] = best(lambda: Delaunay(src))
    vals = np.array([c2, r2], dtype=np.float64).T
    dst = np.array([r2i, c2i]).T
    _, res['fg_interp_linear_device'] = best(lambda: _capi.fg_interp_linear(tri.points, tri.simplices, vals, dst.reshape(-1, 2), device=0))
    rq = np.round(r2i).astype(np.int16); cq = np.round(c2i).astype(np.int16)
    _, res['nearest_keypoint_distance_device'] = best(lambda: pmlib.nearest_keypoint_distance(c2, r2, rq, cq, shape=n2.shape(), device=0))
    _, res['get_initial_rotation'] = best(lambda: pmlib.get_initial_rotation(n1, n2))
    _, res['pm_prelude_total'] = best(lambda: pmlib.pm_prelude(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34))
print(json.dumps({k: round(v * 1e3, 3) for k, v in res.items()}))
