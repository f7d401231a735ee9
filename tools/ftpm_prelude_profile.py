"""Config 4 (FT -> PM): where the first-guess prelude of get_drift_PM spends its time - SciPy Delaunay of the key points,
the device evaluation (sid_fg_interp_linear), the queries left flagged for SciPy.  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from sea_ice_drift_amd import synthetic as syn, _capi, lib
from sea_ice_drift_amd.domain import ArrayNansat
from sea_ice_drift_amd.seaicedrift import SeaIceDrift
size, grid = 10000, 200
img1, img2 = syn.make_pair(size, size, speckle=0.03)
scale = 4e-4
n1 = ArrayNansat(img1, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
n2 = ArrayNansat(img2, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
cg, rg = np.meshgrid(np.rint(np.linspace(100, size - 101, grid)), np.rint(np.linspace(100, size - 101, grid)))
sid = SeaIceDrift(n1, n2)
uft, vft, lon1ft, lat1ft, lon2ft, lat2ft = sid.get_drift_FT(max_drift=3600.0, nFeatures=100000)
x1, y1 = n1.transform_points(lon1ft, lat1ft, 1); x2, y2 = n2.transform_points(lon2ft, lat2ft, 1)
from scipy.spatial import Delaunay
src = np.array([y1, x1]).T; dst = np.array([rg.ravel(), cg.ravel()]).T
for rep in range(3):
    t0 = time.perf_counter(); tri = Delaunay(src); t1 = time.perf_counter()
    vals = np.array([x2, y2], dtype=np.float64).T
    both, sx, doubt = _capi.fg_interp_linear(tri.points, tri.simplices, vals, dst, device=0, details=True); t2 = time.perf_counter()
    print('points %d simplices %d | Delaunay %.1f ms | fg_interp_linear %.1f ms | flagged left %d of %d | integer keypoints %.2f'
          % (len(src), len(tri.simplices), (t1 - t0) * 1e3, (t2 - t1) * 1e3, int(doubt.sum()), len(dst), float((src == np.rint(src)).all(1).mean())))
t0 = time.perf_counter(); r = lib.interpolation_near(x1, y1, x2, y2, cg, rg, first_guess_device=0); print('interpolation_near %.1f ms' % ((time.perf_counter() - t0) * 1e3))
