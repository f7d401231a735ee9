#!/usr/bin/env python3
"""cProfile of one warm pattern_matching call on the e2e benchmark inputs (tools/e2e_bench.py)."""
import contextlib, cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import pmlib, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat
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
angles = list(range(-7, 8))
call = lambda: pmlib.pattern_matching(lon, lat, n1, c1, r1, n2, c2, r2, img_size=34, angles=angles)
with contextlib.redirect_stdout(io.StringIO()):
    call(); call()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
print('warm calls ms:', [round(t * 1e3, 1) for t in ts])
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
