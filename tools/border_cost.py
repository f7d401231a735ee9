#!/usr/bin/env python3
"""Nanoseconds per grid point as a function of the search border (one GPU, 40 000 points of one border each, pair
resident, kernels + fetch): the cost table behind dist.shard_indices_by_cost.  Usage: python3 tools/border_cost.py [K=15]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table
half = (int(sys.argv[1]) - 1) // 2 if len(sys.argv) > 1 else 7
size, grid, s = 10000, 200, 34
img1, img2 = syn.make_pair(size, size)
angles = list(range(-half, half + 1)); rot = rotation_table(angles, 0.0, s)
res = {}
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(img1, img2)
    for b in [20, 21, 22] + list(range(20, 51)):                       # (the first three again: the clocks ramp up)
        g = syn.make_grid(size, size, grid, border=b)
        ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 0.0, angles, rot=rot)
        for _ in range(2):
            ctx.run(); ctx.fetch(want_ij=False)
        t0 = time.perf_counter()
        for _ in range(6):
            ctx.run(); ctx.fetch(want_ij=False)
        res[b] = round((time.perf_counter() - t0) / 6 / g['c1'].size * 1e9, 1)
print(json.dumps({'K': len(angles), 'ns_per_point': res}))
