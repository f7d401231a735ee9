#!/usr/bin/env python3
"""Wall clock of the pieces of pm_dispatch on the benchmark pair (upload excluded / included)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table
size, grid, s = 10000, 200, 34
img1, img2 = syn.make_pair(size, size)
g = syn.make_grid(size, size, grid)
angles = list(range(-7, 8))
res = {}
def best(fn, reps=5):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t = min(t, time.perf_counter() - t0)
    return t * 1e3
with _capi.PMContext(0) as ctx:
    res['upload_pair_pageable_ms'] = best(lambda: (ctx.upload_pair(img1, img2), ctx.sync()), 3)
    res['rotation_table_ms'] = best(lambda: rotation_table(angles, 0.0, s))
    rot = rotation_table(angles, 0.0, s)
    sp = lambda: ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 0.0, angles, rot=rot)
    res['set_points_ms'] = best(sp)
    res['run_and_sync_ms'] = best(lambda: (ctx.run(), ctx.sync()))
    ctx.run()
    res['fetch_ms'] = best(lambda: ctx.fetch(want_ij=False))
    res['set_run_fetch_ms'] = best(lambda: (sp(), ctx.run(), ctx.fetch(want_ij=False)))
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
