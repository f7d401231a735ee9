#!/usr/bin/env python3
"""Run the benchmark workload repeatedly and report any run-to-run difference (race detector)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
img1, img2 = syn.make_pair(size, size)
g = syn.make_grid(size, size, size // 50)
angles = list(range(-7, 8)); rot = rotation_table(angles, 0.0, 34)
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(img1, img2)
    ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, angles, rot=rot)
    ref = None
    for it in range(runs):
        ctx.run(); out, ij = ctx.fetch()
        if ref is None: ref = (out.copy(), ij.copy()); continue
        bad = np.nonzero((ij != ref[1]).any(1) | (out != ref[0]).any(1))[0]
        if len(bad) or it == runs - 1: print('run %d: %d differing points' % (it, len(bad)), [(int(b), ij[b].tolist(), ref[1][b].tolist(), out[b, 3], ref[0][b, 3]) for b in bad[:4]])
