#!/usr/bin/env python3
"""Which points differ between repetitions, and how (debug aid for the determinism soak)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SID_PM_NO_SAMP_TABLE'] = '1'
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table
from oracle import c_oracle
size = 4000
img1, img2 = syn.make_pair(size, size, seed=777)
ang = list(range(-7, 8)); rot = rotation_table(ang, 0.0, 34)
import sys as _s
BORDERS = [int(x) if x != 'mixed' else x for x in _s.argv[1:]] or [20, 28, 44, 'mixed']
for border in BORDERS:
    g = syn.make_grid(size, size, 80, border=border)
    with _capi.PMContext(0) as ctx:
        ctx.upload_pair(img1, img2)
        ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, ang, rot=rot)
        ctx.run(); ref, ref_ij = ctx.fetch()
        exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, ang, rot=rot, nthreads=16)
        bad_ref = np.nonzero(~((ref_ij == exp_ij).all(1) & (ref[:, :4] == exp[:, :4]).all(1)))[0]
        print('border', border, ': first run vs oracle: %d bad' % len(bad_ref), [(int(b), g['border'][b], ref_ij[b].tolist(), exp_ij[b].tolist(), ref[b, 3], exp[b, 3]) for b in bad_ref[:5]])
        nbad = 0
        for it in range(40):
            ctx.run(); out, ij = ctx.fetch()
            same = (ij == exp_ij).all(1) & (out[:, :4] == exp[:, :4]).all(1)
            bad = np.nonzero(~same)[0]
            nbad += len(bad)
            if len(bad) and it < 6:
                print('  run %d: %d bad' % (it, len(bad)), [(int(b), g['border'][b], ij[b].tolist(), exp_ij[b].tolist(), float(out[b, 3]), float(exp[b, 3])) for b in bad[:4]])
        print('border', border, 'total bad over 40 runs:', nbad)
