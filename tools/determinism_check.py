#!/usr/bin/env python3
"""Run a workload repeatedly and report any run-to-run difference (race detector).

    python tools/determinism_check.py [size=4000] [runs=8] [--angles 7] [--img-size 34] [--border mixed|N]
                                      [--no-table]      (on-the-fly sampler instead of the offset table)
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('size', nargs='?', type=int, default=4000)
ap.add_argument('runs', nargs='?', type=int, default=8)
ap.add_argument('--angles', type=int, default=7)
ap.add_argument('--img-size', type=int, default=34)
ap.add_argument('--border', default='mixed')
ap.add_argument('--no-table', action='store_true')
args = ap.parse_args()
if args.no_table:
    os.environ['SID_PM_NO_SAMP_TABLE'] = '1'
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table
size, runs, s = args.size, args.runs, args.img_size
img1, img2 = syn.make_pair(size, size)
g = syn.make_grid(size, size, size // 50, border=args.border if args.border == 'mixed' else int(args.border))
angles = list(range(-args.angles, args.angles + 1)); rot = rotation_table(angles, 0.0, s)
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(img1, img2)
    ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 0.0, angles, rot=rot)
    ref = None
    nbad = 0
    for it in range(runs):
        ctx.run(); out, ij = ctx.fetch()
        if ref is None: ref = (out.copy(), ij.copy()); continue
        same = (ij == ref[1]).all(1) & ((out == ref[0]) | (np.isnan(out) & np.isnan(ref[0]))).all(1)
        bad = np.nonzero(~same)[0]
        nbad += len(bad)
        if len(bad):
            cols = [int(((out[bad, c] != ref[0][bad, c]) & ~(np.isnan(out[bad, c]) & np.isnan(ref[0][bad, c]))).sum()) for c in range(5)]
            print('run %d: %d differing points, indices %d..%d (%s), differing columns c2/r2/a/r/h: %s, ij rows differing: %d, borders %s'
                  % (it, len(bad), bad[0], bad[-1], 'consecutive' if bad[-1] - bad[0] + 1 == len(bad) else 'scattered', cols,
                     int((ij[bad] != ref[1][bad]).any(1).sum()), sorted(set(g['border'][bad].tolist()))[:6]))
            print('   first:', [(int(b), ij[b].tolist(), ref[1][b].tolist(), out[b].tolist(), ref[0][b].tolist()) for b in bad[:2]])
            # is the difference still there when the same buffers are fetched again / the step is repeated?
            out2, ij2 = ctx.fetch()
            print('   refetch of the same step: %d of them still differ' % int((~((ij2[bad] == ref[1][bad]).all(1) & ((out2[bad] == ref[0][bad]) | (np.isnan(out2[bad]) & np.isnan(ref[0][bad]))).all(1))).sum()))
    print('%d runs of %d points (size %d, s=%d, K=%d, border %s%s): %d differing point results'
          % (runs, len(g['c1']), size, s, len(angles), args.border, ', on-the-fly sampler' if args.no_table else '', nbad))
