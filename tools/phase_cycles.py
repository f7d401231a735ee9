#!/usr/bin/env python3
"""Shader-clock cycles per kernel phase for single points (one workgroup on an idle GPU)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table

img1, img2 = syn.make_pair(1000, 1000)
g = syn.make_grid(1000, 1000, 10, border=20)
_A = int(os.environ.get('SID_PHASE_ANGLES', '7'))
angles = list(range(-_A, _A + 1))
rot = rotation_table(angles, 0.0, 34)
names = ['P0a window', 'P1 sums', 'P0b templates', 'P2 sweep', 'P3 argmax', 'P4 winner', 'P5 hessian']
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(img1, img2)
    for b in [int(x) for x in os.environ.get("SID_PHASE_BORDERS", "20,35,50").split(",")]:
        acc = []
        for i in (11, 45, 77, 78):
            d = ctx.debug_point(g['c1'][i], g['r1'][i], g['c2fg'][i], g['r2fg'][i], float(b), 34, 0.0, angles, rot=rot)
            acc.append(np.diff(d['cycles'][:8]))
            c = d['cycles']
            cy = c
            fine = [c[8] - c[2], c[15] - c[8], c[9] - c[15], c[3] - c[9], c[11] - c[10], c[12] - c[11], c[13] - c[5], c[6] - c[13], c[14] - c[6], c[7] - c[14]]
        acc = np.median(np.array(acc), axis=0)
        print('border %d: total %d cycles' % (b, acc.sum()))
        for n, c in zip(names, acc):
            print('   %-14s %8d  %5.1f %%' % (n, c, 100.0 * c / acc.sum()))
        f2 = [cy[16] - cy[1], cy[2] - cy[16], cy[17] - cy[13], cy[18] - cy[17], cy[6] - cy[18], cy[19] - cy[6], cy[14] - cy[19], cy[20] - cy[14], cy[21] - cy[20], cy[22] - cy[21], cy[23] - cy[22], cy[24] - cy[23], cy[7] - cy[24]]
        print('   fine2: wp %d, sums %d | winner item0 mfma %d, norm %d, rest %d | hes compute %d, barrier+dump %d, reduce %d, hist %d, scan %d, compact %d, rank %d, tail %d' % tuple(f2))
        print('   fine: patch+zero %d, sampling(tid0) %d, ones+barrier %d, sums %d | item0 sweep %d, item0 epilogue %d | P4 staging %d, P4 mfma+norm %d | P5 hessian %d, P5 median/std %d' % tuple(fine))
