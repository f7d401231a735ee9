#!/usr/bin/env python3
"""Detector on the 10000x10000 benchmark image: levels on the device from end to end (default) against the host-side selection
and sort (SID_ORB_HOST_SELECT=1); wall clock per image, best of 5, and equality of the results."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import orb, synthetic as syn

size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
img = syn.make_pair(size, size, speckle=0.03)[0]
res, outs = {}, {}
for name, env in (('device', None), ('host_select', '1'), ('device_again', None)):
    os.environ.pop('SID_ORB_HOST_SELECT', None)
    if env:
        os.environ['SID_ORB_HOST_SELECT'] = env
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); out = orb.detect_and_compute(img, full=True); best = min(best, time.perf_counter() - t0)
    res[name + '_ms'] = round(best * 1e3, 3); outs[name] = out
os.environ.pop('SID_ORB_HOST_SELECT', None)
res['keypoints'] = int(len(outs['device'][0]))
res['equal'] = bool(all(np.array_equal(a, b) for a, b in zip(outs['device'], outs['host_select'])))
print(json.dumps(res))
