#!/usr/bin/env python3
"""Where get_drift_FT spends its time on the benchmark pair: detector stages (SID_ORB_VERBOSE on stderr), matcher, host
filters.  Usage: python3 tools/ft_profile.py [size]"""
import contextlib, io, json, os, sys, time
os.environ.setdefault('SID_ORB_VERBOSE', '1') if os.environ.get('SID_ORB_VERBOSE', None) != '' else os.environ.pop('SID_ORB_VERBOSE')
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import ftlib, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat

size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
img1, img2 = syn.make_pair(size, size, speckle=0.03)
scale = 4e-4
n1 = ArrayNansat(img1, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
n2 = ArrayNansat(img2, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
out = {}
for rep in range(2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        kp1, d1 = ftlib.find_key_points(img1, nFeatures=100000)
    t1 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        kp2, d2 = ftlib.find_key_points(img2, nFeatures=100000)
    t2 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        x1, y1, x2, y2 = ftlib.get_match_coords(kp1, d1, kp2, d2)
    t3 = time.perf_counter()
    out = {'detect1_s': t1 - t0, 'detect2_s': t2 - t1, 'match_and_ratio_test_s': t3 - t2, 'keypoints': [len(kp1), len(kp2)], 'matches': int(len(x1))}
print(json.dumps(out))
