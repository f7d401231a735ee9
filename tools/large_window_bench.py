#!/usr/bin/env python3
"""Timings of the large-window pipeline (csrc/pm_large.hip): rotate_and_match on whole images, batches of points with search
borders beyond the LDS launch classes, template sides above 64 - one JSON line per case, with the C oracle on the host beside it
where that finishes in seconds.

    python tools/large_window_bench.py [--oracle]
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn

want_oracle = '--oracle' in sys.argv
if want_oracle:
    from oracle import c_oracle
    c_oracle.build()


def timed(fn, reps):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


with _capi.PMContext(0) as ctx:
    # rotate_and_match with the whole of image 2 as the window (the reference's own test: tests.py:336-337)
    for size, s, K in ((500, 50, 7), (2000, 50, 7), (2000, 35, 3), (4000, 50, 7), (10000, 34, 15)):
        img1, img2 = syn.make_pair(size, size, seed=11)
        angles = list(np.linspace(-3, 3, K))
        rot = my.rotation_table(angles, 0.0, s)
        ctx.upload_pair(img1, img2)
        c1, r1 = size * 0.6, size * 0.2
        ms = timed(lambda: ctx.rotate_and_match(c1, r1, s, 0.0, angles, rot=rot, window=(0, 0, size, size), want_ccm=False, want_template=False), 5)
        ms_ccm = timed(lambda: ctx.rotate_and_match(c1, r1, s, 0.0, angles, rot=rot, window=(0, 0, size, size)), 3)
        rec = dict(case='rotate_and_match whole image', size=size, img_size=s, angles=K, ms=round(ms, 3), ms_with_matrix_download=round(ms_ccm, 3),
                   placements=(size - s + 1) ** 2, gmacs=round(K * (size - s + 1) ** 2 * s * s * 1e-9, 2),
                   mfma_frac=round(2 * K * (size - s + 1) ** 2 * s * s / (ms * 1e-3) / 5e15, 4))
        if want_oracle and size <= 2000:
            t = time.perf_counter()
            c_oracle.rotate_and_match(img1, c1, r1, s, img2, 0.0, angles, rot)
            rec['c_oracle_1core_ms'] = round((time.perf_counter() - t) * 1e3, 1)
        print(json.dumps(rec), flush=True)
    # batches of points with large borders / large templates on the benchmark-sized pair
    size = 10000
    img1, img2 = syn.make_pair(size, size)
    ctx.upload_pair(img1, img2)
    g = syn.make_grid(size, size, 200)
    for border, s, K, n in ((112, 34, 15, 200), (160, 34, 15, 200), (250, 34, 15, 100), (250, 35, 3, 100), (50, 100, 3, 200), (20, 65, 15, 200)):
        sel = np.flatnonzero((g['c2fg'] > border + 300) & (g['c2fg'] < size - border - 300) & (g['r2fg'] > border + 300) & (g['r2fg'] < size - border - 300))[:n]
        angles = list(np.linspace(-7, 7, K)) if K > 3 else [-3, 0, 3]
        v = [g[k][sel] for k in ('c1', 'r1', 'c2fg', 'r2fg')]
        b = np.full(len(sel), float(border))
        ctx.set_points(*v, b, s, 0.0, angles)

        def step():
            ctx.run(); ctx.sync()
        ms = timed(step, 3)
        rec = dict(case='batch through the large-window pipeline', border=border, img_size=s, angles=K, points=len(sel), ms=round(ms, 2),
                   ms_per_point=round(ms / len(sel), 4), points_per_s=round(len(sel) / ms * 1e3, 1))
        if want_oracle:
            m = min(len(sel), 16)
            t = time.perf_counter()
            c_oracle.pm_batch(img1, img2, *[x[:m] for x in v], b[:m], s, 0.0, angles, rot=my.rotation_table(angles, 0.0, s), nthreads=os.cpu_count())
            rec['c_oracle_ms_per_point_all_cores'] = round((time.perf_counter() - t) * 1e3 / m, 2)
            rec['cores'] = os.cpu_count()
        print(json.dumps(rec), flush=True)
