#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: a batch of 10000x10000 pairs streamed back to back, the upload of pair k+1
(copy stream, pinned host memory) overlapping the PM kernels of pair k.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_ice_drift_amd import _capi, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size, grid = 10000, 200
distinct = 2                                   # distinct synthetic pairs cycled through the batch (host RAM)
t0 = time.perf_counter()
host = []
for k in range(distinct):
    a, b = syn.make_pair(size, size, seed=20200123 + 7 * k)
    host.append((torch.from_numpy(a).pin_memory().numpy(), torch.from_numpy(b).pin_memory().numpy()))
t_gen = time.perf_counter() - t0
g = syn.make_grid(size, size, grid)
angles = list(range(-7, 8)); rot = rotation_table(angles, 0.0, 34)
n = len(g['c1'])
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(*host[0], slot=0)
    ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, angles, rot=rot)
    ctx.run(); ref0 = ctx.fetch()
    # serial: upload, run, fetch - nothing overlaps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(npairs):
        ctx.upload_pair(*host[k % distinct], slot=0); ctx.select_pair(0); ctx.run(); out = ctx.fetch()
    t_serial = time.perf_counter() - t0
    # streamed: upload of the next pair overlaps this pair's kernels
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.upload_pair(*host[0], slot=0)
    outs = []
    for k in range(npairs):
        ctx.select_pair(k % 2)
        if k + 1 < npairs: ctx.upload_pair(*host[(k + 1) % distinct], slot=(k + 1) % 2, select=False)
        ctx.run(); outs.append(ctx.fetch())
    t_stream = time.perf_counter() - t0
    ok = all(np.array_equal(outs[k][1], outs[k % distinct][1]) for k in range(npairs)) and np.array_equal(outs[0][1], ref0[1])
print(json.dumps({'metric': 'pairs streamed back to back (10000x10000 px, 200x200 grid, K=15, mixed border), 1 GPU',
                  'pairs': npairs, 'serial_ms_per_pair': t_serial / npairs * 1e3, 'streamed_ms_per_pair': t_stream / npairs * 1e3,
                  'streamed_grid_points_per_s': n * npairs / t_stream, 'upload_mb_per_pair': 2 * size * size / 1e6,
                  'results_consistent': bool(ok), 'host_buffers': 'pinned (torch pin_memory)', 'setup_s': t_gen}))
