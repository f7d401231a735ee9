#!/usr/bin/env python3
"""What a strong-scaling step costs per rank under different ways of sharding the grid: every rank's shard is timed on
the one GPU of this box (kernels + fetch, pair resident); the slowest shard is the predicted step time on `world` GPUs
(before the gather).  Usage: python3 tools/shard_sim.py [world=8] [feedback rounds=0]
With feedback rounds: the cuts of dist.shard_cuts_by_cost are then moved by dist.rebalance_cuts on the MEASURED kernel times
(what bench.py --gpus N does during its warm-up) and every round's per-rank times are reported."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sea_ice_drift_amd import _capi, dist, synthetic as syn
from sea_ice_drift_amd.pmlib import rotation_table

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
feedback = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size, grid, s = 10000, 200, 34
img1, img2 = syn.make_pair(size, size)
g = syn.make_grid(size, size, grid)
angles = list(range(-7, 8)); rot = rotation_table(angles, 0.0, s)
def snake_deal(border, world_size, rank):
    """The round-1/2 partition (kept here for the comparison only): points ordered by border, dealt 0..G-1, G-1..0, ..."""
    order = np.argsort(-np.asarray(border), kind='stable')
    pos = np.arange(order.size)
    k, rnd = pos % world_size, pos // world_size
    return np.sort(order[np.where(rnd % 2 == 0, k, world_size - 1 - k) == rank])


schemes = {'snake deal by border (rounds 1-2)': lambda r: snake_deal(g['border'], world, r),
           'contiguous runs of equal estimated time (dist.shard_indices_by_cost)': lambda r: dist.shard_indices_by_cost(g['border'], world, r)}
out = {}
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(img1, img2)
    def timed(idx, steps=30):
        """(run + fetch, run + sync) in ms: the first is what one GPU pays for a shard on its own (kernels + copy of its results
        to the host); the second is the kernels alone - what a rank of an N-GPU run contributes before the exchange step, where
        the per-rank copy does not exist (the kernels write the gathered block in place)."""
        ctx.set_points(g['c1'][idx], g['r1'][idx], g['c2fg'][idx], g['r2fg'][idx], g['border'][idx], s, 0.0, angles, rot=rot)
        for _ in range(3):
            ctx.run(); ctx.fetch(want_ij=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.run(); ctx.fetch(want_ij=False)
        t1 = time.perf_counter()
        for _ in range(steps):
            ctx.run(); ctx.sync()
        t2 = time.perf_counter()
        return (t1 - t0) / steps * 1e3, (t2 - t1) / steps * 1e3
    full, full_k = timed(np.arange(g['c1'].size))
    for name, fn in schemes.items():
        shards = [fn(r) for r in range(world)]
        assert sorted(np.concatenate(shards).tolist()) == list(range(g['c1'].size))
        both = [timed(ix) for ix in shards]
        ms, mk = [b[0] for b in both], [b[1] for b in both]
        out[name] = {'points_per_rank': [int(len(ix)) for ix in shards], 'ms_per_rank': [round(x, 3) for x in ms],
                     'slowest_ms': round(max(ms), 3), 'predicted_speedup_before_gather': round(full / max(ms), 2),
                     'kernels_only_ms_per_rank': [round(x, 3) for x in mk], 'kernels_only_slowest_ms': round(max(mk), 3),
                     'kernels_only_sum_ms': round(sum(mk), 3)}
    fb = None
    if feedback > 0:
        order, cuts, cost = dist.shard_cuts_by_cost(g['border'], world, s, len(angles))
        fb = {'rounds': [], 'what': 'kernels alone (run + sync), ms per rank; cuts moved by dist.rebalance_cuts on these times'}
        best = (np.inf, cuts)
        for it in range(feedback + 1):
            mk = np.array([timed(dist.indices_of_cut(order, cuts, r), steps=20)[1] for r in range(world)])
            fb['rounds'].append({'points_per_rank': np.diff(cuts).tolist(), 'kernels_only_ms_per_rank': [round(float(x), 3) for x in mk],
                                 'slowest_ms': round(float(mk.max()), 3), 'mean_ms': round(float(mk.mean()), 3)})
            if mk.max() < best[0]:
                best = (float(mk.max()), cuts)
            cuts = dist.rebalance_cuts(cost, cuts, mk)
        mk = np.array([timed(dist.indices_of_cut(order, best[1], r), steps=30)[1] for r in range(world)])   # the kept cuts, once more
        fb['kept'] = {'points_per_rank': np.diff(best[1]).tolist(), 'kernels_only_ms_per_rank': [round(float(x), 3) for x in mk],
                      'slowest_ms': round(float(mk.max()), 3), 'predicted_speedup_before_gather': round(full_k / float(mk.max()), 2)}
print(json.dumps({'world': world, 'measured_feedback': fb, 'full_step_ms_one_gpu': round(full, 3), 'full_step_kernels_only_ms': round(full_k, 3), 'schemes': out}, indent=1))
