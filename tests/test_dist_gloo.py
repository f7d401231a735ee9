"""CPU test of the multi-GPU path's host logic with 2 gloo processes: sharding by border and
the gather that undoes it (on the GPU box the same code runs over RCCL with device tensors)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sea_ice_drift_amd.dist import (PackedGatherer, ResultGatherer, point_cost, shard_indices, shard_indices_by_cost,
                                    shard_size)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(12)
    border = np.clip(np.floor(rng.rayleigh(16.0, n_total)), 20, 50)
    truth = rng.standard_normal((n_total, 5))
    truth_ij = rng.integers(0, 100, (n_total, 3)).astype(np.int32)
    idx = shard_indices(border, world, rank)
    m = shard_size(n_total, world)
    out_local = torch.full((m, 5), float('nan'), dtype=torch.float64)
    ij_local = torch.full((m, 3), -1, dtype=torch.int32)
    out_local[:len(idx)] = torch.from_numpy(truth[idx])          # stands in for the kernel's output
    ij_local[:len(idx)] = torch.from_numpy(truth_ij[idx])
    g = ResultGatherer(n_total, idx, torch.device('cpu'))
    for _ in range(2):                                           # reusable across steps
        out, ij = g.gather(out_local, ij_local)
    # the one-collective form bench.py uses: the "kernel" writes into the packed block in place
    pg = PackedGatherer(n_total, idx, torch.device('cpu'))
    o_v, j_v = pg.local_views()
    o_v.copy_(torch.from_numpy(truth[idx]))
    j_v.copy_(torch.from_numpy(truth_ij[idx]))
    for _ in range(2):
        pg.gather_to_host()
    if rank == 0:
        p_out, p_ij = pg.host_results()
        assert np.array_equal(p_out, truth) and np.array_equal(p_ij, truth_ij)
    # shards of equal estimated cost (what bench.py uses): unequal lengths, block rows = the largest shard
    idx_c = shard_indices_by_cost(border, world, rank)
    pc = PackedGatherer(n_total, idx_c, torch.device('cpu'))
    o_v, j_v = pc.local_views()
    o_v.copy_(torch.from_numpy(truth[idx_c]))
    j_v.copy_(torch.from_numpy(truth_ij[idx_c]))
    pc.gather_to_host()
    if rank == 0:
        c_out, c_ij = pc.host_results()
        assert np.array_equal(c_out, truth) and np.array_equal(c_ij, truth_ij)
    if rank == 0:
        q.put((np.array_equal(out.numpy(), truth), np.array_equal(ij.numpy(), truth_ij),
               float(border[idx].sum()), len(idx)))
    else:
        assert out is None and ij is None
        q.put((None, None, float(border[idx].sum()), len(idx)))
    dist.barrier()
    dist.destroy_process_group()


def test_shards_partition_and_balance():
    rng = np.random.default_rng(1)
    border = np.clip(np.floor(rng.rayleigh(16.0, 1001)), 20, 50)
    for world in (1, 2, 4, 8):
        parts = [shard_indices(border, world, r) for r in range(world)]
        allidx = np.sort(np.concatenate(parts))
        np.testing.assert_array_equal(allidx, np.arange(1001))               # a partition
        assert max(len(p) for p in parts) <= shard_size(1001, world)
        work = [((2 * border[p] + 2) ** 2).sum() for p in parts]
        assert max(work) / min(work) < 1.02                                   # balanced by window size


def test_cost_shards_partition_and_balance():
    rng = np.random.default_rng(2)
    border = np.clip(np.floor(rng.rayleigh(16.0, 40000)), 20, 50)
    for world in (1, 2, 4, 8):
        parts = [shard_indices_by_cost(border, world, r) for r in range(world)]
        np.testing.assert_array_equal(np.sort(np.concatenate(parts)), np.arange(border.size))   # a partition
        # equal estimated TIME: cost of the points + the tails of the shard's launches (dist._shard_times); the bare costs
        # may differ by the tails (the one-per-CU shard gets less work than a three-per-CU one)
        order = np.argsort(-border, kind='stable')
        from sea_ice_drift_amd import _capi
        from sea_ice_drift_amd.dist import _shard_times
        cuts = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
        for r, p in enumerate(parts):                                         # shard r = the r-th run of the border order
            np.testing.assert_array_equal(np.sort(order[cuts[r]:cuts[r + 1]]), p)
        t = _shard_times(point_cost(border[order]), _capi.estimate_residency(border[order]), cuts)
        assert t.max() / t.min() < 1.02
        cost = [point_cost(border[p]).sum() for p in parts]
        assert max(cost) / min(cost) < 1.25                                   # (at 8 shards a two-launch shard carries two tails)
        if world > 1:
            assert cost[0] < cost[-1]                                         # (largest borders: fewest slots, longest tail)
        if world == 8:                                          # a rank holds neighbouring borders, not a bit of each
            assert max(len(np.unique(border[p])) for p in parts) <= 16
    assert shard_indices_by_cost(np.zeros(0), 4, 1).size == 0
    np.testing.assert_array_equal(np.sort(np.concatenate([shard_indices_by_cost(np.full(5, 70.0), 2, r) for r in range(2)])),
                                  np.arange(5))


def test_two_rank_gather_roundtrip():
    world, n_total = 2, 1001
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in res if r[0] is not None]
    assert len(root) == 1 and root[0][0] and root[0][1]
    assert sum(r[3] for r in res) == n_total
