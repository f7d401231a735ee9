"""CPU tests of the multi-GPU path's host logic with 2 and 8 gloo processes: shards of equal estimated time (unequal
lengths) and the one-collective exchange step that undoes them (on the GPU box the same code runs over RCCL with device
tensors and un-permutes with one kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sea_ice_drift_amd.dist import (PackedGatherer, indices_of_cut, per_rank_breakdown, point_cost, rebalance_cuts,
                                    shard_cuts_by_cost, shard_indices_by_cost)


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
    # shards of equal estimated time (what bench.py uses): unequal lengths, block rows = the largest shard (made even)
    idx = shard_indices_by_cost(border, world, rank)
    lens = [len(shard_indices_by_cost(border, world, r)) for r in range(world)]
    pg = PackedGatherer(n_total, idx, torch.device('cpu'))
    assert pg.m == max(lens) + (max(lens) & 1) and pg.n_local == lens[rank]
    o_v, j_v = pg.local_views()
    for step in range(2):                                        # reusable across steps; the second step after a poisoning
        o_v.copy_(torch.from_numpy(truth[idx] + step))           # stands in for the kernels' output, written in place
        j_v.copy_(torch.from_numpy(truth_ij[idx] + step))
        pg.gather_to_host()
        if rank == 0:
            p_out, p_ij = pg.host_results()
            assert np.array_equal(p_out, truth + step) and np.array_equal(p_ij, truth_ij + step)
        pg.poison()
        if rank == 0:
            assert np.isnan(pg.host_results()[0]).all() and (pg.host_results()[1] == -1).all()
    # every rank's step numbers on every rank (bench.py's per-rank breakdown)
    table = per_rank_breakdown([10.0 + rank, float(len(idx))], torch.device('cpu'))
    assert table.shape == (world, 2) and list(table[:, 0]) == [10.0 + r for r in range(world)] and list(table[:, 1]) == lens
    q.put((rank, len(idx), float(point_cost(border[idx]).sum())))
    dist.barrier()
    dist.destroy_process_group()


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
        t = _shard_times(border[order], cuts)
        assert t.max() / t.min() < 1.02
        cost = [point_cost(border[p]).sum() for p in parts]
        # (at 8 shards the rank with the largest borders runs two launches of few slots - one workgroup per CU, two per CU -
        # whose tails are a third of its step: it gets that much less work than a rank of border-20 points)
        assert max(cost) / min(cost) < 1.45
        if world > 1:
            assert cost[0] < cost[-1]                                         # (largest borders: fewest slots, longest tail)
        if world == 8:                                          # a rank holds neighbouring borders, not a bit of each
            assert max(len(np.unique(border[p])) for p in parts) <= 16
    assert shard_indices_by_cost(np.zeros(0), 4, 1).size == 0
    np.testing.assert_array_equal(np.sort(np.concatenate([shard_indices_by_cost(np.full(5, 70.0), 2, r) for r in range(2)])),
                                  np.arange(5))


def test_measured_feedback_moves_the_cuts_towards_equal_times():
    """``rebalance_cuts``: a device whose launches cost something else than the estimate says (here: every residency class off
    by its own factor, plus a fixed tail per shard) - a few rounds of measure / rebalance bring the slowest shard to within 1 %
    of the mean; the cuts stay a partition at every round; degenerate inputs leave them alone."""
    from sea_ice_drift_amd import _capi, synthetic as syn
    g = syn.make_grid(10000, 10000, 200)
    world = 8
    order, cuts, cost = shard_cuts_by_cost(g['border'], world)
    assert cuts[0] == 0 and cuts[-1] == order.size and (np.diff(cuts) > 0).all()
    for r in range(world):
        np.testing.assert_array_equal(indices_of_cut(order, cuts, r), shard_indices_by_cost(g['border'], world, r))
    cls = _capi.estimate_residency(g['border'][order], 34, 15, 1) & 15
    factor = np.where(cls == 1, 0.9, np.where(cls == 2, 1.08, np.where(cls == 3, 1.03, 1.0)))

    def device(cuts):
        return np.array([(cost[a:b] * factor[a:b]).sum() + 4.0e4 for a, b in zip(cuts[:-1], cuts[1:])])
    t0 = device(cuts)
    assert t0.max() / t0.mean() > 1.02
    c = cuts
    for _ in range(6):
        c = rebalance_cuts(cost, c, device(c))
        assert c[0] == 0 and c[-1] == order.size and (np.diff(c) >= 0).all()
        assert sorted(np.concatenate([indices_of_cut(order, c, r) for r in range(world)]).tolist()) == list(range(order.size))
    t = device(c)
    assert t.max() / t.mean() < 1.01 and t.max() < t0.max()
    np.testing.assert_array_equal(rebalance_cuts(cost, cuts, np.full(world, np.nan)), cuts)      # no usable measurement
    np.testing.assert_array_equal(rebalance_cuts(cost, cuts, np.zeros(world)), cuts)
    np.testing.assert_array_equal(rebalance_cuts(cost, cuts, t0 * 0 + 1.0, damping=0.0), cuts)  # no step


def _roundtrip(world, n_total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert sum(r[1] for r in res) == n_total
    return res


def test_two_rank_gather_roundtrip():
    _roundtrip(2, 1001)


def test_eight_rank_gather_roundtrip_with_unequal_shards():
    """BASELINE config 3's rank count on CPU: eight gloo ranks, shards of equal estimated time - the rank with the largest
    borders holds far fewer points than the one with the smallest -, one gather of blocks of the largest shard's rows."""
    res = _roundtrip(8, 6007)
    lens = [r[1] for r in sorted(res)]
    assert max(lens) > 1.5 * min(lens)                            # unequal on purpose


def _feedback_worker(rank, world, port, fail_rank, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sea_ice_drift_amd.dist import rebalance_with_feedback, shard_cuts_by_cost
        border = np.repeat(np.arange(20.0, 51.0), 40)
        order, cuts, cost = shard_cuts_by_cost(border, world)
        state = {'cuts': cuts, 'calls': 0}

        def measure():
            state['calls'] += 1
            if rank == fail_rank and state['calls'] == 2:
                raise RuntimeError('a refused point on rank %d' % rank)
            a, b = state['cuts'][rank], state['cuts'][rank + 1]
            return float(cost[a:b].sum()) * (1.3 if rank == 0 else 1.0) + 5e4   # rank 0's device is slower than the estimate says

        def reshard(c):
            state['cuts'] = c
        try:
            res = rebalance_with_feedback(measure, reshard, cost, cuts, 3, 'cpu')
            q.put((rank, 'ok', [h['kernel_ms'] for h in res['history']], np.diff(res['cuts']).tolist()))
        except RuntimeError as e:
            q.put((rank, 'raised', str(e), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fail_rank', [-1, 1])
def test_measured_feedback_is_collective_safe(fail_rank):
    """``dist.rebalance_with_feedback`` on two gloo ranks: without a failure the slowest rank gets faster and no shard is empty;
    when ONE rank's measurement raises, EVERY rank raises after the round's all_gather - none is left blocking in it."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_feedback_worker, args=(r, world, port, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if fail_rank < 0:
        assert all(r[1] == 'ok' for r in res)
        hist = res[0][2]
        assert hist == res[1][2]                                   # the same numbers on every rank
        assert max(hist[-1]) < max(hist[0]) and min(res[0][3]) >= 1
    else:
        assert all(r[1] == 'raised' for r in res), res
        assert 'rank(s) [1]' in res[0][2] and 'rank(s) [1]' in res[1][2]


def test_rebalance_cuts_keep_every_shard_non_empty():
    cost = np.ones(10)
    cuts = np.array([0, 3, 6, 10])
    new = rebalance_cuts(cost, cuts, np.array([1e-9, 1.0, 1.0]), damping=1.0)    # rank 0 measured ~nothing: it would be given everything
    assert (np.diff(new) >= 1).all() and new[0] == 0 and new[-1] == 10
