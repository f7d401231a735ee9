"""Sharding of grid points over the GPUs of one node and the exchange step that brings their results together.

The reference's only parallelism is data parallelism over independent grid points
(multiprocessing.Pool.map at pmlib.py:442-444: index scatter, pickled 5-tuples back).
Here: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for the tests).  The points, ordered by search border, are cut into one
contiguous run of equal estimated TIME per rank (``shard_indices_by_cost``: a rank holds one or
two neighbouring residency classes, i.e. one or two long launches instead of an eighth of every
class), and ONE collective ends a step - a gather of every rank's packed result block
``[m x 5 float64 | m x 3 int32]`` to rank 0 (``PackedGatherer``), where one kernel
(``sid_pm_unpermute``) puts the rows back into the original point order, writing straight into pinned host
memory.  The payload is tiny (40 000 points -> 2.1 MB), so the gather is latency-bound, not xGMI-link-bound.
"""
import os

import numpy as np


def point_cost(border, img_size=34, n_angles=15, flags=1):
    """Estimated cost of a grid point (nanoseconds on one MI355X; only the ratios matter here).  Computed by the library
    (include/sid_pm.h ``sid_pm_estimate_cost``, host arithmetic) from what the kernel executes for a point of that search
    border - matrix instructions of the sweep and of the winner's NCC matrix, placements - and from the residency class of
    its LDS footprint, for the template side and angle count of the run; it reproduces the staircase that
    tools/border_cost.py measures (placement tiles per output row, work items per wavefront, workgroups per CU) within 4 %."""
    from . import _capi
    return _capi.estimate_cost(np.asarray(border, dtype=np.float64), img_size, n_angles, flags)


def _shard_times(border_sorted, cuts, img_size=34, n_angles=15, flags=1):
    """Estimated kernel time of every shard [cuts[r], cuts[r+1]) of the border-ordered points: the LIBRARY's estimate of one run
    over the shard's points (include/sid_pm.h ``sid_pm_estimate_run_time``: point costs per launch class, launch tails, the
    launcher's own rule for side-by-side launches, large-window points) - the rule lives in one place, next to the launcher."""
    from . import _capi
    t = np.zeros(len(cuts) - 1)
    for r in range(len(cuts) - 1):
        a, b = cuts[r], cuts[r + 1]
        if b > a:
            t[r] = _capi.estimate_run_time(border_sorted[a:b], img_size, n_angles, flags)
    return t


def shard_cuts_by_cost(border, world_size, img_size=34, n_angles=15, flags=1):
    """(order, cuts, cost): the points ordered by border (largest first, stable), the ``world_size + 1`` positions that cut
    that order into contiguous runs of equal estimated TIME, and the estimated cost of every point in that order.  A rank then
    holds one or two neighbouring border classes instead of an eighth of every class, i.e. one or two launches that are eight
    times longer: at 5 000 points per rank the tails of three short launches cost a quarter of the step.
    The time of a run is the cost of its points (``point_cost``) plus the tails of its launches (``_shard_times``): cut by cost
    alone, the rank with the largest borders - one workgroup per CU, 7 rounds of 256 points - came out 6 % slower than the
    others on the GPU (tools/shard_sim.py).

    Every rank's kernels must finish before the gather can complete, so the step time is the slowest rank's kernel time
    plus the exchange step; shortening rank 0's shard would not hide the exchange (it starts when the LAST rank is done).
    Pure function of its arguments: every rank computes the same cuts."""
    border = np.asarray(border)
    order = np.argsort(-border, kind='stable')
    n = order.size
    if n == 0 or world_size <= 1:
        return order, np.array([0] + [n] * max(world_size, 1), dtype=np.int64), np.zeros(n)
    cost = point_cost(border[order], img_size, n_angles, flags)          # (flags of the run: sid_pm.h SID_PM_HES_NORM = 1 ...)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    share = np.full(world_size, cum[-1] / world_size)               # cost each shard is to hold
    best_cuts, best = None, np.inf
    for _ in range(24):                                             # move cost from the slow shards to the fast ones
        cuts = _cuts_for_shares(cost, cum, share)
        t = _shard_times(border[order], cuts, img_size, n_angles, flags)
        if t.max() < best:
            best, best_cuts = t.max(), cuts
        share = np.maximum(share + 0.7 * (t.mean() - t), 0.0)
        share *= cum[-1] / share.sum()
    return order, best_cuts, cost


def _cuts_for_shares(cost, cum, share):
    """Cut positions for the given cost per shard: point k goes to the shard whose cost interval holds the middle of its own."""
    edges = np.cumsum(share)[:-1]
    mid = cum[:-1] + 0.5 * cost
    c = np.searchsorted(mid, edges, side='left')
    return np.concatenate([[0], c, [cost.size]]).astype(np.int64)


def shard_indices_by_cost(border, world_size, rank, img_size=34, n_angles=15, flags=1):
    """Indices (ascending) owned by `rank` under ``shard_cuts_by_cost``."""
    order, cuts, _ = shard_cuts_by_cost(border, world_size, img_size, n_angles, flags)
    return indices_of_cut(order, cuts, rank)


def indices_of_cut(order, cuts, rank):
    """Indices (ascending) of shard `rank` of the border-ordered points cut at ``cuts``."""
    if rank + 1 >= len(cuts):
        return np.zeros(0, dtype=order.dtype)
    return np.sort(order[cuts[rank]:cuts[rank + 1]])


def rebalance_cuts(cost, cuts, measured, damping=0.6):
    """New cut positions from the kernel times the ranks MEASURED with the current ones (``measured[r]``, any unit; the same
    array on every rank, e.g. from ``per_rank_breakdown``).  The estimate prices a launch as cost + tail; what a device does with
    a launch whose last round of workgroups is nearly empty, or with two short launches side by side, it can only fit (the
    slowest of eight simulated shards stays 4 % above the mean).  The measured time of a shard is spread
    over its points in proportion to their estimated costs, the border order is cut into runs of equal spread time, and the
    cuts move ``damping`` of the way there (a launch's tail does not shrink with its length: the full step overshoots).
    Callers iterate - measure, rebalance, measure - and keep the cuts with the smallest slowest time.  Pure function."""
    cuts = np.asarray(cuts, dtype=np.int64)
    measured = np.asarray(measured, dtype=np.float64)
    world = cuts.size - 1
    if world <= 1 or cost.size == 0 or not np.all(np.isfinite(measured)) or measured.min() <= 0.0:
        return cuts.copy()
    w = np.array(cost, dtype=np.float64)
    for r in range(world):
        a, b = cuts[r], cuts[r + 1]
        if b > a:
            w[a:b] *= measured[r] / max(w[a:b].sum(), 1e-300)       # the shard's measured time, spread like its estimated cost
    cum = np.concatenate([[0.0], np.cumsum(w)])
    target = _cuts_for_shares(w, cum, np.full(world, cum[-1] / world))
    new = np.rint(cuts + damping * (target - cuts)).astype(np.int64)
    new[0], new[-1] = 0, cost.size
    new = np.maximum.accumulate(new)
    if cost.size >= world:                                          # every rank keeps at least one point (an empty shard times as
        new = np.maximum(new, np.arange(world + 1))                 # ~0 ms and would drag the next round's cuts towards it)
        new = np.minimum(new, cost.size - (world - np.arange(world + 1)))
    return new


class PackedGatherer(object):
    """One collective per step: every rank's results live in ONE padded byte block
    [m x 5 float64 | m x 3 int32] that the kernels write in place (``local_views``), so the
    exchange step of the path is a single gather of that block to rank `dst` (RCCL over xGMI
    with device tensors), followed there by ONE kernel (``sid_pm_unpermute``) that undoes the sharding and writes the
    rows, in original point order, straight into pinned host memory - gather, kernel, stream synchronise.  With a
    ``gloo`` group (CPU tests, several ranks sharing one device in a dry run) the block is staged through the host and the
    rows are put back with ``index_select``.
    """

    ROW = 5 * 8 + 3 * 4

    def __init__(self, n_total, idx_local, device, group=None, dst=0, force_collective=False, timing=False):
        """``force_collective``: run every collective (the all_reduce and the two gathers) even in a group of ONE rank - the
        RCCL code path of an N-GPU run then executes on a one-GPU box (a world-size-1 ``nccl`` group), where it can be
        tested.  ``timing``: HIP events around the stages of the exchange step on EVERY rank (``timings()``)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.dst = torch, dist, group, dst
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.is_dst = self.rank == dst
        self.n_total, self.device = int(n_total), torch.device(device)
        self.n_local = len(idx_local)
        self.collective = self.distributed and (self.world > 1 or bool(force_collective))
        self.host_staged = self.distributed and dist.get_backend(group) == 'gloo' and self.device.type != 'cpu'
        self.timing = bool(timing) and self.device.type == 'cuda'
        self._acc, self._n_timed, self._pending = [0.0, 0.0, 0.0], 0, []
        # rows of the padded block = the largest shard (the shards of shard_indices_by_cost differ in length), made even so
        # that every block of the stacked buffer keeps its doubles 8-byte aligned
        if self.collective:
            t = torch.tensor([self.n_local], dtype=torch.int64,
                             device='cpu' if dist.get_backend(group) == 'gloo' else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            m = int(t.item())
        else:
            m = self.n_local
        m = max(m, 1)
        self.m = m = m + (m & 1)
        self.block = torch.zeros(m * self.ROW, dtype=torch.uint8, device=self.device)
        self.out_local = self.block[:m * 40].view(torch.float64).view(m, 5)
        self.ij_local = self.block[m * 40:].view(torch.int32).view(m, 3)
        self.out_local.fill_(float('nan'))
        self.ij_local.fill_(-1)
        # one-time index exchange: which original point every row of every rank's block holds
        idx_pad = torch.full((m,), -1, dtype=torch.int64)
        idx_pad[:self.n_local] = torch.as_tensor(np.asarray(idx_local), dtype=torch.int64)
        if not self.collective:
            all_idx = idx_pad[None]
        else:
            gl = [torch.empty_like(idx_pad) for _ in range(self.world)] if self.is_dst else None
            if dist.get_backend(group) == 'gloo':
                dist.gather(idx_pad, gl, dst=dst, group=group)
            else:
                dev_pad = idx_pad.to(self.device)
                dgl = [torch.empty_like(dev_pad) for _ in range(self.world)] if self.is_dst else None
                dist.gather(dev_pad, dgl, dst=dst, group=group)
                gl = [t.cpu() for t in dgl] if self.is_dst else None
            all_idx = torch.stack(gl) if self.is_dst else None
        self.identity = self.zero_copy = self.device_unpermute = False
        if self.is_dst:
            flat = all_idx.reshape(-1)
            rows = torch.nonzero(flat >= 0).reshape(-1)
            perm = torch.full((self.n_total,), -1, dtype=torch.int64)
            perm[flat[rows]] = rows                               # original index -> row of the stacked blocks
            if (perm < 0).any():
                raise ValueError('the shards do not cover all points')
            work_dev = torch.device('cpu') if self.host_staged else self.device
            self.perm = perm.to(work_dev)
            self.stack = torch.empty((self.world, m * self.ROW), dtype=torch.uint8, device=work_dev)
            n = self.n_total
            pin = self.device.type == 'cuda'
            # the results on the host: ONE buffer [n x 5 float64 | n x 3 int32], pinned (device-visible on ROCm: hipHostMalloc)
            self.host = torch.empty(n * self.ROW, dtype=torch.uint8, pin_memory=pin)
            self.host_out = self.host[:n * 40].view(torch.float64).view(n, 5)
            self.host_ij = self.host[n * 40:].view(torch.int32).view(n, 3)
            # one rank, no forced collective, points already in their own order: the block IS the result - no un-permutation
            self.identity = (not self.collective) and m == n and bool((self.perm == torch.arange(n, device=self.perm.device)).all())
            # zero copy: then the kernels write their 52 B per point straight into the pinned host buffer - 2 MB of posted
            # PCIe writes spread over the step instead of a 50 us copy after it.  SID_PM_NO_ZERO_COPY=1: copy as before (A/B runs).
            self.zero_copy = self.identity and pin and os.environ.get('SID_PM_NO_ZERO_COPY') is None
            if self.zero_copy:
                self.host_out.fill_(float('nan'))
                self.host_ij.fill_(-1)
                self.out_local, self.ij_local = self.host_out, self.host_ij
            # N ranks on their own GPUs: one kernel reads the gathered blocks through the permutation and writes the pinned
            # host buffer (sid_pm_unpermute).  SID_PM_NO_DEVICE_UNPERMUTE=1: two index_select + one copy as in round 3 (A/B runs).
            self.device_unpermute = (not self.identity and pin and not self.host_staged and
                                     os.environ.get('SID_PM_NO_DEVICE_UNPERMUTE') is None)
            if self.device_unpermute:
                self.perm32 = self.perm.to(torch.int32)
            elif not self.identity:
                self.full = torch.empty(n * self.ROW, dtype=torch.uint8, device=work_dev)
                self.full_out = self.full[:n * 40].view(torch.float64).view(n, 5)
                self.full_ij = self.full[n * 40:].view(torch.int32).view(n, 3)

    def local_views(self):
        """The [n_local,5] float64 and [n_local,3] int32 tensors the kernels of this rank write."""
        return self.out_local[:self.n_local], self.ij_local[:self.n_local]

    def poison(self):
        """Overwrite this rank's result rows (NaN / -1) - and on `dst` the gathered host copy - so that a following step
        that launched nothing, or gathered nothing, cannot pass a parity check on the values of an earlier step."""
        self.out_local.fill_(float('nan'))
        self.ij_local.fill_(-1)
        if self.is_dst:
            self.host_out.fill_(float('nan'))
            self.host_ij.fill_(-1)
        if self.device.type == 'cuda':
            self.torch.cuda.current_stream(self.device).synchronize()

    def gather_to_host(self):
        """The exchange step.  On `dst` the results are in host memory, in original point order, on return."""
        torch, dist, m = self.torch, self.dist, self.m
        ev = None
        if self.timing:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        if not self.collective:
            stack = self.block[None]
        elif self.host_staged:
            mine = self.block.cpu()
            dist.gather(mine, list(self.stack.unbind(0)) if self.is_dst else None, dst=self.dst, group=self.group)
            stack = self.stack if self.is_dst else None
        else:
            dist.gather(self.block, list(self.stack.unbind(0)) if self.is_dst else None, dst=self.dst, group=self.group)
            stack = self.stack if self.is_dst else None
        if ev:
            ev[1].record()
        if not self.is_dst:
            if ev:                                                 # a sending rank: its part of the gather only; read later -
                self._pending.append((ev[0], ev[1]))               # the rank does not wait for the collective here
            return
        if self.identity:
            if ev:
                ev[2].record()
            if not self.zero_copy:
                self.host.copy_(self.block, non_blocking=True)
        elif self.device_unpermute:
            from . import _capi
            _capi.unpermute(stack.data_ptr(), self.world, m, self.perm32.data_ptr(), self.n_total, self.host_out.data_ptr(),
                            self.host_ij.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
            if ev:
                ev[2].record()                                     # (the kernel's writes ARE the copy to the host)
        else:
            so = stack[:, :m * 40].reshape(-1).view(torch.float64).view(-1, 5) if stack.shape[0] == 1 else \
                stack[:, :m * 40].contiguous().view(torch.float64).view(-1, 5)
            si = stack[:, m * 40:].reshape(-1).view(torch.int32).view(-1, 3) if stack.shape[0] == 1 else \
                stack[:, m * 40:].contiguous().view(torch.int32).view(-1, 3)
            torch.index_select(so, 0, self.perm, out=self.full_out)
            torch.index_select(si, 0, self.perm, out=self.full_ij)
            if ev:
                ev[2].record()
            self.host.copy_(self.full, non_blocking=True)
        if ev:
            ev[3].record()
        if self.device.type == 'cuda':
            torch.cuda.current_stream(self.device).synchronize()
        if ev:
            for k in range(3):
                self._acc[k] += ev[k].elapsed_time(ev[k + 1])
            self._n_timed += 1

    def timings(self, reset=True):
        """Mean milliseconds per exchange step on this rank since the last reset: the gather (on the launch stream it also
        holds the wait for the slowest rank's kernels), and on `dst` the un-permutation and the copy to pinned host memory
        (one kernel does both when ``device_unpermute``: reported under ``unpermute_ms``, ``d2h_ms`` = 0)."""
        for a, b in self._pending:
            b.synchronize()
            self._acc[0] += a.elapsed_time(b)
            self._n_timed += 1
        self._pending = []
        n = max(self._n_timed, 1)
        out = {'gather_ms': self._acc[0] / n, 'unpermute_ms': self._acc[1] / n, 'd2h_ms': self._acc[2] / n, 'steps': self._n_timed}
        if reset:
            self._acc, self._n_timed = [0.0, 0.0, 0.0], 0
        return out

    def host_results(self):
        """NumPy copies of the gathered results (`dst` only)."""
        return self.host_out.numpy().copy(), self.host_ij.numpy().copy()


def per_rank_breakdown(values, device, group=None):
    """All ranks' step numbers on every rank: ``values`` (a short list of floats of THIS rank, e.g. kernel / gather /
    un-permute milliseconds) -> array [world, len(values)].  One all_gather; with no process group the one row."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return mine.numpy()[None].copy()
    world = dist.get_world_size(group)
    on_dev = dist.get_backend(group) != 'gloo'
    mine = mine.to(device) if on_dev else mine
    rows = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine, group=group)
    return torch.stack(rows).cpu().numpy()



def rebalance_with_feedback(measure_ms, reshard, cost, cuts, rounds, device, group=None):
    """Measured feedback on the shard cuts, as a collective-safe loop (used by bench.py --gpus N; any N-GPU caller can).

    ``measure_ms()``  -> this rank's kernel milliseconds with the current cuts (may raise: a refused point, a HIP error);
    ``reshard(cuts)`` -> make this rank run its shard under ``cuts`` (every rank is called with the same cuts).
    Every round is ONE all_gather of (time, ok): a rank whose measurement raised still takes part in the collective - with
    ok = 0 - and then EVERY rank raises, instead of one rank raising while the others block in the all_gather for ever.
    ``rounds`` rounds of ``rebalance_cuts``; the cuts with the smallest slowest rank are kept.  A pure function of the gathered
    times on every rank.  Returns dict(cuts, rounds, kept_slowest_kernel_ms, history)."""
    history, best = [], (float('inf'), cuts)
    for it in range(rounds + 1):
        err = None
        try:
            mine = float(measure_ms())
        except Exception as e:                                       # noqa: reported after the collective
            err, mine = e, float('nan')
        both = per_rank_breakdown([mine, 0.0 if err is not None else 1.0], device, group)
        if not (both[:, 1] > 0.5).all():
            bad = [int(r) for r in np.flatnonzero(both[:, 1] <= 0.5)]
            raise RuntimeError('rebalance round %d: the kernels of rank(s) %s failed%s' % (it, bad, ': %s' % err if err is not None else ''))
        t = both[:, 0]
        history.append({'points': np.diff(cuts).tolist(), 'kernel_ms': [round(float(v), 4) for v in t]})
        if t.max() < best[0]:
            best = (float(t.max()), cuts)
        if it < rounds:
            cuts = rebalance_cuts(cost, cuts, t)
            reshard(cuts)
    if best[1] is not cuts:
        cuts = best[1]
        reshard(cuts)
    return {'cuts': cuts, 'rounds': rounds, 'kept_slowest_kernel_ms': best[0], 'history': history}
