"""Sharding of grid points over the GPUs of one node and the gather of their results.

The reference's only parallelism is data parallelism over independent grid points
(multiprocessing.Pool.map at pmlib.py:442-444: index scatter, pickled 5-tuples back).
Here: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for the tests), points dealt to ranks so that every rank receives an equal
share of every search-window size, and ONE collective at the end - a gather of the
(N/G, 5) float64 (+ (N/G, 3) int32) result blocks to rank 0.  The payload is tiny (40 000
points -> 1.6 MB + 0.5 MB), so the gather is latency-bound, not xGMI-link-bound.
"""
import os

import numpy as np


def shard_indices(border, world_size, rank):
    """Indices (into the original point order) owned by `rank`.

    Work per point grows with (2*border+2)^2, so points are ordered by border (largest
    first, stable) and dealt in snake order (0..G-1, G-1..0, ...): every rank gets the same
    mix of window sizes and no rank is systematically first in every round.
    """
    border = np.asarray(border)
    order = np.argsort(-border, kind='stable')
    pos = np.arange(order.size)
    k, rnd = pos % world_size, pos // world_size
    owner = np.where(rnd % 2 == 0, k, world_size - 1 - k)
    return np.sort(order[owner == rank])


def point_cost(border, img_size=34, n_angles=15):
    """Estimated cost of a grid point (nanoseconds on one MI355X; only the ratios matter here).  Computed by the library
    (include/sid_pm.h ``sid_pm_estimate_cost``, host arithmetic) from what the kernel executes for a point of that search
    border - matrix instructions of the sweep and of the winner's NCC matrix, placements - and from the residency class of
    its LDS footprint, for the template side and angle count of the run; it reproduces the staircase that
    tools/border_cost.py measures (placement tiles per output row, work items per wavefront, workgroups per CU) within 4 %."""
    from . import _capi
    return _capi.estimate_cost(np.asarray(border, dtype=np.float64), img_size, n_angles)


# tail of a launch in run times of one of its workgroups, by residency class (workgroups per CU): with 256 x class points in
# flight the last round is half empty on average; the classes with few slots run the large borders, whose run times differ by
# up to 1.5x inside one launch, and end less evenly.  Fitted to the shards tools/shard_sim.py measures on the GPU
# (8 shards of the benchmark grid: 0.5 everywhere left the one-per-CU shard 6.5 % and the mixed one 7.4 % above the others).
_TAIL_ROUNDS = {1: 1.0, 2: 0.7, 3: 0.5, 4: 0.5}


def _shard_times(cost, cls, cuts):
    """Estimated kernel time of every shard [cuts[r], cuts[r+1]) of the border-ordered points: per launch class the sum of
    the point costs plus the tail of the launch (``_TAIL_ROUNDS`` x the run time of a workgroup = slots x mean cost); a launch
    shorter than one round still takes a full one."""
    t = np.zeros(len(cuts) - 1)
    for r in range(len(cuts) - 1):
        a, b = cuts[r], cuts[r + 1]
        for c in np.unique(cls[a:b]):
            sel = cost[a:b][cls[a:b] == c]
            latency = 256.0 * c * sel.mean()
            t[r] += max(sel.sum() + _TAIL_ROUNDS.get(int(c), 0.5) * latency, latency)
    return t


def shard_indices_by_cost(border, world_size, rank, img_size=34, n_angles=15):
    """Indices owned by `rank` when the points, ordered by border (largest first, stable), are cut into `world_size`
    contiguous runs of equal estimated TIME.  A rank then holds one or two neighbouring border classes instead of an
    eighth of every class, i.e. one or two launches that are eight times longer: at 5 000 points per rank the tails of
    three short launches cost a quarter of the step.  The time of a run is the cost of its points
    (``point_cost``) plus the tails of its launches (``_shard_times``): cut by cost alone, the rank with the largest borders -
    one workgroup per CU, 7 rounds of 256 points - came out 6 % slower than the others on the GPU (tools/shard_sim.py).

    Every rank's kernels must finish before the gather can complete, so the step time is the slowest rank's kernel time
    plus the exchange step; shortening rank 0's shard would not hide the exchange (it starts when the LAST rank is done).
    Pure function of its arguments: every rank computes the same cuts."""
    from . import _capi
    border = np.asarray(border)
    order = np.argsort(-border, kind='stable')
    n = order.size
    if n == 0 or world_size <= 1:
        return np.sort(order) if rank == 0 else np.zeros(0, dtype=order.dtype)
    cost = point_cost(border[order], img_size, n_angles)
    cls = _capi.estimate_residency(border[order], img_size, n_angles)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    share = np.full(world_size, cum[-1] / world_size)               # cost each shard is to hold

    def cuts_for(share):
        # point k goes to the shard whose cost interval holds the middle of its own
        edges = np.cumsum(share)[:-1]
        mid = cum[:-1] + 0.5 * cost
        c = np.searchsorted(mid, edges, side='left')
        return np.concatenate([[0], c, [n]]).astype(np.int64)
    best_cuts, best = None, np.inf
    for _ in range(24):                                             # move cost from the slow shards to the fast ones
        cuts = cuts_for(share)
        t = _shard_times(cost, cls, cuts)
        if t.max() < best:
            best, best_cuts = t.max(), cuts
        share = np.maximum(share + 0.7 * (t.mean() - t), 0.0)
        share *= cum[-1] / share.sum()
    a, b = best_cuts[rank], best_cuts[rank + 1]
    return np.sort(order[a:b])


def shard_size(n_total, world_size):
    """Rows of the padded per-rank block (equal on all ranks so one gather suffices)."""
    return (int(n_total) + world_size - 1) // world_size


class ResultGatherer(object):
    """Gather padded per-rank result blocks to rank `dst` and undo the sharding.

    The index exchange happens once at construction; each `gather` call is then one
    collective per result array (RCCL gather over xGMI with device tensors, gloo with CPU
    tensors) plus an index_copy on rank `dst`.
    """

    def __init__(self, n_total, idx_local, device, group=None, dst=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group, self.dst = group, dst
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n_total = int(n_total)
        self.m = shard_size(n_total, self.world)
        self.device = device
        idx_pad = torch.full((self.m,), -1, dtype=torch.int64, device=device)
        idx_t = torch.as_tensor(np.asarray(idx_local), dtype=torch.int64, device=device)
        if idx_t.numel() > self.m:
            raise ValueError('shard larger than the padded block')
        idx_pad[:idx_t.numel()] = idx_t
        self.is_dst = self.rank == dst
        if self.world == 1:
            all_idx = idx_pad[None]
        else:
            gl = [torch.empty_like(idx_pad) for _ in range(self.world)] if self.is_dst else None
            dist.gather(idx_pad, gl, dst=dst, group=group)
            all_idx = torch.stack(gl) if self.is_dst else None
        if self.is_dst:
            flat = all_idx.reshape(-1)
            self.sel = torch.nonzero(flat >= 0).reshape(-1)        # rows of the stacked blocks that are real
            self.dest = flat[self.sel]                             # their original indices
            self.buf_out = torch.empty((self.world, self.m, 5), dtype=torch.float64, device=device)
            self.buf_ij = torch.empty((self.world, self.m, 3), dtype=torch.int32, device=device)

    def gather(self, out_local, ij_local=None):
        """out_local: float64 [m,5]; ij_local: int32 [m,3] or None.  Returns (out, ij) on dst
        (tensors in original point order, NaN / -1 where no rank produced a row), else (None, None)."""
        torch, dist = self.torch, self.dist
        if tuple(out_local.shape) != (self.m, 5):
            raise ValueError('out_local must be [%d,5]' % self.m)
        if self.world == 1:
            stacked_out = out_local[None]
            stacked_ij = ij_local[None] if ij_local is not None else None
        else:
            dist.gather(out_local, list(self.buf_out.unbind(0)) if self.is_dst else None, dst=self.dst, group=self.group)
            if ij_local is not None:
                dist.gather(ij_local, list(self.buf_ij.unbind(0)) if self.is_dst else None, dst=self.dst,
                            group=self.group)
            if not self.is_dst:
                return None, None
            stacked_out = self.buf_out
            stacked_ij = self.buf_ij if ij_local is not None else None
        out = torch.full((self.n_total, 5), float('nan'), dtype=torch.float64, device=self.device)
        out[self.dest] = stacked_out.reshape(-1, 5)[self.sel]
        ij = None
        if stacked_ij is not None:
            ij = torch.full((self.n_total, 3), -1, dtype=torch.int32, device=self.device)
            ij[self.dest] = stacked_ij.reshape(-1, 3)[self.sel]
        return out, ij


class PackedGatherer(object):
    """One collective per step: every rank's results live in ONE padded byte block
    [m x 5 float64 | m x 3 int32] that the kernels write in place (``local_views``), so the
    exchange step of the path is a single gather of that block to rank `dst` (RCCL over xGMI
    with device tensors), followed there by one index_select per array that undoes the
    sharding and an asynchronous copy into pinned host memory.  With a ``gloo`` group (CPU
    tests, several ranks sharing one device in a dry run) the block is staged through the host.
    """

    ROW = 5 * 8 + 3 * 4

    def __init__(self, n_total, idx_local, device, group=None, dst=0, force_collective=False, timing=False):
        """``force_collective``: run every collective (the all_reduce and the two gathers) even in a group of ONE rank - the
        RCCL code path of an N-GPU run then executes on a one-GPU box (a world-size-1 ``nccl`` group), where it can be
        tested.  ``timing``: HIP events around the three stages of the exchange step (``timings()``)."""
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
        self._ev, self._acc, self._n_timed = [], [0.0, 0.0, 0.0], 0
        # rows of the padded block = the largest shard (the shards of shard_indices_by_cost differ in length)
        if self.collective:
            t = torch.tensor([self.n_local], dtype=torch.int64,
                             device='cpu' if dist.get_backend(group) == 'gloo' else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            m = int(t.item())
        else:
            m = self.n_local
        self.m = m = max(m, 1)
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
            # un-permuted results: one buffer [n x 5 float64 | n x 3 int32] on the device and its pinned twin on the host, so
            # that the copy to the host is ONE transfer (a second 0.5 MB copy costs as much as the first 1.6 MB one)
            n = self.n_total
            # one rank, no forced collective, points already in their own order: the block IS the result - no un-permutation
            self.identity = (not self.collective) and m == n and bool((self.perm == torch.arange(n, device=self.perm.device)).all())
            self.full = torch.empty(n * self.ROW, dtype=torch.uint8, device=work_dev)
            self.full_out = self.full[:n * 40].view(torch.float64).view(n, 5)
            self.full_ij = self.full[n * 40:].view(torch.int32).view(n, 3)
            pin = self.device.type == 'cuda'
            self.host = torch.empty(n * self.ROW, dtype=torch.uint8, pin_memory=pin)
            self.host_out = self.host[:n * 40].view(torch.float64).view(n, 5)
            self.host_ij = self.host[n * 40:].view(torch.int32).view(n, 3)
            # zero copy: with one rank, the points in their own order and no forced collective, the kernels write their 52 B per
            # point straight into the pinned host buffer (device-visible on ROCm: hipHostMalloc) - 2 MB of posted PCIe writes
            # spread over the step instead of a 50 us copy after it.  SID_PM_NO_ZERO_COPY=1: copy as before (A/B runs).
            self.zero_copy = self.identity and pin and os.environ.get('SID_PM_NO_ZERO_COPY') is None
            if self.zero_copy:
                self.host_out.fill_(float('nan'))
                self.host_ij.fill_(-1)
                self.out_local, self.ij_local = self.host_out, self.host_ij

    def local_views(self):
        """The [n_local,5] float64 and [n_local,3] int32 tensors the kernels of this rank write."""
        return self.out_local[:self.n_local], self.ij_local[:self.n_local]

    def gather_to_host(self):
        """The exchange step.  On `dst` the results are in host memory, in original point order, on return."""
        torch, dist, m = self.torch, self.dist, self.m
        ev = None
        if self.timing and self.is_dst:
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
        if not self.is_dst:
            return
        if ev:
            ev[1].record()
        if getattr(self, 'identity', False):
            if ev:
                ev[2].record()
            if not self.zero_copy:
                self.host.copy_(self.block, non_blocking=True)
            if ev:
                ev[3].record()
            if self.device.type == 'cuda':
                torch.cuda.current_stream(self.device).synchronize()
            if ev:
                for k in range(3):
                    self._acc[k] += ev[k].elapsed_time(ev[k + 1])
                self._n_timed += 1
            return
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
        """Mean milliseconds per exchange step on `dst` since the last reset: the gather (on the launch stream it also
        holds the wait for the slowest rank's kernels), the un-permutation, the copy to pinned host memory."""
        n = max(self._n_timed, 1)
        out = {'gather_ms': self._acc[0] / n, 'unpermute_ms': self._acc[1] / n, 'd2h_ms': self._acc[2] / n, 'steps': self._n_timed}
        if reset:
            self._acc, self._n_timed = [0.0, 0.0, 0.0], 0
        return out

    def host_results(self):
        """NumPy copies of the gathered results (`dst` only)."""
        return self.host_out.numpy().copy(), self.host_ij.numpy().copy()
