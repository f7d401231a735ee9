"""Sharding of grid points over the GPUs of one node and the gather of their results.

The reference's only parallelism is data parallelism over independent grid points
(multiprocessing.Pool.map at pmlib.py:442-444: index scatter, pickled 5-tuples back).
Here: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for the tests), points dealt to ranks so that every rank receives an equal
share of every search-window size, and ONE collective at the end - a gather of the
(N/G, 5) float64 (+ (N/G, 3) int32) result blocks to rank 0.  The payload is tiny (40 000
points -> 1.6 MB + 0.5 MB), so the gather is latency-bound, not xGMI-link-bound.
"""
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
