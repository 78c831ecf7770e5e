"""Row-sharded likelihood evaluation over the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests).  Conditioning sets are independent (src/U_NZentries.cpp:39: no cross-row reads or
writes), so each rank owns a contiguous block of rows; locations, nuggets and data are
replicated; the only exchange is ONE all-reduce(sum) of the 8-double partial-sum vector per
evaluation.  The U factor stays sharded on the devices.
"""
from __future__ import annotations

import numpy as np

from ._lib import NSUMS


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous, balanced split (rows beyond the first m all have n0 = m+1 => equal work)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n_rows) // world, ((rank + 1) * n_rows) // world


def allreduce_sums(local_sums, group=None):
    """Sum the per-rank partial sums.  `local_sums`: torch tensor (any device) or numpy array of
    length NSUMS.  Returns the reduced values in the same kind of container."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_sums
    if isinstance(local_sums, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local_sums, dtype=np.float64).copy())
        if dist.get_backend(group) == "nccl":
            # pinned staging both ways and a wait on the current stream only: `tensor.cpu()` right after an RCCL
            # collective costs ~0.2 ms (measured in bench.py), the collective itself ~0.01 ms at 64 bytes
            d = t.pin_memory().cuda(non_blocking=True)
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
            out = torch.empty_like(t).pin_memory()
            out.copy_(d, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return out.numpy().copy()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.numpy()
    dist.all_reduce(local_sums, op=dist.ReduceOp.SUM, group=group)
    return local_sums


class ShardedLikelihood:
    """vecchia_likelihood() for cond.yz='z' with the rows split over the ranks of a process group.

    plan_factory(row_begin, row_end) must return an object with set_data / eval / sums (api.Plan on a
    GPU box; the CPU tests inject a stand-in that computes its shard's sums with the oracle)."""

    def __init__(self, n_rows, z_ord, plan_factory, rank=None, world=None, group=None):
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank, self.world, self.group = rank, world, group
        self.n = int(n_rows)
        self.row_begin, self.row_end = shard_rows(self.n, rank, world)
        self.plan = plan_factory(self.row_begin, self.row_end)
        self.plan.set_data(z_ord)

    def sums(self, covmodel, covparms, nuggets, flags):
        self.plan.eval(covmodel, covparms, nuggets, flags)
        s = np.asarray(self.plan.sums(), dtype=np.float64)
        assert s.shape == (NSUMS,)
        return allreduce_sums(s, self.group)

    def loglik(self, covmodel, covparms, nuggets):
        from ._lib import GPV_WANT_LOGLIK_Z
        s = self.sums(covmodel, covparms, nuggets, GPV_WANT_LOGLIK_Z)
        if s[6] > 0:
            return float("nan")
        return float(-0.5 * (s[2] + s[3] + self.n * np.log(2.0 * np.pi)))
