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
    """Sum the per-rank partial sums.  `local_sums`: torch tensor (any device; with backend "nccl" it must live on this
    rank's GPU) or numpy array of length NSUMS (host backends only: ShardedLikelihood keeps its own device buffer for
    RCCL).  Returns the reduced values in the same kind of container."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_sums
    if isinstance(local_sums, np.ndarray):
        if dist.get_backend(group) == "nccl":
            raise ValueError("allreduce_sums: pass a tensor on this rank's GPU with the nccl backend "
                             "(ShardedLikelihood does; a host array has no device to reduce on)")
        t = torch.from_numpy(np.ascontiguousarray(local_sums, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.numpy()
    dist.all_reduce(local_sums, op=dist.ReduceOp.SUM, group=group)
    return local_sums


def _agree_min(flag: int, group=None, device=None) -> int:
    """MIN of an integer over the ranks of the torch group (on this rank's GPU for backend nccl, on the host otherwise)."""
    import torch
    import torch.distributed as dist
    dev = f"cuda:{device}" if (dist.get_backend(group) == "nccl" and device is not None) else "cpu"
    t = torch.tensor([int(flag)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def negotiate_comm(device, group=None, timeout_s=120.0, log=None):
    """The library's own RCCL communicator (api.Comm) for this rank, or None — THE SAME ANSWER ON EVERY RANK.

    The ranks first agree that each of them could bind RCCL (gpv_rccl_version() > 0 everywhere), then create the
    communicator (rank 0's id travels over the torch group; gpv_comm_create is collective and proves itself with a
    two-double all-reduce), each in a helper thread joined with `timeout_s`, and finally agree that all of them
    succeeded.  Any failure anywhere => every rank returns None and the caller takes the torch.distributed route
    (dist.all_reduce on the plan's device buffer).  Without this agreement a rank whose dlopen failed would wait in
    dist.all_reduce while the others wait inside ncclAllReduce of a communicator it never joined.
    Returns (comm_or_None, reason)."""
    import os
    import threading
    from .api import Comm
    say = log or (lambda m: None)
    forced = os.environ.get("GPV_TORCH_ALLREDUCE", "0") == "1"
    try:
        ver = Comm.rccl_version()
    except Exception as e:                                    # pragma: no cover  (a library without the symbol)
        ver, _ = 0, say(f"gpv_rccl_version failed: {e!r}")
    if _agree_min(1 if (ver >= 2000 and not forced) else 0, group, device) == 0:
        return None, ("GPV_TORCH_ALLREDUCE=1" if forced else f"RCCL not bound on every rank (this rank: version {ver})")
    import torch.distributed as dist
    box = {}
    # the id travels over the torch group on THIS thread (a helper thread's current CUDA device is device 0 on every rank:
    # an nccl broadcast issued from it would run on the wrong GPU for every rank but the first); only the collective
    # gpv_comm_create, which binds its device itself, runs on the helper thread and can be timed out
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ident = None
    try:
        ident = Comm.exchange_id(rank, Comm.torch_exchange(group))
    except Exception as e:
        say(f"communicator id not exchanged: {e!r}")
    if _agree_min(1 if ident is not None else 0, group, device) == 0:
        return None, "the communicator id could not be made or exchanged"

    def work():
        try:
            box["comm"] = Comm(device, rank, world, ident=ident)
        except Exception as e:
            box["err"] = e

    th = threading.Thread(target=work, name="gpv-comm-create", daemon=True)
    th.start()
    th.join(timeout_s)
    ok = (not th.is_alive()) and "comm" in box
    if th.is_alive():
        say(f"gpv_comm_create did not return within {timeout_s:.0f} s")
    elif "err" in box:
        say(f"library communicator unavailable: {box['err']!r}")
    if _agree_min(1 if ok else 0, group, device) == 0:
        box.pop("comm", None)                                 # (a communicator only some ranks hold is destroyed, not used)
        return None, "gpv_comm_create failed or timed out on at least one rank"
    return box["comm"], f"gpv_comm (RCCL {ver})"


class ShardedLikelihood:
    """vecchia_likelihood() for cond.yz='z' with the rows split over the ranks of a process group.

    plan_factory(row_begin, row_end) must return an object with set_data / eval / sums (api.Plan on a
    GPU box; the CPU tests inject a stand-in that computes its shard's sums with the oracle).

    With the "nccl" (= RCCL) backend the rank is bound to ONE GPU: `device` (default: the plan's device, else
    LOCAL_RANK).  The library itself owns an RCCL communicator (api.Comm; its id travels over the torch group once) and
    enqueues the all-reduce of the 8 partial sums and their 64-byte copy to pinned host memory on the evaluation's own
    stream right behind the kernel: torch.distributed is the launcher and the rendezvous, not part of the step.
    The ranks AGREE on that route first (negotiate_comm); if any of them cannot bind RCCL or create the communicator, or
    with GPV_TORCH_ALLREDUCE=1, all of them take the older route (dist.all_reduce on the plan's device buffer).
    `route` says which one runs."""

    def __init__(self, n_rows, z_ord, plan_factory, rank=None, world=None, group=None, device=None, comm_timeout_s=120.0):
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank, self.world, self.group = rank, world, group
        self.n = int(n_rows)
        self.row_begin, self.row_end = shard_rows(self.n, rank, world)
        self._nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if self._nccl:
            import os
            import torch
            if device is None:
                device = int(os.environ.get("LOCAL_RANK", rank))
            if not (0 <= int(device) < torch.cuda.device_count()):
                raise ValueError(f"rank {rank}: GPU {device} does not exist ({torch.cuda.device_count()} visible)")
            torch.cuda.set_device(device)                  # RCCL reduces on the CURRENT device of each rank
        self.plan = plan_factory(self.row_begin, self.row_end)
        self.route = "torch.distributed all_reduce (host backend)" if dist.is_initialized() else "none (1 rank)"
        if self._nccl:
            pdev = getattr(self.plan, "device", device)
            if pdev != device:
                raise ValueError(f"rank {rank}: plan lives on GPU {pdev} but the rank is bound to GPU {device}")
            self.device = device
            # the route is AGREED between the ranks (negotiate_comm): the library's communicator everywhere, or
            # dist.all_reduce everywhere
            self._comm, self.route = (None, "plan without set_comm") if not hasattr(self.plan, "set_comm") else \
                negotiate_comm(device, group, timeout_s=comm_timeout_s)
            self._native = self._comm is not None
            if self._native:
                self.plan.set_comm(self._comm)
            self._stream = torch.cuda.Stream(device=device)
            self._d = torch.zeros(NSUMS, dtype=torch.float64, device=f"cuda:{device}")
            self._h = torch.zeros(NSUMS, dtype=torch.float64).pin_memory()
            self._done = torch.cuda.Event()
        self.plan.set_data(z_ord)

    def sums(self, covmodel, covparms, nuggets, flags):
        if self._nccl and self._native:
            self.plan.eval(covmodel, covparms, nuggets, flags, stream=self._stream.cuda_stream)
            return np.asarray(self.plan.sums(), dtype=np.float64)     # totals of the whole job (polled, not slept on)
        if self._nccl:
            import torch
            import torch.distributed as dist
            with torch.cuda.stream(self._stream):
                self.plan.eval(covmodel, covparms, nuggets, flags, stream=self._stream.cuda_stream,
                               d_sums_out=self._d.data_ptr())
                dist.all_reduce(self._d, op=dist.ReduceOp.SUM, group=self.group)   # the ONE collective: 64 bytes
                self._h.copy_(self._d, non_blocking=True)
                self._done.record(self._stream)
            while not self._done.query():              # poll the event's flag: a blocking wait costs an interrupt wake-up
                pass
            return self._h.numpy().copy()
        self.plan.eval(covmodel, covparms, nuggets, flags)
        s = np.asarray(self.plan.sums(), dtype=np.float64)
        assert s.shape == (NSUMS,)
        return allreduce_sums(s, self.group)

    def loglik(self, covmodel, covparms, nuggets):
        from ._lib import GPV_WANT_LOGLIK_Z
        s = self.sums(covmodel, covparms, nuggets, GPV_WANT_LOGLIK_Z)
        if s[6] > 0:
            return float("-inf")              # failed block => zero row of U => logdet.num = +Inf (R/vecchia_likelihood.R:76)
        return float(-0.5 * (s[2] + s[3] + self.n * np.log(2.0 * np.pi)))
