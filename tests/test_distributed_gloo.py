"""world_size-2 gloo test of the row-sharded likelihood (the N>1 path of bench.py): each rank
evaluates its shard of conditioning sets, ONE all-reduce of the 8 partial sums, same log-likelihood
on every rank as the unsharded oracle value.  On CPU the per-shard sums come from the oracle
(injected plan stand-in); the sharding, reduction and closed form are the product code."""
import os
import socket

import numpy as np
import pytest


class _OraclePlan:
    """Stand-in for api.Plan on a GPU-less box: same interface, shard sums from the oracle."""

    def __init__(self, va, a, b):
        self.va, self.a, self.b = va, a, b

    def set_data(self, z_ord):
        self.z = np.asarray(z_ord)

    def eval(self, covmodel, covparms, nuggets, flags):
        from oracle import r_side as R
        va = self.va
        n = va["locsord"].shape[0]
        tau = np.repeat(np.atleast_1d(nuggets), n) if np.size(nuggets) == 1 else np.asarray(nuggets)
        ent = R.U_NZentries(1, n, va["locsord"], np.nan_to_num(va["U_prep"]["revNNarray"]), va["U_prep"]["revCond"],
                            tau, tau, covmodel, covparms)
        s = np.zeros(8)
        revNN, revCond, L = va["U_prep"]["revNNarray"], va["U_prep"]["revCond"], ent["Lentries"]
        for k in range(self.a, self.b):
            ok = ~np.isnan(revNN[k]); n0 = int(ok.sum())
            idx = revNN[k, ok].astype(int) - 1
            M = L[k, :n0]; d = M[-1]; v = 1 / d ** 2
            ak = float(np.sum(M[:-1] * self.z[idx[:-1]] * (revCond[k, ok][:-1] == 0)))
            mu = -ak / d
            s[2] += np.log(tau[k] + v)
            s[3] += (self.z[k] - mu) ** 2 / (tau[k] + v)
            s[7] += 1
        self._s = s

    def sums(self):
        return self._s


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpvecchia_amd.distributed import ShardedLikelihood
        from oracle import r_side as R
        rng = np.random.default_rng(0)
        n = 301
        locs = rng.random((n, 2)); z = rng.standard_normal(n)
        va = R.vecchia_specify(locs, 8, ordering="none", cond_yz="z")
        sl = ShardedLikelihood(n, z, lambda a, b: _OraclePlan(va, a, b))
        ll = sl.loglik("matern", [1.0, 0.2, 1.5], 0.1)
        q.put((rank, sl.row_begin, sl.row_end, ll))
    finally:
        dist.destroy_process_group()


def test_sharded_likelihood_world2_gloo():
    import torch.multiprocessing as mp
    from oracle import r_side as R
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    locs = rng.random((301, 2)); z = rng.standard_normal(301)
    va = R.vecchia_specify(locs, 8, ordering="none", cond_yz="z")
    ref = R.vecchia_likelihood(z, va, [1.0, 0.2, 1.5], 0.1)
    assert res[0][1:3] == (0, 150) and res[1][1:3] == (150, 301)       # contiguous balanced shards
    assert res[0][3] == res[1][3]                                        # every rank holds the same reduced value
    assert abs(res[0][3] - ref) <= 1e-10 * abs(ref)


def test_shard_rows_partition():
    from gpvecchia_amd.distributed import shard_rows
    for n in (0, 1, 7, 1_000_000):
        for w in (1, 2, 3, 8):
            cuts = [shard_rows(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_rows(10, 2, 2)


def _negotiate_worker(rank, world, port, q, force_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if rank == force_rank:
        os.environ["GPV_TORCH_ALLREDUCE"] = "1"                    # ONE rank cannot (will not) use the library's communicator
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpvecchia_amd.distributed import negotiate_comm
        msgs = []
        comm, why = negotiate_comm(0, None, timeout_s=30, log=msgs.append)
        q.put((rank, comm is None, why, msgs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("force_rank", [1, -1])
def test_ranks_agree_on_the_collective_route_world2_gloo(force_rank):
    """negotiate_comm returns the same answer on every rank and never leaves one rank inside a collective the other does
    not enter: with one rank opting out (force_rank = 1) both fall back; with none opting out on this GPU-less box
    gpv_comm_create fails on both (no device) and both fall back as well — in both cases within seconds, no hang."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, 2, port, q, force_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import torch
    if torch.cuda.device_count() == 0 or force_rank >= 0:
        assert res[0][1] and res[1][1], res                          # both None
    assert res[0][1] == res[1][1]                                    # the same route on both ranks, whatever it is
