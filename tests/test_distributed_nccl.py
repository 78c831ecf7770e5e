"""RCCL ("nccl" backend) path of the row-sharded likelihood on real GPUs: ShardedLikelihood binds each rank to one
device, the set kernel deposits its 8 partial sums in the buffer RCCL all-reduces on the same stream.
world 1 runs in-process on the 1-GPU box; world 2 starts two rank processes and needs two GPUs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _problem():
    from gpvecchia_amd import specify as S
    rng = np.random.default_rng(12)
    n, m = 6001, 20
    locs = rng.random((n, 2)); z = rng.standard_normal(n)
    NN = S.find_ordered_nn(locs, m)
    revNN = NN[:, ::-1].copy()
    revCond = np.where(revNN != 0, 0, -1).astype(np.int8); revCond[:, -1] = 1
    return n, locs, z, revNN, revCond, [1.1, 0.07, 1.5], 0.2


def test_library_owned_communicator_world1():
    """gpv_comm without torch.distributed anywhere: the library binds RCCL itself and all-reduces inside gpv_plan_eval."""
    import torch
    import gpvecchia_amd as G
    n, locs, z, revNN, revCond, cp, tau = _problem()
    ref = G.Plan(locs, revNN, revCond)
    ref.set_data(z)
    ref.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z)
    s1 = ref.sums()
    handed = []

    def exchange(mine):
        handed.append(mine)
        return mine
    comm = G.Comm(0, 0, 1, exchange)
    assert len(handed) == 1 and len(handed[0]) == 128 and any(handed[0])
    plan = G.Plan(locs, revNN, revCond)
    plan.set_data(z)
    plan.set_comm(comm)
    for _ in range(3):                                              # the buffers and the event are reusable
        plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z)
        assert np.array_equal(plan.sums(), s1)
    st = torch.cuda.Stream()
    dev = torch.zeros(G._lib.NSUMS, dtype=torch.float64, device="cuda")
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z | G.GPV_WANT_U, stream=st.cuda_stream, d_sums_out=dev.data_ptr())
    st.synchronize()
    assert np.array_equal(dev.cpu().numpy(), s1)
    assert np.array_equal(plan.Lentries(), (ref.eval("matern", cp, tau, G.GPV_WANT_U), ref.Lentries())[1])
    with pytest.raises(G.GpvError) as e:                            # the posterior pass does not shard
        plan.eval("matern", cp, tau, G.GPV_WANT_DENOM)
    assert e.value.status == 7
    plan.set_comm(None)
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z)
    assert np.array_equal(plan.sums(), s1)
    with pytest.raises(ValueError):
        G.Comm(0, 0, 1, lambda mine: b"short")
    with pytest.raises(G.GpvError):
        G.Comm(99, 0, 1, lambda mine: mine)                         # no such device


@pytest.mark.parametrize("route", ["library", "torch"])
def test_sharded_likelihood_rccl_world1_in_process(route, monkeypatch):
    monkeypatch.setenv("GPV_TORCH_ALLREDUCE", "1" if route == "torch" else "0")
    import torch
    import torch.distributed as dist
    import gpvecchia_amd as G
    from gpvecchia_amd.distributed import ShardedLikelihood
    from oracle import r_side as R
    assert torch.cuda.is_available()
    n, locs, z, revNN, revCond, cp, tau = _problem()
    ref = G.Plan(locs, revNN, revCond)
    ref.set_data(z)
    ref.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z)
    ll1 = G.loglik_z_from_sums(ref.sums(), n)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        sl = ShardedLikelihood(n, z, lambda a, b: G.Plan(locs, revNN, revCond, device=0, row_begin=a, row_end=b), device=0)
        assert sl._nccl and (sl.row_begin, sl.row_end) == (0, n) and sl._native == (route == "library")
        ll = sl.loglik("matern", cp, tau)
        assert ll == ll1                                             # same kernel, same rows, sum of one shard
        assert sl.loglik("matern", cp, tau) == ll                    # the stream/buffer pair is reusable
        with pytest.raises(ValueError):
            ShardedLikelihood(n, z, lambda a, b: G.Plan(locs, revNN, revCond, device=0, row_begin=a, row_end=b), device=1)
    finally:
        dist.destroy_process_group()
    va = dict(U_prep=dict(revNNarray=np.where(revNN == 0, np.nan, revNN.astype(float)),
                          revCond=np.where(revCond < 0, np.nan, revCond.astype(float))), ord_z=np.arange(1, n + 1))
    ent = R.U_NZentries(R.max_threads(), n, locs, revNN, np.where(revCond < 0, 0, revCond).astype(float), np.full(n, tau),
                        np.full(n, tau), "matern", cp)
    ll_ref, _ = R.separable_loglik_condz(va, ent, z, tau)
    assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref)


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import gpvecchia_amd as G
from gpvecchia_amd.distributed import ShardedLikelihood
sys.path.insert(0, os.path.join({root!r}, "tests"))
from test_distributed_nccl import _problem
lr = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
n, locs, z, revNN, revCond, cp, tau = _problem()
sl = ShardedLikelihood(n, z, lambda a, b: G.Plan(locs, revNN, revCond, device=lr, row_begin=a, row_end=b))
ll = sl.loglik("matern", cp, tau)
print("RESULT " + json.dumps(dict(rank=dist.get_rank(), a=sl.row_begin, b=sl.row_end, ll=ll, route=sl.route)), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_sharded_likelihood_rccl_world2(tmp_path):
    import json
    import torch
    import gpvecchia_amd as G
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device); world 1 covers the code path on this box")
    n, locs, z, revNN, revCond, cp, tau = _problem()
    ref = G.Plan(locs, revNN, revCond)
    ref.set_data(z)
    ref.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z)
    ll1 = G.loglik_z_from_sums(ref.sums(), n)
    script = str(tmp_path / "nccl_worker.py")
    open(script, "w").write(_WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = sorted((json.loads(l[7:]) for l in r.stdout.splitlines() if l.startswith("RESULT ")), key=lambda d: d["rank"])
    assert len(res) == 2 and (res[0]["a"], res[0]["b"], res[1]["b"]) == (0, n // 2, n)
    assert res[0]["ll"] == res[1]["ll"] and res[0]["route"] == res[1]["route"]       # the ranks agreed on ONE route
    assert abs(res[0]["ll"] - ll1) <= 1e-12 * abs(ll1)
