"""Developer tool: what one rank pays per step at the per-rank load of an 8-GPU run (n/8 rows), for several ways of getting the
8 sums to the host.  Run under torchrun with one rank (RCCL world 1):
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/step_profile2.py"""
import os, sys, time
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gpvecchia_amd as G
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, m, d = 1_000_000, 30, 2
locs, z, revNN, revCond, a, b = bench.build_workload(n, m, d, 0, 8, device=0)
plan = G.Plan(locs, revNN, revCond, device=0, row_begin=a, row_end=b)
plan.set_data(z)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
sums = torch.zeros(8, dtype=torch.float64, device="cuda"); pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
done = torch.cuda.Event()
cp = np.array([1.0, 0.02, 1.5]); tau = np.array([0.1])


class _Alias:
    def __init__(self, ptr, nel):
        self.__cuda_array_interface__ = {"shape": (nel,), "typestr": "<f8", "data": (ptr, False), "version": 2}


try:
    alias = torch.as_tensor(_Alias(pinned.data_ptr(), 8), device="cuda")
except Exception as e:
    alias = None
    print("alias failed:", repr(e))


def run(name, body, K=400):
    for _ in range(30):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        body()
    el = (time.perf_counter() - t0) / K * 1e6
    print(f"{name:58s} {el:8.1f} us/step   loglik {G.loglik_z_from_sums(pinned.numpy(), n):.6f}", flush=True)
    return el


def poll():
    done.record(ts)
    while not done.query():
        pass


def v0():
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=sums.data_ptr())
    dist.all_reduce(sums)
    pinned.copy_(sums, non_blocking=True)
    poll()


def v0s():
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=sums.data_ptr())
    dist.all_reduce(sums)
    pinned.copy_(sums, non_blocking=True)
    ts.synchronize()


def v2():
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=sums.data_ptr())
    pinned.copy_(sums, non_blocking=True)
    poll()


def v3():
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=pinned.data_ptr())
    poll()


def v4():
    plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=pinned.data_ptr())
    dist.all_reduce(alias)
    poll()


plan.eval("matern", cp, tau, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=sums.data_ptr())
ts.synchronize()
print("kernel ms", plan.last_kernel_ms())
run("V0  dev sums + all_reduce + D2H copy + event poll", v0)
run("V0s same with stream.synchronize()", v0s)
run("V2  no all_reduce: D2H copy + poll", v2)
run("V3  kernel writes pinned host memory, poll (no copy)", v3)
if alias is not None:
    try:
        run("V4  kernel -> pinned, all_reduce ON the pinned alias, poll", v4)
    except Exception as e:
        print("V4 failed:", repr(e))
plan.set_kernel_timing(False)
run("V0  + kernel timing events off", v0)
run("V3  + kernel timing events off", v3)
if alias is not None:
    try:
        run("V4  + kernel timing events off", v4)
    except Exception as e:
        print("V4 failed:", repr(e))
dist.destroy_process_group()
