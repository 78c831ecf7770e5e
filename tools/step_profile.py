"""Developer tool: host-side time of each piece of one bench step at the per-rank load of an 8-GPU run (n/8 rows),
under torchrun with one rank (NCCL world 1)."""
import os, sys, time
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gpvecchia_amd as G
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, m, d = 125_000, 30, 2
locs, z, revNN, revCond, a, b = bench.build_workload(n, m, d, 0, 1, device=0)
plan = G.Plan(locs, revNN, revCond, device=0, row_begin=a, row_end=b)
plan.set_data(z)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
sums = torch.zeros(8, dtype=torch.float64, device="cuda"); pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
T = np.zeros(6); K = 300
for it in range(K + 20):
    t0 = time.perf_counter()
    plan.eval("matern", [1.0, 0.02, 1.5], 0.1, G.GPV_WANT_LOGLIK_Z, stream=ts.cuda_stream, d_sums_out=sums.data_ptr())
    t1 = time.perf_counter()
    dist.all_reduce(sums)
    t2 = time.perf_counter()
    pinned.copy_(sums, non_blocking=True)
    t3 = time.perf_counter()
    ts.synchronize()
    t4 = time.perf_counter()
    ll = G.loglik_z_from_sums(pinned.numpy(), n)
    t5 = time.perf_counter()
    if it >= 20:
        T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0]
print("us per step: eval-enqueue %.1f  all_reduce-enqueue %.1f  copy-enqueue %.1f  sync-wait %.1f  loglik %.1f  total %.1f  (kernel %.1f)" % (*(T / K * 1e6), plan.last_kernel_ms() * 1e3))
dist.destroy_process_group()
