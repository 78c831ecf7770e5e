"""Developer tool: per-step time of the library-owned RCCL route (world 1, rows = n/8) in windows after the communicator's
first use."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gpvecchia_amd as G
import torch

ci, n, m, d, nu, rng_ = bench.CONFIGS["C3"]
locs, z, revNN, revCond, a, b = bench.build_workload(n, m, d, 0, 8, device=0)
plan = G.Plan(locs, revNN, revCond, device=0, row_begin=a, row_end=b)
plan.set_data(z)
plan.set_kernel_timing(False)
cp = [1.0, rng_, nu]
st = torch.cuda.Stream()

def windows(tag, k=3000, w=250):
    ts = np.empty(k + 1)
    ts[0] = time.perf_counter()
    te = np.empty(k)
    for i in range(k):
        plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z, stream=st.cuda_stream)
        te[i] = time.perf_counter()
        plan.sums()
        ts[i + 1] = time.perf_counter()
    dt = 1e6 * np.diff(ts)
    enq = 1e6 * (te - ts[:-1])
    print(tag)
    for s in range(0, k, w):
        x = dt[s:s + w]
        print(f"  steps {s:5d}..{s+w-1:5d}  t={1e3*(ts[s]-ts[0]):7.1f} ms  median {np.median(x):7.1f}  mean {x.mean():7.1f}  max {x.max():8.1f}  enqueue median {np.median(enq[s:s+w]):6.1f} us",
              flush=True)

windows("no communicator", k=1000)
if os.environ.get("COMM_DIAG_TORCH") == "1":
    import torch.distributed as dist
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    if "RANK" in os.environ:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        print("env:", {k: v for k, v in os.environ.items() if any(t in k for t in ("NCCL", "OMP", "TORCH", "HSA", "HIP", "GPU"))})
    else:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if os.environ.get("COMM_DIAG_BARRIER") == "1":
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    windows("torch process group (nccl) initialised, no communicator of ours", k=1000)
    comm = G.Comm.from_torch(0)
else:
    comm = G.Comm(0, 0, 1, lambda x: x)
plan.set_comm(comm)
windows("communicator attached (first use at t = 0)")
plan.set_comm(None)
windows("detached again", k=1000)
print("threads:", len(os.listdir("/proc/self/task")), "cpus allowed:", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max: n/a", e)
