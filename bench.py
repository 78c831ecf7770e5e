#!/usr/bin/env python
"""bench.py — Vecchia log-likelihood evaluations per second on MI355X.

Metric (BASELINE.json): "Vecchia log-likelihood evals/sec at n=1e6, m=30 (1/2/4/8 GPUs)".
One STEP = one likelihood evaluation of the whole data set: the conditioning-set
kernel over every ordered location (covariance block, factorisation, solve, fused
log-likelihood sums), the deterministic reduction, (N > 1) ONE all-reduce of the
8-double partial-sum vector over RCCL, and the scalar log-likelihood on the host.
Everything parameter-independent (locations, neighbour indices, cond flags, data) is
resident in HBM before the timed region, as in the reference where vecchia_specify()
runs once and vecchia_likelihood() once per optimiser step (R/vecchia_wrappers.R:55,72-78).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: n=1e6 uniform 2-D points (numpy default_rng(0)),
ordering='none', exact ordered 30-NN, cond.yz='z', Matern nu=1.5, covparms (1, 0.02, 1.5),
nugget 0.1, z ~ default_rng(1).standard_normal.  Rows shard contiguously over ranks
(strong scaling: the metric fixes n = 1e6).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per conditioning set, mode L (fused likelihood):
#   n0*(4 B index + 1 B cond flag) + 8*d coords + 8 nugget + 8 z  + 16 B of partial sums
# flop model of SURVEY.md §8d: p^3/3 + p^2 + p(p-1)/2 * (3d + 25)
def alg_bytes_per_set(p, d, mode):
    out = 8 * p if mode in ("U", "S") else 16
    return p * 5 + 8 * d + 8 + 8 + out


def flops_per_set(p, d):
    return p ** 3 / 3.0 + p ** 2 + 0.5 * p * (p - 1) * (3 * d + 25)


def build_workload(n, m, d, rank, world, seed=0, sgv=False, device=0):
    from gpvecchia_amd import specify as S
    rng = np.random.default_rng(seed)
    locs = rng.random((n, d))
    z = np.random.default_rng(seed + 1).standard_normal(n)
    a = (rank * n) // world
    b = ((rank + 1) * n) // world
    NN = S.find_ordered_nn_gpu(locs, m, rows=(a, b), device=device)   # this rank's rows only
    revNN = NN[:, ::-1].copy()
    if sgv:
        revCond = S.whichCondOnLatent(NN)[:, ::-1].copy()   # cond.yz='SGV' (R/vecchia_specify.R:182-183)
    else:
        revCond = np.where(revNN != 0, 0, -1).astype(np.int8)  # cond.yz='z' (R/vecchia_specify.R:189-190)
        revCond[:, -1] = 1
    return locs, z, revNN, revCond, a, b


def cpu_baseline(locs, revNN, revCond, covparms, tau, rows_sample):
    """Time the oracle's C restatement of U_NZentries (OpenMP, all host cores) on a bounded
    contiguous sample of the SAME conditioning sets; report extrapolated evals/s."""
    from oracle import r_side as R
    a, b = rows_sample
    sub = revNN[a:b]
    used = np.unique(sub[sub != 0]) - 1
    remap = np.zeros(locs.shape[0] + 1, dtype=np.int64)
    remap[used + 1] = np.arange(1, used.size + 1)
    nn2 = remap[sub]
    # pad the sample to a square problem the oracle signature expects (Nlocs rows): extra rows empty
    Nl = max(used.size, sub.shape[0])
    nnp = np.zeros((Nl, sub.shape[1]), dtype=np.int64)
    nnp[: sub.shape[0]] = nn2
    cdp = np.zeros((Nl, sub.shape[1]))
    cdp[: sub.shape[0]] = np.where(revCond[a:b] < 0, 0, revCond[a:b])
    lp = np.zeros((Nl, locs.shape[1]))
    lp[: used.size] = locs[used]
    nug = np.full(Nl, tau)
    cores = R.max_threads()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        R.U_NZentries(cores, 1, lp, nnp, cdp, nug, nug[:1], "matern", covparms)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    sets_per_s = (b - a) / t
    return dict(value=sets_per_s / locs.shape[0], unit="evals/s", cores=cores, kind="port",
                sample=f"{b - a} of {locs.shape[0]} conditioning sets (rows {a}..{b - 1}, all with n0=m+1), "
                       f"oracle/u_nzentries_oracle.c U_NZentries only, OpenMP schedule(static) on {cores} threads, "
                       f"median of 3 = {t:.3f} s, extrapolated linearly to n",
                sets_per_s=sets_per_s, seconds=t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=30)
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--mode", choices=["L", "U", "S"], default="L",
                    help="L: fused log-likelihood, cond.yz='z' (headline); U: also materialise the U entries in HBM; "
                         "S: the reference's default cond.yz='SGV' with the posterior pass (U2V) on the GPU, 1 GPU only")
    ap.add_argument("--config", choices=["C2", "C3", "C4"], default=None,
                    help="BASELINE.json parity configs (SURVEY.md §8d): C2 n=1e5 m=20 d=2; C3 n=1e6 m=30 d=2 (the default, "
                         "the configuration the metric is quoted on); C4 n=1e6 m=60 d=3 exponential")
    ap.add_argument("--nu", type=float, default=1.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="conditioning sets in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # developer hook (not the measured configuration): GPV_BENCH_BACKEND=gloo lets several ranks share one GPU so that
    # the sharding / reduction logic of N > 1 can be exercised on a single-GPU box; the collective then runs on host copies
    backend = os.environ.get("GPV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun the collective path runs even at N = 1
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def all_reduce_(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    import gpvecchia_amd as G

    if args.config == "C2":
        args.n, args.m, args.d, args.nu = 100_000, 20, 2, 1.5
    elif args.config == "C4":
        args.n, args.m, args.d, args.nu = 1_000_000, 60, 3, 0.5
    n, m, d = args.n, args.m, args.d
    p = m + 1
    rng_ = {("C2"): 0.05}.get(args.config, 0.02 if d == 2 else 0.05)
    covparms = [1.0, rng_, args.nu]
    tau = 0.1
    t_setup = time.time()
    if args.mode == "S" and world > 1:
        raise SystemExit("mode S (SGV posterior pass) does not shard: replicas only (DESIGN.md §6)")
    locs, z, revNN, revCond, a, b = build_workload(n, m, d, rank, world, sgv=(args.mode == "S"), device=local_rank)
    plan = G.Plan(locs, revNN, revCond, device=local_rank, row_begin=a, row_end=b)
    plan.set_data(z)
    if args.mode == "S":
        plan.build_posterior()
    t_setup = time.time() - t_setup

    flags = G.GPV_WANT_LOGLIK_Z | (G.GPV_WANT_U if args.mode == "U" else 0)
    if args.mode == "S":
        flags = G.GPV_WANT_DENOM
    sums = torch.zeros(G._lib.NSUMS, dtype=torch.float64, device="cuda")
    # one explicit (non-null) HIP stream carries the kernel, the all-reduce and the D2H copy of every step;
    # a NULL handle would select the plan's private stream and un-order the consumers below
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    pinned = torch.zeros(G._lib.NSUMS, dtype=torch.float64).pin_memory()

    def step():
        plan.eval("matern", covparms, tau, flags, stream=stream, d_sums_out=sums.data_ptr())
        if use_dist:
            all_reduce_(sums, dist.ReduceOp.SUM)              # the ONE collective: 64 bytes over xGMI
        # the 8 sums reach the host every step through a pinned buffer: copy on the launch stream, then wait for that
        # stream only (after an RCCL collective `sums.cpu()` costs 0.2 ms per step, as much as the kernel of one of 8 shards)
        pinned.copy_(sums, non_blocking=True)
        tstream.synchronize()
        host = pinned.numpy()
        return G.loglik_from_sums(host, n) if args.mode == "S" else G.loglik_z_from_sums(host, n)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    loglik = None
    for _ in range(args.warmup):
        loglik = step()
    kernel_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loglik = step()
        kernel_ms.append(plan.last_kernel_ms())               # hipEvent pair on the launch stream, already complete
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        all_reduce_(te, dist.ReduceOp.MAX)
        elapsed = float(te.item())
        km = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device="cuda")
        all_reduce_(km, dist.ReduceOp.MAX)
        k_ms = float(km.item())
    else:
        k_ms = float(np.mean(kernel_ms))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        rows_rank = b - a
        ab = alg_bytes_per_set(p, d, args.mode) * rows_rank
        achieved = ab / (k_ms * 1e-3) / 1e9
        fl = flops_per_set(p, d) * rows_rank / (k_ms * 1e-3) / 1e12
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tf) and n == 1_000_000 and m == 30 and d == 2 and world == 1 and args.mode == "L":
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "vecchia_loglik_evals_per_sec", "value": args.steps / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"n={n} {d}-D uniform, Matern nu={args.nu}, m={m}, cond.yz={'SGV' if args.mode == 'S' else 'z'}, mode {args.mode} "
                                   f"(BASELINE.json configs[2] geometry; rows sharded over {world} GPU(s))",
                       "n": n, "m": m, "d": d, "covparms": covparms, "nugget": tau, "mode": args.mode,
                       "sharding": f"rows/{world}", "loglik": loglik, "setup_s": round(t_setup, 2)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": f"gpv_sets_kernel<{p},{d}>", "kernel_ms": k_ms,
                         "alg_bytes_per_set": alg_bytes_per_set(p, d, args.mode),
                         "fp64_valu": {"achieved": fl, "peak": 78.6, "unit": "TFLOP/s", "frac": fl / 78.6,
                                       "flops_per_set": flops_per_set(p, d),
                                       "note": "binding roofline: FP64 VALU issue, not HBM (DESIGN.md §4)"}},
        }
        if not args.no_cpu_baseline:
            sample = args.cpu_sample
            if not sample:                      # calibrate: ~4 s of wall per repeat on all cores, 3 repeats
                cal = cpu_baseline(locs, revNN, revCond, covparms, tau, (b - 4000, b))
                sample = int(min(b - a - 2 * p, max(20000, cal["sets_per_s"] * 4.0)))
            lo = b - sample
            out["cpu_baseline"] = cpu_baseline(locs, revNN, revCond, covparms, tau, (lo, b))
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
