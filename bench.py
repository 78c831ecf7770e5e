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

With N > 1 and no RANK in the environment the parent starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything touches the
GPU) and relays rank 0's JSON line; under torchrun (RANK set) it is one of the ranks.

Workload (SURVEY.md §8d, config C3 = BASELINE.json configs[2]): n=1e6 uniform 2-D points
(numpy default_rng(0)), ordering='none', exact ordered 30-NN, cond.yz='z', Matern nu=1.5,
covparms (1, 0.02, 1.5), nugget 0.1, z ~ default_rng(1).standard_normal.  Rows shard contiguously
over ranks (strong scaling: the metric fixes n = 1e6).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TF = 78.6      # MI355X FP64 vector peak (256 CUs x 128 flop/clk x 2.4 GHz), SURVEY.md §8d
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


# algorithmic bytes per conditioning set:
#   n0*(4 B index + 1 B cond flag) + 8*d coords + 8 nugget + 8 z  + out (16 B of partial sums, or 8*n0 B of U entries)
# flop model of SURVEY.md §8d: p^3/3 + p^2 + p(p-1)/2 * (3d + 25)
def alg_bytes_per_set(p, d, mode):
    out = 8 * p if mode in ("U", "S") else 16
    return p * 5 + 8 * d + 8 + 8 + out


def flops_per_set(p, d):
    return p ** 3 / 3.0 + p ** 2 + 0.5 * p * (p - 1) * (3 * d + 25)


CONFIGS = {   # BASELINE.json configs[] index, n, m, d, nu, range
    "C2": (1, 100_000, 20, 2, 1.5, 0.05),
    "C3": (2, 1_000_000, 30, 2, 1.5, 0.02),
    "C4": (3, 1_000_000, 60, 3, 0.5, 0.05),
}


def build_workload(n, m, d, rank, world, seed=0, device=0):
    """ordering='none', cond.yz='z': this rank's rows of the exact ordered-NN arrays (GPU brute force, bit-exact)."""
    from gpvecchia_amd import specify as S
    rng = np.random.default_rng(seed)
    locs = rng.random((n, d))
    z = np.random.default_rng(seed + 1).standard_normal(n)
    a = (rank * n) // world
    b = ((rank + 1) * n) // world
    NN = S.find_ordered_nn_gpu(locs, m, rows=(a, b), device=device)   # this rank's rows only
    revNN = NN[:, ::-1].copy()
    revCond = np.where(revNN != 0, 0, -1).astype(np.int8)  # cond.yz='z' (R/vecchia_specify.R:189-190)
    revCond[:, -1] = 1
    return locs, z, revNN, revCond, a, b


def host_cpu_topology():
    """(logical CPUs this process may run on, physical cores among them) from the affinity mask and /proc/cpuinfo
    (distinct (physical id, core id) pairs); physical = logical where the file does not say."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    try:
        cpu, phys, core = None, None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu, phys, core = int(line.split(":")[1]), None, None
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
                if cpu in allowed and phys is not None:
                    cores.add((phys, core))
    except OSError:
        pass
    logical = len(allowed)
    physical = len(cores) if 0 < len(cores) <= logical else logical
    return logical, physical


def marshal_cpu_sample(locs, revNN, revCond, covparms, tau, rows_sample):
    """The oracle's arguments for the conditioning sets [a, b) of the workload, in the C function's own layout, built ONCE
    outside every timer (oracle.r_side.marshal_U_NZentries: Fortran-order float64 / int64 arrays, outputs allocated and
    touched).  Returns (marshalled arguments, seconds the marshalling took, whether it is the whole data set)."""
    from oracle import r_side as R
    t0 = time.perf_counter()
    a, b = rows_sample
    n = locs.shape[0]
    full = (a == 0 and b == n)
    if full:
        lp, nnp = locs, revNN
        cdp = np.where(revCond < 0, 0, revCond)
        Nl = n
    else:
        sub = revNN[a:b]
        used = np.unique(sub[sub != 0]) - 1
        remap = np.zeros(n + 1, dtype=np.int64)
        remap[used + 1] = np.arange(1, used.size + 1)
        Nl = max(used.size, sub.shape[0])
        nnp = np.zeros((Nl, sub.shape[1]), dtype=np.int64)
        nnp[: sub.shape[0]] = remap[sub]
        cdp = np.zeros((Nl, sub.shape[1]))
        cdp[: sub.shape[0]] = np.where(revCond[a:b] < 0, 0, revCond[a:b])
        lp = np.zeros((Nl, locs.shape[1]))
        lp[: used.size] = locs[used]
    nug = np.full(Nl, tau)
    ms = R.marshal_U_NZentries(1, lp, nnp, cdp, nug, nug[:1], "matern", covparms)
    return ms, time.perf_counter() - t0, full


def cgroup_cpu_state():
    """CPU quota and throttling counters of this process's control group (v2: cpu.max / cpu.stat; v1: cpu.cfs_quota_us /
    cpu.stat), or {} where they cannot be read: a container may see every CPU of the box in its affinity mask and still be
    held to a fraction of them by a quota."""
    out = {}
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            out["cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
            st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
            out["nr_throttled"], out["throttled_usec"] = int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read().strip()
            per = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            out["cpu_max"] = f"{'max' if q == '-1' else q} {per}"
            st = dict(l.split() for l in open("/sys/fs/cgroup/cpu/cpu.stat"))
            out["nr_throttled"], out["throttled_usec"] = int(st.get("nr_throttled", 0)), int(st.get("throttled_time", 0)) // 1000
    except (OSError, ValueError):
        pass
    return out


def time_oracle_function(ms, threads, repeats, probe=None):
    """Seconds of `repeats` calls of the C function oracle_U_NZentries ALONE ("wall-clock of the function
    only"): the ctypes function object on pre-marshalled arguments, nothing else between the two clock reads.  `probe`
    (tests) receives the callable that is timed."""
    fn, cargs = ms.fn, ms.cargs(threads)
    if probe is not None:
        probe(fn)
    times = []
    for _ in range(repeats):
        c0 = time.process_time()
        t0 = time.perf_counter()
        nf = fn(*cargs)
        times.append(time.perf_counter() - t0)
        time_oracle_function.last_cpu_s = time.process_time() - c0     # CPU seconds of all threads over the last call
    return times, int(nf)


def cpu_baseline(locs, revNN, revCond, covparms, tau, rows_sample, repeats=3, keep=None, sweep=True, probe=None):
    """Time the oracle's C restatement of U_NZentries (src/U_NZentries.cpp:37-69; OpenMP schedule(static)) on the
    conditioning sets [a, b) of the SAME workload (the whole data set when it fits the time budget).  The arguments are
    marshalled once, outside the timer; the timed callable is the C function.  `thread_sweep` repeats the call on {1, 8, 16,
    32, 64, physical cores, all logical CPUs} (1 thread: on a bounded row sample), each entry with the CPU seconds it was
    really granted; `value` quotes the FASTEST entry (a container may see every CPU of the box and be held to a few by its
    quota), `all_logical` the reference's own choice (Ncores = detectCores(logical=TRUE), R/U_sparsity.R:76).  keep: a dict
    that receives the oracle's U entries of those sets (`Lentries`, rows a..b-1) for the parity_in_run block — the
    checker's output is compared with the GPU's, never fed back into it."""
    a, b = rows_sample
    n = locs.shape[0]
    logical, physical = host_cpu_topology()
    ms, marshal_s, full = marshal_cpu_sample(locs, revNN, revCond, covparms, tau, rows_sample)
    cg0 = cgroup_cpu_state()
    times, nfail = time_oracle_function(ms, logical, repeats, probe)
    cpu_last = time_oracle_function.last_cpu_s / max(times[-1], 1e-9)        # CPU seconds per wall second of the last call
    if keep is not None:
        keep["Lentries"] = ms.L[: b - a]
        keep["n_failed"] = nfail
        keep["rows"] = (a, b)
    t_all = float(np.median(times))

    def entry(threads, rows, secs, eff):
        # effective_cores = CPU seconds the call consumed / its wall time: what the host actually granted the threads
        return {"threads": threads, "rows": rows, "seconds": secs, "sets_per_s": rows / secs, "effective_cores": round(eff, 1)}
    sw = [entry(logical, b - a, t_all, cpu_last)]
    if sweep:
        for th in sorted({physical, 64, 32, 16, 8} - {logical, 1}, reverse=True):
            if th > logical:
                continue
            tp, _ = time_oracle_function(ms, th, 2 if th != physical else repeats, probe)
            sw.append(entry(th, b - a, float(np.median(tp)), time_oracle_function.last_cpu_s / max(tp[-1], 1e-9)))
        # one thread: a bounded row sample (full rows, n0 = m + 1) — the whole data set would take ~10-20 s per repeat
        r1 = min(b - a, 60000)
        if r1 == b - a:
            m1 = ms
        else:
            m1, _, _ = marshal_cpu_sample(locs, revNN, revCond, covparms, tau, (b - r1, b))
        t1, _ = time_oracle_function(m1, 1, 2, probe)
        sw.append(entry(1, r1, float(np.median(t1)), time_oracle_function.last_cpu_s / max(t1[-1], 1e-9)))
        sw.sort(key=lambda e: e["threads"])
    # `value` is the FASTEST configuration of the sweep over the whole sample (the strongest baseline this host offers): a
    # container may show every CPU of the box and still be held to a few of them by a quota, and then all logical CPUs — the
    # reference's own choice, Ncores = detectCores(logical=TRUE), R/U_sparsity.R:76 — is not the fastest way to run it
    full_rows = [e for e in sw if e["rows"] == b - a]
    best = max(full_rows, key=lambda e: e["sets_per_s"])
    t, sets_per_s, used = best["seconds"], best["sets_per_s"], best["threads"]
    cg1 = cgroup_cpu_state()
    cgroup = dict(cg1)
    if "throttled_usec" in cg0 and "throttled_usec" in cg1:
        cgroup["throttled_usec_during_baseline"] = cg1["throttled_usec"] - cg0["throttled_usec"]
        cgroup["nr_throttled_during_baseline"] = cg1["nr_throttled"] - cg0["nr_throttled"]
    quota = None
    try:
        q, per = cgroup.get("cpu_max", "max 1").split()
        quota = None if q == "max" else float(q) / float(per)
    except ValueError:
        pass
    what = ("oracle/u_nzentries_oracle.c oracle_U_NZentries, the C function ALONE on pre-marshalled arguments (ctypes call "
            f"between two clock reads; marshalling {marshal_s:.2f} s reported apart as marshal_s)")
    host = (f"{used} threads — the fastest of the thread sweep; the host shows {logical} logical CPUs on {physical} physical cores"
            + (f", the container's CPU quota is {quota:g} cores (cgroup cpu.max)" if quota else "")
            + f"; on all {logical} logical CPUs: {t_all:.3f} s")
    if full:
        sample = (f"all {n} conditioning sets (no extrapolation), {what}, OpenMP schedule(static) on {host}; median = {t:.3f} s")
    else:
        sample = (f"{b - a} of {n} conditioning sets (rows {a}..{b - 1}, all with n0=m+1), {what}, OpenMP schedule(static) on "
                  f"{host}; median = {t:.3f} s, EXTRAPOLATED linearly to n")
    return dict(value=sets_per_s / n, unit="evals/s", cores=used, logical_cpus=logical, physical_cores=physical,
                quota_cores=quota, all_logical={"threads": logical, "seconds": t_all, "sets_per_s": (b - a) / t_all},
                kind="port", sample=sample,
                extrapolated=not full, sets_per_s=sets_per_s, seconds=t, marshal_s=marshal_s, thread_sweep=sw,
                cgroup=cgroup, loadavg=[round(x, 1) for x in os.getloadavg()],
                timed_callable="ctypes oracle_U_NZentries",
                note="own C restatement of src/U_NZentries.cpp:39-69 without Armadillo's per-iteration temporaries: "
                     "FASTER than the real reference (BASELINE.md §2), which cannot be built on this box")


def parity_in_run(gpu_L, gpu_loglik, gpu_nfail, kept, revNN, z, tau, n):
    """The bench run's own parity evidence (outside every timed region): the U entries the GPU wrote for this workload
    (one GPV_WANT_U evaluation) against the oracle's — the rows the cpu_baseline leg computed anyway — row by row,
    normwise, and the log-likelihood against the oracle's (closed form of R/vecchia_likelihood.R:63-99 for cond.yz='z')."""
    from oracle import r_side as R
    a, b = kept["rows"]
    ref = kept["Lentries"]
    out = gpu_L[a:b]
    err = np.abs(out - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-300)
    res = {"rows_compared": int(b - a), "max_row_err": float(err.max()), "median_row_err": float(np.median(err)),
           "n_failed": int((~(err <= 1e-8)).sum()), "tol": 1e-8,
           "zero_pattern_equal": bool(np.array_equal(out == 0, ref == 0)),
           "chol_failures": {"hip": int(gpu_nfail), "oracle": int(kept["n_failed"])},
           "what": "U entries (Lentries) of the HIP path vs oracle/u_nzentries_oracle.c on the same inputs, per row "
                   "max|dM|/max|M|; n_failed = rows beyond 1e-8; the oracle is a restatement (parity unpinned, DESIGN.md §3)"}
    if a == 0 and b == n:
        ll_o, _ = R.separable_sums_condz_vectorised(revNN, ref, z, tau)
        res.update(loglik_oracle=float(ll_o), loglik_hip=float(gpu_loglik),
                   loglik_rel_err=float(abs(gpu_loglik - ll_o) / abs(ll_o)))
    return res


def mode_S_oracle(va, z, covparms, tau, gpu_loglik, gpu_sums, gpu_mu_ord):
    """Parity evidence for secondary.mode_S (outside every timed region): the oracle's SPARSE restatement of the R chain
    createU -> U2V -> vecchia_likelihood_U / vecchia_mean (oracle/r_side.py: its own U entries, Matrix::tcrossprod by
    scipy.sparse, its own natural-order sparse Cholesky) on the plan mode S was timed on, against what the GPU returned."""
    from oracle import r_side as R
    t0 = time.time()
    prep = dict(va["U_prep"])
    nn = prep["revNNarray"]
    prep["revNNarray"] = np.where(nn == 0, np.nan, nn.astype(np.float64))
    prep["revCond"] = np.where(prep["revCond"] < 0, np.nan, prep["revCond"].astype(np.float64))
    ova = {k: v for k, v in va.items() if not isinstance(k, tuple)}
    ova["U_prep"] = prep
    Us = R.createU_sparse(ova, covparms, tau)
    V = R.U2V_sparse(Us)
    ll_o, t = R.vecchia_likelihood_U_sparse(z, Us, V=V, terms=True)
    res = {"loglik_oracle": float(ll_o), "rel_err": float(abs(gpu_loglik - ll_o) / abs(ll_o)),
           "logdet_denom_rel_err": float(abs(gpu_sums[2] + t["logdet_denom"]) / abs(t["logdet_denom"])),
           "quadform_denom_rel_err": float(abs(gpu_sums[3] - t["quadform_denom"]) / abs(t["quadform_denom"])),
           "nnz_V": int(V.nnz)}
    if gpu_mu_ord is not None:
        mu_o = R.vecchia_mean_sparse(z, Us, V, ordered=True)
        res["mean_max_abs_err"] = float(np.abs(gpu_mu_ord - mu_o).max())
        res["mean_abs_max"] = float(np.abs(mu_o).max())
    res["oracle_s"] = round(time.time() - t0, 1)
    res["what"] = ("HIP set kernel + posterior pass vs oracle.r_side.{createU_sparse,U2V_sparse,vecchia_likelihood_U_sparse,"
                   "vecchia_mean_sparse} (R/vecchia_prediction.R:62-142, R/vecchia_likelihood.R:63-99) on the same plan")
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def kernel_code_sha256():
    """Hash of the conditioning-set kernel's CODE (gpv_sets_kernel.hpp with comments and whitespace removed): the PMC traffic
    figure in profiles/ is quoted only while it was measured on the code in this tree."""
    import re
    src = open(os.path.join(ROOT, "gpvecchia_amd", "csrc", "gpv_sets_kernel.hpp"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return hashlib.sha256("".join(src.split()).encode()).hexdigest()


def posterior_code_sha256():
    """Hash of the posterior pass's CODE (gpv_posterior.hip, comments and whitespace removed) and of the schedule builder's
    (gpv_api.hip build_posterior_impl lives in gpv_api.hip: the whole file): offline per-level counters under profiles/ are
    quoted only while both are the code they were measured on."""
    import re
    h = hashlib.sha256()
    for f in ("gpv_posterior.hip", "gpv_posterior_ext.h"):
        src = open(os.path.join(ROOT, "gpvecchia_amd", "csrc", f), encoding="utf-8").read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        h.update("".join(src.split()).encode())
    return h.hexdigest()


def count_gpus_sysfs():
    """GPUs of this box counted from the KFD topology in sysfs (nodes with simd_count > 0), honouring the
    HIP/ROCR_VISIBLE_DEVICES lists: the launching parent never opens the driver, not even to count devices."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as fresh processes.  Nothing in this process touches the GPU
    or the driver (the GPUs are counted in sysfs), and nothing is exec'ed."""
    backend = os.environ.get("GPV_BENCH_BACKEND", "nccl")
    ndev = count_gpus_sysfs()
    if backend == "nccl" and ndev < args.gpus:
        print(f"[bench] --gpus {args.gpus} but only {ndev} GPU(s) visible: refusing to report a {args.gpus}-GPU number",
              file=sys.stderr)
        return 2
    # one fresh child per rank with the launcher contract's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), started
    # here rather than through torch.distributed.run so that a rank's exit code (3 = self-check failed, 4 = the collective
    # guard expired) reaches the caller unchanged; when a rank fails, the others are ended by their exact PIDs
    port = str(_free_port())
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    base.update(WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for pr in list(alive):
            code = pr.poll()
            if code is None:
                continue
            alive.remove(pr)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                for other in alive:                       # a failed rank ends the job: the rest would wait in a collective
                    other.terminate()
                t_end = time.time() + 10.0
                for other in alive:
                    try:
                        other.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        other.kill()
                        other.wait()
                alive = []
                break
    return rc


def other_config(name, device, measure_with, steps=10, b2b=None):
    """BASELINE.json configs[1] / configs[3] on one GPU, mode L: evals/s, set-kernel ms, FP64 fraction."""
    import gpvecchia_amd as G
    ci, n, m, d, nu, rng_ = CONFIGS[name]
    t0 = time.time()
    locs, z, revNN, revCond, a, b = build_workload(n, m, d, 0, 1, device=device)
    plan = G.Plan(locs, revNN, revCond, device=device)
    plan.set_data(z)
    ts = time.time() - t0
    cp = [1.0, rng_, nu]
    el, km, ll = measure_with(plan, G.GPV_WANT_LOGLIK_Z, False, steps, 2, cp, 0.1, n)
    kb = b2b(plan, G.GPV_WANT_LOGLIK_Z, cp, 0.1) if b2b is not None else None
    kuse = kb if (kb is not None and km < 0.5) else km       # short launches: the event pair itself is 10 % of the reading
    tf = flops_per_set(m + 1, d) * n / (kuse * 1e-3) / 1e12
    del plan
    return {"value": steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / steps, "kernel_ms": km,
            "kernel_ms_back_to_back": kb, "fp64_frac_from": "kernel_ms_back_to_back" if kuse is kb else "kernel_ms",
            "kernel": f"gpv_sets_kernel<{m + 1},{d}>", "fp64_frac": tf / FP64_PEAK_TF, "fp64_tflops": tf, "loglik": ll,
            "setup_s": round(ts, 2),
            "what": f"BASELINE.json configs[{ci}]: n={n} {d}-D, m={m}, Matern nu={nu}, range {rng_}, cond.yz='z', mode L, 1 GPU"}


def vl_oracle(z, va, cp, post, ll):
    """Parity evidence for secondary.C5_vl (outside every timed region): the Newton LOOP of R/vecchia_laplace_NR.R:88-130 and
    vecchia_laplace_likelihood (:361-416) restated by the oracle on sparse matrices (oracle.r_side.calculate_posterior_VL_sparse:
    createU_sparse -> U2V_sparse -> vecchia_mean_sparse per step), on the same data and plan, against what the GPU returned."""
    from oracle import r_side as R
    t0 = time.time()
    prep = dict(va["U_prep"])
    nn = prep["revNNarray"]
    prep["revNNarray"] = np.where(nn == 0, np.nan, nn.astype(np.float64))
    prep["revCond"] = np.where(prep["revCond"] < 0, np.nan, prep["revCond"].astype(np.float64))
    ova = {k: v for k, v in va.items() if not isinstance(k, tuple)}
    ova["U_prep"] = prep
    tr = []
    ref = R.calculate_posterior_VL_sparse(z, ova, "poisson", cp, trace=tr, snapshot_convg=1e-5)
    ll_ref = R.vecchia_laplace_likelihood_sparse(z, ova, "poisson", cp, post=ref["snapshot"])
    return {"iters_hip": int(post["iter"]), "iters_oracle": int(ref["iter"]), "converged_oracle": bool(ref["cnvgd"]),
            "mean_max_abs_err": float(np.abs(post["mean"] - ref["mean"]).max()), "mean_abs_max": float(np.abs(ref["mean"]).max()),
            "loglik_oracle": float(ll_ref), "loglik_rel_err": float(abs(ll - ll_ref) / abs(ll_ref)),
            "oracle_newton_trace": tr, "oracle_s": round(time.time() - t0, 1),
            "what": "G.calculate_posterior_VL / G.vecchia_laplace_likelihood (device Newton loop) vs "
                    "oracle.r_side.{calculate_posterior_VL_sparse,vecchia_laplace_likelihood_sparse} on the same z and plan"}


def vl_config(device, n=500_000, m=30, parity=True):
    """BASELINE.json configs[4]: Vecchia-Laplace, Poisson data, maxmin + SGV, one GPU (the posterior pass does not shard)."""
    import gpvecchia_amd as G
    rng = np.random.default_rng(0)
    locs = rng.random((n, 2))
    y = 0.8 * np.sin(5 * locs[:, 0]) * np.cos(4 * locs[:, 1]) + 0.3       # a cheap smooth latent field
    z = rng.poisson(np.exp(y)).astype(float)
    cp = [1.0, 0.03, 1.5]
    t0 = time.time()
    va = G.vecchia_specify(locs, m, nn_backend="gpu")                      # defaults: maxmin, SGV
    t_spec = time.time() - t0
    t0 = time.time()
    post = G.calculate_posterior_VL(z, va, "poisson", cp, device=device)   # includes plan upload + posterior structure
    t_first = time.time() - t0
    t_nr = []
    for _ in range(3):
        t0 = time.time()
        post = G.calculate_posterior_VL(z, va, "poisson", cp, device=device)
        t_nr.append(time.time() - t0)
    t_nr = float(np.median(t_nr))
    t_ll = []
    for _ in range(3):
        t0 = time.time()
        ll = G.vecchia_laplace_likelihood(z, va, "poisson", cp, device=device)
        t_ll.append(time.time() - t0)
    par = None
    if parity:
        try:
            par = vl_oracle(z, va, cp, post, ll)
        except Exception as e:                                # noqa: BLE001
            par = {"error": repr(e)}
    return {"parity_in_run": par,
            "ms_per_nr_iter": 1e3 * t_nr / max(post["iter"], 1), "nr_iters": int(post["iter"]), "converged": bool(post["cnvgd"]),
            "nr_loop_s": t_nr, "vecchia_laplace_likelihood_s": float(np.median(t_ll)), "loglik": ll,
            "rmse_latent": float(np.sqrt(np.mean((post["mean"] - y) ** 2))), "specify_s": round(t_spec, 2),
            "first_call_s": round(t_first, 2),
            "what": f"BASELINE.json configs[4]: n={n} 2-D, Poisson, m={m}, ordering='maxmin', cond.yz='SGV'; whole "
                    "calculate_posterior_VL call divided by its Newton steps; 1 GPU (replicas only at N > 1)"}


def dropin_config(n, locs, revNN, revCond, covparms, tau):
    """SURVEY.md §8d mode U+D2H: the literal drop-in gpv_U_NZentries with host buffers in, Lentries/Zentries to host buffers
    out (PCIe inclusive).  The first call builds and caches the device plan; the timed calls are what createU pays per
    optimiser step."""
    import gpvecchia_amd as G
    from gpvecchia_amd import _lib as L
    lf = np.asfortranarray(locs)
    nn = np.asfortranarray(revNN.astype(np.int32))
    cd = np.asfortranarray(np.where(revCond < 0, L.NA_INTEGER, revCond).astype(np.int32))
    nug = np.full(n, tau)
    cp = np.ascontiguousarray(covparms, dtype=np.float64)
    p = nn.shape[1]
    Lent = np.empty((n, p), order="F")
    Z = np.empty(2 * n)
    import ctypes as C
    ci = lambda v: C.byref(C.c_int(int(v)))
    nfail, status, ct = C.c_int(0), C.c_int(0), C.c_char_p(b"matern")
    t = []
    for it in range(5):                               # the C symbol itself, as R's .C() would call it
        t0 = time.perf_counter()
        L.lib().gpv_U_NZentries(ci(1), ci(n), ci(n), ci(lf.shape[1]), ci(p), L.dptr(lf), L.iptr(nn), L.iptr(cd), L.dptr(nug),
                                L.dptr(nug), C.byref(ct), L.dptr(cp), ci(cp.size), L.dptr(Lent), L.dptr(Z), C.byref(nfail),
                                C.byref(status))
        t.append(time.perf_counter() - t0)
        L.check(status.value, "gpv_U_NZentries")
    dk_ok = bool(np.all(Lent[np.arange(0, n, 9973), (nn[::9973] != 0).sum(axis=1) - 1] > 0))
    tf = []
    for it in range(3):                               # outputs allocated anew per call, as R's .C() does: first-touch page faults
        Lf, Zf = np.empty((n, p), order="F"), np.empty(2 * n)
        t0 = time.perf_counter()
        L.lib().gpv_U_NZentries(ci(1), ci(n), ci(n), ci(lf.shape[1]), ci(p), L.dptr(lf), L.iptr(nn), L.iptr(cd), L.dptr(nug),
                                L.dptr(nug), C.byref(ct), L.dptr(cp), ci(cp.size), L.dptr(Lf), L.dptr(Zf), C.byref(nfail),
                                C.byref(status))
        tf.append(time.perf_counter() - t0)
        del Lf, Zf
    L.lib().gpv_plan_cache_clear()
    return {"first_call_ms": 1e3 * t[0], "ms_per_call": 1e3 * float(np.median(t[1:])),
            "ms_per_call_fresh_outputs": 1e3 * float(np.median(tf)), "n_failed": int(nfail.value),
            "diag_positive_on_sample": dk_ok, "bytes_to_host": int(8 * n * p + 16 * n),
            "what": "mode U+D2H: the C symbol gpv_U_NZentries at C3 with host buffers in and out (caller-allocated outputs), plan "
                    "cached from the first call; includes content hash, nuggets H2D, kernel, transpose and the D2H copy"}


def per_rank_step(args, emulate=8, steps=200):
    """What ONE rank pays per step at the per-rank load of an `emulate`-GPU run: rows = n / emulate on this GPU, through the
    RCCL path (a child started with torch.distributed.run, world 1, backend nccl).  Emulated load, NOT a scaling number:
    a real all-reduce adds its xGMI round."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "20",
           "--emulate-world", str(emulate), "--no-cpu-baseline", "--no-secondary", "--config", args.config,
           "--clock-warmup-s", str(args.clock_warmup_s)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        raise RuntimeError(f"child failed ({r.returncode}): {r.stderr[-600:]}")
    j = json.loads(line[-1])
    k_ms = j["roofline"]["kernel_ms"]
    return {"ms_per_step": j["ms_per_step"], "kernel_ms": k_ms, "overhead_us": 1e3 * (j["ms_per_step"] - k_ms),
            "kernel_plus_allreduce_ms_back_to_back": j["roofline"].get("kernel_ms_back_to_back"),
            "rows": j["roofline"]["sets_per_launch"], "emulated_world": emulate, "backend": "nccl (RCCL), world 1",
            "what": "emulated per-rank load, not a scaling number: one GPU takes rank 0's shard of an 8-rank job and runs the "
                    "step of the N > 1 path (kernel with fused reduction, the library's own RCCL all-reduce of 8 doubles on "
                    "the same stream, 64-thread kernel that stores them into pinned memory behind a sequence number the host "
                    "spins on); overhead_us = ms_per_step - kernel_ms is the fixed cost strong scaling pays per step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--m", type=int, default=None)
    ap.add_argument("--d", type=int, default=None)
    ap.add_argument("--mode", choices=["L", "U", "S"], default="L",
                    help="L: fused log-likelihood, cond.yz='z' (headline); U: also materialise the U entries in HBM; "
                         "S: the reference's defaults, ordering='maxmin' + cond.yz='SGV', posterior pass (U2V) on the GPU, "
                         "1 GPU only")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3",
                    help="BASELINE.json configs (SURVEY.md §8d): C2 n=1e5 m=20 d=2; C3 n=1e6 m=30 d=2 (default, the "
                         "configuration the metric is quoted on); C4 n=1e6 m=60 d=3 exponential")
    ap.add_argument("--nu", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the mode U / mode S secondary measurements")
    ap.add_argument("--no-vl-parity", action="store_true",
                    help="skip secondary.C5_vl.parity_in_run (the oracle's sparse Vecchia-Laplace loop at n = 5e5: ~1.5 min of host time)")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0,
                    help="per-repeat wall budget of the CPU baseline; the whole data set is timed when it fits")
    ap.add_argument("--clock-warmup-s", type=float, default=0.1,
                    help="seconds of untimed evaluations before the W warm-up steps of every measurement, so that the timed steps "
                         "run at the steady shader clock (0 = none; reported in the JSON line)")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="developer/secondary measurement: this rank takes the row shard rank 0 of a job of that many ranks "
                         "would own (rows = n / E) while the collectives run over the real world; NOT a scaling number")
    ap.add_argument("--self-check", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--no-self-check", action="store_true",
                    help="N > 1: skip rank 0's evaluation of the unsharded plan (by default it runs after the timed region and "
                         "the N-rank log-likelihood must equal it to 1e-12; exit 3 otherwise)")
    ap.add_argument("--comm-guard-s", type=float, default=300.0,
                    help="N > 1: wall-clock limit for communicator creation + the first all-reduced evaluation (exit 4 beyond it)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # developer hook (not the measured configuration): GPV_BENCH_BACKEND=gloo lets several ranks share one GPU so that
    # the sharding / reduction logic of N > 1 can be exercised on a single-GPU box; the collective then runs on host copies
    backend = os.environ.get("GPV_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"[bench] {world} ranks but {ndev} GPU(s) visible")
    if backend != "nccl":
        local_rank = local_rank % ndev
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

    # the collective of the step belongs to the library (gpv_comm: RCCL bound at run time, include/gpvecchia.h); torch.distributed
    # is the launcher, the rendezvous that carries the communicator's id, and the barrier / max-over-ranks of the timing.
    # GPV_TORCH_ALLREDUCE=1: the round-2 route (dist.all_reduce on the plan's device buffer), kept for A/B
    comm, route = None, ("none (1 rank, no launcher)" if not use_dist else f"torch.distributed all_reduce ({backend})")
    guard = None

    def arm_guard():
        # Wall-clock guard around the first contact with N ranks: creating the communicator and the first all-reduced
        # evaluation must finish within --comm-guard-s, else this rank says why and exits non-zero (the launcher then ends
        # the others): a hung collective must not look like a slow bench.  Disarmed after the first complete step.
        import threading

        def _expired():
            print(f"[bench] rank {rank}: no complete all-reduced evaluation within {args.comm_guard_s:.0f} s of starting the "
                  f"communicator (route so far: {route}); exiting 4.  GPV_TORCH_ALLREDUCE=1 selects the torch.distributed route.",
                  file=sys.stderr, flush=True)
            os._exit(4)
        g = threading.Timer(args.comm_guard_s, _expired)
        g.daemon = True
        g.start()
        return g

    if use_dist and backend == "nccl" and args.mode != "S":
        guard = arm_guard()
        from gpvecchia_amd.distributed import negotiate_comm
        comm, why = negotiate_comm(local_rank, None, timeout_s=min(120.0, args.comm_guard_s / 2),
                                   log=lambda m: print(f"[bench] rank {rank}: {m}", file=sys.stderr, flush=True))
        route = ("library-owned RCCL communicator (gpv_comm), all-reduce enqueued by gpv_plan_eval; " + why) if comm is not None \
            else f"torch.distributed all_reduce (nccl); gpv_comm not used: {why}"

    ci, n, m, d, nu, rng_ = CONFIGS[args.config]
    custom = any(v is not None for v in (args.n, args.m, args.d, args.nu))
    n = args.n or n
    m = args.m or m
    d = args.d or d
    nu = args.nu or nu
    if args.d is not None:
        rng_ = 0.02 if d == 2 else 0.05
    p = m + 1
    covparms = [1.0, rng_, nu]
    tau = 0.1
    if args.mode == "S" and world > 1:
        raise SystemExit("mode S (SGV posterior pass) does not shard: replicas only (DESIGN.md §6)")

    # one explicit (non-null) HIP stream carries the kernel, the all-reduce and the D2H copy of every step;
    # a NULL handle would select the plan's private stream and un-order the consumers below
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    sums = torch.zeros(G._lib.NSUMS, dtype=torch.float64, device="cuda")
    pinned = torch.zeros(G._lib.NSUMS, dtype=torch.float64).pin_memory()
    done = torch.cuda.Event()

    last_sums = np.zeros(G._lib.NSUMS)                    # the totals of the latest step (N > 1: after the all-reduce)
    first_contact_done = [False]

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(plan, flags, denom, steps, warmup):
        return measure_with(plan, flags, denom, steps, warmup, covparms, tau, n)

    def measure_with(plan, flags, denom, steps, warmup, covparms, tau, n, clock_warmup_s=None):
        """W untimed + K timed evaluations of `plan`; returns (seconds, mean set-kernel ms, loglik)."""
        cw_s = args.clock_warmup_s if clock_warmup_s is None else clock_warmup_s
        def step():
            if comm is not None:
                # N > 1 (one process per GPU): the library enqueues kernel, RCCL all-reduce of the 8 sums (64 bytes over xGMI)
                # and the 64-thread kernel that stores them into pinned host memory on this one stream; sums() spins on the
                # sequence number behind them
                plan.eval("matern", covparms, tau, flags, stream=stream)
                host = plan.sums()
                last_sums[:] = host
                return G.loglik_z_from_sums(host, n)
            if use_dist:
                plan.eval("matern", covparms, tau, flags, stream=stream, d_sums_out=sums.data_ptr())
                all_reduce_(sums, dist.ReduceOp.SUM)              # the ONE collective: 64 bytes over xGMI
                # the 8 sums reach the host through a pinned buffer: copy on the launch stream, then poll that stream's event
                pinned.copy_(sums, non_blocking=True)
                done.record(tstream)
                while not done.query():                           # spin on the event's flag: no interrupt-driven wake-up
                    pass
                host = pinned.numpy()
            else:
                # one GPU: exactly what vecchia_likelihood() does.  The kernel that totals the sums stores them into the plan's
                # pinned host buffer followed by the evaluation's sequence number; sums() spins on that number (posterior pass:
                # waits for the stream)
                plan.eval("matern", covparms, tau, flags, stream=stream)
                host = plan.sums()
            last_sums[:] = host
            return G.loglik_from_sums(host, n) if denom else G.loglik_z_from_sums(host, n)
        nonlocal guard
        ll = None
        if guard is None and world > 1 and not first_contact_done[0]:
            guard = arm_guard()                                   # every multi-rank backend: the first all-reduced evaluation
        if guard is not None:
            stall = os.environ.get("GPV_BENCH_STALL_RANK")        # developer switch (tests): this rank sleeps before its
            if stall is not None and int(stall) == rank:          # first evaluation, the others wait in the collective
                time.sleep(float(os.environ.get("GPV_BENCH_STALL_S", "30")))
            step()                                                # first contact: the first all-reduced evaluation of the job
            guard.cancel()
            guard = None
            first_contact_done[0] = True
        # like timeit: no cyclic-GC pass inside the timed region (with torch imported a full collection walks millions of
        # objects: one 55 ms pause was seen in a 70 ms region of 0.18 ms steps; tools/comm_diag.py).  Collected HERE, before the
        # clock warm-up: a collection between the warm-up and the timed steps idles the GPU for ~40 ms and the clock is down again
        import gc
        gc.collect()
        gc.disable()
        try:
            return _measure_gc_off(step, plan, steps, warmup, cw_s, ll)
        finally:
            gc.enable()                                           # also when a step raises

    def _measure_gc_off(step, plan, steps, warmup, cw_s, ll):
        # clock warm-up (untimed, before the W warm-up steps): the GPU's power management drops the shader clock within
        # milliseconds of idling and takes ~35 ms of continuous work to bring it back (tools/clock_ramp.py,
        # profiles/archive/r03_clock_ramp.txt: 1.46 -> 1.26 ms per launch over the first 25 launches at this workload); a timed region of 20
        # steps behind 5 warm-up steps would measure the ramp, not the rate an optimiser loop sees
        if cw_s > 0:
            t_w = time.perf_counter()
            for _ in range(3):
                step()
            cnt = int(min(5000.0, cw_s / max((time.perf_counter() - t_w) / 3, 1e-5))) + 1
            if use_dist:                                          # every rank runs the same number of collectives
                tc = torch.tensor([cnt], dtype=torch.int64, device="cuda")
                all_reduce_(tc, dist.ReduceOp.MAX)
                cnt = int(tc.item())
            for _ in range(cnt):
                step()
        for _ in range(warmup):
            ll = step()
        kms = []
        fence()
        t0 = time.perf_counter()
        trace = [] if os.environ.get("GPV_BENCH_TRACE") else None
        for _ in range(steps):
            ll = step()
            if plan.kernel_timing:
                kms.append(plan.last_kernel_ms())                # hipEvent pair on the launch stream, already complete
            if trace is not None:
                trace.append(time.perf_counter())
        if trace:
            dt = 1e6 * np.diff(np.array([t0] + trace))
            print(f"[bench trace] steps {steps}: median {np.median(dt):.1f} us, mean {dt.mean():.1f}, max {dt.max():.1f}, "
                  f"over 2x median: {int((dt > 2 * np.median(dt)).sum())}; first 5: {np.round(dt[:5], 1)}", file=sys.stderr)
        fence()
        el = time.perf_counter() - t0
        return el, (float(np.mean(kms)) if kms else float("nan")), ll

    def kernel_ms_back_to_back(plan, flags, covparms, tau, launches=50):
        """Mean duration of `launches` evaluations enqueued back to back between ONE event pair on the launch stream (no host
        wait in between; the totals of the last one are awaited).  For launches of ~0.1 ms the per-launch hipEvent pair of
        `last_kernel_ms` adds 7-13 us of its own (two queue packets around the kernel; rocprofv3's kernel trace of the same
        command shows the kernel itself: profiles/r04_C2_kernel_launches.json 90.5 us where the event pair reads 103.7); this
        figure includes the ~1-2 us between two dependent launches instead."""
        was = plan.kernel_timing
        plan.set_kernel_timing(False)
        for _ in range(5):
            plan.eval("matern", covparms, tau, flags, stream=stream)
        plan.sums()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(tstream)
        for _ in range(launches):
            plan.eval("matern", covparms, tau, flags, stream=stream)
        e1.record(tstream)
        plan.sums()
        e1.synchronize()
        plan.set_kernel_timing(was)
        return e0.elapsed_time(e1) / launches

    def roofline(k_ms, rows_rank, mode, traffic=None):
        ab = alg_bytes_per_set(p, d, mode) * rows_rank
        gbs = ab / (k_ms * 1e-3) / 1e9
        tf = flops_per_set(p, d) * rows_rank / (k_ms * 1e-3) / 1e12
        return {"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TF,
                "traffic": traffic, "kernel": f"gpv_sets_kernel<{p},{d}>", "kernel_ms": k_ms, "kernel_ms_from": timing_note,
                "flops_per_set": flops_per_set(p, d), "sets_per_launch": rows_rank,
                "note": "binding roofline is FP64 VALU issue, not HBM or MFMA (blocks are (m+1)x(m+1); DESIGN.md §4); "
                        "flop model of SURVEY.md §8d",
                "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "alg_bytes_per_set": alg_bytes_per_set(p, d, mode)}}

    t_setup = time.time()
    if args.mode == "S":
        rng = np.random.default_rng(0)
        locs = rng.random((n, d))
        z = np.random.default_rng(1).standard_normal(n)
        va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="SGV", nn_backend="gpu")
        plan = G.Plan(va["locsord"], va["U_prep"]["revNNarray"], va["U_prep"]["revCond"], device=local_rank)
        plan.set_data(z[va["ord_z"] - 1])
        plan.build_posterior()
        a, b = 0, n
        flags, denom = G.GPV_WANT_DENOM, True
    else:
        if args.emulate_world > 1:
            if world != 1:
                raise SystemExit("--emulate-world is a one-rank measurement")
            locs, z, revNN, revCond, a, b = build_workload(n, m, d, 0, args.emulate_world, device=local_rank)
        else:
            locs, z, revNN, revCond, a, b = build_workload(n, m, d, rank, world, device=local_rank)
            corrupt = os.environ.get("GPV_BENCH_CORRUPT", "")     # developer switch (tests of the self-check): "data:R" makes
            if corrupt:                                           # rank R evaluate a wrong datum, "rows:R" makes it drop a row
                kind, r_bad = corrupt.split(":")
                if int(r_bad) == rank and kind == "data":
                    z = z.copy()
                    z[a] += 1.0
                elif int(r_bad) == rank and kind == "rows":
                    b -= 1                                        # (the index arrays keep their Nlocs rows: the plan reads rows a..b-1)
        plan = G.Plan(locs, revNN, revCond, device=local_rank, row_begin=a, row_end=b)
        plan.set_data(z)
        if comm is not None:
            plan.set_comm(comm)
        flags, denom = G.GPV_WANT_LOGLIK_Z | (G.GPV_WANT_U if args.mode == "U" else 0), False
    t_setup = time.time() - t_setup

    timing_note = "hipEvent pair around every launch of the timed region, on the launch stream"
    from_idle = None
    k_b2b = None
    if use_dist:
        # at a fraction of a millisecond per step the event pair itself costs 7-10 us (two queue packets per launch): the K
        # timed steps run without it, the kernel's duration comes from an instrumented repeat of the same K steps
        plan.set_kernel_timing(False)
        elapsed, _, loglik = measure(plan, flags, denom, args.steps, args.warmup)
        plan.set_kernel_timing(True)
        _, k_ms, _ = measure(plan, flags, denom, args.steps, 1)
        timing_note = "hipEvent pair around every launch of an instrumented repeat of the K timed steps (events off while timing)"
        k_b2b = kernel_ms_back_to_back(plan, flags, covparms, tau) if (world == 1 and args.mode != "S") else None
    else:
        # first, the protocol of rounds 1 and 2 for the record: W warm-up + K timed steps straight from an idle GPU (no clock
        # warm-up): what a caller sees who evaluates a handful of times and stops; reported as config.from_idle, never as value
        if args.clock_warmup_s > 0 and world == 1:
            time.sleep(0.05)
            el0, km0, _ = measure_with(plan, flags, denom, args.steps, args.warmup, covparms, tau, n, clock_warmup_s=0.0)
            from_idle = {"value": args.steps / el0, "ms_per_step": 1e3 * el0 / args.steps, "kernel_ms": km0,
                         "what": "the same W + K steps started on an idle GPU (shader clock still ramping, DESIGN.md §5); not `value`"}
        elapsed, k_ms, loglik = measure(plan, flags, denom, args.steps, args.warmup)
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        all_reduce_(te, dist.ReduceOp.MAX)
        elapsed = float(te.item())
        km = torch.tensor([k_ms], dtype=torch.float64, device="cuda")
        all_reduce_(km, dist.ReduceOp.MAX)
        k_ms = float(km.item())

    sums_job = last_sums.copy()                           # totals of the last timed step (N > 1: all-reduced over the ranks)
    shard_info = None
    if use_dist and world > 1:
        # every rank's shard and what its GPU holds (after the timed region): rows it owns, device memory in use on its
        # device (hipMemGetInfo: everything resident there, all processes), peak of this process's torch allocator
        free_b, total_b = torch.cuda.mem_get_info()
        mine = {"rank": rank, "rows": [int(a), int(b)], "device": int(local_rank),
                "device_mem_used_gb": round((total_b - free_b) / 2 ** 30, 3)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        shard_info = gathered
    gpu_U = None
    if world == 1 and not use_dist and args.mode == "L" and not args.no_cpu_baseline and p <= 64:
        # parity_in_run, GPU half (untimed): one more evaluation that also writes the U entries, copied to the host
        plan.eval("matern", covparms, tau, flags | G.GPV_WANT_U, stream=stream)
        s_u = plan.sums()
        gpu_U = {"L": plan.Lentries(), "loglik": G.loglik_z_from_sums(s_u, n), "n_failed": int(s_u[6])}

    # N > 1 proves itself (default; --no-self-check skips the second half): (1) the all-reduced row count sums[7] must equal n
    # — every rank contributed its rows exactly once; (2) rank 0 evaluates the UNSHARDED plan after the timed region and the
    # N-rank log-likelihood must equal it to 1e-12 (the shards only change the summation tree, src/U_NZentries.cpp:37-39)
    check = None
    if world > 1 and rank == 0 and args.mode != "S":
        check = {"rows_reduced": float(sums_job[7]), "rows_expected": n, "ranks": world,
                 "rows_ok": bool(sums_job[7] == n), "chol_failures_reduced": float(sums_job[6])}
        check["ok"] = check["rows_ok"]
        if not args.no_self_check:
            l1, z1, nn1, cd1, _, _ = build_workload(n, m, d, 0, 1, device=local_rank)
            full = G.Plan(l1, nn1, cd1, device=local_rank)
            full.set_data(z1)
            full.eval("matern", covparms, tau, flags)
            ll1 = G.loglik_z_from_sums(full.sums(), n)
            del full, l1, z1, nn1, cd1
            rel = abs(ll1 - loglik) / abs(ll1)
            check.update(loglik_1rank=ll1, loglik_nrank=loglik, rel_diff=rel, tol=1e-12, loglik_ok=bool(rel <= 1e-12))
            check["ok"] = bool(check["rows_ok"] and check["loglik_ok"])
        if not check["ok"]:
            print(f"[bench] self-check FAILED: {check}", file=sys.stderr)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        rows_rank = b - a
        traffic = None
        if args.config == "C3" and not custom and world == 1 and args.mode == "L":
            # HBM bytes per launch from the PMC passes of tools/profile_round.sh (profiles/rNN_pmc_traffic.json); quoted only
            # while the kernel code they were measured on is the one in this tree
            import glob
            for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                try:
                    tj = json.load(open(tf))
                    if tj.get("kernel_code_sha256") == kernel_code_sha256():
                        traffic = tj.get("hbm_bytes_per_launch")
                        break
                except Exception:
                    continue
        cond_s = "SGV" if args.mode == "S" else "z"
        ord_s = "maxmin" if args.mode == "S" else "none"
        out = {
            "metric": "vecchia_loglik_evals_per_sec", "value": args.steps / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "clock_warmup_s": args.clock_warmup_s,           # untimed evaluations before the W warm-up steps (see measure_with)
            "config": {"workload": f"n={n} {d}-D uniform, Matern nu={nu}, m={m}, ordering={ord_s}, cond.yz={cond_s}, "
                                   f"mode {args.mode} (" + ("custom sizes" if custom else f"BASELINE.json configs[{ci}]")
                                   + (f"; rows sharded over {world} GPU(s))" if args.emulate_world == 1 else
                                      f"; THIS RANK'S SHARD of an emulated {args.emulate_world}-rank job only)"),
                       "n": n, "m": m, "d": d, "covparms": covparms, "nugget": tau, "mode": args.mode,
                       "sharding": f"rows/{world}", "loglik": loglik, "setup_s": round(t_setup, 2),
                       "collective": route, "ranks": world, "rows_reduced": float(sums_job[7]),
                       "shards": shard_info},
            "roofline": roofline(k_ms, rows_rank, args.mode, traffic),
        }
        if k_b2b is not None:
            out["roofline"]["kernel_ms_back_to_back"] = k_b2b   # (kernel + the rank's all-reduce, no event pair per launch)
        if from_idle is not None:
            out["config"]["from_idle"] = from_idle
        if check is not None:
            out["self_check"] = check
        if world == 1 and not args.no_secondary and args.mode == "L":
            # secondary measurements of the same workload (same JSON line, not `value`):
            #   mode U: the U entries materialised in HBM (the literal createU product, 8*(m+1) B per set written)
            #   mode S: the reference's DEFAULTS, ordering='maxmin' + cond.yz='SGV', denominator (U2V) on the GPU
            sec = {}
            el, km, ll = measure(plan, G.GPV_WANT_LOGLIK_Z | G.GPV_WANT_U, False, args.steps, 1)
            sec["mode_U"] = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                             "kernel_ms": km, "loglik": ll,
                             "hbm_alg_gbs": alg_bytes_per_set(p, d, "U") * n / (km * 1e-3) / 1e9,
                             "what": "cond.yz='z', ordering='none', Lentries written to HBM"}
            if p <= 64:
                try:
                    t0 = time.time()
                    va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="SGV", nn_backend="gpu")
                    ps = G.Plan(va["locsord"], va["U_prep"]["revNNarray"], va["U_prep"]["revCond"], device=local_rank)
                    ps.set_data(z[va["ord_z"] - 1])
                    nlev = ps.build_posterior()
                    ts = time.time() - t0
                    el, km, ll = measure(ps, G.GPV_WANT_DENOM, True, args.steps, 2)
                    sums_S, mu_S = ps.sums().copy(), None
                    pass_roof = None
                    try:
                        # the posterior pass against ITS roofline: L2 line misses (a scattered 16-byte gather moves one 128-byte
                        # line; the chip serves ~54 such lines per ns: tools/ubench/gather_lines.hip) per
                        # nanosecond of pass.  DERIVED FROM OFFLINE COUNTERS: the misses are the per-level PMC sums of this
                        # workload under profiles/ (PMC passes cannot run inside a timed bench), quoted only while the file
                        # says it was measured on the pass code of this tree (`_meta.posterior_code_sha256`); otherwise the
                        # block says "stale" and carries no fraction.  Pass time: this run's evaluation minus its set kernel.
                        import glob
                        pass_ms = 1e3 * el / args.steps - km
                        pass_roof = {"stale": True, "pass_ms": pass_ms, "derived_from": "offline counters",
                                     "note": "no profiles/r*_posterior_levels_pmc.json measured on this tree's posterior code"}
                        for lf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_posterior_levels_pmc.json")), reverse=True):
                            lv = json.load(open(lf))
                            meta = lv.get("_meta", {})
                            if meta.get("posterior_code_sha256") != posterior_code_sha256():
                                continue
                            misses = sum(v.get("TCC_MISS_sum", 0.0) for k_, v in lv.items() if k_ != "_meta")
                            pass_roof = {"bound": "l2_miss_lines", "stale": False, "pass_ms": pass_ms, "l2_line_misses": misses,
                                         "achieved": misses / (pass_ms * 1e6), "peak": 54.0, "unit": "128-byte lines/ns",
                                         "frac": misses / (pass_ms * 1e6) / 54.0,
                                         "derived_from": "offline counters: " + os.path.basename(lf),
                                         "counters_code_sha256": meta.get("posterior_code_sha256"),
                                         "note": "pass_ms = evaluation - set kernel (includes the dense top block, the two "
                                                 "reduction kernels and the launch floor of the narrow levels); misses: "
                                                 "per-level PMC sums of the same workload, measured offline on this pass code"}
                            break
                    except Exception as e:                    # noqa: BLE001
                        pass_roof = {"stale": True, "error": repr(e)}
                    sec["mode_S"] = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                                     "sets_kernel_ms": km, "loglik": ll, "levels": nlev, "setup_s": round(ts, 2),
                                     "what": "the reference's defaults: ordering='maxmin', cond.yz='SGV'; set kernel + "
                                             "posterior pass (U2V) on one GPU; does not shard"}
                    if pass_roof is not None and args.config == "C3" and not custom:      # (the counters are this workload's)
                        sec["mode_S"]["pass_roofline"] = pass_roof
                    # the same evaluation with the posterior mean (R/vecchia_prediction.R:118-126: one more level-scheduled
                    # sweep R^T u = t; what every Newton step of the Vecchia-Laplace loop runs)
                    try:
                        el, km, ll = measure(ps, G.GPV_WANT_DENOM | G.GPV_WANT_MEAN, True, args.steps, 2)
                        sec["mode_S_mean"] = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                                              "sets_kernel_ms": km,
                                              "what": "mode_S plus the posterior mean of the latent field at the observed locations"}
                        mu_S = ps.posterior_mean()
                    except Exception as e:
                        sec["mode_S_mean"] = {"error": repr(e)}
                    del ps
                    # mode L on the reference's DEFAULT ordering ("ordering='none' ... unless the maxmin builder
                    # exists"): maxmin with the cut-9 quirk (R/vecchia_specify.R:103-106), cond.yz='z', same locations / m / covparms
                    try:
                        rn = va["U_prep"]["revNNarray"]
                        rc = np.where(rn != 0, 0, -1).astype(np.int8)       # cond.yz='z' (R/vecchia_specify.R:189-190)
                        rc[:, -1] = 1
                        pm = G.Plan(va["locsord"], rn, rc, device=local_rank)
                        pm.set_data(z[va["ord_z"] - 1])
                        el, km, ll = measure(pm, G.GPV_WANT_LOGLIK_Z, False, args.steps, 2)
                        tfm = flops_per_set(p, d) * n / (km * 1e-3) / 1e12
                        sec["mode_L_maxmin"] = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                                                "kernel_ms": km, "fp64_frac": tfm / FP64_PEAK_TF, "loglik": ll,
                                                "what": "mode L (fused log-likelihood, cond.yz='z') on ordering='maxmin' (the reference's "
                                                        "default ordering incl. its cut-9 quirk) instead of the headline's ordering='none'"}
                        del pm
                    except Exception as e:
                        sec["mode_L_maxmin"] = {"error": repr(e)}
                    if not args.no_cpu_baseline:
                        try:
                            sec["mode_S"]["parity_in_run"] = mode_S_oracle(va, z, covparms, tau, sec["mode_S"]["loglik"], sums_S, mu_S)
                        except Exception as e:
                            sec["mode_S"]["parity_in_run"] = {"error": repr(e)}
                    del va
                except Exception as e:                       # never lose the headline line to a secondary failure
                    sec["mode_S"] = {"error": repr(e)}
            if args.config == "C3" and not custom:
                # the other BASELINE.json configurations and SURVEY.md §8d's third mode, each timed by this same run
                del plan
                for name in ("C2", "C4"):
                    try:
                        sec[name] = other_config(name, local_rank, measure_with, b2b=kernel_ms_back_to_back)
                    except Exception as e:
                        sec[name] = {"error": repr(e)}
                try:
                    sec["C5_vl"] = vl_config(local_rank, parity=not (args.no_cpu_baseline or args.no_vl_parity))
                except Exception as e:
                    sec["C5_vl"] = {"error": repr(e)}
                try:
                    sec["dropin_U_D2H"] = dropin_config(n, locs, revNN, revCond, covparms, tau)
                except Exception as e:
                    sec["dropin_U_D2H"] = {"error": repr(e)}
                try:
                    sec["per_rank_step"] = per_rank_step(args)
                except Exception as e:
                    sec["per_rank_step"] = {"error": repr(e)}
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline and args.mode != "S":
            cal = cpu_baseline(locs, revNN, revCond, covparms, tau, (b - min(b - a - 2 * p, 60000), b), repeats=2, sweep=False)
            kept = {}
            if n / cal["sets_per_s"] <= args.cpu_budget_s:
                out["cpu_baseline"] = cpu_baseline(locs, revNN, revCond, covparms, tau, (0, n), repeats=3, keep=kept)
            else:
                sample = int(min(b - a - 2 * p, max(20000, cal["sets_per_s"] * args.cpu_budget_s)))
                out["cpu_baseline"] = cpu_baseline(locs, revNN, revCond, covparms, tau, (b - sample, b), repeats=3, keep=kept)
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
            if gpu_U is not None:
                try:
                    out["parity_in_run"] = parity_in_run(gpu_U["L"], gpu_U["loglik"], gpu_U["n_failed"], kept, revNN, z, tau, n)
                except Exception as e:
                    out["parity_in_run"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    rc = 0 if (check is None or check["ok"]) else 3
    if use_dist:
        # the verdict of the run is `rc`: a peer that is already gone while this rank tears its group down (gloo reports a
        # closed connection) must not replace it
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                                # noqa: BLE001
            print(f"[bench] rank {rank}: process group teardown: {e!r}", file=sys.stderr, flush=True)
    if rc:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)                                          # (no interpreter teardown with half a process group)


if __name__ == "__main__":
    main()
