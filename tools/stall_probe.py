"""Developer tool: where do multi-millisecond stalls fall in a long run of short evaluations (launch index, time)?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gpvecchia_amd as G

ci, n, m, d, nu, rng_ = bench.CONFIGS["C3"]
for world, k in ((8, 6000), (64, 12000), (1, 1500)):
    locs, z, revNN, revCond, a, b = bench.build_workload(n, m, d, 0, world, device=0)
    plan = G.Plan(locs, revNN, revCond, device=0, row_begin=a, row_end=b)
    plan.set_data(z)
    plan.set_kernel_timing(False)
    cp = [1.0, rng_, nu]
    ts = np.empty(k + 1)
    ts[0] = time.perf_counter()
    for i in range(k):
        plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z)
        plan.sums()
        ts[i + 1] = time.perf_counter()
    dt = 1e3 * np.diff(ts)
    med = np.median(dt)
    big = np.nonzero(dt > max(1.0, 3 * med))[0]
    print(f"rows n/{world}: {k} steps, median {1e3*med:.1f} us; stalls (> max(1 ms, 3x median)):",
          [(int(i), round(float(dt[i]), 1), f"t={1e3*(ts[i]-ts[0]):.0f}ms") for i in big], flush=True)
    del plan
