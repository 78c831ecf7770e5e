"""Developer tool: how much of a posterior-pass evaluation is launch gaps and how much host turnaround?
n = 5e5, m = 30, maxmin + SGV; (a) evaluations enqueued back to back, one wait at the end; (b) one wait per evaluation."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpvecchia_amd as G

n, m = 500_000, 30
rng = np.random.default_rng(0)
locs = rng.random((n, 2)); z = rng.standard_normal(n)
va = G.vecchia_specify(locs, m, nn_backend="gpu")
plan = G.Plan(va["locsord"], va["U_prep"]["revNNarray"], va["U_prep"]["revCond"])
plan.set_data(z[va["ord_z"] - 1]); plan.build_posterior()
plan.set_kernel_timing(False)
cp = [1.0, 0.03, 1.5]
for flags, tag in ((G.GPV_WANT_DENOM, "denominator"), (G.GPV_WANT_MEAN, "denominator + mean")):
    for _ in range(30):
        plan.eval("matern", cp, 0.1, flags); plan.sums()
    K = 40
    t0 = time.perf_counter()
    for _ in range(K):
        plan.eval("matern", cp, 0.1, flags)
    t1 = time.perf_counter()
    plan.sums()
    t2 = time.perf_counter()
    for _ in range(K):
        plan.eval("matern", cp, 0.1, flags); plan.sums()
    t3 = time.perf_counter()
    print(f"{tag:20s}: back to back {1e3*(t2-t0)/K:.3f} ms per eval (host enqueue {1e3*(t1-t0)/K:.3f}); one wait per eval {1e3*(t3-t2)/K:.3f} ms")
