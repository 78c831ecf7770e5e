import cProfile, pstats, sys, time, os
import numpy as np
sys.path.insert(0, '.')
import torch
import gpvecchia_amd as G
n, m = 1_000_000, 30
locs = np.random.default_rng(0).random((n, 2))
G.vecchia_specify(locs[:20000], m, nn_backend="gpu")
os.environ["GPV_TIMING"] = "1"
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="SGV", nn_backend="gpu")
pr.disable()
t1 = time.time()
print("vecchia_specify", t1 - t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
z = np.random.default_rng(1).standard_normal(n)
t0 = time.time()
plan = G.Plan(va["locsord"], va["U_prep"]["revNNarray"], va["U_prep"]["revCond"])
t1 = time.time(); plan.set_data(z[va["ord_z"] - 1]); t2 = time.time()
nl = plan.build_posterior(); t3 = time.time()
print("Plan()", t1 - t0, "set_data", t2 - t1, "build_posterior", t3 - t2, "levels", nl)
