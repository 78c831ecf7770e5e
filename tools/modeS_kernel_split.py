"""Developer tool: what the set kernel pays inside --mode S (maxmin + SGV plan) for each of its outputs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpvecchia_amd as G
n, m = 1_000_000, 30
rng = np.random.default_rng(0)
locs = rng.random((n, 2)); z = rng.standard_normal(n)
va = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="SGV", nn_backend="gpu")
plan = G.Plan(va["locsord"], va["U_prep"]["revNNarray"], va["U_prep"]["revCond"])
plan.set_data(z[va["ord_z"] - 1]); nlev = plan.build_posterior()
cp = [1.0, 0.02, 1.5]
def t(flags, tag):
    for _ in range(60):
        plan.eval("matern", cp, 0.1, flags); plan.sums()
    ks = []
    for _ in range(30):
        plan.eval("matern", cp, 0.1, flags); plan.sums(); ks.append(plan.last_kernel_ms())
    print(f"{tag:40s} set kernel {np.mean(ks):.4f} ms", flush=True)
t(G.GPV_WANT_NUMERATOR, "numerator sums only")
t(G.GPV_WANT_NUMERATOR | G.GPV_WANT_U, "numerator + U entries")
t(G.GPV_WANT_DENOM, "denominator (compact blocks written)")
t(G.GPV_WANT_DENOM | G.GPV_WANT_U, "denominator + U entries")
va2 = G.vecchia_specify(locs, m, ordering="maxmin", cond_yz="z", nn_backend="gpu")
p2 = G.Plan(va2["locsord"], va2["U_prep"]["revNNarray"], va2["U_prep"]["revCond"]); p2.set_data(z[va2["ord_z"] - 1])
plan = p2
t(G.GPV_WANT_LOGLIK_Z, "maxmin ordering, cond.yz='z', likelihood")
