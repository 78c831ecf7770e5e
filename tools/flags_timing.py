"""Developer tool: set-kernel time (hipEvents) by flag combination on a maxmin + SGV plan and on an ordering='none' plan."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpvecchia_amd as G
from gpvecchia_amd import specify as S

n, m = 1_000_000, 30
locs = np.random.default_rng(0).random((n, 2))
z = np.random.default_rng(1).standard_normal(n)
cp, tau = [1.0, 0.02, 1.5], 0.1
for ordering in ("none", "maxmin"):
    va = G.vecchia_specify(locs, m, ordering=ordering, cond_yz="SGV", nn_backend="gpu")
    plan = G.Plan(va["locsord"], va["U_prep"]["revNNarray"], va["U_prep"]["revCond"])
    plan.set_data(z[va["ord_z"] - 1])
    plan.build_posterior()
    for name, fl in (("L", G.GPV_WANT_LOGLIK_Z), ("U", G.GPV_WANT_U), ("U|N", G.GPV_WANT_U | G.GPV_WANT_NUMERATOR),
                     ("U|L", G.GPV_WANT_U | G.GPV_WANT_LOGLIK_Z), ("DENOM", G.GPV_WANT_DENOM)):
        ts = []
        for it in range(8):
            plan.eval("matern", cp, tau, fl)
            plan.sums()
            if it >= 2:
                ts.append(plan.last_kernel_ms())
        print(ordering, name, round(float(np.median(ts)), 4), flush=True)
    del plan
