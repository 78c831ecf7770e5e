"""Developer tool: BASELINE.json configs[4] (n = 5e5 2-D, Vecchia-Laplace Poisson likelihood, m = 30) on one GPU:
wall time of vecchia_specify, of the Newton-Raphson loop (R/vecchia_laplace_NR.R:31-155) and of one
vecchia_laplace_likelihood.  Not part of the product or the tests.

    python tools/vl_bench.py [--n 500000] [--m 30]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpvecchia_amd as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=500_000)
ap.add_argument("--m", type=int, default=30)
a = ap.parse_args()
rng = np.random.default_rng(0)
locs = rng.random((a.n, 2))
y = 0.8 * np.sin(5 * locs[:, 0]) * np.cos(4 * locs[:, 1]) + 0.3          # a cheap smooth latent field
z = rng.poisson(np.exp(y)).astype(float)
cp = [1.0, 0.03, 1.5]
t0 = time.time()
va = G.vecchia_specify(locs, a.m)                                          # defaults: maxmin, SGV
t_spec = time.time() - t0
t0 = time.time()
post = G.calculate_posterior_VL(z, va, "poisson", cp)
t_first = time.time() - t0                                                 # includes plan upload + posterior structure
t0 = time.time()
post = G.calculate_posterior_VL(z, va, "poisson", cp)                      # device loop (gpv_plan_vl_step)
t_nr = time.time() - t0
t0 = time.time()
post_h = G.calculate_posterior_VL(z, va, "poisson", cp, on_device=False)   # family arithmetic in NumPy, data over PCIe every step
t_nr_host = time.time() - t0
t0 = time.time()
ll = G.vecchia_laplace_likelihood(z, va, "poisson", cp)
t_ll = time.time() - t0
print(json.dumps({"n": a.n, "m": a.m, "specify_s": round(t_spec, 2), "first_posterior_s": round(t_first, 2),
                  "nr_loop_s": round(t_nr, 3), "nr_iters": post["iter"], "converged": bool(post["cnvgd"]),
                  "ms_per_nr_iter": round(1e3 * t_nr / max(post["iter"], 1), 2),
                  "ms_per_nr_iter_host_loop": round(1e3 * t_nr_host / max(post_h["iter"], 1), 2), "laplace_loglik_s": round(t_ll, 3),
                  "loglik": ll, "rmse_latent": float(np.sqrt(np.mean((post["mean"] - y) ** 2)))}))
