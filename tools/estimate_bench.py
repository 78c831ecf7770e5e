"""Developer tool: wall time of vecchia_estimate (R/vecchia_wrappers.R:28-106: Nelder-Mead over vecchia_likelihood, the
smoothness varies at every step, i.e. every evaluation takes the general-nu Matern path) on synthetic data.

    python tools/estimate_bench.py [--n 200000] [--m 20]
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
ap.add_argument("--n", type=int, default=200_000)
ap.add_argument("--m", type=int, default=20)
a = ap.parse_args()
rng = np.random.default_rng(0)
locs = rng.random((a.n, 2))
# a smooth random field from a few hundred random Fourier features (cheap stand-in for a GP draw) + noise
W = rng.standard_normal((256, 2)) * 12.0
ph = rng.random(256) * 2 * np.pi
y = np.sqrt(2.0 / 256) * np.cos(locs @ W.T + ph).sum(axis=1)
z = 1.5 + y + 0.3 * rng.standard_normal(a.n)
t0 = time.time()
res = G.vecchia_estimate(z, locs, m=a.m, output_level=0)
t = time.time() - t0
print(json.dumps({"n": a.n, "m": a.m, "seconds": round(t, 2), "evaluations": int(res["n_evals"]),
                  "theta_hat": [round(float(v), 5) for v in res["theta_hat"]]}))
