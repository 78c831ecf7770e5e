"""End-to-end at the reference's defaults (ordering='maxmin', cond.yz='SGV'), n = 1e6, m = 30."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import gpvecchia_amd as G
n, m = 1_000_000, 30
rng = np.random.default_rng(0)
locs = rng.random((n, 2)); z = rng.standard_normal(n)
t = time.time(); va = G.vecchia_specify(locs, m); t_spec = time.time() - t
t = time.time(); ll = G.vecchia_likelihood(z, va, [1.0, 0.02, 1.5], 0.1); t_first = time.time() - t
ts = []
for cp in ([1.1, 0.02, 1.5], [1.0, 0.03, 1.5], [1.0, 0.02, 0.9]):
    t = time.time(); l2 = G.vecchia_likelihood(z, va, cp, 0.1); ts.append(time.time() - t)
print(f"vecchia_specify(n=1e6, m=30, maxmin, SGV): {t_spec:.1f} s; first likelihood (plan upload + posterior structure): "
      f"{t_first:.2f} s; next evaluations: {[round(x * 1e3, 1) for x in ts]} ms; loglik {ll:.6f}")
