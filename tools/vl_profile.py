import numpy as np, time, sys, cProfile, pstats
sys.path.insert(0, '.')
import gpvecchia_amd as G
n, m = 500_000, 30
rng = np.random.default_rng(0)
locs = rng.random((n, 2))
y = 0.8 * np.sin(5 * locs[:, 0]) * np.cos(4 * locs[:, 1]) + 0.3
z = rng.poisson(np.exp(y)).astype(float)
cp = [1.0, 0.03, 1.5]
va = G.vecchia_specify(locs, m, nn_backend="gpu")
post = G.calculate_posterior_VL(z, va, "poisson", cp)
post = G.calculate_posterior_VL(z, va, "poisson", cp)
t = time.time(); post = G.calculate_posterior_VL(z, va, "poisson", cp); print("call s", time.time() - t, "iters", post["iter"])
pr = cProfile.Profile(); pr.enable()
post = G.calculate_posterior_VL(z, va, "poisson", cp)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
