"""Developer tool: where vecchia_specify(n = 1e6, m = 30, maxmin, SGV) spends its time (cProfile, cumulative)."""
import cProfile, pstats, sys
import numpy as np
sys.path.insert(0, '.')
import gpvecchia_amd as G
n, m = 1_000_000, 30
locs = np.random.default_rng(0).random((n, 2))
G.vecchia_specify(locs[:20000], m)                                   # warm the library
pr = cProfile.Profile(); pr.enable()
va = G.vecchia_specify(locs, m)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
