import sys, time, numpy as np
sys.path.insert(0, '.')
from gpvecchia_amd import specify as S
for n in (100000, 1000000):
    locs = np.random.default_rng(0).random((n, 2))
    t = time.time(); a = S.find_ordered_nn_gpu(locs, 30); tg = time.time() - t
    t = time.time(); b = S.find_ordered_nn(locs, 30); tc = time.time() - t
    print(f"n={n} m=30: GPU brute force {tg:.2f}s  host cKDTree {tc:.2f}s  identical={np.array_equal(a, b)}", flush=True)
