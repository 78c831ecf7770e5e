"""Developer tool: time of the exact ordered nearest-neighbour search (gpv_find_ordered_nn) through the grid and, with
GPV_NN_BRUTE=1 in the environment, by brute force.   python tools/nnbench2.py [--n 1000000] [--d 2] [--m 30]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--d", type=int, default=2)
ap.add_argument("--m", type=int, default=30); a = ap.parse_args()
import torch  # noqa
from gpvecchia_amd import specify as S
locs = np.random.default_rng(0).random((a.n, a.d))
S.find_ordered_nn_gpu(locs[:10000], a.m)
t0 = time.time(); NN = S.find_ordered_nn_gpu(locs, a.m); t = time.time() - t0
print(f"n={a.n} d={a.d} m={a.m} GPV_NN_BRUTE={os.environ.get('GPV_NN_BRUTE', '-')}: {t:.3f} s (host prep + kernels + copies), checksum {int(NN.astype(np.int64).sum())}")
