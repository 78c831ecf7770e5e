"""Developer tool: PCIe-inclusive times of the literal drop-in gpv_U_NZentries at n = 1e6, m = 30 (first call builds the
plan, later calls hit the plan cache).  GPV_TIMING=1 prints the library's own phase times on stderr."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpvecchia_amd as G
from gpvecchia_amd import _lib as L
from gpvecchia_amd import specify as S

n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 30
locs = np.asfortranarray(np.random.default_rng(0).random((n, 2)))
NN = S.find_ordered_nn_gpu(locs, m)
revNN = np.asfortranarray(NN[:, ::-1].astype(np.int32))
revCond = np.asfortranarray(np.where(revNN != 0, 0, L.NA_INTEGER).astype(np.int32))
revCond[:, -1] = 1
tau = np.full(n, 0.1)
for it in range(4):
    t0 = time.perf_counter()
    out = G.U_NZentries(1, n, locs, revNN, revCond, tau, tau, "matern", [1.0, 0.02, 1.5 if it < 3 else 0.5])
    print(f"call {it}: {1e3 * (time.perf_counter() - t0):.1f} ms through the Python mirror", flush=True)
