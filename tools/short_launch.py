"""Developer tool: what a SHORT launch of the conditioning-set kernel costs beyond its work.
Times the kernel (hipEvent pair on the launch stream, settled clock) for one (m, d) at several n and fits
t = a + b n: `a` is the fixed cost (dispatch, per-wave prologue, tail quantisation, last-workgroup reduction) that the
per-rank shard of an 8-GPU job and BASELINE config C2 pay once per evaluation.

    python tools/short_launch.py [--m 30] [--d 2] [--sizes 31250,62500,125000,250000,500000] [--flags L]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=30)
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--sizes", default="31250,62500,125000,250000,500000,1000000")
    ap.add_argument("--nu", type=float, default=1.5)
    ap.add_argument("--iters", type=int, default=300)
    a = ap.parse_args()
    import torch  # noqa: F401
    import gpvecchia_amd as G
    from gpvecchia_amd import specify as S
    sizes = [int(x) for x in a.sizes.split(",")]
    nmax = max(sizes)
    locs = np.random.default_rng(0).random((nmax, a.d))
    z = np.random.default_rng(1).standard_normal(nmax)
    NN = S.find_ordered_nn_gpu(locs, a.m)
    revNN = NN[:, ::-1].copy()
    revCond = np.where(revNN != 0, 0, -1).astype(np.int8)
    revCond[:, -1] = 1
    rng_ = {2: 0.02, 3: 0.05}.get(a.d, 0.02) * (1e6 / nmax) ** (1.0 / a.d)
    rows = []
    for n in sizes:
        # rows [0, n) of the SAME plan data (a shard): the density, and with it the arithmetic per set, is the same at every size
        plan = G.Plan(locs, revNN, revCond, row_begin=0, row_end=n)
        plan.set_data(z)
        cp = [1.0, rng_, a.nu]
        t_end = time.perf_counter() + 0.15
        while time.perf_counter() < t_end:                      # settle the shader clock
            plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z); plan.sums()
        km, wall = [], []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z)
            plan.sums()
            wall.append(time.perf_counter() - t0)
            km.append(plan.last_kernel_ms())
        rows.append((n, float(np.median(km)), float(np.min(km)), 1e3 * float(np.median(wall))))
        print(f"n={n:8d}  kernel median {rows[-1][1] * 1e3:8.1f} us  min {rows[-1][2] * 1e3:8.1f} us   step wall {rows[-1][3] * 1e3:8.1f} us",
              flush=True)
        del plan
    x = np.array([r[0] for r in rows], float)
    y = np.array([r[1] for r in rows], float)
    b, a0 = np.polyfit(x, y, 1)
    print(f"fit: kernel = {a0 * 1e3:.1f} us + {b * 1e6:.4f} us per 1000 sets   (m={a.m}, d={a.d}, nu={a.nu}; GPV_GRID_MULT={os.environ.get('GPV_GRID_MULT', '-')})")


if __name__ == "__main__":
    main()
