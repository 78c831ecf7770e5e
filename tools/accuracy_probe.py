"""Developer tool: error of the HIP path and of the double-precision oracle against the extended-precision rows
(oracle.r_side.rows_extended) on ALL rows of ill-conditioned workloads: the distribution of err_hip / err_oracle and of
both errors in units of cond(S) * eps.  Run once per library build (GPV_LIB) to see which arithmetic shortcut of the kernel
costs accuracy.

    GPV_LIB=gpvecchia_amd/libgpvecchia_hip_x.so python tools/accuracy_probe.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(name, locs, NN, cond_mode, cp, tau, G, R, S):
    n = locs.shape[0]
    revNN = NN[:, ::-1].copy()
    if cond_mode == "y":
        revCond = np.where(revNN != 0, 1, -1).astype(np.int8)
    elif cond_mode == "SGV":
        revCond = S.whichCondOnLatent(NN)[:, ::-1].copy()
    else:
        revCond = np.where(revNN != 0, 0, -1).astype(np.int8); revCond[:, -1] = 1
    cd = np.where(revCond < 0, 0, revCond).astype(np.float64)
    ref = R.U_NZentries(R.max_threads(), n, locs, revNN, cd, np.full(n, tau), np.full(n, tau), "matern", cp)["Lentries"]
    out = G.U_NZentries(1, n, locs, revNN, revCond, np.full(n, tau), np.full(n, tau), "matern", cp)["Lentries"]
    ex = R.rows_extended(np.arange(n), locs, revNN, cd, tau, "matern", cp)
    sc = np.maximum(np.abs(ex).max(axis=1), 1e-300)
    eh = np.abs(out - ex).max(axis=1) / sc
    eo = np.abs(ref - ex).max(axis=1) / sc
    d = np.abs(out - ref).max(axis=1) / sc
    hard = np.where(d > 1e-8)[0]
    big = np.where(np.maximum(eh, eo) > 1e-10)[0]
    q = lambda v: "median %.2e  90%% %.2e  99%% %.2e  max %.2e" % tuple(np.quantile(v, [.5, .9, .99, 1.0])) if len(v) else "-"
    print(f"[{name}] n={n} rows beyond 1e-8 (hip vs oracle): {hard.size}; rows with an error > 1e-10: {big.size}")
    print(f"   err_hip    : {q(eh)}")
    print(f"   err_oracle : {q(eo)}")
    if big.size:
        r = eh[big] / np.maximum(eo[big], 1e-300)
        print(f"   err_hip/err_oracle over those {big.size} rows: {q(r)};  rows with ratio > 4: {(r > 4).sum()}, > 10: {(r > 10).sum()}")
        print(f"   sum of err over them: hip {eh[big].sum():.3e}  oracle {eo[big].sum():.3e}   (ratio {eh[big].sum() / eo[big].sum():.2f})")
    if hard.size:
        r = eh[hard] / np.maximum(eo[hard], 1e-300)
        print(f"   rows beyond 1e-8: ratio {q(r)};  failing max(4 err_oracle, 1e-8): {(eh[hard] > np.maximum(4 * eo[hard], 1e-8)).sum()}")
    sys.stdout.flush()


def main():
    import torch  # noqa: F401
    import gpvecchia_amd as G
    from gpvecchia_amd import specify as S
    from oracle import r_side as R
    print("library:", os.environ.get("GPV_LIB", "default"))
    locs = np.random.default_rng(0).random((100_000, 2))
    NN = S.find_ordered_nn_gpu(locs, 20)
    case("C2 SGV m=20 range .05 nu 1.5", locs, NN, "SGV", [1.0, 0.05, 1.5], 0.1, G, R, S)
    case("C2 y   m=20 range .05 nu 0.5", locs, NN, "y", [1.0, 0.05, 0.5], 0.1, G, R, S)
    locs3 = np.random.default_rng(3).random((40_000, 2))
    NN3 = S.find_ordered_nn_gpu(locs3, 30)
    case("n=4e4 y m=30 range .05 nu 1.5", locs3, NN3, "y", [1.0, 0.05, 1.5], 0.1, G, R, S)
    case("n=4e4 SGV m=30 range .1 nu 1.5", locs3, NN3, "SGV", [1.0, 0.1, 1.5], 0.1, G, R, S)
    l1 = np.random.default_rng(5).random((20_000, 1))
    NN1 = S.find_ordered_nn_gpu(l1, 3)
    case("1-D n=2e4 y m=3 range .004 nu 1.5", l1, NN1, "y", [1.3, 0.004, 1.5], 0.1, G, R, S)


if __name__ == "__main__":
    main()
