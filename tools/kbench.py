"""Developer tool: time the conditioning-set kernel of one or more library builds on the
same workload (each build in its own process).  Not part of the product or the tests.

    python tools/kbench.py [--n 1000000] [--configs 30x2,20x2,60x3] lib1.so lib2.so ...
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    import gpvecchia_amd as G
    from gpvecchia_amd import specify as S
    res = {}
    for cfg in args.configs.split(","):
        m, d = (int(x) for x in cfg.split("x"))
        n = args.n if m <= 32 else args.n // 2
        cache = f"/tmp/kbench_{n}_{m}_{d}.npz"
        if os.path.exists(cache):
            z = np.load(cache)
            locs, NN = z["locs"], z["NN"]
        else:
            locs = np.random.default_rng(0).random((n, d))
            NN = S.find_ordered_nn(locs, m)
            np.savez(cache, locs=locs, NN=NN)
        revNN = NN[:, ::-1].copy()
        revCond = np.where(revNN != 0, 0, -1).astype(np.int8)
        revCond[:, -1] = 1
        plan = G.Plan(locs, revNN, revCond)
        plan.set_data(np.random.default_rng(1).standard_normal(n))
        cp = [1.0, 0.02 if d == 2 else 0.05, 1.5]
        out = {}
        for mode, flags in (("L", G.GPV_WANT_LOGLIK_Z), ("U", G.GPV_WANT_U)):
            ts = []
            for it in range(args.iters + 2):
                plan.eval("matern", cp, 0.1, flags)
                s = plan.sums()
                if it >= 2:
                    ts.append(plan.last_kernel_ms())
            out[mode] = round(float(np.median(ts)), 4)
        res[cfg] = out
        del plan
        if args.sgv and m <= 32:
            import time
            t0 = time.time()
            cond = S.whichCondOnLatent(NN)
            t_cond = time.time() - t0
            rc = cond[:, ::-1].copy()
            plan = G.Plan(locs, revNN, rc)
            t0 = time.time()
            nlev = plan.build_posterior()
            t_build = time.time() - t0
            plan.set_data(np.random.default_rng(1).standard_normal(n))
            ts = []
            for it in range(args.iters + 2):
                t0 = time.time()
                plan.eval("matern", cp, 0.1, G.GPV_WANT_DENOM)
                s = plan.sums()
                if it >= 2:
                    ts.append((time.time() - t0) * 1e3)
            out["SGV_eval_wall_ms"] = round(float(np.median(ts)), 3)
            out["SGV_sets_kernel_ms"] = round(plan.last_kernel_ms(), 3)
            out["SGV_levels"] = nlev
            out["SGV_setup_s"] = [round(t_cond, 1), round(t_build, 2)]
            out["SGV_loglik"] = G.loglik_from_sums(s, n)
            del plan
    print("KBENCH " + json.dumps(res), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--configs", default="30x2,20x2,60x3,10x2")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--sgv", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a)
    else:
        libs = a.libs or [os.path.join(ROOT, "gpvecchia_amd", "libgpvecchia_hip.so")]
        for lib in libs:
            env = dict(os.environ, GPV_LIB=os.path.abspath(lib))
            r = subprocess.run([sys.executable, __file__, "--child", "--n", str(a.n), "--configs", a.configs,
                                "--iters", str(a.iters)] + (["--sgv"] if a.sgv else []), env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("KBENCH")]
            print(os.path.basename(lib), line[0] if line else ("FAILED\n" + r.stdout[-2000:] + r.stderr[-2000:]), flush=True)
