"""Developer tool: duration of the conditioning-set kernel launch by launch, from a cold GPU, after an idle gap and
with a host pause between the launches (does the clock ramp shape what a short timed region sees?).

    python tools/clock_ramp.py [--config C3] [--evals 400]
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
    ap.add_argument("--config", default="C3")
    ap.add_argument("--evals", type=int, default=400)
    a = ap.parse_args()
    import bench
    import gpvecchia_amd as G
    ci, n, m, d, nu, rng_ = bench.CONFIGS[a.config]
    locs, z, revNN, revCond, r0, r1 = bench.build_workload(n, m, d, 0, 1, device=0)
    plan = G.Plan(locs, revNN, revCond)
    plan.set_data(z)
    cp = [1.0, rng_, nu]

    def series(k, pause=0.0):
        out = []
        t0 = time.perf_counter()
        for _ in range(k):
            plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z)
            plan.sums()
            out.append(plan.last_kernel_ms())
            if pause:
                time.sleep(pause)
        return np.array(out), time.perf_counter() - t0

    def show(tag, s, wall):
        k = len(s)
        pts = [0, 1, 2, 5, 10, 15, 20, 25, 30, 40, 60, 80, 120, 160, 240, 320, k - 1]
        print(tag, "wall %.3f s" % wall, " ".join(f"{i}:{s[i]*1e3:.0f}" for i in pts if i < k), "| last-50 mean %.1f us" % (1e3 * s[-50:].mean()),
              flush=True)

    s, w = series(a.evals)
    show("cold     ", s, w)
    s, w = series(a.evals)
    show("again    ", s, w)
    for gap in (0.05, 0.5, 3.0):
        time.sleep(gap)
        s, w = series(120)
        show(f"idle {gap:4.2f}s", s, w)
    s, w = series(60, pause=0.01)
    show("10ms pause between launches", s, w)
    s, w = series(60, pause=0.001)
    show("1ms pause between launches", s, w)


if __name__ == "__main__":
    main()
