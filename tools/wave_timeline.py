"""Developer tool: where the time of a SHORT launch of the conditioning-set kernel goes, from wall-clock stamps every
wavefront leaves (a library built with -DGPV_TRACE_TIMES): kernel entry, first task gathered, task loop left, kernel end.

    python -m gpvecchia_amd.build --tag _trace --plist 21,31 --flags=-DGPV_TRACE_TIMES
    GPV_LIB=gpvecchia_amd/libgpvecchia_hip_trace.so python tools/wave_timeline.py [--m 30] [--rows 125000]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=30)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--rows", type=int, default=125_000)
    a = ap.parse_args()
    import torch  # noqa: F401
    import gpvecchia_amd as G
    from gpvecchia_amd import specify as S, _lib as L
    locs = np.random.default_rng(0).random((a.n, 2))
    z = np.random.default_rng(1).standard_normal(a.n)
    NN = S.find_ordered_nn_gpu(locs, a.m, rows=(0, a.rows))
    revNN = NN[:, ::-1].copy()
    revCond = np.where(revNN != 0, 0, -1).astype(np.int8); revCond[:, -1] = 1
    plan = G.Plan(locs, revNN, revCond, row_begin=0, row_end=a.rows)
    plan.set_data(z)
    cp = [1.0, 0.02, 1.5]
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z); plan.sums()
    kms = []
    for _ in range(50):
        plan.eval("matern", cp, 0.1, G.GPV_WANT_LOGLIK_Z); plan.sums(); kms.append(plan.last_kernel_ms())
    lib = L.lib()
    lib.gpv_plan_debug_block_sums.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    half = 16384 // 2
    buf = np.zeros(half * 8)
    grid = C.c_int(0)
    L.check(lib.gpv_plan_debug_block_sums(plan._h, half * 8, half * 8, buf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(grid)), "debug")
    raw = buf.view(np.uint64).reshape(half, 8)
    nw = grid.value * 4
    t = raw[:nw, :4].astype(np.float64) / 100.0                     # 100 MHz wall clock -> us
    tasks = raw[:nw, 4].astype(int)
    if not t[:, 0].any():
        raise SystemExit("no stamps: the library was not built with -DGPV_TRACE_TIMES")
    t0 = t[:, 0].min()
    t -= t0
    q = lambda v: "min %7.2f  median %7.2f  max %7.2f" % (v.min(), np.median(v), v.max())
    print(f"rows {a.rows}, m {a.m}: kernel {np.median(kms) * 1e3:.1f} us by hipEvents; grid {grid.value} workgroups, {nw} wavefronts; "
          f"tasks per wavefront {tasks.min()}..{tasks.max()}")
    print("wave start        (us after the first) :", q(t[:, 0]))
    print("first task gathered - start  (prologue):", q(t[:, 1] - t[:, 0]))
    print("task loop  (first gather -> loop left)  :", q(t[:, 2] - t[:, 1]))
    print("   per task                              :", q((t[:, 2] - t[:, 1]) / np.maximum(tasks, 1)))
    print("loop left  (us after the first start)   :", q(t[:, 2]))
    print("kernel end - loop left (reduce, tail)   :", q(t[:, 3] - t[:, 2]))
    print("last stamp                               : %.2f us  (hipEvent duration minus this = launch + completion overhead)" % t[:, 3].max())
    for tk in np.unique(tasks):
        sel = tasks == tk
        print(f"   waves with {tk} tasks: {sel.sum():5d}, loop left at {q(t[sel, 2])}")
    hw = raw[:nw, 5]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; se = (hw >> 13) & 7
    key = (raw[:nw, 5] >> 4) & 0  # placeholder


if __name__ == "__main__":
    main()
