"""Times the literal drop-in gpv_U_NZentries (host buffers in, host buffers out: plan build, H2D, kernel,
device transpose, D2H) and the plan-API U mode + D2H, at BASELINE config C3."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import gpvecchia_amd as G
from gpvecchia_amd import specify as S
n, m = 1_000_000, 30
locs = np.random.default_rng(0).random((n, 2))
NN = S.find_ordered_nn_gpu(locs, m)
revNN = NN[:, ::-1].copy()
revCond = np.where(revNN != 0, 0, -1).astype(np.int8); revCond[:, -1] = 1
nug = np.full(n, .1)
for it in range(3):
    t = time.time()
    out = G.U_NZentries(1, n, locs, revNN, revCond, nug, nug, "matern", [1, .02, 1.5])
    t_lit = time.time() - t
plan = G.Plan(locs, revNN, revCond)
ts = []
for it in range(4):
    t = time.time()
    plan.eval("matern", [1, .02, 1.5], .1, G.GPV_WANT_U)
    L = plan.Lentries()
    ts.append(time.time() - t)
print(f"literal gpv_U_NZentries (everything from host buffers): {t_lit:.3f} s; plan-resident eval + Lentries to host "
      f"(248 MB D2H + transpose): {min(ts)*1e3:.1f} ms; identical: {np.array_equal(L, out['Lentries'])}")
