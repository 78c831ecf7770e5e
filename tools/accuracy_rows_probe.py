"""Developer tool: where does the posterior mean's error come from when HIP and oracle U rows are equally accurate row by row?
Feeds the HIP U entries through the ORACLE's sparse chain, and measures per-row backward errors (residuals of S x = e_last / d in
extended precision) of both implementations' rows.    python tools/accuracy_rows_probe.py
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: F401
import gpvecchia_amd as G
from gpvecchia_amd import api as A

sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import r_side as R
from test_gpu_fuzz import _oracle_va

seed = 9298
rng = np.random.default_rng(seed)
d = int(rng.integers(1, 4))
n = int(rng.choice([rng.integers(5, 64), rng.integers(64, 130), rng.integers(130, 400), rng.integers(400, 3000)]))
m = int(min(n - 1, rng.integers(2, 45)))
locs = rng.random((n, d)); z = rng.standard_normal(n)
nu = 0.5 if d == 1 else float(rng.choice([0.5, 1.5, 2.5]))
cp = [float(0.5 + rng.random()), float(0.05 + 0.3 * rng.random()), nu]
tau = 0.05 + 0.3 * rng.random(n) if rng.random() < 0.7 else float(0.05 + 0.3 * rng.random())
cond = str(rng.choice(["SGV", "SGV", "y"]))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    va = G.vecchia_specify(locs, m, ordering=str(rng.choice(["maxmin", "none"])), cond_yz=cond)
vb = _oracle_va(va)
prep = vb["U_prep"]
U_obj = A.createU(va, cp, tau)
Lh = np.ascontiguousarray(U_obj["Lentries"])
Us_o = R.createU_sparse(vb, cp, tau)
Lo = np.ascontiguousarray(Us_o["U_entries"]["Lentries"])
ex = R.posterior_extended(z, vb, cp, tau)
mux = np.empty(n); mux[va["ord"] - 1] = ex["mu_ord"]
sc = max(1.0, np.abs(mux).max())


def chain(L):
    Us = R.createU_sparse(vb, cp, tau, U_entries=dict(Lentries=L, Zentries=Us_o["U_entries"]["Zentries"]))
    return R.vecchia_mean_sparse(z, Us, R.U2V_sparse(Us))


print("oracle chain on ORACLE rows vs exact: %.2e" % (np.abs(chain(Lo) - mux).max() / sc))
print("oracle chain on HIP rows    vs exact: %.2e" % (np.abs(chain(Lh) - mux).max() / sc))
# mixtures: HIP rows for the first / second half of the ordering
for cut in (n // 4, n // 2, 3 * n // 4):
    Lm = Lo.copy(); Lm[:cut] = Lh[:cut]
    Lm2 = Lo.copy(); Lm2[cut:] = Lh[cut:]
    print(f"   HIP rows [0,{cut}) + oracle rest: %.2e ;  oracle [0,{cut}) + HIP rest: %.2e" % (np.abs(chain(Lm) - mux).max() / sc, np.abs(chain(Lm2) - mux).max() / sc))
# only the diagonal entry d_k from HIP / only the off-diagonal entries from HIP
n0 = (~np.isnan(prep["revNNarray"])).sum(axis=1)
Ld = Lo.copy(); Ld[np.arange(n), n0 - 1] = Lh[np.arange(n), n0 - 1]
Lf = Lh.copy(); Lf[np.arange(n), n0 - 1] = Lo[np.arange(n), n0 - 1]
print("   oracle rows with HIP's d_k: %.2e ;  HIP rows with the oracle's d_k: %.2e" % (np.abs(chain(Ld) - mux).max() / sc, np.abs(chain(Lf) - mux).max() / sc))
# backward errors in extended precision: S x - e_last / d  relative to |S| |x|
ld = np.longdouble
lo_ = vb["locsord"].astype(ld)
nug = np.broadcast_to(np.asarray(tau, dtype=np.float64), (n,))[va["ord"] - 1]
s5 = np.sqrt(ld(5))
res_h, res_o = [], []
for k in range(n):
    ok = ~np.isnan(prep["revNNarray"][k])
    J = prep["revNNarray"][k, ok].astype(int) - 1
    c = prep["revCond"][k, ok]
    P = lo_[J]
    D = np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(axis=2))
    t = s5 * D / ld(cp[1])
    S = ld(cp[0]) * np.exp(-t) * (1 + t + t * t / 3)
    S[np.diag_indices(len(J))] = ld(cp[0]) + nug[J].astype(ld) * (1 - c)
    for L, out in ((Lh, res_h), (Lo, res_o)):
        x = L[k, :len(J)].astype(ld)
        rhs = np.zeros(len(J), dtype=ld); rhs[-1] = 1 / x[-1]              # S x = e_last / d  (x = S^-1 e_last * d ... d = x_last)
        r = S @ x - rhs
        out.append(float(np.abs(r).max() / (np.abs(S) @ np.abs(x)).max()))
res_h, res_o = np.array(res_h), np.array(res_o)
print("backward error |S x - e/d| / (|S||x|): hip median %.1e max %.1e sum %.2e | oracle median %.1e max %.1e sum %.2e" %
      (np.median(res_h), res_h.max(), res_h.sum(), np.median(res_o), res_o.max(), res_o.sum()))
