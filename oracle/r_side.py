"""oracle/r_side.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain numpy / pure-Python restatement of the R functions that sit either side
of the U_NZentries hot path in GPvecchia (reference checked out read-only at
/root/reference; every function cites the file:line it follows).  Written as
slow, literal loops on purpose: it is the checker, never the thing measured or
shipped.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.

PARITY STATUS: "parity unpinned" against reference-run output (R is absent from
this image, the package cannot be loaded).  Pinned by the identities the
reference itself documents: m = n-1 reproduces the exact multivariate normal
density (vignettes/GPvecchia_vignette.Rmd:129-139) for cond.yz in {z, y, SGV},
and U U^T equals the joint precision of (y, z).

All index arrays here are 1-based with NaN/0 for "missing", exactly like the R
objects, so that the layout contract of R/U_sparsity.R is exercised literally.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


# ----------------------------------------------------------------------------
# C restatement loader (oracle/u_nzentries_oracle.c)
# ----------------------------------------------------------------------------
def build_c_oracle(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("u_nzentries_oracle.c", "sparse_chol_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        # GPV_ORACLE_LIB: another build of the same two C files (tools/sanitize_host.sh: -fsanitize=address,undefined)
        _LIB = ctypes.CDLL(os.environ.get("GPV_ORACLE_LIB") or build_c_oracle())
        dp = ctypes.POINTER(ctypes.c_double)
        lp = ctypes.POINTER(ctypes.c_long)
        _LIB.oracle_U_NZentries.restype = ctypes.c_long
        _LIB.oracle_U_NZentries.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                            ctypes.c_int, dp, lp, dp, dp, dp, ctypes.c_int, dp, dp, dp]
        _LIB.oracle_U_NZentries_mat.restype = ctypes.c_long
        _LIB.oracle_U_NZentries_mat.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                                lp, dp, dp, dp, dp]
        _LIB.oracle_MaternFun.argtypes = [dp, ctypes.c_long, dp, dp]
        _LIB.oracle_EsqeFun.argtypes = [dp, ctypes.c_long, dp, dp]
        _LIB.oracle_max_threads.restype = ctypes.c_int
        _LIB.oracle_rows_extended.restype = ctypes.c_long
        _LIB.oracle_rows_extended.argtypes = [ctypes.c_int, ctypes.c_long, lp, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                              dp, lp, dp, dp, ctypes.c_int, dp, dp, dp]
        _LIB.oracle_long_double_digits.restype = ctypes.c_int
        _LIB.oracle_sparse_chol_symbolic.restype = ctypes.c_long
        _LIB.oracle_sparse_chol_symbolic.argtypes = [ctypes.c_long, lp, lp, lp, lp]
        _LIB.oracle_sparse_chol_numeric.restype = ctypes.c_long
        _LIB.oracle_sparse_chol_numeric.argtypes = [ctypes.c_long, lp, lp, dp, lp, lp, lp, dp]
        _LIB.oracle_sparse_lsolve.restype = None
        _LIB.oracle_sparse_lsolve.argtypes = [ctypes.c_long, lp, lp, dp, dp]
        _LIB.oracle_sparse_ltsolve.restype = None
        _LIB.oracle_sparse_ltsolve.argtypes = [ctypes.c_long, lp, lp, dp, dp]
        ldp = ctypes.POINTER(ctypes.c_longdouble)
        _LIB.oracle_sparse_chol_numeric_ld.restype = ctypes.c_long
        _LIB.oracle_sparse_chol_numeric_ld.argtypes = [ctypes.c_long, lp, lp, ldp, lp, lp, lp, ldp]
        _LIB.oracle_sparse_lsolve_ld.restype = None
        _LIB.oracle_sparse_lsolve_ld.argtypes = [ctypes.c_long, lp, lp, ldp, ldp]
        _LIB.oracle_sparse_ltsolve_ld.restype = None
        _LIB.oracle_sparse_ltsolve_ld.argtypes = [ctypes.c_long, lp, lp, ldp, ldp]
    return _LIB


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def max_threads() -> int:
    return int(_lib().oracle_max_threads())


class MarshalledSets:
    """The arguments of oracle_U_NZentries in the C function's own layout (what Rcpp's converters produce before
    src/U_NZentries.cpp:25 starts: src/RcppExports.cpp:57-63), with caller-allocated, pre-touched outputs.  `cargs(Ncores)` is
    the exact ctypes argument tuple; `fn` is the ctypes function object itself."""

    def __init__(self, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms):
        self.locs = np.asfortranarray(locs, dtype=np.float64)
        self.Nlocs, self.d = self.locs.shape
        nn = np.asarray(revNNarray)
        if nn.dtype.kind == "f":
            nn = np.nan_to_num(nn, nan=0.0)
        self.nn = np.asfortranarray(nn.astype(np.int64))
        self.p = self.nn.shape[1]
        cd = np.asarray(revCondOnLatent)
        if cd.dtype.kind == "f":
            cd = np.nan_to_num(cd, nan=0.0)
        self.cond = np.asfortranarray(cd, dtype=np.float64)
        self.nug = np.ascontiguousarray(nuggets, dtype=np.float64)
        self.nugo = np.ascontiguousarray(nuggets_obsord, dtype=np.float64)
        self.cp = np.ascontiguousarray(covparms, dtype=np.float64)
        self.n = int(n)
        self.code = {"matern": 0, "esqe": 1}.get(covType, 99)
        self.closed_form = not (self.code == 0 and float(self.cp[2]) not in (0.5, 1.5, 2.5))
        self.L = np.zeros((self.Nlocs, self.p), dtype=np.float64, order="F")      # zeros(): pages touched here, not in the call
        self.Z = np.zeros(2 * self.n, dtype=np.float64)
        self.L[...] = 0.0
        self.fn = _lib().oracle_U_NZentries

    def cargs(self, Ncores):
        return (int(Ncores), self.n, int(self.Nlocs), int(self.d), int(self.p), _dptr(self.locs),
                self.nn.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), _dptr(self.cond), _dptr(self.nug), _dptr(self.nugo),
                self.code, _dptr(self.cp), _dptr(self.L), _dptr(self.Z))


def marshal_U_NZentries(n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms):
    """Everything U_NZentries() below does BEFORE the C function is entered, once: bench.py's cpu_baseline times
    `m.fn(*m.cargs(threads))` — the function only — and reports this step separately."""
    return MarshalledSets(n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms)


def U_NZentries(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms):
    """R/RcppExports.R:22-24 -> src/U_NZentries.cpp:25-118 (through the C restatement).

    revNNarray: (Nlocs, p) 1-based with 0 for missing (R/createU.R:146-147).
    Returns dict(Lentries=(Nlocs,p), Zentries=(2n,), n_failed)."""
    m = MarshalledSets(n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms)
    if not m.closed_form:
        # Bessel branch (src/Matern.cpp:72-84): the C restatement has no K_nu, use the numpy one
        return U_NZentries_numpy(n, m.locs, m.nn, m.cond, m.nug, m.nugo, m.cp)
    nf = m.fn(*m.cargs(Ncores))
    if nf < 0:
        raise ValueError(f"{covType} covariance is not implemented")      # src/U_NZentries.cpp:27-29
    return dict(Lentries=m.L, Zentries=m.Z, n_failed=int(nf))


def rows_extended(rows, locs, revNNarray, revCondOnLatent, nuggets, covType, covparms, covVals=None):
    """The adjudicator: rows `rows` (0-based) of Lentries from the definition of src/U_NZentries.cpp:39-69 evaluated in
    extended precision — x87 long double through the C restatement (64-bit significand) for the closed-form covariances
    and the dense variant, 40-digit mpmath for the Bessel branch (src/Matern.cpp:72-84).  Returns (len(rows), p) doubles,
    exact for the given inputs up to ~cond(S) * 1e-19."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    locs = np.asfortranarray(locs, dtype=np.float64)
    Nlocs, d = locs.shape
    nn = np.asfortranarray(np.nan_to_num(np.asarray(revNNarray, dtype=np.float64), nan=0.0).astype(np.int64))
    p = nn.shape[1]
    cond = np.asfortranarray(np.nan_to_num(np.asarray(revCondOnLatent, dtype=np.float64), nan=0.0))
    nug = np.ascontiguousarray(np.broadcast_to(np.asarray(nuggets, dtype=np.float64), (Nlocs,)))
    cp = np.ascontiguousarray(covparms, dtype=np.float64)
    code = {"matern": 0, "esqe": 1}[covType]
    if covVals is None and code == 0 and float(cp[2]) not in (0.5, 1.5, 2.5):
        return _rows_extended_mpmath(rows, locs, nn, cond, nug, cp)
    if _lib().oracle_long_double_digits() < 64:
        raise RuntimeError("long double is not extended precision on this host")
    out = np.zeros((rows.size, p))
    cv = None if covVals is None else np.asfortranarray(covVals, dtype=np.float64)
    _lib().oracle_rows_extended(max_threads(), rows.size, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), Nlocs, d, p,
                                _dptr(locs), nn.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), _dptr(cond), _dptr(nug), code,
                                _dptr(cp), None if cv is None else _dptr(cv), _dptr(out))
    return out


def _rows_extended_mpmath(rows, locs, nn, cond, nug, cp, dps=40):
    import mpmath as mp
    Nlocs, p = nn.shape
    out = np.zeros((len(rows), p))
    with mp.workdps(dps):
        sig2, rng, nu = (mp.mpf(float(v)) for v in cp[:3])
        normcon = sig2 / (mp.mpf(2) ** (nu - 1) * mp.gamma(nu))
        for r, k in enumerate(rows):
            inds = nn[k][nn[k] != 0] - 1
            n0 = len(inds)
            if n0 == 0:
                continue
            S = mp.matrix(n0, n0)
            for a in range(n0):
                for b in range(a, n0):
                    if a == b:
                        S[a, a] = sig2 + mp.mpf(float(nug[inds[a]])) * (1 - mp.mpf(float(cond[k, p - n0 + a])))
                        continue
                    dd = mp.sqrt(sum((mp.mpf(float(locs[inds[a], t])) - mp.mpf(float(locs[inds[b], t]))) ** 2
                                     for t in range(locs.shape[1])))
                    s = dd / rng
                    S[a, b] = S[b, a] = sig2 if dd == 0 else normcon * s ** nu * mp.besselk(nu, s)
            e = mp.matrix(n0, 1)
            e[n0 - 1] = 1
            sol = mp.lu_solve(S, e)
            if not sol[n0 - 1] > 0:
                continue
            out[r, :n0] = [float(v / mp.sqrt(sol[n0 - 1])) for v in sol]
    return out


def U_NZentries_numpy(n, locs, nn, cond, nuggets, nuggets_obsord, covparms):
    """src/U_NZentries.cpp:39-69,111-115 row by row in numpy (any Matern smoothness; small cases only)."""
    from scipy.linalg import solve_triangular
    Nlocs, p = nn.shape
    L = np.zeros((Nlocs, p))
    nfail = 0
    for k in range(Nlocs):
        inds = nn[k][nn[k] != 0] - 1                                   # :44
        n0 = len(inds)
        if n0 == 0:
            continue
        nug = nuggets[inds] * (1.0 - cond[k, p - n0:])                 # :47
        S = MaternFun(rdist(locs[inds]), covparms) + np.diag(nug)      # :48-52
        try:
            Rm = np.linalg.cholesky(S).T                               # :61
            if not np.all(np.isfinite(Rm)):
                raise np.linalg.LinAlgError
            e = np.zeros(n0); e[-1] = 1.0
            L[k, :n0] = solve_triangular(Rm, e)                        # :62-63
        except np.linalg.LinAlgError:
            nfail += 1                                                 # :64-66
    Z = np.zeros(2 * int(n))
    with np.errstate(divide="ignore"):
        Z[0::2] = -1.0 / np.sqrt(nuggets_obsord[: int(n)])
        Z[1::2] = 1.0 / np.sqrt(nuggets_obsord[: int(n)])
    return dict(Lentries=L, Zentries=Z, n_failed=nfail)


def U_NZentries_mat(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covVals, covparms):
    """R/RcppExports.R:26-28 -> src/U_NZentries.cpp:126-197."""
    Nlocs = np.asarray(locs).shape[0]
    nn = np.asfortranarray(np.nan_to_num(np.asarray(revNNarray, dtype=np.float64), nan=0.0).astype(np.int64))
    p = nn.shape[1]
    nugo = np.ascontiguousarray(nuggets_obsord, dtype=np.float64)
    cv = np.asfortranarray(covVals, dtype=np.float64)
    L = np.zeros((Nlocs, p), dtype=np.float64, order="F")
    Z = np.zeros(2 * int(n), dtype=np.float64)
    nf = _lib().oracle_U_NZentries_mat(int(Ncores), int(n), int(Nlocs), int(p),
                                       nn.ctypes.data_as(ctypes.POINTER(ctypes.c_long)), _dptr(nugo), _dptr(cv),
                                       _dptr(L), _dptr(Z))
    return dict(Lentries=np.array(L), Zentries=Z, n_failed=int(nf))


# ----------------------------------------------------------------------------
# covariance functions (numpy; general nu included)
# ----------------------------------------------------------------------------
def MaternFun(distmat, covparms):
    """src/Matern.cpp:24-86, including the general-nu Bessel branch (:72-84).

    Note the general branch has no sqrt(2 nu) scaling of the distance (a quirk
    of the reference that is reproduced, not fixed)."""
    from scipy.special import gamma, kv
    d = np.asarray(distmat, dtype=np.float64)
    sig2, rng, nu = (float(x) for x in covparms[:3])
    out = np.empty_like(d)
    z = d == 0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        s = d / rng
        if nu == 0.5:
            v = sig2 * np.exp(-s)
        elif nu == 1.5:
            v = sig2 * (1 + np.sqrt(3) * s) * np.exp(-np.sqrt(3) * s)
        elif nu == 2.5:
            v = sig2 * np.exp(-s * np.sqrt(5)) * (1 + np.sqrt(5) * s + 5 * s * s / 3)
        else:
            normcon = sig2 / (2.0 ** (nu - 1) * gamma(nu))
            v = normcon * s ** nu * kv(nu, s)
    out[...] = v
    out[z] = sig2
    return out


def EsqeFun(distmat, covparms):
    """src/Esqe.cpp:17-39."""
    d = np.asarray(distmat, dtype=np.float64)
    v = covparms[0] * np.exp(-(d / covparms[1])) + covparms[2] * np.exp(-((d / covparms[3]) ** 2))
    v = np.where(d == 0, covparms[0] + covparms[2], v)
    return v


def rdist(a, b=None):
    a = np.asarray(a, dtype=np.float64)
    b = a if b is None else np.asarray(b, dtype=np.float64)
    ssq = np.zeros((a.shape[0], b.shape[0]))
    for t in range(a.shape[1]):
        ssq += (a[:, t][:, None] - b[:, t][None, :]) ** 2
    return np.sqrt(ssq)


# ----------------------------------------------------------------------------
# orderings (R/ordering_functions.R, src/MaxMin.cpp)
# ----------------------------------------------------------------------------
def order_coordinate(locs, coordinate=None):
    """R/ordering_functions.R:126-128 (R's order() is stable). Returns 1-based."""
    locs = np.asarray(locs, dtype=np.float64)
    cols = list(range(locs.shape[1])) if coordinate is None else list(coordinate)
    return np.argsort(locs[:, cols].sum(axis=1), kind="stable") + 1


def order_maxmin_exact(locs):
    """R/ordering_functions.R:147-150 -> src/MaxMin.cpp:661-738.

    Exact max-min-distance ordering; first point = closest to the centroid,
    strict '<' so the lowest index wins ties (src/MaxMin.cpp:675-707).  The
    reference's heap-based algorithm is quasi-linear; this restatement is the
    O(n^2) definition.  Ties in the max-min distance (regular grids) are broken
    towards the lowest index here; the reference's tie order is an artefact of
    its heap and is unpinned.  Returns 1-based indices."""
    locs = np.asarray(locs, dtype=np.float64)
    n, dim = locs.shape
    avg = np.zeros(dim)
    for i in range(n):
        avg += locs[i]
    avg /= n
    d2 = np.zeros(n)
    for j in range(dim):
        d2 += (locs[:, j] - avg[j]) * (locs[:, j] - avg[j])
    first = int(np.argmin(d2))
    order = [first]
    mind = np.sqrt(((locs - locs[first]) ** 2).sum(axis=1))
    mind[first] = -1.0
    for _ in range(1, n):
        nxt = int(np.argmax(mind))
        order.append(nxt)
        dn = np.sqrt(((locs - locs[nxt]) ** 2).sum(axis=1))
        mind = np.minimum(mind, dn)
        mind[order] = -1.0
    return np.asarray(order, dtype=np.int64) + 1


# ----------------------------------------------------------------------------
# conditioning sets (R/NN_kdtree.R)
# ----------------------------------------------------------------------------
def findOrderedNN(locs, m):
    """R/NN_kdtree.R:73-83 — brute-force ordered nearest neighbours.

    Row j (1-based) = the min(m+1, j) nearest points among locs[1..j] (self
    included, distance 0), ascending distance, R's stable order() => lower index
    wins ties.  NaN-padded on the right.  This is the semantic definition the
    package relies on; for d >= 2 the package calls GpGp::find_ordered_nn
    (R/vecchia_specify.R:159, not vendored), which implements the same
    definition up to tie-breaking."""
    locs = np.asarray(locs, dtype=np.float64)
    n = locs.shape[0]
    NN = np.full((n, m + 1), np.nan)
    for j in range(n):
        dv = rdist(locs[: j + 1], locs[j: j + 1])[:, 0]
        o = np.argsort(dv, kind="stable")[: min(m + 1, j + 1)]
        NN[j, : len(o)] = o + 1
    return NN


def get_knn(x, k):
    """FNN::get.knn(x, k)$nn.index: the k nearest OTHER points of every point, ascending distance (ties: FNN's kd-tree
    order is unspecified; lower index here).  Returns 1-based (n, k)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    out = np.zeros((n, k), dtype=np.int64)
    for i in range(n):
        d = rdist(x, x[i: i + 1])[:, 0]
        d[i] = np.inf                                                   # the point itself is not its own neighbour
        out[i] = np.argsort(d, kind="stable")[:k] + 1
    return out


def order_maxmin_exact_obs_pred(locs, locs_pred):
    """R/ordering_functions.R:174-218, literal (1-based bookkeeping kept)."""
    locs = np.asarray(locs, dtype=np.float64)
    locs_pred = np.asarray(locs_pred, dtype=np.float64)
    ord_ = order_maxmin_exact(locs)                                     # :176
    ord_pred = order_maxmin_exact(locs_pred)                            # :177
    locs_all = np.vstack([locs, locs_pred])                             # :182
    n = locs.shape[0]
    m = int(min(round(np.sqrt(n)), 200))                                # :185 (R rounds half to even, like Python)
    n_pred = locs_pred.shape[0]
    NN = get_knn(locs_all, m)                                           # :189
    NA = -1
    iip = np.concatenate([ord_, n + ord_pred, np.full(n_pred, NA)]).astype(np.int64)   # index_in_position, :192
    poi = np.zeros(n + n_pred + 1, dtype=np.float64)                    # position_of_index (1-based), :193
    poi[iip[: n + n_pred]] = np.arange(1, n + n_pred + 1)
    curlen = n + n_pred                                                 # :197
    nmoved = 0
    for j in range(n + 1, n + 2 * n_pred + 1):                          # :199
        nneigh = int(round(min(m, 1 * (n + n_pred) / (j - nmoved + 1))))   # :203
        idx = iip[j - 1] if j <= len(iip) else NA
        if idx == NA:                                                   # NN[NA, ] is NA: min(..., na.rm = TRUE) = Inf, no move
            continue
        cols = list(range(1, nneigh + 1)) if nneigh >= 1 else [1]       # R: 1:0 == c(1, 0) and index 0 is dropped
        neighbors = NN[idx - 1, [c - 1 for c in cols]]                  # :204
        if np.min(poi[neighbors]) < j:                                  # :205
            nmoved += 1
            curlen += 1
            poi[idx] = curlen
            if curlen > len(iip):                                       # R vectors grow on assignment past the end
                iip = np.concatenate([iip, np.full(curlen - len(iip), NA)])
            iip[curlen - 1] = idx
            iip[j - 1] = NA
    kept = iip[iip != NA]
    return ord_, kept[n: n + n_pred] - n                                # :214


def whichCondOnLatent(NNarray, firstind_pred=None):
    """R/whichCondOnLatent.R:2-26 — the SGV rule, literal (including R's
    is.element(NA, NA) == TRUE semantics on the first rows)."""
    NN = np.asarray(NNarray, dtype=np.float64)
    n, p = NN.shape
    m = p - 1
    if firstind_pred is None:
        firstind_pred = n + 1
    Cond = np.full((n, p), np.nan)
    Cond[0, 0] = 1.0

    def is_element(x, table):
        # R: match() treats NA as matching NA
        res = np.zeros(len(x), dtype=bool)
        tab_has_na = bool(np.any(np.isnan(table)))
        tabvals = set(table[~np.isnan(table)].tolist())
        for i, v in enumerate(x):
            res[i] = tab_has_na if np.isnan(v) else (v in tabvals)
        return res

    for k in range(1, n):
        latents = np.zeros(p)                  # R: rep(0,m) grown to m+1 by assignment
        for ind in range(1, p):
            l = NN[k, ind]
            if not np.isnan(l) and l < firstind_pred:
                li = int(l) - 1
                latents[ind] = np.sum(is_element(NN[k], NN[li] * Cond[li]))
        # latents[0] stays 0 (R vector index 1 is never assigned => 0)
        best = int(np.where(latents == latents.max())[0][0])   # :19 first maximum
        ind = int(NN[k, best]) - 1
        Cond[k] = is_element(NN[k], NN[ind] * Cond[ind]).astype(float)
        with np.errstate(invalid="ignore"):
            Cond[k, NN[k] >= firstind_pred] = 1.0
        Cond[k, 0] = 1.0
        Cond[k, np.isnan(NN[k])] = np.nan
    return Cond


# ----------------------------------------------------------------------------
# U_sparsity / vecchia_specify (R/U_sparsity.R, R/vecchia_specify.R)
# ----------------------------------------------------------------------------
def U_sparsity(locs, NNarray, obs, Cond):
    """R/U_sparsity.R:5-81, literal."""
    NN = np.asarray(NNarray, dtype=np.float64)
    Cond = np.asarray(Cond, dtype=np.float64)
    nnp = np.asarray(locs).shape[0]
    obs = np.asarray(obs, dtype=bool)
    n = int(obs.sum())
    size = nnp + n
    latent_map = np.zeros(nnp, dtype=np.int64)
    observed_map = np.full(nnp, -1, dtype=np.int64)
    cur = 1
    for k in range(nnp):                                  # :19-29
        latent_map[k] = cur
        cur += 1
        if obs[k]:
            observed_map[k] = cur
            cur += 1
    revNN = NN[:, ::-1].copy()                            # :32
    revCond = Cond[:, ::-1].copy()                        # :33
    rowp, coli = [], []
    for k in range(nnp):                                  # :39-56
        inds = revNN[k]
        ok = ~np.isnan(inds)
        inds0 = inds[ok].astype(np.int64) - 1
        rc = revCond[k, ok] == 1.0
        cur_row = latent_map[k]
        cols = np.where(rc, latent_map[inds0], observed_map[inds0])
        rowp.extend([cur_row] * len(inds0))
        coli.extend(cols.tolist())
    Zrow, Zcol = [], []
    for k in range(nnp):                                  # :59-69
        if obs[k]:
            Zrow.extend([observed_map[k]] * 2)
            Zcol.extend([latent_map[k], observed_map[k]])
    return dict(revNNarray=revNN, revCond=revCond, n_cores=os.cpu_count(), size=size,
                rowpointers=np.asarray(rowp + Zrow, dtype=np.int64),
                colindices=np.asarray(coli + Zcol, dtype=np.int64),
                y_ind=latent_map, observed_map=observed_map)


def vecchia_specify(locs, m, ordering=None, cond_yz=None, NNarray=None, locs_pred=None, ordering_pred=None,
                    pred_cond=None):
    """R/vecchia_specify.R:29-240 for conditioning='NN' (no MRA), with or without prediction locations.

    ordering in {'none','coord','maxmin'}; cond_yz in {'z','y','SGV','SGVT','zy','RVP','LK'}.
    NNarray may be supplied (1-based, NaN padded) to bypass the NN search (no-prediction case only)."""
    locs = np.asarray(locs, dtype=np.float64)
    n, dim = locs.shape
    have_pred = locs_pred is not None
    if have_pred:
        locs_pred = np.asarray(locs_pred, dtype=np.float64)
        la = np.vstack([locs, locs_pred])                               # :47-51
        if len({tuple(r) for r in la.tolist()}) < la.shape[0]:
            raise ValueError("Prediction locations contain observed location(s), remove redundancies.")
    if m > n:                                                       # :53-56
        m = n - 1
    if ordering is None:                                            # :83-85
        ordering = "coord" if dim == 1 else "maxmin"
    if pred_cond is None:                                           # :86
        pred_cond = "general"
    if cond_yz is None:                                             # :92-96
        cond_yz = "SGV" if (not have_pred or dim == 1) else "zy"
    if not have_pred:                                               # :100-117
        if ordering == "coord":                                         # :102
            ord_ = order_coordinate(locs)
        elif ordering == "maxmin":                                      # :103-106
            o = order_maxmin_exact(locs)
            cut = min(n, 9)
            ord_ = np.concatenate([o[:1], o[cut:], o[1:cut]])
        elif ordering == "none":                                        # :109-110
            ord_ = np.arange(1, n + 1)
        else:
            raise ValueError(ordering)
        ord_z = ord_.copy()
        locsord = locs[ord_ - 1]
        obs = np.ones(n, dtype=bool)
        ordering_pred = "general"
        n_p = 0
    else:                                                           # :119-149
        n_p = locs_pred.shape[0]
        locs_all = np.vstack([locs, locs_pred])
        observed_obspred = np.concatenate([np.ones(n, bool), np.zeros(n_p, bool)])
        if ordering_pred is None:                                       # :124-126
            ordering_pred = "general" if (dim == 1 and ordering == "coord") else "obspred"
        if ordering_pred == "general":                                  # :127-131
            ord_ = order_coordinate(locs_all) if ordering == "coord" else order_maxmin_exact(locs_all)
            ord_obs = ord_[ord_ <= n]
        else:                                                           # :132-145
            if ordering == "coord":
                ord_obs = order_coordinate(locs)
                ord_pred = order_coordinate(locs_pred)
            elif ordering == "none":
                ord_obs = np.arange(1, n + 1)
                ord_pred = np.arange(1, n_p + 1)
            else:
                ord_obs, ord_pred = order_maxmin_exact_obs_pred(locs, locs_pred)
            ord_ = np.concatenate([ord_obs, ord_pred + n])
        ord_z = ord_obs
        locsord = locs_all[ord_ - 1]
        obs = observed_obspred[ord_ - 1]
    if NNarray is None:
        NNarray = findOrderedNN(locsord, m)                         # :157-159 (semantic twin)
    NNarray = np.asarray(NNarray, dtype=np.float64)
    if have_pred and pred_cond == "independent":                    # :168-178
        if ordering_pred == "obspred":
            for j in range(1, n_p + 1):
                dists = rdist(locsord[n + j - 1: n + j], locsord[:n])[0]
                m_nearest = np.sort(np.argsort(dists, kind="stable")[:m] + 1)[::-1]
                NNarray[n + j - 1] = np.concatenate([[n + j], m_nearest])
    if cond_yz == "SGV":                                            # :182-183
        Cond = whichCondOnLatent(NNarray, firstind_pred=n + 1)
    elif cond_yz == "SGVT":                                         # :184-185
        Cond = np.vstack([whichCondOnLatent(NNarray[:n]), np.ones((n_p, m + 1))])
    elif cond_yz == "y":                                            # :186-187
        Cond = np.full(NNarray.shape, np.nan)
        Cond[~np.isnan(NNarray)] = 1.0
    elif cond_yz == "z":                                            # :189-190
        Cond = np.full(NNarray.shape, np.nan)
        Cond[~np.isnan(NNarray)] = 0.0
        Cond[:, 0] = 1.0
    elif cond_yz in ("RVP", "LK", "zy"):                            # :191-223 response-latent trick
        obs = np.concatenate([np.ones(n, bool), np.zeros(locsord.shape[0], bool)])     # :195
        locsord = np.vstack([locsord[:n], locsord])                     # :196
        NNs = get_knn(locsord[:n], m - 1).astype(np.float64)            # :199
        if cond_yz in ("RVP", "zy"):                                    # :200-203
            prev = NNs < np.arange(1, n + 1)[:, None]
            NNs[prev] += n
        NN_z = np.hstack([np.arange(1, n + 1)[:, None].astype(float), np.full((n, m), np.nan)])      # :206
        NN_y = np.hstack([(np.arange(1, n + 1) + n)[:, None], np.arange(1, n + 1)[:, None], NNs])   # :207
        if not have_pred:                                               # :208-210
            NN_yp = np.zeros((0, m + 1))
            ordering_pred = "obspred"
        else:
            if cond_yz == "zy":                                         # :213-214
                NN_yp = NNarray[n: n + n_p] + n
            else:                                                       # :215-218
                NN_yp = NNarray[n: n + n_p].copy()
                big = NN_yp > n
                NN_yp[big] += n
        NNarray = np.vstack([NN_z, NN_y, NN_yp])                        # :220
        with np.errstate(invalid="ignore"):
            Cond = (NNarray > n).astype(np.float64)                     # :223 (NA > n is NA)
        Cond[np.isnan(NNarray)] = np.nan
        Cond[:, 0] = 1.0
        cond_yz = "zy"
    else:
        raise ValueError(cond_yz)
    U_prep = U_sparsity(locsord, NNarray, obs, Cond)                # :230
    return dict(locsord=locsord, obs=obs, ord=ord_, ord_z=ord_z, ord_pred=ordering_pred,
                U_prep=U_prep, cond_yz=cond_yz, ic0=False, conditioning="NN")


# ----------------------------------------------------------------------------
# createU / vecchia_likelihood (R/createU.R, R/vecchia_likelihood.R, R/vecchia_prediction.R)
# ----------------------------------------------------------------------------
def createU(va, covparms, nuggets, covmodel="matern"):
    """R/createU.R:65-86,141-163,195-199 — NN branch, non-zy, nuggets > 0.

    Returns dict with dense U (size x size), latent mask, ord_z, plus the raw
    U_entries for parity tests."""
    n = int(np.sum(va["obs"]))
    prep = va["U_prep"]
    size = prep["size"]
    latent = np.isin(np.arange(1, size + 1), prep["y_ind"])
    ord_ = va["ord"]
    nug = np.asarray(nuggets, dtype=np.float64)
    if nug.size == 1:                                               # :74
        nug = np.repeat(nug, n)
    nuggets_all = np.concatenate([nug, np.zeros(int(latent.sum()) - n)])   # :75
    ord_all = np.concatenate([ord_[:n], ord_ + n]) if va["cond_yz"] == "zy" else ord_   # :76
    nuggets_all_ord = nuggets_all[ord_all - 1]                      # :77
    nuggets_ord = nuggets_all[va["ord_z"] - 1]                      # :78
    revNN = prep["revNNarray"].copy()
    revCond = prep["revCond"].copy()
    if np.any(nug == 0):                                            # :83-86
        zero_idx = np.where(nuggets_ord == 0)[0] + 1
        revCond[np.isin(revNN, zero_idx)] = 1.0
    revNN0 = np.nan_to_num(revNN, nan=0.0)                          # :146-147
    if isinstance(covmodel, str):
        ent = U_NZentries(prep["n_cores"], n, va["locsord"], revNN0, revCond, nuggets_all_ord,
                          nuggets_ord, covmodel, covparms)          # :152-154
    else:
        ent = U_NZentries_mat(prep["n_cores"], n, va["locsord"], revNN0, revCond, nuggets_all_ord,
                              nuggets_ord, np.asarray(covmodel), covparms)   # :149-151
    # :158-159 — row-major walk of Lentries keeping the first n0 entries of each row
    L = ent["Lentries"]
    n0 = (~np.isnan(revNN)).sum(axis=1)
    vals = np.concatenate([L[k, : n0[k]] for k in range(L.shape[0])] + [ent["Zentries"]])   # :160
    U = np.zeros((size, size))
    # :161-162 sparseMatrix(i=colindices, j=rowpointers, x=...) (duplicates would be summed)
    np.add.at(U, (prep["colindices"] - 1, prep["rowpointers"] - 1), vals)
    obs = va["obs"]
    if va["cond_yz"] == "zy":                                       # :166-171 rows/columns of the dummy y's
        dummy = 2 * np.arange(1, n + 1) - 1
        keepd = np.ones(size, dtype=bool)
        keepd[dummy - 1] = False
        U = U[np.ix_(keepd, keepd)]
        latent = latent[keepd]
        obs = np.delete(obs, np.arange(n, 2 * n))
        size = int(keepd.sum())
    zero_nugg = {}
    if np.any(nug == 0):                                            # :173-193, literal
        inds_U = np.where(np.isinf(np.diag(U)))[0]                  # :178
        cond_on = np.array([np.where(U[:, j] != 0)[0].min() for j in inds_U])     # :179
        keep = np.ones(size, dtype=bool)
        keep[inds_U] = False
        U = U[np.ix_(keep, keep)]                                   # :180
        inds_z = np.where(np.isin(np.where(~latent)[0], inds_U))[0]               # :183
        inds_locs = np.where(np.isin(np.where(latent)[0], cond_on))[0]            # :184
        zero_nugg = dict(inds_U=inds_U + 1, inds_z=inds_z + 1, inds_locs=inds_locs + 1)
        latent = latent.copy()
        latent[cond_on] = False                                     # :188
        latent = latent[keep]                                       # :189
        rest = np.setdiff1d(np.arange(len(ord_)), inds_locs)
        ord_ = np.concatenate([ord_[rest], ord_[inds_locs]])        # :190
        obs = np.concatenate([obs[rest], obs[inds_locs]])           # :191
    return dict(U=U, latent=latent, ord=ord_, obs=obs, ord_z=va["ord_z"], ord_pred=va["ord_pred"],
                cond_yz=va["cond_yz"], ic0=va["ic0"], U_entries=ent, zero_nugg=zero_nugg,
                triplets=(prep["colindices"].copy(), prep["rowpointers"].copy(), vals))


def ic0(ptrs, inds, vals):
    """src/ic0.cpp:43-64 (with dot_prod, :16-31): zero-fill incomplete Cholesky on a lower triangle in compressed-row
    form (indices ascending, diagonal last), literal loops."""
    vals = np.array(vals, dtype=np.float64)
    N = len(ptrs) - 1
    for i in range(N):
        for j in range(ptrs[i], ptrs[i + 1]):
            l1, u1 = ptrs[i], ptrs[i + 1] - 2
            l2, u2 = ptrs[inds[j]], ptrs[inds[j] + 1] - 2
            dp = 0.0
            while l1 <= u1 and l2 <= u2:                            # dot_prod
                if inds[l1] == inds[l2]:
                    dp += vals[l1] * vals[l2]
                    l1 += 1; l2 += 1
                elif inds[l1] < inds[l2]:
                    l1 += 1
                else:
                    l2 += 1
            if inds[j] < i:
                vals[j] = (vals[j] - dp) / vals[ptrs[inds[j] + 1] - 1]
            else:
                vals[j] = np.sqrt(vals[j] - dp)
    return vals


def ichol(M):
    """R/ichol.R:16-59 without a pattern matrix: IC(0) on the pattern of M; returns the UPPER factor (dense)."""
    M = np.asarray(M, dtype=np.float64)
    n = M.shape[0]
    ptrs, inds, vals = [0], [], []
    for c in range(n):                                              # upper triangle by columns == lower by rows
        for r in range(c + 1):
            if M[r, c] != 0.0:
                inds.append(r); vals.append(M[r, c])
        ptrs.append(len(inds))
    v = ic0(ptrs, inds, vals)
    R_ = np.zeros((n, n))
    for c in range(n):
        for p in range(ptrs[c], ptrs[c + 1]):
            R_[inds[p], c] = v[p]
    return R_


def revMat(M):
    """R/vecchia_likelihood.R:103."""
    return M[::-1, ::-1]


def U2V(U_obj):
    """R/vecchia_prediction.R:62-111: V = t(chol(rev(U_y U_y^T))) (or t(ichol(.)) with ic0 = TRUE, :76-77); for 'zy' the
    latent block of U reversed (:68-70); for obs-pred ordering the prediction columns unchanged and a Cholesky of the
    observed block only (:85-107)."""
    U = U_obj["U"]
    latent = U_obj["latent"]
    Uy = U[latent, :]
    chol_rev = (lambda A: ichol(revMat(A)).T) if U_obj.get("ic0", False) else (lambda A: np.linalg.cholesky(revMat(A)))
    if U_obj["cond_yz"] == "zy":                                        # :68-70
        return revMat(Uy[:, latent])
    if U_obj["ord_pred"] != "obspred":                                  # :72-83
        return chol_rev(Uy @ Uy.T)
    last_obs = int(np.max(np.where(~latent)[0])) + 1                    # :87 (1-based position)
    latents_before = int(latent[:last_obs].sum())                       # :88
    latents_after = int(latent[last_obs:].sum())                        # :89
    V_pr = revMat(Uy[:, last_obs:])                                     # :92
    U_oo = Uy[:latents_before, :last_obs]                               # :95
    V_oor = chol_rev(U_oo @ U_oo.T)                                     # :96-100
    V_or = np.vstack([np.zeros((latents_after, latents_before)), V_oor])   # :103-104
    return np.hstack([V_pr, V_or])                                      # :106


def vecchia_likelihood_U(z, U_obj):
    """R/vecchia_likelihood.R:63-99."""
    from scipy.linalg import solve_triangular
    U = U_obj["U"]
    latent = U_obj["latent"]
    zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]      # :68
    const = np.sum(~latent) * np.log(2 * np.pi)                     # :71
    z1 = U[~latent, :].T @ zord                                     # :74
    quadform_num = np.sum(z1 ** 2)                                  # :75
    logdet_num = -2 * np.sum(np.log(np.diag(U)))                    # :76
    if latent.sum() == 0:                                           # :79-81
        logdet_denom = quadform_denom = 0.0
    else:
        z2 = U[latent, :] @ z1                                      # :85-86
        V = U2V(U_obj)                                              # :87
        z3 = solve_triangular(V, z2[::-1], lower=True)              # :88
        quadform_denom = np.sum(z3 ** 2)                            # :89
        logdet_denom = -2 * np.sum(np.log(np.diag(V)))              # :90
    neg2loglik = logdet_num - logdet_denom + quadform_num - quadform_denom + const   # :95
    return -neg2loglik / 2                                          # :96


def removeNAs(z, nuggets):
    """R/vecchia_likelihood.R:45-58: missing data get the mean of the rest and a nugget of var * 1e8 (stats::var: n-1)."""
    z = np.array(z, dtype=np.float64)
    nug = np.atleast_1d(np.array(nuggets, dtype=np.float64))
    na = np.isnan(z)
    if na.any():                                                        # :47
        if nug.size < z.size:                                           # :49-53
            new = np.zeros(z.size)
            new[~na] = nug
            nug = new
        nug[na] = np.var(z[~na], ddof=1) * 1e8                          # :55
        z[na] = np.mean(z[~na])                                         # :56
    return z, nug


def vecchia_likelihood(z, va, covparms, nuggets, covmodel="matern"):
    """R/vecchia_likelihood.R:14-27."""
    z, nuggets = removeNAs(z, nuggets)                                  # :20
    return vecchia_likelihood_U(z, createU(va, covparms, nuggets, covmodel))


def separable_loglik_condz(va, U_entries, z, nuggets):
    """Closed form of vecchia_likelihood_U when cond.yz == 'z' (W = U_y U_y^T is
    diagonal): derived from R/vecchia_likelihood.R:63-99, R/vecchia_prediction.R:74.
    Used to pin the device epilogue; returns (loglik, partial sums)."""
    prep = va["U_prep"]
    revNN = prep["revNNarray"]
    revCond = prep["revCond"]
    L = U_entries["Lentries"]
    n, p = revNN.shape
    zord = np.asarray(z, dtype=np.float64)[va["ord_z"] - 1]
    tau = np.asarray(nuggets, dtype=np.float64)
    tau = np.repeat(tau, n) if tau.size == 1 else tau[va["ord_z"] - 1]
    s = np.zeros(6)
    for k in range(n):
        ok = ~np.isnan(revNN[k])
        n0 = int(ok.sum())
        idx = revNN[k, ok].astype(np.int64) - 1
        c = revCond[k, ok]
        M = L[k, :n0]
        d = M[n0 - 1]
        a = float(np.sum(M[: n0 - 1] * zord[idx[: n0 - 1]] * (c[: n0 - 1] == 0)))
        w = d * d + 1.0 / tau[k]
        s[0] += np.log(d)
        s[1] += np.log(tau[k])
        s[2] += np.log(w)
        s[3] += a * a
        s[4] += zord[k] ** 2 / tau[k]
        s[5] += (d * a - zord[k] / tau[k]) ** 2 / w
    loglik = -0.5 * (-2 * s[0] + s[1] + s[2] + s[3] + s[4] - s[5] + n * np.log(2 * np.pi))
    return loglik, s


def separable_sums_condz_vectorised(revNN, Lentries, zord, tau):
    """The sums of separable_loglik_condz for cond.yz == 'z' plans at full size (n = 1e6 in seconds): the same per-row
    quantities, rows with n0 == p in one vectorised pass, the first rows one by one.  revNN: (n, p) ints, 0 = missing,
    right-aligned; Lentries: (n, p) left-aligned; tau scalar or (n,).  Returns (loglik, s[6]) like separable_loglik_condz."""
    revNN = np.asarray(revNN)
    n, p = revNN.shape
    L = np.asarray(Lentries)
    zord = np.asarray(zord, dtype=np.float64)
    tau = np.broadcast_to(np.asarray(tau, dtype=np.float64), (n,))
    n0 = (revNN != 0).sum(axis=1)
    d = L[np.arange(n), n0 - 1]
    a = np.zeros(n)
    full = n0 == p
    a[full] = np.einsum("ij,ij->i", L[full, : p - 1], zord[revNN[full, : p - 1] - 1])
    for k in np.where(~full)[0]:
        a[k] = L[k, : n0[k] - 1] @ zord[revNN[k, p - n0[k]: p - 1] - 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = d * d + 1.0 / tau
        s = np.array([np.log(d).sum(), np.log(tau).sum(), np.log(w).sum(), (a * a).sum(), (zord ** 2 / tau).sum(),
                      ((d * a - zord / tau) ** 2 / w).sum()])
    loglik = -0.5 * (-2 * s[0] + s[1] + s[2] + s[3] + s[4] - s[5] + n * np.log(2 * np.pi))
    return loglik, s


# ----------------------------------------------------------------------------
# posterior mean and Vecchia-Laplace (R/vecchia_prediction.R, R/vecchia_laplace_NR.R), dense restatements
# ----------------------------------------------------------------------------
def vecchia_mean(z, U_obj, V, both=False):
    """R/vecchia_prediction.R:118-142: mu.obs in original order (and mu.pred with both=True)."""
    from scipy.linalg import solve_triangular
    U = U_obj["U"]
    latent = U_obj["latent"]
    zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]        # :121
    z1 = U[~latent, :].T @ zord                                        # :122
    z2 = U[latent, :] @ z1                                             # :123
    temp = solve_triangular(V, z2[::-1], lower=True)                   # :124
    mu_rev = -solve_triangular(V.T, temp, lower=False)                 # :125
    mu_ord = mu_rev[::-1]                                              # :126
    if len(U_obj["zero_nugg"]) > 0:                                    # :129-132 for zero nugget, observations are posterior means
        obs_zero = zord[U_obj["zero_nugg"]["inds_z"] - 1]
        mu_ord = np.concatenate([mu_ord, obs_zero])
    orig_order = np.argsort(U_obj["ord"], kind="stable")               # :135
    mu = mu_ord[orig_order]                                            # :136
    obs_orig = np.asarray(U_obj["obs"])[orig_order]                    # :137
    if both:
        return mu[obs_orig], mu[~obs_orig]                             # :138-139
    return mu[obs_orig]


def vecchia_prediction_mean(z, va, covparms, nuggets, covmodel="matern", both=False):
    """R/vecchia_prediction.R:17-56 with return.values='meanmat' (mean only)."""
    z, nuggets = removeNAs(z, nuggets)                                 # :22
    U_obj = createU(va, covparms, nuggets, covmodel)                   # :25
    V = U2V(U_obj)                                                     # :28
    return vecchia_mean(z, U_obj, V, both)                             # :34


# ----------------------------------------------------------------------------
# The same R callers on SPARSE matrices, for sizes the dense restatements above cannot hold (n = 1e6): what the reference
# itself does (Matrix::sparseMatrix, tcrossprod, chol, solve).  Pinned to the dense functions in tests/test_oracle.py.
# ----------------------------------------------------------------------------
def U_triplets_vectorised(revNN, revCond, obs):
    """R/U_sparsity.R:19-73 without the loops: (rowpointers, colindices, latent_map, observed_map, size), 1-based, in the
    order the literal U_sparsity above produces them (rows of revNNarray one after the other, valid entries left to
    right, then the Z pairs)."""
    revNN = np.asarray(revNN, dtype=np.float64)
    revCond = np.asarray(revCond, dtype=np.float64)
    obs = np.asarray(obs, dtype=bool)
    nnp = revNN.shape[0]
    n = int(obs.sum())
    pos = np.arange(nnp, dtype=np.int64) + np.concatenate([[0], np.cumsum(obs)[:-1]])   # :19-29 y_k, then z_k if observed
    latent_map = pos + 1
    observed_map = np.where(obs, pos + 2, -1)
    ok = ~np.isnan(revNN)
    k_idx = np.nonzero(ok)[0]
    inds0 = revNN[ok].astype(np.int64) - 1
    rc = revCond[ok] == 1.0
    coli = np.where(rc, latent_map[inds0], observed_map[inds0])         # :44-50
    if np.any(coli < 0):                                                # observed_map is NA there: sparseMatrix() stops
        raise ValueError("U_sparsity: a location without an observation is conditioned on as observed")
    rowp = latent_map[k_idx]
    ko = np.where(obs)[0]
    Zrow = np.repeat(observed_map[ko], 2)                               # :59-69
    Zcol = np.stack([latent_map[ko], observed_map[ko]], axis=1).reshape(-1)
    return (np.concatenate([rowp, Zrow]), np.concatenate([coli, Zcol]), latent_map, observed_map, nnp + n)


def createU_sparse(va, covparms, nuggets, covmodel="matern", U_entries=None):
    """createU above with U as a scipy.sparse CSC matrix (R/createU.R:65-86,141-171; positive nuggets only: the
    zero-nugget surgery of :173-193 stays with the dense restatement).  U_entries: a U_NZentries result to reuse."""
    import scipy.sparse as sp
    n = int(np.sum(va["obs"]))
    prep = va["U_prep"]
    revNN = np.asarray(prep["revNNarray"], dtype=np.float64)
    revCond = np.asarray(prep["revCond"], dtype=np.float64)
    rowp, coli, latent_map, _, size = U_triplets_vectorised(revNN, revCond, va["obs"])
    latent = np.zeros(size, dtype=bool)
    latent[latent_map - 1] = True
    ord_ = va["ord"]
    nug = np.asarray(nuggets, dtype=np.float64)
    if nug.size == 1:                                               # :74
        nug = np.repeat(nug, n)
    if np.any(nug == 0):
        raise ValueError("createU_sparse: zero nuggets are handled by the dense createU only")
    nuggets_all = np.concatenate([nug, np.zeros(int(latent.sum()) - n)])   # :75
    ord_all = np.concatenate([ord_[:n], ord_ + n]) if va["cond_yz"] == "zy" else ord_   # :76
    nuggets_all_ord = nuggets_all[ord_all - 1]                      # :77
    nuggets_ord = nuggets_all[va["ord_z"] - 1]                      # :78
    if U_entries is None:
        U_entries = U_NZentries(prep.get("n_cores", max_threads()), n, va["locsord"], np.nan_to_num(revNN, nan=0.0),
                                revCond, nuggets_all_ord, nuggets_ord, covmodel, covparms)          # :146-154
    L = np.ascontiguousarray(U_entries["Lentries"])
    n0 = (~np.isnan(revNN)).sum(axis=1)
    keep = np.arange(revNN.shape[1])[None, :] < n0[:, None]         # :158-159 first n0 entries of each row, row-major
    vals = np.concatenate([L[keep], U_entries["Zentries"]])         # :160
    U = sp.coo_matrix((vals, (coli - 1, rowp - 1)), shape=(size, size)).tocsc()   # :161-162 (duplicates are summed)
    obs = np.asarray(va["obs"], dtype=bool)
    if va["cond_yz"] == "zy":                                       # :166-171
        keepd = np.ones(size, dtype=bool)
        keepd[2 * np.arange(n)] = False
        U = U[keepd][:, keepd].tocsc()
        latent = latent[keepd]
        obs = np.delete(obs, np.arange(n, 2 * n))
    return dict(U=U, latent=latent, ord=ord_, obs=obs, ord_z=va["ord_z"], ord_pred=va["ord_pred"],
                cond_yz=va["cond_yz"], ic0=va.get("ic0", False), U_entries=U_entries, zero_nugg={})


def sparse_chol_lower(A):
    """The lower Cholesky factor of a sparse symmetric positive definite matrix in the natural order, = t(Matrix::chol(A))
    with Matrix's default pivot = FALSE (R/vecchia_prediction.R:80), through oracle/sparse_chol_oracle.c.
    Returns a scipy.sparse CSC matrix whose columns hold the diagonal first, rows ascending."""
    import scipy.sparse as sp
    Au = sp.triu(A, format="csc")
    Au.sort_indices()
    n = Au.shape[0]
    lp = ctypes.POINTER(ctypes.c_long)
    Ap = np.ascontiguousarray(Au.indptr, dtype=np.int64)
    Ai = np.ascontiguousarray(Au.indices, dtype=np.int64)
    ext = Au.dtype == np.longdouble                                 # the adjudicator's instance (x87 extended precision)
    real = np.longdouble if ext else np.float64
    rp = (lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_longdouble))) if ext else _dptr
    Ax = np.ascontiguousarray(Au.data, dtype=real)
    parent = np.empty(max(n, 1), dtype=np.int64)
    cnt = np.empty(max(n, 1), dtype=np.int64)
    P = lambda a: a.ctypes.data_as(lp)
    nnz = _lib().oracle_sparse_chol_symbolic(n, P(Ap), P(Ai), P(parent), P(cnt))
    if nnz < 0:
        raise MemoryError("oracle_sparse_chol_symbolic")
    Lp = np.concatenate([[0], np.cumsum(cnt[:n])]).astype(np.int64)
    Li = np.empty(max(nnz, 1), dtype=np.int64)
    Lx = np.empty(max(nnz, 1), dtype=real)
    fn = _lib().oracle_sparse_chol_numeric_ld if ext else _lib().oracle_sparse_chol_numeric
    st = fn(n, P(Ap), P(Ai), rp(Ax), P(parent), P(Lp), P(Li), rp(Lx))
    if st != 0:
        raise np.linalg.LinAlgError(f"sparse_chol_lower: pivot {st} is not positive")
    return sp.csc_matrix((Lx[:nnz], Li[:nnz], Lp), shape=(n, n))


def _tri_solve(V, b, transpose=False):
    """Matrix::solve(V, b) / solve(t(V), b) for the lower-triangular CSC factor of sparse_chol_lower (or any lower
    triangular CSC matrix: it is brought to the diagonal-first column layout)."""
    import scipy.sparse as sp
    V = sp.csc_matrix(V)
    V.sort_indices()                                                # lower triangular + sorted rows => diagonal first
    n = V.shape[0]
    lp = ctypes.POINTER(ctypes.c_long)
    Lp = np.ascontiguousarray(V.indptr, dtype=np.int64)
    Li = np.ascontiguousarray(V.indices, dtype=np.int64)
    ext = V.dtype == np.longdouble
    real = np.longdouble if ext else np.float64
    rp = (lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_longdouble))) if ext else _dptr
    Lx = np.ascontiguousarray(V.data, dtype=real)
    if n and not np.array_equal(Li[Lp[:-1]], np.arange(n)):
        raise ValueError("_tri_solve: the matrix is not lower triangular with a full diagonal")
    x = np.array(b, dtype=real).copy()
    if ext:
        fn = _lib().oracle_sparse_ltsolve_ld if transpose else _lib().oracle_sparse_lsolve_ld
    else:
        fn = _lib().oracle_sparse_ltsolve if transpose else _lib().oracle_sparse_lsolve
    fn(n, Lp.ctypes.data_as(lp), Li.ctypes.data_as(lp), rp(Lx), rp(x))
    return x


def _rev_sparse(M):
    """revMat (R/vecchia_likelihood.R:103) of a sparse matrix."""
    import scipy.sparse as sp
    M = sp.coo_matrix(M)
    return sp.coo_matrix((M.data, (M.shape[0] - 1 - M.row, M.shape[1] - 1 - M.col)), shape=M.shape).tocsc()


def U2V_sparse(U_obj):
    """U2V above on sparse matrices (R/vecchia_prediction.R:62-111, ic0 = FALSE): a lower-triangular CSC V.ord."""
    import scipy.sparse as sp
    if U_obj.get("ic0", False):
        raise ValueError("U2V_sparse: ic0 = TRUE is restated densely only (ichol above)")
    U = sp.csr_matrix(U_obj["U"])
    latent = np.asarray(U_obj["latent"], dtype=bool)
    Uy = U[np.where(latent)[0], :]                                      # :66
    chol_rev = lambda A: sparse_chol_lower(_rev_sparse(A))
    if U_obj["cond_yz"] == "zy":                                        # :68-70
        return _rev_sparse(Uy.tocsc()[:, np.where(latent)[0]])
    if U_obj["ord_pred"] != "obspred":                                  # :72-83
        return chol_rev(Uy @ Uy.T)
    last_obs = int(np.max(np.where(~latent)[0])) + 1                    # :87
    latents_before = int(latent[:last_obs].sum())                       # :88
    latents_after = int(latent[last_obs:].sum())                        # :89
    V_pr = _rev_sparse(Uy.tocsc()[:, last_obs:])                        # :92
    U_oo = Uy[:latents_before, :][:, :last_obs]                         # :95
    V_oor = chol_rev(U_oo @ U_oo.T)                                     # :96-100
    V_or = sp.vstack([sp.csc_matrix((latents_after, latents_before)), V_oor])   # :103-104
    return sp.hstack([V_pr, V_or]).tocsc()                              # :106


def vecchia_likelihood_U_sparse(z, U_obj, V=None, terms=False):
    """vecchia_likelihood_U above on a sparse U (R/vecchia_likelihood.R:63-99).  With terms=True also returns
    dict(logdet_num, quadform_num, logdet_denom, quadform_denom, V)."""
    import scipy.sparse as sp
    U = sp.csr_matrix(U_obj["U"])
    latent = np.asarray(U_obj["latent"], dtype=bool)
    zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]      # :68
    const = np.sum(~latent) * np.log(2 * np.pi)                     # :71
    z1 = U[np.where(~latent)[0], :].T @ zord                        # :74
    quadform_num = float(np.sum(z1 ** 2))                           # :75
    logdet_num = -2 * float(np.sum(np.log(U.diagonal())))           # :76
    if latent.sum() == 0:                                           # :79-81
        logdet_denom = quadform_denom = 0.0
    else:
        z2 = U[np.where(latent)[0], :] @ z1                         # :85-86
        if V is None:
            V = U2V_sparse(U_obj)                                   # :87
        z3 = _tri_solve(V, z2[::-1])                                # :88
        quadform_denom = float(np.sum(z3 ** 2))                     # :89
        logdet_denom = -2 * float(np.sum(np.log(V.diagonal())))     # :90
    neg2loglik = logdet_num - logdet_denom + quadform_num - quadform_denom + const   # :95
    if terms:
        return -neg2loglik / 2, dict(logdet_num=logdet_num, quadform_num=quadform_num, logdet_denom=logdet_denom,
                                     quadform_denom=quadform_denom, V=V)
    return -neg2loglik / 2                                          # :96


def vecchia_mean_sparse(z, U_obj, V, both=False, ordered=False):
    """vecchia_mean above on sparse matrices (R/vecchia_prediction.R:118-142); ordered=True returns mu.ord (:126)."""
    import scipy.sparse as sp
    U = sp.csr_matrix(U_obj["U"])
    latent = np.asarray(U_obj["latent"], dtype=bool)
    zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]        # :121
    z1 = U[np.where(~latent)[0], :].T @ zord                          # :122
    z2 = U[np.where(latent)[0], :] @ z1                               # :123
    temp = _tri_solve(V, z2[::-1])                                    # :124
    mu_rev = -_tri_solve(V, temp, transpose=True)                     # :125
    mu_ord = mu_rev[::-1]                                             # :126
    if ordered:
        return mu_ord
    orig_order = np.argsort(U_obj["ord"], kind="stable")              # :135
    mu = mu_ord[orig_order]                                           # :136
    obs_orig = np.asarray(U_obj["obs"], dtype=bool)[orig_order]       # :137
    if both:
        return mu[obs_orig], mu[~obs_orig]                            # :138-139
    return mu[obs_orig]


def vecchia_prediction_mean_sparse(z, va, covparms, nuggets, covmodel="matern", both=False):
    """vecchia_prediction_mean above on sparse matrices (R/vecchia_prediction.R:17-56, return.values='meanmat')."""
    z, nuggets = removeNAs(z, nuggets)                                 # :22
    U_obj = createU_sparse(va, covparms, nuggets, covmodel)            # :25
    V = U2V_sparse(U_obj)                                              # :28
    return vecchia_mean_sparse(z, U_obj, V, both)                      # :34


def vecchia_likelihood_sparse(z, va, covparms, nuggets, covmodel="matern"):
    """vecchia_likelihood above on sparse matrices (R/vecchia_likelihood.R:14-27)."""
    z, nuggets = removeNAs(z, nuggets)                                 # :20
    return vecchia_likelihood_U_sparse(z, createU_sparse(va, covparms, nuggets, covmodel))


def posterior_extended(z, va, covparms, nuggets, covmodel="matern"):
    """The adjudicator for the posterior pass: the chain createU -> U2V -> vecchia_likelihood_U / vecchia_mean
    (R/createU.R:141-171, R/vecchia_prediction.R:62-83,118-126, R/vecchia_likelihood.R:63-99; general ordering, positive
    nuggets, cond.yz != 'zy') in x87 extended precision: U entries from rows_extended (rounded to double: 1e-16), W = U_y
    U_y^T, the Cholesky factor, both triangular solves and every sum in long double.  Exact for the given inputs to
    ~cond(W) * 1e-19.  Returns dict(mu_ord, loglik, logdet_num, quadform_num, logdet_denom, quadform_denom) as doubles."""
    import scipy.sparse as sp
    ld = np.longdouble
    prep = va["U_prep"]
    n = int(np.sum(va["obs"]))
    if va["cond_yz"] == "zy" or va["ord_pred"] == "obspred" or n != va["locsord"].shape[0]:
        raise ValueError("posterior_extended: plans without prediction locations, cond.yz in {'SGV','y','z'}")
    nug = np.asarray(nuggets, dtype=np.float64)
    if nug.size == 1:
        nug = np.repeat(nug, n)
    nuggets_ord = nug[va["ord_z"] - 1]
    Nl = va["locsord"].shape[0]
    Lx = rows_extended(np.arange(Nl), va["locsord"], prep["revNNarray"], prep["revCond"], nug[va["ord"] - 1], covmodel,
                       covparms)
    with np.errstate(divide="ignore"):
        zd = (1.0 / np.sqrt(nuggets_ord.astype(ld)))
    Zx = np.stack([-zd, zd], axis=1).reshape(-1).astype(np.float64)     # src/U_NZentries.cpp:111-115
    Us = createU_sparse(va, covparms, nuggets, covmodel, U_entries=dict(Lentries=Lx, Zentries=Zx))
    U = sp.csr_matrix(Us["U"]).astype(ld)
    latent = np.asarray(Us["latent"], dtype=bool)
    zord = np.asarray(z, dtype=np.float64)[va["ord_z"] - 1].astype(ld)
    z1 = U[np.where(~latent)[0], :].T @ zord
    quadform_num = np.sum(z1 * z1)
    logdet_num = -2 * np.sum(np.log(U.diagonal()))
    Uy = U[np.where(latent)[0], :]
    z2 = Uy @ z1
    V = sparse_chol_lower(_rev_sparse(Uy @ Uy.T))
    z3 = _tri_solve(V, z2[::-1])
    quadform_denom = np.sum(z3 * z3)
    logdet_denom = -2 * np.sum(np.log(V.diagonal()))
    mu_ord = (-_tri_solve(V, z3, transpose=True))[::-1]
    const = ld(n) * np.log(2 * ld(np.pi))
    ll = -(logdet_num - logdet_denom + quadform_num - quadform_denom + const) / 2
    return dict(mu_ord=mu_ord.astype(np.float64), loglik=float(ll), logdet_num=float(logdet_num),
                quadform_num=float(quadform_num), logdet_denom=float(logdet_denom), quadform_denom=float(quadform_denom))


def vl_family(model, likparms=None):
    """R/vecchia_laplace_NR.R:213-276."""
    from scipy.special import gammaln
    lp = dict(alpha=2, sigma=np.sqrt(.1))
    lp.update(likparms or {})
    a, sg = lp["alpha"], lp["sigma"]
    if model == "poisson":
        return dict(hess=lambda y, z: np.exp(y), score=lambda y, z: z - np.exp(y),
                    llh=lambda y, z: np.sum(z * y - np.exp(y) - gammaln(z + 1)))
    if model == "logistic":
        return dict(hess=lambda y, z: np.exp(y) / (1 + np.exp(y)) ** 2, score=lambda y, z: z - np.exp(y) / (1 + np.exp(y)),
                    llh=lambda y, z: np.sum(z * y - np.log(1 + np.exp(y))))
    if model == "gamma":
        return dict(hess=lambda y, z: a * z * np.exp(-y), score=lambda y, z: a * (z * np.exp(-y) - 1),
                    llh=lambda y, z: np.sum(-a * z * np.exp(-y) + (a - 1) * np.log(z) - a * y + a * np.log(a) - gammaln(a)))
    if model == "gaussian":
        return dict(hess=lambda y, z: np.full(len(y), 1 / sg ** 2), score=lambda y, z: (z - y) / sg ** 2,
                    llh=lambda y, z: np.sum(-.5 * (z - y) ** 2 / sg ** 2) - len(y) * (np.log(sg) + np.log(2 * np.pi) / 2))
    raise ValueError(model)


def calculate_posterior_VL(z, va, likelihood_model, covparms, covmodel="matern", likparms=None, max_iter=50,
                           convg=1e-6, prior_mean=None, sparse=False, trace=None, snapshot_convg=None):
    """R/vecchia_laplace_NR.R:31-155, missing observations (NaN in z) included.  sparse=True: every step's
    vecchia_prediction on sparse matrices (vecchia_prediction_mean_sparse: what the reference does, feasible at n = 5e5);
    trace: a list that receives max|y_o - y_prev| of every step (:124); snapshot_convg: a looser threshold — the result
    carries under "snapshot" the posterior the SAME loop would have returned with convg = snapshot_convg (the iterations
    are a prefix of these: vecchia_laplace_likelihood's default 1e-5 against 1e-6 here, without running the loop twice)."""
    predict = vecchia_prediction_mean_sparse if sparse else vecchia_prediction_mean
    z = np.asarray(z, dtype=np.float64)
    fam = vl_family(likelihood_model, likparms)
    pm = np.zeros(len(z)) if prior_mean is None else np.asarray(prior_mean, float)
    obs = np.where(~np.isnan(z))[0]                                    # :45
    z_obs = z[obs]
    y_o = pm.copy()                                                    # :81-82
    if len(y_o) > 1:
        y_o = y_o[obs]                                                 # :84
    convgd, tot = False, 0
    snap = None
    for i in range(1, max_iter + 1):                                   # :88
        y_prev = y_o
        D = 1 / fam["hess"](y_o, z_obs)                                # :93,100
        u = fam["score"](y_o, z_obs)                                   # :101
        pseudo = np.full(len(z), np.nan)                               # :103
        pseudo[obs] = D * u + y_o - pm[obs]                            # :105
        nuggets = np.full(len(z), np.inf)                              # :107
        nuggets[obs] = D                                               # :108
        mu = predict(pseudo, va, covparms, nuggets, covmodel)          # :112-113
        y_o = mu[obs] + pm[obs]                                        # :115
        if trace is not None:
            trace.append(float(np.max(np.abs(y_o - y_prev))))
        if snapshot_convg is not None and snap is None and np.max(np.abs(y_o - y_prev)) < snapshot_convg:
            snap = dict(mean=mu + pm, cnvgd=True, iter=i, t=pseudo + pm, D=D, model_llh=fam["llh"], prior_mean=pm)
        if np.max(np.abs(y_o - y_prev)) < convg:                       # :124
            convgd, tot = True, i
            break
        tot += 1
    out = dict(mean=mu + pm, cnvgd=convgd, iter=tot, t=pseudo + pm, D=D, model_llh=fam["llh"], prior_mean=pm)
    if snapshot_convg is not None:
        out["snapshot"] = snap
    return out


def vecchia_laplace_likelihood(z, va, likelihood_model, covparms, likparms=None, covmodel="matern", max_iter=50,
                               convg=1e-5, prior_mean=None, sparse=False, post_out=None, post=None):
    """R/vecchia_laplace_NR.R:361-416.  sparse=True: the loop and the pseudo-marginal likelihood on sparse matrices;
    post_out: a dict that receives the posterior the loop ended with (mean, iter, cnvgd, D, t); post: the posterior of
    a loop already run with THIS convg (calculate_posterior_VL(..., snapshot_convg=convg)["snapshot"]) instead of :369."""
    z = np.asarray(z, dtype=np.float64)
    if post is None:
        post = calculate_posterior_VL(z, va, likelihood_model, covparms, covmodel, likparms, max_iter, convg, prior_mean,
                                      sparse=sparse)                   # :369-370
    if post_out is not None:
        post_out.update(post)
    if not post["cnvgd"]:
        return -np.inf                                                 # :373
    pm = post["prior_mean"]
    z_pseudo = post["t"] - pm                                          # :381
    D = post["D"]
    if len(D) < len(z_pseudo) and np.any(np.isnan(z_pseudo)):          # :382-387
        full = np.full(len(z_pseudo), np.nan)
        full[~np.isnan(z_pseudo)] = D
        D = full
    marg = (vecchia_likelihood_sparse if sparse else vecchia_likelihood)(z_pseudo, va, covparms, D, covmodel)   # :396-397
    io = ~np.isnan(z)                                                  # :401
    true_llh = post["model_llh"](post["mean"][io], z[io])              # :402
    with np.errstate(invalid="ignore"):
        cond = np.nansum(-0.5 * np.log(2 * np.pi * D) - 0.5 * (z_pseudo - (post["mean"] - pm)) ** 2 / D)   # :405, na.rm
    return marg - cond + true_llh                                      # :408-409


def calculate_posterior_VL_sparse(z, va, likelihood_model, covparms, **kw):
    """calculate_posterior_VL (R/vecchia_laplace_NR.R:31-155) with every Newton step's vecchia_prediction on sparse
    matrices: config 5 (n = 5e5, m = 30) in about a minute.  Pinned to the dense loop in tests/test_oracle.py."""
    return calculate_posterior_VL(z, va, likelihood_model, covparms, sparse=True, **kw)


def vecchia_laplace_likelihood_sparse(z, va, likelihood_model, covparms, **kw):
    """vecchia_laplace_likelihood (R/vecchia_laplace_NR.R:361-416) on sparse matrices."""
    return vecchia_laplace_likelihood(z, va, likelihood_model, covparms, sparse=True, **kw)
