"""Host-side, parameter-independent setup: the counterpart of R/vecchia_specify.R,
R/ordering_functions.R, R/NN_kdtree.R, R/whichCondOnLatent.R and R/U_sparsity.R.

These run once per data set (R/vecchia_specify.R:25-27) and are NOT the hot path
("next" rows); they are vectorised NumPy here so that plans for
1e5..1e6 locations can be built in seconds to minutes.  Index arrays follow the
R objects (1-based) with 0 standing in for R's NA in integer arrays and -1 for
NA in the logical revCond.
"""
from __future__ import annotations

import os

import numpy as np


# ---------------------------------------------------------------------------
# orderings — R/ordering_functions.R
# ---------------------------------------------------------------------------
def order_coordinate(locs, coordinate=None):
    """R/ordering_functions.R:126-128: order(rowSums(locs[,coordinate])), 1-based, stable."""
    locs = np.asarray(locs, dtype=np.float64)
    cols = list(range(locs.shape[1])) if coordinate is None else [c - 1 for c in np.atleast_1d(coordinate)]
    return np.argsort(locs[:, cols].sum(axis=1), kind="stable") + 1


def order_dist_to_point(locs, loc0):
    """R/ordering_functions.R:21-47 (lonlat=FALSE)."""
    locs = np.asarray(locs, dtype=np.float64)
    loc0 = np.asarray(loc0, dtype=np.float64).reshape(1, -1)
    if loc0.shape[1] != locs.shape[1]:
        raise ValueError("location in loc0 not in the same domain as the locations in locs")
    d = np.sqrt(((locs - loc0) ** 2).sum(axis=1))
    return np.argsort(d, kind="stable") + 1


def order_middleout(locs):
    """R/ordering_functions.R:64-81."""
    locs = np.asarray(locs, dtype=np.float64)
    return order_dist_to_point(locs, locs.mean(axis=0))


def order_outsidein(locs):
    """R/ordering_functions.R:98-102."""
    return order_middleout(locs)[::-1].copy()


def order_maxmin_exact(locs, native=True):
    """R/ordering_functions.R:147-150 -> src/MaxMin.cpp:661-738: exact max-min distance
    ordering, first point closest to the centroid.  native=True: the library's quasi-linear
    host routine (gpv_order_maxmin_exact, lazy heap + grid balls); native=False: the O(n^2)
    definition below with a running min-distance vector (its cross-check).  Returns 1-based indices."""
    locs = np.ascontiguousarray(locs, dtype=np.float64)
    n = locs.shape[0]
    if native:
        from . import _lib as L
        lf = np.asfortranarray(locs)
        out = np.empty(n, dtype=np.int32)
        L.check(L.lib().gpv_order_maxmin_exact(L.dptr(lf), n, locs.shape[1], L.iptr(out)), "gpv_order_maxmin_exact")
        return out.astype(np.int64)
    avg = np.zeros(locs.shape[1])
    for i in range(n):                      # sequential sums like src/MaxMin.cpp:679-691
        avg += locs[i]
    avg /= n
    first = int(np.argmin(((locs - avg) ** 2).sum(axis=1)))
    order = np.empty(n, dtype=np.int64)
    order[0] = first
    mind = np.sqrt(((locs - locs[first]) ** 2).sum(axis=1))
    mind[first] = -1.0
    for t in range(1, n):
        nxt = int(np.argmax(mind))
        order[t] = nxt
        np.minimum(mind, np.sqrt(((locs - locs[nxt]) ** 2).sum(axis=1)), out=mind)
        mind[nxt] = -1.0
    return order + 1


def get_knn(x, k, workers=-1):
    """FNN::get.knn(x, k)$nn.index (R/vecchia_specify.R:199, R/ordering_functions.R:189): the k nearest OTHER points of
    every point by ascending distance.  1-based (n, k)."""
    from scipy.spatial import cKDTree
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.shape[0]
    k = int(min(k, n - 1))
    _, ind = cKDTree(x).query(x, k=k + 1, workers=workers)
    ind = ind.reshape(n, -1)
    own = ind == np.arange(n)[:, None]
    # drop the point itself (first hit unless an exact duplicate sorts before it; then drop the last column instead)
    drop = np.where(own.any(axis=1), own.argmax(axis=1), k)
    keep = np.ones_like(ind, dtype=bool)
    keep[np.arange(n), drop] = False
    return ind[keep].reshape(n, k) + 1


def order_maxmin_exact_obs_pred(locs, locs_pred):
    """R/ordering_functions.R:174-218: maxmin ordering of the observed locations, then the prediction locations in their
    own maxmin order, each moved to the end while one of its nearest neighbours is still ahead of it.
    Returns (ord, ord_pred), 1-based."""
    locs = np.ascontiguousarray(locs, dtype=np.float64)
    locs_pred = np.ascontiguousarray(locs_pred, dtype=np.float64)
    ord_ = order_maxmin_exact(locs)
    ord_pred = order_maxmin_exact(locs_pred)
    n, n_pred = locs.shape[0], locs_pred.shape[0]
    m = int(min(round(np.sqrt(n)), 200))                                  # :185
    NN = get_knn(np.vstack([locs, locs_pred]), m)                         # :189
    iip = np.concatenate([ord_, n + ord_pred, np.zeros(2 * n_pred + 8, dtype=np.int64)]).astype(np.int64)   # 0 = NA
    poi = np.zeros(n + n_pred + 1, dtype=np.int64)
    poi[iip[: n + n_pred]] = np.arange(1, n + n_pred + 1)
    curlen, nmoved, N = n + n_pred, 0, n + n_pred
    for j in range(n + 1, n + 2 * n_pred + 1):                            # :199
        idx = iip[j - 1] if j <= iip.size else 0
        if idx == 0:
            continue
        nneigh = int(round(min(m, N / (j - nmoved + 1))))                 # :203
        nneigh = max(nneigh, 1)                                           # R: NN[i, 1:0] is column 1 (index 0 is dropped)
        if poi[NN[idx - 1, :nneigh]].min() < j:                           # :205
            nmoved += 1
            curlen += 1
            poi[idx] = curlen
            if curlen > iip.size:
                iip = np.concatenate([iip, np.zeros(iip.size, dtype=np.int64)])
            iip[curlen - 1] = idx
            iip[j - 1] = 0
    kept = iip[iip != 0]
    return ord_, kept[n: n + n_pred] - n                                  # :214


# ---------------------------------------------------------------------------
# ordered nearest neighbours — R/NN_kdtree.R:73-83 semantics
# ---------------------------------------------------------------------------
def _canon_dist(locs, q, cand):
    """sqrt(sum_t (a_t-b_t)^2) accumulated left to right (src/dist.cpp:10-16 / fields::rdist)."""
    ssq = np.zeros(cand.shape, dtype=np.float64)
    for t in range(locs.shape[1]):
        df = locs[cand, t] - locs[q, t][:, None]
        ssq += df * df
    return np.sqrt(ssq)


def find_ordered_nn(locs, m, workers=-1, rows=None):
    """Exact ordered nearest neighbours: row j (1-based) lists the min(m+1, j) points of
    locs[1..j] closest to locs[j] (self included) by ascending distance, lower index
    first among equal distances (R's stable order(), R/NN_kdtree.R:79-80).  Same
    definition as GpGp::find_ordered_nn used at R/vecchia_specify.R:159, minus its random
    jitter.  Returns int32 (n, m+1), 1-based, 0 = NA.  rows=(a, b) fills only rows a..b-1
    (a row shard of a multi-GPU plan); the other rows stay 0."""
    from scipy.spatial import cKDTree
    locs = np.ascontiguousarray(locs, dtype=np.float64)
    n = locs.shape[0]
    NN = np.zeros((n, m + 1), dtype=np.int32)
    start = min(n, max(2 * (m + 1), 32))
    ra, rb = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    # brute force on the first rows
    for j in range(max(0, ra), min(start, rb)):
        d = _canon_dist(locs, np.array([j]), np.arange(j + 1)[None, :])[0]
        o = np.lexsort((np.arange(j + 1), d))[: min(m + 1, j + 1)]
        NN[j, : len(o)] = o + 1
    lo = start
    while lo < n:
        hi = min(n, 2 * lo)
        if hi <= ra or lo >= rb:
            lo = hi
            continue
        tree = cKDTree(locs[:hi])
        pending = np.arange(max(lo, ra), min(hi, rb))
        kk = min(hi, 2 * (m + 1) + 2)
        while pending.size:
            dq, ind = tree.query(locs[pending], k=kk, workers=workers)
            ind = ind.reshape(len(pending), -1)
            dq = dq.reshape(len(pending), -1)
            ok = ind <= pending[:, None]
            ok &= ind < hi
            cnt = ok.sum(axis=1)
            # exact under ties: the (m+1)-th eligible distance must lie strictly inside the queried ball,
            # otherwise an equidistant point with a lower index may have been cut off by the k-NN query
            dm = np.sort(np.where(ok, dq, np.inf), axis=1)[:, min(m, kk - 1)]
            done = ((cnt >= m + 1) & (dq[:, -1] > dm * (1 + 1e-9))) | (kk >= hi)
            if done.any():
                sel = np.where(done)[0]
                q = pending[sel]
                cand = np.where(ok[sel], ind[sel], q[:, None])         # masked-out slots -> self (deduplicated by key)
                d = _canon_dist(locs, q, cand)
                d = np.where(ok[sel], d, np.inf)
                idxkey = np.where(ok[sel], cand, np.iinfo(np.int64).max)
                o = np.lexsort((idxkey, d), axis=1)[:, : m + 1]
                NN[q] = np.take_along_axis(cand, o, axis=1).astype(np.int32) + 1
                if kk >= hi:                                             # whole prefix queried: drop the padding
                    pad = np.take_along_axis(d, o, axis=1) == np.inf
                    NN[q] = np.where(pad, 0, NN[q])
            pending = pending[~done]
            kk = min(hi, kk * 2)
        lo = hi
    return NN


def find_ordered_nn_gpu(locs, m, rows=None, device=0):
    """Same result as find_ordered_nn (bit-exact), computed by the library's brute-force GPU kernel
    (gpv_find_ordered_nn).  Returns int32 (n, m+1), 1-based, 0 = NA; rows outside `rows` stay 0."""
    from . import _lib as L
    locs = np.asfortranarray(locs, dtype=np.float64)
    n, d = locs.shape
    a, b = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    out = np.zeros((n, m + 1), dtype=np.int32, order="F")
    L.check(L.lib().gpv_find_ordered_nn(int(device), L.dptr(locs), n, d, int(m), a, b, L.iptr(out)), "gpv_find_ordered_nn")
    return np.ascontiguousarray(out)


def whichCondOnLatent(NNarray, firstind_pred=None, native=True):
    """R/whichCondOnLatent.R:2-26 (SGV rule).  NNarray int (n, m+1), 1-based, 0 = NA.
    Returns int8 (n, m+1): 1 TRUE (latent), 0 FALSE (observed), -1 NA.
    native=True runs the C++ host routine of the library (gpv_whichCondOnLatent, O(n m^2));
    native=False the pure-Python restatement below (kept as its cross-check).

    R details reproduced: is.element(NA, x) is TRUE when x contains NA; the 'table' is
    NNarray[l,] * CondOnLatent[l,] (index, 0, or NA); which(...)[1] takes the first maximum."""
    NN = np.asarray(NNarray, dtype=np.int64)
    n, p = NN.shape
    if firstind_pred is None:
        firstind_pred = n + 1
    if native:
        from . import _lib as L
        nn32 = np.asfortranarray(NN.astype(np.int32))
        out = np.empty((n, p), dtype=np.int32, order="F")
        L.check(L.lib().gpv_whichCondOnLatent(L.iptr(nn32), n, p, int(firstind_pred), L.iptr(out)),
                "gpv_whichCondOnLatent")
        return np.where(out == L.NA_INTEGER, -1, out).astype(np.int8)
    Cond = np.full((n, p), -1, dtype=np.int8)
    Cond[0, 0] = 1
    na = NN == 0
    # table[l] = set of indices conditioned-on-latent in row l; table_na[l] = table contains NA
    lat_sets = [None] * n
    lat_sets[0] = {int(NN[0, 0])}
    tab_na = np.zeros(n, dtype=bool)
    tab_na[0] = bool(na[0].any())
    for k in range(1, n):
        row = NN[k]
        row_na = na[k]
        n_na = int(row_na.sum())
        vals = row[~row_na].tolist()
        latents = np.zeros(p, dtype=np.int64)
        for ind in range(1, p):
            l = row[ind]
            if l != 0 and l < firstind_pred:
                s = lat_sets[l - 1]
                cnt = sum(1 for v in vals if v in s)
                if tab_na[l - 1]:
                    cnt += n_na
                latents[ind] = cnt
        best = int(np.argmax(latents))                 # first maximum (which(...)[1])
        ind = int(row[best]) - 1
        if ind == k:                                    # all-zero case: table is the (still all-NA) row k itself
            s, sna = set(), True
        else:
            s, sna = lat_sets[ind], tab_na[ind]
        c = np.array([(v in s) for v in row], dtype=np.int8)
        c[row_na] = 1 if sna else 0
        c[row >= firstind_pred] = 1
        c[0] = 1
        c[row_na] = -1
        Cond[k] = c
        lat_sets[k] = {int(v) for v, cc in zip(row, c) if cc == 1}
        tab_na[k] = bool(row_na.any())
    return Cond


# ---------------------------------------------------------------------------
# U_sparsity — R/U_sparsity.R:5-81, vectorised
# ---------------------------------------------------------------------------
def U_sparsity(locs, NNarray, obs, Cond):
    """Symbolic structure of U.  NNarray int (0 = NA), Cond int8 (-1 = NA).
    Returns the U.prep list of R/U_sparsity.R:78-79 as a dict."""
    NN = np.asarray(NNarray)
    Cond = np.asarray(Cond)
    nnp = np.asarray(locs).shape[0]
    obs = np.asarray(obs, dtype=bool)
    n = int(obs.sum())
    size = nnp + n
    # :19-29 latent_map / observed_map (1-based rows of U)
    latent_map = np.arange(1, nnp + 1, dtype=np.int64) + np.concatenate([[0], np.cumsum(obs[:-1])])
    observed_map = np.where(obs, latent_map + 1, 0)
    revNN = NN[:, ::-1].copy()                                     # :32
    revCond = Cond[:, ::-1].copy()                                 # :33
    ok = revNN != 0
    k_idx, _ = np.nonzero(ok)                                      # row-major walk == the R loop order (:39-56)
    nb = revNN[ok].astype(np.int64) - 1
    cl = revCond[ok] == 1
    rowpointers = latent_map[k_idx]
    colindices = np.where(cl, latent_map[nb], observed_map[nb])
    obs_k = np.nonzero(obs)[0]                                     # :59-69
    Zrow = np.repeat(observed_map[obs_k], 2)
    Zcol = np.stack([latent_map[obs_k], observed_map[obs_k]], axis=1).reshape(-1)
    return dict(revNNarray=revNN, revCond=revCond, n_cores=os.cpu_count() or 1, size=size,
                rowpointers=np.concatenate([rowpointers, Zrow]).astype(np.int64),
                colindices=np.concatenate([colindices, Zcol]).astype(np.int64),
                y_ind=latent_map, observed_map=observed_map)
