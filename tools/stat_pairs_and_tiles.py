#!/usr/bin/env python3
"""Host statistics that decide two kernel designs BEFORE anything is built (no GPU needed).

A. Pair sharing in the set kernel.  A wavefront task evaluates the covariance of the P(P-1)/2 point pairs of each of its 4
   conditioning sets (src/Matern.cpp:46-52 spends P^2 per set).  How many of those 4 x 465 pairs are DISTINCT pairs of
   points?  Measured for the plan's grouping (sets in Morton order of their own point, 4 consecutive sets per task) and for
   the grouping by (octave of the ordering index, Morton) that puts sets of similar neighbour radius together.
B. Tile residency of the posterior pass.  Morton-range tiles of the columns; a column of the leaf level or of levels 1..K
   is INTERIOR to its tile when every column it gathers from (the columns c > k that hold row k) lies in the same tile and
   is interior itself.  Which fraction of all columns (and of all (B, R) pairs) could a tile-resident kernel retire?

Uniform 2-D points, m = 30, the two orderings of the benchmark (none = the generated order; maxmin).  The statistics are
scale-free in n for uniform points, so n = 2e5 stands for 1e6 (pass --n to change)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def morton2(locs, bits=16):
    q = np.minimum((locs * (1 << bits)).astype(np.uint64), (1 << bits) - 1)

    def spread(v):
        v = v & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


def pair_sharing(NN, order_sets, spw=4, sample=20000, seed=0):
    """distinct / total point pairs over tasks of `spw` consecutive sets of `order_sets` (full rows only)."""
    n, p = NN.shape
    rng = np.random.default_rng(seed)
    ntask = len(order_sets) // spw
    pick = rng.choice(ntask, min(sample, ntask), replace=False)
    iu = np.triu_indices(p, 1)
    ratios, pts = [], []
    for t in pick:
        rows = order_sets[t * spw:(t + 1) * spw]
        S = NN[rows]
        if (S == 0).any():
            continue
        a, b = S[:, iu[0]], S[:, iu[1]]
        key = (np.minimum(a, b).astype(np.int64) << 32) | np.maximum(a, b).astype(np.int64)
        ratios.append(np.unique(key).size / key.size)
        pts.append(np.unique(S).size / S.size)
    return float(np.mean(ratios)), float(np.median(ratios)), float(np.mean(pts))


def stat_A(locs, NN, ordname):
    n = NN.shape[0]
    mort = morton2(locs)
    byM = np.argsort(mort, kind="stable")
    res = {}
    res["morton (the plan's grouping)"] = pair_sharing(NN, byM)
    octave = np.floor(np.log2(np.arange(n) + 1.0)).astype(np.int64)
    for sub in (1, 2, 4):
        # buckets of the ordering index: `sub` per octave (neighbour radius ~ index^-1/2: equal ratio = equal radius ratio)
        b = np.floor(sub * np.log2(np.arange(n) + 1.0)).astype(np.int64)
        o = np.lexsort((mort, b))
        res[f"(index bucket, {sub}/octave; morton)"] = pair_sharing(NN, o)
    print(f"  A. ordering={ordname}: distinct/total pairs per task of 4 sets  [mean, median | distinct/total POINTS]")
    for k, v in res.items():
        print(f"     {k:42s} {v[0]:.3f} {v[1]:.3f} | {v[2]:.3f}")
    return res


def stat_B(locs_ord, NN, cond, tiles_cols=(256, 512, 1024, 2048, 4096), Ks=(0, 2, 4, 6, 8, 12)):
    """NN, cond: (n, p) in ordered indexing (1-based neighbours, self first in NN[:,0]); cond True = latent."""
    n, p = NN.shape
    # column k of B holds the latent entries of set k (rows = neighbours conditioned on as latent, and k itself)
    rows_k = []
    src = [[] for _ in range(n)]            # readers: columns c > k that hold row k (k gathers from them)
    nnz = 0
    ks, rs = [], []
    for j in range(1, p):
        ok = (NN[:, j] > 0) & cond[:, j]
        kk = np.nonzero(ok)[0]
        ks.append(kk)
        rs.append(NN[kk, j] - 1)
    ks = np.concatenate(ks)
    rs = np.concatenate(rs)
    nnz = ks.size + n
    # level(k) = 1 + max level of the columns c > k that contain row k (gpv_api.hip build_posterior_impl)
    o = np.argsort(rs, kind="stable")
    rs_s, ks_s = rs[o], ks[o]
    ptr = np.searchsorted(rs_s, np.arange(n + 1))
    lev = np.zeros(n, dtype=np.int32)
    for k in range(n - 1, -1, -1):
        c = ks_s[ptr[k]:ptr[k + 1]]
        if c.size:
            lev[k] = lev[c].max() + 1
    mort = morton2(locs_ord)
    rank = np.empty(n, dtype=np.int64)
    rank[np.argsort(mort, kind="stable")] = np.arange(n)
    cols_per = np.bincount(ks, minlength=n) + 1
    print(f"  B. levels: {lev.max() + 1}; columns per level (first 12): {np.bincount(lev)[:12].tolist()}; nnz(B) = {nnz} "
          f"({nnz / n:.1f} per column)")
    cum = np.cumsum(np.bincount(lev)) / n
    print(f"     cumulative share of columns by level: " + " ".join(f"L{l}:{cum[l]:.2f}" for l in (0, 1, 2, 4, 6, 8, 12) if l < cum.size))
    out = {}
    for T in tiles_cols:
        tile = rank // T
        for K in Ks:
            interior = np.zeros(n, dtype=bool)
            # process by ascending level: interior iff level <= K and all sources are in the same tile and interior
            for l in range(0, K + 1):
                idx = np.nonzero(lev == l)[0]
                if l == 0:
                    interior[idx] = True
                    continue
                for k in idx:
                    c = ks_s[ptr[k]:ptr[k + 1]]
                    interior[k] = bool(np.all(interior[c] & (tile[c] == tile[k])))
            # a leaf column is only worth keeping on chip if someone in the tile reads it; count as retired anyway
            frac_cols = interior.mean()
            frac_pairs = cols_per[interior].sum() / nnz
            # of the gathers (k reads the block of c) how many become on-chip
            gk_int = interior[rs_s.repeat(1)] if False else None
            on_chip = (interior[rs] & interior[ks] & (tile[rs] == tile[ks])).sum() / ks.size
            out[(T, K)] = (frac_cols, frac_pairs, on_chip)
        print(f"     tile = {T:5d} columns (~{T * nnz / n / 1000:.1f} k pairs): " +
              "  ".join(f"K={K}: cols {out[(T, K)][0]:.2f} gathers {out[(T, K)][2]:.2f}" for K in Ks))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--m", type=int, default=30)
    ap.add_argument("--skip-b", action="store_true")
    a = ap.parse_args()
    from gpvecchia_amd import specify as S
    rng = np.random.default_rng(0)
    locs = rng.random((a.n, 2))
    print(f"n = {a.n}, m = {a.m}, uniform 2-D")
    for ordname in ("none", "maxmin"):
        t0 = time.time()
        if ordname == "maxmin":
            ord_ = S.order_maxmin_exact(locs)
            lo = locs[ord_ - 1] if ord_.min() == 1 else locs[ord_]
        else:
            lo = locs
        NN = S.find_ordered_nn(lo, a.m)
        print(f" ordering={ordname}: specify {time.time() - t0:.1f} s")
        stat_A(lo, NN, ordname)
        if ordname == "maxmin" and not a.skip_b:
            Cond = S.whichCondOnLatent(NN)
            stat_B(lo, NN, np.asarray(Cond, dtype=bool))


if __name__ == "__main__":
    main()
