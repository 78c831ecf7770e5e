"""Row-by-row comparison of U entries with the oracle, shared by the GPU parity tests.

The bar (BASELINE.json north_star): every row of Lentries within 1e-8 of the oracle's, normwise
(max|dM| / max|M|).  A row that misses it is NOT waved through on its condition number: both
implementations are then measured against the same definition evaluated in extended precision
(oracle.r_side.rows_extended: x87 long double for the closed forms, 40-digit mpmath for the Bessel
branch) and the HIP row must be as good as the oracle's own double-precision row,

    err_hip <= max(4 * err_oracle, 1e-8)        (errors against the extended-precision row)

i.e. on an ill-conditioned block (two correct fp64 factorisations differ by ~cond(S) * eps there,
SURVEY.md §8d) the Gauss-Jordan sweep of the kernel may lose what dpotrf + dtrtrs lose, not more.
The number of rows that needed adjudication is returned; every caller asserts a cap on it.
"""
import json
import os

import numpy as np

FLAT = 1e-8


def row_errors(out, ref):
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-300)
    return np.abs(out - ref).max(axis=1) / scale


def check_rows(out, ref, locs, revNN, revCond, nuggets, covType, cp, rows=None, covVals=None, flat=FLAT, label=None):
    """out, ref: (r, p) rows of Lentries (HIP path, oracle) for the row numbers `rows` (default: all rows of the plan).
    revNN / revCond in either coding (NaN or 0 / -1 for missing).  Returns dict(rows, max_err, escaped, worst_ratio,
    max_err_hip_exact).  Raises AssertionError when an adjudicated row of the HIP path is worse than 4x the oracle's."""
    from oracle import r_side as R
    out = np.asarray(out)
    ref = np.asarray(ref)
    rows = np.arange(out.shape[0]) if rows is None else np.asarray(rows)
    err = row_errors(out, ref)
    bad = np.where(~(err <= flat))[0]                                   # NaN counts as bad
    res = dict(rows=int(out.shape[0]), max_err=float(np.nanmax(err)) if err.size else 0.0, escaped=int(bad.size),
               worst_ratio=0.0, max_err_hip_exact=0.0)
    if bad.size:
        nn = np.nan_to_num(np.asarray(revNN, dtype=np.float64), nan=0.0)
        cd = np.asarray(revCond, dtype=np.float64)
        cd = np.where(np.isnan(cd) | (cd < 0), 0.0, cd)
        ex = R.rows_extended(rows[bad], locs, nn, cd, nuggets, covType, cp, covVals=covVals)
        sc = np.maximum(np.abs(ex).max(axis=1), 1e-300)
        e_hip = np.abs(out[bad] - ex).max(axis=1) / sc
        e_ref = np.abs(ref[bad] - ex).max(axis=1) / sc
        ratio = e_hip / np.maximum(e_ref, flat / 4)
        res.update(worst_ratio=float(ratio.max()), max_err_hip_exact=float(e_hip.max()),
                   max_err_oracle_exact=float(e_ref.max()))
        worst = int(np.argmax(ratio))
        assert np.all(e_hip <= np.maximum(4 * e_ref, flat)), \
            (label, "row", int(rows[bad][worst]), "err_hip", float(e_hip[worst]), "err_oracle", float(e_ref[worst]))
    log = os.environ.get("GPV_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps(dict(label=label or os.environ.get("PYTEST_CURRENT_TEST", ""), **res)) + "\n")
    return res
