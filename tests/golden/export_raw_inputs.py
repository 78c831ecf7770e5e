"""Writes tests/golden/raw/<case>/: the INPUTS of every committed fixture as raw little-endian arrays an R script can
readBin(), for tests/golden/make_golden_reference.R (which runs the REAL GPvecchia package on them; it cannot run here).

    python tests/golden/export_raw_inputs.py

Per case:  meta.txt (key=value), locs.f64 (n x d, column-major), z.f64, covparms.f64, nuggets.f64 (1 or n values, ORIGINAL
order), and -- so that the R side can pin the hot path on exactly our plan even if its own ordering / neighbour search
breaks a tie differently -- ord.i32 (1-based), NNarray.i32 and Cond.i32 (n x (m+1), column-major, NOT reversed: the
arguments of GPvecchia:::U_sparsity; NA = -2147483648 = R's NA_integer_)."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NA_INT = -2147483648


def main():
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        g = np.load(path, allow_pickle=False)
        name = os.path.basename(path)[:-4]
        out = os.path.join(HERE, "raw", name)
        os.makedirs(out, exist_ok=True)
        locs = np.asarray(g["locs"], dtype="<f8")
        n, d = locs.shape
        nug = np.atleast_1d(np.asarray(g["nuggets"], dtype="<f8"))
        np.asfortranarray(locs).ravel(order="F").tofile(os.path.join(out, "locs.f64"))
        np.asarray(g["z"], dtype="<f8").tofile(os.path.join(out, "z.f64"))
        np.asarray(g["covparms"], dtype="<f8").tofile(os.path.join(out, "covparms.f64"))
        nug.tofile(os.path.join(out, "nuggets.f64"))
        np.asarray(g["ord"], dtype="<i4").tofile(os.path.join(out, "ord.i32"))
        rev = np.asarray(g["revNNarray"])                       # (n, m+1), 0 = NA, reversed columns
        NN = rev[:, ::-1].astype("<i4")
        NN = np.where(NN == 0, NA_INT, NN).astype("<i4")
        cond = np.asarray(g["revCond"])[:, ::-1].astype("<i4")  # -1 = NA
        cond = np.where(cond < 0, NA_INT, cond).astype("<i4")
        np.asfortranarray(NN).ravel(order="F").tofile(os.path.join(out, "NNarray.i32"))
        np.asfortranarray(cond).ravel(order="F").tofile(os.path.join(out, "Cond.i32"))
        with open(os.path.join(out, "meta.txt"), "w") as f:
            f.write(f"n={n}\nd={d}\nm={int(g['m'])}\nordering={str(g['ordering'])}\ncond.yz={str(g['cond'])}\n"
                    f"covmodel={str(g['covmodel'])}\nncovparms={g['covparms'].size}\nnnuggets={nug.size}\n")
        print(name, n, d, int(g["m"]))


if __name__ == "__main__":
    main()
