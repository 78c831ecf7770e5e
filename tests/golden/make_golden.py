"""Regenerates tests/golden/*.npz.

The reference cannot be run here (no R), and its own tests hold no vectors for this path, so
these are REGRESSION fixtures produced by the oracle (oracle/), frozen so that later changes of either the
oracle or the HIP path are caught against fixed numbers.  The 6-point case reproduces the known-answer values
recorded in SURVEY.md §8c digit for digit.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import r_side as R   # noqa: E402


PROVENANCE = ("oracle-generated (oracle/r_side.py + oracle/u_nzentries_oracle.c through this script): a REGRESSION fixture, NOT output of the "
              "GPvecchia package; every field, the posterior quantities V_diag / logdet_denom / quadform_denom / mu_obs included, "
              "comes from the oracle itself.  The only parity pin would be tests/golden/reference_run/ (make_golden_reference.R, never run: "
              "no R in this image)")


def case(name, locs, z, m, ordering, cond, covmodel, covparms, nuggets):
    va = R.vecchia_specify(locs, m, ordering=ordering, cond_yz=cond)
    U = R.createU(va, covparms, nuggets, covmodel)
    ll = R.vecchia_likelihood_U(z, U)
    prep = va["U_prep"]
    # the posterior quantities (row f-1 of SURVEY.md section 8: U2V, the denominator terms, vecchia_mean) of the same case
    V = R.U2V(U)                                                                   # R/vecchia_prediction.R:62-111
    mu_obs = R.vecchia_mean(z, U, V)                                               # :118-142
    lat = U["latent"]
    zord = np.asarray(z)[U["ord_z"] - 1]
    z1 = U["U"][~lat, :].T @ zord
    z2 = U["U"][lat, :] @ z1
    z3 = np.linalg.solve(V, z2[::-1])                                              # R/vecchia_likelihood.R:88 (V lower triangular)
    post = dict(mu_obs=mu_obs, logdet_denom=-2 * np.sum(np.log(np.diag(V))), quadform_denom=np.sum(z3 ** 2),
                V_diag=np.diag(V).copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), provenance=PROVENANCE, locs=locs, z=z, m=m, ordering=ordering, cond=cond, **post,
                        covmodel=covmodel, covparms=np.asarray(covparms, float), nuggets=np.asarray(nuggets, float),
                        ord=va["ord"], revNNarray=np.nan_to_num(prep["revNNarray"]).astype(np.int32),
                        revCond=np.nan_to_num(prep["revCond"], nan=-1).astype(np.int8),
                        Lentries=U["U_entries"]["Lentries"], Zentries=U["U_entries"]["Zentries"], loglik=ll,
                        rowpointers=prep["rowpointers"], colindices=prep["colindices"])
    return ll


if __name__ == "__main__":
    kat = np.array([[0, 0], [1, 0], [0, 1], [1, 1], [.5, .5], [.25, .75]], float)
    zk = np.array([0.1, -0.2, 0.3, 0.4, -0.5, 0.6])
    print("kat_z", case("kat_z", kat, zk, 2, "none", "z", "matern", [1, .5, 1.5], .1))
    print("kat_sgv", case("kat_sgv", kat, zk, 2, "none", "SGV", "matern", [1, .5, 1.5], .1))
    rng = np.random.default_rng(2024)
    locs = rng.random((250, 2)); z = rng.standard_normal(250)
    tau = 0.05 + 0.2 * rng.random(250)
    print(case("rand2d_m10_sgv_nu15", locs, z, 10, "maxmin", "SGV", "matern", [1.3, 0.2, 1.5], tau))
    print(case("rand2d_m10_z_nu05", locs, z, 10, "none", "z", "matern", [0.9, 0.3, 0.5], 0.1))
    print(case("rand2d_m10_y_nu25", locs, z, 10, "coord", "y", "matern", [1.1, 0.1, 2.5], 0.2))
    print(case("rand2d_m10_sgv_esqe", locs, z, 10, "maxmin", "SGV", "esqe", [1.0, 0.3, 0.5, 0.2], 0.1))
    print(case("rand2d_m10_sgv_nu08", locs, z, 10, "maxmin", "SGV", "matern", [1.0, 0.2, 0.8], 0.1))
    l3 = rng.random((150, 3)); z3 = rng.standard_normal(150)
    print(case("rand3d_m30_sgv_nu05", l3, z3, 30, "maxmin", "SGV", "matern", [1.0, 0.3, 0.5], 0.1))
