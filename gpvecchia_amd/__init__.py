"""gpvecchia_amd — MI355X-native engine for GPvecchia's U_NZentries / Vecchia
log-likelihood path.  The arithmetic lives in libgpvecchia_hip.so (HIP, gfx950,
C ABI in include/gpvecchia.h); this package is the host-side mirror of the R API."""
from ._lib import (GPV_WANT_DENOM, GPV_WANT_LOGLIK_Z, GPV_WANT_MEAN, GPV_WANT_MEAN_B, GPV_WANT_NUMERATOR, GPV_WANT_U, GpvError,  # noqa: F401
                   device_count)
from .api import (Comm, EsqeFun, MaternFun, MultiPlan, Plan, ReplicaPlans, U2V, U_NZentries, U_NZentries_mat, createU, loglik_from_sums,  # noqa: F401
                  loglik_z_from_sums, numerator_from_sums, vecchia_likelihood, vecchia_likelihood_U, vecchia_specify)

from .laplace import (calculate_posterior_VL, vecchia_laplace_likelihood,  # noqa: F401,E402
                      vecchia_laplace_likelihood_from_posterior, vecchia_laplace_prediction, vecchia_prediction)

from .wrappers import vecchia_estimate, vecchia_pred  # noqa: F401,E402

__version__ = "0.1.0"
