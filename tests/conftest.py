import os
import sys

import pytest

# torch first: its wheel bundles a HIP runtime with the same soname (libamdhip64.so.7) as /opt/rocm's, which
# libgpvecchia_hip.so links; whichever is loaded first serves both, the other order puts two runtimes in one process
# and a torch stream handle handed to the library would belong to the wrong one
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    """RNG-free 6-point known-answer case (values re-derived by
    tests/golden/make_golden.py from the oracle and cross-checked there)."""
    import numpy as np
    return dict(
        locs=np.array([[0, 0], [1, 0], [0, 1], [1, 1], [.5, .5], [.25, .75]], float),
        z=np.array([0.1, -0.2, 0.3, 0.4, -0.5, 0.6]),
        covparms=[1.0, 0.5, 1.5], nugget=0.1, m=2,
        Lentries_z=np.array([[1, 0, 0],
                             [-0.128171102949723, 1.008994853700693, 0],
                             [-0.024455210030704, -0.125106066807264, 1.009321335625181],
                             [-0.124285439764088, -0.124285439764088, 1.017517359855895],
                             [-0.25951355535058, -0.25951355535058, 1.080270837405132],
                             [-0.750219079497773, -0.750219079497773, 1.604202979586874]]),
        loglik_z=-6.037912476524804,
        last_row_sgv=np.array([-0.751966819744953, -0.859057258396604, 1.656729586247101]),
        loglik_sgv=-6.024353219666226,
    )
