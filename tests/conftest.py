import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def synthetic_weights():
    from thermompnn_amd.weights import synthetic_state_dict
    return synthetic_state_dict(0)


def weights_for_case(g):
    """The synthetic weight set a golden fixture was generated with (make_golden.py stores seed + style)."""
    from thermompnn_amd.weights import synthetic_state_dict
    return synthetic_state_dict(int(g["weight_seed"]), style=str(g["weight_style"]) if "weight_style" in g else "xavier")


def tol_scale(g, ref):
    """Absolute tolerances are quoted for the O(1) tensors of the Xavier draws. The "hot" draw (matrices x 3, biases x 5,
    LayerNorm gamma in [-2, 2]) produces tensors one to two orders of magnitude larger; there the tolerance scales with the
    tensor: x max(1, max|reference tensor| / 4) (4 = the largest hidden-state magnitude of the Xavier draws)."""
    if "weight_style" not in g or str(g["weight_style"]) != "hot":
        return 1.0
    return max(1.0, float(np.nanmax(np.abs(ref))) / 4.0)
