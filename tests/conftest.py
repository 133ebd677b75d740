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


HOT_TOL_DIVISOR = 3.0


def tol_scale(g, ref):
    """Absolute tolerances are quoted for the O(1) tensors of the Xavier draws. The "hot" draw (matrices x 3, biases x 5,
    LayerNorm gamma in [-2, 2]) produces tensors one to two orders of magnitude larger; there the tolerance scales with the
    tensor: x max(1, max|reference tensor| / 3).

    Why 3: the line has to sit clear of the REFERENCE's own fp32 rounding. The imported reference's hot tensors are 0.28-0.49 of
    (1e-5 x max|ref| / 4) away from a float64 evaluation of the same network (test_oracle_golden.
    test_hot_tolerance_sits_above_the_reference_own_rounding), so with a divisor of 4 the line was 2 x that noise and two
    fp32 evaluations with different summation orders — the reference on CPU, any correct kernel — can meet it from opposite sides
    (seen in a round-3 experiment: 1 of 4 074 head outputs at 1.006 of the old line after a GELU variant that moved every other hot
    ratio DOWN). The Xavier draws (and BASELINE's 1e-5 / 1e-4) are untouched."""
    if "weight_style" not in g or str(g["weight_style"]) != "hot":
        return 1.0
    return max(1.0, float(np.nanmax(np.abs(ref))) / HOT_TOL_DIVISOR)
