import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

PARAM_KEYS = ("X", "Z", "U", "logvariance", "loglengthscales", "log_Q", "CC", "DD", "log_Rchols")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def actuator():
    z = np.load(os.path.join(GOLDEN, "actuator_slim.npz"), allow_pickle=False)
    params = {k: z[k] for k in PARAM_KEYS}
    return params, z["Y"], z["control_inputs"]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"golden_{name}.npz"), allow_pickle=False))
