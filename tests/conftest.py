import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """A committed fixture as a dict of torch tensors (numpy scalars stay numpy)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype.kind in "fiub" and a.ndim > 0:
            out[k] = torch.from_numpy(a.copy())
        else:
            out[k] = a
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle_params():
    from oracle import supnerf_oracle as O
    return O.init_decoder_params(seed=0, sigma_bias=-2.0)
