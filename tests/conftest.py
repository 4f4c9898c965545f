import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import packppi_amd  # noqa: E402,F401  (sets HIP_FORCE_DEV_KERNARG before anything initialises the HIP runtime)

GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 20251003


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (Batch, dict of remaining arrays as torch tensors)."""
    from packppi_amd.batch import Batch
    z = np.load(os.path.join(GOLD, name + ".npz"))
    batch, rest = Batch(), {}
    for k in z.files:
        v = z[k]
        if k.startswith("batch."):
            key = k[len("batch."):]
            batch[key] = int(v) if key in ("num_proteins", "max_size") else torch.from_numpy(v)
        elif v.dtype.kind in "fiub":
            rest[k] = torch.from_numpy(np.asarray(v))
        else:
            rest[k] = v
    return batch, rest


@pytest.fixture(scope="session")
def weights():
    from packppi_amd.weights import make_random_state_dict
    return make_random_state_dict(WEIGHT_SEED)


def wrapped_absdiff(a, b):
    d = (a.double() - b.double()).abs()
    return torch.minimum(d, (2 * np.pi - d).abs())
