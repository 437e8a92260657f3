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
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" and z[k].ndim > 0 else z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def seeded_state():
    """state dicts of the seeded model (same seed as tests/golden/make_golden.py)."""
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_

    def get(refine):
        # a fresh CPU model per call: GPU tests move theirs with .to(device)
        m = CDSMVSNet(refine=refine, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75))
        seeded_init_(m, 7)
        return m.eval()
    return get
