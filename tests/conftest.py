import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    """Import the hyphenated package (or a sub-module of it)."""
    name = "prior-diffuse_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def seeded(shape, seed):
    import torch

    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g, dtype=torch.float32)


@pytest.fixture(scope="session")
def weights():
    import torch  # noqa: F401

    synth = pkg("synth")
    cache = {}

    def get(arch, seed=1234):
        if (arch, seed) not in cache:
            cache[(arch, seed)] = synth.make_state_dict(arch, seed)
        return cache[(arch, seed)]

    return get
