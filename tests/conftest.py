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


_ORACLE_RUNS = {}


def oracle_sample_pair(R, weights, feat, x_T, fast, prior="GCRN"):
    """(fp32 CPU oracle result, the same algorithm evaluated in float64) of ``R.sample`` on one input, memoised per
    test session: the 50-step evaluations take ~70 s each on the GPU box's host cores and several tests assert against
    the same input (seed 77, T = 401)."""
    import hashlib

    import torch

    key = (prior, bool(fast), tuple(feat.shape), hashlib.sha1(feat.numpy().tobytes()).hexdigest(),
           hashlib.sha1(x_T.numpy().tobytes()).hexdigest())
    if key not in _ORACLE_RUNS:
        params = pkg("params").params
        w32 = (weights(prior), weights("DiffUNet1"))
        w64 = tuple({k: v.double() for k, v in sd.items()} for sd in w32)
        with torch.no_grad():
            ref, _ = R.sample(prior, w32[0], w32[1], feat, x_T, params.noise_schedule, params.inference_noise_schedule, fast, False)
            exact, _ = R.sample(prior, w64[0], w64[1], feat.double(), x_T.double(), params.noise_schedule,
                                params.inference_noise_schedule, fast, False)
        _ORACLE_RUNS[key] = (ref, exact)
    return _ORACLE_RUNS[key]
