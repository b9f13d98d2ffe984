"""Shared fixtures.  `-m "not gpu"` runs here on CPU; `-m gpu` runs on the MI355X box."""
import json
import os
import sys

import numpy as np
import pytest

# torch first: if a test later touches torch.cuda, its bundled libamdhip64 (same SONAME as
# /opt/rocm's) must be the one HIP runtime in the process, shared with libwhenet_hip.so.
import torch  # noqa: F401

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "headposeestimation-whenet_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    from whenet_hip import weights as W
    return W.synthetic(1234)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        info = json.load(f)
    exp = dict(np.load(os.path.join(GOLDEN, "golden_expected.npz")))
    crops = np.load(os.path.join(GOLDEN, "golden_crops.npy"))
    return {"info": info, "expected": exp, "crops": crops}


@pytest.fixture(scope="session")
def weights_file(weights, tmp_path_factory):
    from whenet_hip import weights as W
    p = tmp_path_factory.mktemp("snap") / "synthetic_seed1234.whnp"
    W.save(str(p), weights)
    return str(p)
