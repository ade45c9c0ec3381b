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


def load_golden(name):
    """-> (cfg dict, {prefix: {key: ndarray}}) from tests/golden/<name>.npz."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    groups = {}
    for k in z.files:
        pre, _, rest = k.partition(".")
        groups.setdefault(pre, {})[rest] = z[k]
    cfg = {}
    for k, v in groups.get("cfg", {}).items():
        v = v.item() if v.shape == () else v
        cfg[k] = v
    return cfg, groups


@pytest.fixture(scope="session")
def golden():
    return load_golden
