import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import numpy as np
    path = os.path.join(GOLDEN_DIR, name)
    if not os.path.isfile(path):
        pytest.skip(f"fixture {name} not generated")
    return np.load(path, allow_pickle=False)


def golden_case(blob, prefix):
    """Sub-dict of an npz whose keys start with ``prefix/``."""
    plen = len(prefix) + 1
    return {k[plen:]: blob[k] for k in blob.files if k.startswith(prefix + "/")}
