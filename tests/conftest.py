import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN_DIR, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    # our own fixture file; numpy.load default allow_pickle=False
    return np.load(os.path.join(GOLDEN_DIR, "arrow_golden.npz"))


def split_batches(flat: np.ndarray, sizes) -> list:
    out, o = [], 0
    for s in sizes:
        s = int(s)
        out.append(np.ascontiguousarray(flat[o : o + s]))
        o += s
    assert o == flat.size
    return out


def sort_rows(*cols):
    """Canonical row order for join outputs: lexicographic by the given columns
    (first column most significant), as host/join/join_test.cc:27-38 does."""
    order = np.lexsort(tuple(reversed(cols)))
    return tuple(c[order] for c in cols)
