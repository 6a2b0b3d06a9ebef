import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["toy_d8", "toy_d16_k32", "toy_d64"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    trip = g["triplets"]
    # reference dataset.py:116: add_edges(triplet[:,2], triplet[:,0]) -> src = t, dst = h
    g["src"], g["dst"], g["etype"] = trip[:, 2].copy(), trip[:, 0].copy(), trip[:, 1].copy()
    g["n"] = int(g["n_nodes"])
    g["R"] = int(g["n_rel"])
    g["W2"] = [g[k] for k in sorted(k for k in g if k.startswith("W2_"))]
    return g


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)


def rel_err(x, y):
    """SURVEY 8c tolerance form: max|x-y| / max(|y|, 1e-3*||y||inf), per tensor."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    scale = np.maximum(np.abs(y), 1e-3 * max(np.abs(y).max(), 1e-30))
    return float(np.max(np.abs(x - y) / scale)) if y.size else 0.0
