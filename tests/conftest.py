import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def split_lists(g):
    """Per-list packed codes / ids out of a g6_ivf_*.npz fixture."""
    sizes = g["list_sizes"]
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    codes = [g["list_codes"][coff[i]:coff[i + 1]] for i in range(len(sizes))]
    ids = [g["ids"][ioff[i]:ioff[i + 1]] for i in range(len(sizes))]
    return codes, ids


G6_TAGS = ["eu20", "an20", "an100", "an100b2", "eu128", "eu20f64"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
