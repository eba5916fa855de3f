import os
import sys

import numpy as np
import pytest

# torch first: its wheel bundles its own HIP runtime under the soname libtinyknn_hip.so asks
# for, so loaded in this order the process holds ONE runtime (as bench.py's does).  The other
# order maps /opt/rocm's runtime for the library and a second one for torch, whose
# initialisation then fails on the GPU box ("No HIP GPUs are available").
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "scripts") not in sys.path:      # bench / test scaffolding (simulated_peers.py)
    sys.path.append(os.path.join(ROOT, "scripts"))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Batches of up to TK_OPT_PAIR_NQ queries (product default 8192) replay their heaps one query per wave with the heap in
# registers.  Most tests here use batches of a few dozen to a few hundred queries and are ABOUT the lane kernels: start
# every index of the suite with a threshold of 4 (single queries and tiny batches take the register heap everywhere;
# tests/test_pair_replay_gpu.py and the heap_mode 3 legs cover it at every size, and at the product default).
os.environ.setdefault("TINYKNN_PAIR_NQ", "4")


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


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the HIP library (hipcc cross-compiles for gfx950 without a GPU) and the
    oracle if they are missing or older than their sources."""
    import subprocess
    csrc = os.path.join(ROOT, "tinyknn_amd", "csrc")
    so = os.path.join(ROOT, "tinyknn_amd", "libtinyknn_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "tinyknn_hip.h"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call(["make", "-s", "-j4", "-C", csrc])
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
