"""Parity at BASELINE.json's full size (configs[1]: N = 1 183 514 x 100, 1087 lists, k = 10):
the index bench.py measures, every scan form / scan mode / heap mode, against the CPU oracle
on 2000+ rows — and configs[2]'s flat DistanceTable.top two-pass at 1M codes."""
import argparse
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full(oracle):
    import torch
    from tinyknn_amd import _lib
    if _lib.device_count() < 1:
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=1,
                              metric="angular", data="glove-like", fit_sample=100000,
                              cache_dir=os.environ.get("TMPDIR", "/tmp"))
    ivf, cent = bench.build_index(args, torch.device("cuda", 0))
    qs = bench.synth_queries(cent, 2048, 4242, kind="glove-like")
    return ivf, bench.oracle_index(ivf), qs


def test_full_size_index_all_modes_vs_oracle(full):
    from tinyknn_amd import _lib
    ivf, ox, qs = full
    assert ivf.data.shape == (1183514, 100) and len(ivf.active_centers) == 1087
    dev = ivf.device_index()
    qn, qp = ivf._prepare(qs.copy())
    L = _lib.lib()
    try:
        for n_probes in (1, 5, 10, 20, 50):
            want = ox.query_batch(qn, 10, n_probes)
            modes = ((0, 0, 0, 1), (1, 0, 0, 1), (2, 1, 0, 1), (2, 2, 0, 1), (2, 0, 1, 1), (2, 0, 2, 1), (0, 0, 0, 2))
            if n_probes in (5, 20):     # the sweep's other points: default path, one batch and pipelined + paired
                modes = ((0, 0, 0, 1), (0, 0, 0, 2))
            for scan_mode, heap_mode, form, depth in modes:
                dev.set_pipeline(depth)
                dev.set_coalesce(2 if depth > 1 else 1)
                dev.set_scan_mode(scan_mode)
                dev.set_heap_mode(heap_mode)
                dev.set_option(_lib.OPT_SCAN_FORM, form)
                got = dev.query_batch(qn, qp, 10, n_probes)
                bad = np.flatnonzero((got != want).any(axis=1))
                assert bad.size == 0, (n_probes, scan_mode, heap_mode, form, depth, bad[:5])
        # raw queries through the exact streamed front end
        np.testing.assert_array_equal(ivf.query_batch(qs, 10, n_probes=10), ox.query_batch(qn, 10, 10))
    finally:
        dev.set_pipeline(1); dev.set_coalesce(1); dev.set_scan_mode(0); dev.set_heap_mode(0)
        dev.set_option(_lib.OPT_SCAN_FORM, 0)


def test_flat_top_two_pass_at_1m_codes(full, oracle):
    """configs[2]: _FastDistanceTable.top over ONE flat TransformedData of the full data set
    (fast_pq.py:284-312: heap of rescore = 2k+10 candidates, exact rescoring) vs the oracle."""
    import time
    from tinyknn_amd.fast_pq import TransformedData
    ivf, ox, qs = full
    pq = ivf.pq
    from tinyknn_amd import _fast_pq
    n = 1 << 20
    rng = np.random.RandomState(3)
    rows = np.sort(rng.choice(len(ivf.data), n, replace=False))
    X = ivf.data[rows]
    td = pq.transform(X, device=True)
    assert isinstance(td, TransformedData) and td.size == n
    qn, _ = ivf._prepare(qs[:64].copy())
    _fast_pq.cache_device_codes = True        # written once, scanned 64 times
    try:
        pq.distance_table(qn[0]).top(td, X, k=10)
        t0 = time.perf_counter()
        got = [pq.distance_table(q).top(td, X, k=10) for q in qn]
        el = time.perf_counter() - t0
    finally:
        _fast_pq.cache_device_codes = False
        _fast_pq.forget_device_codes()
    for q, g in zip(qn, got):
        dt = pq.distance_table(q)
        idx = np.zeros(30, np.int64); val = np.zeros(30, np.int32)
        oracle.init_heap(idx, val, True)
        oracle.query_pq(td.packed, n, dt.tables, idx, val, True, None, oracle.ORDER_AVX)
        exp = idx[oracle.knn_brute1(q, X[idx], 10)]
        np.testing.assert_array_equal(g, exp)
    print(f"flat top() two-pass, 2^20 codes x M=52: {el / len(qn) * 1e3:.2f} ms per query (host API)")
