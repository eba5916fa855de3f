"""The wave-per-query heap replay with the heap in registers (heap.hip: heap_replay_pair_kernel) — the kernel behind
ONE query per call, which is what the reference's own bench times (examples/bench.py:118-137: `ivf.query(q)` in a
Python loop), and behind small batches (TK_OPT_PAIR_NQ, product default 8192 one batch at a time).

Heap arrays (layout included), probe order and final ids against the oracle (ivf.py:106-163 through
_fast_pq_256.pyx:73-123,188-210) for labels that are distinct and labels that repeat (IVF.build(n_probes=2): every
row in two lists, the duplicate test of `insert` on (value, label) entries), at every batch size up to the threshold,
heaps of 1 ... 513 entries (two, four and eight nodes per lane), both coarse (2 n_probes + 10 entries) and list replays; the CPU lemma for the
two-nodes-per-lane formulation is tests/test_pair_heap_lemma.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_index(oracle, ivf):
    L = len(ivf.active_centers)
    return oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                              ivf.pq_transformed_centers.packed,
                              [ivf.pq_transformed_points[i].packed for i in range(L)],
                              [ivf.pq_transformed_points[i].size for i in range(L)],
                              [ivf.ids[i] for i in range(L)], ivf.data)


@pytest.fixture(scope="module", params=[1, 2])
def built(request, oracle):
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(20 + request.param)
    n, d, nq = 40000, 100, 300
    cent = np.random.randn(200, d)
    X = (cent[np.random.randint(200, size=n)] + 0.7 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(200, size=nq)] + 0.7 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 200, FastPQ(2))
    ivf.fit(X[:15000]).build(X, n_probes=request.param)
    return ivf, oracle_index(oracle, ivf), qs


def check_batch(dev, ox, qn, qp, k, n_probes, lo, hi):
    out, dbg = dev.query_batch(qn[lo:hi], qp[lo:hi], k, n_probes, debug=True)
    for qi in range(lo, hi):
        ids, odbg = ox.query(qn[qi], k, n_probes=n_probes, debug=True)
        np.testing.assert_array_equal(dbg["probes"][qi - lo], odbg["probes"], err_msg=f"q{qi}")
        np.testing.assert_array_equal(dbg["heap_idx"][qi - lo], odbg["heap_idx"], err_msg=f"q{qi} k{k} p{n_probes}")
        np.testing.assert_array_equal(dbg["heap_val"][qi - lo], odbg["heap_val"], err_msg=f"q{qi}")
        got = out[qi - lo]
        np.testing.assert_array_equal(got[got != -1] if len(ids) < k else got, ids)


def test_small_batches_take_the_register_heap_and_match_the_oracle(built):
    from tinyknn_amd import _lib
    ivf, ox, qs = built
    dev = ivf.device_index()
    qn, qp = ivf._prepare(qs.copy())
    dev.set_heap_mode(0)
    dev.set_option(_lib.OPT_PAIR_NQ, 256)         # (the product default is 8192; the suite starts indexes at 4: conftest.py)
    try:
        # (k, n_probes): heaps of (n_probes + 1) k + 1 = 21, 61, 111, 121, 94, 129, 3 entries; coarse heaps 12 ... 70
        for k, n_probes in ((10, 1), (10, 5), (10, 10), (10, 11), (3, 30), (1, 127), (1, 1)):
            for lo, hi in ((0, 1), (1, 4), (4, 68), (0, 300)):      # (300 > 256: the lane kernel's batch beside them)
                check_batch(dev, ox, qn, qp, k, n_probes, lo, hi)
        # four nodes per lane (heaps of 130 ... 257 entries: n_probes 12 ... 24 at k = 10) and eight (... 513: n_probes <= 50)
        for k, n_probes in ((10, 12), (10, 20), (10, 24), (2, 127), (10, 25), (10, 50), (4, 127)):
            for lo, hi in ((0, 1), (1, 70)):
                check_batch(dev, ox, qn, qp, k, n_probes, lo, hi)
        # beyond 513 entries the register heap does not apply: the other kernels answer, same arrays
        check_batch(dev, ox, qn, qp, 10, 60, 0, 40)
    finally:
        dev.set_option(_lib.OPT_PAIR_NQ, 4)


def test_forced_for_every_batch_size(built):
    from tinyknn_amd import _lib
    ivf, ox, qs = built
    dev = ivf.device_index()
    qn, qp = ivf._prepare(qs.copy())
    dev.set_heap_mode(3)
    try:
        for scan_mode in (1, 2):
            dev.set_scan_mode(scan_mode)
            check_batch(dev, ox, qn, qp, 10, 10, 0, 300)
        # the duplicate test on (value, label64) entries — what an index of more than 16.7 M rows gets — instead of
        # value8 << 24 | label24: two, four and eight nodes per lane
        dev.set_option(_lib.OPT_LABELS24, 0)
        for n_probes in (10, 20, 40):
            check_batch(dev, ox, qn, qp, 10, n_probes, 0, 120)
    finally:
        dev.set_option(_lib.OPT_LABELS24, 1)
        dev.set_heap_mode(0)
        dev.set_scan_mode(0)


def test_one_query_per_call_is_the_reference_protocol(built):
    """examples/bench.py:118-137: `ivf.query(q, k, n_probes)` per query, raw vectors in, ids out."""
    ivf, ox, qs = built
    for qi in range(60):
        q = qs[qi].copy()
        got = ivf.query(q, 10, n_probes=10)
        qn = qs[qi] / np.linalg.norm(qs[qi])
        want = ox.query(np.ascontiguousarray(qn, dtype=np.float32), 10, n_probes=10)
        np.testing.assert_array_equal(got, want, err_msg=f"q{qi}")
