"""FlatTop (tk_index_top_centers): `_FastDistanceTable.top` (fast_pq.py:284-312) for a batch of
queries over ONE flat code array — row by row the reference's per-query call, which
test_hip_parity pins to the compiled reference, and the oracle directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_top(O, pq, td, X, q, k):
    rescore = min(2 * k + 10, td.size)
    idx = np.zeros(rescore, np.int64)
    val = np.zeros(rescore, np.int32)
    O.init_heap(idx, val, True)
    O.query_pq(td.packed, td.size, pq.distance_table(q).tables, idx, val, True, None, O.ORDER_AVX)
    if rescore <= k:
        return idx
    return idx[O.knn_brute1(q, X[idx], k)]


@pytest.mark.parametrize("n,d,k", [(5003, 100, 10), (20000, 128, 10), (37, 100, 10), (9, 100, 10), (3000, 100, 1)])
def test_flat_top_batch_equals_per_query_top(oracle, n, d, k):
    from tinyknn_amd import FastPQ
    from tinyknn_amd.fast_pq import FlatTop
    rng = np.random.RandomState(n)
    cent = rng.randn(12, d)
    X = (cent[rng.randint(12, size=n)] + 0.6 * rng.randn(n, d)).astype(np.float32)
    qs = (cent[rng.randint(12, size=70)] + 0.6 * rng.randn(70, d)).astype(np.float32)
    pq = FastPQ(2)
    pq.fit(X[:3000] if n >= 3000 else np.concatenate([X] * (3000 // n + 1)))
    assert (pq.R is not None) == (d != 100)
    td = pq.transform(X)
    ft = FlatTop(pq, td, X)
    got = ft.top(qs, k)
    kk = min(k, n)
    assert got.shape == (70, kk)
    for i, q in enumerate(qs):
        np.testing.assert_array_equal(got[i], pq.distance_table(q).top(td, X, k=k))
        if i % 7 == 0:
            np.testing.assert_array_equal(got[i], oracle_top(oracle, pq, td, X, q, kk))
    ft.close()


@pytest.mark.parametrize("d,structured", [(100, True), (128, True), (100, False)])
def test_flat_top_long_rows_on_the_matrix_cores(oracle, d, structured):
    """n >= 2^16 rows: the scan behind the exact head (first n/64 rows) runs as plain sums on the int8
    matrix cores, the replay fetches only the blocks whose minimum passes its bound — same rows as the
    reference's per-query call and the oracle.  Rows without structure fail the per-query check (bound
    at the first plain block above the table's limit): that chunk is answered again exactly, the index
    stays on the exact kernel, rows still identical."""
    from tinyknn_amd import FastPQ
    from tinyknn_amd.fast_pq import FlatTop
    n, nq, k = 70000 + d, 200, 10
    rng = np.random.RandomState(d + structured)
    if structured:
        cent = rng.randn(40, d)
        X = (cent[rng.randint(40, size=n)] + 0.5 * rng.randn(n, d)).astype(np.float32)
        qs = (cent[rng.randint(40, size=nq)] + 0.5 * rng.randn(nq, d)).astype(np.float32)
    else:
        X = rng.randn(n, d).astype(np.float32)
        qs = rng.randn(nq, d).astype(np.float32)
    pq = FastPQ(2)
    pq.fit(X[:4000])
    td = pq.transform(X, device=True)
    ft = FlatTop(pq, td, X)
    for rep in range(2):
        got = ft.top(qs, k)
        for i in range(0, nq, 9):
            np.testing.assert_array_equal(got[i], oracle_top(oracle, pq, td, X, qs[i], k))
        for i in (1, 50, 199):
            np.testing.assert_array_equal(got[i], pq.distance_table(qs[i]).top(td, X, k=k))
    ft.close()


@pytest.mark.parametrize("signd", [True, False])
@pytest.mark.parametrize("chunks,n_off,R", [(4096, 0, 30), (6000, -7, 30), (6000, -16 * 5000, 12), (5000, 0, 2000),
                                            (70000, -3, 30), (4500, 40, 1), (5000, -1, 64), (5000, -1, 65)])
def test_query_pq_long_rows_fresh_heap_compacts_candidates(oracle, signd, chunks, n_off, R):
    """Per-query `query_pq` over >= 4096 blocks with a fresh heap of at most 64 entries: one launch
    replays the first ~sqrt(R chunks) blocks with the heap in registers, compacts the later blocks whose
    minimum is below the bound reached there, and replays those (heap.hip, flat_top_one_kernel).  Same
    heap arrays as the oracle — with n short of (or beyond) the padded rows, signed and unsigned, heaps
    on either side of the 64-entry limit (R = 65, 2000: the general kernel), and a second call that
    continues from the heap of the first (general kernel)."""
    from tinyknn_amd import _fast_pq_avx as F
    from tinyknn_amd import _fast_pq as P
    rng = np.random.RandomState(chunks + R)
    M = 16
    data = rng.randint(0, 2 ** 63, size=(chunks, M), dtype=np.int64).astype(np.uint64)
    data |= rng.randint(0, 2, size=(chunks, M)).astype(np.uint64) << np.uint64(63)
    lo, hi = (-8, 8) if signd else (0, 16)
    t8 = rng.randint(lo, hi, size=(M * 16,)).astype(np.int8 if signd else np.uint8)
    t8[rng.randint(0, M * 16, size=40)] = 7 if signd else 15
    tables = t8.view(np.uint64).copy()
    n = 16 * chunks + n_off
    gi, gv = np.zeros(R, np.int64), np.zeros(R, np.int32)
    wi, wv = np.zeros(R, np.int64), np.zeros(R, np.int32)
    P.init_heap(gi, gv, signd)
    oracle.init_heap(wi, wv, signd)
    F.query_pq_avx(data, n, tables, gi, gv, signd)
    oracle.query_pq(data, n, tables, wi, wv, signd, None, oracle.ORDER_AVX)
    np.testing.assert_array_equal(gi, wi)
    np.testing.assert_array_equal(gv, wv)
    assert (gi < max(n, 0)).all()
    # continue from that heap (not fresh: the general path), other tables
    tables2 = np.roll(t8, 5).view(np.uint64).copy()
    F.query_pq_avx(data, n, tables2, gi, gv, signd)
    oracle.query_pq(data, n, tables2, wi, wv, signd, None, oracle.ORDER_AVX)
    np.testing.assert_array_equal(gi, wi)
    np.testing.assert_array_equal(gv, wv)
