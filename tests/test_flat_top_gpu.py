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
