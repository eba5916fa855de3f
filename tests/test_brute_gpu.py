"""Exact k nearest vectors on the f32 matrix cores (tk_index_knn_brute)
against numpy's knn_brute formula (utils.py:66-86): the same `part` values bit for bit, the k
smallest in ascending (part, row) order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _index(data):
    """a DeviceIndex that only needs its vectors: one list holding everything"""
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.fast_pq import TransformedData
    from tinyknn_amd._transform import transform_data
    n, d = data.shape
    dq = d + (-d) % 8
    ivf = IVF("euclidean", 1, FastPQ(2))
    ivf.pq.centers = np.zeros((16, dq), np.float32)
    ivf.pq.sqrt_n_blocks = float(np.sqrt(dq // 2))
    ivf.active_centers = np.zeros((1, d), np.float32)
    ivf.pq_transformed_centers = TransformedData(1, transform_data(np.zeros((16, dq // 2), np.uint8)))
    pad = (-n) % 16
    ivf.pq_transformed_points = [TransformedData(n, transform_data(np.zeros((n + pad, dq // 2), np.uint8)))]
    ivf.ids = [np.arange(n, dtype=np.int64)]
    ivf.data = data
    return ivf.device_index()


def _numpy_part(X, Y):
    xn = np.einsum("ij,ij->i", X, X)
    yn = np.einsum("ij,ij->i", Y, Y)
    out = np.empty((len(X), len(Y)), np.float32)
    for i in range(0, len(X), 100):            # 100-row chunks: the GEMM shape of the reference
        out[i:i + 100] = xn[i:i + 100, None] + yn[None] - 2 * X[i:i + 100] @ Y.T
    return out


@pytest.mark.parametrize("n,d,nq,k", [(20000, 100, 300, 10), (9000, 17, 131, 1), (5000, 128, 200, 100),
                                      (40000, 64, 1000, 10), (700, 20, 5, 10)])
def test_knn_brute_matches_numpy(n, d, nq, k):
    rng = np.random.RandomState(n + d)
    cent = rng.randn(40, d)
    Y = (cent[rng.randint(40, size=n)] + 0.5 * rng.randn(n, d)).astype(np.float32)
    Y[123] = Y[77]                                   # exact duplicates: ties, lower row first
    Y[n - 1] = Y[77]
    X = (cent[rng.randint(40, size=nq)] + 0.5 * rng.randn(nq, d)).astype(np.float32)
    X[3] = Y[77]                                     # a query ON the duplicated vector
    got = _index(Y).knn_brute(X, k)
    part = _numpy_part(X[:nq - nq % 100 or nq], Y)   # whole chunks: the FMA-chain shape
    for i in range(len(part)):
        order = np.lexsort((np.arange(n), part[i]))[:k]      # ascending (part, row)
        np.testing.assert_array_equal(got[i], order, err_msg=f"query {i}")


def test_knn_brute_recall_of_the_index_itself():
    """the use it is built for: Recall10@10 of IVF.query_batch against it"""
    from tinyknn_amd import IVF, FastPQ
    rng = np.random.RandomState(2)
    cent = rng.randn(100, 50)
    X = (cent[rng.randint(100, size=30000)] + 0.4 * rng.randn(30000, 50)).astype(np.float32)
    qs = (cent[rng.randint(100, size=500)] + 0.4 * rng.randn(500, 50)).astype(np.float32)
    ivf = IVF("euclidean", 170, FastPQ(2))
    ivf.fit(X[:10000]).build(X, n_probes=1)
    truth = ivf.device_index().knn_brute(qs, 10)
    got = ivf.query_batch(qs, 10, n_probes=170)      # every list probed, pass_1 = 1711 candidates
    recall = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(truth, got)])
    assert recall > 0.9, recall
