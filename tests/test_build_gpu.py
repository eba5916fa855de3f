"""Offline build path on the MI355X: tk_encode_pq / tk_assign_lists against
the oracle, the golden fixtures (codes and list memberships produced by the compiled
reference) and the product's own host (numpy) build."""
import numpy as np
import pytest

from conftest import G6_TAGS, golden, split_lists

pytestmark = pytest.mark.gpu


def _same_up_to_exact_ties(oracle, pq, X, a, b):
    """Packed codes a and b may differ only where numpy's own `part` values of the two
    labels are EXACTLY equal (random data does produce a few such ties, mostly for
    dims_per_block = 1, and numpy's AVX-512 argselect then need not return the first)."""
    la, lb = oracle.unpack(a), oracle.unpack(b)
    bad = np.argwhere(la != lb)
    assert len(bad) <= 1e-5 * la.size + 2
    dpb = pq.dims_per_block
    Xp = np.concatenate([X, np.zeros((len(la) - len(X), X.shape[1]), X.dtype)])
    pad = (-X.shape[1]) % (4 * dpb)
    Xp = np.concatenate([Xp, np.zeros((len(Xp), pad), X.dtype)], axis=1)
    if pq.R is not None:
        Xp = Xp @ pq.R.T
    for i, m in bad:
        lo = i - i % 100                                  # the chunk numpy scored the row in
        xc = Xp[lo:lo + 100, m * dpb:(m + 1) * dpb]
        code = pq.centers[:, m * dpb:(m + 1) * dpb]
        part = (np.einsum("ij,ij->i", xc, xc)[:, None] + np.einsum("ij,ij->i", code, code)[None]
                - 2 * xc @ code.T)[i - lo]
        assert part[la[i, m]] == part[lb[i, m]] == part.min()


def _pq(centers, dpb, R=None):
    from tinyknn_amd import FastPQ
    pq = FastPQ(dpb)
    pq.centers = centers
    pq.R = R
    return pq


@pytest.mark.parametrize("dpb,d,rot,f64", [(2, 100, False, False), (1, 100, False, False),
                                           (4, 100, False, False), (2, 128, True, False),
                                           (2, 20, False, True), (8, 64, False, False)])
def test_encode_vs_oracle(oracle, dpb, d, rot, f64):
    rng = np.random.RandomState(3)
    n = 5000
    pad = (-d) % (4 * dpb)
    R = np.linalg.qr(rng.randn(d + pad, d + pad))[0][:64] if rot else None
    dq = 64 if rot else d + pad
    for ties in (False, True):
        X = rng.randn(n, d).astype(np.float64 if f64 else np.float32)
        X[17] = 0                                         # a zero row (what pads a list)
        centers = (rng.randn(16, dq) * 0.8).astype(np.float32)
        if ties:
            # exact ties between centroids: integer rows, half-integer centroids, a duplicate.
            # First occurrence wins (numpy's generic argpartition path; its AVX-512 network
            # may pick another tied entry, so numpy itself is compared on tie-free data only)
            X[100:400] = np.round(X[100:400])
            centers = np.round(centers * 2) / 2
            centers[5] = centers[3]
        pq = _pq(centers, dpb, R)
        got = pq.transform(X, device=True)
        want_n, want = oracle.fastpq_transform(centers, dpb, R, X)
        assert got.size == want_n
        np.testing.assert_array_equal(got.packed, want)
        if not ties:
            host = pq.transform(X, device=False)          # numpy, as the reference
            _same_up_to_exact_ties(oracle, pq, X, got.packed, host.packed)


@pytest.mark.parametrize("tag", G6_TAGS)
def test_encode_reproduces_reference_codes(tag):
    g = golden(f"g6_ivf_{tag}.npz")
    codes, ids = split_lists(g)
    pq = _pq(g["pq_centers"], 2, g["R"] if "R" in g else None)
    for l, rows in enumerate(ids):
        if len(rows):
            np.testing.assert_array_equal(pq.transform(g["data"][rows], device=True).packed, codes[l])
    np.testing.assert_array_equal(pq.transform(g["active_centers"], device=True).packed,
                                  g["center_codes"])


def _assign(X, Y, k, metric):
    from tinyknn_amd import IVF
    ivf = IVF(metric, len(Y))
    ivf.all_centers = Y
    return ivf._nearest_on_device(X, k)


@pytest.mark.parametrize("metric,y64,k,L", [("euclidean", False, 1, 244), ("angular", False, 2, 1087),
                                            ("angular", False, 1, 1087), ("euclidean", False, 1, 33),
                                            ("angular", True, 1, 40), ("euclidean", True, 2, 300),
                                            ("euclidean", False, 2, 2)])
def test_assign_vs_oracle_and_numpy(oracle, metric, y64, k, L):
    from tinyknn_amd.utils import knn_brute
    rng = np.random.RandomState(7)
    n, d = 2317, 100                                      # 23 whole chunks + 17 rows in numpy
    X = rng.randn(n, d).astype(np.float32)
    Y = rng.randn(L, d).astype(np.float64 if y64 else np.float32)
    got = _assign(X, Y, k, metric)
    if k < L:      # k == L: bottom_k_2d returns arange without looking (utils.py:29-30)
        np.testing.assert_array_equal(got[:2300], oracle.assign(X[:2300], Y, k, metric))
    np.testing.assert_array_equal(got, knn_brute(X, Y, k, metric))
    if L > 10:     # exact ties (duplicated centres, incl. centre 0; points ON centres): the
        Y[7] = Y[0]                                       # oracle's dumb_select order
        Y[9] = Y[3]
        Y[L - 1] = Y[0]
        X[:50] = Y[rng.randint(L, size=50)].astype(np.float32)
        got = _assign(X, Y, k, metric)
        np.testing.assert_array_equal(got[:2300], oracle.assign(X[:2300], Y, k, metric))


@pytest.mark.parametrize("tag", [t for t in G6_TAGS if "f64" not in t])
def test_assign_reproduces_reference_lists(tag):
    g = golden(f"g6_ivf_{tag}.npz")
    k = int(g["build_probes"])
    near = _assign(np.ascontiguousarray(g["data"], dtype=np.float32), g["active_centers"], k,
                   str(g["metric"]))
    sizes = g["list_sizes"]
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    for l in range(len(sizes)):
        want = g["ids"][ioff[l]:ioff[l + 1]]
        o = 0
        for j in range(k):
            run = np.nonzero(near[:, j] == l)[0]
            np.testing.assert_array_equal(run, np.sort(want[o:o + len(run)]))
            o += len(run)
        assert o == len(want)


@pytest.mark.parametrize("metric,d,rot,probes", [("angular", 100, False, 1), ("euclidean", 128, True, 2)])
def test_device_build_equals_host_build(metric, d, rot, probes):
    """IVF.build(device=True) == IVF.build(device=False): same lists, ids, codes."""
    from tinyknn_amd import IVF, FastPQ
    rng = np.random.RandomState(1)
    np.random.seed(7)       # sklearn's KMeans and FastPQ.fit draw from numpy's global generator: the same fit every run
    n = 20037
    cent = rng.randn(60, d)
    X = (cent[rng.randint(60, size=n)] + 0.6 * rng.randn(n, d)).astype(np.float32)
    a = IVF(metric, 141, FastPQ(2))
    a.fit(X[:8000])
    b = IVF(metric, 141, FastPQ(2))
    b.all_centers, b.pq = a.all_centers, a.pq
    a.build(X, n_probes=probes, device=False)
    b.build(X, n_probes=probes, device=True)
    assert (a.pq.R is not None) == rot
    np.testing.assert_array_equal(a.active_centers, b.active_centers)
    np.testing.assert_array_equal(a.pq_transformed_centers.packed, b.pq_transformed_centers.packed)
    for l in range(len(a.active_centers)):
        np.testing.assert_array_equal(a.ids[l], b.ids[l])
        ta, tb = a.pq_transformed_points[l], b.pq_transformed_points[l]
        assert isinstance(ta, np.ndarray) == isinstance(tb, np.ndarray)
        if not isinstance(ta, np.ndarray):
            assert ta.size == tb.size
            np.testing.assert_array_equal(ta.packed, tb.packed)
