"""Offline build path: the oracle's restatement of FastPQ.transform and
of IVF.build's list assignment against (a) the codes and list memberships the compiled
reference produced (golden IVF fixtures) and (b) numpy itself on the shapes the path uses."""
import numpy as np
import pytest

from conftest import G6_TAGS, golden, split_lists


def _lists(g):
    sizes = g["list_sizes"]
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    return [g["ids"][ioff[i]:ioff[i + 1]] for i in range(len(sizes))]


@pytest.mark.parametrize("tag", G6_TAGS)
def test_encode_reproduces_reference_codes(oracle, tag):
    g = golden(f"g6_ivf_{tag}.npz")
    codes, ids = split_lists(g)
    R = g["R"] if "R" in g else None
    for l, rows in enumerate(ids):
        if len(rows) == 0:
            continue
        n, packed = oracle.fastpq_transform(g["pq_centers"], 2, R, g["data"][rows])
        assert n == len(rows)
        np.testing.assert_array_equal(packed, codes[l], err_msg=f"{tag} list {l}")
    # the coded coarse centres (ivf.py:96)
    n, packed = oracle.fastpq_transform(g["pq_centers"], 2, R, g["active_centers"])
    np.testing.assert_array_equal(packed, g["center_codes"])


@pytest.mark.parametrize("tag", G6_TAGS)
def test_assignment_reproduces_reference_lists(oracle, tag):
    """ids[l] = rows whose 1st nearest centre is l, then rows whose 2nd is l, each run in the
    order of an (unstable) np.argsort (group_data_by_indices, utils.py:95-150): the MEMBERS
    of each run are what the assignment decides."""
    g = golden(f"g6_ivf_{tag}.npz")
    k = int(g["build_probes"])
    if k > 16:
        pytest.skip("more lists per row than the restatement takes")
    X = np.ascontiguousarray(g["data"], dtype=np.float32)
    if g["data"].dtype != np.float32:
        pytest.skip("float64 data: knn_brute runs in float64 there")
    near = oracle.assign(X, g["active_centers"], k, str(g["metric"]))
    lists = _lists(g)
    for l, want in enumerate(lists):
        o = 0
        for j in range(k):
            run = np.nonzero(near[:, j] == l)[0]
            np.testing.assert_array_equal(run, np.sort(want[o:o + len(run)]),
                                          err_msg=f"{tag} list {l} probe {j}")
            o += len(run)
        assert o == len(want)


@pytest.mark.parametrize("k", [3, 4, 9])
def test_assignment_beyond_two_lists_matches_numpy_here(oracle, k):
    """k >= 3: numpy's argpartition leaves the order of the first k to its quickselect (SIMD on this
    host); it comes out ascending here, which is what the oracle (and the device) restate."""
    from tinyknn_amd.utils import knn_brute
    rng = np.random.RandomState(11 + k)
    X = rng.randn(700, 24).astype(np.float32)
    for Y in (rng.randn(61, 24).astype(np.float32), rng.randn(300, 24)):
        for metric in ("euclidean", "angular"):
            np.testing.assert_array_equal(oracle.assign(X, Y, k, metric), knn_brute(X, Y, k=k, metric=metric))


@pytest.mark.parametrize("dpb,d,rot", [(2, 100, False), (1, 100, False), (4, 100, False), (2, 128, True),
                                       (2, 20, False)])
def test_encode_vs_numpy(oracle, dpb, d, rot):
    rng = np.random.RandomState(5)
    n = 1600
    pad = (-d) % (4 * dpb)
    X = rng.randn(n, d).astype(np.float32)
    R = np.linalg.qr(rng.randn(d + pad, d + pad))[0][:64] if rot else None
    dq = 64 if rot else d + pad
    centers = (rng.randn(16, dq) * 0.7).astype(np.float32)
    Xp = np.concatenate([X, np.zeros((n, pad), np.float32)], axis=1)
    if rot:
        Xp = Xp @ R.T
    cols = Xp.reshape(n, dq // dpb, dpb).transpose(1, 0, 2)
    ccols = centers.reshape(16, -1, dpb).transpose(1, 0, 2)
    want = []
    for col, code in zip(cols, ccols):       # knn_brute(col, code, 1), utils.py:66-86
        yn = np.einsum("ij,ij->i", code, code)
        lab = np.empty((n, 1), dtype=int)
        for i in range(0, n, 100):
            xc = col[i:i + 100]
            part = np.einsum("ij,ij->i", xc, xc)[:, None] + yn[None] - 2 * xc @ code.T
            lab[i:i + 100] = np.argpartition(part, 1, axis=1)[:, :1]
        want.append(lab)
    want = np.hstack(want).astype(np.uint8)
    np.testing.assert_array_equal(oracle.encode_pq(centers, dpb, Xp), want)


@pytest.mark.parametrize("metric,y64,k", [("euclidean", False, 1), ("angular", False, 2),
                                          ("angular", True, 1), ("euclidean", True, 2)])
def test_assign_vs_numpy(oracle, metric, y64, k):
    rng = np.random.RandomState(11)
    n, d, L = 700, 100, 244
    X = rng.randn(n, d).astype(np.float32)
    Y = rng.randn(L, d).astype(np.float64 if y64 else np.float32)
    Xn, Yn = X, Y
    if metric == "angular":
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        Yn = Y / np.linalg.norm(Y, axis=1, keepdims=True)
    want = np.zeros((n, k), dtype=int)
    yn = np.einsum("ij,ij->i", Yn, Yn)
    for i in range(0, n, 100):
        xc = Xn[i:i + 100]
        part = np.einsum("ij,ij->i", xc, xc)[:, None] + yn[None] - 2 * xc @ Yn.T
        want[i:i + 100] = np.argpartition(part, k, axis=1)[:, :k]
    np.testing.assert_array_equal(oracle.assign(X, Y, k, metric), want)
