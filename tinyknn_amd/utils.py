"""numpy helpers with the reference's names and semantics (tinyknn/utils.py).

Only knn_brute1 is on the hot path; it runs on the GPU (tk_knn_brute1).  The rest
are host-side bookkeeping used by fit/build and by the recall measurements.
"""
import time
from contextlib import contextmanager

import numpy as np

from . import _lib


def pad1(arr, m):
    """Zero-pad a vector to a multiple of m.  reference: utils.py:6-11"""
    (s,) = arr.shape
    out = np.zeros((s + (-s) % m,), dtype=arr.dtype)
    out[:s] = arr
    return out


def pad2(arr, m1, m2):
    """Zero-pad a matrix to multiples of (m1, m2).  reference: utils.py:14-19"""
    s1, s2 = arr.shape
    out = np.zeros((s1 + (-s1) % m1, s2 + (-s2) % m2), dtype=arr.dtype)
    out[:s1, :s2] = arr
    return out


def bottom_k(arr, k):
    """reference: utils.py:22-25"""
    if k >= len(arr):
        return np.arange(len(arr))
    return np.argpartition(arr, k)[:k]


def bottom_k_2d(arr, k):
    """reference: utils.py:28-31"""
    if k >= arr.shape[1]:
        return np.resize(np.arange(arr.shape[1]), arr.shape)
    return np.argpartition(arr, k, axis=1)[:, :k]


@contextmanager
def timer(verbose, text):
    """reference: utils.py:34-41"""
    if verbose:
        print(text)
        start = time.time()
    yield
    if verbose:
        print(f"Took {time.time() - start:.1f}s")


def cdist(X, Y, chunk=100):
    """Squared Euclidean distances R[i, j] = |X_i - Y_j|^2.  reference: utils.py:44-63"""
    nx = np.einsum("ij,ij->i", X, X)
    ny = np.einsum("ij,ij->i", Y, Y)
    res = np.zeros((nx.size, ny.size))
    for i in range(0, nx.size, chunk):
        res[i:i + chunk] = nx[i:i + chunk, None] + ny
        res[i:i + chunk] -= 2 * X[i:i + chunk] @ Y.T
    return res


def knn_brute(X, Y, k, metric="euclidean", chunk=100):
    """k nearest rows of Y for every row of X (host numpy; ground truth and
    build-time assignment).  reference: utils.py:66-86"""
    assert k <= Y.shape[0], f"Can't find knn with {k=} and {Y.shape[0]} targets."
    if metric == "angular":
        X = X / np.linalg.norm(X, axis=1, keepdims=True)
        Y = Y / np.linalg.norm(Y, axis=1, keepdims=True)
    elif metric != "euclidean":
        raise ValueError(f"Metric not supported: {metric}")
    n = X.shape[0]
    res = np.zeros((n, k), dtype=int)
    ynorm = np.einsum("ij,ij->i", Y, Y)
    for i in range(0, n, chunk):
        xc = X[i:i + chunk]
        xnorm = np.einsum("ij,ij->i", xc, xc)
        part = xnorm[:, None] + ynorm[None] - 2 * xc @ Y.T
        res[i:i + chunk] = bottom_k_2d(part, k)
    return res


def knn_brute1(x, Y, k):
    """Positions of the k rows of Y closest to x, ascending (GPU: rescore.hip).
    reference: utils.py:89-92.  float32 arithmetic when both are float32, float64
    otherwise, as numpy promotes `Y - x`."""
    x = np.asarray(x)
    Y = np.asarray(Y)
    x = np.ascontiguousarray(x, dtype=np.float32 if x.dtype == np.float32 else np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float32 if Y.dtype == np.float32 else np.float64)
    n, d = Y.shape
    assert x.shape == (d,)
    kk = min(int(k), n)
    out = np.zeros(max(kk, 1), dtype=np.int64)
    got = _lib.check(_lib.lib().tk_knn_brute1(
        x.ctypes.data, int(x.dtype == np.float64), Y.ctypes.data, int(Y.dtype == np.float64),
        n, d, int(k), _lib.ptr(out, _lib._i64p)))
    return out[:got]


def group_data_by_indices(X, indices, k):
    """Rows of X grouped by the list ids in `indices` (N, c): returns (parts, ids)
    with X[i] in parts[indices[i, j]] for every j.  reference: utils.py:95-162
    (same order inside each part: column by column, argsort order within a column)."""
    assert 0 <= np.min(indices) and np.max(indices) < k
    parts = [[] for _ in range(k)]
    ids = [[] for _ in range(k)]
    for j in range(indices.shape[1]):
        col = indices[:, j]
        order = np.argsort(col)
        uniq, counts = np.unique(col[order], return_counts=True)
        start = 0
        for g, cnt in zip(uniq, counts):
            sel = order[start:start + cnt]
            parts[g].append(X[sel])
            ids[g].append(sel)
            start += cnt
    for part, idl in zip(parts, ids):
        if not part:
            part.append(np.empty((0, X.shape[1])))
            idl.append(np.empty(0))
    return [np.vstack(p) for p in parts], [np.hstack(i) for i in ids]
