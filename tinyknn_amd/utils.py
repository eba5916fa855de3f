"""Host-side helpers under the reference's names (tinyknn/utils.py), written for this repository.

Only knn_brute1 is on the hot path; it runs on the GPU (tk_knn_brute1).  The rest is host
bookkeeping for fit / build and the recall measurements: a drop-in needs the same names, argument
meaning and — where list membership or result order depends on it — the same numpy operations in the
same order (argpartition, the |x|^2 + |y|^2 - 2 x.y expansion).  Those few expressions are the
contract, cited per function; nothing else is taken from the reference (AGPL-3.0-or-later).
"""
import time

import numpy as np

from . import _lib


def _zero_extended(arr, multiples):
    """`arr` inside a zero array whose every axis length is rounded up to its multiple."""
    shape = tuple(-(-n // m) * m for n, m in zip(arr.shape, multiples))
    out = np.zeros(shape, dtype=arr.dtype)
    out[tuple(slice(0, n) for n in arr.shape)] = arr
    return out


def pad1(arr, m):
    """A vector zero-extended to a multiple of m (what utils.py:6-11 of the reference returns).
    Like there, anything but a 1-D array is an error."""
    (_,) = arr.shape
    return _zero_extended(arr, (m,))


def pad2(arr, m1, m2):
    """A matrix zero-extended to multiples of (m1, m2) (utils.py:14-19)."""
    _, _ = arr.shape
    return _zero_extended(arr, (m1, m2))


def _k_smallest(arr, k, axis):
    """Positions of the k smallest entries along `axis` in numpy's argpartition order — that order is
    part of the contract (it decides list membership and the order of returned ids: SURVEY 8c) — or
    every position, ascending, when there are no more than k."""
    n = arr.shape[axis]
    if k >= n:
        every = np.arange(n)
        return every if arr.ndim == 1 else np.resize(every, arr.shape)
    part = np.argpartition(arr, k, axis=axis)
    return part[:k] if arr.ndim == 1 else part[:, :k]


def bottom_k(arr, k):
    """utils.py:22-25"""
    return _k_smallest(arr, k, 0)


def bottom_k_2d(arr, k):
    """utils.py:28-31"""
    return _k_smallest(arr, k, 1)


class timer:
    """`with timer(verbose, text):` prints the text, then "Took ...s" (the lines utils.py:34-41
    prints; examples and IVF.fit / build use them).  An exception inside the block passes through
    without the second line, as a generator-based context manager would let it."""

    def __init__(self, verbose, text):
        self.verbose, self.text = verbose, text

    def __enter__(self):
        if self.verbose:
            print(self.text)
            self.t0 = time.time()

    def __exit__(self, exc_type, exc, tb):
        if self.verbose and exc_type is None:
            print(f"Took {time.time() - self.t0:.1f}s")
        return False


def _sq_norms(A):
    return np.einsum("ij,ij->i", A, A)


def cdist(X, Y, chunk=100):
    """Squared Euclidean distances D[i, j] = |X_i - Y_j|^2 as float64, `chunk` rows of X at a time,
    by the expansion |x|^2 + |y|^2 - 2 x.y in the reference's operation order (utils.py:44-63)."""
    nx, ny = _sq_norms(X), _sq_norms(Y)
    out = np.zeros((len(nx), len(ny)))
    for lo in range(0, len(nx), chunk):
        rows = slice(lo, lo + chunk)
        out[rows] = nx[rows, None] + ny
        out[rows] -= 2 * X[rows] @ Y.T
    return out


def knn_brute(X, Y, k, metric="euclidean", chunk=100):
    """k nearest rows of Y for every row of X on the host (ground truth, build-time assignment).
    The distance expression `|x|^2 + |y|^2 - 2 x @ Y.T`, its dtype and the argpartition behind it are
    the reference's (utils.py:66-86): list membership of IVF.build depends on their rounding."""
    assert k <= Y.shape[0], f"Can't find knn with {k=} and {Y.shape[0]} targets."
    if metric not in ("angular", "euclidean"):
        raise ValueError(f"Metric not supported: {metric}")
    if metric == "angular":
        X, Y = (A / np.linalg.norm(A, axis=1, keepdims=True) for A in (X, Y))
    ynorm = _sq_norms(Y)
    out = np.zeros((X.shape[0], k), dtype=int)
    for lo in range(0, X.shape[0], chunk):
        xc = X[lo:lo + chunk]
        out[lo:lo + chunk] = bottom_k_2d(_sq_norms(xc)[:, None] + ynorm[None] - 2 * xc @ Y.T, k)
    return out


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
