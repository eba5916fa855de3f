"""FastPQ / _FastDistanceTable with the reference's API (tinyknn/fast_pq.py), the
query side running on the MI355X through libtinyknn_hip.so.

fit / transform are offline host code (sklearn k-means, numpy) with the
reference's semantics; distance_table, estimate_distances and top — the hot
path — call the HIP kernels (tables.hip, adc_scan.hip, heap.hip, rescore.hip).
"""
import warnings
from collections import namedtuple

import numpy as np

from . import _lib
from ._fast_pq import init_heap
from ._transform import transform_data, transform_tables
from .utils import knn_brute, knn_brute1, pad1, pad2

# The reference selects its AVX kernels and a 4-block padding at import time
# (fast_pq.py:21-27); same switch, same names.
avx = True
if avx:
    from ._fast_pq_avx import query_pq_avx as query_pq, estimate_pq_avx as estimate_pq
    dpad = 4
else:
    from ._fast_pq import query_pq_sse as query_pq, estimate_pq_sse as estimate_pq
    dpad = 2

TransformedData = namedtuple("TransformedData", "size packed")


def _gauss_polar_code():
    """The fixed 16-point code for a 2-d Gaussian: origin, 6 points on the unit
    circle, 9 on radius 2 (reference: fast_pq.py:129-136)."""
    pts = [(0.0, 0.0)]
    for radius, count in ((1, 6), (2, 9)):
        for theta in np.linspace(0, 2 * np.pi, count, endpoint=False):
            pts.append((radius * np.cos(theta), radius * np.sin(theta)))
    return np.array(pts)


# default of FastPQ.transform / IVF.build: False = host numpy as in the reference, True = the
# nearest-centroid / nearest-centre searches on the GPU (build.hip; same codes and lists)
device_build = False


class FastPQ:
    """4-bit product quantizer: 16 centroids per block of `dims_per_block` dims.
    reference: fast_pq.py:33-252"""

    def __init__(self, dims_per_block, use_kmeans=True, rotate_dim=64):
        self.dims_per_block = dims_per_block
        self.centers = None          # (16, d) float32 after fit
        self.sqrt_n_blocks = None
        self.use_kmeans = use_kmeans
        self.rotate_dim = rotate_dim
        self.R = None                # optional (rotate_dim, d) float64 rotation

    # ---- offline ---------------------------------------------------------
    def fit(self, data, verbose=False):
        """reference: fast_pq.py:50-104"""
        assert data.size > 0, "Can't fit no data"
        true_d = data.shape[1]
        dpb = self.dims_per_block
        data = pad2(data, 16, dpad * dpb)
        d = data.shape[1]
        # random rotation, truncated to rotate_dim rows — skipped for 100-d input
        # exactly as the reference does (fast_pq.py:77-82)
        if self.rotate_dim is not None and true_d != 100:
            from scipy.stats import ortho_group
            self.R = ortho_group.rvs(dim=d)
            if d > self.rotate_dim:
                d = self.rotate_dim
                self.R = self.R[:d]
            data = data @ self.R.T
        codebooks = self._fit_code(data, verbose=verbose)     # (M, 16, dpb)
        self.centers = np.array(codebooks, dtype=np.float32).transpose(1, 0, 2).reshape(16, d)
        self.sqrt_n_blocks = np.sqrt(d // dpb)
        return self

    def fit_transform(self, data, verbose=False):
        return self.fit(data, verbose).transform(data, verbose)

    def _fit_code(self, data, verbose=False):
        """One 16-entry codebook per block.  reference: fast_pq.py:109-145"""
        n, d = data.shape
        dpb = self.dims_per_block
        blocks = data.reshape(n, d // dpb, dpb).transpose(1, 0, 2)
        it = range(d // dpb)
        if verbose:
            import tqdm
            it = tqdm.tqdm(it)
        out = []
        if self.use_kmeans:
            import sklearn.cluster
            from sklearn.exceptions import ConvergenceWarning
            km = sklearn.cluster.KMeans(16, n_init=2)
            for m in it:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", category=ConvergenceWarning)
                    km.fit(blocks[m])
                out.append(km.cluster_centers_.copy())
        else:
            assert dpb == 2, "Fixed code only defined for dpb = 2"
            base = _gauss_polar_code()
            for m in it:
                col = blocks[m]
                chol = np.linalg.cholesky(np.cov(col.T, bias=True))
                out.append(base @ chol.T + np.mean(col, axis=0))
        return out

    def encode_labels(self, data, device=None):
        """(n, M) uint8 nearest-centroid labels of rows that are already padded (and
        rotated): the body of transform, fast_pq.py:171-182.  device: run the search on the
        GPU (build.hip, same labels); None = the module default `device_build`."""
        dpb = self.dims_per_block
        n, d = data.shape
        M = d // dpb
        if (device_build if device is None else device) and 16 % dpb == 0:
            is64 = data.dtype != np.float32
            data = np.ascontiguousarray(data, dtype=np.float64 if is64 else np.float32)
            c32 = np.ascontiguousarray(self.centers, dtype=np.float32)
            labels = np.empty((n, M), dtype=np.uint8)
            _lib.check(_lib.lib().tk_encode_pq(_lib.ptr(c32, _lib._f32p), d, dpb, data.ctypes.data,
                                               int(is64), n, _lib.ptr(labels, _lib._u8p)))
            return labels
        blocks = data.reshape(n, M, dpb).transpose(1, 0, 2)
        books = self.centers.reshape(16, M, dpb).transpose(1, 0, 2)
        return np.hstack([knn_brute(col, book, 1) for col, book in zip(blocks, books)]).astype(np.uint8)

    def transform(self, data, verbose=False, device=None):
        """Encode rows to 4-bit codes in the Quick-ADC layout.
        reference: fast_pq.py:147-184.  device=True: the nearest-centroid search runs on
        the GPU (padding and the rotation GEMM stay numpy, as in the reference)."""
        assert self.centers is not None, "PQ has not been fitted"
        if data.size == 0:
            return data
        true_n = data.shape[0]
        dpb = self.dims_per_block
        data = pad2(data, 16, dpad * dpb)
        if self.R is not None:
            data = data @ self.R.T
        if device_build if device is None else device:
            return TransformedData(true_n, transform_data(self.encode_labels(data, True)))
        n, d = data.shape
        M = d // dpb
        blocks = data.reshape(n, M, dpb).transpose(1, 0, 2)
        books = self.centers.reshape(16, M, dpb).transpose(1, 0, 2)
        pairs = zip(blocks, books)
        if verbose:
            import tqdm
            pairs = tqdm.tqdm(pairs, total=M)
        codes = np.hstack([knn_brute(col, book, 1) for col, book in pairs]).astype(np.uint8)
        assert codes.shape == (n, M)
        return TransformedData(true_n, transform_data(codes))

    # ---- query side (GPU) --------------------------------------------------
    def _pq_query(self, q):
        q = pad1(q, dpad * self.dims_per_block)
        if self.R is not None:
            q = q @ self.R.T          # float64 GEMV on the host, as the reference
        return q

    def _table(self, q, signed):
        raw_q = q
        qp = self._pq_query(np.asarray(q))
        tables, shift, scale = build_tables(self, qp[None, :], signed)
        return _FastDistanceTable(qp, raw_q, tables[0], shift[0], scale[0], signed=signed)

    def distance_table(self, q):
        """Signed int8 lookup table for one query.  reference: fast_pq.py:186-222"""
        return self._table(q, True)

    def udistance_table(self, q):
        """Unsigned variant (experimental).  reference: fast_pq.py:224-252"""
        return self._table(q, False)


def build_tables(pq, q_pq, signed=True):
    """Tables for a batch of padded/rotated queries (nq, dq) on the GPU
    (tables.hip).  Returns (tables uint64 (nq, 2M), shift (nq,), scale (nq,))."""
    centers = pq.centers
    f_order = int(not centers.flags.c_contiguous)
    c32 = np.ascontiguousarray(centers, dtype=np.float32)
    dq = c32.shape[1]
    dpb = pq.dims_per_block
    M = dq // dpb
    q_pq = np.asarray(q_pq)
    assert q_pq.ndim == 2 and q_pq.shape[1] == dq
    is64 = q_pq.dtype != np.float32
    q_pq = np.ascontiguousarray(q_pq, dtype=np.float64 if is64 else np.float32)
    nq = q_pq.shape[0]
    tables = np.zeros((nq, M, 16), dtype=np.uint8)
    shift = np.zeros(nq, dtype=q_pq.dtype)
    scale = np.zeros(nq, dtype=np.float64)
    if signed:
        aux = (float(pq.sqrt_n_blocks), 0.0)
    else:
        aux = (float(np.log(M)), float(np.sqrt(M)))
    _lib.check(_lib.lib().tk_build_tables(
        _lib.ptr(c32, _lib._f32p), dq, dpb, f_order, q_pq.ctypes.data, int(is64), nq,
        aux[0], aux[1], int(bool(signed)), _lib.ptr(tables, _lib._u8p), shift.ctypes.data,
        _lib.ptr(scale, _lib._f64p)))
    return tables.reshape(nq, M * 16).view(np.uint64), shift, scale


def estimate_batch(pq, transformed_data, qs, signed=True):
    """estimate_distances for many queries in one call: tables for all rows of `qs`
    (tables.hip), then one pass of the list-major scan kernel over the resident
    codes, four queries per pass.  Returns (nq, n) int8 / uint8 — row i equals
    pq.distance_table(qs[i]).estimate_distances(transformed_data)."""
    from ._fast_pq import device_codes, _buf
    true_n, packed = transformed_data
    packed = _buf(packed, np.uint64, 2, "data")
    qp = np.stack([pq._pq_query(np.asarray(q)) for q in qs])
    tables, _, _ = build_tables(pq, qp, signed)
    order = _lib.ORDER_AVX if avx else _lib.ORDER_SSE
    nq = len(qs)
    out = np.zeros((nq, 2 * len(packed)), dtype=np.uint64)
    h = device_codes(packed)
    if h:
        _lib.check(_lib.lib().tk_codes_estimate(h, _lib.ptr(tables, _lib._u64p), nq,
                                                _lib.ptr(out, _lib._u64p), int(bool(signed)), order))
    else:
        _lib.check(_lib.lib().tk_estimate_pq_batch(
            _lib.ptr(packed, _lib._u64p), packed.shape[0], packed.shape[1],
            _lib.ptr(tables, _lib._u64p), nq, _lib.ptr(out, _lib._u64p), int(bool(signed)), order))
    return out.view(np.int8 if signed else np.uint8)[:, :true_n]


class FlatTop:
    """`pq.distance_table(q).top(transformed_data, data, k)` (fast_pq.py:284-312) for a BATCH of
    queries: the codes and the float32 rows go to HBM once, a call runs tables, one list-major
    scan of all rows for all queries, the exact lane-per-query heap replay and the exact
    rescoring on the device (tk_index_top_centers) — row i of the result equals the reference's
    per-query call.  float32 rows and queries (the per-query API covers float64)."""

    def __init__(self, pq, transformed_data, data):
        true_n, packed = transformed_data
        data = np.ascontiguousarray(data, dtype=np.float32)
        assert len(data) == true_n
        self.pq, self.n, self.d = pq, true_n, data.shape[1]
        L = _lib.lib()
        self._h = L.tk_index_create()
        if not self._h:
            raise _lib.TinyKnnHipError(L.tk_last_error().decode() or "tk_index_create failed")
        c32 = np.ascontiguousarray(pq.centers, dtype=np.float32)
        order = _lib.ORDER_AVX if avx else _lib.ORDER_SSE
        _lib.check(L.tk_index_set_pq(self._h, _lib.ptr(c32, _lib._f32p), c32.shape[1], pq.dims_per_block,
                                     int(not pq.centers.flags.c_contiguous), float(pq.sqrt_n_blocks), order))
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        _lib.check(L.tk_index_set_centers(self._h, _lib.ptr(data, _lib._f32p), true_n, self.d,
                                          _lib.ptr(packed, _lib._u64p), packed.shape[0]))

    def top(self, qs, k=1):
        """(nq, d) float32 queries -> (nq, min(k, n)) int64 row ids."""
        qs = np.ascontiguousarray(qs, dtype=np.float32)
        assert qs.ndim == 2 and qs.shape[1] == self.d
        k = min(int(k), self.n)
        from . import _front
        R = self.pq.R
        if len(qs) and _front.bind() and (R is None or (min(R.shape) >= 2 and R.dtype == np.float64)):
            # pad1 / q @ R.T row by row through numpy's own BLAS on a thread pool (exact)
            pad = (-self.d) % (dpad * self.pq.dims_per_block)
            _, qp = _front.prepare(qs.copy(), False, R, pad)
        else:
            qp = np.stack([self.pq._pq_query(q) for q in qs]) if len(qs) else np.zeros((0, 1), np.float32)
        qp = np.ascontiguousarray(qp)
        out = np.full((len(qs), k), -1, dtype=np.int64)
        _lib.check(_lib.lib().tk_index_top_centers(
            self._h, _lib.ptr(qs, _lib._f32p), qp.ctypes.data, int(qp.dtype != np.float32), len(qs), k,
            _lib.ptr(out, _lib._i64p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            if _lib.owns_handles():
                _lib.lib().tk_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _FastDistanceTable:
    """reference: fast_pq.py:255-312"""

    def __init__(self, q, raw_q, transformed_tables, mean, scale, signed):
        self.q = q
        self.raw_q = raw_q
        self.tables = transformed_tables
        self.mean = mean
        self.scale = scale
        self.signed = signed

    def __repr__(self):
        return (f"FastDistanceTable(q={self.q}, tables={self.tables}, mean={self.mean}, "
                f"scale={self.scale}, signed={self.signed})")

    def estimate_distances(self, transformed_data, out=None, rescale=False):
        """int8/uint8 estimates of every row (adc_scan.hip).
        reference: fast_pq.py:270-282"""
        true_n, packed = transformed_data
        if out is None:
            out = np.zeros(2 * len(packed), dtype=np.uint64)
        estimate_pq(packed, self.tables, out, self.signed)
        res = out.view(np.int8 if self.signed else np.uint8)[:true_n]
        if not rescale:
            return res
        as_float = np.ascontiguousarray(res, dtype=np.float32)
        return self.q @ self.q + (as_float / self.scale + self.mean)

    def top(self, transformed_data, data, k=1, rescore=None):
        """Two-pass nearest rows: PQ heap of `rescore`, then exact distances.
        reference: fast_pq.py:284-312"""
        true_n, packed = transformed_data
        assert len(data) == true_n
        k = min(k, true_n)
        if not rescore:
            rescore = min(2 * k + 10, true_n)
        assert true_n >= rescore >= k
        indices = np.zeros((rescore,), dtype=np.int64)
        values = np.zeros((rescore,), dtype=np.int32)
        init_heap(indices, values, self.signed)
        query_pq(packed, true_n, self.tables, indices, values, self.signed)
        if rescore <= k:
            return indices
        best = knn_brute1(self.raw_q, data[indices], k)
        return indices[best]
