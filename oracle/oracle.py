"""ctypes front-end of the CPU oracle (oracle/tinyknn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under tinyknn_amd/ imports this module.

Function names mirror the reference entry points they restate
(/root/reference/tinyknn/_fast_pq.pyx, _fast_pq_256.pyx, _transform.py,
fast_pq.py, ivf.py, utils.py); see the C file for file:line citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtinyknn_oracle.so")

ORDER_SSE, ORDER_AVX = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "tinyknn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtinyknn_oracle.so"])
    return _SO


_lib = None


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class _Index(C.Structure):
    _fields_ = [
        ("d", C.c_int), ("dq", C.c_int), ("dpb", C.c_int), ("M", C.c_int),
        ("order", C.c_int), ("rotated", C.c_int), ("sqrt_n_blocks", C.c_double),
        ("pq_centers", C.c_void_p), ("n_lists", C.c_int64),
        ("center_codes", C.c_void_p), ("center_chunks", C.c_int64),
        ("active_centers", C.c_void_p), ("list_chunk_off", C.c_void_p),
        ("list_n", C.c_void_p), ("codes", C.c_void_p), ("ids_off", C.c_void_p),
        ("ids", C.c_void_p), ("data", C.c_void_p), ("N", C.c_int64), ("data_is_f64", C.c_int),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.tko_pairwise_sum_f32.restype = C.c_float
        _lib.tko_pairwise_sum_f64.restype = C.c_double
        _lib.tko_einsum_dot_f32.restype = C.c_float
        _lib.tko_einsum_dot_f64.restype = C.c_double
        _lib.tko_bottom_k.restype = C.c_int64
        _lib.tko_ivf_query.restype = C.c_int64
        _lib.tko_ivf_query.argtypes = [
            C.POINTER(_Index), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.tko_ivf_query_batch.argtypes = [
            C.POINTER(_Index), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
            C.c_int, C.c_void_p]
    return _lib


def force_scalar(on):
    lib().tko_force_scalar(int(bool(on)))


def simd():
    return bool(lib().tko_simd())


# ---- layout ---------------------------------------------------------------

def transform_data(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, M = codes.shape
    assert n % 16 == 0 and M % 2 == 0
    out = np.zeros((n // 16, M), dtype=np.uint64)
    lib().tko_pack_codes(_p(codes, C.c_uint8), C.c_int64(n), M, _p(out, C.c_uint64))
    return out


def unpack(packed):
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    chunks, M = packed.shape
    out = np.zeros((chunks * 16, M), dtype=np.uint8)
    lib().tko_unpack_codes(_p(packed, C.c_uint64), C.c_int64(chunks), M, _p(out, C.c_uint8))
    return out


def transform_tables(table):
    table = np.ascontiguousarray(table, dtype=np.uint8)
    M, b = table.shape
    assert b == 16
    out = np.zeros(2 * M, dtype=np.uint64)
    lib().tko_transform_tables(_p(table, C.c_uint8), M, _p(out, C.c_uint64))
    return out


# ---- kernels --------------------------------------------------------------

def estimate_pq(data, tables, out, signd, order=ORDER_AVX):
    assert data.dtype == np.uint64 and data.flags.c_contiguous
    chunks, M = data.shape
    assert tables.shape == (2 * M,) and out.shape[0] >= 2 * chunks
    lib().tko_estimate_pq(_p(data, C.c_uint64), C.c_int64(chunks), M,
                          _p(tables, C.c_uint64), _p(out, C.c_uint64),
                          int(bool(signd)), order)


def init_heap(indices, vals, signd):
    lib().tko_init_heap(_p(indices, C.c_int64), _p(vals, C.c_int32), len(indices),
                        int(bool(signd)))


def insert(indices, vals, i, v):
    lib().tko_heap_insert(_p(indices, C.c_int64), _p(vals, C.c_int32), len(indices),
                          C.c_int64(int(i)), C.c_int32(int(v)))


def insert_is(indices, vals, i, v):
    lib().tko_heap_insert_is(_p(indices, C.c_int64), _p(vals, C.c_int32), len(indices),
                             C.c_int64(int(i)), C.c_int32(int(v)))


def query_pq(data, n, tables, indices, vals, signd, labels=None, order=ORDER_AVX,
             stats=None):
    assert data.dtype == np.uint64 and data.flags.c_contiguous
    chunks, M = data.shape
    assert indices.dtype == np.int64 and vals.dtype == np.int32
    lab = None if labels is None else _p(np.ascontiguousarray(labels, dtype=np.int64), C.c_int64)
    st = None if stats is None else _p(stats, C.c_int64)
    lib().tko_query_pq_stats(_p(data, C.c_uint64), C.c_int64(chunks), M, C.c_int64(n),
                             _p(tables, C.c_uint64), _p(indices, C.c_int64),
                             _p(vals, C.c_int32), len(indices), int(bool(signd)), lab,
                             order, st)


# ---- tables ---------------------------------------------------------------

def distance_table(centers, dpb, q_pq, sqrt_n_blocks, signed=True):
    """q_pq: the padded (and rotated, if any) query; dtype selects the f32/f64 path.
    Returns (table (M,16) uint8, shift, scale).  The memory order of `centers`
    (C, or the F-ordered view FastPQ.fit leaves for dims_per_block == 1) selects
    the summation order of the mean, as in numpy."""
    f_order = int(not centers.flags.c_contiguous)
    centers = np.ascontiguousarray(centers, dtype=np.float32)
    dq = centers.shape[1]
    M = dq // dpb
    table = np.zeros((M, 16), dtype=np.uint8)
    scale = C.c_double()
    if signed:
        aux = (C.c_double(float(sqrt_n_blocks)), C.c_int(f_order))
    else:
        aux = (C.c_double(float(np.log(M))), C.c_double(float(np.sqrt(M))))
    if q_pq.dtype == np.float32:
        q = np.ascontiguousarray(q_pq)
        shift = C.c_float()
        fn = lib().tko_distance_table_f32 if signed else lib().tko_udistance_table_f32
        fn(_p(centers, C.c_float), dq, dpb, _p(q, C.c_float), *aux, _p(table, C.c_uint8),
           C.byref(shift), C.byref(scale))
        return table, np.float32(shift.value), np.float64(scale.value)
    q = np.ascontiguousarray(q_pq, dtype=np.float64)
    shift = C.c_double()
    fn = lib().tko_distance_table_f64 if signed else lib().tko_udistance_table_f64
    fn(_p(centers, C.c_float), dq, dpb, _p(q, C.c_double), *aux, _p(table, C.c_uint8),
       C.byref(shift), C.byref(scale))
    return table, np.float64(shift.value), np.float64(scale.value)


# ---- rescoring ------------------------------------------------------------

def _fx(a):
    a = np.asarray(a)
    return np.ascontiguousarray(a, dtype=np.float32 if a.dtype == np.float32 else np.float64)


def sqdist_gather(x, Y, idx):
    x, Y = _fx(x), _fx(Y)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.zeros(len(idx), dtype=np.float64)
    lib().tko_sqdist_gather(C.c_void_p(x.ctypes.data), int(x.dtype == np.float64),
                            C.c_void_p(Y.ctypes.data), int(Y.dtype == np.float64),
                            C.c_int64(Y.shape[0]), Y.shape[1], _p(idx, C.c_int64),
                            C.c_int64(len(idx)), _p(out, C.c_double))
    return out


def bottom_k(dists, k):
    dists = np.ascontiguousarray(dists, dtype=np.float64)
    out = np.zeros(max(len(dists), 1), dtype=np.int64)
    n = lib().tko_bottom_k(_p(dists, C.c_double), C.c_int64(len(dists)), C.c_int64(k),
                           _p(out, C.c_int64))
    return out[:n]


def knn_brute1(x, Y_rows, k):
    """Y_rows already gathered (as the reference passes data[indices])."""
    d = sqdist_gather(x, Y_rows, np.arange(len(Y_rows)))
    return bottom_k(d, k)


# ---- IVF ------------------------------------------------------------------

class OracleIndex:
    """Flat, C-visible copy of a built IVF (the attributes the reference's IVF
    object holds after fit+build: ivf.py:14-17,77-102)."""

    def __init__(self, pq_centers, dpb, R, sqrt_n_blocks, active_centers,
                 center_codes, list_codes, list_sizes, ids, data, order=ORDER_AVX):
        self.pq_centers = np.ascontiguousarray(pq_centers, dtype=np.float32)
        self.dq = self.pq_centers.shape[1]
        self.dpb = int(dpb)
        self.M = self.dq // self.dpb
        self.R = None if R is None else np.ascontiguousarray(R, dtype=np.float64)
        self.active_centers = np.ascontiguousarray(active_centers, dtype=np.float32)
        self.center_codes = np.ascontiguousarray(center_codes, dtype=np.uint64)
        assert self.center_codes.shape[1] == self.M
        self.n_lists = len(list_codes)
        offs = np.zeros(self.n_lists + 1, dtype=np.int64)
        for i, c in enumerate(list_codes):
            offs[i + 1] = offs[i] + (0 if c is None else c.shape[0])
        self.list_chunk_off = offs
        self.codes = np.zeros((max(int(offs[-1]), 1), self.M), dtype=np.uint64)
        for i, c in enumerate(list_codes):
            if c is not None and c.shape[0]:
                self.codes[offs[i]:offs[i + 1]] = c
        self.list_n = np.ascontiguousarray(list_sizes, dtype=np.int64)
        self.ids_off = np.zeros(self.n_lists + 1, dtype=np.int64)
        self.ids_off[1:] = np.cumsum([len(x) for x in ids])
        self.ids = (np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int64) for x in ids]))
                    if self.n_lists else np.zeros(1, np.int64))
        if len(self.ids) == 0:
            self.ids = np.zeros(1, np.int64)
        self.data = _fx(data)
        self.d = self.data.shape[1]
        s = _Index()
        s.d, s.dq, s.dpb, s.M, s.order = self.d, self.dq, self.dpb, self.M, order
        s.rotated = 0 if self.R is None else 1
        s.sqrt_n_blocks = float(sqrt_n_blocks)
        s.pq_centers = self.pq_centers.ctypes.data
        s.n_lists = self.n_lists
        s.center_codes = self.center_codes.ctypes.data
        s.center_chunks = self.center_codes.shape[0]
        s.active_centers = self.active_centers.ctypes.data
        s.list_chunk_off = self.list_chunk_off.ctypes.data
        s.list_n = self.list_n.ctypes.data
        s.codes = self.codes.ctypes.data
        s.ids_off = self.ids_off.ctypes.data
        s.ids = self.ids.ctypes.data
        s.data = self.data.ctypes.data
        s.N = self.data.shape[0]
        s.data_is_f64 = int(self.data.dtype == np.float64)
        self._s = s

    def pq_query(self, qn):
        """Pad (fast_pq.py:202) and, if fitted with a rotation, rotate in numpy
        float64 (fast_pq.py:203-204; a BLAS GEMV that is not restated in C)."""
        qn = np.asarray(qn)
        pad = (-qn.shape[-1]) % (4 * self.dpb)
        qp = np.concatenate([qn, np.zeros(qn.shape[:-1] + (pad,), qn.dtype)], axis=-1)
        if self.R is not None:
            qp = qp @ self.R.T
        else:
            assert qp.shape[-1] == self.dq
        return np.ascontiguousarray(qp)

    def query(self, qn, k, n_probes=1, pass_1=None, debug=False):
        """qn: float32 query AFTER the metric's normalisation (ivf.py:125-127)."""
        qn = np.ascontiguousarray(qn, dtype=np.float32)
        qp = self.pq_query(qn)
        R = pass_1 if pass_1 else (n_probes + 1) * k + 1
        out = np.full(max(R, k), -1, dtype=np.int64)
        probes = np.full(max(2 * n_probes + 10, 1), -1, dtype=np.int64)
        hidx = np.zeros(R, dtype=np.int64)
        hval = np.zeros(R, dtype=np.int32)
        table = np.zeros((self.M, 16), dtype=np.uint8)
        n = lib().tko_ivf_query(C.byref(self._s), qn.ctypes.data, qp.ctypes.data, k,
                                n_probes, int(pass_1 or 0), out.ctypes.data,
                                probes.ctypes.data, hidx.ctypes.data, hval.ctypes.data,
                                table.ctypes.data)
        if debug:
            kc = min(n_probes, self.n_lists)
            return out[:n], dict(probes=probes[:kc], heap_idx=hidx, heap_val=hval, table=table)
        return out[:n]

    def query_batch(self, qn, k, n_probes=1, pass_1=None):
        qn = np.ascontiguousarray(qn, dtype=np.float32)
        qp = self.pq_query(qn)
        out = np.full((len(qn), k), -1, dtype=np.int64)
        lib().tko_ivf_query_batch(C.byref(self._s), qn.ctypes.data, qp.ctypes.data,
                                  C.c_int64(len(qn)), k, n_probes, int(pass_1 or 0),
                                  out.ctypes.data)
        return out


# ---- offline build path -----------------------------------

def encode_pq(centers, dpb, data_pq):
    """labels (n, M) uint8 of FastPQ.transform's per-block knn_brute(col, code, 1)
    (fast_pq.py:174-181); data_pq: padded (rotated) rows, float32 or float64."""
    centers = np.ascontiguousarray(centers, dtype=np.float32)
    is64 = data_pq.dtype != np.float32
    data_pq = np.ascontiguousarray(data_pq, dtype=np.float64 if is64 else np.float32)
    n, dq = data_pq.shape
    assert centers.shape == (16, dq)
    labels = np.empty((n, dq // dpb), dtype=np.uint8)
    lib().tko_encode_pq(_p(centers, C.c_float), dq, int(dpb), C.c_void_p(data_pq.ctypes.data),
                        int(is64), C.c_int64(n), _p(labels, C.c_uint8))
    return labels


def fastpq_transform(centers, dpb, R, data):
    """FastPQ.transform (fast_pq.py:147-184) -> (true_n, packed)"""
    true_n = data.shape[0]
    pr, pc = (-data.shape[0]) % 16, (-data.shape[1]) % (4 * dpb)
    padded = np.zeros((data.shape[0] + pr, data.shape[1] + pc), dtype=data.dtype)
    padded[:data.shape[0], :data.shape[1]] = data
    if R is not None:
        padded = padded @ R.T          # BLAS GEMM: not restated, numpy as in the reference
    return true_n, transform_data(encode_pq(centers, dpb, padded))


def assign(X, Y, k, metric):
    """knn_brute(X, Y, k, metric) (utils.py:66-86) for k <= 16 (k <= 2: numpy's dumb_select;
    k >= 3: ascending by (value, position) — numpy's own order there is its SIMD quickselect's,
    ascending on the fixture host, not pinned by the reference); a trailing 1-row chunk
    (len(X) % 100 == 1) is a GEMV in numpy and stays numpy."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    y64 = Y.dtype != np.float32
    Y = np.ascontiguousarray(Y, dtype=np.float64 if y64 else np.float32)
    out = np.empty((len(X), k), dtype=np.int64)
    rc = lib().tko_assign(_p(X, C.c_float), C.c_int64(len(X)), X.shape[1],
                          C.c_void_p(Y.ctypes.data), int(y64), C.c_int64(len(Y)), int(k),
                          int(metric == "angular"), _p(out, C.c_int64))
    assert rc == 0
    return out
