"""Drop-in for the reference's Cython module tinyknn._fast_pq (SSE order),
bound to the gfx950 kernels through the C ABI (include/tinyknn_hip.h).

    estimate_pq_sse, query_pq_sse   reference: _fast_pq.pyx:101-206
    init_heap, insert, insert_is    reference: _fast_pq.pyx:240-307
"""
import numpy as np

from . import _lib
from ._lib import ORDER_AVX, ORDER_SSE


def _buf(a, dtype, ndim, name):
    # what Cython's typed-memoryview acquisition enforces before entry
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch for {name}: expected {np.dtype(dtype)}")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions for {name}")
    if not a.flags.c_contiguous:
        raise ValueError(f"ndarray is not C-contiguous: {name}")
    return a


def _estimate(data, tables, out, signd, order):
    data = _buf(data, np.uint64, 2, "data")
    tables = _buf(tables, np.uint64, 1, "tables")
    out = _buf(out, np.uint64, 1, "out")
    chunks, M = data.shape
    assert tables.shape[0] >= 2 * M and out.shape[0] >= 2 * chunks
    if not out.flags.writeable:
        raise ValueError("buffer source array is read-only")
    _lib.check(_lib.lib().tk_estimate_pq(
        _lib.ptr(data, _lib._u64p), chunks, M, _lib.ptr(tables, _lib._u64p),
        _lib.ptr(out, _lib._u64p), int(bool(signd)), order))


def _query(data, n, tables, indices, vals, signd, labels, order):
    data = _buf(data, np.uint64, 2, "data")
    tables = _buf(tables, np.uint64, 1, "tables")
    indices = _buf(indices, np.int64, 1, "indices")
    vals = _buf(vals, np.int32, 1, "vals")
    chunks, M = data.shape
    assert tables.shape[0] >= 2 * M and len(indices) == len(vals)
    lab = None
    if labels is not None:
        labels = _buf(labels, np.int64, 1, "labels")
        lab = _lib.ptr(labels, _lib._i64p)
    _lib.check(_lib.lib().tk_query_pq(
        _lib.ptr(data, _lib._u64p), chunks, M, int(n), _lib.ptr(tables, _lib._u64p),
        _lib.ptr(indices, _lib._i64p), _lib.ptr(vals, _lib._i32p), len(indices),
        int(bool(signd)), lab, order))


def estimate_pq_sse(data, tables, out, signd):
    _estimate(data, tables, out, signd, ORDER_SSE)


def query_pq_sse(data, n, tables, indices, vals, signd, labels=None):
    _query(data, n, tables, indices, vals, signd, labels, ORDER_SSE)


def init_heap(indices, vals, signd):
    indices = _buf(indices, np.int64, 1, "indices")
    vals = _buf(vals, np.int32, 1, "vals")
    _lib.check(_lib.lib().tk_init_heap(_lib.ptr(indices, _lib._i64p), _lib.ptr(vals, _lib._i32p),
                                       len(indices), int(bool(signd))))


def insert(indices, vals, i, v):
    indices = _buf(indices, np.int64, 1, "indices")
    vals = _buf(vals, np.int32, 1, "vals")
    _lib.check(_lib.lib().tk_heap_insert(_lib.ptr(indices, _lib._i64p), _lib.ptr(vals, _lib._i32p),
                                         len(indices), int(i), int(v)))


def insert_is(indices, vals, i, v):
    indices = _buf(indices, np.int64, 1, "indices")
    vals = _buf(vals, np.int32, 1, "vals")
    _lib.check(_lib.lib().tk_heap_insert_is(_lib.ptr(indices, _lib._i64p),
                                            _lib.ptr(vals, _lib._i32p), len(indices), int(i), int(v)))
