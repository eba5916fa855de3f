"""Drop-in for the reference's Cython module tinyknn._fast_pq (SSE order),
bound to the gfx950 kernels through the C ABI (include/tinyknn_hip.h).

    estimate_pq_sse, query_pq_sse   reference: _fast_pq.pyx:101-206
    init_heap, insert, insert_is    reference: _fast_pq.pyx:240-307
"""
import numpy as np

from . import _lib
from ._lib import ORDER_AVX, ORDER_SSE  # noqa: F401  (re-exported for _fast_pq_avx)


def _buf(a, dtype, ndim, name):
    # what Cython's typed-memoryview acquisition enforces before entry
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch for {name}: expected {np.dtype(dtype)}")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions for {name}")
    if not a.flags.c_contiguous:
        raise ValueError(f"ndarray is not C-contiguous: {name}")
    return a


# Opt-in: code arrays that are scanned repeatedly (FastPQ.transform output: one array,
# thousands of queries, examples/example.py:60-66) can be kept in HBM between calls, keyed
# by the identity of the numpy array and released when it is garbage collected.  OFF by
# default: the reference kernels always read the live buffer, and a resident copy cannot see
# an in-place mutation of the numpy array (set cache_device_codes = True only for arrays that
# are not written to afterwards; forget_device_codes drops a copy).
cache_device_codes = False
_CACHE_MIN_BYTES = 16 * 1024
_CACHE_MAX_ARRAYS = 512
_codes_cache = {}      # id(array) -> (weakref, handle, (data pointer, shape))


def _drop(key):
    ent = _codes_cache.pop(key, None)
    if ent is not None and _lib.owns_handles():
        _lib.lib().tk_codes_free(ent[1])


def forget_device_codes(data=None):
    """Drop the HBM copy of one packed array (or, with no argument, of all of them)."""
    for k in (list(_codes_cache) if data is None else [id(data)]):
        _drop(k)


def device_codes(data):
    """HBM-resident handle (tk_codes*) of a packed code array, or None."""
    if not cache_device_codes or data.nbytes < _CACHE_MIN_BYTES:
        return None
    key = id(data)
    sig = (data.ctypes.data, data.shape)
    ent = _codes_cache.get(key)
    if ent is not None and ent[0]() is data and ent[2] == sig:
        return ent[1]
    _drop(key)                      # a stale entry of a dead array with the same id
    h = _lib.lib().tk_codes_upload(_lib.ptr(data, _lib._u64p), data.shape[0], data.shape[1])
    if not h:
        _lib.check(-2)
    import weakref

    def _gone(_ref, key=key, h=h):
        ent = _codes_cache.get(key)
        if ent is not None and ent[1] == h:
            try:
                _drop(key)
            except Exception:
                pass

    _codes_cache[key] = (weakref.ref(data, _gone), h, sig)
    while len(_codes_cache) > _CACHE_MAX_ARRAYS:     # evict the oldest entry BY KEY
        oldest = next(iter(_codes_cache))
        if oldest == key:
            break
        _drop(oldest)
    return h


def _estimate(data, tables, out, signd, order):
    data = _buf(data, np.uint64, 2, "data")
    tables = _buf(tables, np.uint64, 1, "tables")
    out = _buf(out, np.uint64, 1, "out")
    chunks, M = data.shape
    assert tables.shape[0] >= 2 * M and out.shape[0] >= 2 * chunks
    if not out.flags.writeable:
        raise ValueError("buffer source array is read-only")
    h = device_codes(data)
    if h:
        _lib.check(_lib.lib().tk_codes_estimate(h, _lib.ptr(tables, _lib._u64p), 1,
                                                _lib.ptr(out, _lib._u64p), int(bool(signd)), order))
        return
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
    h = device_codes(data)
    if h:
        _lib.check(_lib.lib().tk_codes_query(
            h, int(n), _lib.ptr(tables, _lib._u64p), _lib.ptr(indices, _lib._i64p),
            _lib.ptr(vals, _lib._i32p), len(indices), int(bool(signd)), lab, order))
        return
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
