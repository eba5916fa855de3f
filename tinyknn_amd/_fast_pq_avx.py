"""Drop-in for the reference's Cython module tinyknn._fast_pq_avx (AVX order:
two saturating accumulators merged at the end; _fast_pq_256.pyx:52-156), the
variant the reference's public API uses (fast_pq.py:21-24)."""
from ._fast_pq import _estimate, _query, init_heap, insert, insert_is  # noqa: F401
from ._lib import ORDER_AVX


def estimate_pq_avx(data, tables, out, signd):
    _estimate(data, tables, out, signd, ORDER_AVX)


def query_pq_avx(data, n, tables, indices, vals, signd, labels=None):
    _query(data, n, tables, indices, vals, signd, labels, ORDER_AVX)
