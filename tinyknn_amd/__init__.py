"""tinyknn_amd — MI355X-native drop-in for the hot path of thomasahle/tinyknn.

Same package surface as the reference (tinyknn/__init__.py:1-6); the Quick-ADC
scan, the bounded top-R heap, the distance tables and the exact rescoring run as
hand-written HIP kernels for gfx950 behind a C ABI (include/tinyknn_hip.h).
"""
from . import _transform
from . import _fast_pq
from .fast_pq import FastPQ, avx
from .ivf import IVF
from . import utils
from .utils import bottom_k, bottom_k_2d, cdist, knn_brute, group_data_by_indices

__all__ = ["FastPQ", "IVF", "avx", "utils", "bottom_k", "bottom_k_2d", "cdist", "knn_brute",
           "group_data_by_indices", "_transform", "_fast_pq"]
