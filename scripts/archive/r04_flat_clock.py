"""Diagnostic build (-DTK_FLAT_CLOCK): phase clocks of flat_top_one_kernel, read from the pinned page."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tinyknn_amd import FastPQ, _fast_pq
n, d, k = 1_000_000, 128, 10
X, cent_ = bench.synth(n, 0, d, 10, kind="clustered")
qs = bench.synth_queries(cent_, 20, 110, kind="clustered")
pq = FastPQ(2); pq.fit(X[:30000])
td = pq.transform(X, device=True)
_fast_pq.cache_device_codes = True
for q in qs:
    pq.distance_table(q).top(td, X, k=k)
