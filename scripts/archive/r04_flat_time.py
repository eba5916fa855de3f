import sys, time, json
import numpy as np
sys.path.insert(0, '/root/repo')
import bench
from tinyknn_amd import FastPQ
from tinyknn_amd.fast_pq import FlatTop
kind = sys.argv[1] if len(sys.argv) > 1 else "sift-clustered"
n, d, k = 1_000_000, 128, 10
X, cent = bench.synth(n, 0, d, 10, kind=kind)
pq = FastPQ(2); pq.fit(X[:30000])
td = pq.transform(X, device=True)
ft = FlatTop(pq, td, X)
for nqb in (2048, 10000, 10000, 10000, 4000):
    qb = bench.synth_queries(cent, nqb, 111 + nqb, kind=kind)
    t0 = time.perf_counter(); g = ft.top(qb, k); t = time.perf_counter() - t0
    print(kind, nqb, "queries:", round(t * 1e3, 1), "ms =", round(nqb / t), "queries/s", flush=True)
