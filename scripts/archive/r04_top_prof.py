"""Where one per-query `_FastDistanceTable.top` call spends its time (host pieces by cProfile)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tinyknn_amd import FastPQ, _fast_pq
n, d, nq, k = 1_000_000, 128, 200, 10
X, cent_ = bench.synth(n, 0, d, 10, kind="clustered")
qs = bench.synth_queries(cent_, nq, 110, kind="clustered")
pq = FastPQ(2); pq.fit(X[:30000])
td = pq.transform(X, device=True)
_fast_pq.cache_device_codes = True
pq.distance_table(qs[0]).top(td, X, k=k)
t0 = time.perf_counter(); dts = [pq.distance_table(q) for q in qs]; t1 = time.perf_counter()
print("distance_table ms", (t1 - t0) / nq * 1e3)
t0 = time.perf_counter(); [dt.top(td, X, k=k) for dt in dts]; t1 = time.perf_counter()
print("top ms", (t1 - t0) / nq * 1e3)
pr = cProfile.Profile(); pr.enable()
for q in qs: pq.distance_table(q).top(td, X, k=k)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
