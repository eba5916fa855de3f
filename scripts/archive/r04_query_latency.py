"""Latency of the per-query drop-in call `IVF.query(q, k, n_probes)` (ivf.py:106-163), one query per call as
examples/bench.py:118-137 times the reference, on the headline index; with cProfile of the host side."""
import argparse, cProfile, pstats, sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
ap = argparse.ArgumentParser()
a = ap.parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 1, "angular", "glove-like", 100000
a.cache_dir, a.data_file = os.environ.get("TMPDIR", "/tmp"), None
ivf, cent = bench.build_index(a, "cuda:0")
qs = bench.synth_queries(cent, 2000, 110, kind="glove-like")
for q in qs[:50]:
    ivf.query(q.copy(), 10, 10)
t0 = time.perf_counter()
for q in qs[:1000]:
    ivf.query(q.copy(), 10, 10)
t = (time.perf_counter() - t0) / 1000
print(json.dumps({"per_query_ms": t * 1e3, "queries_per_s": 1 / t}))
pr = cProfile.Profile(); pr.enable()
for q in qs[1000:1400]:
    ivf.query(q.copy(), 10, 10)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
# the same index, batches of growing size through the drop-in batch call (raw queries in, ids out, exact)
out = []
for nq in (1, 8, 64, 512, 4096, 10000):
    Q = bench.synth_queries(cent, nq, 200 + nq, kind="glove-like")
    ivf.query_batch(Q.copy(), 10, 10)
    reps = max(3, min(200, 20000 // nq))
    t0 = time.perf_counter()
    for _ in range(reps):
        ivf.query_batch(Q.copy(), 10, 10)
    t = (time.perf_counter() - t0) / reps
    out.append({"nq": nq, "ms_per_call": t * 1e3, "queries_per_s": nq / t})
print(json.dumps({"query_batch_by_size": out}))
# one query per call with the other replay kernels (tk_index_set_heap_mode: 0 lane-per-query, 2 packed wave-per-query, 1 general)
res = {}
dev = ivf.device_index()
for mode in (0, 2, 1, 0):
    dev.set_heap_mode(mode)
    for q in qs[:30]:
        ivf.query(q.copy(), 10, 10)
    t0 = time.perf_counter()
    for q in qs[:600]:
        ivf.query(q.copy(), 10, 10)
    res.setdefault(str(mode), []).append((time.perf_counter() - t0) / 600 * 1e3)
dev.set_heap_mode(0)
print(json.dumps({"per_query_ms_by_heap_mode": res}))
