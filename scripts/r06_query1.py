"""One query per call on the GloVe-shaped index (the reference's own protocol, examples/bench.py:118-137): stage times
of small batches and the call's wall time, with the wave-per-query register heap (TK_OPT_PAIR_NQ) and without, and the
oracle (CPU, one core) per query beside it.  usage: python scripts/r06_query1.py"""
import argparse, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tinyknn_amd import _lib
a = argparse.ArgumentParser().parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 1, "angular", "glove-like", 100000
a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
device = torch.device("cuda:0")
ivf, cent = bench.build_index(a, device)
dev = ivf.device_index()
qs = bench.synth_queries(cent, 8000, 12345, kind=a.data)
qn, qp = ivf._prepare(qs.copy())
qp = np.ascontiguousarray(qp)
for pair_nq in (0, 1 << 20):
    dev.set_option(_lib.OPT_PAIR_NQ, pair_nq)
    for nq in (1, 4, 16, 64, 256, 1024, 2048, 4096, 8000):
        n = 300 if nq <= 16 else 40 if nq <= 256 else 6 if nq <= 2048 else 2
        for _ in range(10):
            dev.query_batch(qn[:nq], qp[:nq], 10, 10)
        dev.set_profiling(1)
        t0 = time.perf_counter()
        for i in range(n):
            lo = (i * nq) % (len(qn) - nq + 1)
            dev.query_batch(qn[lo:lo + nq], qp[lo:lo + nq], 10, 10)
        wall = (time.perf_counter() - t0) / n
        prof = dev.last_profile()
        dev.set_profiling(0)
        print(json.dumps({"pair_nq": pair_nq, "queries_per_call": nq, "wall_ms_per_call": round(wall * 1e3, 4),
                          "stage_ms": {k: round(v, 4) for k, v in prof[0].items()},
                          "stage_sum_ms": round(sum(prof[0].values()), 4)}), flush=True)
dev.set_option(_lib.OPT_PAIR_NQ, 256)
# the drop-in call itself
for _ in range(20):
    ivf.query(qs[0].copy(), k=10, n_probes=10)
got = []
t0 = time.perf_counter()
for i in range(2000):
    got.append(ivf.query(qs[i].copy(), k=10, n_probes=10))
t_gpu = (time.perf_counter() - t0) / 2000
ox = bench.oracle_index(ivf)
t0 = time.perf_counter()
want = [ox.query(np.ascontiguousarray(qn[i]), 10, 10) for i in range(2000)]
t_cpu = (time.perf_counter() - t0) / 2000
same = sum(int(len(g) == len(w) and (np.asarray(g) == np.asarray(w)).all()) for g, w in zip(got, want))
print(json.dumps({"ivf.query ms per call": round(t_gpu * 1e3, 4), "queries_per_s": round(1 / t_gpu),
                  "oracle ms per query (one core, prepared rows)": round(t_cpu * 1e3, 4),
                  "identical_rows": same, "rows": 2000}), flush=True)
