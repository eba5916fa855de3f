"""One query per call on the GloVe-shaped index (the reference's own protocol, examples/bench.py:118-137): where the
0.5 ms of IVF.query go — stage times of a one-query batch, and the call's wall time.  usage: python scripts/r05_query1.py"""
import argparse, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = argparse.ArgumentParser().parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 1, "angular", "glove-like", 100000
a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
device = torch.device("cuda:0")
ivf, cent = bench.build_index(a, device)
dev = ivf.device_index()
qs = bench.synth_queries(cent, 2000, 12345, kind=a.data)
qn, qp = ivf._prepare(qs.copy())
qp = np.ascontiguousarray(qp)
for nq in (1, 4, 16):
    for _ in range(50):
        dev.query_batch(qn[:nq], qp[:nq], 10, 10)
    dev.set_profiling(1)
    t0 = time.perf_counter()
    n = 300
    for i in range(n):
        dev.query_batch(qn[i * nq:(i + 1) * nq], qp[i * nq:(i + 1) * nq], 10, 10)
    wall = (time.perf_counter() - t0) / n
    prof = dev.last_profile()
    dev.set_profiling(0)
    print(json.dumps({"queries_per_call": nq, "wall_ms_per_call": round(wall * 1e3, 4), "stage_ms": {k: round(v, 4) for k, v in prof[0].items()},
                      "stage_sum_ms": round(sum(prof[0].values()), 4)}), flush=True)
# the drop-in call itself
for _ in range(20):
    ivf.query(qs[0], k=10, n_probes=10)
t0 = time.perf_counter()
for i in range(300):
    ivf.query(qs[i], k=10, n_probes=10)
print(json.dumps({"ivf.query ms per call": round((time.perf_counter() - t0) / 300 * 1e3, 4)}), flush=True)
