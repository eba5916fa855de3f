"""Stage times of ONE query per call on the GloVe-shaped build(n_probes=2) index, register heap (label24 / label64) and lane kernel."""
import argparse, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tinyknn_amd import _lib
a = argparse.ArgumentParser().parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 2, "angular", "glove-like", 100000
a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
ivf, cent = bench.build_index(a, torch.device("cuda:0"))
dev = ivf.device_index()
qs = bench.synth_queries(cent, 400, 12345, kind=a.data)
qn, qp = ivf._prepare(qs.copy())
qp = np.ascontiguousarray(qp)
for tag, pair_nq, l24 in (("lanes", 0, 1), ("register heap label24", 8192, 1), ("register heap label64", 8192, 0)):
    dev.set_option(_lib.OPT_PAIR_NQ, pair_nq)
    dev.set_option(_lib.OPT_LABELS24, l24)
    for _ in range(10):
        dev.query_batch(qn[:1], qp[:1], 10, 10)
    dev.set_profiling(1)
    t0 = time.perf_counter()
    for i in range(200):
        dev.query_batch(qn[i:i + 1], qp[i:i + 1], 10, 10)
    wall = (time.perf_counter() - t0) / 200
    prof = dev.last_profile()
    dev.set_profiling(0)
    print(json.dumps({"form": tag, "wall_ms_per_call": round(wall * 1e3, 4), "stage_ms": {k: round(v, 4) for k, v in prof[0].items()}}), flush=True)
