"""ivf.query(q) per call over the reference bench's n_probes sweep (examples/bench.py:118-137: n_probes += int(sqrt(n_probes))
until recall 0.9), register heap (2 / 4 / 8 nodes per lane) against the lane / packed kernels (TK_OPT_PAIR_NQ = 0), and the
CPU oracle per query; GloVe-shaped index, build_probes 1 and 2.  usage: python scripts/r06_query1_probes.py"""
import argparse, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tinyknn_amd import _lib
for bp in (1, 2):
    a = argparse.ArgumentParser().parse_args([])
    a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, bp, "angular", "glove-like", 100000
    a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
    ivf, cent = bench.build_index(a, torch.device("cuda:0"))
    dev = ivf.device_index()
    qs = bench.synth_queries(cent, 400, 12345, kind=a.data)
    qn, _ = ivf._prepare(qs.copy())
    ox = bench.oracle_index(ivf)
    for n_probes in (1, 2, 4, 8, 10, 13, 16, 20, 24, 28, 38, 50):
        row = {"build_probes": bp, "n_probes": n_probes, "R": (n_probes + 1) * 10 + 1}
        t0 = time.perf_counter()
        want = [ox.query(np.ascontiguousarray(qn[i]), 10, n_probes) for i in range(200)]
        row["oracle_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        for tag, pair_nq in (("lanes_ms", 0), ("register_heap_ms", 8192)):
            dev.set_option(_lib.OPT_PAIR_NQ, pair_nq)
            for i in range(10):
                ivf.query(qs[i].copy(), k=10, n_probes=n_probes)
            t0 = time.perf_counter()
            got = [ivf.query(qs[i].copy(), k=10, n_probes=n_probes) for i in range(200)]
            row[tag] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
            row["identical_" + tag[:-3]] = sum(int(len(g) == len(w) and (np.asarray(g) == np.asarray(w)).all()) for g, w in zip(got, want))
        print(json.dumps(row), flush=True)
    dev.close() if hasattr(dev, "close") else None
