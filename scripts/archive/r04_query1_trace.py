"""Kernel statistics of the one-query drop-in call IVF.query (run under rocprofv3 --kernel-trace --stats)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = argparse.ArgumentParser().parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 1, "angular", "glove-like", 100000
a.cache_dir, a.data_file = os.environ.get("TMPDIR", "/tmp"), None
ivf, cent = bench.build_index(a, "cuda:0")
qs = bench.synth_queries(cent, 600, 110, kind="glove-like")
for q in qs:
    ivf.query(q.copy(), 10, 10)
