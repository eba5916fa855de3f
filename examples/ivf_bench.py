#!/usr/bin/env python3
"""Recall-vs-QPS sweep of the GPU IVF — the measurement protocol of the
reference's examples/bench.py:108-137 (n_probes grows by int(sqrt(n_probes)) until
Recall10@10 >= 0.9), on a .npy file or on the synthetic GloVe-100-shaped set of
bench.py.  Prints the reference's `Recall10@10:` / `Queries/second:` lines (its
plot scripts parse those), measured over whole batches on the device.

    python examples/ivf_bench.py --n 200000 --metric angular
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyknn_amd import FastPQ, IVF, knn_brute              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("filename", nargs="?", default=None, help=".npy data (else synthetic)")
ap.add_argument("--n", type=int, default=200000)
ap.add_argument("--d", type=int, default=100)
ap.add_argument("--n-queries", type=int, default=10000)
ap.add_argument("--k-neighbours", type=int, default=10)
ap.add_argument("--metric", choices=["euclidean", "angular"], default="angular")
ap.add_argument("--a", type=float, default=1.0, help="n_clusters = int(a * sqrt(n))")
ap.add_argument("--build-probes", type=int, nargs="+", default=[1])
ap.add_argument("--per-query", type=int, default=0,
                help="also time this many queries ONE ivf.query(q) per call, exactly as the reference's loop does "
                     "(examples/bench.py:118-137); printed on its own line")
args = ap.parse_args()

np.random.seed(10)
if args.filename:
    data = np.load(args.filename).astype(np.float32)
    np.random.shuffle(data)
else:
    cent = np.random.randn(300, args.d)
    tot = args.n + args.n_queries
    data = (cent[np.random.randint(300, size=tot)] + 0.7 * np.random.randn(tot, args.d)).astype(np.float32)
data, queries = data[:-args.n_queries], data[-args.n_queries:]
n, d = data.shape
k = args.k_neighbours
n_clusters = int(args.a * n ** 0.5)
print(f"num_points={n}, num_dims={d}, num_queries={len(queries)}, dims_per_block=2, num_clusters={n_clusters}")
truth = knn_brute(queries[:1000], data, k, metric=args.metric)       # recall on 1000 queries
ivf = IVF(args.metric, n_clusters, FastPQ(2))
ivf.fit(data[np.random.choice(n, min(n, 100000), replace=False)])
for build_probes in args.build_probes:
    print(f"Adding each point to {build_probes} lists...")
    ivf.build(data, n_probes=build_probes)
    ivf.query_batch(queries[:16], k, 1)                               # upload + warm-up
    recall, n_probes = 0.0, 1
    while recall < 0.9 and n_probes <= n_clusters:
        # the first call at a new n_probes re-sizes the index's workspaces (and, the very first,
        # creates the streaming session with its page-locked staging): timed apart, like the
        # reference's bench keeps its build out of the query loop
        t0 = time.time()
        found = ivf.query_batch(queries, k, n_probes=n_probes)
        first = time.time() - t0
        reps = 5
        t0 = time.time()
        for _ in range(reps):
            found = ivf.query_batch(queries, k, n_probes=n_probes)
        qps = reps * len(queries) / (time.time() - t0)
        print(f"(n_probes={n_probes}; first call incl. workspace set-up: {first * 1e3:.1f} ms)")
        recall = float(np.mean([len(set(t) & set(g)) / k for t, g in zip(truth, found[:1000])]))
        print(f"Recall{k}@{k}:", recall)
        print("Queries/second:", qps)
        if args.per_query:
            m = min(args.per_query, len(queries))
            for q in queries[:8]:
                ivf.query(q.copy(), k=k, n_probes=n_probes)
            t0 = time.time()
            for q in queries[:m]:
                ivf.query(q.copy(), k=k, n_probes=n_probes)
            print("(one ivf.query(q) per call) queries/second:", m / (time.time() - t0))
        n_probes += int(n_probes ** 0.5)
