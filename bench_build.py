#!/usr/bin/env python3
"""bench_build.py — the step before the hot path on one MI355X:
nearest-centre assignment (IVF.build, ivf.py:85) and PQ encoding (FastPQ.transform,
fast_pq.py:147-184) of a GloVe-100-shaped data set, device against the reference's numpy
code on the host cores, same outputs.

    python bench_build.py --n 1183514 --d 100 --n-clusters 1087
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1183514)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--n-clusters", type=int, default=1087)
    ap.add_argument("--metric", default="angular")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="rows timed through numpy")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.utils import knn_brute
    X, cent = bench.synth(args.n, 0, args.d, 10)
    ang = args.metric == "angular"
    rng = np.random.RandomState(11)
    sample = X[rng.choice(len(X), 100000, replace=False)]
    if ang:
        sample = sample / np.linalg.norm(sample, axis=1, keepdims=True)
    C = bench.quick_kmeans(sample, args.n_clusters, 8, 10, torch.device("cuda", 0))
    ivf = IVF(args.metric, args.n_clusters, FastPQ(2))
    C = C.astype(np.float32)          # what sklearn's KMeans returns for float32 data
    ivf.all_centers = C / np.linalg.norm(C, axis=1, keepdims=True) if ang else C
    ivf.pq.fit(sample[:30000])
    data = X.copy()
    if ang:
        data /= np.linalg.norm(data, axis=1, keepdims=True)
    M = ivf.pq.centers.shape[1] // 2
    out = {"workload": f"N={args.n} d={args.d} {args.metric}, {args.n_clusters} centres "
                       f"({ivf.all_centers.dtype}), FastPQ dpb=2 M={M}"}

    # ---- assignment
    ivf._nearest_on_device(data[:1000], 1)            # load the library, warm up
    ts = []
    for _ in range(args.reps):
        t = time.perf_counter()
        near = ivf._nearest_on_device(data, 1)
        ts.append(time.perf_counter() - t)
    cs = min(args.cpu_sample, args.n) // 100 * 100
    t = time.perf_counter()
    want = knn_brute(data[:cs], ivf.all_centers, 1, args.metric)
    tcpu = time.perf_counter() - t
    out["assign"] = {"device_rows_per_s_incl_pcie": args.n / min(ts), "device_s": min(ts),
                     "numpy_rows_per_s": cs / tcpu, "numpy_threads": os.cpu_count(),
                     "numpy_sample_rows": cs, "identical_on_sample": bool((near[:cs] == want).all()),
                     "flops_per_row": 2 * args.d * args.n_clusters}
    # ---- encoding (unrotated float32 for d=100; rotation, when fitted, is a host GEMM)
    pad = (-args.d) % 8
    rows = np.concatenate([data, np.zeros((len(data), pad), np.float32)], axis=1) if pad else data
    if ivf.pq.R is not None:
        rows = rows @ ivf.pq.R.T
    ivf.pq.encode_labels(rows[:1600], True)
    ts = []
    for _ in range(args.reps):
        t = time.perf_counter()
        lab = ivf.pq.encode_labels(rows, True)
        ts.append(time.perf_counter() - t)
    cs = min(args.cpu_sample, args.n) // 1600 * 1600
    t = time.perf_counter()
    want = ivf.pq.encode_labels(rows[:cs], False)
    tcpu = time.perf_counter() - t
    # labels may differ from numpy's only where numpy's own distances of the two centroids are
    # EXACTLY equal (its AVX-512 argselect need not return the first of tied entries)
    bad = np.argwhere(lab[:cs] != want)
    ties = 0
    for i, m in bad:
        lo = i - i % 100
        xc = rows[lo:lo + 100, 2 * m:2 * m + 2]
        code = ivf.pq.centers[:, 2 * m:2 * m + 2]
        part = (np.einsum("ij,ij->i", xc, xc)[:, None] + np.einsum("ij,ij->i", code, code)[None]
                - 2 * xc @ code.T)[i - lo]
        ties += int(part[lab[i, m]] == part[want[i, m]] == part.min())
    out["encode"] = {"device_rows_per_s_incl_pcie": args.n / min(ts), "device_s": min(ts),
                     "numpy_rows_per_s": cs / tcpu, "numpy_sample_rows": cs,
                     "identical_on_sample": bool(len(bad) == 0),
                     "labels_differing": int(len(bad)), "of_which_exact_ties_in_numpy": ties,
                     "labels_compared": int(want.size),
                     "bytes_per_row": rows.shape[1] * rows.itemsize + M}
    # ---- whole build
    t = time.perf_counter()
    ivf.build(X, n_probes=1, device=True)
    out["build_device_s"] = time.perf_counter() - t
    print(json.dumps(out))


if __name__ == "__main__":
    main()
