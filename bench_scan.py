#!/usr/bin/env python3
"""bench_scan.py — kernel-only microbenchmark of the Quick-ADC scan at sizes that
do NOT fit the 256 MB Infinity Cache (SURVEY §8d "kernel-only microbench"): random
packed codes (N/16, M), realistic tables in [-4, 23], the flat scan of every code
for nq queries.  Prints one JSON line per case: algorithmic GB/s = nq * N * M/2
bytes / kernel time (HIP events), next to the code bytes actually resident.

    python bench_scan.py --log2n 26 27 --M 32 52 --nq 1 4 16
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, nargs="+", default=[24, 26])
    ap.add_argument("--M", type=int, nargs="+", default=[32, 52])
    ap.add_argument("--nq", type=int, nargs="+", default=[1, 4, 16])
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from tinyknn_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(0)
    for M in args.M:
        for lg in args.log2n:
            n = 1 << lg
            chunks = n // 16
            packed = rng.integers(0, 2**63, size=(chunks, M), dtype=np.int64).astype(np.uint64)
            h = L.tk_codes_upload(_lib.ptr(packed, _lib._u64p), chunks, M)
            assert h
            for nq in args.nq:
                tables = rng.integers(-4, 24, size=(nq, M, 16)).astype(np.int8).view(np.uint8)
                t_dev = torch.from_numpy(tables).to(dev)
                out = torch.empty((nq, chunks * 16), dtype=torch.uint8, device=dev)
                run = lambda: _lib.check(L.tk_codes_estimate_dev(h, t_dev.data_ptr(), nq, out.data_ptr(), 1, 1, stream))
                run(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                alg = nq * n * (M // 2) + nq * n  # code bytes per (query, code) + int8 out
                print(json.dumps({"kernel": "scan_units" if nq >= 4 else "scan_flat", "N": n, "M": M,
                                  "nq": nq, "code_bytes": n * M // 2, "ms": ms,
                                  "algorithmic_GBps": alg / ms / 1e6,
                                  "frac_of_8TBps": alg / ms / 1e6 / 8000.0}), flush=True)
            L.tk_codes_free(h)
            del packed


if __name__ == "__main__":
    main()
