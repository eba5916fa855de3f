"""Phase timers of heap_replay_lanes_kernel on the bench batch (tk_debug_replay_timers).
usage: python scripts/replay_timers.py [--build-probes 1|2] [--n-probes 10]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-probes", type=int, default=1)
    ap.add_argument("--n-probes", type=int, default=10)
    ap.add_argument("--nq", type=int, default=10000)
    a = ap.parse_args()
    import torch
    import bench
    from tinyknn_amd import _lib
    args = argparse.Namespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=a.build_probes,
                              metric="angular", data="glove-like", fit_sample=100000,
                              cache_dir=os.environ.get("TMPDIR", "/tmp"))
    torch.cuda.set_device(0)
    ivf, cent = bench.build_index(args, torch.device("cuda", 0))
    dev = ivf.device_index()
    dev.set_pipeline(1)
    qs = bench.synth_queries(cent, a.nq, 110, kind="glove-like")
    qn, qp = ivf._prepare(qs.copy())
    q_dev = torch.from_numpy(qn).cuda()
    qp_dev = torch.from_numpy(np.ascontiguousarray(qp)).cuda()
    out = torch.full((a.nq, 10), -1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), False, a.nq, 10, a.n_probes,
                                      out.data_ptr(), stream=st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    dev.set_profiling(1)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    stages, _, _ = dev.last_profile()
    dev.set_profiling(0)
    L = _lib.lib()
    L.tk_debug_replay_timers.restype = C.c_int
    L.tk_debug_replay_timers.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    sums = (C.c_uint64 * 8)()
    wg = C.c_int(0)
    assert L.tk_debug_replay_timers(1, a.nq, sums, C.byref(wg)) == 0
    run()
    torch.cuda.synchronize()
    assert L.tk_debug_replay_timers(0, 0, sums, C.byref(wg)) == 0
    w = max(wg.value, 1)
    names = ["total_cycles", "search_cycles", "insert_cycles", "lds_sift_cycles", "rounds", "lds_iterations",
             "search_iterations", "segments"]
    per = {n: sums[i] / w for i, n in enumerate(names)}
    per["other_cycles"] = per["total_cycles"] - per["search_cycles"] - per["insert_cycles"]
    print(json.dumps({"build_probes": a.build_probes, "n_probes": a.n_probes, "workgroups": wg.value,
                      "heap_stage_ms": stages["heap"], "per_wave": per,
                      "cycles_per_round": {"search": per["search_cycles"] / max(per["rounds"], 1),
                                           "insert": per["insert_cycles"] / max(per["rounds"], 1),
                                           "of_which_lds_sift": per["lds_sift_cycles"] / max(per["rounds"], 1)}}))


if __name__ == "__main__":
    main()
