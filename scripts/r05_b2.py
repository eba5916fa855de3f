"""IVF.build(n_probes=2) on the GloVe-shaped set (the bench's sweep point, the reference's default build, ivf.py:53):
rate of the pipelined mode with the TWIN form of the lane replay against the hash-set form it replaces, with and
without pairs of calls and the plain path; insert rounds of the replay; 500 rows against the oracle.
usage: python scripts/r05_b2.py [build_probes]"""
import argparse, sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tinyknn_amd import _lib

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
only = len(sys.argv) > 2 and sys.argv[2] == "only"      # the default configuration alone (profiler runs)
a = argparse.ArgumentParser().parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, bp, "angular", "glove-like", 100000
a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
device = torch.device("cuda:0")
ivf, cent = bench.build_index(a, device)
dev = ivf.device_index()
tl, to = dev.twin_table()
print(json.dumps({"twin_table": list(tl.shape), "rows_with_a_copy": int((tl[:, 0] >= 0).sum()) if tl.size else 0}), flush=True)
bs = []
for b in range(2):
    qs_b = bench.synth_queries(cent, a.nq, a.seed + 100 + 1000 * b, kind=a.data)
    qn_b, qp_b = ivf._prepare(qs_b.copy())
    bs.append(dict(qn=qn_b, q_dev=torch.from_numpy(qn_b).to(device), qp_dev=torch.from_numpy(np.ascontiguousarray(qp_b)).to(device),
                   out=torch.full((a.nq, a.k), -1, dtype=torch.int64, device=device)))
st = torch.cuda.current_stream().cuda_stream
ox = bench.oracle_index(ivf)
want = ox.query_batch(bs[0]["qn"][:500], a.k, 10)
lazy_arg = int(sys.argv[3]) if len(sys.argv) > 3 else -1
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 2           # replay streams of the pipelined mode
for twin, co, plain, lazy in (((1, 2, True, lazy_arg),) if only else
                              ((0, 1, True, -1), (1, 1, False, -1), (1, 1, True, -1), (1, 2, False, -1), (1, 2, True, -1),
                               (1, 2, True, 0), (1, 2, True, 1), (1, 2, "always", -1))):
    dev.set_option(_lib.OPT_REPLAY_TWIN, twin)
    dev.set_option(_lib.OPT_REPLAY_LAZY, lazy)
    dev.set_plain_scan(plain)
    r = bench.timed_rate(dev, bs, False, a.nq, a.k, 10, st, depth, co)
    torch.cuda.synchronize()
    got = bs[0]["out"].cpu().numpy()[:500]
    pst = dev.plain_stats() or {}
    print(json.dumps({"twin": twin, "coalesce": co, "plain": str(plain), "lazy": lazy, "depth": depth, "M_qps": round(r["queries_per_s"] / 1e6, 2),
                      "ms": round(r["ms_per_step"], 4), "identical_rows": int((want == got).all(axis=1).sum()),
                      "plain_state": pst.get("state"), "flagged": pst.get("flagged_queries")}), flush=True)
if only:
    sys.exit(0)
# stage times alone and the replay's rounds (one batch in flight)
dev.set_pipeline(1)
dev.set_plain_scan(True)
for twin, lazy in ((1, 0), (1, 1), (0, 0)):
    dev.set_option(_lib.OPT_REPLAY_TWIN, twin)
    dev.set_option(_lib.OPT_REPLAY_LAZY, lazy)
    dev.set_option(_lib.OPT_REPLAY_COUNT, 1)
    dev.set_profiling(1)
    for _ in range(6):
        dev.query_batch_dev(bs[0]["q_dev"].data_ptr(), bs[0]["qp_dev"].data_ptr(), False, a.nq, a.k, 10, bs[0]["out"].data_ptr(), stream=st)
    torch.cuda.synchronize()
    prof = dev.last_profile()
    print(json.dumps({"twin": twin, "lazy": lazy, "alone_ms": {k_: round(v_, 4) for k_, v_ in prof[0].items()}, "replay": dev.replay_stats()}), flush=True)
    dev.set_profiling(0)
    dev.set_option(_lib.OPT_REPLAY_COUNT, 0)
