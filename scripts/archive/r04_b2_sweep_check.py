"""Why does the sweep's build(n_probes=2) point read lower than the stand-alone run?  timed_rate on a b2 index with
different window lengths / warm-ups."""
import argparse, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = argparse.ArgumentParser().parse_args([])
a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 2, "angular", "glove-like", 100000
a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
device = torch.device("cuda:0")
ivf, cent = bench.build_index(a, device)
dev = ivf.device_index()
for nb in (2, 4):
    bs = []
    for b in range(nb):
        qs_b = bench.synth_queries(cent, a.nq, a.seed + 100 + 1000 * b, kind=a.data)
        qn_b, qp_b = ivf._prepare(qs_b.copy())
        bs.append(dict(qn=qn_b, q_dev=torch.from_numpy(qn_b).to(device), qp_dev=torch.from_numpy(np.ascontiguousarray(qp_b)).to(device),
                       out=torch.full((a.nq, a.k), -1, dtype=torch.int64, device=device)))
    for steps in (40, 200, 40):
        for co in (2, 1):
            r = bench.timed_rate(dev, bs, False, a.nq, a.k, 10, torch.cuda.current_stream().cuda_stream, 2, co, steps=steps)
            print(json.dumps({"batches": nb, "steps": steps, "coalesce_arg": co, "M_qps": round(r["queries_per_s"] / 1e6, 2), "ms": round(r["ms_per_step"], 4)}), flush=True)
