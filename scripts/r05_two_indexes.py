"""Does an index run slower behind another index of the same process?  (the bench's sweep builds the build(n_probes=2)
index behind the headline index and reads 13.1 M queries/s where the stand-alone run reads 15.2 M)
usage: python scripts/r05_two_indexes.py [close_first]"""
import argparse, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

def mk(bp):
    a = argparse.ArgumentParser().parse_args([])
    a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, bp, "angular", "glove-like", 100000
    a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
    return a

device = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
close_first = len(sys.argv) > 1 and sys.argv[1] == "close_first"

def batches(ivf, cent, a):
    bs = []
    for b in range(2):
        qs_b = bench.synth_queries(cent, a.nq, a.seed + 100 + 1000 * b, kind=a.data)
        qn_b, qp_b = ivf._prepare(qs_b.copy())
        bs.append(dict(qn=qn_b, q_dev=torch.from_numpy(qn_b).to(device), qp_dev=torch.from_numpy(np.ascontiguousarray(qp_b)).to(device),
                       out=torch.full((a.nq, a.k), -1, dtype=torch.int64, device=device)))
    return bs

def rate(tag, dev, bs, a):
    r = bench.timed_rate(dev, bs, False, a.nq, a.k, 10, st, 2, 2)
    print(json.dumps({"what": tag, "M_qps": round(r["queries_per_s"] / 1e6, 2), "ms": round(r["ms_per_step"], 4)}), flush=True)

a1, a2 = mk(1), mk(2)
ivf1, cent1 = bench.build_index(a1, device)
dev1 = ivf1.device_index()
bs1 = batches(ivf1, cent1, a1)
rate("b1 alone in the process", dev1, bs1, a1)
dev1.set_pipeline(1)
if close_first:
    dev1.close()
ivf2, cent2 = bench.build_index(a2, device)
dev2 = ivf2.device_index()
bs2 = batches(ivf2, cent2, a2)
rate("b2 behind b1" + (" (closed)" if close_first else " (alive, pipeline 1)"), dev2, bs2, a2)
rate("b2 again", dev2, bs2, a2)
if not close_first:
    rate("b1 again", dev1, bs1, a1)
    dev1.set_pipeline(1)
    rate("b2 a third time", dev2, bs2, a2)
