"""round 5: what does a replay of the captured pipelined mode look like on the device?  Captures `--gsteps` steps
(pairs of calls, depth 2) + join as ONE hipGraph, replays it, and — run under rocprofv3 --kernel-trace — lets
scripts/r05_graph_trace_read.py compare queues / overlap of the replays with the stream-launched run beside it."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--gsteps", type=int, default=32)
a = ap.parse_args()
args = argparse.Namespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=1, metric="angular", data="glove-like",
                          cache_dir="/tmp", fit_sample=100000, data_file=None, nq=10000, k=10, n_probes=10)
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
from tinyknn_amd import _lib
_lib.check(_lib.lib().tk_set_device(0))
ivf, cent = B.build_index(args, device)
dev = ivf.device_index()
batches = []
for b in range(4):
    qs = B.synth_queries(cent, args.nq, args.seed + 100 + 1000 * b, kind=args.data)
    qn, qp = ivf._prepare(qs.copy())
    batches.append(dict(q_dev=torch.from_numpy(qn).to(device), qp_dev=torch.from_numpy(np.ascontiguousarray(qp)).to(device),
                        out=torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device)))
stream = torch.cuda.current_stream().cuda_stream
un = B.timed_rate(dev, batches, False, args.nq, args.k, args.n_probes, stream, 2, 2)
print("stream-launched", un, flush=True)
# marker kernels around the phases: torch fills of distinctive sizes
import ctypes
def marker():       # a kernel with a name nothing else in the run has (read_only_kernel)
    v = ctypes.c_double(0.0)
    _lib.check(_lib.lib().tk_measure_read_bandwidth(1 << 20, 1, ctypes.byref(v)))
    torch.cuda.synchronize()
gouts = [torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device) for _ in range(4)]

def gstep(i, st):
    bb = batches[i % 4]
    dev.query_batch_dev(bb["q_dev"].data_ptr(), bb["qp_dev"].data_ptr(), False, args.nq, args.k, args.n_probes,
                        gouts[i % 4].data_ptr(), stream=st)

side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for i in range(a.gsteps):
        gstep(i, side.cuda_stream)
    dev.join(side.cuda_stream)
torch.cuda.synchronize()
dev.quiesce()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cs = torch.cuda.current_stream().cuda_stream
    for i in range(a.gsteps):
        gstep(i, cs)
    dev.join(cs)
dev.quiesce()
g.replay()
torch.cuda.synchronize()
marker()                        # graph replays start
t0 = time.perf_counter()
for _ in range(6):
    g.replay()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 6
marker()                        # graph replays end
print("graph", {"ms_per_step": el / a.gsteps * 1e3, "queries_per_s": args.nq * a.gsteps / el}, flush=True)
# the same steps stream-launched between two more markers
for rep in range(6):
    for i in range(a.gsteps):
        gstep(i, stream)
    dev.join(stream)
torch.cuda.synchronize()
marker()
