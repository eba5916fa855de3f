"""round 5: ONE rank's share of a W-rank list partition, alone in its process (so that a kernel trace holds
nothing else).  Prints the bench leg's dict.  usage: python scripts/r05_rank_share.py [--world 8] [--workload glove|c5]
[--n ...] [--steps 20] [--depth 4] [--plain 1]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--workload", default="glove")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--depth", type=int, default=4)
ap.add_argument("--plain", type=int, default=1)
ap.add_argument("--windows", type=int, default=0)
ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--co", type=int, default=0)
ap.add_argument("--exchange", default="auto")
ap.add_argument("--replay-lazy", type=int, default=-1, help="TK_OPT_REPLAY_LAZY of every engine (A/B)")
ap.add_argument("--clusters", type=int, default=0)
a = ap.parse_args()
glove = a.workload == "glove"
args = argparse.Namespace(n=a.n or (1183514 if glove else 25_000_000), d=100 if glove else 128,
                          n_clusters=a.clusters or (1087 if glove else 5000), seed=10, build_probes=1,
                          metric="angular" if glove else "euclidean", data="glove-like", cache_dir="/tmp", fit_sample=100000,
                          data_file=None, nq=a.nq, k=10, n_probes=10, workload=a.workload, shard_depth=a.depth,
                          shard_plain=a.plain, warmup=5, steps=a.steps, windows=a.windows, backend="nccl", shard_coarse="home",
                          shard_counts="device", rank_share=a.world, shard_coalesce=0, rank_share_exchange=a.exchange)
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
from tinyknn_amd import _lib
_lib.check(_lib.lib().tk_set_device(0))
if a.replay_lazy >= 0:       # A/B: the lane replay's form on every shard handle of this process
    from tinyknn_amd import multi_gpu as MG
    _init = MG._HipShardEngine.__init__

    def _patched(self, *x, **kw):
        _init(self, *x, **kw)
        self.dev.set_option(_lib.OPT_REPLAY_LAZY, a.replay_lazy)

    MG._HipShardEngine.__init__ = _patched
ivf, cent = B.build_index(args, device) if glove else B.build_index_c5(args, device)
dev = ivf.device_index()
# the unsharded rate beside it (pipelined, pairs of calls), as bench.py's sweep measures it
batches = []
for b in range(4):
    qs = B.synth_queries(cent, args.nq, args.seed + 100 + 1000 * b, kind=args.data)
    qn, qp = ivf._prepare(qs.copy())
    batches.append(dict(q_dev=torch.from_numpy(qn).to(device), qp_dev=torch.from_numpy(np.ascontiguousarray(qp)).to(device),
                        out=torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device)))
f64 = qp.dtype != np.float32
pairs = 2 * args.nq <= dev.max_sub_batch(args.k, args.n_probes)
un = B.timed_rate(dev, batches, f64, args.nq, args.k, args.n_probes, torch.cuda.current_stream().cuda_stream, 2, 2 if pairs else 1)
dev.set_pipeline(1)
qn_t, qp_t, want = B.shard_inputs(args, ivf, cent, dev, device)
rs = B.rank_share_leg(args, ivf, device, qn_t, qp_t, want, a.world, a.co)
W = rs["world"]
rs["unsharded_ms_per_step"] = un["ms_per_step"]
rs["target_ms_per_step_at_0.7_efficiency"] = un["ms_per_step"] / (0.7 * W)
rs["implied_strong_scaling_efficiency_without_links"] = un["ms_per_step"] / W / rs["ms_per_step"]
print(json.dumps(rs), flush=True)
