"""round 5 debug: when does the one-phase sharded scan raise bit 4 / do exchange regions overflow in flight?"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--force", type=int, default=1)
a = ap.parse_args()
args = argparse.Namespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=1, metric="angular", data="glove-like",
                          cache_dir="/tmp", fit_sample=100000, data_file=None, nq=10000, k=10, n_probes=10)
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
from tinyknn_amd import _lib
_lib.check(_lib.lib().tk_set_device(0))
ivf, cent = B.build_index(args, device)
dev = ivf.device_index()
qn_t, qp_t, want = B.shard_inputs(args, ivf, cent, dev, device)
if a.force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
from tinyknn_amd.multi_gpu import ListShardedIndex
for depth, co, plain in ((1, 1, True), (4, 1, True), (4, 3, True), (4, 3, "two-phase")):
    idx = ListShardedIndex(ivf, depth=depth, coalesce=co, force_collectives=bool(a.force), plain=plain, exchange="dense")
    for i in range(3):
        got = idx.query_prepared(qn_t, qp_t, 10, 10)
    print(f"depth {depth} co {co} plain {plain}: sync rows same {(got == want).all(axis=1).sum()} plain_failed {idx._plain_failed} "
          f"cap {idx.capacity}", flush=True)
    if co > 1:
        qc, pc = torch.cat([qn_t] * co), torch.cat([qp_t] * co)
        for i in range(3):
            gotc = idx.query_prepared(qc, pc, 10, 10)
        print("   coalesced sync: plain_failed", idx._plain_failed, "cap", idx.capacity, flush=True)
    t_prev = None
    for rep in range(6):
        outs = []
        t0 = time.perf_counter()
        for _ in range(24):
            o = idx.submit(qn_t, qp_t, 10, 10)
            if o is not None:
                outs.append(o)
        try:
            idx.join()
            msg = "ok"
        except RuntimeError as e:
            msg = str(e)[:60]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        flags = [int(o[:, -1].sum().item()) for o in outs]
        print(f"   in flight rep {rep}: {el * 1e3:.2f} ms for 24 steps ({24 * 10000 / el / 1e6:.2f} M q/s) flags {flags} {msg} "
              f"one_phase_now {idx._one_phase_now(10, 10, None)} mem {torch.cuda.memory_allocated() / 1e9:.2f} GB", flush=True)
    del idx
    torch.cuda.empty_cache()
