#!/usr/bin/env python3
"""An IVF index whose inverted lists are sharded by cluster id over the GPUs of one node
(the reference has no multi-device code).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/sharded.py                      # RCCL over xGMI
    python examples/sharded.py                   # one GPU: the same code, the exchange is a copy
    python -m torch.distributed.run ... examples/sharded.py --backend gloo   # rehearsal on one GPU

Every rank builds the same index from the seed (vectors generated in HBM, never on the host),
keeps the codes of the lists it owns, and every rank receives all the ids — identical to what
the unsharded index and the reference return."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--d", type=int, default=96)
ap.add_argument("--n-clusters", type=int, default=1500)
ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--n-probes", type=int, default=10)
ap.add_argument("--backend", default="nccl")
ap.add_argument("--exchange", default="auto", choices=["auto", "dense", "filtered"])
args = ap.parse_args()

import torch                                       # noqa: E402
import torch.distributed as dist                   # noqa: E402
from tinyknn_amd import IVF, FastPQ                # noqa: E402
from tinyknn_amd.ivf import synth_rows             # noqa: E402
from tinyknn_amd.multi_gpu import ListShardedIndex # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
if world > 1:
    dist.init_process_group(args.backend, rank=rank, world_size=world)

seed, sigma = 7, 0.6
cent = np.random.RandomState(1).randn(300, args.d).astype(np.float32)
ivf = IVF("euclidean", args.n_clusters, FastPQ(2))
np.random.seed(0)        # fit draws its k-means starts from numpy's global RNG: every rank the same index
ivf.fit(synth_rows(60000, args.d, seed, cent, sigma))          # a sample of the same generator
t0 = time.perf_counter()
ivf.build_resident(args.rows, args.d, seed, cent, sigma)          # generate + assign + encode + pack in HBM
unsharded = None
if rank == 0 and world == 1:
    unsharded = ivf.device_index()
idx = ListShardedIndex(ivf, exchange=args.exchange)            # shards the resident index in place
if rank == 0:
    print(f"{args.rows} x {args.d} in {len(ivf.list_sizes)} lists over {world} rank(s), built and sharded in "
          f"{(time.perf_counter() - t0) * 1e3:.0f} ms; rank 0 keeps {int((idx.owner == 0).sum())} lists")

qs = synth_rows(args.nq, args.d, seed + 1, cent, sigma)
ids = idx.query_batch(qs, args.k, n_probes=args.n_probes)      # same batch on every rank, all ids back
t0 = time.perf_counter()
ids = idx.query_batch(qs, args.k, n_probes=args.n_probes)
dt = time.perf_counter() - t0
if rank == 0:
    print(f"{args.nq} raw queries -> ids in {dt * 1e3:.1f} ms ({args.nq / dt / 1e6:.2f} M queries/s, host "
          f"preparation and copies included), first row {ids[0].tolist()}")
if world > 1:
    dist.destroy_process_group()
