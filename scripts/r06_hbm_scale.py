"""The exact code scan where it streams HBM (bench.hbm_scale_leg: 1 GiB of codes, nq = 1 / 4 / 16), alone — for A/Bs of
the flat kernel (TK_FLAT_UNROLL)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for _ in range(3):
    r = bench.hbm_scale_leg(dev)
    print(json.dumps([(c["nq"], round(c["ms"], 4), round(c["min_hbm_GBps"]), round(c["frac"], 4)) for c in r["cases"]]))
