"""Eight list-shards of the bench index simulated on ONE GPU (tests/test_shard_gpu.py's
simulate_world): default capacity does not overflow, ids equal the unsharded index's."""
import sys, types, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench
from test_shard_gpu import simulate_world
args = types.SimpleNamespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=1, metric="angular",
                             data="glove-like", cache_dir="/tmp", fit_sample=100000, workload="glove")
ivf, cent = bench.build_index(args, torch.device("cuda", 0))
qs = bench.synth_queries(cent, 10000, 110)
qn, qp = ivf._prepare(qs.copy())
want = ivf.device_index().query_batch(qn, qp, 10, 10)
for W in (8, 4):
    for ex in ("dense", "filtered"):
        t = time.time()
        st = {}
        ids, flags, cap = simulate_world(ivf, W, qn, qp, 10, 10, exchange=ex, stats=st)
        extra = ""
        if ex == "filtered":
            extra = (f", records {st['records']} x 20 B = {st['records'] * 20 / W / 1e6:.2f} MB per rank against "
                     f"{st['dense_blocks'] * 16 / W / 1e6:.2f} MB of whole segments")
        print(f"W={W} {ex}: capacity {cap} uint4 per region ({W * cap * 16 / 1e6:.1f} MB all-to-all per rank a priori, "
              f"longest stream {st['usage']} uint4 -> {W * int(1.25 * st['usage']) * 16 / 1e6:.1f} MB trimmed), "
              f"overflow flags {flags.tolist()}, identical rows {int((ids == want).all(axis=1).sum())}/10000{extra}, "
              f"{time.time() - t:.1f}s")

# the exchange size bench.py --gpus 8 uses (8 steps coalesced: 80 000 queries per exchange)
qn8, qp8 = np.concatenate([qn] * 8), np.concatenate([qp] * 8)
for ex in ("dense", "filtered"):
    t = time.time()
    st = {}
    ids, flags, cap = simulate_world(ivf, 8, qn8, qp8, 10, 10, exchange=ex, stats=st)
    same = min(int((ids[j * 10000:(j + 1) * 10000] == want).all(axis=1).sum()) for j in range(8))
    print(f"W=8 {ex}, 80000 queries per exchange: capacity {cap} uint4, longest stream {st['usage']}, overflow flags "
          f"{flags.tolist()}, identical rows per step >= {same}/10000, {time.time() - t:.1f}s")
