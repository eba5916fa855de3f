import numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, tinyknn_amd
from tinyknn_amd import IVF, FastPQ
from oracle import oracle
np.random.seed(11)
n, nq, d = 40000, 1800, 100
cent = np.random.randn(200, d)
X = (cent[np.random.randint(200, size=n)] + 0.6 * np.random.randn(n, d)).astype(np.float32)
qs = (cent[np.random.randint(200, size=nq)] + 0.6 * np.random.randn(nq, d)).astype(np.float32)
ivf = IVF("angular", 180, FastPQ(2)); ivf.fit(X[:15000]).build(X, n_probes=1)
L = len(ivf.active_centers)
ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers, ivf.pq_transformed_centers.packed,
    [ivf.pq_transformed_points[i].packed for i in range(L)], [ivf.pq_transformed_points[i].size for i in range(L)], [ivf.ids[i] for i in range(L)], ivf.data)
want = ox.query_batch(ivf._prepare(qs.copy())[0], 10, 5)
dev = ivf.device_index()
for co in (1, 2):
    dev.set_pipeline(2); dev.set_coalesce(co)
    type(dev).CHUNK = 500
    got = dev.query_raw(qs, 10, 5)
    bad = np.nonzero((got != want).any(axis=1))[0]
    print("coalesce", co, "bad rows", len(bad), bad[:10], bad[-10:] if len(bad) else "")
    if len(bad):
        print(got[bad[0]], want[bad[0]])
