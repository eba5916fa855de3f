import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyknn_amd import FastPQ, IVF
from tinyknn_amd.ivf import DeviceIndex
np.random.seed(10)
n, d, nq = 200000, 100, 10000
cent = np.random.randn(300, d)
data = (cent[np.random.randint(300, size=n + nq)] + 0.7 * np.random.randn(n + nq, d)).astype(np.float32)
data, queries = data[:-nq], data[-nq:]
ivf = IVF("angular", 447, FastPQ(2))
ivf.fit(data[:100000]).build(data, n_probes=1, device=True)
for chunk in (10000, 5000, 2500, 1250):
    DeviceIndex.CHUNK = chunk
    ivf._dev = None
    for n_probes in (1, 5, 10):
        ivf.query_batch(queries, 10, n_probes=n_probes)
        t0 = time.time()
        for _ in range(5):
            ivf.query_batch(queries, 10, n_probes=n_probes)
        print("chunk", chunk, "n_probes", n_probes, "q/s", round(5 * nq / (time.time() - t0)))
