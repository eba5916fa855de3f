"""DistanceTable.top(k) of ONE query per call (fast_pq.py:284-312; examples/example.py's loop) over arrays of 16 000 ... 1M rows:
ms per call of the table build (host numpy, as the reference) and of top() (scan + heap replay on the device + host rescoring),
with the code array resident in HBM (tinyknn_amd._fast_pq.cache_device_codes, as examples/example.py sets it) or uploaded per call
(TINYKNN_CACHE_CODES=0).
usage: python scripts/r06_top_one.py"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinyknn_amd
import tinyknn_amd._fast_pq as _fp
_fp.cache_device_codes = os.environ.get("TINYKNN_CACHE_CODES", "1") == "1"      # examples/example.py keeps the code array in HBM
for n, d in ((16000, 128), (60000, 128), (250000, 128), (1000000, 128)):
    np.random.seed(10)
    cent = np.random.randn(100, d)
    X = (cent[np.random.randint(100, size=n)] + 0.7 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(100, size=300)] + 0.7 * np.random.randn(300, d)).astype(np.float32)
    pq = tinyknn_amd.FastPQ(2)
    pq.fit(X[:30000])
    td = pq.transform(X)
    for k in (10, 50):
        dts = [pq.distance_table(qs[i]) for i in range(205)]
        t0 = time.perf_counter()
        for i in range(200):
            pq.distance_table(qs[i])
        t_tab = (time.perf_counter() - t0) / 200
        for i in range(5):
            dts[200 + i].top(td, X, k)
        t0 = time.perf_counter()
        got = [dts[i].top(td, X, k) for i in range(200)]
        t = (time.perf_counter() - t0) / 200
        print(json.dumps({"rows": n, "k": k, "heap": 2 * k + 10, "ms_table_host": round(t_tab * 1e3, 4), "ms_top": round(t * 1e3, 4),
                          "codes_resident": _fp.cache_device_codes}), flush=True)
