import numpy as np, sys
sys.path.insert(0, '/root/repo')
from tinyknn_amd import IVF, FastPQ
from tinyknn_amd.utils import knn_brute
rng = np.random.RandomState(1)
n, d = 20037, 128
cent = rng.randn(60, d)
X = (cent[rng.randint(60, size=n)] + 0.6 * rng.randn(n, d)).astype(np.float32)
a = IVF("euclidean", 141, FastPQ(2)); a.fit(X[:8000])
print("centres dtype", a.all_centers.dtype, np.__version__)
try:
    print(np.show_runtime())
except Exception as e:
    print(e)
nh = knn_brute(X, a.all_centers, k=2, metric="euclidean")
nd = a._nearest_on_device(X, 2)
bad = np.flatnonzero((nh != nd).any(axis=1))
print("rows where device != numpy:", len(bad), bad[:10])
for r in bad[:5]:
    print(r, nh[r], nd[r])
col = np.ascontiguousarray(nh[:, 1])
o1, o2 = np.argsort(col), np.argsort(nh[:, 1])
print("argsort contiguous == argsort strided view:", np.array_equal(o1, o2), "repeat equal:", np.array_equal(o1, np.argsort(col)))
