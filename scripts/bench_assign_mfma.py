import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from tinyknn_amd import IVF, FastPQ
X, cent = bench.synth(1183514, 0, 100, 10)
rng=np.random.RandomState(1)
C = X[rng.choice(len(X),1087,replace=False)].astype(np.float32)
C /= np.linalg.norm(C,axis=1,keepdims=True)
data = X/np.linalg.norm(X,axis=1,keepdims=True)
for name, Y in (("f32 centres (MFMA)", C), ("f64 centres (VALU)", C.astype(np.float64))):
    ivf=IVF("angular",1087,FastPQ(2)); ivf.all_centers=Y
    ivf._nearest_on_device(data[:1000],1)
    ts=[]
    for _ in range(3):
        t=time.perf_counter(); near=ivf._nearest_on_device(data,1); ts.append(time.perf_counter()-t)
    print(name, "rows/s incl PCIe", len(data)/min(ts), "s", min(ts))
from tinyknn_amd.utils import knn_brute
want=knn_brute(data[:20000], C, 1, "angular")
ivf=IVF("angular",1087,FastPQ(2)); ivf.all_centers=C
print("identical to numpy on 20000 rows:", bool((ivf._nearest_on_device(data[:20000],1)==want).all()))
