import numpy as np, threading
rng = np.random.RandomState(1)
col = rng.randint(0, 141, size=20037).astype(np.int64)
two = np.stack([rng.randint(0, 141, size=20037), col], axis=1).astype(np.int64)
ref = np.argsort(col)
diff = 0
for i in range(300):
    a = np.argsort(two[:, 1]) if i % 2 else np.argsort(col.copy())
    diff += not np.array_equal(a, ref)
print("argsort differs from first result in", diff, "of 300 calls")
# with BLAS / other threads busy
stop = False
def burn():
    x = np.random.randn(400, 400)
    while not stop:
        x = x @ x; x /= np.abs(x).max()
ts = [threading.Thread(target=burn) for _ in range(8)]
[t.start() for t in ts]
diff = 0
for i in range(300):
    diff += not np.array_equal(np.argsort(two[:, 1]), ref)
stop = True
[t.join() for t in ts]
print("... with 8 threads in BLAS:", diff, "of 300")
