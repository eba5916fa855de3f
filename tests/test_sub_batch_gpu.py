"""A device-resident batch beyond ONE workspace (tk_index_max_sub_batch): tk_index_query_batch_dev cuts it into equal
parts, each through the pipeline by itself — the rows must be the oracle's whatever the cut.  The workspace size is read
once per process (TINYKNN_WORKSPACE_GB), so the case runs in a child process with the smallest one (0.25 GB)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from tinyknn_amd import IVF, FastPQ, _lib
from oracle import oracle
assert _lib.device_count() >= 1, "no GPU visible"
np.random.seed(5)
n, d, nq0 = 40000, 48, 1500
cent = np.random.randn(150, d)
X = (cent[np.random.randint(150, size=n)] + 0.6 * np.random.randn(n, d)).astype(np.float32)
qs = (cent[np.random.randint(150, size=nq0)] + 0.6 * np.random.randn(nq0, d)).astype(np.float32)
ivf = IVF("euclidean", 160, FastPQ(2))
ivf.fit(X[:15000]).build(X, n_probes=1)
L = len(ivf.active_centers)
ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                        ivf.pq_transformed_centers.packed,
                        [ivf.pq_transformed_points[i].packed for i in range(L)],
                        [ivf.pq_transformed_points[i].size for i in range(L)],
                        [ivf.ids[i] for i in range(L)], ivf.data)
qn0, qp0 = ivf._prepare(qs.copy())
k, n_probes = 10, 100
want0 = ox.query_batch(qn0, k, n_probes)
dev = ivf.device_index()
ms = dev.max_sub_batch(k, n_probes)
reps = -(-(2 * ms + 5) // nq0)
sel = np.concatenate([np.random.permutation(nq0) for _ in range(reps)])[:2 * ms + 5]     # 2 ms + 5 rows: three parts
qn, qp, want = np.ascontiguousarray(qn0[sel]), np.ascontiguousarray(qp0[sel]), want0[sel]
nq = len(qn)
assert ms < nq <= 3 * ms, (ms, nq)
st = torch.cuda.current_stream().cuda_stream
q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(qp).cuda()
f64 = qp.dtype != np.float32
for depth, co in ((1, 1), (2, 1), (2, 2)):
    dev.set_pipeline(depth)
    dev.set_coalesce(co)
    for rep in range(2):
        out = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
        dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), f64, nq, k, n_probes, out.data_ptr(), stream=st)
        # ... and a batch of one workspace exactly behind it (pairs up with nothing: the big one went in parts)
        out2 = torch.full((ms, k), -7, dtype=torch.int64, device="cuda")
        dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), f64, ms, k, n_probes, out2.data_ptr(), stream=st)
        dev.join(st)
        torch.cuda.synchronize()
        got, got2 = out.cpu().numpy(), out2.cpu().numpy()
        bad = int((got != want).any(axis=1).sum()) + int((got2 != want[:ms]).any(axis=1).sum())
        assert bad == 0, (depth, co, rep, bad)
dev.set_pipeline(1)
print("SUB_BATCH_OK", ms, nq)
'''


def test_batch_beyond_one_workspace(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, TINYKNN_WORKSPACE_GB="0.25")
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SUB_BATCH_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    ms, nq = (int(x) for x in r.stdout.split("SUB_BATCH_OK")[1].split()[:2])
    assert ms < nq
