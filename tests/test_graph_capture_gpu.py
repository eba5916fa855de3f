"""hipGraph capture of the device-pointer path (BASELINE configs[3]: "hipGraph-captured batch").

The fused slot descriptors of the coarse rescoring live in an immutable per-workspace pool whose
uploads come from page-locked memory (api_internal.h): a capture made right after ONE warm-up
call on a fresh index — the warm-up is the plain path's PROBE batch, the captured call then sees
another plain state, i.e. another descriptor — must replay the rows of the stream-launched calls,
also after later stream-launched batches with yet other descriptors have run in between."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle):
    import torch
    from tinyknn_amd import IVF, FastPQ, _lib
    assert _lib.device_count() >= 1, "no GPU visible"
    np.random.seed(5)
    n, nq, d = 60000, 2500, 100
    cent = np.random.randn(150, d)
    X = (cent[np.random.randint(150, size=n)] + 0.6 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(150, size=nq)] + 0.6 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 120, FastPQ(2))
    ivf.fit(X[:15000]).build(X, n_probes=1)
    L = len(ivf.active_centers)
    ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                            ivf.pq_transformed_centers.packed,
                            [ivf.pq_transformed_points[i].packed for i in range(L)],
                            [ivf.pq_transformed_points[i].size for i in range(L)],
                            [ivf.ids[i] for i in range(L)], ivf.data)
    qn, qp = ivf._prepare(qs.copy())
    return torch, ivf, ox, qn, np.ascontiguousarray(qp)


@pytest.mark.parametrize("depth", [1, 2])
def test_capture_right_after_one_warm_up_call(setup, depth):
    torch, ivf, ox, qn, qp = setup
    from tinyknn_amd.ivf import DeviceIndex
    nq = len(qn)
    want = ox.query_batch(qn, 10, 8)
    dev = DeviceIndex(ivf)                      # a FRESH index: plain state PROBE
    dev.set_pipeline(depth)
    q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(qp).cuda()
    side = torch.cuda.Stream()
    n_calls = 1 if depth == 1 else 5
    outs = [torch.full((nq, 10), -1, dtype=torch.int64, device="cuda") for _ in range(n_calls)]

    def calls(st):
        for o in outs:
            dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), False, nq, 10, 8, o.data_ptr(), stream=st)
        dev.join(st)

    with torch.cuda.stream(side):
        # exactly one warm-up pass (depth 1: ONE call): workspaces, events and streams exist afterwards
        calls(side.cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want)
    dev.quiesce()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        calls(torch.cuda.current_stream().cuda_stream)
    dev.quiesce()
    for rep in range(2):
        for o in outs:
            o.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        for o in outs:
            np.testing.assert_array_equal(o.cpu().numpy(), want)
        # stream-launched batches in between, with other plain states (other descriptors in the
        # workspaces' pools): what the graph reads must not change
        dev.set_plain_scan(False if rep == 0 else "always")
        tmp = torch.full((nq, 10), -1, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2 * depth + 6):
            dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), False, nq, 10, 8, tmp.data_ptr(), stream=st)
        dev.join(st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(tmp.cpu().numpy(), want)
        dev.quiesce()
    for o in outs:
        o.fill_(-1)
    g.replay()
    torch.cuda.synchronize()
    for o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want)
    del g
    dev.close()
