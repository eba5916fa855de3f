"""tk_index_set_coalesce(ix, 2): pairs of consecutive calls run through the pipeline as ONE batch.
The rows each call gets back must be the rows of separate calls (= the oracle's): equal and unequal
batch sizes, an odd number of calls (the last one is launched alone by the join), a call that
cannot join the held one (other n_probes), completion events per call, rotated float64 tables."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle):
    import torch
    import tinyknn_amd
    from tinyknn_amd import IVF, FastPQ, _lib
    assert _lib.device_count() >= 1, "no GPU visible"
    out = {}
    for tag, metric, d in (("an100", "angular", 100), ("eu64", "euclidean", 48)):
        np.random.seed(11)
        n, nq = 40000, 1800
        cent = np.random.randn(200, d)
        X = (cent[np.random.randint(200, size=n)] + 0.6 * np.random.randn(n, d)).astype(np.float32)
        qs = (cent[np.random.randint(200, size=nq)] + 0.6 * np.random.randn(nq, d)).astype(np.float32)
        ivf = IVF(metric, 180, FastPQ(2))
        ivf.fit(X[:15000]).build(X, n_probes=1)
        L = len(ivf.active_centers)
        ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                                ivf.pq_transformed_centers.packed,
                                [ivf.pq_transformed_points[i].packed for i in range(L)],
                                [ivf.pq_transformed_points[i].size for i in range(L)],
                                [ivf.ids[i] for i in range(L)], ivf.data)
        qn, qp = ivf._prepare(qs.copy())
        out[tag] = (ivf, ox, qn, np.ascontiguousarray(qp))
    return torch, out


@pytest.mark.parametrize("pair_nq", [4, 8192])
@pytest.mark.parametrize("tag", ["an100", "eu64"])
def test_pairs_of_calls_are_the_rows_of_separate_calls(setup, tag, pair_nq):
    """pair_nq: the suite's default (4: the lane kernels replay) and the product's (8192: launches of up to 4 096 queries
    — every pair here — replay through the register heap with batches in flight)."""
    from tinyknn_amd import _lib
    torch, fx = setup
    ivf, ox, qn, qp = fx[tag]
    f64 = qp.dtype != np.float32
    dev = ivf.device_index()
    dev.set_option(_lib.OPT_PAIR_NQ, pair_nq)
    dev.set_pipeline(2)
    dev.set_coalesce(2)
    st = torch.cuda.current_stream().cuda_stream
    want = {p: ox.query_batch(qn, 10, p) for p in (3, 10)}
    q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(qp).cuda()
    esz = 8 if f64 else 4
    dq = qp.shape[1]
    # (row range, n_probes) of every call; 7 calls: pairs (0,1), (2,3) of unequal sizes, call 4 held
    # and launched alone because call 5 has another n_probes, (5,6)
    calls = [((0, 900), 10), ((900, 1800), 10), ((0, 300), 10), ((300, 1001), 10), ((1001, 1500), 10),
             ((0, 700), 3), ((700, 1800), 3)]
    outs = []
    evs = [torch.cuda.Event() for _ in calls]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    for ((a, b), p), e in zip(calls, evs):
        o = torch.full((b - a, 10), -1, dtype=torch.int64, device="cuda")
        outs.append(o)
        dev.query_batch_dev(q_dev.data_ptr() + a * qn.shape[1] * 4, qp_dev.data_ptr() + a * dq * esz, f64, b - a,
                            10, p, o.data_ptr(), stream=st, done_event=e.cuda_event)
    # an odd call at the end is held until the join
    last = torch.full((500, 10), -1, dtype=torch.int64, device="cuda")
    dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), f64, 500, 10, 3, last.data_ptr(), stream=st)
    dev.join(st)
    for e in evs:
        e.synchronize()             # every call's own completion event fires
    torch.cuda.synchronize()
    for ((a, b), p), o in zip(calls, outs):
        np.testing.assert_array_equal(o.cpu().numpy(), want[p][a:b])
    np.testing.assert_array_equal(last.cpu().numpy(), want[3][:500])
    # and the same calls one batch each
    dev.set_coalesce(1)
    for (a, b), p in calls[:3]:
        o = torch.full((b - a, 10), -1, dtype=torch.int64, device="cuda")
        dev.query_batch_dev(q_dev.data_ptr() + a * qn.shape[1] * 4, qp_dev.data_ptr() + a * dq * esz, f64, b - a,
                            10, p, o.data_ptr(), stream=st)
        dev.join(st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(o.cpu().numpy(), want[p][a:b])
    dev.set_pipeline(1)
    dev.set_option(_lib.OPT_PAIR_NQ, 4)


def test_streaming_session_over_a_coalescing_index(setup):
    """IVF.query_batch's streaming sessions (raw queries in, ids out) on an index that pairs calls:
    waiting for a ticket whose batch is still held launches it."""
    torch, fx = setup
    ivf, ox, qn, qp = fx["an100"]
    np.random.seed(5)
    raw = (qn * (1.0 + np.random.rand(len(qn), 1))).astype(np.float32)      # un-normalised rows of the same directions
    dev = ivf.device_index()
    dev.set_pipeline(2)
    dev.set_coalesce(2)
    want = ox.query_batch(ivf._prepare(raw.copy())[0], 10, 5)
    chunk = type(dev).CHUNK
    type(dev).CHUNK = 500            # 4 submits (1800 rows: 500, 500, 500, 300)
    try:
        got = dev.query_raw(raw, 10, 5)
    finally:
        type(dev).CHUNK = chunk
    np.testing.assert_array_equal(got, want)
    dev.set_coalesce(1)
    dev.set_pipeline(1)


@pytest.mark.parametrize("n_calls", [1, 2, 3])
def test_drain_of_few_batches(setup, n_calls):
    """A join behind one, two or three pipelined calls.  (With exactly two, the list scan of the
    first batch used to be enqueued behind the table build of the second only — which was recorded
    BEFORE the first batch's descriptors: found in round 4 through the coalesced sessions.)"""
    torch, fx = setup
    ivf, ox, qn, qp = fx["an100"]
    dev = ivf.device_index()
    dev.set_pipeline(2)
    dev.set_coalesce(1)
    st = torch.cuda.current_stream().cuda_stream
    want = ox.query_batch(qn, 10, 8)
    q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(qp).cuda()
    for rep in range(3):
        torch.cuda.synchronize()        # an idle device: nothing queued ahead hides a missing wait
        outs = [torch.full((len(qn), 10), -1, dtype=torch.int64, device="cuda") for _ in range(n_calls)]
        for o in outs:
            dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), False, len(qn), 10, 8, o.data_ptr(), stream=st)
        dev.join(st)
        torch.cuda.synchronize()
        for o in outs:
            np.testing.assert_array_equal(o.cpu().numpy(), want)
    dev.set_pipeline(1)


@pytest.mark.parametrize("n_probes", [1, 64, 65, 100])
def test_slot_descriptors_by_the_coarse_rescoring_or_by_their_own_kernel(setup, n_probes):
    """Up to 64 probed lists the wave that ranked a query's lists writes their scan descriptors and counts the
    pairs (rescore.hip: slots_epilogue); above, make_slots_kernel does, as a launch of its own.  Pairs of calls
    of unequal sizes through the pipeline on either side of the limit: the oracle's rows."""
    torch, fx = setup
    ivf, ox, qn, qp = fx["an100"]
    dev = ivf.device_index()
    dev.set_pipeline(2)
    dev.set_coalesce(2)
    st = torch.cuda.current_stream().cuda_stream
    nq = 600
    want = ox.query_batch(qn[:nq], 10, n_probes)
    q_dev, qp_dev = torch.from_numpy(qn[:nq]).cuda(), torch.from_numpy(qp[:nq]).cuda()
    dq = qp.shape[1]
    outs = []
    for a, b in ((0, 250), (250, 600), (0, 600)):
        o = torch.full((b - a, 10), -1, dtype=torch.int64, device="cuda")
        outs.append(((a, b), o))
        dev.query_batch_dev(q_dev.data_ptr() + a * qn.shape[1] * 4, qp_dev.data_ptr() + a * dq * 4, False, b - a,
                            10, n_probes, o.data_ptr(), stream=st)
    dev.join(st)
    torch.cuda.synchronize()
    for (a, b), o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want[a:b])
    dev.set_coalesce(1)
    dev.set_pipeline(1)
