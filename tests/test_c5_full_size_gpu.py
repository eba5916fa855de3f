"""BASELINE configs[4] at its STATED size on one GPU: 100M x 128 float32 generated in HBM (51 GB),
10 000 inverted lists of ~10 000 rows, PQ rotated to 64 dims (M = 32, float64 table math), built on
the device (tk_index_build_dev), n_probes 10, k 10.

Checked against the CPU oracle fed with the exported lists (its vector file is sparse: only the heap
candidates' rows travel): the first 300 rows of a batch of 10 000 queries, and probe lists + heap
arrays (layout included) of 40 of them — with one batch in flight, and pipelined with pairs of calls
as one batch (the mode `bench.py --workload c5` times).  Then ONE simulated rank of a W = 8 list
partition of the same index (its 1/8 of the lists, sharded where the codes lie) must return the same
rows for its home queries.  Needs ~80 GB of HBM; skipped on a smaller device."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, N_LISTS, NQ, SEED = 100_000_000, 128, 10_000, 10_000, 10
ROWS, HEAPS = 300, 40


def synth_rows(n, d, seed, centres, sigma, row0=0):
    from tinyknn_amd import _lib
    out = np.zeros((n, d), dtype=np.float32)
    c = np.ascontiguousarray(centres, dtype=np.float32)
    _lib.check(_lib.lib().tk_synth_rows(_lib.ptr(out, _lib._f32p), row0, n, d, seed, c.ctypes.data, len(c),
                                        float(sigma)))
    return out


@pytest.fixture(scope="module")
def c5(oracle, tmp_path_factory):
    import torch
    from tinyknn_amd import IVF, FastPQ
    free, total = torch.cuda.mem_get_info()
    if total < 150e9:
        pytest.skip("configs[4] at 100M x 128 needs a 288 GB device")
    cent = np.random.RandomState(SEED).randn(3000, D).astype(np.float32)
    ivf = IVF("euclidean", N_LISTS, FastPQ(2))
    sample = synth_rows(400_000, D, SEED, cent, 0.7)        # rows 0 .. 400k of the data set itself
    # coarse centres = rows of the data set: every centre is the nearest centre of at least one row
    # (tk_index_build_dev refuses an empty list in front of a used one, like utils.py:128)
    ivf.all_centers = sample[np.random.RandomState(3).choice(len(sample), N_LISTS, replace=False)].copy()
    np.random.seed(5)
    ivf.pq.fit(sample[:20000])
    del sample
    ivf.build_resident(N, D, SEED, cent, 0.7)
    dev = ivf.device_index()
    sizes, codes, ids = dev.export_lists()
    assert sizes.sum() == N and sizes.min() > 0
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    tmp = tmp_path_factory.mktemp("c5")
    data = np.memmap(str(tmp / "rows.f32"), dtype=np.float32, mode="w+", shape=(N, D))      # sparse
    ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                            ivf.pq_transformed_centers.packed,
                            [codes[coff[i]:coff[i + 1]] for i in range(N_LISTS)], list(sizes),
                            [ids[ioff[i]:ioff[i + 1]] for i in range(N_LISTS)], data)
    assert ox.data is data or np.may_share_memory(ox.data, data), "the oracle copied the sparse vector file"
    del codes, ids
    qs = synth_rows(NQ, D, SEED + 101, cent, 0.7)
    qn, qp = ivf._prepare(qs.copy())
    yield torch, ivf, dev, ox, data, qn, np.ascontiguousarray(qp), sizes
    dev.close()
    try:
        os.unlink(str(tmp / "rows.f32"))
    except OSError:
        pass


def test_100m_x_128_against_the_oracle(c5):
    torch, ivf, dev, ox, data, qn, qp, sizes = c5
    assert np.median(sizes) > 50 * 111          # lists far longer than the heap of (10 + 1) * 10 + 1 entries
    dev.set_pipeline(1)
    _, dbg = dev.query_batch(qn[:ROWS], qp[:ROWS], 10, 10, debug=True)          # heaps of the checked rows
    rows = np.unique(dbg["heap_idx"][dbg["heap_idx"] >= 0])
    data[rows] = dev.read_rows(rows)
    want = ox.query_batch(qn[:ROWS], 10, 10)
    for i in range(HEAPS):
        _, w = ox.query(qn[i], 10, 10, debug=True)
        np.testing.assert_array_equal(w["probes"], dbg["probes"][i])
        np.testing.assert_array_equal(w["heap_idx"], dbg["heap_idx"][i])
        np.testing.assert_array_equal(w["heap_val"], dbg["heap_val"][i])
    # one batch of 10 000 queries in flight (list-major scans, plain sums on the matrix cores)
    got = dev.query_batch(qn, qp, 10, 10)
    np.testing.assert_array_equal(got[:ROWS], want)
    st = dev.plain_stats()
    # pipelined, pairs of calls as one batch of 20 000 queries
    dev.set_pipeline(2)
    dev.set_coalesce(2)
    q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(qp).cuda()
    s_ = torch.cuda.current_stream().cuda_stream
    outs = [torch.full((NQ, 10), -1, dtype=torch.int64, device="cuda") for _ in range(5)]
    for o in outs:
        dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), qp.dtype != np.float32, NQ, 10, 10, o.data_ptr(),
                            stream=s_)
    dev.join(s_)
    torch.cuda.synchronize()
    for o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), got)
    dev.set_coalesce(1)
    dev.set_pipeline(1)
    print("plain path of the one-batch call:", st)


def test_one_rank_of_eight_on_the_same_index(c5):
    """W = 8 list partition, rank 3: the index is sharded where it lies (tk_index_shard_resident); the
    segments of the other seven ranks come from seven more passes of the same device with their
    owner maps (a 1-GPU box cannot host eight RCCL ranks).  Home rows = the unsharded rows."""
    torch, ivf, dev, ox, data, qn, qp, sizes = c5
    from tinyknn_amd.multi_gpu import ListShardedIndex
    from simulated_peers import SimulatedPeers
    nq = 4000
    want = dev.query_batch(qn[:nq], qp[:nq], 10, 10)
    qn_t, qp_t = torch.from_numpy(qn[:nq]).cuda(), torch.from_numpy(qp[:nq]).cuda()
    peers = SimulatedPeers(ivf, world=8, rank=3)
    try:
        idx = ListShardedIndex(ivf, simulate=peers, depth=2, exchange="dense")
        got = idx.query_prepared(qn_t, qp_t, 10, 10)
        lo, hi = peers.home_range(nq)
        np.testing.assert_array_equal(got[lo:hi], want[lo:hi])
        idx.exchange = "filtered"
        peers.reset()
        got = idx.query_prepared(qn_t, qp_t, 10, 10)
        np.testing.assert_array_equal(got[lo:hi], want[lo:hi])
    finally:
        peers.close()
