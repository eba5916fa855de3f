"""IVF.build(n_probes = 2 / 3) (the reference's default is 2, ivf.py:53): every row in two / three lists, and
`insert`'s duplicate test (_fast_pq.pyx:284-287) decides what a label's later copies do.  The lane replay's TWIN
form (heap.hip; lemma: tests/test_twin_dedupe_lemma.py) against the oracle: heap arrays with their layout and
final ids, with the exact kernel alone and with the plain sums on the matrix cores behind the heads, lazy and
staged, one batch in flight and pipelined in pairs; and against the hash-set form it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 3])
def setup(request, oracle):
    import torch
    from tinyknn_amd import IVF, FastPQ, _lib
    assert _lib.device_count() >= 1, "no GPU visible"
    b = request.param
    np.random.seed(20 + b)
    n, nq, d = 120000, 3000, 100
    cent = np.random.randn(400, d)
    X = (cent[np.random.randint(400, size=n)] + 0.7 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(400, size=nq)] + 0.7 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 150, FastPQ(2))
    ivf.fit(X[:30000]).build(X, n_probes=b)
    L = len(ivf.active_centers)
    ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                            ivf.pq_transformed_centers.packed,
                            [ivf.pq_transformed_points[i].packed for i in range(L)],
                            [ivf.pq_transformed_points[i].size for i in range(L)],
                            [ivf.ids[i] for i in range(L)], ivf.data)
    qn, qp = ivf._prepare(qs.copy())
    return torch, _lib, b, ivf, ox, qn, np.ascontiguousarray(qp)


def test_twin_table_lists_every_other_copy(setup):
    torch, _lib, b, ivf, ox, qn, qp = setup
    dev = ivf.device_index()
    tl, to = dev.twin_table()
    assert tl.shape == (sum(len(x) for x in ivf.ids), b - 1)
    ids = np.concatenate(ivf.ids)
    off = np.concatenate([[0], np.cumsum([len(x) for x in ivf.ids])])
    assert (tl >= 0).all()
    flat = off[tl] + to                           # flat rows of the other copies
    assert (ids[flat] == ids[:, None]).all()      # they carry the row's label
    assert (flat != np.arange(len(ids))[:, None]).all()
    if b == 3:
        assert (flat[:, 0] != flat[:, 1]).all()


@pytest.mark.parametrize("plain", [False, "always"])
def test_heaps_and_ids_against_the_oracle(setup, plain):
    torch, _lib, b, ivf, ox, qn, qp = setup
    dev = ivf.device_index()
    dev.set_plain_scan(plain)                    # exact kernel only / plain sums behind the heads
    try:
        for n_probes in (2, 10, 25):
            want = ox.query_batch(qn, 10, n_probes)
            ref = None
            for lazy, twin in ((0, 1), (1, 1), (0, 0)):
                dev.set_option(_lib.OPT_REPLAY_LAZY, lazy)
                dev.set_option(_lib.OPT_REPLAY_TWIN, twin)
                out, dbg = dev.query_batch(qn, qp, 10, n_probes, debug=True)
                np.testing.assert_array_equal(out, want)
                if ref is None:
                    ref = dbg
                    for qi in range(0, len(qn), 97):
                        _, odbg = ox.query(qn[qi], 10, n_probes=n_probes, debug=True)
                        np.testing.assert_array_equal(dbg["heap_idx"][qi], odbg["heap_idx"], err_msg=f"q{qi} p{n_probes}")
                        np.testing.assert_array_equal(dbg["heap_val"][qi], odbg["heap_val"], err_msg=f"q{qi} p{n_probes}")
                else:
                    np.testing.assert_array_equal(dbg["heap_idx"], ref["heap_idx"])
                    np.testing.assert_array_equal(dbg["heap_val"], ref["heap_val"])
    finally:
        dev.set_option(_lib.OPT_REPLAY_LAZY, -1)
        dev.set_option(_lib.OPT_REPLAY_TWIN, 1)
        dev.set_plain_scan(True)


def test_forced_rescan_of_every_query(setup):
    """table limits forced down: every query of a plain batch fails the lemma's check, is scanned again exactly and
    replayed by the packed kernel with the reference's scan of the labels"""
    torch, _lib, b, ivf, ox, qn, qp = setup
    dev = ivf.device_index()
    dev.set_plain_scan("always")
    dev.set_option(_lib.OPT_PLAIN_LIMIT, -128)
    try:
        want = ox.query_batch(qn[:400], 10, 8)
        np.testing.assert_array_equal(dev.query_batch(qn[:400], qp[:400], 10, 8), want)
    finally:
        dev.set_option(_lib.OPT_PLAIN_LIMIT, 0x7fffffff)
        dev.set_plain_scan(True)


def test_pipelined_pairs(setup):
    torch, _lib, b, ivf, ox, qn, qp = setup
    f64 = qp.dtype != np.float32
    dev = ivf.device_index()
    dev.set_pipeline(2)
    dev.set_coalesce(2)
    try:
        st = torch.cuda.current_stream().cuda_stream
        want = ox.query_batch(qn, 10, 10)
        q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(qp).cuda()
        esz = 8 if f64 else 4
        dq = qp.shape[1]
        calls = [(0, 700), (700, 1500), (1500, 1501), (1501, 2400), (2400, 3000), (0, 3000), (100, 2100)]
        outs = []
        for rep in range(2):                      # (the second round runs with the plain path's verdict in)
            for a, e in calls:
                o = torch.full((e - a, 10), -1, dtype=torch.int64, device="cuda")
                outs.append((a, e, o))
                dev.query_batch_dev(q_dev.data_ptr() + a * qn.shape[1] * 4, qp_dev.data_ptr() + a * dq * esz, f64,
                                    e - a, 10, 10, o.data_ptr(), stream=st)
        dev.join(st)
        torch.cuda.synchronize()
        for a, e, o in outs:
            np.testing.assert_array_equal(o.cpu().numpy(), want[a:e])
    finally:
        dev.set_coalesce(1)
        dev.set_pipeline(1)


def _oracle_of(oracle, ivf):
    L = len(ivf.active_centers)
    return oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                              ivf.pq_transformed_centers.packed,
                              [ivf.pq_transformed_points[i].packed for i in range(L)],
                              [ivf.pq_transformed_points[i].size for i in range(L)],
                              [ivf.ids[i] for i in range(L)], ivf.data)


@pytest.mark.parametrize("damage", ["other_code", "same_list", "none"])
def test_lists_that_do_not_meet_the_premises_keep_the_label_test(oracle, damage):
    """The TWIN form rests on: copies of a label lie in different lists and carry the same code.  IVF.build's lists do;
    an uploaded index may not (the reference's `insert` compares LABELS, whatever the codes).  The library checks both
    on the device when the lists arrive and keeps the label-based duplicate test for an index that fails; either way
    the heaps and ids are the oracle's for the lists as given."""
    from conftest import golden
    from test_hip_parity import ivf_from_fixture
    from tinyknn_amd.fast_pq import TransformedData
    g = golden("g6_ivf_an100b2.npz")
    ivf = ivf_from_fixture(None, g)
    big = int(np.argmax(g["list_sizes"]))
    if damage == "other_code":          # every 5th row of the longest list gets another code: its copy elsewhere keeps the old one
        td = ivf.pq_transformed_points[big]
        pk = np.array(td.packed, copy=True)
        b = pk.view(np.uint8).reshape(pk.shape[0], pk.shape[1] // 2, 16)
        b[:, :, ::5] ^= 0x35
        ivf.pq_transformed_points[big] = TransformedData(td.size, pk)
    elif damage == "same_list":         # a label twice in one list
        ids = np.array(ivf.ids[big], copy=True)
        ids[1::7] = ids[0:-1:7][:len(ids[1::7])]
        ivf.ids[big] = ids
    dev = ivf.device_index()
    assert (dev.twin_table_width() > 0) == (damage == "none")
    ox = _oracle_of(oracle, ivf)
    for n_probes in (5, 10):
        out, dbg = dev.query_batch(g["qn"], g["qpq"], 10, n_probes, debug=True)
        for qi in range(len(g["qn"])):
            ids_, odbg = ox.query(g["qn"][qi], 10, n_probes=n_probes, debug=True)
            np.testing.assert_array_equal(dbg["heap_idx"][qi], odbg["heap_idx"], err_msg=f"{damage} q{qi} p{n_probes}")
            np.testing.assert_array_equal(dbg["heap_val"][qi], odbg["heap_val"], err_msg=f"{damage} q{qi} p{n_probes}")
            np.testing.assert_array_equal(out[qi][:len(ids_)], ids_)


def test_a_sharded_rank_vouches_only_for_lists_it_has_checked():
    """tk_index_set_lists_shard hands a rank the codes of its own lists only: the library cannot check the premises, the
    Python wrapper does on the host (all lists are there) and vouches — or does not."""
    from conftest import golden
    from test_hip_parity import ivf_from_fixture
    from tinyknn_amd.fast_pq import TransformedData
    from tinyknn_amd.ivf import DeviceIndex
    from tinyknn_amd.multi_gpu import shard_lists
    from tinyknn_amd import _lib
    g = golden("g6_ivf_an100b2.npz")
    ivf = ivf_from_fixture(None, g)
    owner = shard_lists(np.asarray(g["list_sizes"], dtype=np.int64), 2)
    good = DeviceIndex(ivf, owner, 0, 2)
    assert good.twin_table_width() > 0 and good.shard_plain(10, 5, None)
    good.close()
    td = ivf.pq_transformed_points[0]
    pk = np.array(td.packed, copy=True)
    pk.view(np.uint8)[3] ^= 0x11
    ivf.pq_transformed_points[0] = TransformedData(td.size, pk)
    bad = DeviceIndex(ivf, owner, 0, 2)
    assert not bad.shard_plain(10, 5, None)          # no TWIN replay: the plain path stays "on request only"
    bad.close()


def test_default_build_at_full_size(oracle):
    """BASELINE configs[1]'s data set built the reference's DEFAULT way — IVF.build(n_probes=2), ivf.py:53: 2.37 M stored
    rows in 1087 lists — as bench.py's sweep measures it: ids of 2048 queries at n_probes 1 / 10 / 20 against the
    oracle, one batch at a time and pipelined + paired, both forms of the duplicate test; heap arrays of a sample."""
    import argparse, os, sys
    import torch
    from tinyknn_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    args = argparse.Namespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=2,
                              metric="angular", data="glove-like", fit_sample=100000,
                              cache_dir=os.environ.get("TMPDIR", "/tmp"), data_file=None)
    ivf, cent = bench.build_index(args, torch.device("cuda", 0))
    ox = bench.oracle_index(ivf)
    qs = bench.synth_queries(cent, 2048, 777, kind="glove-like")
    qn, qp = ivf._prepare(qs.copy())
    dev = ivf.device_index()
    assert dev.twin_table_width() == 1 and sum(len(x) for x in ivf.ids) == 2 * 1183514
    try:
        for n_probes in (1, 10, 20):
            want = ox.query_batch(qn, 10, n_probes)
            for twin, depth in ((1, 1), (1, 2), (0, 1)):
                dev.set_option(_lib.OPT_REPLAY_TWIN, twin)
                dev.set_pipeline(depth)
                dev.set_coalesce(2 if depth > 1 else 1)
                got = dev.query_batch(qn, qp, 10, n_probes)
                bad = np.flatnonzero((got != want).any(axis=1))
                assert bad.size == 0, (n_probes, twin, depth, bad[:5])
            dev.set_pipeline(1); dev.set_coalesce(1); dev.set_option(_lib.OPT_REPLAY_TWIN, 1)
            _, dbg = dev.query_batch(qn[:64], qp[:64], 10, n_probes, debug=True)
            for qi in range(0, 64, 7):
                _, odbg = ox.query(qn[qi], 10, n_probes=n_probes, debug=True)
                np.testing.assert_array_equal(dbg["heap_idx"][qi], odbg["heap_idx"])
                np.testing.assert_array_equal(dbg["heap_val"][qi], odbg["heap_val"])
    finally:
        dev.set_pipeline(1); dev.set_coalesce(1); dev.set_option(_lib.OPT_REPLAY_TWIN, 1)
