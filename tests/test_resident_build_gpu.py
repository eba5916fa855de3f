"""Device-resident build (tk_index_build_dev): vectors generated in HBM,
lists and codes built on the device.  Checked (a) against IVF.build on the same vectors brought
to the host — same active centres, same list memberships, same codes — and (b) end to end
against the CPU oracle fed with what the index exports: probe lists, heap arrays (layout
included) and final ids."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def synth_rows(n, d, seed, centres=None, sigma=1.0, row0=0):
    from tinyknn_amd import _lib
    out = np.zeros((n, d), dtype=np.float32)
    c = None if centres is None else np.ascontiguousarray(centres, dtype=np.float32)
    _lib.check(_lib.lib().tk_synth_rows(_lib.ptr(out, _lib._f32p), row0, n, d, seed,
                                        None if c is None else c.ctypes.data,
                                        0 if c is None else len(c), float(sigma)))
    return out


def oracle_from_resident(O, ivf):
    """OracleIndex over what a resident index exports; the rescoring vectors are fetched whole
    (small test sizes)."""
    dev = ivf.device_index()
    sizes, codes, ids = dev.export_lists()
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    L = len(sizes)
    data = dev.read_rows(np.arange(dev.N))
    return O.OracleIndex(ivf.pq.centers, ivf.pq.dims_per_block, ivf.pq.R, ivf.pq.sqrt_n_blocks,
                         ivf.active_centers, ivf.pq_transformed_centers.packed,
                         [codes[coff[i]:coff[i + 1]] for i in range(L)], list(sizes),
                         [ids[ioff[i]:ioff[i + 1]] for i in range(L)], data), (sizes, codes, ids)


def test_generator_is_a_function_of_seed_and_row():
    cent = np.random.RandomState(0).randn(7, 33).astype(np.float32)
    a = synth_rows(1000, 33, 5, cent, 0.7)
    b = np.concatenate([synth_rows(300, 33, 5, cent, 0.7), synth_rows(700, 33, 5, cent, 0.7, row0=300)])
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a, synth_rows(1000, 33, 6, cent, 0.7))
    z = synth_rows(200000, 8, 1)                      # plain N(0, 1)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.01


@pytest.mark.parametrize("metric", ["angular", "euclidean"])
def test_resident_build_equals_host_build(oracle, metric):
    """Unrotated PQ (d = 100 as GloVe: fast_pq.py:77 skips the rotation), N a multiple of 100:
    every arithmetic step is the one IVF.build(device=True) / numpy takes, so active centres,
    list memberships and codes must agree; only the order inside a list may differ."""
    from tinyknn_amd import IVF, FastPQ
    untransform_data = oracle.unpack
    n, d, nq, seed = 30000, 100, 300, 11
    cent = np.random.RandomState(1).randn(40, d).astype(np.float32)
    X = synth_rows(n, d, seed, cent, 0.7)
    host = IVF(metric, 60, FastPQ(2))
    host.fit(X[:8000])
    host.build(X, n_probes=1, device=True)
    res = IVF(metric, 60, FastPQ(2))
    res.all_centers, res.pq = host.all_centers, host.pq
    res.build_resident(n, d, seed, cent, 0.7)
    dev = res.device_index()
    np.testing.assert_array_equal(res.active_centers, host.active_centers)
    np.testing.assert_array_equal(res.pq_transformed_centers.packed, host.pq_transformed_centers.packed)
    np.testing.assert_array_equal(dev.read_rows(np.arange(n)), host.data)
    sizes, codes, ids = dev.export_lists()
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    for i in range(len(sizes)):
        hi = np.asarray(host.ids[i], dtype=np.int64)
        mine = ids[ioff[i]:ioff[i + 1]]
        np.testing.assert_array_equal(mine, np.sort(hi))            # ascending row order here
        hl = untransform_data(host.pq_transformed_points[i].packed)[:len(hi)]
        ml = untransform_data(codes[coff[i]:coff[i + 1]])
        np.testing.assert_array_equal(ml[:len(hi)], hl[np.argsort(hi, kind="stable")])
        # rows that pad the last chunk carry the zero vector's code, as the host build's do
        full = untransform_data(host.pq_transformed_points[i].packed)
        if len(full) > len(hi):
            np.testing.assert_array_equal(ml[len(hi):], full[len(hi):])
    # end to end vs the oracle on the exported index
    ox, _ = oracle_from_resident(oracle, res)
    qs = synth_rows(nq, d, seed + 1, cent, 0.7)
    qn, qp = res._prepare(qs.copy())
    for n_probes in (1, 5, 10):
        got, dbg = dev.query_batch(qn, qp, 10, n_probes, debug=True)
        np.testing.assert_array_equal(got, ox.query_batch(qn, 10, n_probes))
        for i in range(0, nq, 37):
            _, want = ox.query(qn[i], 10, n_probes, debug=True)
            np.testing.assert_array_equal(dbg["probes"][i], want["probes"])
            np.testing.assert_array_equal(dbg["heap_idx"][i], want["heap_idx"])
            np.testing.assert_array_equal(dbg["heap_val"][i], want["heap_val"])


def test_resident_build_rotated_vs_oracle(oracle):
    """Rotated PQ (128 -> 64 dims, float64 tables), euclidean, ragged list sizes: the index as
    built on the device answers exactly as the oracle does on the exported lists."""
    from tinyknn_amd import IVF, FastPQ
    n, d, nq, seed = 40037, 128, 200, 3
    cent = np.random.RandomState(2).randn(25, d).astype(np.float32)
    sample = synth_rows(6000, d, seed, cent, 0.9)
    ivf = IVF("euclidean", 50, FastPQ(2))
    ivf.fit(sample)
    assert ivf.pq.R is not None and ivf.pq.R.shape == (64, 128)
    ivf.build_resident(n, d, seed, cent, 0.9)
    dev = ivf.device_index()
    ox, (sizes, codes, ids) = oracle_from_resident(oracle, ivf)
    assert sizes.sum() == n and len(np.unique(ids)) == n
    qs = synth_rows(nq, d, seed + 1, cent, 0.9)
    qn, qp = ivf._prepare(qs.copy())
    for n_probes in (1, 4, 12):
        for depth in (1, 2):
            dev.set_pipeline(depth)
            np.testing.assert_array_equal(dev.query_batch(qn, qp, 10, n_probes), ox.query_batch(qn, 10, n_probes))
    dev.set_pipeline(1)
    np.testing.assert_array_equal(ivf.query_batch(qs, 10, n_probes=4), ox.query_batch(qn, 10, 4))


@pytest.mark.parametrize("n,n_clusters", [(500, 64), (100, 7), (1600, 3)])
def test_resident_build_small_and_sparse(oracle, n, n_clusters):
    """More centres than some lists can fill (inactive centres are dropped as ivf.py:91 does),
    lists of one row, N below one slab: same active centres and memberships as the host build,
    same answers as the oracle over the exported index."""
    from tinyknn_amd import IVF, FastPQ
    d, seed = 100, 21
    cent = np.random.RandomState(7).randn(5, d).astype(np.float32) * 3
    X = synth_rows(n, d, seed, cent, 0.5)
    host = IVF("euclidean", n_clusters, FastPQ(2))
    host.fit(np.concatenate([X, synth_rows(max(0, 400 - n), d, seed + 5, cent, 0.5)]))
    # centres nobody is nearest to: far away from all the data
    host.all_centers = np.concatenate([host.all_centers, np.full((5, d), 1e3, host.all_centers.dtype)])
    host.n_clusters += 5
    host.pq_transformed_points = [None] * host.n_clusters
    host.ids = [None] * host.n_clusters
    host.build(X, n_probes=1, device=True)
    res = IVF("euclidean", host.n_clusters, FastPQ(2))
    res.all_centers, res.pq = host.all_centers, host.pq
    res.build_resident(n, d, seed, cent, 0.5)
    assert len(res.active_centers) == len(host.active_centers) < host.n_clusters
    np.testing.assert_array_equal(res.active_centers, host.active_centers)
    sizes, codes, ids = res.device_index().export_lists()
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    for i in range(len(sizes)):
        np.testing.assert_array_equal(ids[ioff[i]:ioff[i + 1]], np.sort(np.asarray(host.ids[i], np.int64)))
    ox, _ = oracle_from_resident(oracle, res)
    qs = synth_rows(50, d, seed + 1, cent, 0.5)
    qn, qp = res._prepare(qs.copy())
    for n_probes in (1, 3, 100):
        np.testing.assert_array_equal(res.device_index().query_batch(qn, qp, 10, n_probes),
                                      ox.query_batch(qn, 10, n_probes))


@pytest.mark.parametrize("metric", ["angular", "euclidean"])
def test_resident_build_two_lists_per_row(oracle, metric):
    """IVF.build(n_probes=2), the reference's default: every row in its two nearest lists, a
    list's column-0 members before its column-1 members (utils.py:131-150), labels repeat and
    the replay runs its duplicate test.  Same memberships and codes as the host build; same
    answers as the oracle over the exported lists."""
    from tinyknn_amd import IVF, FastPQ
    unpack = oracle.unpack
    n, d, nq, seed = 20000, 100, 250, 31
    cent = np.random.RandomState(3).randn(30, d).astype(np.float32)
    X = synth_rows(n, d, seed, cent, 0.7)
    host = IVF(metric, 48, FastPQ(2))
    host.fit(X[:6000])
    host.build(X, n_probes=2, device=True)
    res = IVF(metric, 48, FastPQ(2))
    res.all_centers, res.pq = host.all_centers, host.pq
    res.build_resident(n, d, seed, cent, 0.7, n_probes=2)
    np.testing.assert_array_equal(res.active_centers, host.active_centers)
    dev = res.device_index()
    sizes, codes, ids = dev.export_lists()
    assert sizes.sum() == 2 * n
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    for i in range(len(sizes)):
        hi = np.asarray(host.ids[i], dtype=np.int64)
        mine = ids[ioff[i]:ioff[i + 1]]
        np.testing.assert_array_equal(np.sort(mine), np.sort(hi))
        # two ascending runs: the list's column-0 members, then its column-1 members
        brk = np.flatnonzero(np.diff(mine) < 0)
        assert len(brk) <= 1
        hl = unpack(host.pq_transformed_points[i].packed)[:len(hi)]
        ml = unpack(codes[coff[i]:coff[i + 1]])[:len(hi)]
        np.testing.assert_array_equal(ml[np.argsort(mine, kind="stable")], hl[np.argsort(hi, kind="stable")])
    ox, _ = oracle_from_resident(oracle, res)
    qs = synth_rows(nq, d, seed + 1, cent, 0.7)
    qn, qp = res._prepare(qs.copy())
    for n_probes in (1, 5, 12):
        got, dbg = dev.query_batch(qn, qp, 10, n_probes, debug=True)
        np.testing.assert_array_equal(got, ox.query_batch(qn, 10, n_probes))
        for i in range(0, nq, 41):
            _, want = ox.query(qn[i], 10, n_probes, debug=True)
            np.testing.assert_array_equal(dbg["heap_idx"][i], want["heap_idx"])
            np.testing.assert_array_equal(dbg["heap_val"][i], want["heap_val"])


def test_reduced_c5_lists_far_longer_than_the_heap(oracle, tmp_path):
    """BASELINE configs[4] scaled to one test: 6 M x 128 generated and built in HBM, 1 500 lists of
    ~4 000 rows (36 heaps' worth each), PQ rotated to 64 dims (M = 32, float64 table math),
    n_probes 10: probe lists, heap arrays (layout included) and final ids against the CPU oracle
    fed with the exported lists; the oracle's vector file is sparse (only the candidates' rows
    travel).  One batch in flight AND batches in flight (the plain-sum kernel, head mode, the
    pipelined streams) must both reproduce it."""
    import torch
    from tinyknn_amd import IVF, FastPQ
    n, d, nq, seed, n_lists = 6_000_000, 128, 2000, 17, 1500
    cent = np.random.RandomState(2).randn(400, d).astype(np.float32)
    ivf = IVF("euclidean", n_lists, FastPQ(2))
    sample = synth_rows(200_000, d, seed, cent, 0.7)
    # coarse centres = rows of the data set itself (the sample is its first rows): every centre is
    # then the nearest centre of at least one row — tk_index_build_dev, like the reference's
    # group_data_by_indices (utils.py:128), refuses an empty list in front of a used one
    rng = np.random.RandomState(3)
    ivf.all_centers = sample[rng.choice(len(sample), n_lists, replace=False)].copy()
    np.random.seed(5)
    ivf.pq.fit(sample[:20000])
    ivf.build_resident(n, d, seed, cent, 0.7)
    dev = ivf.device_index()
    dev.set_plain_scan("always")            # (mode 0 would decide per batch: tests/test_plain_scan_gpu.py)
    sizes, codes, ids = dev.export_lists()
    assert sizes.min() > 0 and np.median(sizes) > 10 * 111
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    L = len(sizes)
    data = np.memmap(str(tmp_path / "rows.f32"), dtype=np.float32, mode="w+", shape=(n, d))
    ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                            ivf.pq_transformed_centers.packed, [codes[coff[i]:coff[i + 1]] for i in range(L)],
                            list(sizes), [ids[ioff[i]:ioff[i + 1]] for i in range(L)], data)
    assert ox.data is data or np.may_share_memory(ox.data, data)
    qs = synth_rows(nq, d, seed + 1, cent, 0.7)
    qn, qp = ivf._prepare(qs.copy())
    for n_probes in (10, 3):
        out, dbg = dev.query_batch(qn, qp, 10, n_probes, debug=True)        # one batch in flight
        rows = np.unique(dbg["heap_idx"][dbg["heap_idx"] >= 0])
        data[rows] = dev.read_rows(rows)
        want = ox.query_batch(qn, 10, n_probes)
        np.testing.assert_array_equal(out, want)
        for i in range(0, nq, 37):
            _, w = ox.query(qn[i], 10, n_probes, debug=True)
            np.testing.assert_array_equal(w["probes"], dbg["probes"][i])
            np.testing.assert_array_equal(w["heap_idx"], dbg["heap_idx"][i])
            np.testing.assert_array_equal(w["heap_val"], dbg["heap_val"][i])
        st = dev.plain_stats()
        # (n_probes = 3: 4 pairs per list — such a batch goes to the query-major kernel, not the plain one)
        assert n_probes != 10 or (st["plain_units"] > 0 and st["head_pair_records"] >= nq // 2), st
        # batches in flight
        dev.set_pipeline(2)
        q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(np.ascontiguousarray(qp)).cuda()
        s_ = torch.cuda.current_stream().cuda_stream
        outs = [torch.full((nq, 10), -1, dtype=torch.int64, device="cuda") for _ in range(4)]
        for o in outs:
            dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), qp.dtype != np.float32, nq, 10, n_probes,
                                o.data_ptr(), stream=s_)
        dev.join(s_)
        torch.cuda.synchronize()
        for o in outs:
            np.testing.assert_array_equal(o.cpu().numpy(), want)
        dev.set_pipeline(1)
    dev.close()
