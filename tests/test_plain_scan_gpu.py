"""plain_scan.hip on the GPU: the probed lists behind the first ones scored as plain sums on the
int8 matrix cores.  Results must be IDENTICAL to the exact kernel's (and so to the reference's):
ids, probe lists and heap arrays (layout included), with the automatic rule, with every query
forced through the re-scan path, and with the path switched off."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

G6 = ["an100", "an100b2", "eu128", "an20", "eu20"]


@pytest.fixture(scope="module")
def tk():
    import tinyknn_amd
    from tinyknn_amd import _lib
    assert _lib.device_count() >= 1, "no GPU visible"
    return tinyknn_amd


@pytest.fixture()
def limit_hook(tk):
    """limit_hook(dev, v): cap every query's table limit on this index (tk_index_set_option)."""
    from tinyknn_amd import _lib
    return lambda dev, v: dev.set_option(_lib.OPT_PLAIN_LIMIT, int(v))


@pytest.mark.parametrize("tag", G6)
@pytest.mark.parametrize("mode", ["auto", "rescan-all", "off"])
@pytest.mark.parametrize("heap_mode", [0, 3])      # the lane replay / the wave-per-query register heap: both make the check
def test_golden_ids_and_heaps(tk, tag, mode, limit_hook, heap_mode):
    from test_hip_parity import ivf_from_fixture
    g = golden(f"g6_ivf_{tag}.npz")
    ivf = ivf_from_fixture(tk, g)
    dev = ivf.device_index()
    dev.set_heap_mode(heap_mode)
    dev.set_scan_mode(2)            # list-major: the form the plain kernel rides with
    dev.set_plain_scan("always" if mode != "off" else False)
    if mode == "rescan-all":
        limit_hook(dev, -128)       # no bound is <= -128 ... every query with a plain slot is redone
    plain_seen = 0
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        out, dbg = dev.query_batch(g["qn"], g["qpq"], 10, n_probes, debug=True)
        plain_seen = max(plain_seen, dev.plain_stats()["plain_units"])
        np.testing.assert_array_equal(dbg["probes"], g[f"probes_p{n_probes}"])
        np.testing.assert_array_equal(dbg["heap_idx"], g[f"heap_idx_p{n_probes}"])
        np.testing.assert_array_equal(dbg["heap_val"], g[f"heap_val_p{n_probes}"])
        np.testing.assert_array_equal(out, g[f"ids_p{n_probes}"])
        # small heaps make more slots plain (they are full after fewer rows)
        for pass_1 in (3, 17):
            a, da = dev.query_batch(g["qn"], g["qpq"], 10, n_probes, pass_1=pass_1, debug=True)
            plain_seen = max(plain_seen, dev.plain_stats()["plain_units"])
            dev.set_plain_scan(False)
            b, db = dev.query_batch(g["qn"], g["qpq"], 10, n_probes, pass_1=pass_1, debug=True)
            dev.set_plain_scan("always" if mode != "off" else False)
            np.testing.assert_array_equal(da["heap_idx"], db["heap_idx"])
            np.testing.assert_array_equal(da["heap_val"], db["heap_val"])
            np.testing.assert_array_equal(a, b)
    # the matrix-core kernel really ran on this fixture (the lists are ~45 rows: with the default heaps most
    # slots stay "head" / exact; the small heaps above leave rows to the plain kernel)
    assert mode == "off" or plain_seen > 0, (tag, mode)


def test_pipelined_batches_vs_oracle(tk, oracle):
    """60k x 100 angular, batches in flight (depth 2), plain on: rows of every batch equal the
    oracle's; a table with entries far below -128 in total (scaled-up query) is left to the exact
    kernel (TK_PLAIN_NEVER) and still answers identically."""
    import torch
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(10)
    n, d, nq = 60000, 100, 2000
    cent = np.random.randn(300, d)
    X = (cent[np.random.randint(300, size=n)] + 0.7 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(300, size=nq)] + 0.7 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 244, FastPQ(2))
    ivf.fit(X[:20000]).build(X, n_probes=1)
    L = len(ivf.active_centers)
    ox = oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                            ivf.pq_transformed_centers.packed,
                            [ivf.pq_transformed_points[i].packed for i in range(L)],
                            [ivf.pq_transformed_points[i].size for i in range(L)],
                            [ivf.ids[i] for i in range(L)], ivf.data)
    qn, qp = ivf._prepare(qs.copy())
    dev = ivf.device_index()
    dev.set_pipeline(2)
    dev.set_plain_scan("always")
    q_dev, qp_dev = torch.from_numpy(qn).cuda(), torch.from_numpy(np.ascontiguousarray(qp)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for n_probes in (2, 10, 20):
        want = ox.query_batch(qn, 10, n_probes)
        outs = [torch.full((nq, 10), -1, dtype=torch.int64, device="cuda") for _ in range(5)]
        for o in outs:
            dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), False, nq, 10, n_probes, o.data_ptr(), stream=st)
        dev.join(st)
        torch.cuda.synchronize()
        for o in outs:
            np.testing.assert_array_equal(o.cpu().numpy(), want)


def test_automatic_mode_pauses_on_data_without_structure(tk):
    """Mode 0 (default): one probe batch goes the plain way; on iid vectors far more than 1 % of its
    queries fail the lemma's condition, the path is paused (the batches behind the probe run on the
    exact kernel alone) and every batch — probe, waiting, paused — returns the rows of mode "off".
    On clustered rows the probe passes and the path stays on."""
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(3)
    n, d, nq = 40000, 64, 1500
    for structured in (False, True):
        if structured:
            cent = np.random.randn(200, d)
            X = (cent[np.random.randint(200, size=n)] + 0.4 * np.random.randn(n, d)).astype(np.float32)
            qs = (cent[np.random.randint(200, size=nq)] + 0.4 * np.random.randn(nq, d)).astype(np.float32)
        else:
            X = np.random.randn(n, d).astype(np.float32)
            qs = np.random.randn(nq, d).astype(np.float32)
        ivf = IVF("euclidean", 150, FastPQ(2))
        ivf.fit(X[:15000]).build(X, n_probes=1)
        qn, qp = ivf._prepare(qs.copy())
        dev = ivf.device_index()
        dev.set_scan_mode(2)
        dev.set_plain_scan(False)
        want = dev.query_batch(qn, qp, 10, 10)
        dev.set_plain_scan(True)                      # automatic: next batch is the probe
        assert dev.plain_stats()["state"] == "probe"
        got = dev.query_batch(qn, qp, 10, 10)         # (the host API waits for its batch)
        st = dev.plain_stats()
        np.testing.assert_array_equal(got, want)
        assert st["plain_units"] > 0
        if structured:
            assert st["flagged_queries"] * 100 <= nq and st["state"] == "on"
        else:
            assert st["flagged_queries"] * 100 > nq and st["state"] == "paused" and st["pause_left"] >= 256
        for _ in range(3):
            np.testing.assert_array_equal(dev.query_batch(qn, qp, 10, 10), want)
        st2 = dev.plain_stats()
        assert (st2["plain_units"] > 0) == structured
        assert st2["state"] == ("on" if structured else "paused")


@pytest.mark.parametrize("d,metric", [(12, "euclidean"), (40, "angular"), (64, "angular"), (72, "euclidean"), (104, "angular")])
def test_every_register_shape_of_the_plain_kernel(tk, d, metric):
    """M = d / 2 blocks: 6, 20, 32, 36, 52 -> the guarded forms for P <= 8 / 16 / 26 block pairs and the
    unguarded ones at P = 16 and P = 26, the latter two also with the table operand read from LDS.  Plain pinned on,
    list-major: heap arrays (layout included) and ids equal to the exact kernel's, with small and
    default heaps."""
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(d)
    n, nq, n_lists = 30000, 600, 60
    cent = np.random.randn(80, d)
    X = (cent[np.random.randint(80, size=n)] + 0.5 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(80, size=nq)] + 0.5 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF(metric, n_lists, FastPQ(2))
    ivf.fit(X[:10000]).build(X, n_probes=1)
    qn, qp = ivf._prepare(qs.copy())
    dev = ivf.device_index()
    dev.set_scan_mode(2)
    from tinyknn_amd import _lib
    for n_probes, pass_1 in ((4, None), (8, 7), (8, None)):
        dev.set_plain_scan(False)
        want, dw = dev.query_batch(qn, qp, 10, n_probes, pass_1=pass_1, debug=True)
        dev.set_plain_scan("always")
        got, dg = dev.query_batch(qn, qp, 10, n_probes, pass_1=pass_1, debug=True)
        st = dev.plain_stats()
        assert st["plain_units"] > 0 and st["plain_pairs"] > 0, st
        np.testing.assert_array_equal(dg["heap_idx"], dw["heap_idx"])
        np.testing.assert_array_equal(dg["heap_val"], dw["heap_val"])
        np.testing.assert_array_equal(got, want)


def test_automatic_mode_cycle_on_off_probe_on(tk):
    """Mode 0 end to end: probe -> on (clustered queries), a batch of queries far from every row
    (their bound never falls to the table's limit: > 1 % flagged) -> paused, the pause runs out ->
    probe -> on again.  Every batch returns the rows of mode "off"."""
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(21)
    n, d, nq = 30000, 64, 400
    cent = np.random.randn(150, d)
    X = (cent[np.random.randint(150, size=n)] + 0.4 * np.random.randn(n, d)).astype(np.float32)
    near = (cent[np.random.randint(150, size=nq)] + 0.4 * np.random.randn(nq, d)).astype(np.float32)
    far = (6.0 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("euclidean", 120, FastPQ(2))
    ivf.fit(X[:12000]).build(X, n_probes=1)
    dev = ivf.device_index()
    dev.set_scan_mode(2)
    qa, pa = ivf._prepare(near.copy())
    qb, pb = ivf._prepare(far.copy())
    dev.set_plain_scan(False)
    want_a, want_b = dev.query_batch(qa, pa, 10, 8), dev.query_batch(qb, pb, 10, 8)
    dev.set_plain_scan(True)
    assert dev.plain_stats()["state"] == "probe"
    np.testing.assert_array_equal(dev.query_batch(qa, pa, 10, 8), want_a)
    assert dev.plain_stats()["state"] == "on"
    np.testing.assert_array_equal(dev.query_batch(qb, pb, 10, 8), want_b)
    st = dev.plain_stats()
    assert st["state"] == "paused" and st["flagged_queries"] * 100 > nq and st["pause_left"] >= 256, st
    left = st["pause_left"]
    for i in range(left + 2):           # the pause runs out on small batches
        got = dev.query_batch(qa[:32], pa[:32], 10, 8)
        if i % 64 == 0:
            np.testing.assert_array_equal(got, want_a[:32])
        if dev.plain_stats()["state"] != "paused":
            break
    assert dev.plain_stats()["state"] in ("probe", "wait", "on"), dev.plain_stats()
    np.testing.assert_array_equal(dev.query_batch(qa, pa, 10, 8), want_a)
    np.testing.assert_array_equal(dev.query_batch(qa, pa, 10, 8), want_a)
    assert dev.plain_stats()["state"] == "on", dev.plain_stats()
