"""GPU parity tests: the HIP path, called through the C ABI, against the golden
vectors of the compiled reference and against the CPU oracle on seeded inputs.
Bit-exact for every integer artefact (G1-G6); 1e-4 for dequantised floats (G7)."""
import numpy as np
import pytest

from conftest import golden, split_lists, G6_TAGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tk():
    import tinyknn_amd
    from tinyknn_amd import _lib
    assert _lib.device_count() >= 1, "no GPU visible"
    return tinyknn_amd


def _est(tk, name):
    from tinyknn_amd._fast_pq import estimate_pq_sse
    from tinyknn_amd._fast_pq_avx import estimate_pq_avx
    return estimate_pq_sse if name == "sse" else estimate_pq_avx


def _qry(tk, name):
    from tinyknn_amd._fast_pq import query_pq_sse
    from tinyknn_amd._fast_pq_avx import query_pq_avx
    return query_pq_sse if name == "sse" else query_pq_avx


# ---- G3 --------------------------------------------------------------------

def test_estimate_golden(tk):
    from tinyknn_amd._transform import transform_data, transform_tables
    g = golden("g3_estimate.npz")
    for ci in range(5):
        d = transform_data(g[f"codes_{ci}"])
        t = transform_tables(g[f"table_{ci}"])
        for signed in (1, 0):
            for name in ("sse", "avx"):
                out = np.zeros(2 * len(d), dtype=np.uint64)
                _est(tk, name)(d, t, out, bool(signed))
                np.testing.assert_array_equal(out.view(np.uint8), g[f"out_{ci}_{signed}_{name}"],
                                              err_msg=f"case {ci} signed={signed} {name}")
    out = np.zeros(2, dtype=np.uint64)
    _est(tk, "sse")(transform_data(g["kat_codes"]), transform_tables(g["kat_table"]), out, False)
    assert out.view(np.uint8)[0] == 26 and not out.view(np.uint8)[1:].any()


@pytest.mark.parametrize("n,M", [(16, 2), (16, 4), (48, 6), (1024, 32), (1040, 52), (16 * 4097, 52),
                                 (16 * 1000, 32), (16 * 333, 104)])
def test_estimate_vs_oracle(tk, oracle, n, M):
    rng = np.random.default_rng(n * 131 + M)
    codes = rng.integers(0, 16, size=(n, M)).astype(np.uint8)
    d = oracle.transform_data(codes)
    for tab in (rng.integers(0, 256, size=(M, 16)).astype(np.uint8),          # saturating
                rng.integers(-4, 24, size=(M, 16)).astype(np.int8).view(np.uint8)):  # realistic
        t = oracle.transform_tables(tab)
        for signed in (True, False):
            for name, order in (("sse", oracle.ORDER_SSE), ("avx", oracle.ORDER_AVX)):
                exp = np.zeros(2 * len(d), dtype=np.uint64)
                oracle.estimate_pq(d, t, exp, signed, order)
                out = np.zeros(2 * len(d), dtype=np.uint64)
                _est(tk, name)(d, t, out, signed)
                np.testing.assert_array_equal(out, exp, err_msg=f"{n} {M} {signed} {name}")


def test_estimate_linearity_full_size(tk):
    """Size-independent property at the C1/C3 scale: with a table that is zero
    except for one block, the estimate is that block's lookup (no saturation)."""
    from tinyknn_amd._transform import transform_data, transform_tables
    rng = np.random.default_rng(7)
    n, M = 16 * 62500, 32          # 1M codes, SIFT-like
    codes = rng.integers(0, 16, size=(n, M)).astype(np.uint8)
    d = transform_data(codes)
    for m in (0, 13, 31):
        tab = np.zeros((M, 16), dtype=np.uint8)
        tab[m] = rng.integers(-100, 100, size=16).astype(np.int8).view(np.uint8)
        out = np.zeros(2 * len(d), dtype=np.uint64)
        _est(tk, "avx")(d, transform_tables(tab), out, True)
        np.testing.assert_array_equal(out.view(np.uint8), tab[m][codes[:, m]])


# ---- G4 --------------------------------------------------------------------

def test_query_pq_golden(tk):
    from tinyknn_amd._transform import transform_data, transform_tables
    from tinyknn_amd._fast_pq import init_heap
    g = golden("g4_query.npz")
    checked = 0
    for ci, R, n, M, signed, use_labels in g["meta"]:
        d1, d2 = transform_data(g[f"codes1_{ci}"]), transform_data(g[f"codes2_{ci}"])
        t = transform_tables(g[f"table_{ci}"])
        l1 = g[f"labels1_{ci}"] if use_labels else None
        l2 = g[f"labels2_{ci}"] if use_labels else None
        for name in ("sse", "avx"):
            if f"idx_{ci}_{name}" not in g:
                continue
            idx = np.zeros(R, np.int64)
            val = np.zeros(R, np.int32)
            init_heap(idx, val, bool(signed))
            for dd, ll in ((d1, l1), (d2, l2), (d1, l1)):
                _qry(tk, name)(dd, int(n), t, idx, val, bool(signed), ll)
            np.testing.assert_array_equal(idx, g[f"idx_{ci}_{name}"], err_msg=f"case {ci} {name}")
            np.testing.assert_array_equal(val, g[f"val_{ci}_{name}"], err_msg=f"case {ci} {name}")
            checked += 1
    assert checked >= 50


@pytest.mark.parametrize("seed", range(6))
def test_query_pq_vs_oracle_random(tk, oracle, seed):
    """Heavy ties, evictions, stale-bound inserts, duplicate labels, chained lists,
    n not a multiple of 16, heaps that are NOT valid max-heaps on entry."""
    from tinyknn_amd._fast_pq import init_heap
    rng = np.random.default_rng(1000 + seed)
    for trial in range(12):
        M = int(rng.choice([4, 8, 32, 52]))
        n = int(rng.choice([1, 15, 16, 17, 100, 1000, 5000, 70000]))
        R = int(rng.choice([1, 2, 3, 21, 30, 111, 511, 1000, 1400, 5000]))
        signed = bool(rng.integers(0, 2))
        npad = n + (-n) % 16
        codes = rng.integers(0, 16, size=(npad, M)).astype(np.uint8)
        tab = (rng.integers(-6, 12, size=(M, 16)).astype(np.int8).view(np.uint8) if signed
               else rng.integers(0, 8, size=(M, 16)).astype(np.uint8))
        d, t = oracle.transform_data(codes), oracle.transform_tables(tab)
        labels = None
        if rng.integers(0, 2):
            labels = rng.integers(0, max(2, n // 2), size=npad).astype(np.int64) + 10**12
        idx = np.zeros(R, np.int64); val = np.zeros(R, np.int32)
        init_heap(idx, val, signed)
        if trial % 4 == 3:      # arbitrary caller-provided arrays (ivf.py:137-138 uses np.full)
            val[:] = rng.integers(-200, 300, size=R)
            idx[:] = rng.integers(-1, 5, size=R)
        e_idx, e_val = idx.copy(), val.copy()
        for rep in range(2):
            _qry(tk, "avx")(d, n, t, idx, val, signed, labels)
            oracle.query_pq(d, n, t, e_idx, e_val, signed, labels, oracle.ORDER_AVX)
        np.testing.assert_array_equal(idx, e_idx, err_msg=f"seed {seed} trial {trial}")
        np.testing.assert_array_equal(val, e_val, err_msg=f"seed {seed} trial {trial}")


def test_topk_multiset(tk):
    """reference tests/test_pq.py:85-140: heap of size n holds exactly the
    estimates below the maximum."""
    from tinyknn_amd import FastPQ
    from tinyknn_amd._fast_pq import init_heap, estimate_pq_sse, query_pq_sse
    np.random.seed(10)
    for n in (1, 3, 9, 20, 50):
        for signed in (True, False):
            X = np.random.randn(n, 11).astype(np.float32)
            pq = FastPQ(dims_per_block=2)
            _, data = pq.fit_transform(X)
            for q in np.random.randn(3, 11).astype(np.float32):
                dt = pq.distance_table(q)
                out = np.zeros(2 * len(data), dtype=np.uint64)
                estimate_pq_sse(data, dt.tables, out, signed)
                est = out.view(np.int8 if signed else np.uint8)[:n]
                idx = np.zeros(n, np.int64); val = np.zeros(n, np.int32)
                init_heap(idx, val, signed)
                query_pq_sse(data, n, dt.tables, idx, val, signed)
                maxv = 127 if signed else 255
                got = np.sort(val[val < maxv])
                exp = np.sort(est)[np.sort(est) < maxv]
                np.testing.assert_array_equal(got, exp)


def test_large_labels(tk):
    # reference tests/test_pq.py:143-158
    from tinyknn_amd import FastPQ
    from tinyknn_amd._fast_pq import init_heap, query_pq_sse
    np.random.seed(10)
    n, d, k = 100, 10, 100
    X = np.random.randn(n, d).astype(np.float32)
    q = np.random.randn(d).astype(np.float32)
    pq = FastPQ(2)
    _, data = pq.fit_transform(X)
    dtable = pq.distance_table(q)
    indices = np.empty((k,), dtype=np.int64); values = np.empty((k,), dtype=np.int32)
    labels = np.arange(n, dtype=np.int64) + 10**12
    init_heap(indices, values, True)
    query_pq_sse(data, n, dtable.tables, indices, values, True, labels)
    indices.sort()
    np.testing.assert_array_equal(indices, labels)


def test_heap_primitives_golden(tk):
    from tinyknn_amd._fast_pq import init_heap, insert, insert_is
    g = golden("g4_query.npz")
    idx = np.empty(3, np.int64); val = np.empty(3, np.int32)
    init_heap(idx, val, True)
    np.testing.assert_array_equal(idx, [-1, -1, -1]); np.testing.assert_array_equal(val, [127] * 3)
    init_heap(idx, val, False)
    np.testing.assert_array_equal(val, [255] * 3)
    idx = np.empty(2, np.int64); val = np.empty(2, np.int32)
    init_heap(idx, val, True)
    insert(idx, val, 1, 10); insert(idx, val, 1, 10)
    np.testing.assert_array_equal(idx, [-1, 1]); np.testing.assert_array_equal(val, [127, 10])
    idx = np.empty(13, np.int64); val = np.empty(13, np.int32)
    init_heap(idx, val, True)
    idx2, val2 = idx.copy(), val.copy()
    for t, (lab, v) in enumerate(g["heap_ops"][:150]):
        insert(idx, val, lab, v); insert_is(idx2, val2, lab, v)
        np.testing.assert_array_equal(idx, g["heap_trace_idx"][t])
        np.testing.assert_array_equal(val, g["heap_trace_val"][t])
        np.testing.assert_array_equal(idx2, g["heap_is_trace_idx"][t])
        np.testing.assert_array_equal(val2, g["heap_is_trace_val"][t])


# ---- G5 / G7 ---------------------------------------------------------------

class _PQ:
    pass


def test_distance_tables_golden(tk):
    from tinyknn_amd.fast_pq import build_tables, _FastDistanceTable, TransformedData
    g = golden("g5_tables.npz")
    for ci, d, dpb, n, size, rotated, f_order in g["meta"]:
        pq = _PQ()
        pq.centers = np.asfortranarray(g[f"centers_{ci}"]) if f_order else g[f"centers_{ci}"]
        pq.dims_per_block = int(dpb)
        pq.sqrt_n_blocks = float(g[f"sqrt_n_blocks_{ci}"])
        qpq = g[f"qpq_{ci}"]
        for signed, tn, sn, cn in ((True, "tables", "shift", "scale"), (False, "utables", "ushift", "uscale")):
            tables, shift, scale = build_tables(pq, qpq, signed)
            np.testing.assert_array_equal(tables, g[f"{tn}_{ci}"], err_msg=f"case {ci} signed={signed}")
            np.testing.assert_array_equal(shift, g[f"{sn}_{ci}"])
            assert shift.dtype == g[f"{sn}_{ci}"].dtype
            np.testing.assert_array_equal(scale, g[f"{cn}_{ci}"])
        tables, shift, scale = build_tables(pq, qpq, True)
        td = TransformedData(int(size), g[f"packed_{ci}"])
        for qi in range(len(qpq)):
            dt = _FastDistanceTable(qpq[qi], None, tables[qi], shift[qi], scale[qi], True)
            resc = dt.estimate_distances(td, rescale=True)
            np.testing.assert_allclose(resc, g[f"est_rescaled_{ci}"][qi], rtol=0, atol=1e-4)


@pytest.mark.parametrize("M,dpb,is64", [(52, 2, False), (32, 2, True), (32, 2, False), (100, 1, False),
                                        (6, 4, True), (200, 2, False), (12, 8, False)])
def test_distance_tables_vs_oracle(tk, oracle, M, dpb, is64):
    from tinyknn_amd.fast_pq import build_tables
    rng = np.random.default_rng(M * 7 + dpb)
    dq = M * dpb
    pq = _PQ()
    pq.centers = rng.standard_normal((16, dq)).astype(np.float32)
    if dpb == 1:
        pq.centers = np.asfortranarray(pq.centers)
    pq.dims_per_block = dpb
    pq.sqrt_n_blocks = np.sqrt(M)
    qs = rng.standard_normal((200, dq)).astype(np.float64 if is64 else np.float32)
    qs[:50] *= 0.05
    for signed in (True, False):
        tables, shift, scale = build_tables(pq, qs, signed)
        for qi in range(len(qs)):
            t, sh, sc = oracle.distance_table(pq.centers, dpb, qs[qi], pq.sqrt_n_blocks, signed)
            np.testing.assert_array_equal(tables[qi], oracle.transform_tables(t), err_msg=f"q{qi} {signed}")
            assert sh == shift[qi] and sc == scale[qi], (qi, signed)


# ---- rescoring ---------------------------------------------------------------

@pytest.mark.parametrize("n,d,k", [(30, 100, 10), (111, 100, 10), (511, 100, 10), (211, 128, 10),
                                   (50, 20, 10), (12, 10, 3), (5, 7, 10), (1101, 100, 100), (64, 3, 1)])
def test_knn_brute1_vs_oracle(tk, oracle, n, d, k):
    from tinyknn_amd.utils import knn_brute1
    rng = np.random.default_rng(n + d)
    for t in range(10):
        Y = rng.standard_normal((n, d)).astype(np.float32)
        x = rng.standard_normal(d).astype(np.float32)
        if t % 3 == 0 and n > 8:
            Y[3] = Y[7]
        np.testing.assert_array_equal(knn_brute1(x, Y, k), oracle.knn_brute1(x, Y, k))


@pytest.mark.parametrize("xd,yd", [(np.float32, np.float64), (np.float64, np.float32),
                                    (np.float64, np.float64)])
def test_knn_brute1_float64_vs_oracle(tk, oracle, xd, yd):
    """numpy promotes `Y - x`: any float64 operand makes the rescoring float64."""
    from tinyknn_amd.utils import knn_brute1
    rng = np.random.default_rng(11)
    for n, d, k in [(30, 100, 10), (111, 100, 10), (211, 128, 10), (50, 20, 10), (40, 7, 5)]:
        for t in range(6):
            Y = rng.standard_normal((n, d)).astype(yd)
            x = rng.standard_normal(d).astype(xd)
            np.testing.assert_array_equal(knn_brute1(x, Y, k), oracle.knn_brute1(x, Y, k))


# ---- G6 --------------------------------------------------------------------

class _State:
    """An IVF-shaped object filled from a fixture (what fit+build leave behind)."""


def ivf_from_fixture(tk, g):
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.fast_pq import TransformedData
    codes, ids = split_lists(g)
    pq = FastPQ(2)
    pq.centers = g["pq_centers"]
    pq.sqrt_n_blocks = float(g["sqrt_n_blocks"])
    pq.R = g["R"] if "R" in g else None
    ivf = IVF(str(g["metric"]), len(codes), None)
    ivf.pq = pq
    ivf.active_centers = g["active_centers"]
    ivf.pq_transformed_centers = TransformedData(int(g["center_size"]), g["center_codes"])
    ivf.pq_transformed_points = [TransformedData(int(s), c) for s, c in zip(g["list_sizes"], codes)]
    ivf.ids = ids
    ivf.data = g["data"]
    return ivf


@pytest.mark.parametrize("tag", G6_TAGS)
def test_ivf_query_golden(tk, tag):
    g = golden(f"g6_ivf_{tag}.npz")
    ivf = ivf_from_fixture(tk, g)
    k = 10
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        # batch path, from the reference's own normalised queries
        qp = ivf._prepare(np.array(g["qn"], copy=True)) if False else None
        dev = ivf.device_index()
        dev.set_scan_mode(2)
        out2 = dev.query_batch(g["qn"], g["qpq"], k, n_probes)
        np.testing.assert_array_equal(out2, g[f"ids_p{n_probes}"])
        # every form of the rescoring kernel (lane per row / rows staged through LDS in tiles of 64 /
        # of 32: the default) sums a row in numpy's order: the same ids
        from tinyknn_amd import _lib
        try:
            for form in (0, 1, 2):
                dev.set_option(_lib.OPT_RESCORE_FORM, form)
                np.testing.assert_array_equal(dev.query_batch(g["qn"], g["qpq"], k, n_probes),
                                              g[f"ids_p{n_probes}"])
        finally:
            dev.set_option(_lib.OPT_RESCORE_FORM, 2)
        dev.set_scan_mode(0)
        # the lazy lane replay (blocks fetched only where their minimum passes: the form long lists get by
        # themselves) and the staged one are the same replay: the reference's heap arrays, layout included
        # ... and so are, for labels that repeat (an100b2), the two forms of `insert`'s duplicate test: decided from
        # the positions of a row's other copies (TWIN, the default) and by the hash set of the labels in the heap
        for lazy, twin in ((1, 1), (0, 1), (0, 0)):
            dev.set_option(_lib.OPT_REPLAY_LAZY, lazy)
            dev.set_option(_lib.OPT_REPLAY_TWIN, twin)
            out, dbg = dev.query_batch(g["qn"], g["qpq"], k, n_probes, debug=True)
            np.testing.assert_array_equal(dbg["probes"], g[f"probes_p{n_probes}"])
            np.testing.assert_array_equal(dbg["heap_idx"], g[f"heap_idx_p{n_probes}"])
            np.testing.assert_array_equal(dbg["heap_val"], g[f"heap_val_p{n_probes}"])
            np.testing.assert_array_equal(out, g[f"ids_p{n_probes}"])
        dev.set_option(_lib.OPT_REPLAY_LAZY, -1)
        dev.set_option(_lib.OPT_REPLAY_TWIN, 1)
        # the wave-per-query replay with the heap in registers (heaps of up to 129 entries; the kernel of one query per
        # call): forced, and as small batches get it by themselves at the product's default threshold
        for mode, pair_nq in ((3, 0), (0, 8192)):
            dev.set_heap_mode(mode)
            dev.set_option(_lib.OPT_PAIR_NQ, pair_nq)
            out, dbg = dev.query_batch(g["qn"], g["qpq"], k, n_probes, debug=True)
            np.testing.assert_array_equal(dbg["probes"], g[f"probes_p{n_probes}"])
            np.testing.assert_array_equal(dbg["heap_idx"], g[f"heap_idx_p{n_probes}"], err_msg=f"pair {mode} {pair_nq}")
            np.testing.assert_array_equal(dbg["heap_val"], g[f"heap_val_p{n_probes}"])
            np.testing.assert_array_equal(out, g[f"ids_p{n_probes}"])
        dev.set_heap_mode(0)
        dev.set_option(_lib.OPT_PAIR_NQ, 4)
        out, dbg = dev.query_batch(g["qn"], g["qpq"], k, n_probes, debug=True)
        np.testing.assert_array_equal(dbg["probes"], g[f"probes_p{n_probes}"])
        np.testing.assert_array_equal(dbg["heap_idx"], g[f"heap_idx_p{n_probes}"])
        np.testing.assert_array_equal(dbg["heap_val"], g[f"heap_val_p{n_probes}"])
        np.testing.assert_array_equal(out, g[f"ids_p{n_probes}"])
        # public single-query path from the RAW queries (host normalisation + rotation)
        for qi in range(0, len(g["qs"]), 5):
            ids = ivf.query(g["qs"][qi].copy(), k, n_probes=n_probes)
            exp = g[f"ids_p{n_probes}"][qi]
            np.testing.assert_array_equal(ids, exp[exp != -1])
        batch = ivf.query_batch(g["qs"], k, n_probes=n_probes)
        np.testing.assert_array_equal(batch, g[f"ids_p{n_probes}"])


def test_ivf_vs_oracle_larger(tk, oracle):
    """A 60k x 100 angular index built by the product's host code, 400 queries,
    n_probes in {1, 5, 10, 20}: ids, probe order and heap arrays vs the oracle."""
    from tinyknn_amd import IVF, FastPQ
    from test_oracle_golden import load_oracle_index  # noqa
    np.random.seed(10)
    n, d, nq = 60000, 100, 400
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
    for n_probes, heap_mode, scan_mode in ((1, 0, 2), (5, 0, 1), (10, 0, 2), (10, 1, 1), (10, 2, 2), (20, 0, 0),
                                           (50, 0, 2), (50, 1, 1), (50, 2, 0), (1, 3, 0), (5, 3, 2), (10, 3, 1),
                                           (11, 3, 0), (20, 3, 0), (50, 3, 2)):      # (3: the register heap: 2 / 4 / 8 nodes per lane)
        ivf.device_index().set_heap_mode(heap_mode)   # lane-per-query / wave-per-query replay
        ivf.device_index().set_scan_mode(scan_mode)   # query-major / list-major scan
        out, dbg = ivf.device_index().query_batch(qn, qp, 10, n_probes, debug=True)
        for qi in range(nq):
            ids, odbg = ox.query(qn[qi], 10, n_probes=n_probes, debug=True)
            np.testing.assert_array_equal(dbg["probes"][qi], odbg["probes"], err_msg=f"q{qi}")
            np.testing.assert_array_equal(dbg["heap_idx"][qi], odbg["heap_idx"], err_msg=f"q{qi}")
            np.testing.assert_array_equal(dbg["heap_val"][qi], odbg["heap_val"], err_msg=f"q{qi}")
            np.testing.assert_array_equal(out[qi][out[qi] != -1] if len(ids) < 10 else out[qi], ids)


def _oracle_index(oracle, ivf):
    L = len(ivf.active_centers)
    return oracle.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                              ivf.pq_transformed_centers.packed,
                              [ivf.pq_transformed_points[i].packed for i in range(L)],
                              [ivf.pq_transformed_points[i].size for i in range(L)],
                              [ivf.ids[i] for i in range(L)], ivf.data)


@pytest.mark.parametrize("build_probes", [1, 2, 3])
def test_ivf_saturating_and_duplicates(tk, oracle, build_probes):
    """Far-apart clusters: int8 sums hit both rails, coarse heaps keep -1 sentinels
    (which wrap to the last list, fast_pq.py:311 / ivf.py:141), and with
    build_probes=2 / 3 every point sits in two / three lists (dedupe across lists: the lane
    replay's TWIN form by default, the queries whose probe list wraps by the packed kernel)."""
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(3)
    n, d, nq = 6000, 20, 200
    cent = np.random.randn(12, d) * 8
    X = (cent[np.random.randint(12, size=n)] + 0.5 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(12, size=nq)] + 0.5 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("euclidean", 40, FastPQ(2))
    ivf.fit(X).build(X, n_probes=build_probes)
    ox = _oracle_index(oracle, ivf)
    qn, qp = ivf._prepare(qs.copy())
    saw_sentinel = 0
    for n_probes in (3, 8, 20):
        for heap_mode in (0, 1, 2, 3, 4):      # (4: the register heap with label64 entries instead of label24)
            from tinyknn_amd import _lib
            ivf.device_index().set_option(_lib.OPT_LABELS24, 0 if heap_mode == 4 else 1)
            heap_mode = min(heap_mode, 3)
            ivf.device_index().set_heap_mode(heap_mode)
            ivf.device_index().set_scan_mode(1 + heap_mode % 2)
            out, dbg = ivf.device_index().query_batch(qn, qp, 10, n_probes, debug=True)
            for qi in range(nq):
                ids, odbg = ox.query(qn[qi], 10, n_probes=n_probes, debug=True)
                saw_sentinel += int((odbg["probes"] < 0).any())
                np.testing.assert_array_equal(dbg["probes"][qi], odbg["probes"], err_msg=f"q{qi}")
                np.testing.assert_array_equal(dbg["heap_idx"][qi], odbg["heap_idx"], err_msg=f"q{qi} p{n_probes} m{heap_mode}")
                np.testing.assert_array_equal(dbg["heap_val"][qi], odbg["heap_val"], err_msg=f"q{qi}")
                np.testing.assert_array_equal(out[qi][out[qi] != -1] if len(ids) < 10 else out[qi], ids)
    print("queries with a -1 probe:", saw_sentinel)


def test_config_c1_flat_scan(tk, oracle):
    """BASELINE configs[0] (examples/example.py): N=16000 d=128 random, FastPQ(2)
    => rotated to 64 dims, M=32, float64 table math.  Per query: distance_table +
    estimate_distances, and the two-pass top(); all against the oracle."""
    from tinyknn_amd import FastPQ
    np.random.seed(10)
    n, d = 16000, 128
    X = np.random.randn(n, d).astype(np.float32)
    qs = np.random.randn(40, d).astype(np.float32)
    pq = FastPQ(dims_per_block=2, use_kmeans=True)
    pq.fit(X[:4000])
    data = pq.transform(X)
    assert pq.R is not None and pq.centers.shape == (16, 64) and data.packed.shape == (1000, 32)
    for signed in (True, False):
        for q in qs[:20]:
            dt = pq.distance_table(q) if signed else pq.udistance_table(q)
            qp = np.concatenate([q, np.zeros(0, np.float32)]) @ pq.R.T
            table, shift, scale = oracle.distance_table(pq.centers, 2, qp, pq.sqrt_n_blocks, signed)
            np.testing.assert_array_equal(dt.tables, oracle.transform_tables(table))
            assert dt.mean == shift and dt.scale == scale and dt.mean.dtype == np.float64
            est = dt.estimate_distances(data)
            exp = np.zeros(2 * len(data.packed), dtype=np.uint64)
            oracle.estimate_pq(data.packed, dt.tables, exp, signed, oracle.ORDER_AVX)
            np.testing.assert_array_equal(est, exp.view(np.int8 if signed else np.uint8)[:n])
    # the batched form (one launch for all queries); default: the live host buffer, like the
    # reference kernels; opt-in: codes resident in HBM between calls
    from tinyknn_amd.fast_pq import estimate_batch
    from tinyknn_amd import _fast_pq
    assert _fast_pq.cache_device_codes is False and _fast_pq.device_codes(data.packed) is None
    dt0 = pq.distance_table(qs[0])
    live = estimate_batch(pq, data, qs[:1])[0]
    np.testing.assert_array_equal(dt0.estimate_distances(data), live)
    _fast_pq.cache_device_codes = True
    try:
        assert _fast_pq.device_codes(data.packed)
        for signed in (True, False):
            allq = estimate_batch(pq, data, qs, signed)
            for i in (0, 3, 39):
                dt = pq.distance_table(qs[i]) if signed else pq.udistance_table(qs[i])
                np.testing.assert_array_equal(allq[i], dt.estimate_distances(data))
    finally:
        _fast_pq.cache_device_codes = False
        _fast_pq.forget_device_codes()
    # in-place mutation of the packed array is seen by the next call (no stale device copy)
    saved = data.packed[:8].copy()
    data.packed[:8] = data.packed[8:16]
    est = dt0.estimate_distances(data)
    exp = np.zeros(2 * len(data.packed), dtype=np.uint64)
    oracle.estimate_pq(data.packed, dt0.tables, exp, True, oracle.ORDER_AVX)
    np.testing.assert_array_equal(est, exp.view(np.int8)[:n])
    assert not np.array_equal(est, live)
    data.packed[:8] = saved
    for q in qs:
        dt = pq.distance_table(q)
        got = dt.top(data, X, k=10)
        # oracle restatement of fast_pq.py:284-312
        idx = np.zeros(30, np.int64); val = np.zeros(30, np.int32)
        oracle.init_heap(idx, val, True)
        oracle.query_pq(data.packed, n, dt.tables, idx, val, True, None, oracle.ORDER_AVX)
        exp = idx[oracle.knn_brute1(q, X[idx], 10)]
        np.testing.assert_array_equal(got, exp)


def test_config_c3_euclidean_rotated_ivf(tk, oracle):
    """BASELINE configs[2] shape (SIFT-like: euclidean, d=128 rotated to 64 dims,
    M=32), 40k points, n_probes sweep; probes, heap arrays and ids vs the oracle."""
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(10)
    n, d, nq = 40000, 128, 200
    X = np.clip(np.abs(np.random.randn(n, d)) * 40, 0, 218).round().astype(np.float32)
    qs = np.clip(np.abs(np.random.randn(nq, d)) * 40, 0, 218).round().astype(np.float32)
    ivf = IVF("euclidean", 200, FastPQ(2))
    ivf.fit(X[:10000]).build(X, n_probes=1)
    assert ivf.pq.R is not None and ivf.pq.centers.shape == (16, 64)
    ox = _oracle_index(oracle, ivf)
    qn, qp = ivf._prepare(qs.copy())
    assert qp.dtype == np.float64
    for n_probes in (1, 5, 20):
        out, dbg = ivf.device_index().query_batch(qn, qp, 10, n_probes, debug=True)
        for qi in range(nq):
            ids, odbg = ox.query(qn[qi], 10, n_probes=n_probes, debug=True)
            np.testing.assert_array_equal(dbg["probes"][qi], odbg["probes"], err_msg=f"q{qi}")
            np.testing.assert_array_equal(dbg["heap_idx"][qi], odbg["heap_idx"], err_msg=f"q{qi}")
            np.testing.assert_array_equal(dbg["heap_val"][qi], odbg["heap_val"], err_msg=f"q{qi}")
            np.testing.assert_array_equal(out[qi][out[qi] != -1] if len(ids) < 10 else out[qi], ids)
        # the public single-query API on the raw query
        for qi in (0, 7, 99):
            np.testing.assert_array_equal(ivf.query(qs[qi].copy(), 10, n_probes=n_probes), out[qi][out[qi] != -1])


def test_pipelined_batches_and_join(tk, oracle):
    """tk_index_set_pipeline: several batches in flight give the same ids as one."""
    import ctypes as C
    from tinyknn_amd import IVF, FastPQ, _lib
    np.random.seed(5)
    n, d, nq = 20000, 100, 300
    X = np.random.randn(n, d).astype(np.float32)
    ivf = IVF("angular", 100, FastPQ(2))
    ivf.fit(X[:5000]).build(X, n_probes=1)
    dev = ivf.device_index()
    batches = []
    for b in range(5):
        qs = np.random.randn(nq, d).astype(np.float32)
        batches.append(ivf._prepare(qs))
    ref = [dev.query_batch(qn, qp, 10, 8) for qn, qp in batches]
    L = _lib.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        return p
    dev.set_pipeline(3)
    try:
        bufs = []
        for qn, qp in batches:
            qn = np.ascontiguousarray(qn, np.float32); qp = np.ascontiguousarray(qp, np.float32)
            dq_, dp_, do_ = dmalloc(qn.nbytes), dmalloc(qp.nbytes), dmalloc(nq * 10 * 8)
            assert hip.hipMemcpy(dq_, qn.ctypes.data, qn.nbytes, 1) == 0
            assert hip.hipMemcpy(dp_, qp.ctypes.data, qp.nbytes, 1) == 0
            bufs.append((dq_, dp_, do_))
        for dq_, dp_, do_ in bufs:          # 5 batches over 3 workspaces, no sync in between
            dev.query_batch_dev(dq_, dp_, False, nq, 10, 8, do_)
        dev.join(0)
        assert hip.hipDeviceSynchronize() == 0
        for (dq_, dp_, do_), exp in zip(bufs, ref):
            out = np.zeros((nq, 10), np.int64)
            assert hip.hipMemcpy(out.ctypes.data, do_, out.nbytes, 2) == 0
            np.testing.assert_array_equal(out, exp)
            for p in (dq_, dp_, do_):
                hip.hipFree(p)
    finally:
        dev.set_pipeline(1)


def test_pipelined_calls_of_different_shapes(tk, oracle):
    """Pipelined mode with calls that differ in batch size, k and n_probes, more calls than
    workspaces, a join in the middle: every call's ids equal the oracle's."""
    import ctypes as C
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(6)
    n, d = 30000, 100
    cent = np.random.randn(50, d)
    X = (cent[np.random.randint(50, size=n)] + 0.6 * np.random.randn(n, d)).astype(np.float32)
    ivf = IVF("angular", 120, FastPQ(2))
    ivf.fit(X[:6000]).build(X, n_probes=1)
    ox = _oracle_index(oracle, ivf)
    dev = ivf.device_index()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        return p

    shapes = [(300, 10, 8), (17, 5, 1), (1000, 10, 20), (64, 1, 3), (1, 10, 8), (513, 20, 5),
              (300, 10, 8), (2000, 10, 10), (3, 50, 100), (257, 10, 2), (1000, 10, 20), (40, 10, 8),
              (700, 3, 15)]
    dev.set_pipeline(2)
    try:
        calls = []
        for i, (nq, k, n_probes) in enumerate(shapes):
            qs = (cent[np.random.randint(50, size=nq)] + 0.6 * np.random.randn(nq, d)).astype(np.float32)
            qn, qp = ivf._prepare(qs)
            qn = np.ascontiguousarray(qn, np.float32)
            qp = np.ascontiguousarray(qp, np.float32)
            dq_, dp_, do_ = dmalloc(qn.nbytes), dmalloc(qp.nbytes), dmalloc(nq * k * 8)
            assert hip.hipMemcpy(dq_, qn.ctypes.data, qn.nbytes, 1) == 0
            assert hip.hipMemcpy(dp_, qp.ctypes.data, qp.nbytes, 1) == 0
            dev.query_batch_dev(dq_, dp_, False, nq, k, n_probes, do_)
            calls.append((qn, nq, k, n_probes, dq_, dp_, do_))
            if i == 6:
                dev.join(0)                      # a join with calls on both sides of it
        dev.join(0)
        assert hip.hipDeviceSynchronize() == 0
        for qn, nq, k, n_probes, dq_, dp_, do_ in calls:
            out = np.zeros((nq, k), np.int64)
            assert hip.hipMemcpy(out.ctypes.data, do_, out.nbytes, 2) == 0
            np.testing.assert_array_equal(out, ox.query_batch(qn, k, n_probes),
                                          err_msg=f"nq={nq} k={k} n_probes={n_probes}")
            for p in (dq_, dp_, do_):
                hip.hipFree(p)
    finally:
        dev.set_pipeline(1)


def test_hipgraph_captured_batch(tk):
    """BASELINE configs[3]: the whole batch pipeline captured as ONE hipGraph and
    replayed (fixed-shape launches, no host sync, no allocation after reserve)."""
    import ctypes as C
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(6)
    n, d, nq = 20000, 100, 500
    X = np.random.randn(n, d).astype(np.float32)
    ivf = IVF("angular", 100, FastPQ(2))
    ivf.fit(X[:5000]).build(X, n_probes=1)
    dev = ivf.device_index()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        return p
    for n_probes in (1, 5, 10, 20):
        qs = [ivf._prepare(np.random.randn(nq, d).astype(np.float32)) for _ in range(3)]
        ref = [dev.query_batch(qn, qp, 10, n_probes) for qn, qp in qs]   # also warms + reserves
        dq_, dp_, do_ = dmalloc(nq * d * 4), dmalloc(nq * 104 * 4), dmalloc(nq * 10 * 8)
        dev.reserve(nq, 10, n_probes)
        assert hip.hipStreamBeginCapture(stream, 0) == 0          # hipStreamCaptureModeGlobal
        dev.query_batch_dev(dq_, dp_, False, nq, 10, n_probes, do_, stream=stream)
        graph = C.c_void_p()
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
        gexec = C.c_void_p()
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        for (qn, qp), exp in zip(qs, ref):
            qn = np.ascontiguousarray(qn, np.float32); qp = np.ascontiguousarray(qp, np.float32)
            assert hip.hipMemcpyAsync(dq_, qn.ctypes.data, qn.nbytes, 1, stream) == 0
            assert hip.hipMemcpyAsync(dp_, qp.ctypes.data, qp.nbytes, 1, stream) == 0
            assert hip.hipGraphLaunch(gexec, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
            out = np.zeros((nq, 10), np.int64)
            assert hip.hipMemcpy(out.ctypes.data, do_, out.nbytes, 2) == 0
            np.testing.assert_array_equal(out, exp, err_msg=f"n_probes {n_probes}")
        hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
        for p in (dq_, dp_, do_):
            hip.hipFree(p)
