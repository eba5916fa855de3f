"""The CPU oracle against the golden vectors produced by the compiled reference
(tests/golden/make_golden.py) and against the reference's own known answers
(/root/reference/tests/test_transform.py, test_heap.py, test_pq.py)."""
import heapq
import random
from functools import reduce

import numpy as np
import pytest

from conftest import golden, split_lists, G6_TAGS


# ---- G1 / G2 ---------------------------------------------------------------

@pytest.mark.parametrize("tag", ["a", "b"])
def test_layout_golden(oracle, tag):
    g = golden("g1_layout.npz")
    codes = g[f"codes_{tag}"]
    packed = oracle.transform_data(codes)
    np.testing.assert_array_equal(packed, g[f"packed_{tag}"])
    np.testing.assert_array_equal(oracle.unpack(packed), codes)
    np.testing.assert_array_equal(oracle.unpack(g[f"packed_{tag}"]), g[f"unpacked_{tag}"])
    np.testing.assert_array_equal(oracle.transform_tables(g[f"table_{tag}"]), g[f"ttable_{tag}"])


def test_transpose_nibble_positions(oracle):
    # reference tests/test_transform.py:80-101
    np.random.seed(10)
    n, d = 16 * 13, 2 * 7
    data0 = np.random.randint(16, size=(n, d)).astype(np.uint8)
    data = oracle.transform_data(data0)
    assert data.shape == (n // 16, d)
    shifts = np.arange(15, -1, -1, dtype=np.uint64) * 4
    data = (data[..., np.newaxis] >> shifts) & 0xF
    assert data[0, 0, -1] == data0[0][0]
    assert data[0, 0, -2] == data0[0][1]
    assert data[0, 0, -3] == data0[1][0]
    assert data[0, 0, -4] == data0[1][1]
    assert data[0, 1, -1] == data0[8][0]
    assert data[0, 1, -2] == data0[8][1]
    assert data[0, 2, -1] == data0[0][2]
    assert data[0, 2, -2] == data0[0][3]


# ---- G3 --------------------------------------------------------------------

@pytest.mark.parametrize("scalar", [True, False])
def test_estimate_golden(oracle, scalar):
    g = golden("g3_estimate.npz")
    oracle.force_scalar(scalar)
    try:
        for ci in range(5):
            d = oracle.transform_data(g[f"codes_{ci}"])
            t = oracle.transform_tables(g[f"table_{ci}"])
            for signed in (1, 0):
                for name, order in (("sse", oracle.ORDER_SSE), ("avx", oracle.ORDER_AVX)):
                    out = np.zeros(2 * len(d), dtype=np.uint64)
                    oracle.estimate_pq(d, t, out, signed, order)
                    np.testing.assert_array_equal(out.view(np.uint8), g[f"out_{ci}_{signed}_{name}"])
        out = np.zeros(2, dtype=np.uint64)
        oracle.estimate_pq(oracle.transform_data(g["kat_codes"]),
                           oracle.transform_tables(g["kat_table"]), out, False, oracle.ORDER_SSE)
        np.testing.assert_array_equal(out.view(np.uint8), g["kat_out"])
        assert out.view(np.uint8)[0] == 26 and not out.view(np.uint8)[1:].any()
    finally:
        oracle.force_scalar(False)


def _sat8(x, y):
    return max(-128, min(127, x + y))


@pytest.mark.parametrize("i,j", [(i, j) for i in (1, 4, 9) for j in (1, 5, 9)])
def test_estimate_python_model(oracle, i, j):
    # reference tests/test_transform.py:20-58 (SSE order) and test_pq.py:39-49 (AVX order)
    random.seed(100 * i + j)
    n, d = 16 * i, 4 * j
    dat = np.array([[random.randrange(16) for _ in range(d)] for _ in range(n)], dtype=np.uint8)
    tab = np.array([[random.randrange(-128, 128) for _ in range(16)] for _ in range(d)])
    data, tables = oracle.transform_data(dat), oracle.transform_tables(tab.astype(np.int8).view(np.uint8))
    out = np.zeros(2 * len(data), dtype=np.uint64)
    oracle.estimate_pq(data, tables, out, True, oracle.ORDER_SSE)
    exp = [reduce(_sat8, (int(tab[m][dat[r][m]]) for m in range(d)), 0) for r in range(n)]
    np.testing.assert_array_equal(out.view(np.int8), np.array(exp))
    oracle.estimate_pq(data, tables, out, True, oracle.ORDER_AVX)
    exp = []
    for r in range(n):
        a = [0, 0]
        for m in range(d):
            a[(m >> 1) & 1] = _sat8(a[(m >> 1) & 1], int(tab[m][dat[r][m]]))
        exp.append(_sat8(a[0], a[1]))
    np.testing.assert_array_equal(out.view(np.int8), np.array(exp))
    utab = np.abs(tab) % (256 // d * 2 + 1)
    oracle.estimate_pq(data, oracle.transform_tables(utab.astype(np.uint8)), out, False, oracle.ORDER_AVX)
    exp = np.minimum([sum(int(utab[m][dat[r][m]]) for m in range(d)) for r in range(n)], 255)
    np.testing.assert_array_equal(out.view(np.uint8), exp)


# ---- G4 --------------------------------------------------------------------

def test_query_pq_golden(oracle):
    g = golden("g4_query.npz")
    checked = 0
    for ci, R, n, M, signed, use_labels in g["meta"]:
        d1 = oracle.transform_data(g[f"codes1_{ci}"])
        d2 = oracle.transform_data(g[f"codes2_{ci}"])
        t = oracle.transform_tables(g[f"table_{ci}"])
        l1 = g[f"labels1_{ci}"] if use_labels else None
        l2 = g[f"labels2_{ci}"] if use_labels else None
        for name, order in (("sse", oracle.ORDER_SSE), ("avx", oracle.ORDER_AVX)):
            if f"idx_{ci}_{name}" not in g:
                continue
            idx = np.zeros(R, np.int64)
            val = np.zeros(R, np.int32)
            oracle.init_heap(idx, val, signed)
            for dd, ll in ((d1, l1), (d2, l2), (d1, l1)):
                oracle.query_pq(dd, int(n), t, idx, val, signed, ll, order)
            np.testing.assert_array_equal(idx, g[f"idx_{ci}_{name}"])
            np.testing.assert_array_equal(val, g[f"val_{ci}_{name}"])
            checked += 1
    assert checked >= 50


def test_heap_golden(oracle):
    g = golden("g4_query.npz")
    idx = np.empty(3, np.int64); val = np.empty(3, np.int32)
    oracle.init_heap(idx, val, True)
    np.testing.assert_array_equal(idx, g["heap_init_idx"])
    np.testing.assert_array_equal(val, g["heap_init_val"])
    idx = np.empty(2, np.int64); val = np.empty(2, np.int32)
    oracle.init_heap(idx, val, True)
    oracle.insert(idx, val, 1, 10); oracle.insert(idx, val, 1, 10)
    np.testing.assert_array_equal(idx, [-1, 1])          # reference tests/test_heap.py:44-49
    np.testing.assert_array_equal(val, [127, 10])
    idx = np.empty(13, np.int64); val = np.empty(13, np.int32)
    oracle.init_heap(idx, val, True)
    idx2, val2 = idx.copy(), val.copy()
    for t, (lab, v) in enumerate(g["heap_ops"]):
        oracle.insert(idx, val, lab, v)
        oracle.insert_is(idx2, val2, lab, v)
        np.testing.assert_array_equal(idx, g["heap_trace_idx"][t])
        np.testing.assert_array_equal(val, g["heap_trace_val"][t])
        np.testing.assert_array_equal(idx2, g["heap_is_trace_idx"][t])
        np.testing.assert_array_equal(val2, g["heap_is_trace_val"][t])


def test_heap_vs_heapq(oracle):
    # reference tests/test_heap.py:52-64
    np.random.seed(10)
    idx = np.empty(10, np.int64); val = np.empty(10, np.int32)
    oracle.init_heap(idx, val, True)
    py = [(-127, -1)] * 10
    for t in range(1000):
        top = -py[0][0]
        assert top == val[0]
        v = np.random.randint(10000 // (t + 1))
        if v < val[0]:
            oracle.insert(idx, val, t, v)
        if v < top:
            heapq.heappop(py)
            heapq.heappush(py, (-v, t))
        assert set(val) == {-vi for vi, _ in py}


# ---- G5 / G7 ---------------------------------------------------------------

def test_distance_tables_golden(oracle):
    g = golden("g5_tables.npz")
    for ci, d, dpb, n, size, rotated, f_order in g["meta"]:
        centers = g[f"centers_{ci}"]
        if f_order:
            centers = np.asfortranarray(centers)
        snb = float(g[f"sqrt_n_blocks_{ci}"])
        packed = g[f"packed_{ci}"]
        for qi in range(len(g[f"qs_{ci}"])):
            qpq = g[f"qpq_{ci}"][qi]
            assert (qpq.dtype == np.float64) == bool(rotated)
            table, shift, scale = oracle.distance_table(centers, int(dpb), qpq, snb, True)
            np.testing.assert_array_equal(oracle.transform_tables(table), g[f"tables_{ci}"][qi])
            assert shift == g[f"shift_{ci}"][qi] and shift.dtype == g[f"shift_{ci}"].dtype
            assert scale == g[f"scale_{ci}"][qi]
            # G7: estimate_distances(rescale=True), fast_pq.py:281-282, <= 1e-4
            out = np.zeros(2 * len(packed), dtype=np.uint64)
            oracle.estimate_pq(packed, oracle.transform_tables(table), out, True, oracle.ORDER_AVX)
            est = out.view(np.int8)[:size].astype(np.float32)
            resc = qpq @ qpq + (est / scale + shift)
            np.testing.assert_allclose(resc, g[f"est_rescaled_{ci}"][qi], rtol=0, atol=1e-4)
            utable, ushift, uscale = oracle.distance_table(centers, int(dpb), qpq, snb, False)
            np.testing.assert_array_equal(oracle.transform_tables(utable), g[f"utables_{ci}"][qi])
            assert ushift == g[f"ushift_{ci}"][qi] and uscale == g[f"uscale_{ci}"][qi]
            oracle.estimate_pq(packed, oracle.transform_tables(utable), out, False, oracle.ORDER_AVX)
            np.testing.assert_array_equal(out.view(np.uint8)[:size], g[f"uest_{ci}"][qi])


# ---- G6 --------------------------------------------------------------------

def load_oracle_index(oracle, g):
    codes, ids = split_lists(g)
    R = g["R"] if "R" in g else None
    return oracle.OracleIndex(g["pq_centers"], 2, R, float(g["sqrt_n_blocks"]),
                              g["active_centers"], g["center_codes"], codes,
                              g["list_sizes"], ids, g["data"])


@pytest.mark.parametrize("tag", G6_TAGS)
def test_ivf_query_golden(oracle, tag):
    g = golden(f"g6_ivf_{tag}.npz")
    ix = load_oracle_index(oracle, g)
    k = 10
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        for qi, qn in enumerate(g["qn"]):
            ids, dbg = ix.query(qn, k, n_probes=n_probes, debug=True)
            np.testing.assert_array_equal(dbg["probes"], g[f"probes_p{n_probes}"][qi])
            np.testing.assert_array_equal(dbg["heap_idx"], g[f"heap_idx_p{n_probes}"][qi])
            np.testing.assert_array_equal(dbg["heap_val"], g[f"heap_val_p{n_probes}"][qi])
            exp = g[f"ids_p{n_probes}"][qi]
            np.testing.assert_array_equal(ids, exp[exp != -1] if len(ids) < k else exp)
            if n_probes == int(g["probes_list"][0]):
                np.testing.assert_array_equal(oracle.transform_tables(dbg["table"]), g["tables"][qi])
        batch = ix.query_batch(g["qn"], k, n_probes=n_probes)
        np.testing.assert_array_equal(batch, g[f"ids_p{n_probes}"])
