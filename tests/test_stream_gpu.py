"""Streaming sessions (tk_stream_*, front.hip): raw queries in, ids out, batches overlapped —
the same ids as the reference's per-query IVF.query (golden fixtures generated from the
compiled reference) whatever the batching, the pipeline depth or the order of the waits."""
import numpy as np
import pytest

from conftest import G6_TAGS, golden
from test_hip_parity import ivf_from_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tk():
    import tinyknn_amd
    from tinyknn_amd import _lib
    if _lib.device_count() < 1:
        pytest.skip("no GPU")
    return tinyknn_amd


@pytest.mark.parametrize("tag", G6_TAGS)
@pytest.mark.parametrize("depth", [1, 2])
def test_stream_matches_golden(tk, tag, depth):
    g = golden(f"g6_ivf_{tag}.npz")
    ivf = ivf_from_fixture(tk, g)
    dev = ivf.device_index()
    dev.set_pipeline(depth)
    qs = np.ascontiguousarray(g["qs"], dtype=np.float32)
    nq, k = len(qs), 10
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        want = g[f"ids_p{n_probes}"]
        # ragged batches, more submits than slots, results collected by slot reuse + drain
        st = dev.stream(16, k, n_probes, slots=3)
        out = np.full((nq, k), -7, dtype=np.int64)
        o, sizes = 0, [16, 1, 7, 16, 3, 16, 16, 2]
        i = 0
        while o < nq:
            m = min(sizes[i % len(sizes)], nq - o)
            st.submit(qs[o:o + m], out[o:o + m])
            o += m
            i += 1
        st.drain()
        st.close()
        np.testing.assert_array_equal(out, want)
    dev.set_pipeline(1)


def test_wait_in_any_order_and_prepared_rows(tk):
    g = golden("g6_ivf_an100.npz")
    ivf = ivf_from_fixture(tk, g)
    dev = ivf.device_index()
    dev.set_pipeline(2)
    qs = np.ascontiguousarray(g["qs"], dtype=np.float32)
    k, n_probes = 10, 5
    want = g[f"ids_p{n_probes}"]
    st = dev.stream(4, k, n_probes, slots=8)
    outs, tickets = [], []
    for o in range(0, 24, 4):
        out = np.full((4, k), -7, dtype=np.int64)
        tickets.append(st.submit(qs[o:o + 4], out))
        outs.append(out)
    for j in (5, 0, 3):            # newest first: forces the pipeline flush
        st.wait(tickets[j])
        np.testing.assert_array_equal(outs[j], want[4 * j:4 * j + 4])
    st.wait(tickets[5])            # twice is harmless
    # prepared rows through the same session
    qn, qp = ivf._prepare(qs[:4].copy())
    out = np.full((4, k), -7, dtype=np.int64)
    st.wait(st.submit_prepared(qn, None, out))
    np.testing.assert_array_equal(out, want[:4])
    st.drain()
    for j in range(6):
        np.testing.assert_array_equal(outs[j], want[4 * j:4 * j + 4])
    st.close()
    dev.set_pipeline(1)


def test_query_raw_chunks_and_query_batch_agree(tk):
    """IVF.query_batch (exact, streamed in chunks) == the synchronous debug path == per-query."""
    g = golden("g6_ivf_eu128.npz")          # rotated PQ: float64 table-build queries
    ivf = ivf_from_fixture(tk, g)
    dev = ivf.device_index()
    qs = np.ascontiguousarray(g["qs"], dtype=np.float32)
    big = np.concatenate([qs] * 5)
    old = dev.CHUNK
    try:
        type(dev).CHUNK = 32           # several chunks in flight
        for n_probes in (1, 5):
            want = np.concatenate([g[f"ids_p{n_probes}"]] * 5)
            np.testing.assert_array_equal(ivf.query_batch(big, 10, n_probes=n_probes), want)
            qn, qp = ivf._prepare(big.copy())
            np.testing.assert_array_equal(dev.query_batch(qn, qp, 10, n_probes), want)
            sync, _ = dev.query_batch(qn[:40], qp[:40], 10, n_probes, debug=True)
            np.testing.assert_array_equal(sync, want[:40])
    finally:
        type(dev).CHUNK = old


def test_stream_argument_errors(tk):
    g = golden("g6_ivf_an20.npz")
    ivf = ivf_from_fixture(tk, g)
    dev = ivf.device_index()
    st = dev.stream(4, 10, 2)
    qs = np.ascontiguousarray(g["qs"][:8], dtype=np.float32)
    with pytest.raises(AssertionError):
        st.submit(qs, np.zeros((8, 10), np.int64))       # nq > max_nq
    with pytest.raises(AssertionError):
        st.wait(99)                                       # unknown ticket
    st.close()
