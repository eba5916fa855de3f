"""The documented limits of the C ABI fail loudly and leave the index usable:
heap size, queries per sharded batch, fast-mode dimension, assignment k, lane-kernel heap sizes."""
import numpy as np
import pytest

from conftest import golden
from test_hip_parity import ivf_from_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    g = golden("g6_ivf_an100.npz")
    return g, ivf_from_fixture(None, g)


def test_heap_larger_than_lds_is_refused_and_index_survives(small):
    from tinyknn_amd import _lib
    g, ivf = small
    dev = ivf.device_index()
    with pytest.raises((_lib.TinyKnnHipError, AssertionError)):
        dev.query_batch(g["qn"], g["qpq"], 10, 5, pass_1=5461)       # R * 12 + 16 > 64 KiB
    np.testing.assert_array_equal(dev.query_batch(g["qn"], g["qpq"], 10, 5), g["ids_p5"])


@pytest.mark.parametrize("pass_1", [149, 150, 232, 574, 575, 1500])
def test_heap_sizes_around_the_lane_kernel_limits(small, oracle, pass_1):
    """pass_1 = 574 is the largest heap of the lane-per-query kernel, 149 with the duplicate test;
    one more switches to the wave-per-query kernel — same ids either side."""
    from test_oracle_golden import load_oracle_index
    g, ivf = small
    ox = load_oracle_index(oracle, g)
    got = ivf.device_index().query_batch(g["qn"], g["qpq"], 10, 5, pass_1=pass_1)
    np.testing.assert_array_equal(got, ox.query_batch(g["qn"], 10, 5, pass_1))
    gb = golden("g6_ivf_an100b2.npz")                               # repeating labels
    ivfb = ivf_from_fixture(None, gb)
    oxb = load_oracle_index(oracle, gb)
    got = ivfb.device_index().query_batch(gb["qn"], gb["qpq"], 10, 5, pass_1=pass_1)
    np.testing.assert_array_equal(got, oxb.query_batch(gb["qn"], 10, 5, pass_1))


def test_sharded_batch_larger_than_131072_is_refused(small):
    import torch
    from tinyknn_amd import _lib
    from tinyknn_amd.multi_gpu import _HipShardEngine, shard_lists
    g, ivf = small
    owner = shard_lists(g["list_sizes"], 1)
    e = _HipShardEngine(ivf, owner, 0, 1, 1)
    nq = 131073
    qn = torch.zeros((nq, ivf.data.shape[1]), dtype=torch.float32, device="cuda")
    qp = torch.zeros((nq, ivf.pq.centers.shape[1]), dtype=torch.float32, device="cuda")
    send = torch.zeros(16 * 1024, dtype=torch.uint8, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(AssertionError, match="131072"):      # argument errors: the reference layer's convention
        e.scan(0, qn, qp, 10, 5, None, 1024, send, flag)
    e.dev.close()


def test_fast_mode_refuses_angular_beyond_128_dims():
    from tinyknn_amd import IVF, FastPQ, _lib
    rng = np.random.RandomState(0)
    X = rng.randn(2000, 160).astype(np.float32)
    ivf = IVF("angular", 8, FastPQ(2, use_kmeans=False, rotate_dim=None))
    ivf.fit(X).build(X, n_probes=1)
    qs = rng.randn(5, 160).astype(np.float32)
    exact = ivf.query_batch(qs, 5, n_probes=3)                      # the exact path has no such limit
    assert exact.shape == (5, 5)
    with pytest.raises((_lib.TinyKnnHipError, AssertionError)):
        ivf.query_batch(qs, 5, n_probes=3, fast=True)


def test_assign_lists_k_range():
    """k = 1 .. 9 on the device (examples/bench.py:108-111 sweeps build_probes 1 .. 9); beyond: refused."""
    from tinyknn_amd import _lib
    X = np.zeros((100, 16), dtype=np.float32)
    Y = np.eye(16, dtype=np.float32)[:12]
    yn = np.einsum("ij,ij->i", Y, Y)
    out = np.zeros((100, 10), dtype=np.int64)
    rc = _lib.lib().tk_assign_lists(_lib.ptr(X, _lib._f32p), 100, 16, 0, Y.ctypes.data, 0, yn.ctypes.data,
                                    12, 10, _lib.ptr(out, _lib._i64p))
    assert rc < 0 and b"k must be 1 .. 9" in _lib.lib().tk_last_error()


@pytest.mark.parametrize("k", [3, 5, 9])
@pytest.mark.parametrize("y64", [False, True])
def test_assign_lists_up_to_nine_lists_per_row(k, y64):
    """knn_brute(X, Y, k) for k = 3 .. 9 on the device = numpy's argpartition(part, k)[:, :k] of numpy's
    own distances on this host (ascending there) = the oracle's restatement."""
    from oracle import oracle as O
    from tinyknn_amd import _lib
    from tinyknn_amd.utils import knn_brute
    rng = np.random.RandomState(k)
    n, d, L = 1200, 100, 157
    X = rng.randn(n, d).astype(np.float32)
    Y = rng.randn(L, d).astype(np.float64 if y64 else np.float32)
    for metric in ("euclidean", "angular"):
        Yn = Y / np.linalg.norm(Y, axis=1, keepdims=True) if metric == "angular" else Y
        Yn = np.ascontiguousarray(Yn)
        yn = np.ascontiguousarray(np.einsum("ij,ij->i", Yn, Yn))
        out = np.zeros((n, k), dtype=np.int64)
        _lib.check(_lib.lib().tk_assign_lists(_lib.ptr(X, _lib._f32p), n, d, int(metric == "angular"),
                                              Yn.ctypes.data, int(y64), yn.ctypes.data, L, k,
                                              _lib.ptr(out, _lib._i64p)))
        want = knn_brute(X, Y, k=k, metric=metric)
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(out, O.assign(X, Y, k, metric))


def test_device_assignment_defers_to_this_hosts_numpy_for_k_above_two(monkeypatch):
    """The ORDER of argpartition's first k is numpy's implementation detail (ascending on the fixture
    host).  IVF.build(device=True) checks the device's k > 2 assignment against this host's numpy on
    sampled chunks and falls back to numpy when they differ — simulated here by a numpy that returns
    the columns in another order."""
    import tinyknn_amd
    from tinyknn_amd import ivf as ivf_mod
    from tinyknn_amd.utils import knn_brute as real
    rng = np.random.RandomState(4)
    X = rng.randn(3000, 32).astype(np.float32)
    index = tinyknn_amd.IVF("euclidean", 40, tinyknn_amd.FastPQ(2))
    index.all_centers = X[:40].copy()
    got = index._nearest_on_device(X, 3)
    np.testing.assert_array_equal(got, real(X, index.all_centers, k=3, metric="euclidean"))      # this host: ascending
    monkeypatch.setattr(ivf_mod, "knn_brute", lambda *a, **k: real(*a, **k)[:, ::-1].copy())
    with pytest.warns(UserWarning, match="falls back to numpy"):
        got = index._nearest_on_device(X, 3)
    np.testing.assert_array_equal(got, real(X, index.all_centers, k=3, metric="euclidean")[:, ::-1])


def test_one_handle_from_several_threads_is_serialised(small):
    """The reference's entry points are nogil and re-entrant; a tk_index handle is not — its entry
    points take a per-handle lock, so concurrent callers are serialised instead of corrupting
    the pipeline state (ctypes releases the GIL during the calls)."""
    import threading
    g, ivf = small
    dev = ivf.device_index()
    errors = []

    def worker(n_probes):
        try:
            for _ in range(15):
                out, dbg = dev.query_batch(g["qn"], g["qpq"], 10, n_probes, debug=True)
                np.testing.assert_array_equal(out, g[f"ids_p{n_probes}"])
                np.testing.assert_array_equal(dbg["probes"], g[f"probes_p{n_probes}"])
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=worker, args=(p,)) for p in (1, 2, 5, 10, 5, 2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors[:2]
