"""Exact host front end (front.hip, _front.py): the host side of IVF.query, ivf.py:125-128 +
fast_pq.py:200-204, computed by the same BLAS calls numpy makes — bit-identical to the
reference's numpy arithmetic, without the Python loop.  Host code only: runs without a GPU.
"""
import numpy as np
import pytest

from tinyknn_amd import _front, _lib


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.fixture(scope="module")
def bound():
    if not _front.bind():
        pytest.skip("numpy's BLAS could not be bound here: " + str(_front.info()["why"]))
    return _front.info()


def test_binds_the_library_numpy_loaded(bound):
    assert bound["ok"] and bound["path"]
    assert "blas" in bound["path"].lower()
    assert _lib.lib().tk_host_blas_bound() == 1


@pytest.mark.parametrize("d", [1, 3, 20, 37, 96, 100, 128, 200, 960])
def test_normalisation_is_numpys_bit_for_bit(bound, d):
    rng = np.random.RandomState(d)
    raw = (rng.randn(257, d) * rng.lognormal(size=(257, 1))).astype(np.float32)
    want, _ = _front.numpy_prepare(raw.copy(), True, None, 0)
    got = raw.copy()
    out, qp = _front.prepare(got, True, None, 0)
    assert out is got and qp is got                      # in place, as the reference normalises q
    assert np.array_equal(bits(want), bits(got))


def test_row_alignment_does_not_matter(bound):
    rng = np.random.RandomState(7)
    base = rng.randn(100 * 64 + 8).astype(np.float32)
    for off in range(8):
        raw = base[off:off + 6400].reshape(64, 100)
        want, _ = _front.numpy_prepare(raw.copy(), True, None, 4)
        buf = np.empty_like(base)
        view = buf[off:off + 6400].reshape(64, 100)
        view[:] = raw
        got, qp = _front.prepare(view, True, None, 4)
        assert np.array_equal(bits(want), bits(got))
        assert qp.shape == (64, 104) and np.array_equal(qp[:, :100], got) and not qp[:, 100:].any()


@pytest.mark.parametrize("d,pad,rd,angular", [(128, 0, 64, False), (128, 0, 64, True), (100, 4, 64, True),
                                              (20, 4, 16, False), (11, 5, 8, True), (300, 4, 64, False)])
def test_rotation_is_numpys_matmul_bit_for_bit(bound, d, pad, rd, angular):
    rng = np.random.RandomState(d + rd)
    R = rng.randn(rd, d + pad)
    raw = rng.randn(131, d).astype(np.float32)
    wn, wp = _front.numpy_prepare(raw.copy(), angular, R, pad)
    gn, gp = _front.prepare(raw.copy(), angular, R, pad)
    assert wp.dtype == gp.dtype == np.float64 and gp.shape == (131, rd)
    assert np.array_equal(bits(wn), bits(gn))
    assert np.array_equal(bits(wp), bits(gp))


def test_zero_rows_and_zero_vectors(bound):
    e = np.zeros((0, 100), np.float32)
    qn, qp = _front.prepare(e, True, None, 4)
    assert qn.shape == (0, 100) and qp.shape == (0, 104)
    z = np.zeros((3, 100), np.float32)
    z[1, 5] = 2.0
    with np.errstate(all="ignore"):
        want, _ = _front.numpy_prepare(z.copy(), True, None, 0)
    got, _ = _front.prepare(z.copy(), True, None, 0)
    assert np.array_equal(bits(want), bits(got))          # 0/0 -> nan exactly like numpy


def test_thread_count_does_not_change_results(bound):
    rng = np.random.RandomState(3)
    raw = rng.randn(5000, 100).astype(np.float32)
    R = rng.randn(64, 104)
    L = _lib.lib()
    before = L.tk_host_threads(0)
    outs = []
    for th in (1, 3, 8):
        assert L.tk_host_threads(th) == th
        outs.append(_front.prepare(raw.copy(), True, R, 4))
    L.tk_host_threads(before)
    for qn, qp in outs[1:]:
        assert np.array_equal(bits(qn), bits(outs[0][0])) and np.array_equal(bits(qp), bits(outs[0][1]))


def test_ivf_prepare_uses_it_and_matches_the_numpy_loop(bound):
    import tinyknn_amd
    rng = np.random.RandomState(5)
    for metric, d, rot in (("angular", 100, False), ("euclidean", 128, True), ("angular", 24, True)):
        ivf = tinyknn_amd.IVF(metric, 4, tinyknn_amd.FastPQ(2))
        pq = ivf.pq
        dpadded = d + (-d) % 8
        pq.R = rng.randn(16, dpadded) if rot else None
        pq.centers = rng.randn(16, 16 if rot else dpadded).astype(np.float32)
        qs = rng.randn(300, d).astype(np.float32)
        wn, wp = _front.numpy_prepare(qs.copy(), metric == "angular", pq.R, dpadded - d)
        gn, gp = ivf._prepare(qs.copy())
        assert np.array_equal(bits(wn), bits(gn)) and np.array_equal(bits(wp), bits(gp))


def test_unbound_library_refuses_instead_of_restating(monkeypatch):
    """tk_prepare_queries_host has no arithmetic of its own: with a path that holds no cblas
    symbols the bind fails and the state is reported, nothing is computed."""
    L = _lib.lib()
    assert L.tk_host_blas_bind(b"/nonexistent/libblas.so") != 0
    assert b"dlopen" in L.tk_last_error()
    assert L.tk_host_blas_bind(_lib.lib_path().encode()) != 0        # a library without cblas_sdot
    assert b"cblas" in L.tk_last_error()
