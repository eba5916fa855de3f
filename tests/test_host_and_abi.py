"""CPU-side checks: the C-ABI library loads and exports every declared symbol,
compute calls fail loudly without a GPU, host-side layout code matches the
golden vectors, and the offline fit/build code keeps the reference's contracts."""
import os
import re

import numpy as np
import pytest

from conftest import golden, ROOT


def _have_gpu():
    from tinyknn_amd import _lib
    return _lib.device_count() > 0


def test_abi_exports_every_declared_symbol():
    from tinyknn_amd import _lib
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "tinyknn_hip.h")).read()
    declared = set(re.findall(r"\b(tk_[a-z0-9_]+)\s*\(", header))
    declared.discard("tk_index")
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in tinyknn_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.tk_version() >= 1


def test_no_cpu_fallback():
    if _have_gpu():
        pytest.skip("GPU present")
    from tinyknn_amd import _lib
    from tinyknn_amd._fast_pq import init_heap
    from tinyknn_amd._fast_pq_avx import estimate_pq_avx
    with pytest.raises(_lib.TinyKnnHipError):
        init_heap(np.zeros(3, np.int64), np.zeros(3, np.int32), True)
    with pytest.raises(_lib.TinyKnnHipError):
        estimate_pq_avx(np.zeros((1, 4), np.uint64), np.zeros(8, np.uint64), np.zeros(2, np.uint64), True)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tinyknn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"
    # outside tests/, only bench.py (cpu_baseline leg) and __graft_entry__.smoke() may use it
    import glob
    for f in glob.glob(os.path.join(ROOT, "*.py")) + glob.glob(os.path.join(ROOT, "examples", "*.py")) + \
            glob.glob(os.path.join(ROOT, "include", "*.h")):
        if os.path.basename(f) in ("bench.py", "__graft_entry__.py"):
            continue
        assert "oracle" not in open(f).read(), f"{f} mentions the oracle"


@pytest.mark.parametrize("tag", ["a", "b"])
def test_layout_golden(tag):
    from tinyknn_amd._transform import transform_data, unpack, transform_tables
    g = golden("g1_layout.npz")
    packed = transform_data(g[f"codes_{tag}"])
    np.testing.assert_array_equal(packed, g[f"packed_{tag}"])
    np.testing.assert_array_equal(unpack(packed), g[f"codes_{tag}"])
    np.testing.assert_array_equal(transform_tables(g[f"table_{tag}"]), g[f"ttable_{tag}"])
    with pytest.raises(AssertionError):
        transform_data(np.zeros((15, 4), np.uint8))
    with pytest.raises(AssertionError):
        transform_data(np.full((16, 4), 16, np.uint8))


def test_typed_buffer_errors():
    from tinyknn_amd._fast_pq import init_heap, estimate_pq_sse
    with pytest.raises(ValueError):
        init_heap(np.zeros(3, np.int32), np.zeros(3, np.int32), True)
    with pytest.raises(ValueError):
        estimate_pq_sse(np.zeros((1, 4), np.uint32), np.zeros(8, np.uint64), np.zeros(2, np.uint64), True)
    with pytest.raises(ValueError):
        estimate_pq_sse(np.zeros((2, 8), np.uint64)[:, ::2], np.zeros(8, np.uint64), np.zeros(4, np.uint64), True)


def test_fit_contracts():
    from tinyknn_amd import FastPQ, IVF
    with pytest.raises(AssertionError):                       # reference tests/test_pq.py:93-97
        FastPQ(2).fit(np.zeros((0, 8), np.float32))
    with pytest.raises(AssertionError):
        IVF("manhattan", 4)
    pq = FastPQ(2)
    with pytest.raises(AssertionError):
        pq.transform(np.zeros((4, 8), np.float32))
    np.random.seed(1)
    X = np.random.randn(100, 10).astype(np.float32)
    pq.fit(X)
    assert pq.centers.shape == (16, 16) and pq.centers.dtype == np.float32   # 10 -> 16 = 2 * (4*dpb)
    assert pq.R is not None and pq.R.shape == (16, 16)
    td = pq.transform(X)
    assert td.size == 100 and td.packed.shape == (7, 8) and td.packed.dtype == np.uint64
    td2 = pq.transform(X)
    np.testing.assert_array_equal(td.packed, td2.packed)     # test_pq.py:100-108
    pq100 = FastPQ(2).fit(np.random.randn(64, 100).astype(np.float32))
    assert pq100.R is None and pq100.centers.shape == (16, 104)   # the 100-d special case
    assert not FastPQ(1).fit(np.random.randn(64, 100).astype(np.float32)).centers.flags.c_contiguous


def test_build_contracts():
    from tinyknn_amd import FastPQ, IVF
    from tinyknn_amd.utils import group_data_by_indices
    np.random.seed(2)
    X = np.random.randn(300, 8).astype(np.float32)
    ivf = IVF("angular", 6, FastPQ(2))
    ivf.fit(X).build(X, n_probes=2)
    L = len(ivf.active_centers)
    assert sum(ivf.pq_transformed_points[i].size for i in range(L)) == 2 * len(X)
    assert all(ivf.ids[i].dtype == np.int64 for i in range(L))
    np.testing.assert_allclose(np.linalg.norm(ivf.data, axis=1), 1, rtol=1e-5)
    import pickle
    ivf2 = pickle.loads(pickle.dumps(ivf))                    # examples/bench.py:88-103
    assert ivf2._dev is None and ivf2.n_clusters == 6
    # reference tests/test_utils.py:50-72
    idx = np.random.randint(0, 5, size=(50, 2))
    parts, ids = group_data_by_indices(X[:50], idx, 5)
    for g in range(5):
        mask = (idx == g).any(axis=1)
        assert sorted(map(tuple, parts[g])) == sorted(map(tuple, np.vstack([X[:50][idx[:, j] == g] for j in range(2)])))
        assert set(ids[g]) == set(np.nonzero(mask)[0])


def test_index_persistence_roundtrips(tmp_path):
    """pickle (what the reference's bench does, examples/bench.py:88-103) and the flat
    save/load give back the same index; no GPU involved."""
    import pickle
    from conftest import split_lists
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.fast_pq import TransformedData
    g = golden("g6_ivf_eu128.npz")
    codes, ids = split_lists(g)
    ivf = IVF(str(g["metric"]), len(codes), FastPQ(2))
    ivf.pq.centers, ivf.pq.sqrt_n_blocks, ivf.pq.R = g["pq_centers"], float(g["sqrt_n_blocks"]), g["R"]
    ivf.active_centers = g["active_centers"]
    ivf.pq_transformed_centers = TransformedData(int(g["center_size"]), g["center_codes"])
    ivf.pq_transformed_points = [TransformedData(int(s), c) for s, c in zip(g["list_sizes"], codes)]
    ivf.ids, ivf.data = ids, g["data"]
    path = str(tmp_path / "index.npz")
    ivf.save(path)
    # a path without the suffix: np.savez appends ".npz", load must find the same file
    ivf.pq.use_kmeans, ivf.pq.rotate_dim = False, 32
    bare = str(tmp_path / "idx")
    ivf.save(bare)
    assert (tmp_path / "idx.npz").exists()
    other = IVF.load(bare)
    assert other.pq.use_kmeans is False and other.pq.rotate_dim == 32
    for other in (IVF.load(path), other, pickle.loads(pickle.dumps(ivf))):
        assert other.metric == ivf.metric and other.pq.dims_per_block == 2
        np.testing.assert_array_equal(other.pq.centers, ivf.pq.centers)
        np.testing.assert_array_equal(other.pq.R, ivf.pq.R)
        np.testing.assert_array_equal(other.active_centers, ivf.active_centers)
        np.testing.assert_array_equal(other.pq_transformed_centers.packed, ivf.pq_transformed_centers.packed)
        np.testing.assert_array_equal(other.data, ivf.data)
        for a, b, ia, ib in zip(other.pq_transformed_points, ivf.pq_transformed_points, other.ids, ivf.ids):
            assert a.size == b.size
            np.testing.assert_array_equal(a.packed, b.packed)
            np.testing.assert_array_equal(ia, ib)


def test_a_forked_child_does_not_release_the_parents_device_handles():
    """A process forked from the one that loaded the library (multiprocessing.Manager, fork start method)
    holds copies of the Python wrappers; their close() / __del__ must not call into the library — the HIP
    context does not survive a fork, and a garbage collection in such a child once aborted a GPU test run."""
    import os
    from tinyknn_amd import _lib
    _lib.lib()
    assert _lib.owns_handles()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        os.write(w, b"1" if _lib.owns_handles() else b"0")
        os._exit(0)
    os.close(w)
    assert os.read(r, 1) == b"0"
    os.waitpid(pid, 0)
    os.close(r)


def test_copies_carry_one_code_host_check():
    """ivf._copies_carry_one_code (what a list-sharded rank vouches for before the TWIN replay): hashes per row
    instead of the code bytes — same verdicts on lists that meet the premises, that carry another code on one copy,
    and that hold a label twice."""
    from types import SimpleNamespace
    from tinyknn_amd.ivf import _copies_carry_one_code
    from tinyknn_amd._transform import transform_data

    rng = np.random.RandomState(3)
    M, n_rows, L = 8, 1000, 6
    codes = rng.randint(0, 16, size=(n_rows, M)).astype(np.uint8)

    def make(assign, codes_of=lambda lst, rows: codes[rows]):
        tds, ids = [], []
        for lst in range(L):
            rows = np.flatnonzero((assign == lst).any(axis=1))
            pad = (-len(rows)) % 16
            c = np.concatenate([codes_of(lst, rows), np.zeros((pad, M), np.uint8)])
            tds.append(SimpleNamespace(packed=transform_data(c), size=len(rows)) if len(rows) else np.empty((0, M)))
            ids.append(rows.astype(np.int64))
        return SimpleNamespace(pq_transformed_points=tds, ids=ids)

    first = rng.randint(0, L - 1, size=n_rows)
    assign = np.stack([first, (first + 1 + rng.randint(0, L - 2, size=n_rows)) % (L - 1)], axis=1)   # two DIFFERENT lists, list L-1 empty
    assert (assign[:, 0] != assign[:, 1]).all()
    assert _copies_carry_one_code(make(assign), L)
    # one copy of one row with another code
    victim = int(np.flatnonzero(assign[:, 1] == 2)[0])

    def damaged(lst, rows):
        c = codes[rows].copy()
        if lst == 2:
            c[rows == victim, 3] ^= 1
        return c
    assert not _copies_carry_one_code(make(assign, damaged), L)
    # a label twice in one list
    ivf = make(assign)
    ivf.ids[1] = ivf.ids[1].copy()
    ivf.ids[1][1] = ivf.ids[1][0]
    assert not _copies_carry_one_code(ivf, L)


def test_failed_plain_check_is_demoted_one_level_at_a_time():
    """ListShardedIndex._note_plain_failure (bit 4 of a batch's flag word): one-phase -> head form -> two-phase -> an error,
    never the same form again (every rank sees the same gathered flag, so every rank takes the same step)."""
    from tinyknn_amd.multi_gpu import ListShardedIndex
    idx = ListShardedIndex.__new__(ListShardedIndex)
    idx._plain_failed, idx._head_failed = set(), set()
    idx._one_phase = idx._head_phase = True
    idx._exchange_kind = lambda k, n_probes, pass_1: "dense"
    idx._use_plain = lambda k, n_probes, pass_1: True
    a, b = (10, 10, None), (10, 5, None)
    assert idx._scan_form(*a) == "one" and idx._scan_form(*b) == "one"
    idx._note_plain_failure({a})
    assert idx._scan_form(*a) == "head" and idx._scan_form(*b) == "one"
    idx._note_plain_failure({a})
    assert idx._scan_form(*a) == "two"
    with pytest.raises(RuntimeError, match="every scan form"):
        idx._note_plain_failure({a})
    # an engine without the head form: straight to two-phase, then the error
    idx2 = ListShardedIndex.__new__(ListShardedIndex)
    idx2._plain_failed, idx2._head_failed = set(), set()
    idx2._one_phase, idx2._head_phase = True, False
    idx2._exchange_kind, idx2._use_plain = idx._exchange_kind, idx._use_plain
    idx2._note_plain_failure({a})
    assert idx2._scan_form(*a) == "two"
    with pytest.raises(RuntimeError):
        idx2._note_plain_failure({a})
