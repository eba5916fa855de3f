"""Randomised shapes: small indexes with awkward sizes (d not a multiple of the PQ padding,
more probes than lists, k above the candidate count, empty and tiny lists, repeated labels,
batches that are not a multiple of anything) through every execution mode of the device
index, each compared with the CPU oracle row by row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.RandomState(seed)
    d = int(rng.choice([3, 8, 17, 20, 33, 100, 130]))
    n_clusters = int(rng.choice([1, 2, 7, 16, 17, 40]))
    n = int(rng.choice([n_clusters + 1, 60, 333, 1500, 3000]))
    n = max(n, 17 * max(1, n_clusters // 8))       # FastPQ.fit wants >= 16 distinct points
    return dict(
        d=d, n=n, n_clusters=n_clusters,
        metric=str(rng.choice(["euclidean", "angular"])),
        build_probes=int(min(n_clusters, rng.choice([1, 1, 2, 3]))),
        k=int(rng.choice([1, 5, 10, 50])),
        n_probes=int(rng.choice([1, 2, 5, n_clusters, n_clusters + 3])),
        nq=int(rng.choice([1, 3, 65, 200])),
        spread=float(rng.choice([0.3, 1.0, 6.0])),     # 6.0: far-apart clusters, int8 rails
    )


import os

# TINYKNN_FUZZ_SEEDS=n widens the sweep for a one-off run (round 2: 400 seeds green after the
# replay / hash-set / scan-grid changes)
@pytest.mark.parametrize("seed", range(int(os.environ.get("TINYKNN_FUZZ_SEEDS", "48"))))
def test_random_index_all_modes(oracle, seed):
    from tinyknn_amd import IVF, FastPQ
    from test_hip_parity import _oracle_index
    c = _case(seed)
    rng = np.random.RandomState(1000 + seed)
    cent = rng.randn(max(c["n_clusters"], 3), c["d"]) * c["spread"]
    X = (cent[rng.randint(len(cent), size=c["n"])] + 0.5 * rng.randn(c["n"], c["d"])).astype(np.float32)
    qs = (cent[rng.randint(len(cent), size=c["nq"])] + 0.5 * rng.randn(c["nq"], c["d"])).astype(np.float32)
    ivf = IVF(c["metric"], c["n_clusters"], FastPQ(2))
    try:
        ivf.fit(X).build(X, n_probes=c["build_probes"])
    except AssertionError:
        pytest.skip("the host build rejects this shape exactly as the reference does")
    if len(ivf.active_centers) != c["n_clusters"]:
        pytest.skip("inactive centres: the reference's grouping asserts on them")
    ox = _oracle_index(oracle, ivf)
    qn, qp = ivf._prepare(qs.copy())
    want = ox.query_batch(qn, c["k"], c["n_probes"])
    dev = ivf.device_index()
    for heap_mode, scan_mode in ((0, 0), (1, 1), (2, 2), (0, 2), (3, 0)):
        dev.set_heap_mode(heap_mode)
        dev.set_scan_mode(scan_mode)
        got = dev.query_batch(qn, qp, c["k"], c["n_probes"])
        np.testing.assert_array_equal(got, want, err_msg=f"{c} heap_mode={heap_mode} scan_mode={scan_mode}")
    dev.set_heap_mode(0)
    dev.set_scan_mode(0)
    # pipelined workspaces and streams (the host entry point joins after every call; calls
    # truly in flight are covered by test_hip_parity.py::test_pipelined_batches_and_join)
    dev.set_pipeline(2)
    outs = [dev.query_batch(qn, qp, c["k"], c["n_probes"]) for _ in range(4)]
    dev.set_pipeline(1)
    for o in outs:
        np.testing.assert_array_equal(o, want, err_msg=f"{c} pipelined")
    # device build of the same index answers identically
    if c["build_probes"] <= 2 and X.shape[1] <= 128:
        dv = IVF(c["metric"], c["n_clusters"], FastPQ(2))
        dv.all_centers, dv.pq = ivf.all_centers, ivf.pq
        dv.build(X, n_probes=c["build_probes"], device=True)
        np.testing.assert_array_equal(dv.query_batch(qs, c["k"], c["n_probes"]), want, err_msg=f"{c} device build")


@pytest.mark.parametrize("seed", range(int(os.environ.get("TINYKNN_FUZZ_SEEDS", "32"))))
def test_random_index_sharded_both_exchanges(oracle, seed):
    """(TINYKNN_FUZZ_SEEDS=400: green after the filtered exchange went in.)
    The same awkward shapes over 1-4 simulated ranks: dense and filtered exchange (lists shorter
    than the heap — the bound stays at its rail and everything travels —, empty lists, probe
    lists that wrap, repeated labels, one query) against the oracle."""
    from tinyknn_amd import IVF, FastPQ
    from test_hip_parity import _oracle_index
    from test_shard_gpu import simulate_world
    c = _case(500 + seed)
    rng = np.random.RandomState(2000 + seed)
    cent = rng.randn(max(c["n_clusters"], 3), c["d"]) * c["spread"]
    X = (cent[rng.randint(len(cent), size=c["n"])] + 0.5 * rng.randn(c["n"], c["d"])).astype(np.float32)
    qs = (cent[rng.randint(len(cent), size=c["nq"])] + 0.5 * rng.randn(c["nq"], c["d"])).astype(np.float32)
    ivf = IVF(c["metric"], c["n_clusters"], FastPQ(2))
    try:
        ivf.fit(X).build(X, n_probes=c["build_probes"])
    except AssertionError:
        pytest.skip("the host build rejects this shape exactly as the reference does")
    if len(ivf.active_centers) != c["n_clusters"]:
        pytest.skip("inactive centres: the reference's grouping asserts on them")
    ox = _oracle_index(oracle, ivf)
    qn, qp = ivf._prepare(qs.copy())
    want = ox.query_batch(qn, c["k"], c["n_probes"])
    world = int(rng.choice([1, 2, 3, 4]))
    worst = c["nq"] * min(c["n_probes"], c["n_clusters"]) * (c["n"] // 16 + 2)
    for ex in ("dense", "filtered"):
        st = {}
        ids, flags, _ = simulate_world(ivf, world, qn, qp, c["k"], c["n_probes"], capacity=worst,
                                       exchange=ex, stats=st,
                                       coarse=str(rng.choice(["home", "replicated"])))
        assert not flags.any()
        np.testing.assert_array_equal(ids, want, err_msg=f"{c} world={world} {ex}")
        if ex == "filtered":
            assert st["records"] <= st["dense_blocks"]
