"""List-sharded index on the MI355X.  A 1-GPU box cannot run W RCCL ranks,
so the W ranks are (a) simulated in one process — W shard handles on the same device,
the all-to-all done by hand on the region layout the C ABI documents — and (b) run as two
gloo processes sharing the GPU (collectives staged through the host).  Either way the ids
must equal the unsharded index's, the golden fixtures' and the oracle's."""
import os
import sys

import numpy as np
import pytest

from conftest import G6_TAGS, golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def simulate_world(ivf, world, qn, qp, k, n_probes, capacity=None, pass_1=None, owner=None,
                   coarse="home", engines=None, sizes=None, exchange="dense", stats=None, plain=True,
                   one_phase=False):
    """-> (ids (nq, k), overflow flags (world,), capacity).  coarse="home": every simulated rank
    runs the coarse stage of its home queries only and the probe lists are gathered by hand;
    "replicated": every rank derives all probe lists itself."""
    import torch
    from tinyknn_amd.multi_gpu import _HipShardEngine, shard_capacity, shard_lists
    L = ivf.active_centers.shape[0]
    if sizes is None:
        sizes = np.array([0 if isinstance(t, np.ndarray) else t.size
                          for t in ivf.pq_transformed_points[:L]], dtype=np.int64)
    if owner is None:
        owner = shard_lists(sizes, world)
    nq = len(qn)
    if capacity is None:
        capacity = shard_capacity(sizes, owner, world, nq, n_probes)
    qh = -(-nq // world)
    qn_t = torch.from_numpy(np.ascontiguousarray(qn, dtype=np.float32)).cuda()
    qp_t = torch.from_numpy(np.ascontiguousarray(qp)).cuda()
    if engines is None:
        engines = [_HipShardEngine(ivf, owner, r, world, 1) for r in range(world)]
    sends = [torch.full((world, capacity * 16), 0xAB, dtype=torch.uint8, device="cuda")
             for _ in range(world)]
    flags = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(world)]
    p_all = None
    if coarse == "home":
        kc = min(n_probes, L)
        homes_p = [torch.full((qh * kc,), -7, dtype=torch.int64, device="cuda") for _ in range(world)]
        for r, e in enumerate(engines):
            e.coarse(0, qn_t, qp_t, k, n_probes, pass_1, homes_p[r])
        p_all = torch.cat(homes_p).contiguous()                                 # all-gather
    # the scan: in two phases (first lists exactly -> bound, min-reduced -> the rest on the matrix
    # cores where the bound allows) wherever the engine says it applies, else in one
    two_phase = plain and engines[0].plain_ok(k, n_probes, pass_1)
    b_red = None
    if one_phase == "head":
        # tk_index_shard_scan_head_dev -> all-reduce(MIN) -> tk_index_shard_scan_plain_dev(bound): the check
        # at home cannot fail
        assert exchange == "dense"
        heads = [torch.zeros(nq, dtype=torch.uint8, device="cuda") for _ in range(world)]
        for r, e in enumerate(engines):
            e.scan_head(0, qn_t, qp_t, k, n_probes, pass_1, capacity, sends[r], flags[r], heads[r], probes_all=p_all)
        b_head = torch.stack(heads).min(dim=0).values.contiguous()
        for r, e in enumerate(engines):
            e.scan_plain(0, qn_t, qp_t, k, n_probes, pass_1, capacity, sends[r], flags[r], probes_all=p_all,
                         bound=b_head)
    elif one_phase:
        # tk_index_shard_scan_plain_dev: heads exactly, the rest on the matrix cores, no bound exchange;
        # the home replays check the lemma (flag bit 4)
        assert exchange == "dense"
        for r, e in enumerate(engines):
            e.scan_plain(0, qn_t, qp_t, k, n_probes, pass_1, capacity, sends[r], flags[r], probes_all=p_all)
    elif two_phase:
        firsts = [torch.zeros(nq, dtype=torch.uint8, device="cuda") for _ in range(world)]
        for r, e in enumerate(engines):
            e.scan_first(0, qn_t, qp_t, k, n_probes, pass_1, capacity, sends[r], flags[r], firsts[r],
                         probes_all=p_all)
        b_red = torch.stack(firsts).min(dim=0).values.contiguous()             # all-reduce(MIN)
        for r, e in enumerate(engines):
            e.scan_rest(0, qn_t, k, n_probes, pass_1, capacity, sends[r], b_red)
        if stats is not None:
            stats["two_phase"] = True
            stats["plain"] = [e.dev.shard_plain_stats(0) for e in engines]
    else:
        for r, e in enumerate(engines):
            e.scan(0, qn_t, qp_t, k, n_probes, pass_1, capacity, sends[r], flags[r], probes_all=p_all)
    homes = []
    if exchange in ("filtered", "filtered-regions"):
        bounds = [torch.zeros(nq, dtype=torch.uint8, device="cuda") for _ in range(world)]
        for r, e in enumerate(engines):
            e.bound(0, qn_t, k, n_probes, pass_1, capacity, sends[r].view(-1), bounds[r])
        b_all = torch.stack(bounds).min(dim=0).values.contiguous()             # all-reduce(MIN)
        if b_red is not None:       # the first lists were scored exactly: the same bound either way
            assert torch.equal(b_all, b_red)
        assert (torch.stack(bounds) != 255).sum(dim=0).le(1).all()             # one owner per query
        if stats is not None:
            stats["bound"] = b_all.cpu().numpy()
        counts, recs = [], []
        region = None
        if exchange == "filtered-regions":
            region = capacity if stats is None or "region" not in stats else int(stats["region"])
        for r, e in enumerate(engines):
            c = torch.full((3 * world,), -5, dtype=torch.int32, device="cuda")
            rec = torch.full((world * (region or capacity), 5), -9, dtype=torch.int32, device="cuda")
            if region:
                e.filter_regions(0, qn_t, k, n_probes, pass_1, capacity, sends[r].view(-1), b_all, c, rec,
                                 region, flags[r])
            else:
                e.filter(0, qn_t, k, n_probes, pass_1, capacity, sends[r].view(-1), b_all, c, rec)
            counts.append(c.cpu().numpy())
            recs.append(rec)
        if stats is not None:
            stats["max_region_count"] = int(max(c[:world].max() for c in counts))
            stats["records"] = int(sum(c[:world].sum() for c in counts))
            stats["dense_blocks"] = int(sum(c[2 * world:].sum() for c in counts))
        for h, e in enumerate(engines if region else []):
            # equal-split all-to-all of the regions and of the counts; the counts stay on the device
            rrec = torch.cat([recs[s_].view(world, region, 5)[h] for s_ in range(world)]).contiguous()
            rcnt = torch.tensor([int(counts[s_][h]) for s_ in range(world)], dtype=torch.int32, device="cuda")
            out = torch.zeros(qh * k, dtype=torch.int64, device="cuda")
            e.finish_regions(0, qn_t, k, n_probes, pass_1, rrec, rcnt, region, out, flags[h])
            homes.append(out.view(qh, k))
        for h, e in enumerate([] if region else engines):
            parts = []
            for s_ in range(world):                                             # all-to-all (splits)
                o = int(counts[s_][:h].sum())
                parts.append(recs[s_][o:o + int(counts[s_][h])])
            rrec = torch.cat(parts).contiguous()
            out = torch.zeros(qh * k, dtype=torch.int64, device="cuda")
            e.finish_filtered(0, qn_t, k, n_probes, pass_1, rrec, rrec.shape[0], out, flags[h])
            homes.append(out.view(qh, k))
    else:
        for h, e in enumerate(engines):
            recv = torch.stack([sends[s][h] for s in range(world)]).contiguous()   # all-to-all
            out = torch.zeros(qh * k, dtype=torch.int64, device="cuda")
            e.finish(0, qn_t, k, n_probes, pass_1, capacity, recv, out, flag=flags[h] if one_phase else None)
            homes.append(out.view(qh, k))
    torch.cuda.synchronize()
    if stats is not None:
        stats["usage"] = max(e.usage(0) for e in engines)
    ids = torch.cat(homes).cpu().numpy()
    assert (ids[nq:] == -1).all()
    for e in engines:
        e.dev.close()
    return ids[:nq], np.array([int(f.item()) for f in flags]), capacity


@pytest.mark.parametrize("tag", G6_TAGS)
@pytest.mark.parametrize("world,coarse", [(1, "home"), (2, "home"), (3, "home"), (2, "replicated")])
def test_sharded_golden(tag, world, coarse):
    from test_hip_parity import ivf_from_fixture
    g = golden(f"g6_ivf_{tag}.npz")
    ivf = ivf_from_fixture(None, g)
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, coarse=coarse)
        assert not flags.any()
        np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])


@pytest.mark.parametrize("tag", G6_TAGS)
@pytest.mark.parametrize("world,coarse", [(1, "home"), (3, "home"), (2, "replicated")])
def test_sharded_filtered_golden(tag, world, coarse):
    """Filtered exchange (bound after the first list, blocks below it as records): the golden ids."""
    from test_hip_parity import ivf_from_fixture
    g = golden(f"g6_ivf_{tag}.npz")
    ivf = ivf_from_fixture(None, g)
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, coarse=coarse,
                                       exchange="filtered")
        assert not flags.any()
        np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])
        # fixed record regions + counts read on the device (tk_index_shard_*_regions_dev): same ids;
        # regions exactly as large as the largest count pass, one record less raises the flag
        st = {}
        ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, coarse=coarse,
                                       exchange="filtered-regions", stats=st)
        assert not flags.any()
        np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])
        need = st["max_region_count"]
        if need > 1:
            for region, ok in ((need, True), (need - 1, False)):
                ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, coarse=coarse,
                                               exchange="filtered-regions", stats={"region": region})
                assert flags.any() != ok
                if ok:
                    np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])


@pytest.mark.parametrize("world", [1, 3])
def test_two_phase_scan_on_the_matrix_cores(world):
    """tk_index_shard_scan_first_dev / _rest_dev: first lists exactly, their bound min-reduced, the
    lists behind them on the plain kernel for the queries whose bound allows — golden ids for both
    exchanges; with repeating labels beside the TWIN replay (beside the hash-set replay only on request:
    tk_index_set_plain_scan(ix, 2)), same ids."""
    from test_hip_parity import ivf_from_fixture
    from tinyknn_amd.multi_gpu import _HipShardEngine, shard_lists
    for tag in ("an100", "eu128", "an100b2"):
        g = golden(f"g6_ivf_{tag}.npz")
        ivf = ivf_from_fixture(None, g)
        sizes = np.array([0 if isinstance(t, np.ndarray) else t.size
                          for t in ivf.pq_transformed_points[:ivf.active_centers.shape[0]]], dtype=np.int64)
        owner = shard_lists(sizes, world)
        for n_probes in (5, 10):
            for exchange in ("dense", "filtered-regions"):
                engines = [_HipShardEngine(ivf, owner, r, world, 1) for r in range(world)]
                if tag.endswith("b2"):
                    # repeating labels: the plain path goes with the TWIN form of the lane replay (the default);
                    # beside the hash-set replay only on request
                    from tinyknn_amd import _lib
                    engines[0].dev.set_option(_lib.OPT_REPLAY_TWIN, 0)
                    assert not engines[0].plain_ok(10, n_probes, None)
                    engines[0].dev.set_plain_scan("always")
                    assert engines[0].plain_ok(10, n_probes, None)
                    engines[0].dev.set_plain_scan(True)
                    engines[0].dev.set_option(_lib.OPT_REPLAY_TWIN, 1)
                assert engines[0].plain_ok(10, n_probes, None)
                assert not engines[0].plain_ok(10, 1, None)                 # one list: nothing behind it
                st = {}
                ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, owner=owner,
                                               engines=engines, exchange=exchange, stats=st)
                assert st.get("two_phase") and not flags.any()
                np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])
                nq_ = len(g["qn"])
                assert all(p_["plain_queries"] == st["plain"][0]["plain_queries"] for p_ in st["plain"])
                # (the fixtures' lists are shorter than the default heap: few bounds are low enough here —
                #  the small heaps below send most queries the plain way)
                assert sum(p_["plain_pairs"] for p_ in st["plain"]) == st["plain"][0]["plain_queries"] * (n_probes - 1)
                for pass_1 in (3, 40):      # small heaps: low bounds, most queries plain; both ways
                    st = {}
                    a, _, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, pass_1=pass_1,
                                             owner=owner, exchange=exchange, stats=st)
                    if not tag.endswith("b2"):
                        assert pass_1 != 3 or st["plain"][0]["plain_queries"] >= nq_ // 2, st["plain"]
                    b, _, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, pass_1=pass_1,
                                             owner=owner, exchange=exchange, plain=False)
                    np.testing.assert_array_equal(a, b)


def test_sharded_overflow_is_flagged_and_harmless():
    """A region that is too small raises the flag on the ranks involved and nothing is
    written out of bounds; the exact capacity (largest stream) passes."""
    from test_hip_parity import ivf_from_fixture
    from tinyknn_amd.multi_gpu import shard_lists, shard_positions
    g = golden("g6_ivf_an100.npz")
    ivf = ivf_from_fixture(None, g)
    world, n_probes = 2, 5
    st = {}
    ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, capacity=4, stats=st)
    assert flags.any()
    owner = shard_lists(g["list_sizes"], world)
    chunks = (g["list_sizes"] + 15) // 16
    probes = g[f"probes_p{n_probes}"].copy()
    probes[probes < 0] += len(chunks)
    src, pos = shard_positions(probes, chunks, owner, world, 10 ** 9)
    exact = int((pos + chunks[probes]).max())
    assert st["usage"] == exact          # tk_index_shard_usage: the longest stream, fitted or not
    ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, capacity=exact)
    assert not flags.any()
    np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])
    _, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, capacity=exact - 1)
    assert flags.any()


@pytest.mark.parametrize("build_probes,world,coarse", [(1, 4, "home"), (2, 3, "home"), (1, 8, "home"),
                                                       (1, 8, "replicated")])
def test_sharded_vs_unsharded_larger(oracle, build_probes, world, coarse):
    """60k x 100 angular index, 1003 queries (not a multiple of the world size): the sharded
    pipeline returns the ids of the unsharded one (itself pinned to the oracle), for distinct
    (lane replay) and repeating labels (duplicate test); a sample is checked against the
    oracle directly."""
    from tinyknn_amd import IVF, FastPQ
    from test_hip_parity import _oracle_index
    np.random.seed(10)
    n, d, nq = 60000, 100, 1003
    cent = np.random.randn(300, d)
    X = (cent[np.random.randint(300, size=n)] + 0.7 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(300, size=nq)] + 0.7 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 244, FastPQ(2))
    ivf.fit(X[:20000]).build(X, n_probes=build_probes)
    qn, qp = ivf._prepare(qs.copy())
    ox = _oracle_index(oracle, ivf)
    for n_probes in (1, 10, 30):
        want = ivf.device_index().query_batch(qn, qp, 10, n_probes)
        ids, flags, cap = simulate_world(ivf, world, qn, qp, 10, n_probes, coarse=coarse)
        assert not flags.any(), f"default capacity {cap} overflowed"
        np.testing.assert_array_equal(ids, want)
        np.testing.assert_array_equal(ids[:60], ox.query_batch(qn[:60], 10, n_probes))


@pytest.mark.parametrize("build_probes,world", [(1, 4), (2, 3), (1, 8)])
def test_sharded_filtered_vs_unsharded_larger(oracle, build_probes, world):
    """The 60k x 100 index through the filtered exchange: ids of the unsharded pipeline for
    distinct and repeating labels, pass_1 heaps too; and the exchange is the smaller the longer
    the lists are against the heap (here ~250 rows per list)."""
    from tinyknn_amd import IVF, FastPQ
    np.random.seed(10)
    n, d, nq = 60000, 100, 1003
    cent = np.random.randn(300, d)
    X = (cent[np.random.randint(300, size=n)] + 0.7 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(300, size=nq)] + 0.7 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 244, FastPQ(2))
    ivf.fit(X[:20000]).build(X, n_probes=build_probes)
    qn, qp = ivf._prepare(qs.copy())
    from test_hip_parity import _oracle_index
    ox = _oracle_index(oracle, ivf)
    for n_probes, pass_1 in ((1, None), (10, None), (30, None), (10, 40), (5, 700)):
        want = ivf.device_index().query_batch(qn, qp, 10, n_probes, pass_1)
        st = {}
        ids, flags, cap = simulate_world(ivf, world, qn, qp, 10, n_probes, pass_1=pass_1,
                                         exchange="filtered", stats=st)
        assert not flags.any(), f"default capacity {cap} overflowed"
        np.testing.assert_array_equal(ids, want)
        assert 0 < st["records"] <= st["dense_blocks"]
        # the reduced bound IS the oracle's bound after the first probed list (fresh heap, query_pq)
        R = pass_1 if pass_1 else (n_probes + 1) * 10 + 1
        off = ox.list_chunk_off
        for i in range(0, nq, 29):
            _, dbg = ox.query(qn[i], 10, n_probes, pass_1, debug=True)
            l0 = int(dbg["probes"][0]) % ox.n_lists
            hidx, hval = np.zeros(R, np.int64), np.zeros(R, np.int32)
            oracle.init_heap(hidx, hval, True)
            codes = np.ascontiguousarray(ox.codes[off[l0]:off[l0 + 1]])
            if len(codes):
                oracle.query_pq(codes, int(ox.list_n[l0]), oracle.transform_tables(dbg["table"]), hidx, hval, True)
            assert int(st["bound"][i]) == ((int(hval[0]) & 0xff) ^ 0x80), (i, n_probes, pass_1)
        if n_probes == 10 and pass_1 == 40:
            assert 20 * st["records"] < 16 * st["dense_blocks"] * 0.6, st


def test_filtered_public_class_world1():
    """ListShardedIndex(exchange="filtered") without a process group, pipelined submit too."""
    import torch
    from test_hip_parity import ivf_from_fixture
    from tinyknn_amd.multi_gpu import ListShardedIndex
    g = golden("g6_ivf_an100.npz")
    ivf = ivf_from_fixture(None, g)
    idx = ListShardedIndex(ivf, exchange="filtered", depth=2, coalesce=2)
    np.testing.assert_array_equal(idx.query_batch(g["qs"], 10, n_probes=5), g["ids_p5"])
    assert 0 < idx.bytes_sent and idx.bytes_dense > 0
    qn, qp = ivf._prepare(np.array(g["qs"], dtype=np.float32))
    qn_t, qp_t = torch.from_numpy(qn).cuda(), torch.from_numpy(np.ascontiguousarray(qp)).cuda()
    outs = [idx.submit(qn_t, qp_t, 10, 10) for _ in range(4)]
    idx.join()
    torch.cuda.synchronize()
    nq, qh = len(qn), 2 * len(qn)
    for o in outs[1::2]:
        got = o.cpu().numpy()
        assert not got[:, -1].any()
        both = got[:, :-1].reshape(qh, 10)
        np.testing.assert_array_equal(both[:nq], g["ids_p10"])
        np.testing.assert_array_equal(both[nq:], g["ids_p10"])


def test_sharded_world1_public_class():
    """ListShardedIndex without a process group: the all-to-all is a copy."""
    from test_hip_parity import ivf_from_fixture
    from tinyknn_amd.multi_gpu import ListShardedIndex
    g = golden("g6_ivf_eu128.npz")
    ivf = ivf_from_fixture(None, g)
    idx = ListShardedIndex(ivf)
    np.testing.assert_array_equal(idx.query_batch(g["qs"], 10, n_probes=5), g["ids_p5"])
    idx.capacity[(len(g["qs"]), 10)] = 2           # overflow -> repeated with room for the longest stream
    np.testing.assert_array_equal(idx.query_batch(g["qs"], 10, n_probes=10), g["ids_p10"])
    cap = idx.capacity[(len(g["qs"]), 10)]
    assert cap > 2
    np.testing.assert_array_equal(idx.query_batch(g["qs"], 10, n_probes=10), g["ids_p10"])
    assert idx.capacity[(len(g["qs"]), 10)] == cap                         # settled
    idx.capacity[(len(g["qs"]), 5)] = 10 ** 6      # far too roomy -> trimmed to 1.35 x the longest stream seen,
    for _ in range(2):                              # on the evidence of three batches (one ran above)
        np.testing.assert_array_equal(idx.query_batch(g["qs"], 10, n_probes=5), g["ids_p5"])
    assert idx.capacity[(len(g["qs"]), 5)] < 10 ** 6 // 2


def _gloo_gpu_worker(rank, world, port, ret, exchange):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_hip_parity import ivf_from_fixture
        from tinyknn_amd.multi_gpu import ListShardedIndex
        g = golden("g6_ivf_an100b2.npz")
        ivf = ivf_from_fixture(None, g)
        idx = ListShardedIndex(ivf, depth=2, exchange=exchange)
        ret[rank] = {p: idx.query_batch(g["qs"], 10, n_probes=p) for p in (1, 5, 10)}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["dense", "filtered"])
def test_sharded_two_processes_one_gpu(exchange):
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() + len(exchange)) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gloo_gpu_worker, args=(2, port, ret, exchange), nprocs=2, join=True)
    g = golden("g6_ivf_an100b2.npz")
    for r in range(2):
        for p in (1, 5, 10):
            np.testing.assert_array_equal(ret[r][p], g[f"ids_p{p}"])


def _rccl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from test_hip_parity import ivf_from_fixture
        from tinyknn_amd.multi_gpu import ListShardedIndex
        out = {}
        for tag in ("g6_ivf_an100", "g6_ivf_an100b2"):      # distinct labels / every point in two lists
            g = golden(tag + ".npz")
            ivf = ivf_from_fixture(None, g)
            qn, qp = ivf._prepare(np.array(g["qs"], dtype=np.float32))
            qn_t, qp_t = torch.from_numpy(qn).cuda(), torch.from_numpy(np.ascontiguousarray(qp)).cuda()
            for exchange in ("dense", "filtered", "filtered-host-counts"):
                idx = ListShardedIndex(ivf, depth=4, exchange=exchange.split("-")[0], force_collectives=True,
                                       counts="host" if exchange.endswith("host-counts") else "device")
                assert idx.backend == "nccl" and idx.world == 1 and idx.force
                assert idx.counts == ("host" if exchange.endswith("host-counts") else "device")
                for p in (1, 5, 10):
                    out[(tag, exchange, p)] = idx.query_batch(g["qs"], 10, n_probes=p)
                # four batches in flight, each on its own stream: the collectives of different
                # batches are issued from different streams
                outs = [idx.submit(qn_t, qp_t, 10, 10) for _ in range(8)]
                idx.join()
                torch.cuda.synchronize()
                out[(tag, exchange, "flight")] = [o.cpu().numpy() for o in outs[-4:]]
                idx.engine.dev.close()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_sharded_rccl_world1_forced_collectives():
    """The RCCL branch on the one-GPU box: ONE rank, backend nccl, the exchanges forced through
    torch.distributed (TINYKNN_FORCE_COLLECTIVES / force_collectives) instead of device copies —
    uint8 MIN all-reduce of the bounds, equal-split all_to_all_single of the distance regions,
    variable-split all_to_all_single of the (n, 5) int32 records (counts="host") and the equal-split
    one of the fixed record regions with the counts beside them (counts="device", the default: no
    host synchronisation in a batch), all_gather_into_tensor of probe lists and ids, MAX all-reduce
    of the region sizes at join(), four batches in flight on four streams.  Golden ids, every
    exchange, distinct and repeating labels."""
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, port, ret), nprocs=1, join=True)
    out = ret[0]
    for tag in ("g6_ivf_an100", "g6_ivf_an100b2"):
        g = golden(tag + ".npz")
        nq = len(g["qs"])
        for exchange in ("dense", "filtered", "filtered-host-counts"):
            for p in (1, 5, 10):
                np.testing.assert_array_equal(out[(tag, exchange, p)], g[f"ids_p{p}"])
            for got in out[(tag, exchange, "flight")]:
                assert not got[:, -1].any()
                np.testing.assert_array_equal(got[:, :-1].reshape(-1, 10)[:nq], g["ids_p10"])


def test_resident_index_sharded_in_place(oracle):
    """IVF.build_resident on every simulated rank (same seed: same vectors, lists and codes),
    each rank's index sharded where it lies (tk_index_shard_resident): ids of the unsharded
    resident index and of the oracle over its exported lists."""
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.multi_gpu import _HipShardEngine, shard_lists
    from test_resident_build_gpu import oracle_from_resident, synth_rows
    n, d, nq, seed, world = 30011, 128, 257, 9, 3
    cent = np.random.RandomState(4).randn(20, d).astype(np.float32)
    proto = IVF("euclidean", 40, FastPQ(2))
    proto.fit(synth_rows(5000, d, seed, cent, 0.8))

    def build():
        ivf = IVF("euclidean", 40, FastPQ(2))
        ivf.all_centers, ivf.pq = proto.all_centers, proto.pq
        return ivf.build_resident(n, d, seed, cent, 0.8)

    whole = build()
    ox, _ = oracle_from_resident(oracle, whole)
    qs = synth_rows(nq, d, seed + 1, cent, 0.8)
    qn, qp = whole._prepare(qs.copy())
    owner = shard_lists(whole.list_sizes, world)
    ranks = [build() for _ in range(world)]
    engines = [_HipShardEngine(ranks[r], owner, r, world, 1, resident=True) for r in range(world)]
    for n_probes in (1, 6):
        want = whole.device_index().query_batch(qn, qp, 10, n_probes)
        np.testing.assert_array_equal(want, ox.query_batch(qn, 10, n_probes))
        ids, flags, _ = simulate_world(whole, world, qn, qp, 10, n_probes, owner=owner,
                                       engines=engines, sizes=whole.list_sizes)
        assert not flags.any()
        np.testing.assert_array_equal(ids, want)
        engines = [_HipShardEngine.__new__(_HipShardEngine) for _ in range(world)]   # closed by simulate_world
        for r, e in enumerate(engines):
            ranks[r] = build()
            e.dev = ranks[r].device_index()
            e.dev.shard_resident(owner, r, world)
            e.dev.set_pipeline(1)
            e.device = "cuda"


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("tag", G6_TAGS)
def test_one_phase_scan_golden(tag, world):
    """tk_index_shard_scan_plain_dev on the golden indexes: heads of the first lists exactly, everything
    else as plain sums, the home replay checks — ids of the fixtures, no flag raised (where the form
    does not apply — repeating labels, n_probes 1 — the call is the exact kernel's).  With the table
    limits forced down (TK_OPT_PLAIN_LIMIT) a home rank must raise bit 4 instead of returning rows it
    cannot vouch for."""
    from tinyknn_amd import _lib
    from tinyknn_amd.multi_gpu import _HipShardEngine, shard_lists
    from test_hip_parity import ivf_from_fixture
    g = golden(f"g6_ivf_{tag}.npz")
    ivf = ivf_from_fixture(None, g)
    nq = len(g["qn"])
    qh = -(-nq // world)
    for n_probes in g["probes_list"]:
        n_probes = int(n_probes)
        ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, one_phase=True)
        assert not (flags & 3).any()
        # bit 4 on a home rank = one of ITS queries met its first plain block with a bound above the
        # table's limit (short lists: the exact head is two lists and the heap is still filling; the
        # unsharded index scans such a query again, tests/test_plain_scan_gpu.py): that rank's rows are
        # not vouched for, every other rank's are the fixture's
        want = g[f"ids_p{n_probes}"]
        for h in range(world):
            if not flags[h] & 4:
                np.testing.assert_array_equal(ids[h * qh:(h + 1) * qh], want[h * qh:(h + 1) * qh])
    # forced failure: no bound is at or below a limit of -128
    n_probes = int(max(g["probes_list"]))
    L = ivf.active_centers.shape[0]
    sizes = np.array([0 if isinstance(t, np.ndarray) else t.size for t in ivf.pq_transformed_points[:L]], dtype=np.int64)
    owner = shard_lists(sizes, world)
    engines = [_HipShardEngine(ivf, owner, r, world, 1) for r in range(world)]
    for e in engines:
        e.dev.set_option(_lib.OPT_PLAIN_LIMIT, -128)
    ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, one_phase=True, engines=engines)
    if not (flags & 4).any():
        np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])      # the form did not apply: exact kernel
    assert not (flags & 3).any()
    # behind the head bounds nothing can fail: the fixture's ids, clean flag words — with the tables' own
    # limits, and with limits no bound can meet (every query then stays on the exact kernel)
    for forced in (False, True):
        for n_probes in g["probes_list"]:
            n_probes = int(n_probes)
            engines = [_HipShardEngine(ivf, owner, r, world, 1) for r in range(world)]
            for e in engines:
                e.dev.set_option(_lib.OPT_PLAIN_LIMIT, -128 if forced else 0x7fffffff)
            ids, flags, _ = simulate_world(ivf, world, g["qn"], g["qpq"], 10, n_probes, one_phase="head", engines=engines)
            assert not flags.any(), (forced, n_probes, flags)
            np.testing.assert_array_equal(ids, g[f"ids_p{n_probes}"])


@pytest.mark.parametrize("pair_nq", ["4", "100000"])
def test_simulated_peers_one_rank_of_a_partition(oracle, pair_nq, monkeypatch):
    """ListShardedIndex(simulate=SimulatedPeers(...)): ONE rank of a W-rank partition with the other
    ranks' contributions recorded from clone shards on the same device — its home rows are the
    unsharded rows, for every rank, dense (one- and two-phase) and filtered exchange, with batches in flight;
    the unsharded index stays usable beside the clones.  pair_nq: the home queries' replays by the lane kernel
    (the suite's default threshold) and by the wave-per-query register heap (every handle made in this test:
    the check of the plain scan's lemma then raises the batch's flag word from that kernel)."""
    monkeypatch.setenv("TINYKNN_PAIR_NQ", pair_nq)
    import torch
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.multi_gpu import ListShardedIndex
    from simulated_peers import SimulatedPeers
    np.random.seed(21)
    n, nq, d = 50000, 1203, 100
    cent = np.random.randn(120, d)
    X = (cent[np.random.randint(120, size=n)] + 0.6 * np.random.randn(n, d)).astype(np.float32)
    qs = (cent[np.random.randint(120, size=nq)] + 0.6 * np.random.randn(nq, d)).astype(np.float32)
    ivf = IVF("angular", 150, FastPQ(2))
    ivf.fit(X[:15000]).build(X, n_probes=1)
    qn, qp = ivf._prepare(qs.copy())
    dev = ivf.device_index()
    want = dev.query_batch(qn, qp, 10, 6)
    qn_t, qp_t = torch.from_numpy(qn).cuda(), torch.from_numpy(np.ascontiguousarray(qp)).cuda()
    for world, rank in ((4, 0), (4, 3), (8, 5)):
        peers = SimulatedPeers(ivf, world=world, rank=rank)
        lo, hi = peers.home_range(nq)
        for kw in (dict(exchange="dense"), dict(exchange="dense", plain="two-phase"), dict(exchange="filtered"),
                   dict(exchange="dense", plain=False), dict(exchange="dense", plain="head")):
            idx = ListShardedIndex(ivf, simulate=peers, depth=2, **kw)
            peers.reset()
            got = idx.query_prepared(qn_t, qp_t, 10, 6)
            np.testing.assert_array_equal(got[lo:hi], want[lo:hi])
            assert (got[:lo] == -1).all() and (got[hi:] == -1).all()        # the other ranks' rows are theirs
            outs = [idx.submit(qn_t, qp_t, 10, 6) for _ in range(5)]
            idx.join()
            torch.cuda.synchronize()
            qh = -(-nq // world)
            for o in outs:
                rows = o[:, :-1].reshape(world * qh, 10)[:nq].cpu().numpy()
                np.testing.assert_array_equal(rows[lo:hi], want[lo:hi])
                assert not o[:, -1].any()
        peers.close()
    np.testing.assert_array_equal(dev.query_batch(qn, qp, 10, 6), want)          # the lender is intact
