"""N > 1 path on CPU: two gloo ranks, queries sharded, result all-gather.  The
per-rank compute engine is the CPU oracle here (tests may use it); on a GPU box
the default engine is the HIP DeviceIndex."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tag, nq, k, n_probes, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import golden
        from test_oracle_golden import load_oracle_index
        from oracle import oracle as O
        from tinyknn_amd.multi_gpu import ReplicaGroup
        g = golden(f"g6_ivf_{tag}.npz")
        ox = load_oracle_index(O, g)

        class HostSide:     # the part of IVF the replica group needs: _prepare
            def _prepare(self, qs):
                return qs, ox.pq_query(qs)

        def engine(qn, qp, k, n_probes, pass_1):
            return ox.query_batch(qn, k, n_probes, pass_1)

        grp = ReplicaGroup(HostSide(), engine=engine)
        out = grp.query_batch(g["qn"][:nq], k, n_probes)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nq", [24, 7, 1])
def test_replica_group_two_ranks(nq):
    import torch.multiprocessing as mp
    from conftest import golden
    tag, k, n_probes = "an100", 10, 5
    port = 29500 + (os.getpid() + nq) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, tag, nq, k, n_probes, ret), nprocs=2, join=True)
    g = golden(f"g6_ivf_{tag}.npz")
    exp = g[f"ids_p{n_probes}"][:nq]
    for r in range(2):
        np.testing.assert_array_equal(ret[r], exp)


def _shard_worker(rank, world, port, tag, nq, k, n_probes, tiny, coarse, ret, exchange="dense", counts="device"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import golden
        from test_oracle_golden import load_oracle_index
        from shard_cpu_engine import OracleShardEngine
        from oracle import oracle as O
        from tinyknn_amd.multi_gpu import ListShardedIndex, shard_lists
        g = golden(f"g6_ivf_{tag}.npz")
        ox = load_oracle_index(O, g)

        class HostSide:
            def _prepare(self, qs):
                return qs, ox.pq_query(qs)

        owner = shard_lists(g["list_sizes"], world)
        eng = OracleShardEngine(O, ox, owner, rank, world)
        idx = ListShardedIndex(HostSide(), engine=eng, owner=owner, list_sizes=g["list_sizes"],
                               coarse=coarse, exchange=exchange, counts=counts)
        assert idx.counts == counts
        if tiny:
            idx.capacity[(nq, n_probes)] = 3      # overflows: the batch must be repeated
        if tiny == "region":
            idx.capacity.pop((nq, n_probes))
            idx.record_region[(nq, n_probes)] = 2    # the record regions overflow, the streams fit
        out = idx.query_batch(g["qn"][:nq], k, n_probes)
        if exchange == "filtered" and counts == "device":
            out2 = idx.query_batch(g["qn"][:nq], k, n_probes)
            assert (out2 == out).all()
            out2 = idx.query_batch(g["qn"][:nq], k, n_probes)      # third look: regions trimmed
            assert (out2 == out).all() and (nq, n_probes) in idx.record_region
        ret[rank] = (out, idx.capacity[(nq, n_probes)], getattr(eng, "coarse_calls", 0),
                     idx.bytes_sent, idx.bytes_dense)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nq,tiny,coarse", [(2, 24, False, "home"), (2, 7, True, "home"),
                                                  (3, 24, False, "home"), (2, 1, False, "home"),
                                                  (3, 23, False, "home"), (2, 24, False, "replicated"),
                                                  (3, 7, True, "replicated")])
def test_list_sharded_index_gloo(world, nq, tiny, coarse):
    """Lists sharded by cluster id over `world` gloo ranks: the probe lists of the home queries
    are all-gathered (coarse="home": checked against every rank's own derivation by the CPU
    engine), the all-to-all carries every segment to the right place of the right home rank
    (checked byte for byte), overflow repeats the batch, ids equal the reference's on every rank."""
    import torch.multiprocessing as mp
    from conftest import golden
    tag, k, n_probes = "an100", 10, 5
    port = 31500 + (os.getpid() * 7 + nq + world) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, port, tag, nq, k, n_probes, tiny, coarse, ret), nprocs=world,
             join=True)
    g = golden(f"g6_ivf_{tag}.npz")
    exp = g[f"ids_p{n_probes}"][:nq]
    for r in range(world):
        np.testing.assert_array_equal(ret[r][0], exp)
        assert not tiny or ret[r][1] > 3
        assert (ret[r][2] > 0) == (coarse == "home")


@pytest.mark.parametrize("counts", ["device", "host"])
@pytest.mark.parametrize("world,nq,tiny,coarse,n_probes", [(2, 24, False, "home", 5), (3, 23, False, "home", 10),
                                                           (2, 7, True, "replicated", 5), (2, 1, False, "home", 5),
                                                           (2, 24, "region", "home", 5)])
def test_list_sharded_filtered_exchange_gloo(world, nq, tiny, coarse, n_probes, counts):
    """exchange="filtered": the bound after the first probed list is min-reduced over the ranks
    (the CPU engine replays that list with the oracle's query_pq), the records travel — with
    variable splits read on the host (counts="host"), or in fixed regions with equal splits and the
    counts beside them (counts="device": nothing is read on the host inside a batch; "region":
    regions of 2 records overflow, the batch is repeated with larger ones) — and the home rank's
    engine checks that exactly the blocks the rule names arrived, byte for byte; ids equal the
    reference's."""
    if tiny == "region" and counts == "host":
        pytest.skip("regions belong to counts='device'")
    import torch.multiprocessing as mp
    from conftest import golden
    tag, k = "an100", 10
    port = 35500 + (os.getpid() * 5 + nq + world + n_probes) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, port, tag, nq, k, n_probes, tiny, coarse, ret, "filtered", counts),
             nprocs=world, join=True)
    g = golden(f"g6_ivf_{tag}.npz")
    exp = g[f"ids_p{n_probes}"][:nq]
    for r in range(world):
        np.testing.assert_array_equal(ret[r][0], exp)
        assert tiny is not True or ret[r][1] > 3
    sent = sum(ret[r][3] for r in range(world))
    dense = sum(ret[r][4] for r in range(world))
    assert dense > 0 and sent > 0      # (the fixture's lists are shorter than the heap: little to drop)


def test_shard_lists_and_capacity():
    from tinyknn_amd.multi_gpu import shard_capacity, shard_lists, shard_positions
    rng = np.random.RandomState(3)
    sizes = rng.randint(0, 4000, size=300)
    chunks = (sizes + 15) // 16
    for world in (1, 2, 8):
        owner = shard_lists(sizes, world)
        load = np.bincount(owner, weights=chunks, minlength=world)
        assert load.max() - load.min() <= chunks.max()
        np.testing.assert_array_equal(owner, shard_lists(sizes, world))    # deterministic
        cap = shard_capacity(sizes, owner, world, 1000, 10)
        assert 1 <= cap <= -(-1000 // world) * 10 * chunks.max()
        # size-biased random probes fit the default capacity
        probes = rng.choice(300, size=(1000, 10), p=chunks / chunks.sum())
        src, pos = shard_positions(probes, chunks, owner, world, cap)
        assert (pos >= 0).all()
        # segments of one (source, home) stream tile its region without gaps or overlap
        qh = -(-1000 // world)
        for h in range(world):
            for s in range(world):
                m = src[h * qh:(h + 1) * qh] == s
                st = pos[h * qh:(h + 1) * qh][m]
                ln = chunks[probes[h * qh:(h + 1) * qh]][m]
                np.testing.assert_array_equal(st, np.cumsum(ln) - ln)


def test_shard_bounds():
    from tinyknn_amd.multi_gpu import shard_bounds
    for nq in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(nq, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nq
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _coalesce_worker(rank, world, port, tag, nq, k, n_probes, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import golden
        from test_oracle_golden import load_oracle_index
        from shard_cpu_engine import OracleShardEngine
        from oracle import oracle as O
        from tinyknn_amd.multi_gpu import ListShardedIndex, shard_lists
        g = golden(f"g6_ivf_{tag}.npz")
        ox = load_oracle_index(O, g)
        owner = shard_lists(g["list_sizes"], world)
        eng = OracleShardEngine(O, ox, owner, rank, world)
        idx = ListShardedIndex(None, engine=eng, owner=owner, list_sizes=g["list_sizes"], coalesce=2)
        qn = torch.from_numpy(np.ascontiguousarray(g["qn"][:nq]))
        qp = torch.from_numpy(np.ascontiguousarray(ox.pq_query(g["qn"][:nq])))
        first = idx.submit(qn, qp, k, n_probes)             # waits for a second batch
        out = idx.submit(qn.flip(0).contiguous(), qp.flip(0).contiguous(), k, n_probes)
        third = idx.submit(qn, qp, k, n_probes)             # flushed alone by join()
        tail = idx.join()                                   # ... which hands its rows back
        assert tail is idx.last_flushed
        qh1 = -(-nq // world)
        tail_rows = tail[:, :-1].reshape(world * qh1, k)[:nq].numpy().copy()
        # a batch with other arguments never shares an exchange with the queued one
        a = idx.submit(qn, qp, k, n_probes)
        b = idx.submit(qn, qp, k, 2)                        # flushes `a` alone first, then waits
        assert a is None and b is None
        alone = idx.last_flushed[:, :-1].reshape(world * qh1, k)[:nq].numpy().copy()
        other = idx.join()[:, :-1].reshape(world * qh1, k)[:nq].numpy().copy()
        # a submit()ted batch that overflows its regions is reported by join(), which grows the
        # capacity; the batch submitted again is then answered
        idx2 = ListShardedIndex(None, engine=eng, owner=owner, list_sizes=g["list_sizes"])
        idx2.capacity[(nq, n_probes)] = 3
        idx2.submit(qn, qp, k, n_probes)
        raised = False
        try:
            idx2.join()
        except RuntimeError as e:
            raised = "overflowed" in str(e)
        assert raised and idx2.capacity[(nq, n_probes)] > 3
        while True:
            again = idx2.submit(qn, qp, k, n_probes)
            try:
                idx2.join()
                break
            except RuntimeError:
                pass
        again_rows = again[:, :-1].reshape(world * qh1, k)[:nq].numpy().copy()
        qh = -(-2 * nq // world)
        ret[rank] = (first is None, third is None, out[:, :-1].reshape(world * qh, k)[:2 * nq].numpy().copy(),
                     bool(out[:, -1].any()), tail_rows, alone, other, again_rows)
    finally:
        dist.destroy_process_group()


def test_list_sharded_coalesced_submits_gloo():
    """Two consecutive submits answered as ONE sharded batch (coalesce=2): rows of both, in
    order; an odd batch left over is flushed by join()."""
    import torch.multiprocessing as mp
    from conftest import golden
    tag, k, n_probes, nq, world = "an100", 10, 5, 11, 2
    port = 35500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_coalesce_worker, args=(world, port, tag, nq, k, n_probes, ret), nprocs=world, join=True)
    exp = golden(f"g6_ivf_{tag}.npz")[f"ids_p{n_probes}"][:nq]
    exp2 = golden(f"g6_ivf_{tag}.npz")["ids_p2"][:nq]
    for r in range(world):
        first_none, third_none, rows, overflow, tail_rows, alone, other, again_rows = ret[r]
        np.testing.assert_array_equal(again_rows, exp)      # after the overflow join() reported
        assert first_none and third_none and not overflow
        np.testing.assert_array_equal(rows[:nq], exp)
        np.testing.assert_array_equal(rows[nq:], exp[::-1])
        np.testing.assert_array_equal(tail_rows, exp)       # join() returns the batch it flushed
        np.testing.assert_array_equal(alone, exp)           # flushed when the arguments changed
        np.testing.assert_array_equal(other, exp2)


def _one_phase_worker(rank, world, port, tag, nq, k, n_probes, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import golden
        from test_oracle_golden import load_oracle_index
        from shard_cpu_engine import OracleShardEngine
        from oracle import oracle as O
        from tinyknn_amd.multi_gpu import ListShardedIndex, shard_lists
        g = golden(f"g6_ivf_{tag}.npz")
        ox = load_oracle_index(O, g)

        class HostSide:
            def _prepare(self, qs):
                return qs, ox.pq_query(qs)

        owner = shard_lists(g["list_sizes"], world)
        eng = OracleShardEngine(O, ox, owner, rank, world)
        idx = ListShardedIndex(HostSide(), engine=eng, owner=owner, list_sizes=g["list_sizes"])
        assert idx._one_phase
        out = idx.query_batch(g["qn"][:nq], k, n_probes)
        calls_ok = eng.plain_calls
        # ONE rank's home replay reports a query the plain sums do not cover: the flag travels with the
        # ids, EVERY rank repeats the batch in the two-phase form and keeps that form for these arguments
        if rank == world - 1:
            eng.fail_plain = 1
        out2 = idx.query_batch(g["qn"][:nq], k, n_probes)
        failed = (k, n_probes, None) in idx._plain_failed
        rest = getattr(eng, "head_calls", 0)
        out3 = idx.query_batch(g["qn"][:nq], k, n_probes)
        # ... and a submit()ted batch in flight: join() raises, the batch submitted again is answered
        idx2 = ListShardedIndex(None, engine=eng, owner=owner, list_sizes=g["list_sizes"])
        qn = torch.from_numpy(np.ascontiguousarray(g["qn"][:nq]))
        qp = torch.from_numpy(np.ascontiguousarray(ox.pq_query(g["qn"][:nq])))
        if rank == 0:
            eng.fail_plain = 1
        idx2.submit(qn, qp, k, n_probes)
        raised = ""
        try:
            idx2.join()
        except RuntimeError as e:
            raised = str(e)
        again = idx2.submit(qn, qp, k, n_probes)
        idx2.join()
        qh = -(-nq // world)
        ret[rank] = (out, out2, out3, calls_ok, failed, rest, eng.plain_calls, raised,
                     again[:, :-1].reshape(world * qh, k)[:nq].numpy().copy(), bool(again[:, -1].any()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_list_sharded_one_phase_scan_and_its_flag_gloo(world):
    """Dense exchange, scan in one phase (tk_index_shard_scan_plain_dev's protocol): no bound
    all-reduce; a home query that fails the replay's check raises bit 4 of the batch's flag word on
    ONE rank, the all-gathered word switches every rank to the scan behind the head bounds and the batch is
    answered again — synchronously (query_batch) and for batches in flight (submit / join)."""
    import torch.multiprocessing as mp
    from conftest import golden
    tag, k, n_probes, nq = "an100", 10, 5, 19
    port = 37500 + (os.getpid() * 3 + world) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_one_phase_worker, args=(world, port, tag, nq, k, n_probes, ret), nprocs=world, join=True)
    exp = golden(f"g6_ivf_{tag}.npz")[f"ids_p{n_probes}"][:nq]
    for r in range(world):
        out, out2, out3, calls_ok, failed, rest, plain_calls, raised, again, again_flag = ret[r]
        for o in (out, out2, out3, again):
            np.testing.assert_array_equal(o, exp)
        assert calls_ok == 1 and failed and rest >= 1       # one-phase, then the repeat went behind the head bounds
        assert plain_calls == 3                             # (+1 failed attempt, +1 idx2's first submit; none after)
        assert "submit the batches again" in raised and not again_flag


def _mismatch_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tinyknn_amd.multi_gpu import ListShardedIndex
        sizes = np.array([40, 17, 99, 3 + rank])        # rank 1 built a different index

        class Eng:
            device = "cpu"

        try:
            ListShardedIndex(object(), engine=Eng(), list_sizes=sizes)
            ret[rank] = "accepted"
        except RuntimeError as e:
            ret[rank] = str(e)
    finally:
        dist.destroy_process_group()


def test_list_sharded_index_rejects_different_indexes():
    """Positions in the exchange are computed from replicated state, not transmitted: ranks whose
    indexes differ (an unseeded fit per rank) must be told at construction, on every rank."""
    import torch.multiprocessing as mp
    port = 37500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mismatch_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        assert "different indexes" in ret[r]


def test_exchange_auto_rule():
    """exchange="auto": filtered where the size-weighted mean list holds >= 32 heaps' worth of rows
    (replicated values only: every rank takes the same branch)."""
    from tinyknn_amd.multi_gpu import ListShardedIndex

    class Eng:
        device = "cpu"

    short = ListShardedIndex(object(), engine=Eng(), list_sizes=np.full(1087, 1100), exchange="auto")
    assert short._exchange_kind(10, 10, None) == "dense"            # heap 111: 1100 < 32 * 111
    assert short._exchange_kind(1, 1, None) == "filtered"           # heap 3
    long_ = ListShardedIndex(object(), engine=Eng(), list_sizes=np.full(10000, 10000), exchange="auto")
    assert long_._exchange_kind(10, 10, None) == "filtered"
    assert long_._exchange_kind(10, 10, 5000) == "dense"            # pass_1 = 5000 rows per heap
    assert ListShardedIndex(object(), engine=Eng(), list_sizes=[5], exchange="dense")._exchange_kind(1, 1, None) == "dense"
