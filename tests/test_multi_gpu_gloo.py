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


def test_shard_bounds():
    from tinyknn_amd.multi_gpu import shard_bounds
    for nq in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(nq, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nq
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
