"""One rank of a W-rank list partition on a 1-GPU box — bench / test scaffolding, NOT part of the product package.

`ListShardedIndex(ivf, simulate=SimulatedPeers(ivf, world, rank))` takes this object in place of its collectives
(`all_gather` / `all_to_all` / `all_reduce_min` / `usage` / `context` / `live_engine`): bench.py's
`list_sharded.rank_share_W8`, tests/test_shard_gpu.py and tests/test_c5_full_size_gpu.py use it.  It lived in
tinyknn_amd/multi_gpu.py until round 6.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tinyknn_amd.multi_gpu import _HipShardEngine, shard_lists  # noqa: E402


class SimulatedPeers:
    """The OTHER ranks of a W-rank list partition, played by this device, so that ONE rank's share of a
    sharded batch — its 1/W of the lists for all queries, the replay and rescoring of its home queries,
    every exchange buffer filled as the collectives would have filled it — can be run, checked and timed
    where W RCCL ranks cannot exist (`ListShardedIndex(ivf, simulate=peers)`; bench.py's
    `list_sharded.rank_share_W8`).

    Every rank is a clone shard of the unsharded index (DeviceIndex.clone_shard: the replicated arrays
    are borrowed, only the owned codes are per rank — a 100M x 128 index keeps its 51 GB of vectors
    once).  For a batch the peers are run ONCE through the real entry points, on the same inputs, and
    what they would have put on the links is recorded: the gathered probe lists, the min-reduced bound,
    the (source -> this rank) regions of the all-to-all, the record regions and counts of the filtered
    exchange.  A "collective" of the live rank is then a device copy of the recorded contribution of
    the others (the bytes a real all-to-all would land in this rank's HBM) plus its own live part.
    The unsharded index stays usable beside it."""

    def __init__(self, ivf, world, rank, owner=None):
        import torch
        self.torch = torch
        self.ivf, self.world, self.rank = ivf, int(world), int(rank)
        self.base = ivf.device_index()
        sizes = getattr(ivf, "list_sizes", None)
        if sizes is None or getattr(ivf, "pq_transformed_points", 0) is not None:
            sizes = [0 if isinstance(t, np.ndarray) else t.size
                     for t in ivf.pq_transformed_points[:ivf.active_centers.shape[0]]]
        self.list_sizes = np.array(sizes, dtype=np.int64)
        self.owner = shard_lists(self.list_sizes, self.world) if owner is None else np.asarray(owner, dtype=np.int32)
        self._engines = {}
        self._live = None
        self._rec = {}
        self.records = 0        # recording passes so far (each runs every peer once: a second or more)

    def home_range(self, nq):
        qh = -(-nq // self.world)
        return self.rank * qh, min(nq, (self.rank + 1) * qh)

    def _engine(self, r):
        if r not in self._engines:
            self._engines[r] = _HipShardEngine(self.ivf, self.owner, r, self.world, 1,
                                               dev=self.base.clone_shard(self.owner, r, self.world))
        return self._engines[r]

    def live_engine(self, depth):
        """The timed rank's own engine (`depth` workspace slots); the recording pass uses another clone."""
        if self._live is None:
            self._live = _HipShardEngine(self.ivf, self.owner, self.rank, self.world, depth,
                                         dev=self.base.clone_shard(self.owner, self.rank, self.world))
        return self._live

    def reset(self):
        """Forget the recorded contributions (other queries follow)."""
        self._rec = {}

    def close(self):
        for e in list(self._engines.values()) + ([self._live] if self._live is not None else []):
            e.dev.close()
        self._engines, self._live, self._rec = {}, None, {}

    # -- what the ranks contribute to one batch
    def context(self, qn, qp, k, n_probes, pass_1, capacity, coarse="home", kind="dense", region=0, form="exact"):
        # recorded once per batch SHAPE and form: the live rank must keep submitting the queries the record
        # was made from (bench.py does; reset() forgets) — a fingerprint of the content would cost every
        # batch a host synchronisation, and the tensors' addresses change with every concatenation
        key = (qn.shape[0], k, n_probes, pass_1, int(capacity), coarse, kind, int(region), form)
        if key not in self._rec:
            if len(self._rec) >= 4:             # (a few batch shapes at a time: the regions are large)
                self._rec.pop(next(iter(self._rec)))
            self._rec[key] = self._record(qn, qp, k, n_probes, pass_1, int(capacity), coarse, kind, int(region), form)
        return self._rec[key]

    def _record(self, qn, qp, k, n_probes, pass_1, capacity, coarse, kind, region, form):
        t, W, me = self.torch, self.world, self.rank
        self.records += 1
        t.cuda.synchronize()
        nq = qn.shape[0]
        qh = -(-nq // W)
        kc = min(n_probes, len(self.list_sizes))
        rec = dict(usage=0)
        with t.cuda.stream(t.cuda.Stream()):
            p_all = None
            if coarse == "home":
                homes = [t.zeros(qh * kc, dtype=t.int64, device="cuda") for _ in range(W)]
                for r in range(W):
                    self._engine(r).coarse(0, qn, qp, k, n_probes, pass_1, homes[r])
                p_all = t.cat(homes).contiguous()
            rec["p_all"] = p_all
            flag = t.zeros(1, dtype=t.int32, device="cuda")
            bound = None
            mine = t.empty((W, capacity * 16), dtype=t.uint8, device="cuda")        # (source s -> this rank)
            sends = {}
            if form == "two" or kind == "filtered":
                firsts = []
                for r in range(W):
                    sends[r] = t.empty((W, capacity * 16), dtype=t.uint8, device="cuda")
                    b_ = t.zeros(nq, dtype=t.uint8, device="cuda")
                    if form == "two":
                        self._engine(r).scan_first(0, qn, qp, k, n_probes, pass_1, capacity, sends[r], flag, b_,
                                                   probes_all=p_all)
                    else:
                        self._engine(r).scan(0, qn, qp, k, n_probes, pass_1, capacity, sends[r], flag, probes_all=p_all)
                        self._engine(r).bound(0, qn, k, n_probes, pass_1, capacity, sends[r].view(-1), b_)
                    firsts.append(b_)
                bound = t.stack(firsts).min(dim=0).values.contiguous()
                if form == "two":
                    for r in range(W):
                        self._engine(r).scan_rest(0, qn, k, n_probes, pass_1, capacity, sends[r], bound)
                for r in range(W):
                    mine[r].copy_(sends[r][me])
                    rec["usage"] = max(rec["usage"], 0)
            elif form == "head":
                firsts = []
                for r in range(W):
                    sends[r] = t.empty((W, capacity * 16), dtype=t.uint8, device="cuda")
                    b_ = t.zeros(nq, dtype=t.uint8, device="cuda")
                    self._engine(r).scan_head(0, qn, qp, k, n_probes, pass_1, capacity, sends[r], flag, b_, probes_all=p_all)
                    firsts.append(b_)
                bound = t.stack(firsts).min(dim=0).values.contiguous()
                for r in range(W):
                    self._engine(r).scan_plain(0, qn, qp, k, n_probes, pass_1, capacity, sends[r], flag, probes_all=p_all,
                                               bound=bound)
                    mine[r].copy_(sends[r][me])
                sends = {}
            else:
                buf = t.empty((W, capacity * 16), dtype=t.uint8, device="cuda")
                for r in range(W):
                    eng = self._engine(r)
                    (eng.scan_plain if form == "one" else eng.scan)(0, qn, qp, k, n_probes, pass_1, capacity, buf, flag,
                                                                     probes_all=p_all)
                    mine[r].copy_(buf[me])
            rec["bound"] = bound
            rec["recv"] = mine
            if kind == "filtered":
                assert region > 0, "simulated peers: the filtered exchange with counts=\"device\""
                rrec = t.empty((W * region, 5), dtype=t.int32, device="cuda")
                rcnt = t.zeros(W, dtype=t.int32, device="cuda")
                for r in range(W):
                    c = t.zeros(3 * W, dtype=t.int32, device="cuda")
                    rr = t.empty((W * region, 5), dtype=t.int32, device="cuda")
                    self._engine(r).filter_regions(0, qn, k, n_probes, pass_1, capacity, sends[r].view(-1), bound, c, rr,
                                                   region, flag)
                    rrec[r * region:(r + 1) * region].copy_(rr[me * region:(me + 1) * region])
                    rcnt[r] = c[me]
                rec["rrec"], rec["rcounts"] = rrec, rcnt
            t.cuda.current_stream().synchronize()
            rec["usage"] = max(int(self._engine(r).usage(0)) for r in range(W))
            rec["flag"] = int(flag.item())
        t.cuda.synchronize()
        return rec

    # -- the live rank's "collectives"
    def all_gather(self, what, ctx, out, inp):
        n = inp.numel()
        if what == "probes":
            out.copy_(ctx["p_all"])
        # (ids: the other ranks' rows were written when the buffer was made, _buffers; their flag words — the last
        #  element of every rank's row — are what the recording pass saw: an overflow of a PEER's region makes the live
        #  rank grow its capacity and record again, as the gathered flag of a real world would)
        elif what == "ids":
            out.view(self.world, n)[:, -1] = int(ctx.get("flag", 0)) & 3
        out.view(self.world, n)[self.rank].copy_(inp)

    def all_to_all(self, what, ctx, recv, send):
        W, me = self.world, self.rank
        src = ctx["recv"] if what == "segments" else ctx["rcounts"] if what == "counts" else ctx["rrec"]
        recv.view(-1).copy_(src.view(-1))
        n = recv.numel() // W
        recv.view(W, n)[me].copy_(send.view(W, n)[me])

    def all_reduce_min(self, ctx, t_):
        self.torch.minimum(t_, ctx["bound"], out=t_)

    def usage(self, ctx):
        return ctx["usage"]
