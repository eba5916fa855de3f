"""Multi-GPU serving: one process per GPU.  Two modes.

ReplicaGroup — replicas of the index, queries sharded (no data-path collective).
ListShardedIndex — inverted lists sharded by cluster id, queries broadcast, one
all-to-all of int8 distance bytes per batch + the all-gather of the ids (below).

Replica mode:

The path shards by QUERY: every rank holds the whole index in its own HBM (the
GloVe-100 index is 0.5 GB, the 100M x 128 one 53 GB — both far below 288 GB) and
answers a contiguous slice of the batch with the single-GPU pipeline, so the data
path needs no collective.  The only exchange is the optional all-gather of the
(nq, k) result ids (80 B per query over RCCL/xGMI) when every rank wants the full
answer.  The reference has no multi-device code; this is new design.

`engine` is the per-rank compute callable; the default is the HIP DeviceIndex of
the given IVF.  Tests inject a CPU engine to exercise the sharding and gather
logic under gloo without a GPU.
"""
import numpy as np


class _Everything:
    """A set that holds everything (ListShardedIndex(plain="head"): every argument tuple has "failed")."""

    def __contains__(self, x):
        return True

    def add(self, x):
        pass

    def __ior__(self, other):
        return self

    def __iter__(self):
        return iter(())


def shard_bounds(nq, world, rank):
    """Balanced contiguous slice [lo, hi) of nq queries for `rank`."""
    base, extra = divmod(nq, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ReplicaGroup:
    def __init__(self, ivf, group=None, engine=None):
        import torch.distributed as dist
        self.ivf = ivf
        self.group = group
        self.dist = dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if engine is None:
            dev = ivf.device_index()          # raises if the HIP library / GPU is missing
            engine = dev.query_batch
        self.engine = engine

    def query_shard(self, qs, k, n_probes=1, pass_1=None):
        """This rank's slice of the batch: returns (lo, hi, ids (hi-lo, k))."""
        qs = np.array(qs, dtype=np.float32, order="C", copy=True)
        lo, hi = shard_bounds(len(qs), self.world, self.rank)
        qn, qp = self.ivf._prepare(qs[lo:hi])
        ids = self.engine(qn, qp, k, n_probes, pass_1) if hi > lo else np.zeros((0, k), np.int64)
        return lo, hi, ids

    def query_batch(self, qs, k, n_probes=1, pass_1=None):
        """Every rank passes the same (nq, d) batch and receives all (nq, k) ids."""
        import torch
        lo, hi, ids = self.query_shard(qs, k, n_probes, pass_1)
        if self.world == 1:
            return ids
        nq = len(qs)
        width = -(-nq // self.world)               # largest shard
        buf = np.full((width, k), -1, dtype=np.int64)
        buf[:hi - lo] = ids
        t = torch.from_numpy(buf)
        if self.dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        res = np.empty((nq, k), dtype=np.int64)
        for r, o in enumerate(out):
            a, b = shard_bounds(nq, self.world, r)
            res[a:b] = o.cpu().numpy()[:b - a]
        return res


# ---------------------------------------------------------------------------
# list-sharded index (C ABI: tk_index_set_lists_shard,
# tk_index_shard_scan_dev, tk_index_shard_finish_dev)

def shard_lists(list_sizes, world):
    """owner (n_lists,) int32: lists partitioned by cluster id, size-balanced (longest
    list first onto the least-loaded rank; ties -> lowest list id / lowest rank), so that
    every rank computes the same partition from the replicated list sizes."""
    chunks = (np.asarray(list_sizes, dtype=np.int64) + 15) // 16
    owner = np.zeros(len(chunks), dtype=np.int32)
    load = np.zeros(world, dtype=np.int64)
    for l in np.argsort(-chunks, kind="stable"):
        r = int(np.argmin(load))
        owner[l] = r
        load[r] += chunks[l]
    return owner


def shard_capacity(list_sizes, owner, world, nq, n_probes, slack=1.5):
    """uint4 (16 distances) per (source, home) region of the all-to-all buffers.
    Expected stream length = queries per home x probes x size-biased mean list length x
    the source's share of the codes, times `slack`; a batch that overflows raises the
    device flag and is repeated with twice the capacity (exactness never depends on it)."""
    chunks = (np.asarray(list_sizes, dtype=np.float64) + 15) // 16
    tot = max(chunks.sum(), 1.0)
    per_probe = (chunks * chunks).sum() / tot          # E[chunks | list hit with p ~ size]
    share = max(np.bincount(owner, weights=chunks, minlength=world).max() / tot, 1.0 / world)
    qh = -(-nq // world)
    kc = min(n_probes, len(chunks))
    worst = qh * kc * int(chunks.max() if len(chunks) else 1)
    want = int(slack * qh * kc * per_probe * share) + 1024
    return int(max(1, min(want, worst)))


def shard_positions(probes, chunks, owner, world, capacity):
    """Host restatement of shard_positions_kernel (shard.hip), used by tests and CPU
    engines.  probes (nq, S) list ids (already wrapped to >= 0); chunks (n_lists,).
    Returns (src, pos): owner rank and offset in uint4 units inside the (src -> home)
    region of every (query, slot) segment, or pos = -1 where the region overflowed."""
    nq, S = probes.shape
    qh = -(-nq // world)
    src = owner[probes]
    n = chunks[probes].astype(np.int64)
    pos = np.full((nq, S), -1, dtype=np.int64)
    for h in range(world):
        blk = slice(h * qh, min(nq, (h + 1) * qh))
        for s_ in range(world):
            m = src[blk] == s_
            c = np.where(m, n[blk], 0).ravel()
            start = np.cumsum(c) - c
            ok = m.ravel() & (start + c <= capacity)
            pv = pos[blk].ravel()
            pv[ok] = start[ok]
            pos[blk] = pv.reshape(pos[blk].shape)
    return src, pos


class _HipShardEngine:
    """scan / finish on the MI355X (torch tensors carry the device buffers)."""

    def __init__(self, ivf, owner, rank, world, depth, resident=False, dev=None):
        from .ivf import DeviceIndex
        if dev is not None:
            self.dev = dev              # e.g. DeviceIndex.clone_shard: another rank's shard on this device
        elif resident:
            # the index was built in HBM (IVF.build_resident): shard it where it lies
            self.dev = ivf.device_index()
            self.dev.shard_resident(owner, rank, world)
        else:
            self.dev = DeviceIndex(ivf, owner=owner, rank=rank, world=world)
        self.dev.set_pipeline(depth)
        self.device = "cuda"

    def coarse(self, slot, qn, qp, k, n_probes, pass_1, probes_home):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_coarse_dev(slot, qn.data_ptr(), qp.data_ptr(), qp.dtype == torch.float64,
                                  qn.shape[0], k, n_probes, pass_1, probes_home.data_ptr(), stream=st)

    @property
    def table_bytes(self):
        return self.dev.M * 16

    def scan(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_scan_dev(slot, qn.data_ptr(), qp.data_ptr(), qp.dtype == torch.float64,
                                qn.shape[0], k, n_probes, pass_1, capacity, send.data_ptr(),
                                flag.data_ptr(), stream=st,
                                probes_all_ptr=None if probes_all is None else probes_all.data_ptr())

    # the scan in two phases, the second on the matrix cores (tk_index_shard_scan_first_dev / _rest_dev)
    def plain_ok(self, k, n_probes, pass_1):
        return self.dev.shard_plain(k, n_probes, pass_1)

    def scan_first(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, bound, probes_all=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_scan_first_dev(slot, qn.data_ptr(), qp.data_ptr(), qp.dtype == torch.float64,
                                      qn.shape[0], k, n_probes, pass_1, capacity, send.data_ptr(),
                                      flag.data_ptr(), bound.data_ptr(), stream=st,
                                      probes_all_ptr=None if probes_all is None else probes_all.data_ptr())

    def scan_rest(self, slot, qn, k, n_probes, pass_1, capacity, send, bound):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_scan_rest_dev(slot, qn.shape[0], k, n_probes, pass_1, capacity, send.data_ptr(),
                                     bound.data_ptr(), stream=st)

    # ... in ONE phase, heads exactly and the rest on the matrix cores, checked by the home replay
    # (tk_index_shard_scan_plain_dev)
    def scan_plain(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=None, bound=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_scan_plain_dev(slot, qn.data_ptr(), qp.data_ptr(), qp.dtype == torch.float64,
                                      qn.shape[0], k, n_probes, pass_1, capacity, send.data_ptr(),
                                      flag.data_ptr(), stream=st,
                                      probes_all_ptr=None if probes_all is None else probes_all.data_ptr(),
                                      bound_ptr=None if bound is None else bound.data_ptr())

    # ... behind one byte per query: heads of the first lists + the bound after them (tk_index_shard_scan_head_dev)
    def scan_head(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, bound, probes_all=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_scan_head_dev(slot, qn.data_ptr(), qp.data_ptr(), qp.dtype == torch.float64,
                                     qn.shape[0], k, n_probes, pass_1, capacity, send.data_ptr(),
                                     flag.data_ptr(), bound.data_ptr(), stream=st,
                                     probes_all_ptr=None if probes_all is None else probes_all.data_ptr())

    def finish(self, slot, qn, k, n_probes, pass_1, capacity, recv, out_home, flag=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_finish_dev(slot, qn.data_ptr(), qn.shape[0], k, n_probes, pass_1, capacity,
                                  recv.data_ptr(), out_home.data_ptr(), stream=st,
                                  flag_ptr=None if flag is None else flag.data_ptr())

    def usage(self, slot):
        return self.dev.shard_usage(slot)

    # filtered exchange (tk_index_shard_bound_dev / _filter_dev / _finish_filtered_dev)
    def bound(self, slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_bound_dev(slot, qn.shape[0], k, n_probes, pass_1, capacity,
                                 scan_buf.data_ptr(), bound.data_ptr(), stream=st)

    def filter(self, slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound, counts, records):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_filter_dev(slot, qn.shape[0], k, n_probes, pass_1, capacity,
                                  scan_buf.data_ptr(), bound.data_ptr(), counts.data_ptr(),
                                  records.data_ptr(), stream=st)

    def finish_filtered(self, slot, qn, k, n_probes, pass_1, records, n_records, out_home, flag):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_finish_filtered_dev(slot, qn.data_ptr(), qn.shape[0], k, n_probes, pass_1,
                                           records.data_ptr(), n_records, out_home.data_ptr(),
                                           flag.data_ptr(), stream=st)


    # ... without the host synchronisation: fixed regions, counts read on the device
    def filter_regions(self, slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound, counts, records,
                       region, flag, acc=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_filter_regions_dev(slot, qn.shape[0], k, n_probes, pass_1, capacity,
                                          scan_buf.data_ptr(), bound.data_ptr(), counts.data_ptr(),
                                          records.data_ptr(), region, flag.data_ptr(),
                                          acc_ptr=None if acc is None else acc.data_ptr(), stream=st)

    def finish_regions(self, slot, qn, k, n_probes, pass_1, records, counts_recv, region, out_home, flag):
        import torch
        st = torch.cuda.current_stream().cuda_stream
        self.dev.shard_finish_regions_dev(slot, qn.data_ptr(), qn.shape[0], k, n_probes, pass_1,
                                          records.data_ptr(), counts_recv.data_ptr(), region,
                                          out_home.data_ptr(), flag.data_ptr(), stream=st)


class ListShardedIndex:
    """One rank of an IVF index whose inverted lists are sharded by cluster id.

    Every rank passes the same batch; query i lives on its home rank i // ceil(nq/world).
    Per batch and rank: distance tables for all queries, the coarse stage for the HOME
    queries, all-gather of the probe lists (nq x n_probes x 8 B), scan of the owned
    (query, list) segments straight into the send buffer, ONE all-to-all of int8 distance
    bytes (1/26-1/16 of the code bytes scanned), exact heap replay + rescoring of the home
    queries, all-gather of the ids.  `coarse="replicated"` runs the coarse stage for all
    queries on every rank instead (no probe all-gather, but a third of a batch's work is then
    not divided by the world size).  Results are identical to the unsharded index (and to the
    reference) by construction: the home rank replays the same distance rows in the same order.

    `exchange="auto"` picks per call (see _exchange_kind).  `exchange="filtered"`: the bound never increases from one 16-code block to the
    next, so after the query's FIRST probed list — replayed by its owner, the bound min-reduced
    over the ranks, 1 byte per query — only the blocks of the later lists with a distance below
    that bound can matter; they travel as (destination, 16 bytes) records with the splits the
    ranks exchange first, and the home rank replays rows in which every other block holds the
    largest value.  Same ids; a fraction of the bytes (bytes_sent / bytes_dense count them).
    `counts="device"` (default where the engine has filter_regions / finish_regions): the records
    of a home rank go to a fixed region of `record_region` records, so the all-to-all has equal
    splits and is enqueued without reading a count — the counts travel beside it and the home rank
    reads them on the device; NO host synchronisation in a batch.  Regions start at the region
    capacity of the dense exchange (which can never overflow) and are trimmed to 1.35 x the largest
    count seen wherever the host looks at a batch anyway (query_prepared, join); an overflow is the
    batch's overflow flag, handled as for `capacity`.  `counts="host"`: exact variable splits, one
    host synchronisation per batch (the round-2 form).

    `plain=True` (default; where the engine has scan_first / scan_rest and says plain_ok): the scan
    in two phases — the first probed lists exactly, their bound B1 min-reduced over the ranks (one
    byte per query; the same reduction the filtered exchange needs, which then does not repeat it),
    the lists behind them on the int8 matrix cores for every query whose B1 is at most the limit of
    its table (plain_scan.hip's lemma), exactly for the others.  Same ids.

    `engine`: object with coarse/scan/finish (default: the HIP engine); tests inject a CPU one.
    """

    def __init__(self, ivf, group=None, engine=None, depth=1, owner=None, list_sizes=None,
                 coarse="home", coalesce=1, exchange="dense", calibrate=True, force_collectives=None,
                 counts="device", plain=True, simulate=None):
        import os
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ivf, self.group = ivf, group
        # simulate: a SimulatedPeers — this object is then ONE rank of peers.world whose collectives
        # are answered locally with what the other ranks would have contributed (no process group)
        self._sim = simulate
        on = dist.is_available() and dist.is_initialized() and simulate is None
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.backend = dist.get_backend(group) if on else None
        if simulate is not None:
            self.world, self.rank = simulate.world, simulate.rank
            owner = simulate.owner
            engine = simulate.live_engine(depth)
        # force_collectives (or TINYKNN_FORCE_COLLECTIVES=1): with ONE rank the exchanges are
        # device copies; forced, they go through torch.distributed all the same — on a one-GPU box
        # that is how the RCCL code path (uint8 MIN all-reduce, equal- and variable-split
        # all_to_all_single, all_gather_into_tensor, issued from several streams) gets executed
        if force_collectives is None:
            force_collectives = os.environ.get("TINYKNN_FORCE_COLLECTIVES", "0") not in ("", "0")
        self.force = bool(force_collectives)
        if self.force and not on:
            raise RuntimeError("force_collectives needs an initialised process group (world size 1 is fine)")
        resident = getattr(ivf, "pq_transformed_points", 0) is None     # IVF.build_resident
        if list_sizes is None and resident:
            list_sizes = ivf.list_sizes
        if list_sizes is None:
            list_sizes = [0 if isinstance(t, np.ndarray) else t.size
                          for t in ivf.pq_transformed_points[:ivf.active_centers.shape[0]]]
        self.list_sizes = np.array(list_sizes, dtype=np.int64)
        self.owner = shard_lists(self.list_sizes, self.world) if owner is None else owner
        self.depth = depth
        self._check_same_index(ivf)
        assert coarse in ("home", "replicated")
        self.coarse = coarse
        assert exchange in ("dense", "filtered", "auto")
        self.exchange = exchange
        sz = self.list_sizes.astype(np.float64)
        self._mean_list = float((sz * sz).sum() / max(sz.sum(), 1.0))    # size-weighted mean rows
        self.calibrate = calibrate
        self.bytes_sent = 0         # filtered exchange: bytes this rank put on the wire ...
        self._bytes_dense = 0       # ... against 16 B x the blocks of whole segments (bytes_dense)
        # submit() answers `coalesce` consecutive batches as ONE sharded batch: the latency-bound
        # stages (two heap replays of the home queries, three collectives) cost the same for
        # 1250 as for 3750 home queries, so what bounds a rank is batches per second, not queries
        self.coalesce = max(1, int(coalesce))
        self._queue = []
        self.engine = engine if engine is not None else _HipShardEngine(
            ivf, self.owner, self.rank, self.world, depth, resident=resident)
        self.device = self.engine.device
        assert counts in ("device", "host")
        if counts == "device" and not (hasattr(self.engine, "filter_regions") and
                                       hasattr(self.engine, "finish_regions")):
            counts = "host"
        self.counts = counts
        self.plain = bool(plain) and all(hasattr(self.engine, a) for a in ("plain_ok", "scan_first", "scan_rest"))
        # dense exchange: the scan in ONE phase (heads exactly, the rest on the matrix cores, the home
        # replay checks the lemma per query and flags the batch: bit 4) where the engine has it;
        # plain="two-phase" keeps the form with the bound after the first list for the dense exchange too
        self._one_phase = self.plain and plain != "two-phase" and hasattr(self.engine, "scan_plain")
        # ... and where that check fails (bit 4: one query in 20 000 of the 100M x 128 index), or on request
        # (plain="head"): the same scan behind ONE byte per query — the bound after the HEAD of the first
        # list, replayed by its owner and min-reduced — which keeps such queries on the exact kernel
        self._head_phase = self._one_phase and hasattr(self.engine, "scan_head")
        self._plain_failed = set()  # (k, n_probes, pass_1) whose one-phase batch raised bit 4
        if plain == "head":
            self._plain_failed = _Everything()
        # ... and arguments whose HEAD form raised bit 4 as well (by construction it cannot; if it ever does, the
        # batch must not be repeated in the same form for ever, inside a collective loop): two-phase scan from then on
        self._head_failed = set()
        self.record_region = {}     # (nq, n_probes) -> records per home-rank region (counts="device")
        self._rec_seen = {}         # ... largest per-home count of the batches looked at so far
        self._acc = None            # device: [largest per-home record count, records, dense blocks] since reset
        self._acc_keys = set()
        self._plain_ok = {}
        self.capacity = {}          # (nq, n_probes) -> uint4 per region, grows on overflow
        self._bufs = {}
        self._pbufs = {}
        self._fbufs = {}
        self._calls = 0
        self._multi = self.device == "cuda" and depth > 1      # batches in flight
        self._streams = None        # one stream per batch in flight, created on first use (_slot_streams)
        # dense exchange, batches in flight: streams by ROLE, as the unsharded pipeline has them
        #  — every scan on ONE stream, in order (two scans at once only stretch each
        # other), the coarse stages + probe all-gathers on a front stream, exchange + replay +
        # rescoring + id gather alternating between two more; handed over by events.  One stream per
        # BATCH (the older form, still what the filtered exchange uses) lets scans of different
        # batches run against each other.  Measured at W = 1 (profiles/r03/shard_streams_ab.txt): no
        # gain — 11.97 M queries/s coalesced / 10.3 M at one step per exchange by role, 11.84 M / 11.2 M
        # by batch: what a sharded batch costs beyond the unsharded one is its own kernels (positions,
        # unpack, the exact scan of every list), not the stream structure.  Off by default;
        # TINYKNN_SHARD_ROLES=1 selects it.
        # One communicator per batch in flight.  torch runs every collective of a process group on that
        # group's ONE internal RCCL stream, in call order: with a single group the probe all-gather of
        # batch b+1 queues behind the id all-gather of batch b, which waits for b's replay and rescoring
        # — the batches "in flight" ran one after the other.  A group per slot lets their exchanges
        # overlap; every rank issues the same sequence on each group.
        self._slot_groups = None
        self._cg = group
        # One communicator per batch in flight lets the collectives of different batches overlap — verified with ONE
        # rank (the forced-collectives rehearsal).  RCCL has never run this at world > 1 (no multi-GPU node so far), and
        # concurrent communicators can deadlock where their kernels cannot all be resident: until such a run has passed,
        # world > 1 takes the process group's single communicator (torch serialises its collectives on one internal
        # stream: correct for any issue order the ranks share) unless TINYKNN_SHARD_COMMS=1 asks for one per batch.
        if (on and depth > 1 and self.backend == "nccl" and (self.world > 1 or self.force) and
                os.environ.get("TINYKNN_SHARD_COMMS", "1" if self.world == 1 else "0") != "0"):
            ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
            self._slot_groups = [dist.new_group(ranks=ranks, backend="nccl") for _ in range(depth)]
        self._roles = None
        roles = os.environ.get("TINYKNN_SHARD_ROLES", "0")
        if self._multi and roles == "1":
            self._roles = dict(front=torch.cuda.Stream(), scan=torch.cuda.Stream(),
                               replay=[torch.cuda.Stream(), torch.cuda.Stream()], done={})
        elif self._multi and roles in ("2", "3") and hasattr(self.engine, "dev"):
            # the LIBRARY's process-wide streams: its high-priority front stream for the coarse stages, its
            # replay streams for exchange + replay + rescoring (tk_shared_stream) — the streams the
            # unsharded pipeline runs on, mapped onto HIP's four hardware queues with the scan stream
            from . import _lib
            ext = lambda role, i: torch.cuda.ExternalStream(int(_lib.lib().tk_shared_stream(role, i)))
            self._roles = dict(front=ext(0, 0), scan=torch.cuda.Stream(),
                               replay=[ext(1, i) for i in range(2 if roles == "2" else 3)], done={})
        self.stage_events = [] if os.environ.get("TINYKNN_SHARD_STAGE_EVENTS", "0") == "1" else None
        self._pending = []
        self._inflight_inputs = []  # concatenated inputs of coalesced batches in flight (released by join())
        self._deferred = None
        self._need = {}             # (nq, n_probes) -> longest streams seen by query_prepared
        self._ovf = None            # per slot: OR of the flag words of its submit()ted batches (device)
        self._ovf_keys = set()      # capacities to grow if the first counter is non-zero at join()
        self._ovf_args = set()      # (k, n_probes, pass_1) of those batches
        self.last_flushed = None    # gathered tensor of the batch join() flushed (coalesce > 1)

    @property
    def bytes_dense(self):
        """16 B x the blocks this rank scored in filtered batches (what the dense form carries at
        least); with counts="device" the tally lives on the device: reading synchronises."""
        n = self._bytes_dense
        if self._acc is not None:
            n += 16 * int(self._acc[2].item())
        return n

    @bytes_dense.setter
    def bytes_dense(self, v):
        self._bytes_dense = int(v)
        if self._acc is not None:
            self._acc[1:].zero_()

    @property
    def records_sent(self):
        """records (20 B each) the filtered batches of this rank really held, counts="device"."""
        return 0 if self._acc is None else int(self._acc[1].item())

    def _check_same_index(self, ivf):
        """Every rank must hold the SAME index (same centres, same lists): positions in the
        exchange are computed, not transmitted.  An index fitted per rank from an unseeded RNG
        differs silently — compare a checksum once, at construction."""
        if (self.world == 1 and not self.force) or self._sim is not None:
            return
        import zlib
        crc = zlib.crc32(np.ascontiguousarray(self.list_sizes).tobytes())
        ac = getattr(ivf, "active_centers", None)
        if ac is not None:
            crc = zlib.crc32(np.ascontiguousarray(ac).tobytes(), crc)
        t = self.torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        lo = t.tensor([crc], dtype=t.int64, device=dev)
        hi = lo.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        if int(lo.item()) != int(hi.item()):
            raise RuntimeError("ListShardedIndex: the ranks hold different indexes (list sizes or centres "
                               "differ) — build every rank's index from the same data and the same seed")

    # -- collectives (RCCL on device tensors; any other backend is staged through the host)
    def _all_to_all(self, recv, send, what="segments"):
        if self._sim is not None:
            self._sim.all_to_all(what, self._sim_ctx, recv, send)
        elif self.world == 1 and not self.force:
            recv.copy_(send)
        elif self.backend == "nccl" or self.device == "cpu":
            self.dist.all_to_all_single(recv, send, group=self._cg)
        else:
            r, s_ = self.torch.empty(recv.shape, dtype=recv.dtype), send.cpu()
            self.dist.all_to_all_single(r, s_, group=self._cg)
            recv.copy_(r)

    def _all_gather(self, out, inp, what="ids"):
        if self._sim is not None:
            self._sim.all_gather(what, self._sim_ctx, out, inp)
        elif self.world == 1 and not self.force:
            out.copy_(inp)
        elif self.backend == "nccl" or self.device == "cpu":
            self.dist.all_gather_into_tensor(out, inp, group=self._cg)
        else:
            o = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu(), group=self._cg)
            out.copy_(o)

    def _all_reduce_min(self, t_):
        if self._sim is not None:
            return self._sim.all_reduce_min(self._sim_ctx, t_)
        if self.world == 1 and not self.force:
            return
        if self.backend == "nccl" or self.device == "cpu":
            self.dist.all_reduce(t_, op=self.dist.ReduceOp.MIN, group=self._cg)
        else:
            h = t_.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN, group=self._cg)
            t_.copy_(h)

    def _all_to_all_rows(self, recv, send, rsplit, ssplit):
        """Variable splits along dim 0."""
        if self.world == 1 and not self.force:
            recv.copy_(send)
        elif self.backend == "nccl" or self.device == "cpu":
            self.dist.all_to_all_single(recv, send, rsplit, ssplit, group=self._cg)
        else:
            r = self.torch.empty(recv.shape, dtype=recv.dtype)
            self.dist.all_to_all_single(r, send.cpu(), rsplit, ssplit, group=self._cg)
            recv.copy_(r)

    def _filtered_buffers(self, slot, nq, capacity, region=0):
        key = (nq, capacity, region)
        if self._fbufs.get(slot, (None,))[0] != key:
            t, W = self.torch, self.world
            mk = lambda n, dt: t.empty(n, dtype=dt, device=self.device)
            self._fbufs[slot] = (key, dict(
                bound=mk(nq, t.uint8), counts=mk(3 * W, t.int32), rcounts=mk(W, t.int32),
                rec=mk((W * (region or capacity), 5), t.int32),
                rrec=mk((W * region if region else 1024, 5), t.int32)))
        return self._fbufs[slot][1]

    def _region(self, nq, n_probes, capacity):
        """records per home-rank region: the dense capacity (never overflows) until batches have
        been looked at, then what they needed (_note_region)."""
        r = self.record_region.get((nq, n_probes))
        return int(capacity if r is None else min(r, capacity))

    def _filtered_regions(self, slot, qn, k, n_probes, pass_1, capacity, b, out_home, bound=None):
        """The whole filtered exchange, enqueued: bound -> min all-reduce -> filter into fixed
        regions -> all-to-all of the counts and (equal splits) of the regions -> finish."""
        W, t = self.world, self.torch
        nq = qn.shape[0]
        region = self._region(nq, n_probes, capacity)
        f = self._filtered_buffers(slot, nq, capacity, region)
        if bound is None:       # (the two-phase scan has reduced it already)
            self.engine.bound(slot, qn, k, n_probes, pass_1, capacity, b["send"], f["bound"])
            self._all_reduce_min(f["bound"])
            bound = f["bound"]
        if self._acc is None:       # [largest per-home count, records, blocks scored]: atomics in the filter
            self._acc = t.zeros(3, dtype=t.int64, device=self.device)
        self.engine.filter_regions(slot, qn, k, n_probes, pass_1, capacity, b["send"], bound,
                                   f["counts"], f["rec"], region, b["flag"], self._acc)
        if W == 1 and not self.force:       # one rank: nothing travels, the regions are read where they lie
            rrec, rcounts = f["rec"], f["counts"]
        else:
            self._all_to_all(f["rcounts"], f["counts"][:W], "counts")
            self._all_to_all(f["rrec"], f["rec"], "records")
            rrec, rcounts = f["rrec"], f["rcounts"]
        self.engine.finish_regions(slot, qn, k, n_probes, pass_1, rrec, rcounts, region,
                                   out_home, b["flag"])
        self._acc_keys.add((nq, n_probes))
        self.bytes_sent += 20 * W * region + nq + 4 * W
        return f

    def _note_region(self, keys, need):
        """`need`: the largest per-home record count of the batches since the last look, already
        max-reduced over the ranks (every rank must arrive at the same region)."""
        for key in keys:
            seen = self._rec_seen.setdefault(key, [])
            seen.append(int(need))
            if self.calibrate and len(seen) >= 3:
                self.record_region[key] = int(1.5 * max(seen)) + 64

    def _max_over_ranks(self, v):
        if self._sim is not None:
            return int(v)
        if self.world > 1 or self.force:
            t_ = self.torch.tensor([int(v)], dtype=self.torch.int64,
                                   device=self.device if self.backend == "nccl" else "cpu")
            self.dist.all_reduce(t_, op=self.dist.ReduceOp.MAX, group=self.group)
            v = int(t_.item())
        return int(v)

    def _take_rec_need(self):
        """Largest per-home record count of the filtered batches (counts="device") since the last
        look, max-reduced over the ranks, and the (nq, n_probes) they had — or (None, ()).
        Synchronises; every rank calls it at the same point."""
        if self._acc is None or not self._acc_keys:
            return None, ()
        if self.device == "cuda":
            # the filters of the batches in flight add to the tally from THEIR streams: a read (and the
            # reset behind it) ordered only on the current stream lost counts on a fresh box — regions
            # were then trimmed below what the next batch needed (the round-4 driver run's overflow)
            self.torch.cuda.synchronize()
        v = int(self._acc[0].item())
        self._acc[0] = 0
        keys, self._acc_keys = self._acc_keys, set()
        return self._max_over_ranks(v), keys

    def _filtered_front(self, slot, qn, k, n_probes, pass_1, capacity, b, bound=None):
        """bound -> min all-reduce -> filter -> all-to-all of the counts (all enqueued)."""
        W = self.world
        f = self._filtered_buffers(slot, qn.shape[0], capacity)
        if bound is None:       # (the two-phase scan has reduced it already)
            self.engine.bound(slot, qn, k, n_probes, pass_1, capacity, b["send"], f["bound"])
            self._all_reduce_min(f["bound"])
            bound = f["bound"]
        assert self._sim is None, "simulated peers: counts=\"device\" only"
        self.engine.filter(slot, qn, k, n_probes, pass_1, capacity, b["send"], bound,
                           f["counts"], f["rec"])
        self._all_to_all(f["rcounts"], f["counts"][:W], "counts")
        return f

    def _filtered_back(self, qn, f):
        """the one host synchronisation (split sizes), then the records -> f["rrec"][:n]."""
        W = self.world
        cnt = f["counts"].cpu().tolist()
        ssplit = cnt[:W]
        self._bytes_dense += 16 * sum(cnt[2 * W:])
        rsplit = f["rcounts"].cpu().tolist()
        n_s, n_r = sum(ssplit), sum(rsplit)
        self.bytes_sent += 20 * n_s + qn.shape[0] + 4 * W
        if W == 1 and not self.force:       # one rank: the records are read where they lie
            return f["rec"], n_s
        if f["rrec"].shape[0] < n_r:
            f["rrec"] = self.torch.empty((int(1.25 * n_r) + 1024, 5), dtype=self.torch.int32,
                                         device=self.device)
        self._all_to_all_rows(f["rrec"][:n_r], f["rec"][:n_s], rsplit, ssplit)
        return f["rrec"], n_r

    def _buffers(self, slot, nq, k, capacity):
        key = (slot, nq, k, capacity)
        if self._bufs.get(slot, (None,))[0] != key:
            t, W = self.torch, self.world
            qh = -(-nq // W)
            mk = lambda n, dt: t.empty(n, dtype=dt, device=self.device)
            home = mk(qh * k + 1, t.int64)
            # the batch's flag word IS the last element of the home rows (its low 32 bits, which is what the
            # engines write): it travels with the ids without a copy
            all_ = mk(W * (qh * k + 1), t.int64)
            if self._sim is not None:
                # simulated peers: their rows of the gathered ids are theirs (-1 here) and their flag words 0 —
                # constant, written once; a batch's "all-gather" then only copies this rank's row in
                all_.fill_(-1)
                all_.view(W, qh * k + 1)[:, -1] = 0
            self._bufs[slot] = (key, dict(
                send=mk(W * capacity * 16, t.uint8), recv=mk(W * capacity * 16, t.uint8),
                flag=home[qh * k:].view(t.int32), home=home, all=all_))
        return self._bufs[slot][1]

    def _use_plain(self, k, n_probes, pass_1):
        """Two-phase scan for these arguments?  (replicated state only: the same on every rank)"""
        if not self.plain:
            return False
        key = (k, n_probes, pass_1)
        if key not in self._plain_ok:
            self._plain_ok[key] = bool(self.engine.plain_ok(k, n_probes, pass_1))
        return self._plain_ok[key]

    def _one_phase_now(self, k, n_probes, pass_1):
        """One-phase plain scan for this batch?  Replicated state only (the flag words that switch it off
        are all-gathered), so every rank answers alike."""
        return (self._one_phase and (k, n_probes, pass_1) not in self._plain_failed and
                self._exchange_kind(k, n_probes, pass_1) == "dense" and self._use_plain(k, n_probes, pass_1))

    def _note_plain_failure(self, args):
        """Bit 4 of a batch's flag word: demote these arguments one level — one-phase -> head form -> two-phase.
        A batch that raises it in a form that does not arm the check (two-phase, exact) is a bug: say so instead of
        repeating it for ever (every rank sees the same gathered flag and raises alike)."""
        for a in args:
            if a in self._plain_failed or not self._head_phase:
                if a in self._head_failed:
                    raise RuntimeError("ListShardedIndex: the plain scan's check failed for %r in every scan form; "
                                       "construct the index with plain=False" % (a,))
                self._head_failed.add(a)
            if not isinstance(self._plain_failed, _Everything):
                self._plain_failed.add(a)

    def _head_phase_now(self, k, n_probes, pass_1):
        """The one-phase scan behind the head bounds: the dense exchange's form once the optimistic one
        has failed for these arguments (or plain="head")."""
        return (self._head_phase and (k, n_probes, pass_1) in self._plain_failed and
                (k, n_probes, pass_1) not in self._head_failed and
                self._exchange_kind(k, n_probes, pass_1) == "dense" and self._use_plain(k, n_probes, pass_1))

    def _scan_form(self, k, n_probes, pass_1):
        return ("one" if self._one_phase_now(k, n_probes, pass_1) else
                "head" if self._head_phase_now(k, n_probes, pass_1) else
                "two" if self._use_plain(k, n_probes, pass_1) else "exact")

    def _finish(self, slot, qn, k, n_probes, pass_1, capacity, recv, b, qh):
        if b.get("one_phase"):
            self.engine.finish(slot, qn, k, n_probes, pass_1, capacity, recv, b["home"][:qh * k], flag=b["flag"])
        else:
            self.engine.finish(slot, qn, k, n_probes, pass_1, capacity, recv, b["home"][:qh * k])

    def _scan(self, slot, qn, qp, k, n_probes, pass_1, capacity, b, probes_all=None):
        """The scan of the owned segments into b["send"]; returns the min-reduced bound after the
        first probed lists (two-phase form) or None."""
        b["one_phase"] = False
        if not self._use_plain(k, n_probes, pass_1):
            self.engine.scan(slot, qn, qp, k, n_probes, pass_1, capacity, b["send"], b["flag"],
                             probes_all=probes_all)
            return None
        if self._one_phase_now(k, n_probes, pass_1):
            self.engine.scan_plain(slot, qn, qp, k, n_probes, pass_1, capacity, b["send"], b["flag"],
                                   probes_all=probes_all)
            b["one_phase"] = True
            return None
        nq = qn.shape[0]
        if self._head_phase_now(k, n_probes, pass_1):
            if b.get("bound") is None or b["bound"].shape[0] != nq:
                b["bound"] = self.torch.empty(nq, dtype=self.torch.uint8, device=self.device)
            self.engine.scan_head(slot, qn, qp, k, n_probes, pass_1, capacity, b["send"], b["flag"], b["bound"],
                                  probes_all=probes_all)
            self._all_reduce_min(b["bound"])
            self.engine.scan_plain(slot, qn, qp, k, n_probes, pass_1, capacity, b["send"], b["flag"],
                                   probes_all=probes_all, bound=b["bound"])
            b["one_phase"] = True       # (the finish takes the flag word; by construction it stays clean)
            return None
        if b.get("bound") is None or b["bound"].shape[0] != nq:
            b["bound"] = self.torch.empty(nq, dtype=self.torch.uint8, device=self.device)
        self.engine.scan_first(slot, qn, qp, k, n_probes, pass_1, capacity, b["send"], b["flag"],
                               b["bound"], probes_all=probes_all)
        self._all_reduce_min(b["bound"])
        self.engine.scan_rest(slot, qn, k, n_probes, pass_1, capacity, b["send"], b["bound"])
        return b["bound"]

    def _coarse_home(self, slot, qn, qp, k, n_probes, pass_1):
        """The home-sharded coarse stage of a batch + the all-gather behind it; returns the gathered probe lists.
        (Every rank builds every query's distance table.  Building a table only on its query's home rank and
        all-gathering it — 832 bytes per query at M = 52 — was built and measured in round 5: 0.115–0.117 against
        0.112 ms per step as one rank's share of W = 8, 58 MB more per batch on the links; removed in round 6.)"""
        nq = qn.shape[0]
        p_home, p_all = self._probe_buffers(slot, nq, n_probes)
        self.engine.coarse(slot, qn, qp, k, n_probes, pass_1, p_home)
        self._all_gather(p_all, p_home, "probes")
        return p_all

    def _probe_buffers(self, slot, nq, n_probes):
        kc = min(n_probes, len(self.list_sizes))
        key = (nq, kc)
        if self._pbufs.get(slot, (None,))[0] != key:
            t, W = self.torch, self.world
            qh = -(-nq // W)
            self._pbufs[slot] = (key, t.empty(qh * kc, dtype=t.int64, device=self.device),
                                 t.empty(W * qh * kc, dtype=t.int64, device=self.device))
        return self._pbufs[slot][1:]

    def _enqueue(self, qn, qp, k, n_probes, pass_1, capacity):
        """One batch on the current stream; returns the gathered (world, qh*k+1) tensor
        (last column: the rank's overflow flag)."""
        return self._enqueue_back(self._enqueue_front(qn, qp, k, n_probes, pass_1, capacity))

    def _enqueue_front(self, qn, qp, k, n_probes, pass_1, capacity):
        """Everything that needs no host decision.  Dense exchange: the whole batch.  Filtered:
        up to the exchange of the record counts; _enqueue_back reads them and does the rest."""
        nq = qn.shape[0]
        slot = self._calls % self.depth
        self._calls += 1
        self._cg = self._slot_groups[slot] if self._slot_groups else self.group
        b = self._buffers(slot, nq, k, capacity)
        qh = -(-nq // self.world)
        b["flag"].zero_()
        self._set_sim_ctx(qn, qp, k, n_probes, pass_1, capacity)
        if self.coarse == "home":
            p_all = self._coarse_home(slot, qn, qp, k, n_probes, pass_1)
            bound = self._scan(slot, qn, qp, k, n_probes, pass_1, capacity, b, probes_all=p_all)
        else:
            bound = self._scan(slot, qn, qp, k, n_probes, pass_1, capacity, b)
        st = dict(slot=slot, qn=qn, k=k, n_probes=n_probes, pass_1=pass_1, capacity=capacity, b=b,
                  out=b["all"].view(self.world, qh * k + 1), f=None)
        if self._exchange_kind(k, n_probes, pass_1) == "filtered" and self.counts == "device":
            self._filtered_regions(slot, qn, k, n_probes, pass_1, capacity, b, b["home"][:qh * k], bound)
            self._gather_ids(b, qh, k)
        elif self._exchange_kind(k, n_probes, pass_1) == "filtered":
            st["f"] = self._filtered_front(slot, qn, k, n_probes, pass_1, capacity, b, bound)
        else:
            recv = b["send"]            # one rank: nothing travels, the segments are read where they lie
            if self.world > 1 or self.force:
                self._all_to_all(b["recv"], b["send"])
                recv = b["recv"]
            self._finish(slot, qn, k, n_probes, pass_1, capacity, recv, b, qh)
            self._gather_ids(b, qh, k)
        return st

    def _set_sim_ctx(self, qn, qp, k, n_probes, pass_1, capacity):
        if self._sim is None:
            return
        kind = self._exchange_kind(k, n_probes, pass_1)
        self._sim_ctx = self._sim.context(
            qn, qp, k, n_probes, pass_1, capacity, coarse=self.coarse, kind=kind,
            region=self._region(qn.shape[0], n_probes, capacity) if kind == "filtered" else 0,
            form=self._scan_form(k, n_probes, pass_1))

    def _exchange_kind(self, k, n_probes, pass_1):
        """"auto": the filter drops what is not below the bound after the first list — worth its
        host synchronisation only where a list holds many heaps' worth of rows (measured at W = 1:
        0.36 of the bytes and -15 % queries/s at 1 100-row lists with a heap of 111; 0.21 of the
        bytes and +7 % at 10 000-row lists).  The rule depends on replicated values only, so every
        rank takes the same branch."""
        if self.exchange != "auto":
            return self.exchange
        heap = int(pass_1) if pass_1 else (n_probes + 1) * k + 1
        return "filtered" if self._mean_list >= 32 * heap else "dense"

    def _gather_ids(self, b, qh, k):
        self._all_gather(b["all"], b["home"])       # (the last element of the home rows is the flag word)

    def _enqueue_back(self, st):
        if st["f"] is not None:
            self._cg = self._slot_groups[st["slot"]] if self._slot_groups else self.group
            b, qn, k = st["b"], st["qn"], st["k"]
            qh = -(-qn.shape[0] // self.world)
            rrec, n_r = self._filtered_back(qn, st["f"])
            self.engine.finish_filtered(st["slot"], qn, k, st["n_probes"], st["pass_1"], rrec, n_r,
                                        b["home"][:qh * k], b["flag"])
            self._gather_ids(b, qh, k)
            st["f"] = None
        return st["out"]

    def _capacity(self, nq, n_probes):
        key = (nq, n_probes)
        if key not in self.capacity:
            self.capacity[key] = shard_capacity(self.list_sizes, self.owner, self.world, nq, n_probes)
        return self.capacity[key]

    def _usage(self, slot):
        """Longest (source -> home) stream of the slot's last batch over all ranks, in uint4
        (None if the engine does not report it)."""
        if not hasattr(self.engine, "usage"):
            return None
        u = int(self.engine.usage(slot))
        if self._sim is not None:
            return max(u, self._sim.usage(self._sim_ctx))
        if self.world > 1 or self.force:
            t_ = self.torch.tensor([u], dtype=self.torch.int64,
                                   device=self.device if self.backend == "nccl" else "cpu")
            self.dist.all_reduce(t_, op=self.dist.ReduceOp.MAX, group=self.group)
            u = int(t_.item())
        return u

    def query_prepared(self, qn, qp, k, n_probes=1, pass_1=None):
        """qn / qp: tensors on the engine's device (normalised queries, table-build queries).
        Synchronous; repeats the batch with a larger capacity if a region overflowed, and
        trims the capacity to 1.25 x the longest stream seen (the regions of the dense exchange
        travel whole: the a-priori estimate carries ~2x the bytes the segments need)."""
        nq = qn.shape[0]
        qh = -(-nq // self.world)
        self._finish_deferred()
        while True:
            cap = self._capacity(nq, n_probes)
            slot = self._calls % self.depth
            g = self._enqueue(qn, qp, k, n_probes, pass_1, cap)
            g = g.cpu().numpy()
            if (g[:, -1] & 2).any():
                raise RuntimeError("filtered exchange: a record outside the home rank's rows")
            if (g[:, -1] & 4).any():
                # a home query failed the one-phase plain scan's check (every rank sees the gathered flag):
                # these arguments take the two-phase form from now on; the batch again
                self._note_plain_failure({(k, n_probes, pass_1)})
                if not (g[:, -1] & 1).any():
                    self._take_rec_need()
                    continue
            need = self._usage(slot)
            # (the device-side maximum is shared by every batch since the last look: batches still in
            #  flight from submit(), possibly with another (nq, n_probes), are credited with it too —
            #  an over-estimate for them, never a lost count)
            rec_need, rec_keys = self._take_rec_need()
            key = (nq, n_probes)
            if not g[:, -1].any():
                # trim only on the evidence of SEVERAL batches (their longest stream, +35 %): one
                # batch's streams say little about the next batch's, and a pipelined submit() only
                # learns of an overflow at join()
                if need is not None and self.calibrate:
                    seen = self._need.setdefault(key, [])
                    seen.append(need)
                    if len(seen) >= 3 and int(1.2 * max(seen)) + 64 < 0.9 * cap:
                        self.capacity[key] = int(1.2 * max(seen)) + 64
                if rec_need is not None:
                    self._note_region(set(rec_keys) | {key}, rec_need)
                return g[:, :-1].reshape(self.world * qh, k)[:nq]
            if rec_need is not None and rec_need > self._region(nq, n_probes, cap):
                # the record regions were too small (counts="device"); the streams may have fitted
                self.record_region[key] = int(1.25 * rec_need) + 64
                self._rec_seen.pop(key, None)
                if need is not None and need <= cap:
                    continue
            worst = qh * min(n_probes, len(self.list_sizes)) * int((self.list_sizes.max() + 15) // 16)
            assert cap < worst, "overflow at the worst-case capacity"
            grow = 2 * cap if need is None else max(int(1.25 * need) + 64, cap + 1)
            self.capacity[(nq, n_probes)] = min(grow, worst)

    def query_batch(self, qs, k, n_probes=1, pass_1=None):
        """Every rank passes the same (nq, d) batch and receives all (nq, k) ids
        (rows padded with -1 like IVF.query_batch)."""
        t = self.torch
        qs = np.array(qs, dtype=np.float32, order="C", copy=True)
        out = np.empty((len(qs), k), dtype=np.int64)
        for lo in range(0, len(qs), 32768):       # (a sharded batch may hold up to 131072 queries)
            qn, qp = self.ivf._prepare(qs[lo:lo + 32768])
            out[lo:lo + len(qn)] = self.query_prepared(
                t.from_numpy(np.ascontiguousarray(qn)).to(self.device),
                t.from_numpy(np.ascontiguousarray(qp)).to(self.device), k, n_probes, pass_1)
        return out

    # -- batches in flight (bench): each on its own stream, checked at join()
    def submit(self, qn, qp, k, n_probes=1, pass_1=None):
        """Enqueue one batch; returns the gathered (world, qh*k+1) tensor of the sharded batch
        it went into (rows in query order over the coalesced batches, last column = overflow
        flag), or None while the batch waits for `coalesce - 1` more (join() flushes)."""
        if self.coalesce > 1:
            if self._queue and self._qargs != (k, n_probes, pass_1):
                self._flush()           # batches with other arguments never share an exchange
            self._queue.append((qn, qp))
            self._qargs = (k, n_probes, pass_1)
            if len(self._queue) < self.coalesce:
                return None
            return self._flush()
        return self._submit_one(qn, qp, k, n_probes, pass_1)

    def _flush(self):
        if not self._queue:
            return None
        t = self.torch
        k, n_probes, pass_1 = self._qargs
        made = len(self._queue) > 1
        if made and self._roles is not None and self.coarse == "home" and self._exchange_kind(k, n_probes, pass_1) == "dense":
            # by role: the concatenation belongs to the front stream (the table build reads it there);
            # on the caller's stream it would sit in a queue a later stage shares
            front = self._roles["front"]
            front.wait_stream(t.cuda.current_stream())
            with t.cuda.stream(front):
                qn = t.cat([a for a, _ in self._queue])
                qp = t.cat([b for _, b in self._queue])
        else:
            qn = t.cat([a for a, _ in self._queue]) if made else self._queue[0][0]
            qp = t.cat([b for _, b in self._queue]) if made else self._queue[0][1]
        self._queue = []
        self.last_flushed = self._submit_one(qn, qp, k, n_probes, pass_1, own_inputs=made)
        return self.last_flushed

    def _submit_one(self, qn, qp, k, n_probes=1, pass_1=None, own_inputs=False):
        t = self.torch
        if own_inputs and self.device == "cuda":
            # the concatenated inputs of a coalesced batch are OURS, created on the caller's stream and
            # read by a batch that runs on another one: without these references (dropped at join()) the
            # caching allocator hands their blocks to the NEXT batch's concatenation while this batch is
            # still in flight — the queries' block to the table-build queries, say — and the batch reads
            # garbage: an intermittent "overflow" / failed plain check in rounds 3-4, never a wrong row
            # in a synchronous call
            self._inflight_inputs.append((qn, qp))
        cap = self._capacity(qn.shape[0], n_probes)
        self._ovf_keys.add((qn.shape[0], n_probes))
        self._ovf_args.add((k, n_probes, pass_1))
        if not self._multi:
            out = self._enqueue(qn, qp, k, n_probes, pass_1, cap)
            self._note_flags(out, (self._calls - 1) % self.depth)
            return out
        if self._roles is not None and self._exchange_kind(k, n_probes, pass_1) == "dense":
            return self._submit_roles(qn, qp, k, n_probes, pass_1, cap)
        st = self._slot_streams()[self._calls % self.depth]
        st.wait_stream(t.cuda.current_stream())
        with t.cuda.stream(st):
            state = self._enqueue_front(qn, qp, k, n_probes, pass_1, cap)
            if state["f"] is None:
                self._note_flags(state["out"], state["slot"])
        # filtered exchange: the host must read this batch's record counts before it can enqueue
        # the second half — it does so only after the NEXT batch's first half is in the queue
        # (same order on every rank), so that the device is never idle while the host waits
        self._finish_deferred()
        self._deferred = (st, state)
        if state["f"] is None:
            self._deferred = None
        return state["out"]

    def _slot_streams(self):
        if self._streams is None:
            self._streams = [self.torch.cuda.Stream() for _ in range(self.depth)]
        return self._streams

    def _submit_roles(self, qn, qp, k, n_probes, pass_1, cap):
        """One dense-exchange batch over the role streams (see __init__)."""
        t = self.torch
        R = self._roles
        nq = qn.shape[0]
        slot = self._calls % self.depth
        n = self._calls
        self._calls += 1
        self._cg = self._slot_groups[slot] if self._slot_groups else self.group
        b = self._buffers(slot, nq, k, cap)
        qh = -(-nq // self.world)
        self._set_sim_ctx(qn, qp, k, n_probes, pass_1, cap)
        cur = t.cuda.current_stream()
        first = R["front"] if self.coarse == "home" else R["scan"]
        first.wait_stream(cur)                       # the caller's queries
        if slot in R["done"]:                        # the slot's buffers: free once its last batch is out
            first.wait_event(R["done"][slot])
        p_all = None
        ev = None
        if self.stage_events is not None:      # diagnosis: when each stage of a batch starts / ends on its stream
            ev = [t.cuda.Event(enable_timing=True) for _ in range(5)]
            self.stage_events.append(ev)
            ev[0].record(first)
        if self.coarse == "home":
            with t.cuda.stream(R["front"]):
                b["flag"].zero_()
                p_all = self._coarse_home(slot, qn, qp, k, n_probes, pass_1)
                if ev:
                    ev[1].record()
            R["scan"].wait_stream(R["front"])
        with t.cuda.stream(R["scan"]):
            if self.coarse != "home":
                b["flag"].zero_()
            if ev:
                ev[2].record()
            self._scan(slot, qn, qp, k, n_probes, pass_1, cap, b, probes_all=p_all)
            scanned = t.cuda.Event(enable_timing=ev is not None)
            scanned.record(R["scan"])
            if ev:
                ev[3] = scanned
        rs = R["replay"][n % len(R["replay"])]
        rs.wait_event(scanned)
        with t.cuda.stream(rs):
            recv = b["send"]
            if self.world > 1 or self.force:
                self._all_to_all(b["recv"], b["send"])
                recv = b["recv"]
            self._finish(slot, qn, k, n_probes, pass_1, cap, recv, b, qh)
            self._gather_ids(b, qh, k)
            out = b["all"].view(self.world, qh * k + 1)
            self._note_flags(out, slot)
            done = t.cuda.Event(enable_timing=ev is not None)
            done.record(rs)
            if ev:
                ev[4] = done
        R["done"][slot] = done
        return out

    def batch_streams(self):
        """The streams submit()ted batches end on ([] = the current stream): a stream that waits for all
        of them is behind every batch submitted so far, without blocking any of them."""
        if not self._multi:
            return []
        return (list(self._roles["replay"]) if self._roles is not None else []) + list(self._streams or [])

    def flush_host_decisions(self):
        """Filtered exchange with counts="host": the second half of the last batch waits for the host to
        read its record counts; enqueue it now (synchronises).  Nothing to do in every other mode."""
        self._finish_deferred()

    def _finish_deferred(self):
        if self._deferred is not None:
            st, state = self._deferred
            self._deferred = None
            with self.torch.cuda.stream(st):
                self._note_flags(self._enqueue_back(state), state["slot"])

    def _note_flags(self, out, slot):
        """The flag column of a submit()ted batch (every rank's overflow / bad-record / plain-check flags)
        is OR-ed into the slot's accumulator, on the batch's stream (one small kernel; batches of a slot
        share a stream, so no atomics): join() reads the accumulators once."""
        acc = self._ovf.get(slot) if self._ovf else None
        if acc is None:
            if self._ovf is None:
                self._ovf = {}
            acc = self._ovf[slot] = self.torch.zeros(self.world, dtype=self.torch.int64, device=self.device)
        acc.bitwise_or_(out[:, -1])

    def join(self):
        """Flushes a partly filled coalesced batch (its gathered tensor: the return value and
        `last_flushed`), re-joins the batches in flight, and RAISES if any batch since the last
        join overflowed its exchange regions (its rows are then invalid): the capacities have been
        grown, submit the batches again."""
        flushed = self._flush()
        self._finish_deferred()
        if self._multi:
            cur = self.torch.cuda.current_stream()
            for st in self._streams or []:
                cur.wait_stream(st)
            if self._roles is not None:
                for st in [self._roles["front"], self._roles["scan"]] + self._roles["replay"]:
                    cur.wait_stream(st)
        if self._inflight_inputs:
            if self.device == "cuda":       # (the batches that read them have to be DONE, not only ordered)
                self.torch.cuda.current_stream().synchronize()
            self._inflight_inputs = []
        if self._ovf is not None and self._ovf_keys:
            # (synchronises with the batches in flight)
            seen = 0
            for acc in self._ovf.values():
                for x in acc.cpu().tolist():
                    seen |= int(x)
            bad, plain_bad = seen & 3, seen & 4
            keys, self._ovf_keys = self._ovf_keys, set()
            args, self._ovf_args = self._ovf_args, set()
            rec_need, rec_keys = self._take_rec_need()
            if seen:
                for acc in self._ovf.values():
                    acc.zero_()
            if plain_bad:
                self._note_plain_failure(args)
                if not bad:
                    raise RuntimeError("ListShardedIndex: a home query of a batch in flight failed the one-phase plain "
                                       "scan's check (its rows are invalid); these arguments take the two-phase scan "
                                       "from now on — submit the batches again")
            if bad:
                for key in keys:
                    nq, n_probes = key
                    qh = -(-nq // self.world)
                    worst = qh * min(n_probes, len(self.list_sizes)) * int((self.list_sizes.max() + 15) // 16)
                    if rec_need is not None and key in rec_keys and \
                            rec_need > self._region(nq, n_probes, self._capacity(nq, n_probes)):
                        self.record_region[key] = int(1.25 * rec_need) + 64
                        self._rec_seen.pop(key, None)
                    self.capacity[key] = min(2 * self._capacity(nq, n_probes), worst)
                    self._need.pop(key, None)
                raise RuntimeError("ListShardedIndex: a batch in flight overflowed its exchange regions (or "
                                   "carried a bad record); capacities doubled — submit the batches again")
            if rec_need is not None:
                self._note_region(rec_keys, rec_need)
        return flushed
