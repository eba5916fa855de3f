"""CPU stand-in for the per-rank device engine of ListShardedIndex (test infrastructure:
built on the oracle).  `scan` scores the segments this rank owns with the oracle's
estimate_pq and packs them exactly where shard_positions_kernel would; `finish` checks
every received segment of the home queries against a local recomputation and answers them
with the oracle's full query — so a misplaced, missing or foreign byte in the exchange
fails the test."""
import numpy as np

from tinyknn_amd.multi_gpu import shard_positions


class OracleShardEngine:
    device = "cpu"

    def __init__(self, O, ox, owner, rank, world):
        self.O, self.ox, self.owner, self.rank, self.world = O, ox, owner, rank, world
        self.chunks = np.diff(ox.list_chunk_off)

    def _front(self, qn, k, n_probes, pass_1):
        probes, tables = [], []
        for q in qn:
            _, dbg = self.ox.query(q, k, n_probes, pass_1, debug=True)
            p = dbg["probes"].copy()
            p[p < 0] += self.ox.n_lists            # ivf.py:141 list indexing from the end
            probes.append(p)
            tables.append(self.O.transform_tables(dbg["table"]))
        return np.array(probes), tables

    def _segment(self, l, table):
        off = self.ox.list_chunk_off
        codes = np.ascontiguousarray(self.ox.codes[off[l]:off[l + 1]])
        out = np.zeros(2 * len(codes), dtype=np.uint64)
        if len(codes):
            self.O.estimate_pq(codes, table, out, True)
        return out.view(np.uint8)

    def coarse(self, slot, qn, qp, k, n_probes, pass_1, probes_home):
        """Probe lists of this rank's home queries only (rows past nq: 0)."""
        qn = qn.numpy()
        qh = -(-len(qn) // self.world)
        out = probes_home.numpy().reshape(qh, -1)
        out[:] = 0
        lo, hi = self.rank * qh, min(len(qn), (self.rank + 1) * qh)
        if hi > lo:
            for i in range(lo, hi):
                _, dbg = self.ox.query(qn[i], k, n_probes, pass_1, debug=True)
                out[i - lo] = dbg["probes"]        # unwrapped, as the coarse stage leaves them
        self.coarse_calls = getattr(self, "coarse_calls", 0) + 1

    def scan(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=None):
        qn = qn.numpy()
        probes, tables = self._front(qn, k, n_probes, pass_1)
        if probes_all is not None:
            # the gathered probe lists must be what every rank would have derived itself
            got = probes_all.numpy().reshape(-1, probes.shape[1])[:len(qn)].copy()
            got[got < 0] += self.ox.n_lists
            np.testing.assert_array_equal(got, probes)
        src, pos = shard_positions(probes, self.chunks, self.owner, self.world, capacity)
        buf = send.numpy().reshape(self.world, capacity * 16)
        buf[:] = 0xAB                               # stale bytes must never be consumed
        qh = -(-len(qn) // self.world)
        for i in range(len(qn)):
            for s in range(probes.shape[1]):
                if src[i, s] != self.rank:
                    continue
                if pos[i, s] < 0:
                    flag[0] = 1
                    continue
                seg = self._segment(probes[i, s], tables[i])
                buf[i // qh, pos[i, s] * 16: pos[i, s] * 16 + len(seg)] = seg

    def finish(self, slot, qn, k, n_probes, pass_1, capacity, recv, out_home):
        qn = qn.numpy()
        probes, tables = self._front(qn, k, n_probes, pass_1)
        src, pos = shard_positions(probes, self.chunks, self.owner, self.world, capacity)
        buf = recv.numpy().reshape(self.world, capacity * 16)
        qh = -(-len(qn) // self.world)
        out = out_home.numpy().reshape(qh, k)
        out[:] = -1
        for i in range(self.rank * qh, min(len(qn), (self.rank + 1) * qh)):
            for s in range(probes.shape[1]):
                if pos[i, s] < 0:
                    continue                        # overflowed: the batch is repeated
                seg = self._segment(probes[i, s], tables[i])
                got = buf[src[i, s], pos[i, s] * 16: pos[i, s] * 16 + len(seg)]
                np.testing.assert_array_equal(got, seg)
            out[i - self.rank * qh] = self.ox.query_batch(qn[i:i + 1], k, n_probes, pass_1)[0]
