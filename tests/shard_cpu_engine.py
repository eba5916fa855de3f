"""CPU stand-in for the per-rank device engine of ListShardedIndex (test infrastructure:
built on the oracle).  `scan` scores the segments this rank owns with the oracle's
estimate_pq and packs them exactly where shard_positions_kernel would; `finish` checks
every received segment of the home queries against a local recomputation and answers them
with the oracle's full query — so a misplaced, missing or foreign byte in the exchange
fails the test.  Filtered exchange: `bound` replays the first probed list with the oracle's
query_pq from a fresh heap, `filter` packs the blocks below the reduced bound, and
`finish_filtered` checks that exactly the blocks the rule names arrived, byte for byte."""
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
        self.last_capacity = capacity
        probes, tables = self._front(qn, k, n_probes, pass_1)
        if probes_all is not None:
            # the gathered probe lists must be what every rank would have derived itself
            got = probes_all.numpy().reshape(-1, probes.shape[1])[:len(qn)].copy()
            got[got < 0] += self.ox.n_lists
            np.testing.assert_array_equal(got, probes)
        src, pos = shard_positions(probes, self.chunks, self.owner, self.world, capacity)
        _, free = shard_positions(probes, self.chunks, self.owner, self.world, 10 ** 12)
        self.last_usage = int((free + self.chunks[probes]).max()) if probes.size else 0
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

    # ---- the scan in two phases (tk_index_shard_scan_first_dev / _rest_dev).  This engine has no
    # second kernel: phase 1 scores everything and computes the bound, phase 2 checks that the
    # bound it is handed is the min-reduced one every rank must hold.
    def plain_ok(self, k, n_probes, pass_1):
        return n_probes >= 2

    def scan_first(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, bound, probes_all=None):
        self.scan(slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=probes_all)
        self.bound(slot, qn, k, n_probes, pass_1, capacity, send, bound)
        self.first_bound = bound.numpy().copy()

    def scan_rest(self, slot, qn, k, n_probes, pass_1, capacity, send, bound):
        b = bound.numpy()
        mine = self.first_bound != 255
        np.testing.assert_array_equal(b[mine], self.first_bound[mine])      # my queries: my value survived the MIN
        qn_ = qn.numpy()
        probes, _ = self._front(qn_, k, n_probes, pass_1)
        owned_elsewhere = (self.owner[probes[:, 0]] != self.rank)
        assert (self.first_bound[owned_elsewhere] == 255).all()
        self.rest_calls = getattr(self, "rest_calls", 0) + 1

    # ---- the scan in ONE phase (tk_index_shard_scan_plain_dev).  This engine has no second kernel: it
    # scores every segment exactly (bytes that trivially satisfy the lemma); `fail_plain` makes the
    # home replay of the next `fail_plain` such batches report a query that failed the check, which
    # is how the tests drive the bit-4 protocol (flag travels with the ids, the batch is repeated
    # in the two-phase form).
    fail_plain = 0

    def scan_plain(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=None, bound=None):
        if bound is not None:       # behind scan_head: the bytes are the min-reduced head bounds every rank must hold
            b = bound.numpy()
            mine = self.head_bound != 255
            np.testing.assert_array_equal(b[mine], self.head_bound[mine])
            self.head_calls = getattr(self, "head_calls", 0) + 1
            return
        self.scan(slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=probes_all)
        self.plain_calls = getattr(self, "plain_calls", 0) + 1

    def scan_head(self, slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, bound, probes_all=None):
        """tk_index_shard_scan_head_dev: (this engine scores everything at once) + the bound after the HEAD
        of the first probed list — its first ceil(2 R / 16) chunks — for the queries whose first list is mine."""
        self.scan(slot, qn, qp, k, n_probes, pass_1, capacity, send, flag, probes_all=probes_all)
        qn_ = qn.numpy()
        probes, tables = self._front(qn_, k, n_probes, pass_1)
        R = self._heap_size(k, n_probes, pass_1)
        E = (2 * R + 15) // 16
        out = bound.numpy()
        out[:] = 255
        off = self.ox.list_chunk_off
        for i in range(len(qn_)):
            l = probes[i, 0]
            if self.owner[l] != self.rank:
                continue
            idx = np.zeros(R, dtype=np.int64)
            val = np.zeros(R, dtype=np.int32)
            self.O.init_heap(idx, val, True)
            codes = np.ascontiguousarray(self.ox.codes[off[l]:off[l + 1]][:E])
            if len(codes):
                self.O.query_pq(codes, min(int(self.ox.list_n[l]), 16 * len(codes)), tables[i], idx, val, True)
            out[i] = (int(val[0]) & 0xff) ^ 0x80
        self.head_bound = out.copy()

    def finish(self, slot, qn, k, n_probes, pass_1, capacity, recv, out_home, flag=None):
        if flag is not None and self.fail_plain > 0:
            self.fail_plain -= 1
            flag.numpy()[0] |= 4
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

    def usage(self, slot):
        """longest stream of the last scan, fitted or not (tk_index_shard_usage)"""
        return self.last_usage

    # ---- filtered exchange: same three calls as the HIP engine
    def _heap_size(self, k, n_probes, pass_1):
        return int(pass_1) if pass_1 else (n_probes + 1) * k + 1        # ivf.py:135-136

    def bound(self, slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound):
        qn = qn.numpy()
        probes, tables = self._front(qn, k, n_probes, pass_1)
        out = bound.numpy()
        out[:] = 255
        off = self.ox.list_chunk_off
        for i in range(len(qn)):
            l = probes[i, 0]
            if self.owner[l] != self.rank:
                continue
            idx = np.zeros(self._heap_size(k, n_probes, pass_1), dtype=np.int64)
            val = np.zeros(len(idx), dtype=np.int32)
            self.O.init_heap(idx, val, True)
            codes = np.ascontiguousarray(self.ox.codes[off[l]:off[l + 1]])
            if len(codes):
                self.O.query_pq(codes, int(self.ox.list_n[l]), tables[i], idx, val, True)
            out[i] = (int(val[0]) & 0xff) ^ 0x80

    def _passing(self, seg, slot, b):
        """chunk indices of a segment that travel under bound key b"""
        m = seg.view(np.int8).reshape(-1, 16).min(axis=1).astype(np.int64) + 128
        return np.arange(len(m)) if slot == 0 else np.flatnonzero(m < int(b))

    def filter(self, slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound, counts, records):
        qn = qn.numpy()
        probes, tables = self._front(qn, k, n_probes, pass_1)
        src, pos = shard_positions(probes, self.chunks, self.owner, self.world, capacity)
        buf = scan_buf.numpy().reshape(self.world, capacity * 16)
        b = bound.numpy()
        qh = -(-len(qn) // self.world)
        cap = probes.shape[1] * int(self.chunks.max())                  # Plan.cap (api_index.hip make_plan)
        prefix = np.concatenate([np.zeros((len(qn), 1), np.int64),
                                 np.cumsum(self.chunks[probes], axis=1)], axis=1)
        per_home = [[] for _ in range(self.world)]
        dense = np.zeros(self.world, dtype=np.int64)
        for i in range(len(qn)):
            h = i // qh
            for s in range(probes.shape[1]):
                if src[i, s] != self.rank or pos[i, s] < 0:
                    continue
                n = int(self.chunks[probes[i, s]])
                seg = buf[h, pos[i, s] * 16:(pos[i, s] + n) * 16]
                dense[h] += n
                for c in self._passing(seg, s, b[i]):
                    rec = np.empty(5, dtype=np.int32)
                    rec[0] = (i - h * qh) * cap + prefix[i, s] + c
                    rec[1:] = seg[16 * c:16 * c + 16].view(np.int32)
                    per_home[h].append(rec)
        cnt = counts.numpy()
        cnt[:] = 0
        cnt[:self.world] = [len(x) for x in per_home]
        cnt[2 * self.world:] = dense
        flat = [r for x in per_home for r in x]
        if flat:
            records.numpy()[:len(flat)] = np.array(flat)

    def finish_filtered(self, slot, qn, k, n_probes, pass_1, records, n_records, out_home, flag):
        qn = qn.numpy()
        probes, tables = self._front(qn, k, n_probes, pass_1)
        qh = -(-len(qn) // self.world)
        cap = probes.shape[1] * int(self.chunks.max())
        _, pos = shard_positions(probes, self.chunks, self.owner, self.world, self.last_capacity)
        rec = records.numpy()[:n_records]
        got = {int(r[0]): r[1:].tobytes() for r in rec}
        assert len(got) == n_records, "a block arrived twice"
        out = out_home.numpy().reshape(qh, k)
        out[:] = -1
        want = 0
        # the bound every rank must have used: after the first list, from a fresh heap
        for i in range(self.rank * qh, min(len(qn), (self.rank + 1) * qh)):
            idx = np.zeros(self._heap_size(k, n_probes, pass_1), dtype=np.int64)
            val = np.zeros(len(idx), dtype=np.int32)
            self.O.init_heap(idx, val, True)
            l0 = probes[i, 0]
            off = self.ox.list_chunk_off
            codes = np.ascontiguousarray(self.ox.codes[off[l0]:off[l0 + 1]])
            if len(codes):
                self.O.query_pq(codes, int(self.ox.list_n[l0]), tables[i], idx, val, True)
            b = (int(val[0]) & 0xff) ^ 0x80
            f0 = 0
            for s in range(probes.shape[1]):
                seg = self._segment(probes[i, s], tables[i])
                for c in (self._passing(seg, s, b) if pos[i, s] >= 0 else []):   # overflowed: repeated
                    key = (i - self.rank * qh) * cap + f0 + int(c)
                    assert got.get(key) == seg[16 * c:16 * c + 16].tobytes(), (i, s, c)
                    want += 1
                f0 += int(self.chunks[probes[i, s]])
            out[i - self.rank * qh] = self.ox.query_batch(qn[i:i + 1], k, n_probes, pass_1)[0]
        assert want == n_records, "blocks that should not have travelled"

    # ---- the same without the host synchronisation: fixed regions, counts read "on the device"
    def filter_regions(self, slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound, counts, records,
                       region, flag, acc=None):
        import torch
        W = self.world
        compact = torch.zeros((W * capacity, 5), dtype=torch.int32)
        self.filter(slot, qn, k, n_probes, pass_1, capacity, scan_buf, bound, counts, compact)
        cnt = counts.numpy()[:W]
        if acc is not None:                             # the caller's books (atomics on the device)
            a = acc.numpy()
            a[0] = max(int(a[0]), int(cnt.max()))
            a[1] += int(cnt.sum())
            a[2] += int(counts.numpy()[2 * W:].sum())
        rec = records.numpy().reshape(W, region, 5)
        rec[:] = -9                                     # (stale bytes must never be read as records)
        o = 0
        for h in range(W):
            n = int(cnt[h])
            rec[h, :min(n, region)] = compact.numpy()[o:o + min(n, region)]
            if n > region:
                flag.numpy()[0] |= 1
            o += n

    def finish_regions(self, slot, qn, k, n_probes, pass_1, records, counts_recv, region, out_home, flag):
        import torch
        W = self.world
        cnt = counts_recv.numpy()[:W]
        if (cnt > region).any():                        # a sender overflowed (its flag says so): the
            out_home.numpy()[:] = -1                    # batch is repeated, nothing to check here
            return
        rec = records.numpy().reshape(W, region, 5)
        got = np.concatenate([rec[s_, :int(cnt[s_])] for s_ in range(W)]) if cnt.sum() else np.zeros((0, 5), np.int32)
        self.finish_filtered(slot, qn, k, n_probes, pass_1, torch.from_numpy(np.ascontiguousarray(got)),
                             len(got), out_home, flag)

