"""IVF with the reference's API (tinyknn/ivf.py); queries run on the MI355X.

fit / build are offline host code with the reference's semantics.  After build the
index (PQ codebook, coded coarse centres, inverted lists, ids, rescoring vectors)
is uploaded once into HBM (`DeviceIndex`); `query` and `query_batch` then run the
kernel pipeline of libtinyknn_hip.so: distance tables -> coarse scan + heap +
rescoring -> probed-list scan -> exact heap replay -> exact rescoring.
"""
import ctypes as C
import weakref

import numpy as np

from . import _front, _lib
from .fast_pq import FastPQ, avx, dpad
from .utils import group_data_by_indices, knn_brute, timer


def synth_rows(n, d, seed, centres=None, sigma=1.0, row0=0):
    """Rows [row0, row0 + n) of the seeded device generator (devbuild.hip synth_rows_kernel:
    a pure function of (seed, row)), copied to the host — the vectors IVF.build_resident
    generates in HBM, e.g. to fit centres on a sample or to draw queries."""
    out = np.zeros((n, d), dtype=np.float32)
    c = None if centres is None else np.ascontiguousarray(centres, dtype=np.float32)
    _lib.check(_lib.lib().tk_synth_rows(_lib.ptr(out, _lib._f32p), int(row0), int(n), int(d), int(seed),
                                        None if c is None else c.ctypes.data,
                                        0 if c is None else len(c), float(sigma)))
    return out


class QueryStream:
    """Streaming session on a DeviceIndex (C ABI: tk_stream_*): raw float32 queries on the
    host in, ids on the host out, batch after batch; the exact host preparation
    (ivf.py:125-128 through numpy's own BLAS, see _front.py), the copies and the kernels of
    consecutive batches overlap.  submit() returns a ticket; the ids land in the array given
    to submit() by wait(ticket) / drain()."""

    def __init__(self, dev, max_nq, k, n_probes, pass_1=None, slots=8):
        if not _front.bind():
            raise _lib.TinyKnnHipError("no BLAS bound for the exact host front end: " +
                                       str(_front.info()["why"]))
        self._dev = dev                 # keeps the index alive
        dev._live_streams.add(self)     # ... and the index closes its sessions before it goes
        self.max_nq, self.k = int(max_nq), int(k)
        R = dev._R
        self._s = _lib.lib().tk_stream_create(
            dev.handle, self.max_nq, self.k, int(n_probes), int(pass_1 or 0), int(dev.angular),
            None if R is None else R.ctypes.data, 0 if R is None else R.shape[1], int(slots))
        if not self._s:
            raise _lib.TinyKnnHipError(_lib.lib().tk_last_error().decode())
        self._keep = {}

    def submit(self, qs, out):
        """qs (nq, d) float32 C-contiguous raw queries (not modified); out (nq, k) int64."""
        assert qs.dtype == np.float32 and qs.flags.c_contiguous and qs.shape[1] == self._dev.d
        assert out.dtype == np.int64 and out.flags.c_contiguous and out.shape == (len(qs), self.k)
        t = _lib.check(_lib.lib().tk_stream_submit(self._s, qs.ctypes.data, len(qs), out.ctypes.data))
        self._keep[t % 64] = (qs, out)
        return t

    def submit_prepared(self, qn, q_pq, out):
        """Already prepared rows (what DeviceIndex.query_batch takes)."""
        assert qn.dtype == np.float32 and qn.flags.c_contiguous and qn.shape[1] == self._dev.d
        assert out.dtype == np.int64 and out.flags.c_contiguous and out.shape == (len(qn), self.k)
        t = _lib.check(_lib.lib().tk_stream_submit_prepared(
            self._s, qn.ctypes.data, None if q_pq is None else q_pq.ctypes.data, len(qn),
            out.ctypes.data))
        self._keep[t % 64] = (qn, q_pq, out)
        return t

    def set_probes(self, n_probes, pass_1=None):
        _lib.check(_lib.lib().tk_stream_set_probes(self._s, int(n_probes), int(pass_1 or 0)))

    def wait(self, ticket):
        _lib.check(_lib.lib().tk_stream_wait(self._s, int(ticket)))

    def drain(self):
        _lib.check(_lib.lib().tk_stream_drain(self._s))
        self._keep.clear()

    def prepare_seconds(self):
        return _lib.lib().tk_stream_prepare_seconds(self._s)

    def close(self):
        if getattr(self, "_s", None):
            if _lib.owns_handles():
                _lib.lib().tk_stream_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentData:
    """Stands where IVF.data stood when the vectors live in HBM only (IVF.build_resident):
    shape, dtype and rows by id (tk_index_read_rows)."""

    def __init__(self, dev):
        self._dev = dev
        self.shape = (dev.N, dev.d)
        self.dtype = np.dtype(np.float32)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, rows):
        rows = np.asarray(rows, dtype=np.int64)
        return self._dev.read_rows(rows).reshape(rows.shape + (self.shape[1],))


def _copies_carry_one_code(ivf, n_lists):
    """Rows with equal ids lie in different lists and carry equal codes — what IVF.build's lists have by construction
    (ivf.py:77-102) and what the lane replay's TWIN form rests on (heap.hip).  The library checks it on the device
    where it holds every list's codes; a rank of a list-sharded index uploads only its own lists, so this is the
    same check on the host, over all lists, before the rank vouches (TK_OPT_TWIN_VOUCH)."""
    # Two independent multilinear hashes (mod 2^64, fixed odd-seeded multipliers) per row instead of the M / 2 code bytes
    # themselves: 16 B per stored row on every rank at construction, not 3 x M / 2 (a 100M-row build(n_probes=2) index
    # is 200M stored rows).  Unequal codes hash alike with probability < 2^-100.
    labels, lists, sigs = [], [], []
    mult = None
    for i in range(n_lists):
        td = ivf.pq_transformed_points[i]
        if isinstance(td, np.ndarray) or td.size == 0:
            continue
        pk = np.ascontiguousarray(td.packed, dtype=np.uint64)              # (chunks, M): per chunk M / 2 groups of 16 bytes
        P = pk.shape[1] // 2
        if mult is None or mult.shape[0] != P:
            mult = np.random.RandomState(0x7151).randint(0, 2 ** 63, size=(P, 2), dtype=np.int64).astype(np.uint64) * 2 + 1
        h = np.empty((td.size, 2), dtype=np.uint64)
        by = pk.view(np.uint8).reshape(pk.shape[0], P, 16)                 # [chunk][pair][row of the chunk]
        for lo in range(0, pk.shape[0], 4096):                              # (bounded temporaries)
            part = by[lo:lo + 4096].transpose(0, 2, 1).reshape(-1, P)      # rows of these chunks, a code per row
            n = min(part.shape[0], td.size - 16 * lo)
            if n > 0:
                h[16 * lo:16 * lo + n] = part[:n].astype(np.uint64) @ mult
        labels.append(np.asarray(ivf.ids[i], dtype=np.int64)[:td.size])
        lists.append(np.full(td.size, i, dtype=np.int32))
        sigs.append(h)
    if not labels:
        return True
    labels, lists, sigs = np.concatenate(labels), np.concatenate(lists), np.concatenate(sigs)
    order = np.lexsort((lists, labels))
    la, li = labels[order], lists[order]
    same = la[1:] == la[:-1]
    if (same & (li[1:] == li[:-1])).any():          # two copies in one list
        return False
    del la, li
    sg = sigs[order]
    return bool((sg[1:][same] == sg[:-1][same]).all())


class DeviceIndex:
    """HBM-resident copy of a built IVF (C ABI: tk_index_*)."""

    def __init__(self, ivf, owner=None, rank=0, world=1):
        """owner (n_lists,) int32 + rank/world: list-sharded index — only the codes of the
        lists with owner[l] == rank are uploaded (tinyknn_hip.h, tk_index_set_lists_shard)."""
        L = _lib.lib()
        pq = ivf.pq
        self._h = L.tk_index_create()
        if not self._h:
            raise _lib.TinyKnnHipError(L.tk_last_error().decode() or "tk_index_create failed")
        centers = pq.centers
        f_order = int(not centers.flags.c_contiguous)
        c32 = np.ascontiguousarray(centers, dtype=np.float32)
        self.dq = c32.shape[1]
        self.dpb = pq.dims_per_block
        order = _lib.ORDER_AVX if avx else _lib.ORDER_SSE
        _lib.check(L.tk_index_set_pq(self._h, _lib.ptr(c32, _lib._f32p), self.dq, self.dpb,
                                     f_order, float(pq.sqrt_n_blocks), order))
        ac = np.ascontiguousarray(ivf.active_centers, dtype=np.float32)
        self.n_lists, self.d = ac.shape
        csize, cpacked = ivf.pq_transformed_centers
        assert csize == self.n_lists
        cpacked = np.ascontiguousarray(cpacked, dtype=np.uint64)
        _lib.check(L.tk_index_set_centers(self._h, _lib.ptr(ac, _lib._f32p), self.n_lists, self.d,
                                          _lib.ptr(cpacked, _lib._u64p), cpacked.shape[0]))
        M = self.dq // self.dpb
        sizes, packed, ids = [], [], []
        for i in range(self.n_lists):
            td = ivf.pq_transformed_points[i]
            if isinstance(td, np.ndarray):      # FastPQ.transform(empty) returns the raw array
                sizes.append(0)
                continue
            sizes.append(td.size)
            if owner is None or owner[i] == rank:
                packed.append(np.ascontiguousarray(td.packed, dtype=np.uint64))
            ids.append(np.asarray(ivf.ids[i], dtype=np.int64)[:td.size])
        sizes = np.array(sizes, dtype=np.int64)
        codes = (np.ascontiguousarray(np.concatenate(packed)) if packed
                 else np.zeros((1, M), dtype=np.uint64))
        allids = (np.ascontiguousarray(np.concatenate(ids)) if ids else np.zeros(1, np.int64))
        self.list_sizes = sizes
        self.rank, self.world = int(rank), int(world)
        if owner is None:
            _lib.check(L.tk_index_set_lists(self._h, _lib.ptr(sizes, _lib._i64p),
                                            _lib.ptr(codes, _lib._u64p),
                                            _lib.ptr(allids, _lib._i64p)))
        else:
            own = np.ascontiguousarray(owner, dtype=np.int32)
            assert own.shape == (self.n_lists,)
            _lib.check(L.tk_index_set_lists_shard(self._h, _lib.ptr(sizes, _lib._i64p),
                                                  _lib.ptr(own, _lib._i32p), self.rank,
                                                  self.world, _lib.ptr(codes, _lib._u64p),
                                                  _lib.ptr(allids, _lib._i64p)))
        # IVF.data keeps the dtype of the X passed to build (ivf.py:77); float64 vectors
        # are rescored in float64 like numpy would
        is64 = ivf.data.dtype != np.float32
        data = np.ascontiguousarray(ivf.data, dtype=np.float64 if is64 else np.float32)
        _lib.check(L.tk_index_set_data(self._h, data.ctypes.data, int(is64), data.shape[0],
                                       data.shape[1]))
        self.code_bytes = int(codes.nbytes)
        self.angular = ivf.metric == "angular"
        self._R = None
        self._streams = {}
        self._live_streams = weakref.WeakSet()
        if pq.R is not None:    # fast mode (device front end) needs the rotation on the device
            R = np.ascontiguousarray(pq.R, dtype=np.float64)
            _lib.check(L.tk_index_set_rotation(self._h, R.ctypes.data, R.shape[1]))
            self._R = R
        # labels that repeat on a list-sharded rank: the library cannot check the twin table's premises against codes
        # it was not given — the host can (all lists are here), and vouches
        if owner is not None and self.twin_table_width() > 0 and _copies_carry_one_code(ivf, self.n_lists):
            self.set_option(_lib.OPT_TWIN_VOUCH, 1)

    @classmethod
    def resident(cls, ivf, N, d):
        """An index whose N float32 vectors are produced IN HBM and never visit the host
        (tk_index_alloc_data; filled by synth_data or through `data_ptr`, then build_dev)."""
        L = _lib.lib()
        pq = ivf.pq
        self = cls.__new__(cls)
        self._h = L.tk_index_create()
        if not self._h:
            raise _lib.TinyKnnHipError(L.tk_last_error().decode() or "tk_index_create failed")
        c32 = np.ascontiguousarray(pq.centers, dtype=np.float32)
        self.dq, self.dpb = c32.shape[1], pq.dims_per_block
        _lib.check(L.tk_index_set_pq(self._h, _lib.ptr(c32, _lib._f32p), self.dq, self.dpb,
                                     int(not pq.centers.flags.c_contiguous), float(pq.sqrt_n_blocks),
                                     _lib.ORDER_AVX if avx else _lib.ORDER_SSE))
        self.data_ptr = L.tk_index_alloc_data(self._h, int(N), int(d))
        if not self.data_ptr:
            raise _lib.TinyKnnHipError(L.tk_last_error().decode() or "tk_index_alloc_data failed")
        self.N, self.d, self.n_lists = int(N), int(d), 0
        self.rank, self.world = 0, 1
        self.angular = ivf.metric == "angular"
        self._R = None if pq.R is None else np.ascontiguousarray(pq.R, dtype=np.float64)
        self._streams = {}
        self._live_streams = weakref.WeakSet()
        self.list_sizes = None
        self.code_bytes = 0
        return self

    def synth_data(self, seed, centres=None, sigma=1.0, row0=0, n=None):
        """rows [row0, row0 + n) = centres[c(row)] + sigma * N(0, 1) from the seeded
        counter-based generator (devbuild.hip): a pure function of (seed, row)."""
        n = self.N - row0 if n is None else n
        c = None if centres is None else np.ascontiguousarray(centres, dtype=np.float32)
        _lib.check(_lib.lib().tk_index_synth_data(
            self._h, int(row0), int(n), int(seed), None if c is None else c.ctypes.data,
            0 if c is None else len(c), float(sigma)))

    def build_dev(self, all_centers, n_probes=1):
        """IVF.build(n_probes=1 or 2) on the resident vectors (tk_index_build_dev) -> n_active."""
        A = np.ascontiguousarray(all_centers, dtype=np.float32)
        Y = A
        if self.angular:
            Y = np.ascontiguousarray(Y / np.linalg.norm(Y, axis=1, keepdims=True))   # utils.py:75
        ynorm2 = np.ascontiguousarray(np.einsum("ij,ij->i", Y, Y))    # utils.py:80
        n_active = C.c_int64(0)
        R = self._R
        _lib.check(_lib.lib().tk_index_build_dev(
            self._h, int(self.angular), _lib.ptr(A, _lib._f32p), _lib.ptr(Y, _lib._f32p),
            _lib.ptr(ynorm2, _lib._f32p), len(Y), int(n_probes),
            None if R is None else R.ctypes.data, 0 if R is None else R.shape[1], C.byref(n_active)))
        self.n_lists = int(n_active.value)
        self.list_sizes = self.export_lists(codes=False, ids=False)[0]
        self.code_bytes = int(((self.list_sizes + 15) // 16).sum()) * (self.dq // self.dpb) * 8
        return self.n_lists

    def shard_resident(self, owner, rank, world):
        """This complete index becomes rank `rank`'s shard of a list-sharded index, in place
        (tk_index_shard_resident): only the codes of the lists it owns stay in HBM."""
        own = np.ascontiguousarray(owner, dtype=np.int32)
        assert own.shape == (self.n_lists,)
        prev = getattr(self, "_sharded_as", None)
        if prev is not None:        # a second ListShardedIndex on the same IVF: same partition only
            if prev[1:] == (int(rank), int(world)) and np.array_equal(prev[0], own):
                return
            raise RuntimeError("this index is already list-sharded in place with another partition")
        _lib.check(_lib.lib().tk_index_shard_resident(self._h, _lib.ptr(own, _lib._i32p), int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)
        self._sharded_as = (own.copy(), int(rank), int(world))

    def export_lists(self, codes=True, ids=True):
        """(list_sizes, packed codes (chunks, M) uint64 or None, ids or None) back on the host
        in the reference's formats (tk_index_export_lists)."""
        L = _lib.lib()
        sizes = np.zeros(self.n_lists, dtype=np.int64)
        _lib.check(L.tk_index_export_lists(self._h, sizes.ctypes.data, None, None))
        M = self.dq // self.dpb
        pk = np.zeros((int(((sizes + 15) // 16).sum()), M), dtype=np.uint64) if codes else None
        lab = np.zeros(int(sizes.sum()), dtype=np.int64) if ids else None
        if codes or ids:
            _lib.check(L.tk_index_export_lists(self._h, None, None if pk is None else pk.ctypes.data,
                                               None if lab is None else lab.ctypes.data))
        return sizes, pk, lab

    def export_centers(self):
        """(active_centers (n_lists, d) float32, their packed codes) (tk_index_export_centers)."""
        ac = np.zeros((self.n_lists, self.d), dtype=np.float32)
        cc = np.zeros(((self.n_lists + 15) // 16, self.dq // self.dpb), dtype=np.uint64)
        _lib.check(_lib.lib().tk_index_export_centers(self._h, ac.ctypes.data, cc.ctypes.data))
        return ac, cc

    def read_rows(self, rows):
        """IVF.data[rows] for float32 vectors held in HBM (tk_index_read_rows)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64).ravel()
        out = np.zeros((len(rows), self.d), dtype=np.float32)
        _lib.check(_lib.lib().tk_index_read_rows(self._h, _lib.ptr(rows, _lib._i64p), len(rows),
                                                 _lib.ptr(out, _lib._f32p)))
        return out

    @property
    def handle(self):
        return self._h

    def close(self):
        # every session on this index, the caller's own included: a session that outlived the
        # index would drain and join a freed handle
        for st in list(getattr(self, "_live_streams", ())):
            st.close()
        self._streams = {}
        if getattr(self, "_h", None):
            if _lib.owns_handles():
                _lib.lib().tk_index_destroy(self._h)
            self._h = None

    # ---- streaming (exact) ---------------------------------------------------
    def max_sub_batch(self, k, n_probes, pass_1=None):
        return _lib.check(_lib.lib().tk_index_max_sub_batch(self._h, int(k), int(n_probes),
                                                            int(pass_1 or 0)))

    def stream(self, max_nq, k, n_probes, pass_1=None, slots=8):
        """A new streaming session (see QueryStream); the caller closes it."""
        return QueryStream(self, max_nq, k, n_probes, pass_1, slots)

    CHUNK = 10000       # queries per sub-batch of the chunked host API

    def _cached_stream(self, nq, k, n_probes, pass_1):
        """One session per k: its page-locked staging does not depend on n_probes, so a sweep
        over n_probes only re-sizes the index's workspaces (tk_stream_set_probes)."""
        key = int(k)
        chunk = min(self.CHUNK, self.max_sub_batch(k, n_probes, pass_1))
        want = min(int(nq), chunk)
        st = self._streams.get(key)
        if st is not None and (st.max_nq < want or st.max_nq > chunk):
            st.close()
            st = None
        if st is None:
            cap = min(chunk, max(64, 1 << (want - 1).bit_length()))
            st = self._streams[key] = QueryStream(self, cap, k, n_probes, pass_1)
        else:
            st.set_probes(n_probes, pass_1)
        return st

    def query_raw(self, qs, k, n_probes, pass_1=None):
        """Exact IVF.query for every row of qs (raw float32 queries on the host): chunks of
        up to CHUNK rows through a streaming session, so that the host preparation and the
        copies of a chunk overlap the kernels of the chunks before it."""
        qs = np.ascontiguousarray(qs, dtype=np.float32)
        nq = qs.shape[0]
        assert qs.shape[1] == self.d
        out = np.full((nq, k), -1, dtype=np.int64)
        if nq == 0:
            return out
        st = self._cached_stream(nq, k, n_probes, pass_1)
        for o in range(0, nq, st.max_nq):
            st.submit(qs[o:o + st.max_nq], out[o:o + st.max_nq])
        st.drain()
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query_batch(self, qn, q_pq, k, n_probes, pass_1=None, debug=False):
        """qn: (nq, d) float32 normalised queries; q_pq: (nq, dq) table-build queries."""
        qn = np.ascontiguousarray(qn, dtype=np.float32)
        is64 = q_pq.dtype != np.float32
        q_pq = np.ascontiguousarray(q_pq, dtype=np.float64 if is64 else np.float32)
        nq = qn.shape[0]
        assert qn.shape[1] == self.d and q_pq.shape == (nq, self.dq)
        out = np.full((nq, k), -1, dtype=np.int64)
        # the session pads unrotated queries on the device: only for q_pq = pad1(qn)
        plain = is64 or (np.array_equal(q_pq[:, :self.d], qn) and not q_pq[:, self.d:].any())
        if not debug and nq > 0 and plain and _front.bind():
            # prepared rows through the streaming session (pinned staging, async copies)
            st = self._cached_stream(nq, k, n_probes, pass_1)
            for o in range(0, nq, st.max_nq):
                st.submit_prepared(qn[o:o + st.max_nq],
                                   q_pq[o:o + st.max_nq] if is64 else None, out[o:o + st.max_nq])
            st.drain()
            return out
        R = pass_1 if pass_1 else (n_probes + 1) * k + 1
        probes = hidx = hval = None
        if debug:
            probes = np.zeros((nq, min(n_probes, self.n_lists)), dtype=np.int64)
            hidx = np.zeros((nq, R), dtype=np.int64)
            hval = np.zeros((nq, R), dtype=np.int32)
        _lib.check(_lib.lib().tk_index_query_batch(
            self._h, _lib.ptr(qn, _lib._f32p), q_pq.ctypes.data, int(is64), nq, int(k),
            int(n_probes), int(pass_1 or 0), _lib.ptr(out, _lib._i64p),
            None if probes is None else _lib.ptr(probes, _lib._i64p),
            None if hidx is None else _lib.ptr(hidx, _lib._i64p),
            None if hval is None else _lib.ptr(hval, _lib._i32p)))
        if debug:
            return out, dict(probes=probes, heap_idx=hidx, heap_val=hval)
        return out

    def query_batch_raw(self, qs, k, n_probes, pass_1=None):
        """Fast mode: raw float32 queries, normalisation / padding / rotation on the device
        (tk_index_prepare_dev: within 1 ulp of the host's BLAS results, not bit-identical)."""
        qs = np.ascontiguousarray(qs, dtype=np.float32)
        assert qs.shape[1] == self.d
        out = np.full((qs.shape[0], k), -1, dtype=np.int64)
        _lib.check(_lib.lib().tk_index_query_batch_raw(
            self._h, _lib.ptr(qs, _lib._f32p), qs.shape[0], int(self.angular), int(k), int(n_probes),
            int(pass_1 or 0), _lib.ptr(out, _lib._i64p)))
        return out

    def knn_brute(self, qn, k):
        """Exact k nearest rows of IVF.data for normalised queries (ground truth of recall;
        tk_index_knn_brute: f32 MFMA, numpy's distances bit for bit), ascending."""
        qn = np.ascontiguousarray(qn, dtype=np.float32)
        assert qn.shape[1] == self.d
        out = np.empty((qn.shape[0], k), dtype=np.int64)
        _lib.check(_lib.lib().tk_index_knn_brute(self._h, _lib.ptr(qn, _lib._f32p), qn.shape[0],
                                                 int(k), _lib.ptr(out, _lib._i64p)))
        return out

    def query_batch_dev(self, qn_ptr, qpq_ptr, qpq_is_f64, nq, k, n_probes, out_ptr,
                        pass_1=None, stream=0, done_event=None):
        """Device pointers in, device pointer out, enqueued on `stream` (no sync).
        done_event: a hipEvent_t (integer handle) recorded behind the batch's last kernel, on
        whichever internal stream that runs (tk_index_query_batch_dev_ex).
        With set_pipeline(depth > 1) the kernels of a call run up to three calls later, and with
        set_coalesce(2) they read the queries from, and write the ids to, these very buffers (no
        staging copy): all three buffers belong to the library until join() — or the call's
        done_event — as include/tinyknn_hip.h says for tk_index_set_pipeline."""
        if done_event is None:
            _lib.check(_lib.lib().tk_index_query_batch_dev(
                self._h, qn_ptr, qpq_ptr, int(qpq_is_f64), nq, int(k), int(n_probes),
                int(pass_1 or 0), out_ptr, stream))
        else:
            _lib.check(_lib.lib().tk_index_query_batch_dev_ex(
                self._h, qn_ptr, qpq_ptr, int(qpq_is_f64), nq, int(k), int(n_probes),
                int(pass_1 or 0), out_ptr, None, C.c_void_p(int(done_event)), stream))

    def shard_coarse_dev(self, slot, qn_ptr, qpq_ptr, qpq_is_f64, nq, k, n_probes, pass_1,
                         probes_home_ptr, stream=0):
        """Tables for all queries + the coarse stage of this rank's home queries
        (tk_index_shard_coarse_dev); the caller all-gathers the probe lists."""
        _lib.check(_lib.lib().tk_index_shard_coarse_dev(
            self._h, int(slot), qn_ptr, qpq_ptr, int(qpq_is_f64), nq, int(k), int(n_probes),
            int(pass_1 or 0), probes_home_ptr, stream))

    def shard_scan_dev(self, slot, qn_ptr, qpq_ptr, qpq_is_f64, nq, k, n_probes, pass_1, capacity,
                       send_ptr, flag_ptr, stream=0, probes_all_ptr=None):
        """Scan of the owned segments into the send buffer (tk_index_shard_scan_dev);
        probes_all_ptr None: the replicated coarse stage runs inside this call."""
        _lib.check(_lib.lib().tk_index_shard_scan_dev(
            self._h, int(slot), qn_ptr, qpq_ptr, int(qpq_is_f64), nq, int(k), int(n_probes),
            int(pass_1 or 0), probes_all_ptr, int(capacity), send_ptr, flag_ptr, stream))

    def shard_finish_dev(self, slot, qn_ptr, nq, k, n_probes, pass_1, capacity, recv_ptr, out_ptr,
                         stream=0, flag_ptr=None):
        """Second half, after the all-to-all (tk_index_shard_finish_dev).  flag_ptr: the batch's flag
        word — required behind shard_scan_plain_dev (bit 4: a home query failed the plain path's check)."""
        _lib.check(_lib.lib().tk_index_shard_finish_dev(
            self._h, int(slot), qn_ptr, nq, int(k), int(n_probes), int(pass_1 or 0),
            int(capacity), recv_ptr, out_ptr, flag_ptr, stream))

    def shard_scan_plain_dev(self, slot, qn_ptr, qpq_ptr, qpq_is_f64, nq, k, n_probes, pass_1, capacity,
                             send_ptr, flag_ptr, stream=0, probes_all_ptr=None, bound_ptr=None):
        """The owned segments in ONE phase as the unsharded pipeline scores them: heads exactly, the rest
        as plain sums on the matrix cores, checked by the home rank's replay
        (tk_index_shard_scan_plain_dev; falls back to shard_scan_dev's kernel where that does not apply).
        bound_ptr: the min-reduced bounds of shard_scan_head_dev — queries above their table's limit
        stay exact, the check at home cannot fail."""
        _lib.check(_lib.lib().tk_index_shard_scan_plain_dev(
            self._h, int(slot), qn_ptr, qpq_ptr, int(bool(qpq_is_f64)), nq, int(k), int(n_probes),
            int(pass_1 or 0), probes_all_ptr, int(capacity), send_ptr, flag_ptr, bound_ptr, stream))

    def shard_scan_head_dev(self, slot, qn_ptr, qpq_ptr, qpq_is_f64, nq, k, n_probes, pass_1, capacity,
                            send_ptr, flag_ptr, bound_ptr, stream=0, probes_all_ptr=None):
        """Heads of the first probed lists this rank owns, exactly, + the bound after them
        (tk_index_shard_scan_head_dev); the caller min-reduces the bytes, then shard_scan_plain_dev(bound_ptr)."""
        _lib.check(_lib.lib().tk_index_shard_scan_head_dev(
            self._h, int(slot), qn_ptr, qpq_ptr, int(bool(qpq_is_f64)), nq, int(k), int(n_probes),
            int(pass_1 or 0), probes_all_ptr, int(capacity), send_ptr, flag_ptr, bound_ptr, stream))

    def clone_shard(self, owner, rank, world):
        """Rank `rank`'s shard of this complete unsharded index as a NEW handle on the same device
        (tk_index_clone_shard): it borrows the replicated arrays — this index must outlive it — and owns
        the codes of its lists.  How several ranks of a list partition are played on one GPU."""
        own = np.ascontiguousarray(owner, dtype=np.int32)
        assert own.shape == (self.n_lists,)
        h = _lib.lib().tk_index_clone_shard(self._h, _lib.ptr(own, _lib._i32p), int(rank), int(world))
        if not h:
            raise _lib.TinyKnnHipError(_lib.lib().tk_last_error().decode() or "tk_index_clone_shard failed")
        c = DeviceIndex.__new__(DeviceIndex)
        c._h = h
        c._source = self            # keeps the lender alive
        for a in ("dq", "dpb", "n_lists", "d", "list_sizes", "angular", "_R", "code_bytes"):
            setattr(c, a, getattr(self, a, None))
        c.N = getattr(self, "N", None)
        c.rank, c.world = int(rank), int(world)
        c._streams = {}
        c._live_streams = weakref.WeakSet()
        return c

    def shard_usage(self, slot):
        """Longest stream (uint4) of the slot's last shard_scan_dev (tk_index_shard_usage; syncs)."""
        import ctypes
        v = ctypes.c_int64(0)
        _lib.check(_lib.lib().tk_index_shard_usage(self._h, int(slot), ctypes.byref(v)))
        return int(v.value)

    def shard_bound_dev(self, slot, nq, k, n_probes, pass_1, capacity, scan_ptr, bound_ptr, stream=0):
        """Bound after the first probed list, for the queries whose first list this rank owns
        (tk_index_shard_bound_dev); the caller min-reduces the bytes over the ranks."""
        _lib.check(_lib.lib().tk_index_shard_bound_dev(
            self._h, int(slot), nq, int(k), int(n_probes), int(pass_1 or 0), int(capacity),
            scan_ptr, bound_ptr, stream))

    def shard_filter_dev(self, slot, nq, k, n_probes, pass_1, capacity, scan_ptr, bound_ptr,
                         counts_ptr, records_ptr, stream=0):
        """Blocks below the bound as records grouped by home rank, compact (tk_index_shard_filter_dev, region 0)."""
        _lib.check(_lib.lib().tk_index_shard_filter_dev(
            self._h, int(slot), nq, int(k), int(n_probes), int(pass_1 or 0), int(capacity),
            scan_ptr, bound_ptr, counts_ptr, records_ptr, 0, None, None, stream))

    def shard_finish_filtered_dev(self, slot, qn_ptr, nq, k, n_probes, pass_1, records_ptr,
                                  n_records, out_ptr, flag_ptr, stream=0):
        """Received compact records -> rows, replay, rescoring (tk_index_shard_finish_filtered_dev)."""
        _lib.check(_lib.lib().tk_index_shard_finish_filtered_dev(
            self._h, int(slot), qn_ptr, nq, int(k), int(n_probes), int(pass_1 or 0), records_ptr,
            int(n_records), None, 0, out_ptr, flag_ptr, stream))

    def shard_plain(self, k, n_probes, pass_1=None):
        """Does the two-phase scan with the matrix-core kernel apply (tk_index_shard_plain)?"""
        r = _lib.lib().tk_index_shard_plain(self._h, int(k), int(n_probes), int(pass_1 or 0))
        if r < 0:
            _lib.check(r)
        return bool(r)

    def shard_scan_first_dev(self, slot, qn_ptr, qpq_ptr, qpq_is_f64, nq, k, n_probes, pass_1, capacity,
                             send_ptr, flag_ptr, bound_ptr, stream=0, probes_all_ptr=None):
        """Phase 1 of the two-phase sharded scan: first slots exactly + their bound
        (tk_index_shard_scan_first_dev); the caller min-reduces the bound over the ranks."""
        _lib.check(_lib.lib().tk_index_shard_scan_first_dev(
            self._h, int(slot), qn_ptr, qpq_ptr, int(bool(qpq_is_f64)), nq, int(k), int(n_probes),
            int(pass_1 or 0), probes_all_ptr, int(capacity), send_ptr, flag_ptr, bound_ptr, stream))

    def shard_scan_rest_dev(self, slot, nq, k, n_probes, pass_1, capacity, send_ptr, bound_ptr, stream=0):
        """Phase 2: the slots behind the first, on the plain kernel where the reduced bound allows
        (tk_index_shard_scan_rest_dev)."""
        _lib.check(_lib.lib().tk_index_shard_scan_rest_dev(
            self._h, int(slot), nq, int(k), int(n_probes), int(pass_1 or 0), int(capacity), send_ptr,
            bound_ptr, stream))

    def shard_plain_stats(self, slot=0):
        """What the slot's last two-phase scan did on this rank (tk_index_shard_plain_stats)."""
        o = np.zeros(4, dtype=np.int64)
        _lib.check(_lib.lib().tk_index_shard_plain_stats(self._h, int(slot), _lib.ptr(o, _lib._i64p)))
        return dict(plain_pairs=int(o[0]), plain_tiles=int(o[1]), exact_pair_records_behind_first=int(o[2]),
                    plain_queries=int(o[3]))

    def shard_filter_regions_dev(self, slot, nq, k, n_probes, pass_1, capacity, scan_ptr, bound_ptr,
                                 counts_ptr, records_ptr, region, flag_ptr, acc_ptr=None, stream=0):
        """... into fixed regions of `region` records per home rank (tk_index_shard_filter_dev, region >= 1)."""
        _lib.check(_lib.lib().tk_index_shard_filter_dev(
            self._h, int(slot), nq, int(k), int(n_probes), int(pass_1 or 0), int(capacity),
            scan_ptr, bound_ptr, counts_ptr, records_ptr, int(region), flag_ptr, acc_ptr, stream))

    def shard_finish_regions_dev(self, slot, qn_ptr, nq, k, n_probes, pass_1, records_ptr,
                                 counts_recv_ptr, region, out_ptr, flag_ptr, stream=0):
        """Received regions + their counts on the device -> rows, replay, rescoring
        (tk_index_shard_finish_filtered_dev with the received counts)."""
        _lib.check(_lib.lib().tk_index_shard_finish_filtered_dev(
            self._h, int(slot), qn_ptr, nq, int(k), int(n_probes), int(pass_1 or 0), records_ptr,
            0, counts_recv_ptr, int(region), out_ptr, flag_ptr, stream))

    def replay_stats(self):
        """What the lane replays of the probed lists did since set_option(OPT_REPLAY_COUNT, 1) / the last call
        (tk_index_replay_stats; synchronises): insert rounds over all waves, the most one wave ran, waves,
        16-block segments walked."""
        o = np.zeros(4, dtype=np.int64)
        _lib.check(_lib.lib().tk_index_replay_stats(self._h, _lib.ptr(o, _lib._i64p)))
        return dict(rounds=int(o[0]), max_rounds_of_a_wave=int(o[1]), waves=int(o[2]), segments=int(o[3]))

    def twin_table_width(self):
        """Other copies listed per stored row (tk_index_twin_table): 0 = no table (distinct labels, ...)."""
        w = np.zeros(1, np.int32)
        _lib.check(_lib.lib().tk_index_twin_table(self._h, None, _lib.ptr(w, _lib._i32p), None, None))
        return int(w[0])

    def twin_table(self):
        """(list, offset) of every stored row's other copies, two (rows, w) int32 arrays (tk_index_twin_table):
        what the lane replay decides `insert`'s duplicate test from where labels repeat (build n_probes >= 2).
        w = 0: no table."""
        rows, w = np.zeros(1, np.int64), np.zeros(1, np.int32)
        _lib.check(_lib.lib().tk_index_twin_table(self._h, _lib.ptr(rows, _lib._i64p), _lib.ptr(w, _lib._i32p), None, None))
        tl = np.zeros((int(rows[0]), int(w[0])), np.int32)
        to = np.zeros_like(tl)
        if tl.size:
            _lib.check(_lib.lib().tk_index_twin_table(self._h, None, None, _lib.ptr(tl, _lib._i32p), _lib.ptr(to, _lib._i32p)))
        return tl, to

    def reserve(self, nq, k, n_probes, pass_1=None):
        _lib.check(_lib.lib().tk_index_reserve(self._h, nq, int(k), int(n_probes), int(pass_1 or 0)))

    def set_pipeline(self, depth):
        """Number of batches in flight for query_batch_dev (see tinyknn_hip.h)."""
        _lib.check(_lib.lib().tk_index_set_pipeline(self._h, int(depth)))

    def set_option(self, option, value):
        """Per-index A/B and test options (tk_index_set_option): _lib.OPT_PAIR_NQ (batches of up to this many queries
        replay their heaps one query per wave, heap in registers: 8192 one batch at a time, at most 4096 per launch pipelined),
        _lib.OPT_LABELS24, _lib.OPT_SCAN_FORM,
        _lib.OPT_RESCORE_FORM, _lib.OPT_PLAIN_LIMIT, _lib.OPT_REPLAY_LAZY, _lib.OPT_REPLAY_COUNT,
        _lib.OPT_REPLAY_TWIN."""
        _lib.check(_lib.lib().tk_index_set_option(self._h, int(option), int(value)))

    def set_coalesce(self, n):
        """2: pairs of consecutive query_batch_dev calls run as one batch (tk_index_set_coalesce);
        the first call of a pair is held until the second arrives (join() launches it alone).  A
        held call has enqueued nothing: a `done_event` passed to it is recorded only once the partner
        call or join() has run — do not wait on it before (pending() > 0 says work is still owed)."""
        _lib.check(_lib.lib().tk_index_set_coalesce(self._h, int(n)))

    def join(self, stream=0):
        _lib.check(_lib.lib().tk_index_join(self._h, stream))

    def plain_stats(self):
        """What the plain (matrix-core) scan did for the last batch (tk_index_plain_stats)."""
        o = np.zeros(8, dtype=np.int64)
        _lib.check(_lib.lib().tk_index_plain_stats(self._h, _lib.ptr(o, _lib._i64p)))
        return dict(plain_units=int(o[0]), plain_pairs=int(o[1]), exact_pair_records=int(o[2]),
                    head_pair_records=int(o[3]), flagged_queries=int(o[4]), plain_unit_chunk_pairs=int(o[5]),
                    state=("probe", "wait", "on", "paused")[int(o[6]) & 3], pause_left=int(o[7]))

    def quiesce(self):
        """Wait for everything enqueued and forget its completion events (before a stream
        capture of the pipelined mode: tinyknn_hip.h, tk_index_quiesce)."""
        _lib.check(_lib.lib().tk_index_quiesce(self._h))

    def set_heap_mode(self, mode):
        """0: automatic (small batches: one query per wave with the heap in registers — _lib.OPT_PAIR_NQ; else one
        query per lane), 1: general wave kernel, 2: packed wave kernel, 3: the register heap for every batch size
        (heaps of up to 513 entries)."""
        _lib.check(_lib.lib().tk_index_set_heap_mode(self._h, int(mode)))

    def set_scan_mode(self, mode):
        """0: automatic, 1: query-major scan, 2: list-major scan (see tinyknn_hip.h)."""
        _lib.check(_lib.lib().tk_index_set_scan_mode(self._h, int(mode)))

    def set_plain_scan(self, on):
        """Probed lists behind the first ones as plain sums on the int8 matrix cores where that
        is provably the same replay (tinyknn_hip.h: tk_index_set_plain_scan); True = automatic
        (default: the path has to prove itself on a probe batch, and is paused while more than
        1 % of the queries need the exact re-scan), False = the exact kernel for every list,
        "always" = no pausing (tests, A/B).  Identical results in every mode."""
        mode = 2 if on == "always" else (0 if on else 1)
        _lib.check(_lib.lib().tk_index_set_plain_scan(self._h, mode))

    def set_profiling(self, on):
        _lib.check(_lib.lib().tk_index_set_profiling(self._h, int(on)))

    def last_profile(self):
        ms = (C.c_float * 8)()
        b = C.c_double()
        n = C.c_int32()
        _lib.check(_lib.lib().tk_index_last_profile(self._h, ms, C.byref(b), C.byref(n)))
        names = ["tables", "coarse_scan", "coarse_heap", "coarse_rescore", "scan", "heap", "rescore"]
        self.last_plain_kernel_ms = float(ms[7])      # the plain kernel alone (0: no recorded batch ran it)
        return dict(zip(names, list(ms)[:7])), b.value, n.value


class IVF:
    """reference: ivf.py:8-163"""

    def __init__(self, metric, n_clusters, pq=None):
        assert metric in ["euclidean", "angular"]
        self.metric = metric
        self.pq = FastPQ(dims_per_block=2) if pq is None else pq
        assert self.pq.centers is None, "PQ should not be pre-fitted"
        self.pq_transformed_points = [None] * n_clusters
        self.pq_transformed_centers = [None] * n_clusters
        self.n_clusters = n_clusters
        self.ids = [None] * n_clusters
        self._dev = None

    # device handles are not picklable (the reference pickles (pq, ivf), bench.py:88-103)
    def __getstate__(self):
        self._require_host_copy("pickle")
        st = dict(self.__dict__)
        st["_dev"] = None
        return st

    def _require_host_copy(self, what):
        """An index built in HBM (build_resident) keeps its lists, codes and vectors on the
        device only; what needs them on the host says so instead of failing on a None."""
        if getattr(self, "pq_transformed_points", 0) is None:
            raise RuntimeError(f"IVF.{what}: this index was built in HBM (build_resident): its lists, codes "
                               "and vectors live on the device only — device_index().export_lists() / "
                               "read_rows() fetch them")

    # ---- offline ---------------------------------------------------------
    def fit(self, X, verbose=False):
        """Coarse k-means centres + PQ codebook.  reference: ivf.py:19-51"""
        import sklearn.cluster
        n, d = X.shape
        assert n >= 1
        with timer(verbose, "Fitting IVF cluster centers..."):
            km = sklearn.cluster.KMeans(n_clusters=self.n_clusters, n_init=1, verbose=verbose)
            if self.metric == "angular":
                # spherical data: normalise the points, then the centres (ivf.py:38-45)
                X = X / np.linalg.norm(X, axis=1, keepdims=True)
                self.all_centers = km.fit(X).cluster_centers_
                self.all_centers /= np.linalg.norm(self.all_centers, axis=1, keepdims=True)
            else:
                self.all_centers = km.fit(X).cluster_centers_
        with timer(verbose, "Fitting PQ to data..."):
            self.pq.fit(X, verbose=verbose)
        return self

    def build(self, X, n_probes=2, verbose=False, device=None):
        """Assign every point to its n_probes nearest centres and encode the lists.
        reference: ivf.py:53-104.  device=True (default: fast_pq.device_build): the two
        searches — nearest centres per point, nearest centroid per block — run on the GPU
        (build.hip) and give the lists and codes numpy gives; everything else (normalisation,
        the rotation GEMM, grouping, packing) is the same host code."""
        assert n_probes <= self.n_clusters, (
            f"Can't assign points to {n_probes} clusters, as index only has {self.n_clusters}")
        from . import fast_pq as _fp
        device = _fp.device_build if device is None else device
        self._dev = None
        self.data = data = X.copy()
        if self.metric == "angular":
            data /= np.linalg.norm(data, axis=1, keepdims=True)
        with timer(verbose, "Computing nearest clusters..."):
            if device:
                nearest = self._nearest_on_device(data, n_probes)
            else:
                nearest = knn_brute(data, self.all_centers, k=n_probes, metric=self.metric)
        with timer(verbose, "PQ Transforming active centers..."):
            self.active_centers = np.ascontiguousarray(
                self.all_centers[np.unique(nearest)], dtype=np.float32)
            self.pq_transformed_centers = self.pq.transform(self.active_centers, device=device)
        with timer(verbose, "Transforming points..."):
            n_active = self.active_centers.shape[0]
            if device:
                self._encode_lists_on_device(data, nearest, n_active)
            else:
                groups, self.ids = group_data_by_indices(data, nearest, n_active)
                for i in range(n_active):
                    self.pq_transformed_points[i] = self.pq.transform(groups[i])
        return self

    def _nearest_on_device(self, data, n_probes):
        """knn_brute(data, all_centers, n_probes, metric) (utils.py:66-86): whole 100-row
        chunks on the GPU, the last partial chunk — a differently shaped GEMM — in numpy."""
        Y = np.asarray(self.all_centers)
        exact_ok = (data.dtype == np.float32 and n_probes <= 9 and n_probes < len(Y)
                    and data.shape[1] <= 384
                    and (self.metric != "angular" or data.shape[1] <= 128))
        if not exact_ok:
            return knn_brute(data, Y, k=n_probes, metric=self.metric)
        y64 = Y.dtype != np.float32
        Y = np.ascontiguousarray(Y, dtype=np.float64 if y64 else np.float32)
        if self.metric == "angular":
            Y = Y / np.linalg.norm(Y, axis=1, keepdims=True)          # utils.py:75
        ynorm2 = np.ascontiguousarray(np.einsum("ij,ij->i", Y, Y))    # utils.py:80
        n = data.shape[0]
        full = n - n % 100
        out = np.zeros((n, n_probes), dtype=np.int64)
        X = np.ascontiguousarray(data[:full])
        _lib.check(_lib.lib().tk_assign_lists(
            _lib.ptr(X, _lib._f32p), full, X.shape[1], int(self.metric == "angular"),
            Y.ctypes.data, int(y64), ynorm2.ctypes.data, Y.shape[0], int(n_probes),
            _lib.ptr(out, _lib._i64p)))
        if full < n:
            out[full:] = knn_brute(data[full:], self.all_centers, k=n_probes, metric=self.metric)
        if n_probes > 2 and full > 0 and not self._assign_order_holds(data, out, n_probes, full):
            # numpy's argpartition leaves the ORDER of the first k unspecified; the device writes them
            # ascending by (value, index), which is what this host's numpy was seen to return — but that
            # is an implementation detail of its argselect (x86-simd-sort on AVX-512; other on AVX2, ARM,
            # older numpy).  Column order decides list order (group_data_by_indices appends column by
            # column): where this host's numpy orders differently, numpy's answer is the reference's.
            import warnings
            warnings.warn("tinyknn_amd: numpy.argpartition on this host does not return the first k ascending "
                          "(k = %d); list assignment falls back to numpy's knn_brute" % n_probes)
            return knn_brute(data, self.all_centers, k=n_probes, metric=self.metric)
        return out

    def _assign_order_holds(self, data, nearest, n_probes, full, chunks=24):
        """Self-check of the k > 2 device assignment against THIS host's numpy: whole 100-row chunks of
        knn_brute (utils.py:81-85 works in chunks of 100), spread over the data."""
        n_chunks = full // 100
        if n_chunks == 0:
            return True
        pick = np.unique(np.linspace(0, n_chunks - 1, min(chunks, n_chunks)).astype(np.int64))
        for c in pick:
            rows = slice(100 * int(c), 100 * int(c) + 100)
            want = knn_brute(data[rows], self.all_centers, k=n_probes, metric=self.metric)
            if not np.array_equal(want, nearest[rows]):
                return False
        return True

    def _encode_lists_on_device(self, data, nearest, n_active):
        """ivf.py:98-102 with ONE pass over the points: a row's code does not depend on the
        list it lands in, so all rows are encoded once (slabs of rows: pad, rotate on the
        host as the reference does, nearest centroids on the GPU) and the per-list arrays
        are gathered from the labels; the rows that pad a list to a multiple of 16 carry the
        code of the zero vector, as pad2 + transform give them (fast_pq.py:165)."""
        from ._transform import transform_data
        from .fast_pq import TransformedData
        pq = self.pq
        dpb = pq.dims_per_block
        n, d = data.shape
        pad = (-d) % (dpad * dpb)
        dq = pq.centers.shape[1]
        M = dq // dpb
        labels = np.empty((n, M), dtype=np.uint8)
        slab = 1 << 20
        for o in range(0, n, slab):
            rows = data[o:o + slab]
            if pad:
                rows = np.concatenate([rows, np.zeros((len(rows), pad), rows.dtype)], axis=1)
            if pq.R is not None:
                rows = rows @ pq.R.T
            labels[o:o + slab] = pq.encode_labels(rows, True)
        zero = pq.encode_labels(np.zeros((16, dq), dtype=np.float64 if pq.R is not None else data.dtype),
                                True)[0]
        # grouping as group_data_by_indices does (utils.py:95-162), without copying the vectors
        assert 0 <= nearest.min() and nearest.max() < n_active       # utils.py:128
        ids = [[] for _ in range(n_active)]
        for j in range(nearest.shape[1]):
            col = nearest[:, j]
            order = np.argsort(col)
            uniq, counts = np.unique(col[order], return_counts=True)
            start = 0
            for g, cnt in zip(uniq, counts):
                ids[g].append(order[start:start + cnt])
                start += cnt
        self.ids = [np.hstack(i) if i else np.empty(0) for i in ids]
        for i in range(n_active):
            sel = self.ids[i].astype(np.int64)
            if len(sel) == 0:
                self.pq_transformed_points[i] = np.empty((0, d))     # fast_pq.py:162-163
                continue
            lab = labels[sel]
            padrows = (-len(sel)) % 16
            if padrows:
                lab = np.concatenate([lab, np.repeat(zero[None], padrows, axis=0)])
            self.pq_transformed_points[i] = TransformedData(len(sel), transform_data(lab))

    # ---- persistence ---------------------------------------------------------
    # The reference pickles (pq, ivf) (examples/bench.py:88-103); that works here too
    # (__getstate__ drops the device handle).  save/load is the flat form the device upload
    # wants: CSR offsets + one array per component, no Python objects.
    def save(self, path):
        """Flat binary (.npz): PQ codebook (+ rotation), coarse centres and their codes, list
        sizes, packed codes and ids of all lists concatenated list-major, rescoring vectors."""
        self._require_host_copy("save")
        L = len(self.active_centers)
        M = self.pq.centers.shape[1] // self.pq.dims_per_block
        tds = [self.pq_transformed_points[i] for i in range(L)]
        sizes = np.array([0 if isinstance(t, np.ndarray) else t.size for t in tds], dtype=np.int64)
        codes = [t.packed for t in tds if not isinstance(t, np.ndarray)]
        ids = [np.asarray(self.ids[i], dtype=np.int64)[:sizes[i]] for i in range(L)]
        extra = {} if self.pq.R is None else {"R": self.pq.R}
        if getattr(self, "all_centers", None) is not None:
            extra["all_centers"] = self.all_centers
        path = self._npz_path(path)
        np.savez(path, format_version=1, metric=self.metric, n_clusters=self.n_clusters,
                 use_kmeans=int(self.pq.use_kmeans), rotate_dim=-1 if self.pq.rotate_dim is None else int(self.pq.rotate_dim),
                 dims_per_block=self.pq.dims_per_block, pq_centers=self.pq.centers,
                 pq_centers_f_order=int(not self.pq.centers.flags.c_contiguous),
                 sqrt_n_blocks=self.pq.sqrt_n_blocks, active_centers=self.active_centers,
                 center_size=self.pq_transformed_centers.size,
                 center_codes=self.pq_transformed_centers.packed, list_sizes=sizes,
                 list_codes=(np.concatenate(codes) if codes else np.zeros((0, M), np.uint64)),
                 ids=(np.concatenate(ids) if ids else np.zeros(0, np.int64)), data=self.data, **extra)

    @staticmethod
    def _npz_path(path):
        """np.savez appends '.npz' to a path without that suffix; save and load agree on it."""
        path = str(path)
        return path if path.endswith(".npz") else path + ".npz"

    @classmethod
    def load(cls, path, data=None):
        """Inverse of save.  `data`: the rescoring vectors if the file was written without them
        being wanted twice (pass the array to avoid keeping two copies)."""
        from .fast_pq import TransformedData
        z = np.load(cls._npz_path(path), allow_pickle=False)
        assert int(z["format_version"]) == 1
        pq = FastPQ(int(z["dims_per_block"]),
                    use_kmeans=bool(int(z["use_kmeans"])) if "use_kmeans" in z else True,
                    rotate_dim=(None if int(z["rotate_dim"]) < 0 else int(z["rotate_dim"])) if "rotate_dim" in z else 64)
        ivf = cls(str(z["metric"]), int(z["n_clusters"]), pq)
        c = z["pq_centers"]
        ivf.pq.centers = np.asfortranarray(c) if int(z["pq_centers_f_order"]) else c
        ivf.pq.sqrt_n_blocks = float(z["sqrt_n_blocks"])
        ivf.pq.R = z["R"] if "R" in z else None
        if "all_centers" in z:
            ivf.all_centers = z["all_centers"]
        ivf.active_centers = z["active_centers"]
        ivf.pq_transformed_centers = TransformedData(int(z["center_size"]), z["center_codes"])
        sizes = z["list_sizes"]
        coff = np.concatenate([[0], np.cumsum((sizes + 15) // 16)])
        ioff = np.concatenate([[0], np.cumsum(sizes)])
        codes, ids = z["list_codes"], z["ids"]
        d = ivf.active_centers.shape[1]
        ivf.pq_transformed_points = [
            TransformedData(int(sizes[i]), codes[coff[i]:coff[i + 1]]) if sizes[i] else np.empty((0, d))
            for i in range(len(sizes))]
        ivf.ids = [ids[ioff[i]:ioff[i + 1]] for i in range(len(sizes))]
        ivf.data = z["data"] if data is None else data
        return ivf

    def build_resident(self, N, d, seed, centres=None, sigma=1.0, verbose=False, n_probes=1):
        """IVF.build(X, n_probes=1 or 2) (ivf.py:53-104) for N synthetic float32 vectors that are
        generated IN HBM (seeded, devbuild.hip) and never visit the host — the way the
        100M x 128 configuration is assembled.  Needs all_centers and a
        fitted pq (fit()).  Everything runs on the device: nearest centre per row, PQ codes,
        grouping by list (rows of a list in ascending row order, where numpy's unstable
        argsort leaves their order unspecified), packing.  The rotation of a rotated PQ is
        the device front end's float64 FMA chain, not numpy's DGEMM, so a code can differ
        from IVF.build's where that flips a nearest centroid.  Afterwards: active_centers,
        pq_transformed_centers and list_sizes on the host; data / ids / codes stay in HBM
        (device_index().export_lists() / read_rows() fetch them for a checker)."""
        from .fast_pq import TransformedData
        assert self.pq.centers is not None and getattr(self, "all_centers", None) is not None
        with timer(verbose, "Generating vectors in HBM..."):
            dev = DeviceIndex.resident(self, N, d)
            dev.synth_data(seed, centres, sigma)
        if n_probes > 2:
            # no host fallback here: refuse k > 2 unless this host's numpy orders argpartition's first k
            # the way the device does (see _nearest_on_device), checked on generated rows
            probe = IVF(self.metric, self.n_clusters, FastPQ(self.pq.dims_per_block))
            probe.all_centers = self.all_centers
            rows = dev.read_rows(np.arange(min(N - N % 100, 2400), dtype=np.int64))
            if self.metric == "angular":
                rows = rows / np.linalg.norm(rows, axis=1, keepdims=True)
            got = probe._nearest_on_device(np.ascontiguousarray(rows, dtype=np.float32), n_probes)
            if len(rows) and not np.array_equal(got, knn_brute(rows, self.all_centers, k=n_probes, metric=self.metric)):
                raise RuntimeError("build_resident(n_probes=%d): numpy.argpartition on this host does not return the "
                                   "first k ascending, which the device assignment assumes for k > 2; build with "
                                   "n_probes <= 2 or on the host (IVF.build)" % n_probes)
        with timer(verbose, "Building lists on the device..."):
            L = dev.build_dev(self.all_centers, n_probes)
        self.active_centers, cc = dev.export_centers()
        self.pq_transformed_centers = TransformedData(L, cc)
        self.list_sizes = dev.list_sizes
        self.data = ResidentData(dev)
        self.ids = self.pq_transformed_points = None
        self._dev = dev
        return self

    # ---- queries (GPU) -----------------------------------------------------
    def device_index(self):
        if self._dev is None:
            self._dev = DeviceIndex(self)
        return self._dev

    def _unsharded_device_index(self):
        dev = self.device_index()
        if dev.world != 1 or getattr(dev, "_sharded_as", None) is not None:
            raise RuntimeError("this IVF's device index has been list-sharded in place (ListShardedIndex on an "
                               f"index built in HBM: rank {dev.rank} of {dev.world}); query it through the "
                               "ListShardedIndex")
        return dev

    def _prepare(self, qs):
        """Host side of ivf.py:125-128: float32, metric normalisation (numpy, in
        place for a contiguous float32 input as in the reference), padding and the
        optional float64 rotation of the table-build query."""
        pq = self.pq
        dq = pq.centers.shape[1]
        d = qs.shape[1]
        pad = (-d) % (dpad * pq.dims_per_block)
        R = pq.R
        if R is not None and (min(R.shape) < 2 or R.dtype != np.float64):
            # 1-row / 1-column products take other numpy code paths (dot, no BLAS)
            qn, qp = _front.numpy_prepare(qs, self.metric == "angular", R, pad)
        else:
            # the same two BLAS calls numpy makes per row, from a thread pool (_front.py)
            qn, qp = _front.prepare(qs, self.metric == "angular", R, pad)
        assert qp.shape[1] == dq
        return qn, qp

    def query(self, q, k, n_probes=1, pass_1=None):
        """Top-k ids for one query.  reference: ivf.py:106-163"""
        q = np.ascontiguousarray(q, dtype=np.float32)
        assert self.data.shape[1] == q.shape[0]
        qn, qp = self._prepare(q[None, :])
        out = self._unsharded_device_index().query_batch(qn, qp, k, n_probes, pass_1)[0]
        return out[out != -1] if out[-1] == -1 else out

    def query_batch(self, qs, k, n_probes=1, pass_1=None, fast=False):
        """(nq, d) queries -> (nq, k) int64 ids, rows padded with -1 when the
        reference would return fewer than k ids.  (The reference's README shows a
        2-d `ivf.query(queries, ...)` that its code does not support; this is that
        call.)  fast=True: normalisation, padding and rotation run on the device instead
        of numpy's per-query BLAS calls (35 ms per 10 000 queries on the host) — within
        1 ulp of them, so a rare id can differ from the reference's; the default is exact."""
        self._unsharded_device_index()
        if fast:
            return self.device_index().query_batch_raw(qs, k, n_probes, pass_1)
        R = self.pq.R
        if _front.bind() and (R is None or (min(R.shape) >= 2 and R.dtype == np.float64)):
            # exact: chunks stream through preparation (numpy's BLAS on a thread pool),
            # pinned H2D, the kernels and D2H, overlapped
            return self.device_index().query_raw(qs, k, n_probes, pass_1)
        qs = np.array(qs, dtype=np.float32, order="C", copy=True)
        qn, qp = self._prepare(qs)
        return self.device_index().query_batch(qn, qp, k, n_probes, pass_1)
