"""Exact host front end of IVF.query (reference: ivf.py:125-128, fast_pq.py:200-204).

The reference prepares a query on the host with two BLAS calls — np.linalg.norm
(cblas_sdot) and, for a rotated PQ, `q @ R.T` (cblas_dgemv) — whose summation orders
belong to the BLAS build numpy links.  front.hip calls THE SAME two functions of THE
SAME shared object from a thread pool; this module finds that shared object, binds it
(tk_host_blas_bind) and proves the binding against numpy itself before it is used.
If the proof fails (an unknown BLAS), `prepare` stays the reference's own numpy loop:
exact either way, the binding only removes the Python loop.
"""
import os

import numpy as np

from . import _lib

_state = {"tried": False, "ok": False, "path": None, "why": None}


def _numpy_blas_candidates():
    """Shared objects of this process that can hold numpy's cblas symbols, numpy's own
    bundled library first."""
    np.dot(np.ones(4, np.float32), np.ones(4, np.float32))     # make sure it is mapped
    seen, out = set(), []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if not path.startswith("/") or path in seen:
                    continue
                seen.add(path)
                base = os.path.basename(path).lower()
                if any(t in base for t in ("openblas", "blas", "mkl_rt", "blis", "accelerate")):
                    out.append(path)
    except OSError:
        pass
    npdir = os.path.dirname(os.path.dirname(np.__file__))
    out.sort(key=lambda p: (0 if os.path.join(npdir, "numpy.libs") in p else
                            1 if "numpy" in p else 2))
    return out


def numpy_prepare(qs, angular, R, pad):
    """The reference's arithmetic, row by row in numpy (ivf.py:125-127, fast_pq.py:202-204):
    the checker of the bound path and the fallback when no BLAS could be bound."""
    if angular:
        for row in qs:
            row /= np.linalg.norm(row)
    qp = qs if pad == 0 else np.concatenate([qs, np.zeros((len(qs), pad), qs.dtype)], axis=1)
    if R is not None:
        qp = np.stack([row @ R.T for row in qp]) if len(qp) else np.zeros((0, R.shape[0]))
    return qs, qp


def _c_prepare(qs, angular, R, pad):
    L = _lib.lib()
    nq, d = qs.shape
    qp = None
    if R is not None:
        Rc = np.ascontiguousarray(R, dtype=np.float64)
        qp = np.empty((nq, Rc.shape[0]), dtype=np.float64)
        _lib.check(L.tk_prepare_queries_host(qs.ctypes.data, nq, d, int(angular), qs.ctypes.data,
                                             Rc.ctypes.data, Rc.shape[0], Rc.shape[1], qp.ctypes.data))
        return qs, qp
    _lib.check(L.tk_prepare_queries_host(qs.ctypes.data, nq, d, int(angular), qs.ctypes.data,
                                         None, d + pad, d + pad, None))
    qp = qs if pad == 0 else np.concatenate([qs, np.zeros((nq, pad), qs.dtype)], axis=1)
    return qs, qp


def _self_check():
    """The bound calls against numpy on rows of awkward lengths and alignments."""
    rng = np.random.RandomState(1234)
    for d, pad, rd in ((100, 4, 0), (37, 3, 0), (128, 0, 64), (20, 4, 16), (1, 7, 4)):
        base = rng.randn(67 * d + 3).astype(np.float32)
        for off in (0, 1, 3):                      # rows at 4-byte granular alignments
            raw = base[off:off + 67 * d].reshape(67, d)
            R = rng.randn(rd, d + pad) if rd else None
            a, ap = numpy_prepare(raw.copy(), True, R, pad)
            buf = np.empty(67 * d + 3, np.float32)
            view = buf[off:off + 67 * d].reshape(67, d)
            view[:] = raw
            b, bp = _c_prepare(view, True, R, pad)
            if not (np.array_equal(a.view(np.uint32), b.view(np.uint32)) and
                    ap.dtype == bp.dtype and np.array_equal(ap.view(np.uint8), bp.view(np.uint8))):
                return False
    return True


def bind():
    """Bind numpy's BLAS once; returns True when the fast exact path is usable."""
    if _state["tried"]:
        return _state["ok"]
    _state["tried"] = True
    forced = os.environ.get("TINYKNN_HOST_BLAS")
    L = _lib.lib()
    for path in ([forced] if forced else _numpy_blas_candidates()):
        if L.tk_host_blas_bind(path.encode()) != 0:
            continue
        try:
            if _self_check():
                _state.update(ok=True, path=path)
                return True
        except Exception as e:        # noqa: BLE001 - any failure means "not proven"
            _state["why"] = repr(e)
    _state["why"] = _state["why"] or "no shared object of this process reproduces numpy's norm / matmul"
    return False


def info():
    bind()
    return dict(_state, threads=_lib.lib().tk_host_threads(0) if _state["ok"] else 0)


def prepare(qs, angular, R, pad):
    """qs: C-contiguous float32 (nq, d), normalised IN PLACE when angular (as the reference
    does to its query); returns (qn, q_pq) with q_pq = pad1(qn) [@ R.T]."""
    assert qs.dtype == np.float32 and qs.flags.c_contiguous and qs.ndim == 2
    if len(qs) and bind():
        return _c_prepare(qs, angular, R, pad)
    return numpy_prepare(qs, angular, R, pad)
