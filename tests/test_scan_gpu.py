"""The library's own exclusive prefix sum (devbuild.hip: one launch, decoupled look-back, no hipcub) against numpy.

The per-batch sums of a list-sharded rank (segment positions, pair offsets: shard.hip) go through it on the device;
tk_scan_exclusive_host runs the same kernel on host arrays so that every length around the tile size (2 048 elements
per workgroup) and well beyond one look-back window (64 tiles) is checked, both element widths, in place too."""
import ctypes as C

import numpy as np
import pytest

from tinyknn_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("is64", [0, 1])
def test_exclusive_scan_matches_numpy(is64):
    L = _lib.lib()
    rng = np.random.RandomState(5 + is64)
    dt = np.int64 if is64 else np.int32
    for n in (0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 5000, 64 * 2048 - 1, 64 * 2048 + 7, 1_000_003, 3_500_000):
        if is64:
            a = rng.randint(0, 1 << 40, size=n).astype(dt)          # sums far beyond 32 bits
        else:
            a = rng.randint(0, 500, size=n).astype(dt)
        want = np.zeros(n, dtype=dt)
        if n > 1:
            want[1:] = np.cumsum(a[:-1], dtype=dt)
        out = np.full(n, -7, dtype=dt)
        _lib.check(L.tk_scan_exclusive_host(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n, is64))
        assert (out == want).all(), n
        b = a.copy()                                                  # in place
        _lib.check(L.tk_scan_exclusive_host(b.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), n, is64))
        assert (b == want).all(), n


def test_exclusive_scan_negative_values():
    L = _lib.lib()
    a = np.random.RandomState(1).randint(-1000, 1000, size=300_001).astype(np.int64)
    want = np.concatenate([[0], np.cumsum(a[:-1])])
    out = np.empty_like(a)
    _lib.check(L.tk_scan_exclusive_host(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(a), 1))
    assert (out == want).all()
