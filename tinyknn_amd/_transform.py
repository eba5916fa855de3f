"""Quick-ADC memory formats (host side, numpy).

Same formats as the reference's tinyknn/_transform.py — they are the contract
between FastPQ.transform and the scan kernels — written here as plain byte
shuffles.  Viewed as bytes, chunk c / block pair p / row r of the packed array is
byte ``c*8*M + 16*p + r`` and holds ``code[16c+r, 2p] | code[16c+r, 2p+1] << 4``.
"""
import numpy as np


def transform_data(data0):
    """(n, M) 4-bit codes -> uint64 (n/16, M).  reference: _transform.py:4-77"""
    data0 = np.asarray(data0)
    n, d = data0.shape
    assert n % 16 == 0, "Number of rows must be divisible by 16"
    assert np.all(data0 < 16) and np.all(0 <= data0), "Input must be 4 bit values"
    assert d % 2 == 0, "Number of blocks must be even"
    nib = data0.astype(np.uint8).reshape(n // 16, 16, d // 2, 2)
    pair = nib[..., 0] | (nib[..., 1] << 4)          # (chunk, row, pair)
    by = np.ascontiguousarray(pair.transpose(0, 2, 1))  # (chunk, pair, row)
    return by.reshape(n // 16, 8 * d).view(np.uint64)


def unpack(transformed_data):
    """Inverse of transform_data.  reference: _transform.py:80-111"""
    td = np.ascontiguousarray(transformed_data, dtype=np.uint64)
    chunks, d = td.shape
    by = td.view(np.uint8).reshape(chunks, d // 2, 16).transpose(0, 2, 1)  # (chunk,row,pair)
    out = np.empty((chunks, 16, d // 2, 2), dtype=np.uint8)
    out[..., 0] = by & 15
    out[..., 1] = by >> 4
    return out.reshape(chunks * 16, d)


def transform_tables(tables0):
    """(M, 16) uint8 -> uint64 (2M,).  reference: _transform.py:114-138"""
    d, b = tables0.shape
    assert b == 16
    assert tables0.dtype == np.uint8
    return np.ascontiguousarray(tables0).reshape(-1).view(np.uint64)
