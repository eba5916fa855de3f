"""The argument behind the filtered exchange (shard.hip), checked on the CPU with
the oracle's heap primitives and the golden fixtures:
  (1) replaying precomputed distance rows block by block with the reference's rule (bound at
      block start, insert without re-check, refresh after a block with a hit) reproduces the
      oracle's heap arrays, layout included;
  (2) replacing every block of the LATER probed lists whose minimum is not below B1 — the bound
      after the first probed list — by the largest value leaves those arrays unchanged;
  (3) B1 follows from the multiset of heap values alone (what shard_first_bound_kernel keeps)."""
import numpy as np
import pytest

from conftest import golden


def _replay(O, rows, R, stop_after=None):
    idx = np.zeros(R, dtype=np.int64)
    val = np.zeros(R, dtype=np.int32)
    O.init_heap(idx, val, True)
    for j, (d, n, lab) in enumerate(rows):
        bound = np.int8(np.uint8(int(val[0]) & 0xff).view(np.int8))
        for c in range(len(d) // 16):
            blk = d[16 * c:16 * c + 16]
            hit = np.flatnonzero(blk < bound)
            if not len(hit):
                continue
            for r in hit:
                if 16 * c + r < n:
                    O.insert(idx, val, int(lab[16 * c + r]), int(blk[r]))
            bound = np.int8(np.uint8(int(val[0]) & 0xff).view(np.int8))
        if stop_after is not None and j == stop_after:
            break
    return idx, val


def _bound_by_multiset(d, n, R):
    hist = np.zeros(256, dtype=np.int64)
    hist[255] = R
    mx = 255
    for c in range(len(d) // 16):
        bnd = mx
        for r in range(16):
            kv = int(d[16 * c + r]) + 128
            if kv < bnd and 16 * c + r < n:
                hist[mx] -= 1
                hist[kv] += 1
                if kv > mx:
                    mx = kv
                else:
                    while hist[mx] == 0:
                        mx -= 1
    return mx


@pytest.mark.parametrize("tag", ["an100", "an100b2", "eu128"])
@pytest.mark.parametrize("n_probes", [5, 10])
def test_filtered_rows_replay_identically(oracle, tag, n_probes):
    from test_oracle_golden import load_oracle_index
    O = oracle
    g = golden(f"g6_ivf_{tag}.npz")
    ox = load_oracle_index(O, g)
    off = ox.list_chunk_off
    k = 10
    dropped = total = 0
    for qn in g["qn"][:16]:
        ids, dbg = ox.query(qn, k, n_probes, debug=True)
        probes = dbg["probes"].copy()
        probes[probes < 0] += ox.n_lists
        table = O.transform_tables(dbg["table"])
        R = len(dbg["heap_val"])
        rows = []
        for l in probes:
            codes = np.ascontiguousarray(ox.codes[off[l]:off[l + 1]])
            out = np.zeros(2 * len(codes), dtype=np.uint64)
            if len(codes):
                O.estimate_pq(codes, table, out, True)
            rows.append((out.view(np.int8).copy(), int(ox.list_n[l]),
                         ox.ids[ox.ids_off[l]:ox.ids_off[l + 1]]))
        idx, val = _replay(O, rows, R)
        np.testing.assert_array_equal(idx, dbg["heap_idx"])                 # (1)
        np.testing.assert_array_equal(val, dbg["heap_val"])
        _, v1 = _replay(O, rows, R, stop_after=0)
        b1 = int(np.uint8(int(v1[0]) & 0xff).view(np.int8))
        assert _bound_by_multiset(rows[0][0], rows[0][1], R) == b1 + 128    # (3)
        filt = [rows[0]]
        for d, n, lab in rows[1:]:
            d = d.copy()
            m = d.reshape(-1, 16).min(axis=1)
            d.reshape(-1, 16)[m >= b1] = 127
            dropped += int((m >= b1).sum())
            total += len(m)
            filt.append((d, n, lab))
        idx2, val2 = _replay(O, filt, R)
        np.testing.assert_array_equal(idx2, idx)                            # (2)
        np.testing.assert_array_equal(val2, val)
    # (the fixtures' lists are short: where the first list holds fewer than R rows, B1 stays 127
    # and everything travels — correct, only not smaller)
    assert total > dropped
    print(f"{tag} n_probes={n_probes}: {dropped}/{total} blocks of the later lists need not travel")


@pytest.mark.parametrize("seed", range(40))
def test_filtered_rows_random_streams(oracle, seed):
    """The same two statements on random distance streams with the shapes the fixtures lack: lists
    much longer than the heap, heavy ties (narrow value range), labels that repeat across lists
    (the duplicate test fires), values at both int8 rails, lists shorter than the heap, R = 1."""
    O = oracle
    rng = np.random.RandomState(seed)
    R = int(rng.choice([1, 3, 30, 111, 300]))
    n_lists = int(rng.choice([2, 5, 10]))
    spread = float(rng.choice([2.0, 12.0, 60.0]))
    pool = int(rng.choice([500, 5000, 10 ** 6]))          # small pool: labels repeat across lists
    rows = []
    for _ in range(n_lists):
        n = int(rng.choice([1, 15, 16, 17, 200, 1500, 4000]))
        d = np.clip(np.round(rng.randn(-(-n // 16) * 16) * spread + rng.randint(-40, 40)), -128, 127).astype(np.int8)
        lab = rng.choice(pool, size=n, replace=pool < n) if pool >= n else rng.randint(0, pool, size=n)
        if pool >= n:
            lab = rng.choice(pool, size=n, replace=False)      # distinct inside one list, as in an IVF list
        rows.append((d, n, lab.astype(np.int64)))
    idx, val = _replay(O, rows, R)
    _, v1 = _replay(O, rows, R, stop_after=0)
    b1 = int(np.uint8(int(v1[0]) & 0xff).view(np.int8))
    if len(set(rows[0][2].tolist())) == len(rows[0][2]):
        assert _bound_by_multiset(rows[0][0], rows[0][1], R) == b1 + 128
    filt = [rows[0]]
    for d, n, lab in rows[1:]:
        d = d.copy()
        m = d.reshape(-1, 16).min(axis=1)
        d.reshape(-1, 16)[m >= b1] = 127
        filt.append((d, n, lab))
    idx2, val2 = _replay(O, filt, R)
    np.testing.assert_array_equal(idx2, idx)
    np.testing.assert_array_equal(val2, val)
