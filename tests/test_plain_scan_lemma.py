"""The lemma behind plain_scan.hip, on the CPU with the oracle's kernels (no GPU):

For a signed table T, chains c = the blocks one saturating accumulator of the reference adds up
(AVX order: (m >> 1) & 1 == c, _fast_pq_256.pyx:135-156; SSE order: all blocks), N_c = the
chain's largest possible negative mass and C = 127 - N_0 - N_1: if N_0, N_1 <= 128 then for every
code with plain sum S:  S < C  =>  saturated value == max(S, -128);  S >= C  =>  value >= C.
And a replay whose bound is <= C from the first block on is the same replay on either value set.
"""
import numpy as np
import pytest


def limits(T, avx):
    neg = np.maximum(0, -T.min(axis=1))
    chain = ((np.arange(len(T)) >> 1) & 1) if avx else np.zeros(len(T), int)
    n0, n1 = int(neg[chain == 0].sum()), int(neg[chain == 1].sum())
    return n0, n1, 127 - n0 - n1


@pytest.mark.parametrize("order", ["avx", "sse"])
@pytest.mark.parametrize("M", [4, 12, 32, 52])
def test_clamped_plain_sum_is_the_saturated_value_below_the_limit(oracle, order, M):
    rng = np.random.default_rng(M * 7 + (order == "avx"))
    avx = order == "avx"
    checked = conform = 0
    for trial in range(60):
        # table shapes from the realistic (-4..23) to the adversarial (full range, skewed)
        kind = trial % 5
        if kind == 4:
            T = rng.integers(-2, 24, size=(M, 16))
        elif kind == 0:
            T = rng.integers(-4, 24, size=(M, 16))
        elif kind == 1:
            T = rng.integers(-128, 128, size=(M, 16))
        elif kind == 2:
            T = rng.integers(-9, 60, size=(M, 16))
        else:
            T = rng.integers(-3, 4, size=(M, 16)) * rng.integers(1, 40, size=(M, 1))
        T = T.astype(np.int8).astype(np.int64)
        n0, n1, C = limits(T, avx)
        n = 16 * 40
        codes = rng.integers(0, 16, size=(n, M)).astype(np.uint8)
        packed = oracle.transform_data(codes)
        tt = oracle.transform_tables(T.astype(np.int8).view(np.uint8))
        out = np.zeros(2 * len(packed), np.uint64)
        oracle.estimate_pq(packed, tt, out, True, oracle.ORDER_AVX if avx else oracle.ORDER_SSE)
        v = out.view(np.int8)[:n].astype(np.int64)
        S = T[np.arange(M)[None, :], codes.astype(np.int64)].sum(axis=1)
        checked += 1
        if n0 > 128 or n1 > 128:
            continue                      # the kernel leaves such a query to the exact path
        conform += 1
        low = S < C
        np.testing.assert_array_equal(v[low], np.maximum(S[low], -128))
        assert (v[~low] >= C).all()
        o = np.clip(S, -128, 127)
        assert (o[~low] >= C).all()
    assert conform >= checked // 6        # (one chain of 52 blocks rarely stays above -128: SSE order, M = 52)


def test_replay_over_clamped_plain_sums_is_the_same_replay(oracle):
    """Heap arrays (layout included) after a first list on exact values and later lists on
    clamp(plain sum) equal the all-exact replay whenever the bound after the first list <= C."""
    rng = np.random.default_rng(3)
    M, R = 52, 30
    same = tried = 0
    for trial in range(40):
        T = rng.integers(-4, 24, size=(M, 16)).astype(np.int64)
        n0, n1, C = limits(T, True)
        assert n0 <= 128 and n1 <= 128
        tt = oracle.transform_tables(T.astype(np.int8).view(np.uint8))
        # rows near the query: most blocks take the table's smallest entry (sums around -100 .. 60)
        best = T.argmin(axis=1)
        lists = []
        for _ in range(4):
            n_ = 16 * int(rng.integers(3, 12))
            li = rng.integers(0, 16, size=(n_, M)).astype(np.uint8)
            near = rng.random((n_, M)) < rng.uniform(0.75, 1.0, size=(n_, 1))
            li[near] = np.broadcast_to(best, (n_, M))[near]
            lists.append(li)
        hi, hv = np.full(R, -1, np.int64), np.full(R, 127, np.int32)
        gi, gv = hi.copy(), hv.copy()
        base = 0
        b1 = None
        for s, codes in enumerate(lists):
            packed = oracle.transform_data(codes)
            n = len(codes) - int(rng.integers(0, 5))
            labels = np.arange(base, base + len(codes), dtype=np.int64)
            base += len(codes)
            oracle.query_pq(packed, n, tt, hi, hv, True, labels=labels)
            if s == 0:
                oracle.query_pq(packed, n, tt, gi, gv, True, labels=labels)
                b1 = int(np.int8(hv[0] & 0xff))
            else:
                S = T[np.arange(M)[None, :], codes.astype(np.int64)].sum(axis=1)
                o = np.clip(S, -128, 127)
                for b0 in range(0, len(codes), 16):
                    bound = int(np.int8(gv[0] & 0xff))
                    for r_ in np.nonzero(o[b0:b0 + 16] < bound)[0]:
                        if b0 + r_ < n:
                            oracle.insert(gi, gv, int(labels[b0 + r_]), int(o[b0 + r_]))
        if b1 <= C:
            tried += 1
            same += int(np.array_equal(hi, gi) and np.array_equal(hv, gv))
    assert tried >= 20 and same == tried
