"""The lane-parallel form of `insert` (_fast_pq.pyx:274-307) that flat_top_one_kernel runs with the heap
in registers (heap.hip: reg_heap_insert), restated in numpy and checked against the oracle's loop.

Claim: the sift of `insert` follows the heap's MAX-CHILD path (left child on ties), which does not depend
on the inserted value; values never rise along it, so the nodes the loop passes are the root and the path
nodes above v: each takes its path child's entry, the last one takes (label, v).  Every node decides from
its sibling, its ancestors' flags and its path child alone — what one lane per node can do at once."""
import numpy as np
import pytest


def insert_by_lanes(idx, val, label, v):
    """All R 'lanes' at once, as heap.hip does it (no loop over levels)."""
    R = len(val)
    if (idx == label).any():                       # the duplicate test, :284-287
        return
    t = np.arange(R)
    nxt = np.concatenate([val[1:], [0]])           # vals[t + 1]
    prv = np.concatenate([[0], val[:-1]])          # vals[t - 1]
    odd = t % 2 == 1
    larger = np.where(odd, (t + 1 >= R) | (val >= nxt), (t == 0) | (val > prv))
    onp = np.ones(R, bool)                         # every ancestor down to the root's child is a larger child
    for node in range(R):
        a = node
        while a > 0:
            onp[node] &= larger[a]
            a = (a - 1) // 2
    l = 2 * t + 1
    has_child = l < R
    c = np.where(has_child, l + 1 - larger[np.minimum(l, R - 1)].astype(int), 0)
    c = np.minimum(c, R - 1)
    cv, ci = val[c], idx[c]
    passed = onp & ((t == 0) | (val > v))
    up = passed & has_child & (cv > v)
    new_val = np.where(up, cv, np.where(passed, v, val))
    new_idx = np.where(up, ci, np.where(passed, label, idx))
    val[:] = new_val
    idx[:] = new_idx


@pytest.mark.parametrize("R", [1, 2, 3, 7, 12, 30, 31, 63, 64])
@pytest.mark.parametrize("spread", [3, 40, 250])
def test_lane_parallel_insert_equals_the_loop(oracle, R, spread):
    rng = np.random.RandomState(R * 1000 + spread)
    for signd in (True, False):
        wi, wv = np.zeros(R, np.int64), np.zeros(R, np.int32)
        oracle.init_heap(wi, wv, signd)
        gi, gv = wi.copy(), wv.copy()
        lo = -128 if signd else 0
        for step in range(600):
            # values concentrated near the current root: ties with the root, its children and each other;
            # now and then a value ABOVE the root (a stale bound lets such rows through, :111-123) and a
            # label that is already in the heap
            base = int(wv[0])
            v = int(np.clip(base - rng.randint(0, spread) + (rng.randint(0, 6) if step % 11 == 0 else 0), lo, lo + 255))
            label = int(wi[rng.randint(R)]) if step % 17 == 5 and wi.max() >= 0 else step
            oracle.insert(wi, wv, label, v)
            insert_by_lanes(gi, gv, label, v)
            assert (gi == wi).all() and (gv == wv).all(), (R, signd, step)
