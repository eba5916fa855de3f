"""The lemma behind the TWIN form of the lane replay (heap.hip), on the CPU (no GPU).

`insert` (_fast_pq.pyx:274-307) begins with a scan of all R labels: `if i == indices[j]: return`.
With IVF.build(n_probes = b >= 2) (ivf.py:53) every row sits in b lists and its copies carry the SAME
value (same code, same table).  The replay (`query_pq_*`, _fast_pq_256.pyx:73-123) compares a block
against the bound captured at its start and inserts every passing row.  Let

    f = the smallest value of any root evicted so far (+inf before the first insert).

Then (proof in DESIGN.md §3.3):
  (a) every entry of the heap is <= f, except possibly the root;
  (b) a row that passes its block's bound and has an EARLIER copy among the rows replayed so far
      finds a copy of its label in the heap
        - always                                     if v <  f,
        - iff the root is one of the earlier copies   if v >  f,
        - iff some entry is one of the earlier copies if v == f   (the only case that needs a scan);
  (c) a passing row without an earlier copy never finds its label.
So the duplicate test needs no set of labels: a position-keyed heap, f, and the positions of a row's
earlier copies decide it.  The test replays streams with the reference's loop (a Python restatement
checked against the oracle's `insert`) and checks the rule at every passing row.
"""
import numpy as np
import pytest


def ref_insert(idx, val, i, v):
    """_fast_pq.pyx:274-307; True if (i, v) went in"""
    R = len(idx)
    for j in range(R):
        if idx[j] == i:
            return False
    j = 0
    while True:
        nxt, nv = j, v
        l, r = 2 * j + 1, 2 * j + 2
        if l < R and val[l] > nv:
            nxt, nv = l, val[l]
        if r < R and val[r] > nv:
            nxt, nv = r, val[r]
        if nxt == j:
            break
        val[j], idx[j] = val[nxt], idx[nxt]
        j = nxt
    val[j], idx[j] = v, i
    return True


def replay_with_rule(vals, labels, R, top=127, stats=None):
    """vals / labels: the stream in replay order (a multiple of 16 rows; label < 0 = padding row).
    Runs the reference's loop with labels in the heap and, next to it, a heap of POSITIONS decided
    by the rule; asserts they agree entry by entry after every block."""
    n = len(vals)
    hi, hv = [-1] * R, [top] * R            # the reference's heap (labels)
    pi, pv = [-1] * R, [top] * R            # the rule's heap (positions)
    f = top + 1
    first_pos = {}                          # label -> positions of its copies seen so far (passing or not)
    for b0 in range(0, n, 16):
        bound = hv[0]
        assert bound == pv[0]
        for r in range(16):
            p = b0 + r
            L, v = int(labels[p]), int(vals[p])
            if L < 0:
                continue
            earlier = first_pos.get(L, ())
            if v < bound:
                # ---- the rule
                if not earlier:
                    dup = False
                    kind = "first"
                elif v < f:
                    dup = True
                    kind = "below"
                elif v > f:
                    dup = pv[0] == v and pi[0] in earlier
                    kind = "above"
                else:
                    dup = any(pv[j] == v and pi[j] in earlier for j in range(R))
                    kind = "scan"
                if stats is not None:
                    stats[kind] = stats.get(kind, 0) + 1
                # ---- the reference
                went_in = ref_insert(hi, hv, L, v)
                assert went_in == (not dup), (kind, p, L, v, f, bound)
                if not dup:
                    f = min(f, pv[0])
                    ref_insert(pi, pv, p, v)        # positions are distinct: its scan never fires
                # (a): every entry but the root is <= f
                assert all(x <= f for x in pv[1:])
            first_pos[L] = earlier + (p,)
        assert hv == pv
        assert [(-1 if q < 0 else int(labels[q])) for q in pi] == hi
    return hi, hv


def make_stream(rng, n_lists, rows, b, spread, R):
    """labels in `b` of n_lists + 3 lists (three lists are never replayed: copies outside the probed
    set), one value per label, every list padded to whole blocks"""
    n_labels = n_lists * rows // b
    value = np.clip(np.rint(rng.normal(40, spread, n_labels)), -128, 126).astype(np.int64)
    member = [[] for _ in range(n_lists + 3)]
    for L in range(n_labels):
        for c in rng.choice(n_lists + 3, size=b, replace=False):
            member[c].append(L)
    vals, labels = [], []
    for c in range(n_lists):
        m = np.array(member[c], np.int64)
        rng.shuffle(m)
        pad = (-len(m)) % 16
        labels.append(np.concatenate([m, -np.ones(pad, np.int64)]))
        vals.append(np.concatenate([value[m], np.zeros(pad, np.int64)]))
    return np.concatenate(vals), np.concatenate(labels)


def test_ref_insert_is_the_oracles(oracle):
    rng = np.random.default_rng(5)
    for R in (1, 2, 7, 30, 111):
        hi, hv = np.full(R, -1, np.int64), np.full(R, 127, np.int32)
        li, lv = [-1] * R, [127] * R
        for _ in range(400):
            i, v = int(rng.integers(0, 60)), int(rng.integers(-20, 40))
            oracle.insert(hi, hv, i, v)
            ref_insert(li, lv, i, v)
        assert li == hi.tolist() and lv == hv.tolist()


@pytest.mark.parametrize("b", [2, 3])
@pytest.mark.parametrize("spread", [2.0, 6.0, 25.0])       # many ties ... few ties
def test_rule_decides_every_duplicate_test(b, spread):
    rng = np.random.default_rng(100 * b + int(spread))
    stats = {}
    for trial in range(6):
        R = int(rng.choice([1, 3, 12, 31, 111]))
        vals, labels = make_stream(rng, n_lists=5, rows=int(rng.integers(40, 400)), b=b, spread=spread, R=R)
        replay_with_rule(vals, labels, R, stats=stats)
    assert stats.get("first", 0) > 0 and stats.get("below", 0) > 0


def test_rule_on_streams_made_to_reach_its_rare_branches():
    """Falling values, copies a block or two behind their originals, small heaps: several rows of one
    block pass, the floor f moves inside a block, roots rise — the branches real lists hardly reach."""
    rng = np.random.default_rng(77)
    stats = {}
    for trial in range(300):
        R = int(rng.choice([1, 2, 3, 5, 8, 20]))
        n = 16 * int(rng.integers(4, 14))
        vals = np.zeros(n, np.int64)
        labels = np.arange(n, dtype=np.int64)
        level = 100
        for p in range(n):
            level -= int(rng.integers(0, 3))
            vals[p] = level + int(rng.integers(-3, 4))
        # a third of the rows become copies of a row 1 .. 40 positions back (not of the same block:
        # a list holds a label once)
        for p in rng.permutation(n)[: n // 3]:
            src = p - int(rng.integers(1, 41))
            if src >= 0 and src // 16 != p // 16 and labels[src] == src and labels[p] == p:
                labels[p], vals[p] = labels[src], vals[src]
        replay_with_rule(vals, labels, R, stats=stats)
    assert min(stats.get(k, 0) for k in ("first", "below", "above", "scan")) > 20, stats


def test_unsigned_range_and_heads_that_fill_the_heap():
    """values 0..255 (udistance_table), top = 255; lists shorter than the heap (fresh entries leave late)"""
    rng = np.random.default_rng(9)
    for trial in range(5):
        vals, labels = make_stream(rng, n_lists=6, rows=24, b=2, spread=5.0, R=111)
        replay_with_rule(np.clip(vals + 100, 0, 254), labels, 111, top=255)
