"""The two-registers-per-lane form of `insert` (_fast_pq.pyx:274-307) that the wave-per-query replay runs for heaps of
up to 129 entries (heap.hip: pair_heap_insert — IVF.query's heap of (n_probes + 1) k + 1 = 111 entries at the
reference's bench settings), restated in numpy over 64 "lanes" and checked against the oracle's loop.

Layout: the root is wave-uniform; lane L holds the two CHILDREN of node L — node 2L+1 in slot 0, node 2L+2 in slot 1
— so "which child of node L is larger" (the left one on ties: `vl > v`, then `vr > nxt_val`) is a comparison inside
the lane, and one ballot B of it over the wave describes the whole max-child path: node t is on it iff every ancestor
chose the child on the chain down to t (a precomputed (mask, bits) pair per lane).  The sift of `insert` follows that
path whatever v is, and values never rise along it, so with CE[L] = the larger child's entry of node L:
    root            <- CE[0]                         if CE[0].v > v   else (label, v)
    larger child c of an on-path node L with CE[L].v > v
                    <- CE[c]                         if c has a lane and CE[c].v > v   else (label, v)
— B, the path, CE and the fetch of CE[c] do not depend on v: one compare and two selects per lane remain once the
candidate is known.  Nodes >= R hold a value below every candidate and are never taken."""
import numpy as np
import pytest

SENT = -(1 << 20)


class PairHeap:
    def __init__(self, R, fresh):
        assert 1 <= R <= 129
        self.R = R
        L = np.arange(64)
        self.v = np.full((2, 64), SENT, dtype=np.int64)
        self.i = np.full((2, 64), -1, dtype=np.int64)
        for s in (0, 1):
            self.v[s][2 * L + 1 + s < R] = fresh
        self.rv, self.ri = fresh, -1
        # lane L > 0 = node L: its ancestors (as lanes) and the child each of them must have chosen
        self.ancmask = np.zeros(64, dtype=object)
        self.ancbits = np.zeros(64, dtype=object)
        for lane in range(64):
            m = b = 0
            t = lane
            while t > 0:
                p = (t - 1) // 2
                m |= 1 << p
                b |= ((t - 1) & 1) << p
                t = p
            self.ancmask[lane], self.ancbits[lane] = m, b

    def insert(self, label, v):
        L = np.arange(64)
        live = np.stack([2 * L + 1 < self.R, 2 * L + 2 < self.R])
        if self.ri == label or ((self.i == label) & live).any():        # :284-287
            return
        B = self.v[1] > self.v[0]                                       # the right child is the larger one
        ball = sum(1 << int(l) for l in L[B])
        onpath = np.array([((ball ^ self.ancbits[l]) & self.ancmask[l]) == 0 for l in L])
        ch = B.astype(int)
        cv = np.where(B, self.v[1], self.v[0])
        ci = np.where(B, self.i[1], self.i[0])
        c = 2 * L + 1 + ch
        has_lane = c < 64
        fv = np.where(has_lane, cv[np.minimum(c, 63)], SENT)            # CE[c]: one cross-lane fetch
        fi = np.where(has_lane, ci[np.minimum(c, 63)], -1)
        upd = onpath & (cv > v)
        nv = np.where(fv > v, fv, v)
        ni = np.where(fv > v, fi, label)
        new_rv, new_ri = (cv[0], ci[0]) if cv[0] > v else (v, label)
        for s in (0, 1):
            sel = upd & (ch == s)
            self.v[s] = np.where(sel, nv, self.v[s])
            self.i[s] = np.where(sel, ni, self.i[s])
        self.rv, self.ri = new_rv, new_ri

    def arrays(self):
        idx, val = np.empty(self.R, np.int64), np.empty(self.R, np.int32)
        idx[0], val[0] = self.ri, self.rv
        for t in range(1, self.R):
            idx[t], val[t] = self.i[(t - 1) & 1][(t - 1) >> 1], self.v[(t - 1) & 1][(t - 1) >> 1]
        return idx, val


@pytest.mark.parametrize("R", [1, 2, 3, 4, 7, 12, 30, 63, 64, 65, 66, 111, 127, 128, 129])
@pytest.mark.parametrize("spread", [3, 40, 250])
def test_pair_layout_insert_equals_the_loop(oracle, R, spread):
    rng = np.random.RandomState(R * 1000 + spread)
    for signd in (True, False):
        wi, wv = np.zeros(R, np.int64), np.zeros(R, np.int32)
        oracle.init_heap(wi, wv, signd)
        H = PairHeap(R, int(wv[0]))
        lo = -128 if signd else 0
        for step in range(700):
            # values near the current root: ties with the root, its children and each other; now and then a value ABOVE
            # the root (a stale bound lets such rows through, _fast_pq_256.pyx:111-123) and a label already in the heap
            base = int(wv[0])
            v = int(np.clip(base - rng.randint(0, spread) + (rng.randint(0, 6) if step % 11 == 0 else 0), lo, lo + 255))
            label = int(wi[rng.randint(R)]) if step % 17 == 5 and wi.max() >= 0 else step
            oracle.insert(wi, wv, label, v)
            H.insert(label, v)
            gi, gv = H.arrays()
            assert (gi == wi).all() and (gv == wv).all(), (R, signd, step)


def test_pair_layout_insert_on_falling_and_constant_streams(oracle):
    """Every sift reaches a leaf (falling values) / stops at the root (constant values): the two ends of the path."""
    for R in (5, 111, 129):
        for kind in ("falling", "constant", "rising"):
            wi, wv = np.zeros(R, np.int64), np.zeros(R, np.int32)
            oracle.init_heap(wi, wv, True)
            H = PairHeap(R, 127)
            for step in range(400):
                v = {"falling": max(-128, 126 - step // 2), "constant": 5, "rising": min(127, -128 + step)}[kind]
                oracle.insert(wi, wv, step, v)
                H.insert(step, v)
            gi, gv = H.arrays()
            assert (gi == wi).all() and (gv == wv).all(), (R, kind)


# ---------------------------------------------------------------------------------------------------------------------
# G groups of node pairs per lane (heap.hip: the same kernel for heaps of up to 128 G + 1 entries — G = 2: 257, the
# heap of n_probes <= 24 at k = 10): lane L, group g holds the children of node t = 64 g + L.  The path is still one
# ballot per group; a node's ancestors are all below 64 G / 2, i.e. in the lower half of the groups; CE[c] of the
# chosen child c = 2 t + 1 + right lives in lane c & 63 of group c >> 6 (if c < 64 G).
class GroupHeap:
    def __init__(self, R, fresh, G):
        assert 1 <= R <= 128 * G + 1
        self.R, self.G = R, G
        t = np.arange(64 * G)                       # internal-node slots: node t's children pair
        self.v = np.full((2, 64 * G), SENT, dtype=np.int64)
        self.i = np.full((2, 64 * G), -1, dtype=np.int64)
        for s in (0, 1):
            self.v[s][2 * t + 1 + s < R] = fresh
        self.rv, self.ri = fresh, -1
        self.anc = []                               # per node t: list of (ancestor a, bit the ancestor must have chosen)
        for node in range(64 * G):
            ch, x = [], node
            while x > 0:
                par = (x - 1) // 2
                ch.append((par, (x - 1) & 1))
                x = par
            self.anc.append(ch)

    def insert(self, label, v):
        G, R = self.G, self.R
        t = np.arange(64 * G)
        live = np.stack([2 * t + 1 < R, 2 * t + 2 < R])
        if self.ri == label or ((self.i == label) & live).any():
            return
        B = self.v[1] > self.v[0]                   # G ballots of 64 bits, concatenated
        onpath = np.array([all(bool(B[a]) == bool(bit) for a, bit in self.anc[n]) for n in t])
        ch = B.astype(int)
        cv = np.where(B, self.v[1], self.v[0])
        ci = np.where(B, self.i[1], self.i[0])
        c = 2 * t + 1 + ch
        has = c < 64 * G
        fv = np.where(has, cv[np.minimum(c, 64 * G - 1)], SENT)
        fi = np.where(has, ci[np.minimum(c, 64 * G - 1)], -1)
        upd = onpath & (cv > v)
        nv = np.where(fv > v, fv, v)
        ni = np.where(fv > v, fi, label)
        new_root = (cv[0], ci[0]) if cv[0] > v else (v, label)
        for s in (0, 1):
            sel = upd & (ch == s)
            self.v[s] = np.where(sel, nv, self.v[s])
            self.i[s] = np.where(sel, ni, self.i[s])
        self.rv, self.ri = new_root

    def arrays(self):
        idx, val = np.empty(self.R, np.int64), np.empty(self.R, np.int32)
        idx[0], val[0] = self.ri, self.rv
        for n in range(1, self.R):
            idx[n], val[n] = self.i[(n - 1) & 1][(n - 1) >> 1], self.v[(n - 1) & 1][(n - 1) >> 1]
        return idx, val


def test_group_layout_ancestors_live_in_the_lower_groups():
    """What the kernel's per-lane constants rest on: the ancestors of internal node t < 64 G are all < 32 G."""
    for G in (1, 2, 4):
        for node in range(64 * G):
            x = node
            while x > 0:
                x = (x - 1) // 2
                assert x < 32 * G


@pytest.mark.parametrize("G,R", [(2, 130), (2, 131), (2, 193), (2, 211), (2, 256), (2, 257), (2, 111), (4, 258), (4, 511), (4, 513)])
@pytest.mark.parametrize("spread", [3, 60])
def test_group_layout_insert_equals_the_loop(oracle, G, R, spread):
    rng = np.random.RandomState(R * 7 + spread + G)
    for signd in (True, False):
        wi, wv = np.zeros(R, np.int64), np.zeros(R, np.int32)
        oracle.init_heap(wi, wv, signd)
        H = GroupHeap(R, int(wv[0]), G)
        lo = -128 if signd else 0
        for step in range(900):
            base = int(wv[0])
            v = int(np.clip(base - rng.randint(0, spread) + (rng.randint(0, 6) if step % 11 == 0 else 0), lo, lo + 255))
            label = int(wi[rng.randint(R)]) if step % 17 == 5 and wi.max() >= 0 else step
            oracle.insert(wi, wv, label, v)
            H.insert(label, v)
            if step % 7 == 0 or step > 880:
                gi, gv = H.arrays()
                assert (gi == wi).all() and (gv == wv).all(), (G, R, signd, step)
