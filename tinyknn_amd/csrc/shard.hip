// shard.hip — the exchange step of a list-sharded index.
//
// Inverted lists are partitioned by cluster id over W ranks (owner[l]); every rank sees
// every query, derives the same probe order (the coarse stage is replicated) and scores
// only the (query, probed list) segments whose list it owns.  Because the reference pushes
// all probed lists of a query through ONE order-dependent heap (ivf.py:137-150), partial
// top-k's cannot be merged; what travels is the int8 distance bytes themselves (1/26-1/16
// of the code bytes), to the query's home rank, which replays the heap exactly.
//
// Fixed-shape exchange (no host synchronisation): the stream (source s -> home h) is the
// concatenation, in (query, probe slot) order, of the segments of h's queries that s owns,
// in a region of `C` uint4 of the all-to-all buffers.  Source and home compute the same
// positions from the replicated probe lists, so no metadata travels.  A stream longer than
// C raises `flag`; the caller then repeats the batch with a larger C.
#include "kernels.h"

// Entry i < nq*S: sender role — (query i/S, slot i%S), owned if the list is mine.
// Entry nq*S + p*qh*S + e: receiver role — entry e of MY home queries, owned if peer p has it.
// One exclusive prefix sum over the owned lengths of all entries (tk_scan_exclusive) gives every
// stream's running offset: a stream's entries are consecutive, so offset = P[i] - P[first entry
// of the stream].  (Round 1 walked each stream with ONE workgroup: 0.5 ms for the 300 000
// entries of a 30 000-query batch at W = 1, a fifth of the batch.)
__device__ __forceinline__ bool shard_entry(int64_t i, const int64_t *__restrict__ probes,
                                            const int *__restrict__ slot_prefix, int S, int64_t nq,
                                            int64_t n_lists, const int *__restrict__ owner, int me,
                                            int64_t qh, int64_t &qi, int &sl, int &len)
{
    const int64_t n_s = nq * S;
    int own;
    if (i < n_s) {
        qi = i / S;
        sl = (int)(i - qi * S);
        own = me;
    } else {
        const int64_t j = i - n_s;
        own = (int)(j / (qh * S));
        const int64_t e = j - (int64_t)own * qh * S;
        qi = (int64_t)me * qh + e / S;
        sl = (int)(e % S);
    }
    len = 0;
    if (qi >= nq) return false;
    int64_t cl = probes[qi * S + sl];
    if (cl < 0) cl += n_lists;
    if (owner[cl] != own) return false;
    len = slot_prefix[qi * (S + 1) + sl + 1] - slot_prefix[qi * (S + 1) + sl];
    return true;
}

__global__ void shard_lens_kernel(const int64_t *__restrict__ probes,
                                  const int *__restrict__ slot_prefix, int S, int64_t nq,
                                  int64_t n_lists, const int *__restrict__ owner, int me, int W,
                                  int64_t qh, long long *__restrict__ lens)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_tot = nq * S + (int64_t)W * qh * S;
    if (i > n_tot) return;
    int64_t qi;
    int sl, len = 0;
    if (i < n_tot) shard_entry(i, probes, slot_prefix, S, nq, n_lists, owner, me, qh, qi, sl, len);
    lens[i] = len;                                     // entry n_tot: 0, so that P[n_tot] = total
}

// spos (nq*S): position in the send buffer of every slot I own (region = home rank), -1
// otherwise.  rpos (qh*S): position in the receive buffer of the slots of MY queries
// (region = owner).  usage[0..W): lengths of my send streams, [W..2W): of my receive streams.
__global__ void shard_place_kernel(const int64_t *__restrict__ probes,
                                   const int *__restrict__ slot_prefix, int S, int64_t nq,
                                   int64_t n_lists, const int *__restrict__ owner, int me, int W,
                                   int64_t qh, int64_t C, const long long *__restrict__ P,
                                   int *__restrict__ spos, int *__restrict__ rpos,
                                   int *__restrict__ flag, long long *__restrict__ usage)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_s = nq * S, seg = qh * S, n_tot = n_s + (int64_t)W * seg;
    if (i < 2 * W && usage) {
        int64_t a, b;
        if (i < W) {
            a = i * seg;
            b = a + seg;
            a = a > n_s ? n_s : a;
            b = b > n_s ? n_s : b;
        } else {
            a = n_s + (i - W) * seg;
            b = a + seg;
        }
        usage[i] = P[b] - P[a];
    }
    if (i >= n_tot) return;
    int64_t qi;
    int sl, len;
    const bool mine = shard_entry(i, probes, slot_prefix, S, nq, n_lists, owner, me, qh, qi, sl, len);
    if (i < n_s) {
        int where = -1;
        if (mine) {
            const int64_t h = qi / qh;
            int64_t a = h * seg;
            a = a > n_s ? n_s : a;
            const int64_t run = P[i] - P[a];
            if (run + len <= C) where = (int)(h * C + run);
            else atomicOr(flag, 1);
        }
        spos[i] = where;
    } else if (mine) {
        const int64_t peer = (i - n_s) / seg;
        const int64_t run = P[i] - P[n_s + peer * seg];
        if (run + len <= C) rpos[(qi - (int64_t)me * qh) * S + sl] = (int)(peer * C + run);
        else {
            rpos[(qi - (int64_t)me * qh) * S + sl] = -1;
            atomicOr(flag, 1);
        }
    }
}

// lens / P: nq*S + W*qh*S + 1 int64 each (a batch may score more than 2^31 blocks in
// total); tmp: room for tk_scan_exclusive64 over that many
int tk_launch_shard_positions(const int64_t *probes, const int *slot_prefix, int S, int64_t nq,
                              int64_t n_lists, const int *owner, int me, int W, int64_t qh,
                              int64_t C, int *spos, int *rpos, int *flag, long long *usage,
                              long long *lens, long long *P, void *tmp, size_t tmp_bytes,
                              hipStream_t s)
{
    if (nq == 0 || S == 0) return 0;
    const int64_t n1 = nq * S + (int64_t)W * qh * S + 1;
    const unsigned grid = (unsigned)((n1 + 255) / 256);
    hipLaunchKernelGGL(shard_lens_kernel, dim3(grid), dim3(256), 0, s, probes, slot_prefix, S, nq,
                       n_lists, owner, me, W, qh, lens);
    if (tk_scan_exclusive64(tmp, &tmp_bytes, lens, P, n1, s)) return -1;
    hipLaunchKernelGGL(shard_place_kernel, dim3(grid), dim3(256), 0, s, probes, slot_prefix, S, nq,
                       n_lists, owner, me, W, qh, C, P, spos, rpos, flag, usage);
    return 0;
}

// (query, slot) records of the lists this rank owns, grouped by list, for the list-major
// scan; the record's row offset is the segment's position in the send buffer.  Segments
// that did not fit become padding records.
__global__ void shard_pairs_fill_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                        int64_t n_lists, const int *__restrict__ owner, int me,
                                        const int *__restrict__ spos,
                                        const int *__restrict__ pair_off, int *__restrict__ cursor,
                                        int *__restrict__ pair_q, int *__restrict__ pair_f0,
                                        int s_lo, int s_hi, const uint8_t *__restrict__ sel, int want,
                                        int ovf_pos)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * S) return;
    const int64_t q = i / S;
    const int sl = (int)(i - q * S);
    if (sl < s_lo || sl >= s_hi) return;             // (two-phase scan: first lists, then the rest)
    if (sel && (int)sel[q] != want) return;          // ... the rest by kernel: exact / plain
    int64_t cl = probes[i];
    if (cl < 0) cl += n_lists;
    if (owner[cl] != me) return;
    const int pos = atomicAdd(&cursor[cl], 1);
    const int p = spos[i];
    // a segment that did not fit its region (the batch is flagged and repeated): a padding record
    // for the exact kernel; the plain kernel has none — it scores the pair to ovf_pos, the tail of
    // the buffer, where the longest list fits
    if (ovf_pos >= 0) {
        pair_q[pair_off[cl] + pos] = (int)q;
        pair_f0[pair_off[cl] + pos] = p >= 0 ? p : ovf_pos;
    } else {
        pair_q[pair_off[cl] + pos] = p >= 0 ? (int)q : -1;
        pair_f0[pair_off[cl] + pos] = p >= 0 ? p : 0;
    }
}

void tk_launch_shard_pairs_fill(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                const int *owner, int me, const int *spos, const int *pair_off,
                                int *cursor, int *pair_q, int *pair_f0, hipStream_t s, int s_lo,
                                int s_hi, const uint8_t *sel, int want, int ovf_pos)
{
    const int64_t np = nq * S;
    if (np == 0) return;
    hipLaunchKernelGGL(shard_pairs_fill_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s,
                       probes, S, nq, n_lists, owner, me, spos, pair_off, cursor, pair_q, pair_f0,
                       s_lo, s_hi, sel, want, ovf_pos);
}

// ---- two-phase scan (tk_index_shard_scan_first_dev / _rest_dev): pairs per list of the first slots ...
__global__ void shard_count_first_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                         int64_t n_lists, const int *__restrict__ owner, int me,
                                         int *__restrict__ count)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int64_t cl = probes[q * S];
    if (cl < 0) cl += n_lists;
    if (owner[cl] == me) atomicAdd(&count[cl], 1);
}

// ... and of the slots behind them, by kernel.  plain_q[q] = 1: the bound after the query's first
// list — min-reduced over the ranks, the same byte everywhere — is at most the table's limit
// (plain_scan.hip's lemma: from there on the replay over clamp(plain sums) IS the replay over the
// reference's values), so every rank scores its lists of that query on the matrix cores; else exactly.
__global__ void shard_count_rest_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                        int64_t n_lists, const int *__restrict__ owner, int me,
                                        const uint8_t *__restrict__ bound, const int *__restrict__ qlim,
                                        int allow, uint8_t *__restrict__ plain_q,
                                        int *__restrict__ count_exact, int *__restrict__ count_plain)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    bool wrapped = false;
    for (int sl = 0; sl < S; sl++) wrapped |= probes[q * S + sl] < 0;
    const int b = (int)(int8_t)(bound[q] ^ 0x80u);             // (order key -> the distance byte)
    const int lim = qlim[q];
    const bool pl = allow && !wrapped && lim != TK_PLAIN_NEVER && b <= lim;
    plain_q[q] = pl ? 1 : 0;
    for (int sl = 1; sl < S; sl++) {
        int64_t cl = probes[q * S + sl];
        if (cl < 0) cl += n_lists;
        if (owner[cl] == me) atomicAdd(pl ? &count_plain[cl] : &count_exact[cl], 1);
    }
}

void tk_launch_shard_count_first(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                 const int *owner, int me, int *count, hipStream_t s)
{
    if (nq == 0 || S == 0) return;
    hipLaunchKernelGGL(shard_count_first_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s,
                       probes, S, nq, n_lists, owner, me, count);
}

void tk_launch_shard_count_rest(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                const int *owner, int me, const uint8_t *bound, const int *qlim,
                                int allow, uint8_t *plain_q, int *count_exact, int *count_plain,
                                hipStream_t s)
{
    if (nq == 0 || S == 0) return;
    hipLaunchKernelGGL(shard_count_rest_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s,
                       probes, S, nq, n_lists, owner, me, bound, qlim, allow, plain_q, count_exact,
                       count_plain);
}

// Received segments -> the home queries' distance rows (the layout the replay kernels
// read) + the per-chunk minima the scan kernels would have written.
// One wave per (home query, slot), four to a workgroup (one-wave workgroups were bound by
// the dispatch rate: 300 000 of them for a 30 000-query exchange took 0.39 ms).
template <bool SIGNED>
__global__ __launch_bounds__(256) void shard_unpack_kernel(
    const uint4 *__restrict__ recv, const int *__restrict__ rpos,
    const int *__restrict__ slot_prefix, int S, int64_t n_pairs, uint4 *__restrict__ dist,
    int64_t cap, uint8_t *__restrict__ mins, int64_t min_stride)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_pairs) return;
    const int64_t qi = b / S;
    const int s = (int)(b - qi * S);
    const int f0 = slot_prefix[qi * (S + 1) + s];
    const int n = slot_prefix[qi * (S + 1) + s + 1] - f0;
    const int p = rpos[b];
    for (int c = threadIdx.x & 63; c < n; c += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p >= 0) v = recv[(int64_t)p + c];
        dist[qi * cap + f0 + c] = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int m = SIGNED ? 127 : 255;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t byte = (w[j] >> (8 * t)) & 0xffu;
                const int x = SIGNED ? (int)(int8_t)byte : (int)byte;
                m = x < m ? x : m;
            }
        mins[qi * min_stride + f0 + c] = (uint8_t)m;
    }
}

void tk_launch_shard_unpack(const uint4 *recv, const int *rpos, const int *slot_prefix, int S,
                            int64_t nq_home, uint4 *dist, int64_t cap, uint8_t *mins,
                            int64_t min_stride, int signd, hipStream_t s)
{
    if (nq_home == 0 || S == 0) return;
    const int64_t n_pairs = nq_home * S;
    const unsigned grid = (unsigned)((n_pairs + 3) / 4);
    if (signd)
        hipLaunchKernelGGL(shard_unpack_kernel<true>, dim3(grid), dim3(256), 0, s, recv, rpos,
                           slot_prefix, S, n_pairs, dist, cap, mins, min_stride);
    else
        hipLaunchKernelGGL(shard_unpack_kernel<false>, dim3(grid), dim3(256), 0, s, recv, rpos,
                           slot_prefix, S, n_pairs, dist, cap, mins, min_stride);
}

// One-phase sharded scan on the matrix cores (tk_index_shard_scan_plain_dev): the home rank's lane replay
// marks a query (flags[q] == 2) whose bound at its first plain block was above the limit of its table —
// the lemma of plain_scan.hip does not cover it.  The unsharded index scans such a query again exactly;
// a home rank does not hold the codes, so the BATCH is flagged (bit 4 of its overflow word, which
// travels with the ids) and the caller repeats it on the exact kernel.
__global__ void shard_flag_plain_kernel(const unsigned char *__restrict__ flags, int64_t nq, int *__restrict__ flag)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = q < nq && flags[q] == 2;
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 4);
}

void tk_launch_shard_flag_plain(const unsigned char *flags, int64_t nq, int *flag, hipStream_t s)
{
    if (nq == 0) return;
    hipLaunchKernelGGL(shard_flag_plain_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, flags, nq, flag);
}

// ---------------------------------------------------------------------------
// Filtered exchange.  Every insert of a block is below the bound
// captured at the block's start (_fast_pq_256.pyx:73,111-123), so the bound never increases
// from one block to the next; a distance that is not below B1 — the bound after the query's
// FIRST probed list — can therefore never enter the heap, whatever comes before it.  The
// owner of a query's first list replays that list from the fresh heap (B1), B1 is min-reduced
// over the ranks (1 byte per query), and of the later lists only the 16-distance blocks whose
// minimum is below B1 travel, as (destination, 16 bytes) records.  The home rank drops them
// into distance rows pre-filled with the largest value (what pad rows carry, `pos < n`), where
// a missing block behaves exactly like the real one: no byte of it is below any bound.
//
// Signed tables: the order key of a distance byte is byte ^ 0x80 (unsigned compare).
__device__ __forceinline__ uint32_t key8(uint32_t b) { return (b ^ 0x80u) & 0xffu; }

// B1 by value multiset: insert() (_fast_pq.pyx:274-307) replaces the root — the maximum of a
// valid max-heap — and sifts down, so the VALUES of the heap after an insert are the old ones
// minus the maximum plus the new one, whatever the layout; labels of one list are distinct and
// the heap starts empty, so the duplicate test never fires here.  One query per wave: the 64
// lanes fetch 64 blocks' minima and — where a minimum is below the bound so far — distances at
// once (coalesced, no dependent loads), park them in LDS, and the wave walks them in order
// with uniform state: 16 lanes test a block's distances, the passing ones enter the multiset —
// 256 counters in LDS, the count of the current maximum in a register: an insert below the
// maximum is one ds_add and a decrement.
__global__ __launch_bounds__(256) void shard_first_bound_kernel(
    const int64_t *__restrict__ probes, const int *__restrict__ slot_prefix,
    const int *__restrict__ slot_n, int S, int64_t nq, int64_t n_lists,
    const int *__restrict__ owner, int me, const int *__restrict__ spos,
    const uint4 *__restrict__ scan, const uint8_t *__restrict__ smins, int R,
    uint8_t *__restrict__ bound, int max_chunks)
{
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint4 s_blk[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    if (q >= nq) return;
    int64_t cl = probes[q * S];
    if (cl < 0) cl += n_lists;
    const int p = spos[q * S];
    if (owner[cl] != me || p < 0) {
        if (lane == 0) bound[q] = 255;
        return;
    }
    uint32_t *hist = s_hist[wv];
    for (int b = lane; b < 256; b += 64) hist[b] = b == 255 ? (uint32_t)R : 0u;   // init_heap: R x 127
    int nch = slot_prefix[q * (S + 1) + 1] - slot_prefix[q * (S + 1)];
    if (max_chunks > 0 && max_chunks < nch) nch = max_chunks;      // (the bound after the list's HEAD only)
    int n = slot_n[q * S];
    n = n < 0 ? 0 : n;
    int mx = 255;                                      // wave-uniform: the bound so far ...
    uint32_t cm = (uint32_t)R;                         // ... and how many heap entries hold it
    for (int c0 = 0; c0 < nch; c0 += 64) {
        const int c = c0 + lane;
        const uint32_t mk = c < nch ? key8(smins[(int64_t)p + c]) : 255u;
        if (mk < (uint32_t)mx) s_blk[wv][lane] = scan[(int64_t)p + c];
        // (same wave: LDS traffic is ordered, no barrier needed between the lanes' stores and
        //  lane 0's loads — but the compiler must not reorder them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // wave-uniform walk over the parked blocks: lanes 0..15 test the 16 distances of a block
        // at once, the passing ones are inserted in row order with uniform (scalar) state
        // (only the blocks whose minimum is below the bound at the start of this round of 64 —
        //  the bound only falls from block to block, so the others cannot pass later either)
        uint64_t cand = __ballot(mk < (uint32_t)mx);
        while (cand) {
            const int j = __builtin_ctzll(cand);
            cand &= cand - 1;
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)mk, j);
            if (kj >= (uint32_t)mx) continue;
            const uint32_t wsel = reinterpret_cast<const uint32_t *>(&s_blk[wv][j])[(lane >> 2) & 3];
            const uint32_t kv = key8(wsel >> (8 * (lane & 3)));
            // bound at block start = mx: every lane that passes is inserted without re-check
            uint64_t mask = __ballot(lane < 16 && kv < (uint32_t)mx && 16 * (c0 + j) + lane < n);
            while (mask) {
                const int e = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int kve = __builtin_amdgcn_readlane((int)kv, e);
                if (kve < mx) {
                    if (lane == 0) atomicAdd(&hist[kve], 1u);
                    if (--cm == 0) {
                        if (lane == 0) hist[mx] = 0;
                        do {
                            mx--;
                            cm = (uint32_t)__builtin_amdgcn_readfirstlane((int)hist[mx]);
                        } while (cm == 0);
                    }
                } else if (kve > mx) {                 // the maximum leaves, kve is the new one
                    if (lane == 0) hist[mx] = cm - 1;
                    mx = kve;
                    cm = 1;
                }                                      // kve == mx: one out, one in
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) bound[q] = (uint8_t)mx;
}

void tk_launch_shard_first_bound(const int64_t *probes, const int *slot_prefix, const int *slot_n,
                                 int S, int64_t nq, int64_t n_lists, const int *owner, int me,
                                 const int *spos, const uint4 *scan, const uint8_t *smins, int R,
                                 uint8_t *bound, hipStream_t s, int max_chunks)
{
    if (nq == 0 || S == 0) return;
    hipLaunchKernelGGL(shard_first_bound_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s,
                       probes, slot_prefix, slot_n, S, nq, n_lists, owner, me, spos, scan, smins, R,
                       bound, max_chunks);
}

// tk_index_shard_scan_plain_dev behind tk_index_shard_scan_head_dev: a query whose bound after the head of
// its first list (min-reduced over the ranks: order key, byte ^ 0x80) is above the limit of its table is
// taken off the plain path — its limit becomes TK_PLAIN_NEVER, make_slots then keeps all its lists exact.
__global__ void shard_mask_limits_kernel(const uint8_t *__restrict__ bound, int64_t nq, int *__restrict__ qlim)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int b = (int)(int8_t)(bound[q] ^ 0x80u);
    if (b > qlim[q]) qlim[q] = TK_PLAIN_NEVER;
}

void tk_launch_shard_mask_limits(const uint8_t *bound, int64_t nq, int *qlim, hipStream_t s)
{
    if (nq == 0) return;
    hipLaunchKernelGGL(shard_mask_limits_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, bound, nq, qlim);
}

// One wave per (query, slot) this rank owns.  PACK = false: blocks that pass -> pair_cnt
// (0 for pairs of other ranks), all owned blocks -> dense[256][W] partial sums (for the books;
// spread over 256 addresses: one counter per home rank serialised 75 000 atomics at W = 1).  The pairs
// are in (query, slot) order and a home rank's queries are consecutive, so an exclusive
// prefix sum of pair_cnt (tk_scan_exclusive) IS the layout of the records: region h starts at
// pair_off[h * qh * S].  PACK = true: the records at their pair's offset, in block order.
template <bool PACK>
__global__ __launch_bounds__(256) void shard_filter_kernel(
    const int64_t *__restrict__ probes, const int *__restrict__ slot_prefix, int S, int64_t nq,
    int64_t n_lists, const int *__restrict__ owner, int me, int W, int64_t qh, int64_t cap,
    const int *__restrict__ spos, const uint4 *__restrict__ scan,
    const uint8_t *__restrict__ smins, const uint8_t *__restrict__ bound,
    int *__restrict__ pair_cnt, const int *__restrict__ pair_off, int *__restrict__ dense,
    int *__restrict__ rec, int region)
{
    __shared__ int s_dense[4];
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int mine = 0;                                      // owned blocks of this wave's pair
    if (i < nq * S) {
        const int64_t q = i / S;
        const int sl = (int)(i - q * S);
        int64_t cl = probes[i];
        if (cl < 0) cl += n_lists;
        const int p = spos[i];
        if (owner[cl] != me || p < 0) {
            if (!PACK && lane == 0) pair_cnt[i] = 0;
        } else {
            const int f0 = slot_prefix[q * (S + 1) + sl];
            const int nch = slot_prefix[q * (S + 1) + sl + 1] - f0;
            const uint32_t b = sl == 0 ? 256u : (uint32_t)bound[q];     // the first list travels whole
            const int h = (int)(q / qh);
            if (!PACK) {
                int cnt = 0;
                int c = lane;
                const uint8_t *mp = smins + (int64_t)p;
                for (; c + 192 < nch; c += 256) {      // four loads in flight
                    const uint32_t m0 = mp[c], m1 = mp[c + 64], m2 = mp[c + 128], m3 = mp[c + 192];
                    cnt += (key8(m0) < b) + (key8(m1) < b) + (key8(m2) < b) + (key8(m3) < b);
                }
                for (; c < nch; c += 64) cnt += key8(mp[c]) < b;
                for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
                if (lane == 0) pair_cnt[i] = cnt;
                mine = nch;
            } else if (pair_cnt[i] > 0) {
                int base = pair_off[i];
                int room = 0x7fffffff;
                if (region > 0) {
                    // fixed regions (tk_index_shard_filter_regions_dev): home rank h's records start
                    // at h * region whatever the other homes hold; what does not fit is dropped
                    // (shard_counts_kernel raises the overflow flag)
                    int64_t a = (int64_t)h * qh;
                    a = a > nq ? nq : a;
                    const int local = base - pair_off[a * S];
                    room = region - local;
                    base = h * region + local;
                }
                const int hdr0 = (int)((q - (int64_t)h * qh) * cap + f0);
                const uint8_t *mp = smins + (int64_t)p;
                const uint4 *sp = scan + (int64_t)p;
                for (int c0 = 0; c0 < nch; c0 += 256) {    // four rounds of 64 with their loads in flight
                    bool pass[4];
                    uint4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int c = c0 + 64 * u + lane;
                        pass[u] = c < nch && key8(mp[c < nch ? c : 0]) < b;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (pass[u]) v[u] = sp[c0 + 64 * u + lane];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint64_t m = __ballot(pass[u]);
                        const int ahead = __popcll(m & ((1ull << lane) - 1));
                        if (pass[u] && ahead < room) {
                            const int at = base + ahead;
                            int *r = rec + (int64_t)at * 5;
                            r[0] = hdr0 + c0 + 64 * u + lane;
                            r[1] = (int)v[u].x; r[2] = (int)v[u].y; r[3] = (int)v[u].z; r[4] = (int)v[u].w;
                        }
                        base += __popcll(m);
                        room -= __popcll(m);
                    }
                }
            }
        }
    }
    if (!PACK) {
        // the four pairs of a workgroup are consecutive: almost always one home rank
        if (lane == 0) s_dense[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int64_t i0 = (int64_t)blockIdx.x * 4;
            int acc = 0, hprev = -1;
            for (int t = 0; t < 4 && i0 + t < nq * S; t++) {
                const int h = (int)((i0 + t) / S / qh);
                if (h != hprev && acc) { atomicAdd(&dense[(blockIdx.x & 255) * W + hprev], acc); acc = 0; }
                hprev = h;
                acc += s_dense[t];
            }
            if (acc) atomicAdd(&dense[(blockIdx.x & 255) * W + hprev], acc);
        }
    }
}

// counts[h] = records for home rank h, from the prefix sums (pair_off has nq * S + 1 entries)
__global__ void shard_counts_kernel(const int *__restrict__ pair_off, int S, int64_t nq, int W,
                                    int64_t qh, const int *__restrict__ tally,
                                    int *__restrict__ counts, int region, int *__restrict__ flag,
                                    long long *__restrict__ acc)
{
    const int h = threadIdx.x + blockIdx.x * blockDim.x;
    if (h >= W) return;
    int64_t a = (int64_t)h * qh, b = a + qh;
    a = a > nq ? nq : a;
    b = b > nq ? nq : b;
    counts[h] = pair_off[b * S] - pair_off[a * S];
    if (region > 0 && counts[h] > region && flag) atomicOr(flag, 1);    // (the home rank reads min(count, region))
    int t = 0;
    for (int r = 0; r < 256; r++) t += tally[r * W + h];
    counts[2 * W + h] = t;
    if (acc) {      // the caller's books, kept on the device: largest region, records, blocks scored
        atomicMax(&acc[0], (long long)counts[h]);
        atomicAdd((unsigned long long *)&acc[1], (unsigned long long)counts[h]);
        atomicAdd((unsigned long long *)&acc[2], (unsigned long long)t);
    }
}

int tk_scan_exclusive(void *tmp, size_t *tmp_bytes, const int *in, int *out, int64_t n, hipStream_t s);

int tk_launch_shard_filter(const int64_t *probes, const int *slot_prefix, int S, int64_t nq,
                           int64_t n_lists, const int *owner, int me, int W, int64_t qh,
                           int64_t cap, const int *spos, const uint4 *scan, const uint8_t *smins,
                           const uint8_t *bound, int *pair_cnt, int *pair_off, void *tmp,
                           size_t tmp_bytes, int *tally, int *counts, int *rec, hipStream_t s,
                           int region, int *flag, long long *acc)
{
    if (nq == 0 || S == 0) return 0;
    const unsigned grid = (unsigned)((nq * S + 3) / 4);
    hipLaunchKernelGGL(shard_filter_kernel<false>, dim3(grid), dim3(256), 0, s, probes, slot_prefix,
                       S, nq, n_lists, owner, me, W, qh, cap, spos, scan, smins, bound, pair_cnt,
                       (const int *)nullptr, tally, rec, 0);
    // pair_cnt has one more entry (zero, set by the caller) so that pair_off[nq * S] = total
    if (tk_scan_exclusive(tmp, &tmp_bytes, pair_cnt, pair_off, nq * S + 1, s)) return -1;
    hipLaunchKernelGGL(shard_counts_kernel, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, s, pair_off,
                       S, nq, W, qh, tally, counts, region, flag, acc);
    hipLaunchKernelGGL(shard_filter_kernel<true>, dim3(grid), dim3(256), 0, s, probes, slot_prefix,
                       S, nq, n_lists, owner, me, W, qh, cap, spos, scan, smins, bound, pair_cnt,
                       (const int *)pair_off, tally, rec, region);
    return 0;
}

// Home side: rows of the home queries filled with the largest value, then the received blocks
// dropped where their headers say, with the block minimum the scan would have written.
__global__ __launch_bounds__(256) void shard_fill_rows_kernel(
    const int *__restrict__ slot_prefix, int S, uint4 *__restrict__ dist, int64_t cap,
    uint8_t *__restrict__ mins, int64_t min_stride)
{
    const int64_t qi = blockIdx.x;
    const int len = slot_prefix[qi * (S + 1) + S];
    const uint4 f = make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu);
    for (int c = threadIdx.x; c < len; c += 256) dist[qi * cap + c] = f;
    int64_t ml = ((int64_t)len + 15) / 16 * 16;
    if (ml > min_stride) ml = min_stride;
    for (int c = threadIdx.x; c < ml; c += 256) mins[qi * min_stride + c] = 0x7f;
}

__global__ __launch_bounds__(256) void shard_scatter_kernel(
    const int *__restrict__ rec, int64_t n_rec, int64_t rows, uint4 *__restrict__ dist,
    int64_t cap, uint8_t *__restrict__ mins, int64_t min_stride, int *__restrict__ bad,
    const int *__restrict__ counts_recv, int region)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rec) return;
    if (region > 0) {       // fixed regions: source rank s sent counts_recv[s] records at s * region
        const int64_t src = i / region;
        if (i - src * region >= (int64_t)counts_recv[src]) return;
    }
    const int *r = rec + i * 5;
    const int64_t hdr = r[0];
    if (hdr < 0 || hdr >= rows * cap) {            // a record that is not ours: never write it
        atomicOr(bad, 2);
        return;
    }
    const uint4 v = make_uint4((uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3], (uint32_t)r[4]);
    dist[hdr] = v;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    int m = 127;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int x = (int)(int8_t)((w[j] >> (8 * t)) & 0xffu);
            m = x < m ? x : m;
        }
    const int64_t qi = hdr / cap;
    mins[qi * min_stride + (hdr - qi * cap)] = (uint8_t)m;
}

void tk_launch_shard_expand(const int *rec, int64_t n_rec, const int *slot_prefix, int S,
                            int64_t nq_home, uint4 *dist, int64_t cap, uint8_t *mins,
                            int64_t min_stride, int *bad, hipStream_t s, const int *counts_recv,
                            int region)
{
    if (nq_home == 0 || S == 0) return;
    hipLaunchKernelGGL(shard_fill_rows_kernel, dim3((unsigned)nq_home), dim3(256), 0, s,
                       slot_prefix, S, dist, cap, mins, min_stride);
    if (n_rec > 0)
        hipLaunchKernelGGL(shard_scatter_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0,
                           s, rec, n_rec, nq_home, dist, cap, mins, min_stride, bad, counts_recv,
                           counts_recv ? region : 0);
}
