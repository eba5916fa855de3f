// heap.hip — exact replay of the reference's bounded top-R "heap" on gfx950.
//
// Replaces the selection half of query_pq_sse (_fast_pq.pyx:153-206) /
// query_pq_avx (_fast_pq_256.pyx:73-123) and the heap primitives init_heap,
// insert, insert_is (_fast_pq.pyx:240-307).
//
// The reference's result is a function of stream order and heap layout (SURVEY §0
// fact 2): per 16-code block it compares against the bound captured at block
// start, inserts EVERY passing lane without re-checking, and refreshes the bound
// once per block.  `insert` drops a label that is already present anywhere and
// otherwise replaces the root and sifts down (left child unless the right one is
// strictly greater).  This file replays exactly that from the int8 distances the scan
// kernels wrote.  Three kernels, all bit-exact (tests compare heap arrays, layout
// included, with the compiled reference):
//
//   heap_replay_lanes_kernel   one query per LANE (64 per wave), packed 32-bit entries
//                              in LDS columns, register-prefetched distance blocks, block
//                              minima, top heap levels in registers; fresh heaps only;
//                              with DEDUPE also labels that can repeat.  The batch path.
//   heap_replay_packed_kernel  one query per WAVE on the same packed entries (heaps too
//                              big for the lane kernel's LDS budget).
//   heap_replay_kernel         the general form below: int64 labels, arbitrary incoming
//                              heaps (tk_query_pq continues whatever the caller passes).
//
// The general kernel, one wavefront per query:
//
//   * the int8 distances were produced by adc_scan.hip (16 per chunk);
//   * a step covers 64 blocks (1024 codes): lane b holds block b's 16 bytes and
//     votes "some byte < bound" with the bound at step start.  The bound never
//     increases while the heap is a valid max-heap of 8-bit values, so the vote is
//     a superset of the blocks the reference would enter (when the caller hands
//     in arrays that are not such a heap, every block is examined instead);
//   * voted blocks are visited in ascending order; each is re-tested against the
//     live bound (= its true block-start bound, all earlier blocks being done),
//     passing lanes are inserted in lane order, the bound is refreshed after the
//     block, and the remaining votes are re-filtered;
//   * the heap (int64 id, int32 value) lives in LDS; the duplicate-label scan is a
//     64-lane compare + ballot, the sift-down runs wave-uniformly.
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"

// Heap state shared by the lanes of a wave lives in LDS and is accessed through
// address-space-3 volatile pointers: volatile keeps every access in program order
// (one lane stores, all lanes load), and the explicit address space keeps them
// ds_read/ds_write — a volatile access through a generic pointer compiles to
// flat_load/flat_store sc0 sc1 with a vmcnt(0) wait each, ~10x slower.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile int32_t lds_vi32;
typedef __attribute__((address_space(3))) volatile int64_t lds_vi64;
#define LDS_PTR(T, p) ((T *)(__attribute__((address_space(3))) unsigned char *)(p))

template <bool SIGNED>
__device__ __forceinline__ bool byte_lt(uint32_t d, uint32_t bound8)
{
    if (SIGNED) return (int)(int8_t)d < (int)(int8_t)bound8;
    return d < bound8;
}

template <bool SIGNED>
__device__ __forceinline__ bool any_lt16(const uint4 v, uint32_t bound8)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int b = 0; b < 4; b++) any |= byte_lt<SIGNED>((w[i] >> (8 * b)) & 0xffu, bound8);
    return any;
}

// The heap arrays are wave-uniform state; reading them through readfirstlane keeps
// the replay's control flow on the scalar unit.
__device__ __forceinline__ int32_t lds_i32(lds_vi32 *p)
{
    return __builtin_amdgcn_readfirstlane(*p);
}
__device__ __forceinline__ int64_t lds_i64(lds_vi64 *p)
{
    int64_t v = *p;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// insert (_fast_pq.pyx:274-307).  Called by all 64 lanes with wave-uniform
// arguments.  Only lane 0 stores: 64 lanes storing to one address serialise in the
// LDS pipe (a 64-way same-bank write), and the accesses are volatile and
// in-order per wave, so every lane's later loads see lane 0's stores.
__device__ __forceinline__ void heap_insert(lds_vi64 *hidx, lds_vi32 *hval, int R,
                                            int64_t label, int32_t v, int lane)
{
    bool dup = false;
    for (int t = lane; t < R; t += 64) dup |= (hidx[t] == label);
    if (__builtin_amdgcn_ballot_w64(dup)) return;  // :284-287
    int j = 0;
    for (;;) {
        int nxt = j;
        int32_t nxt_val = v;
        int l = 2 * j + 1, r = 2 * j + 2;
        if (l < R) {
            int32_t vl = lds_i32(&hval[l]);
            if (vl > nxt_val) { nxt = l; nxt_val = vl; }
        }
        if (r < R) {
            int32_t vr = lds_i32(&hval[r]);
            if (vr > nxt_val) { nxt = r; nxt_val = vr; }
        }
        if (nxt == j) {
            if (lane == 0) { hval[j] = v; hidx[j] = label; }
            break;
        }
        const int64_t moved = lds_i64(&hidx[nxt]);
        if (lane == 0) { hval[j] = nxt_val; hidx[j] = moved; }
        j = nxt;
    }
}

// insert_is (_fast_pq.pyx:256-271)
__device__ __forceinline__ void heap_insert_is(lds_vi64 *hidx, lds_vi32 *hval,
                                               int R, int64_t label, int32_t v, int lane)
{
    bool dup = false;
    for (int t = lane; t < R; t += 64) dup |= (hidx[t] == label);
    if (__builtin_amdgcn_ballot_w64(dup)) return;
    int j = 0;
    while (j + 1 != R) {
        int32_t nv = lds_i32(&hval[j + 1]);
        if (!(nv > v)) break;
        const int64_t moved = lds_i64(&hidx[j + 1]);
        if (lane == 0) { hidx[j] = moved; hval[j] = nv; }
        j++;
    }
    if (lane == 0) { hidx[j] = label; hval[j] = v; }
}

// Workgroup = 4 independent waves (4 queries): gfx950 admits only ~8 workgroups
// per CU, so single-wave workgroups would cap residency at 2 waves per SIMD.
#define TK_HEAP_WAVES 4

template <bool SIGNED>
__global__ __launch_bounds__(64 * TK_HEAP_WAVES) void heap_replay_kernel(
    const uint4 *__restrict__ dist, int64_t cap, const int *__restrict__ slot_prefix,
    const int *__restrict__ slot_n, const int64_t *__restrict__ slot_label_off, int S,
    const int64_t *__restrict__ labels, int64_t *__restrict__ heap_idx,
    int32_t *__restrict__ heap_val, int R, int slots_uniform,
    const unsigned char *__restrict__ only_flagged, int64_t nq,
    const uint8_t *__restrict__ mins, int64_t cap_min)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    const size_t wstride = ((size_t)R * 12 + 15) & ~(size_t)15;
    lds_vi64 *hidx = LDS_PTR(lds_vi64, smem + wave * wstride);
    lds_vi32 *hval = LDS_PTR(lds_vi32, smem + wave * wstride + (size_t)R * 8);
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (q >= nq) return;   // wave-uniform; the kernel uses no workgroup barrier

    if (only_flagged) {
        // second pass behind the lane-per-query kernel: only the queries it skipped,
        // starting from a fresh heap
        if (!only_flagged[q]) return;
        for (int t = lane; t < R; t += 64) {
            hidx[t] = -1;
            hval[t] = SIGNED ? 127 : 255;
        }
    } else {
        for (int t = lane; t < R; t += 64) {
            hidx[t] = heap_idx[q * R + t];
            hval[t] = heap_val[q * R + t];
        }
    }
    // Is the incoming array a max-heap of values in the 8-bit range?  Then the
    // bound (low 8 bits of the root) can only go down and stale votes are safe.
    bool bad = false;
    for (int t = lane; t < R; t += 64) {
        int32_t v = hval[t];
        if (SIGNED ? (v < -128 || v > 127) : (v < 0 || v > 255)) bad = true;
        if (t > 0 && hval[(t - 1) >> 1] < v) bad = true;
    }
    const bool no_skip = __builtin_amdgcn_ballot_w64(bad) != 0;
    uint32_t bound = (uint32_t)lds_i32(&hval[0]) & 0xffu;  // _fast_pq_256.pyx:73

    const int64_t qs = slots_uniform ? 0 : q;  // one descriptor row shared by all queries
    const int *prefix = slot_prefix + qs * (S + 1);
    const uint4 *drow = dist + q * cap;
    const uint4 never = SIGNED ? make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu)
                               : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    for (int s = 0; s < S; s++) {
        const int c0 = prefix[s];
        const int nchunks = prefix[s + 1] - c0;
        const int64_t n = slot_n[qs * S + s];
        const int64_t loff = slot_label_off[qs * S + s];
        const int64_t *lab = loff < 0 ? nullptr : labels + loff;
        // 64 blocks per step — or, with the scan's per-block minima and a heap whose bound can
        // only fall, 1024: lane j looks at the 16 minima of blocks [sbase + 16 j, +16) and only
        // groups that can hold a hit are entered, 16 blocks at a time (one long list against a
        // small heap, the flat DistanceTable.top: 977 steps of 1 KiB become 61 of 1 KiB of
        // minima plus the few groups that matter)
        const bool by_mins = mins != nullptr && !no_skip && (c0 & 15) == 0;
        const int width = by_mins ? 16 : 64;
        for (int sbase = 0; sbase < nchunks; sbase += by_mins ? 1024 : 64) {
          uint64_t groups = 1;
          if (by_mins) {
              const int gb = sbase + 16 * lane;
              bool gv = false;
              if (gb < nchunks) {
                  const uint4 m16 = *(const uint4 *)(mins + q * cap_min + c0 + gb);
                  gv = any_lt16<SIGNED>(m16, bound);
              }
              groups = __builtin_amdgcn_ballot_w64(gv);
          }
          while (groups) {
            const int gj = __builtin_ctzll(groups);
            groups &= groups - 1;
            const int base = by_mins ? sbase + 16 * gj : sbase;
            const int b = base + lane;
            const bool have = lane < width && b < nchunks;
            uint4 dd = never;
            if (have) dd = drow[c0 + b];
            bool vote = have && (no_skip || any_lt16<SIGNED>(dd, bound));
            uint64_t mask = __builtin_amdgcn_ballot_w64(vote);
            while (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint32_t d0 = __builtin_amdgcn_readlane(dd.x, j);
                const uint32_t d1 = __builtin_amdgcn_readlane(dd.y, j);
                const uint32_t d2 = __builtin_amdgcn_readlane(dd.z, j);
                const uint32_t d3 = __builtin_amdgcn_readlane(dd.w, j);
                // the reference's cmp_mask of this block against the live bound
                uint32_t bits = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    uint32_t w = r < 4 ? d0 : r < 8 ? d1 : r < 12 ? d2 : d3;
                    uint32_t by = (w >> (8 * (r & 3))) & 0xffu;
                    bits |= (uint32_t)byte_lt<SIGNED>(by, bound) << r;
                }
                if (!bits) continue;
                const int64_t pos0 = 16 * (int64_t)(base + j);
                while (bits) {
                    const int r = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const int64_t pos = pos0 + r;
                    if (pos < n) {  // _fast_pq_256.pyx:111
                        const int64_t label = lab ? lab[pos] : pos;
                        uint32_t w = r < 4 ? d0 : r < 8 ? d1 : r < 12 ? d2 : d3;
                        uint32_t by = (w >> (8 * (r & 3))) & 0xffu;
                        int32_t v = SIGNED ? (int32_t)(int8_t)by : (int32_t)by;
                        heap_insert(hidx, hval, R, label, v, lane);
                    }
                }
                bound = (uint32_t)lds_i32(&hval[0]) & 0xffu;  // :123
                if (!no_skip && mask) {
                    vote = vote && any_lt16<SIGNED>(dd, bound);
                    mask &= __builtin_amdgcn_ballot_w64(vote);
                }
            }
          }
        }
    }
    for (int t = lane; t < R; t += 64) {
        heap_idx[q * R + t] = hidx[t];
        heap_val[q * R + t] = hval[t];
    }
}

void tk_launch_heap_replay(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                           const int *slot_n, const int64_t *slot_label_off, int S,
                           const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R,
                           int signd, int slots_uniform, const unsigned char *only_flagged,
                           hipStream_t s, const uint8_t *mins, int64_t cap_min)
{
    if (nq == 0 || R == 0) return;
    const size_t wstride = ((size_t)R * 12 + 15) & ~(size_t)15;
    int waves = (int)(64 * 1024 / wstride);   // large heaps: fewer query-waves per workgroup
    waves = waves < 1 ? 1 : (waves > TK_HEAP_WAVES ? TK_HEAP_WAVES : waves);
    size_t lds = wstride * waves;
    dim3 grid((unsigned)((nq + waves - 1) / waves)), block(64 * waves);
    if (signd)
        hipLaunchKernelGGL(heap_replay_kernel<true>, grid, block, lds, s, dist, cap, slot_prefix,
                           slot_n, slot_label_off, S, labels, heap_idx, heap_val, R, slots_uniform,
                           only_flagged, nq, mins, cap_min);
    else
        hipLaunchKernelGGL(heap_replay_kernel<false>, grid, block, lds, s, dist, cap, slot_prefix,
                           slot_n, slot_label_off, S, labels, heap_idx, heap_val, R, slots_uniform,
                           only_flagged, nq, mins, cap_min);
}

// ---------------------------------------------------------------------------
// One query over rows far longer than its heap (`_FastDistanceTable.top` per query, fast_pq.py:297-302:
// 62 500 blocks at 1M rows against a heap of 2k+10 entries), ONE launch of ONE workgroup.
//
// What one such call costs in the general kernel above is latency, not work: ~300 inserts whose sift
// goes through LDS round trips, and a dependent global load for every group of blocks that can hold a
// row.  Here
//   * the heap (R <= 64) lives in three VGPRs across the lanes of wave 0 — entry t in lane t — and
//     `insert` (_fast_pq.pyx:274-307) runs on the scalar unit with v_readlane and a one-lane select;
//   * wave 0 replays the first `h` blocks (256 per batch of loads);
//   * the bound never rises from block to block while the array is a heap (every insert of a block is
//     below the bound captured at the block's start, _fast_pq_256.pyx:73-123), so a later block whose
//     minimum (written by the scan) is not below the bound reached after the head can never insert a
//     row: all 16 waves compact the others, in order, into `cdist` with their block numbers;
//   * wave 0 replays the compact array — same blocks in the same order as the reference's loop would
//     have entered, positions recovered from the block numbers, `pos < n` as there.
// The heap starts fresh (the caller checked): nothing is read from the caller's arrays.
struct RegHeap {
    int32_t v;        // lane t: vals[t]
    uint32_t lo, hi;  // lane t: indices[t]
};

// `insert` (_fast_pq.pyx:274-307) without its loop.  The sift goes down the heap's MAX-CHILD path —
// at node j the larger child (the left one on a tie: `vl > v`, then `vr > nxt_val`), as long as that
// child is above v — and that path does not depend on v: lane t knows from its sibling whether it is
// the larger child of its parent, one ballot makes that a mask, and a node is on the path when every
// ancestor is.  Values never rise along the path (it is a heap), so the nodes the loop would have
// passed are the root and the path nodes above v; each takes its path child's entry, the last of them
// takes (label, v).  Five cross-lane reads and ~80 vector instructions instead of five dependent
// scalar rounds.
struct RegHeapLane {   // what lane t knows about node t of a heap of R entries; fixed for the launch
    uint64_t anc;      // bits of t and its ancestors below the root
    int a_next, a_prev, odd, last, root, in_heap, has_child, l, lsh;
};
__device__ __forceinline__ RegHeapLane reg_heap_lane(int lane, int R)
{
    RegHeapLane K;
    K.anc = 0;
    for (int t = lane; t > 0; t = (t - 1) >> 1) K.anc |= 1ull << t;
    K.a_next = 4 * (lane < 63 ? lane + 1 : 63);
    K.a_prev = 4 * (lane > 0 ? lane - 1 : 0);
    K.odd = lane & 1;
    K.last = lane + 1 >= R;
    K.root = lane == 0;
    K.in_heap = lane < R;
    K.l = 2 * lane + 1;
    K.has_child = K.l < R;
    K.lsh = K.l & 63;
    return K;
}

__device__ __forceinline__ void reg_heap_insert(RegHeap &H, const RegHeapLane &K, int64_t label, int32_t v)
{
    const uint32_t llo = (uint32_t)label, lhi = (uint32_t)((uint64_t)label >> 32);
    if (__builtin_amdgcn_ballot_w64((K.in_heap & (int)(H.lo == llo) & (int)(H.hi == lhi)) != 0)) return;  // :284-287
    const int32_t next = __builtin_amdgcn_ds_bpermute(K.a_next, H.v);   // vals[t + 1]
    const int32_t prev = __builtin_amdgcn_ds_bpermute(K.a_prev, H.v);   // vals[t - 1]
    const int larger = K.odd ? (K.last | (int)(H.v >= next)) : (K.root | (int)(H.v > prev));
    const uint64_t M = __builtin_amdgcn_ballot_w64((larger & K.in_heap) != 0);
    const int onp = (~M & K.anc) == 0;                        // every node from t up to the root's child is the larger child
    const int c = K.l + 1 - (int)((M >> K.lsh) & 1);          // the larger child (a lone left child is one)
    const int ca = (c & 63) << 2;
    const int32_t cv = __builtin_amdgcn_ds_bpermute(ca, H.v);
    const uint32_t clo = (uint32_t)__builtin_amdgcn_ds_bpermute(ca, (int)H.lo);
    const uint32_t chi = (uint32_t)__builtin_amdgcn_ds_bpermute(ca, (int)H.hi);
    const int passed = onp & (K.root | (int)(H.v > v));
    const int up = passed & K.has_child & (int)(cv > v);
    H.v = up ? cv : passed ? v : H.v;
    H.lo = up ? clo : passed ? llo : H.lo;
    H.hi = up ? chi : passed ? lhi : H.hi;
}

// blocks d[0 .. nb) in order, mn[i] the minimum of block i (written by the scan); block i is block
// bmap[i] of the original array (identity without bmap)
template <bool SIGNED>
__device__ __forceinline__ void reg_heap_span(RegHeap &H, const RegHeapLane &K, uint32_t &bound, const uint4 *d, const uint8_t *mn,
                                              int nb, const int *bmap, int64_t n, int lane)
{
    for (int s0 = 0; s0 < nb; s0 += 256) {
        uint4 dd4[4];
        int bm4[4];
        uint32_t mn4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int b = s0 + 64 * i + lane;
            dd4[i] = make_uint4(0, 0, 0, 0);
            bm4[i] = b;
            mn4[i] = SIGNED ? 0x7fu : 0xffu;   // never below a bound
            if (b < nb) {
                dd4[i] = d[b];
                mn4[i] = mn[b];
                if (bmap) bm4[i] = bmap[b];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 dd = dd4[i];
            uint64_t mask = __builtin_amdgcn_ballot_w64(byte_lt<SIGNED>(mn4[i], bound));
            while (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint32_t d0 = __builtin_amdgcn_readlane(dd.x, j);
                const uint32_t d1 = __builtin_amdgcn_readlane(dd.y, j);
                const uint32_t d2 = __builtin_amdgcn_readlane(dd.z, j);
                const uint32_t d3 = __builtin_amdgcn_readlane(dd.w, j);
                const int64_t pos0 = 16 * (int64_t)__builtin_amdgcn_readlane(bm4[i], j);
                // the reference's cmp_mask of this block against the live bound: row r in lane r
                const uint32_t wr = lane < 4 ? d0 : lane < 8 ? d1 : lane < 12 ? d2 : d3;
                const uint32_t myb = (wr >> (8 * (lane & 3))) & 0xffu;
                uint32_t bits = (uint32_t)__builtin_amdgcn_ballot_w64(lane < 16 && byte_lt<SIGNED>(myb, bound));
                while (bits) {
                    const int r = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const int64_t pos = pos0 + r;
                    if (pos < n) {   // _fast_pq_256.pyx:111
                        const uint32_t w = r < 4 ? d0 : r < 8 ? d1 : r < 12 ? d2 : d3;
                        const uint32_t by = (w >> (8 * (r & 3))) & 0xffu;
                        reg_heap_insert(H, K, pos, SIGNED ? (int32_t)(int8_t)by : (int32_t)by);
                    }
                }
                bound = (uint32_t)__builtin_amdgcn_readlane(H.v, 0) & 0xffu;   // :123
                if (mask) mask &= __builtin_amdgcn_ballot_w64(byte_lt<SIGNED>(mn4[i], bound));
            }
        }
    }
}

template <bool SIGNED>
__global__ __launch_bounds__(1024) void flat_top_one_kernel(const uint4 *__restrict__ dist,
                                                            const uint8_t *__restrict__ mins, int chunks, int h,
                                                            int64_t n, int R, uint4 *__restrict__ cdist,
                                                            int *__restrict__ cblock, int64_t *__restrict__ out_idx,
                                                            int32_t *__restrict__ out_val)
{
#ifdef TK_FLAT_CLOCK
    int64_t *dbg = (int64_t *)(out_idx + 1024);
    int n_ins = 0;
#define CLK(i) if (tid == 0) { dbg[2 * (i)] = __builtin_readcyclecounter(); dbg[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); }
#else
#define CLK(i)
#endif
    __shared__ uint32_t s_bound;
    __shared__ int s_wave[16], s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RegHeap H = {SIGNED ? 127 : 255, 0xffffffffu, 0xffffffffu};   // init_heap (_fast_pq.pyx:311-315)
    uint32_t bound = SIGNED ? 127u : 255u;
    const RegHeapLane K = reg_heap_lane(lane, R);
    const int rest = chunks - h;
    const int per = ((rest + 1023) / 1024 + 15) & ~15;
    const int b_lo = h + tid * per;
    uint4 pm[4];   // minima of the thread's first 64 blocks for the compaction: in flight during the head replay
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int b0 = b_lo + 16 * g;
        pm[g] = make_uint4(0, 0, 0, 0);
        if (16 * g < per && b0 < chunks) pm[g] = *(const uint4 *)(mins + b0);
    }
    CLK(0)
    if (wave == 0) {
        reg_heap_span<SIGNED>(H, K, bound, dist, mins, h, nullptr, n, lane);
        if (lane == 0) s_bound = bound;
    }
    CLK(1)
    __syncthreads();
    const uint32_t hb = s_bound;
    uint8_t *cmin = (uint8_t *)(cblock + chunks);   // the kept blocks' minima, behind their numbers
    // compaction: thread t owns blocks [h + t per, + per), per a multiple of 16 (h is one too); the
    // minima of its first 64 were loaded before the head replay
    uint64_t keep = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t w[4] = {pm[g].x, pm[g].y, pm[g].z, pm[g].w};
#pragma unroll
        for (int k = 0; k < 16; k++)
            keep |= (uint64_t)byte_lt<SIGNED>((w[k >> 2] >> (8 * (k & 3))) & 0xffu, hb) << (16 * g + k);
    }
    {   // blocks past the thread's range or the array's end
        int valid = per < 64 ? per : 64;
        if (b_lo + valid > chunks) valid = chunks - b_lo;
        keep = valid <= 0 ? 0 : valid >= 64 ? keep : keep & ((1ull << valid) - 1ull);
    }
    int cnt = __builtin_popcountll(keep);
    for (int g = 64; g < per; g += 16) {   // more than 64 blocks per thread: above 2^16 + h blocks
        const int b0 = b_lo + g;
        if (b0 >= chunks) break;
        const uint4 m16 = *(const uint4 *)(mins + b0);
        const uint32_t w[4] = {m16.x, m16.y, m16.z, m16.w};
#pragma unroll
        for (int k = 0; k < 16; k++)
            cnt += (b0 + k < chunks) && byte_lt<SIGNED>((w[k >> 2] >> (8 * (k & 3))) & 0xffu, hb);
    }
    int inc = cnt;   // inclusive scan over the wave, then over the 16 waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int at = inc - cnt;
    for (int w = 0; w < wave; w++) at += s_wave[w];
    if (tid == 1023) s_total = at + cnt;
    while (keep) {
        const int b = b_lo + __builtin_ctzll(keep);
        keep &= keep - 1;
        cdist[at] = dist[b];
        cblock[at] = b;
        cmin[at] = mins[b];
        at++;
    }
    for (int g = 64; g < per; g += 16) {
        const int b0 = b_lo + g;
        if (b0 >= chunks) break;
        const uint4 m16 = *(const uint4 *)(mins + b0);
        const uint32_t w[4] = {m16.x, m16.y, m16.z, m16.w};
#pragma unroll
        for (int k = 0; k < 16; k++)
            if ((b0 + k < chunks) && byte_lt<SIGNED>((w[k >> 2] >> (8 * (k & 3))) & 0xffu, hb)) {
                cdist[at] = dist[b0 + k];
                cblock[at] = b0 + k;
                cmin[at] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
                at++;
            }
    }
    __threadfence();
    __syncthreads();
    if (wave != 0) return;
    CLK(2)
    reg_heap_span<SIGNED>(H, K, bound, cdist, cmin, s_total, cblock, n, lane);
    CLK(3)
#ifdef TK_FLAT_CLOCK
    if (tid == 0) dbg[8] = s_total;
#endif
    if (lane < R) {
        out_idx[lane] = (int64_t)(((uint64_t)H.hi << 32) | H.lo);
        out_val[lane] = H.v;
    }
}

void tk_launch_flat_top_one(const uint4 *dist, const uint8_t *mins, int chunks, int h, int64_t n, int R, int signd,
                            uint4 *cdist, int *cblock, int64_t *out_idx, int32_t *out_val, hipStream_t s)
{
    if (signd)
        hipLaunchKernelGGL(flat_top_one_kernel<true>, dim3(1), dim3(1024), 0, s, dist, mins, chunks, h, n, R, cdist,
                           cblock, out_idx, out_val);
    else
        hipLaunchKernelGGL(flat_top_one_kernel<false>, dim3(1), dim3(1024), 0, s, dist, mins, chunks, h, n, R, cdist,
                           cblock, out_idx, out_val);
}

// ---------------------------------------------------------------------------
// Lane-per-query replay (the throughput path of IVF.query batches).
//
// The wave-per-query kernel above spends 64 lanes on one scalar heap.  When
//   (a) every heap starts fresh (ivf.py:137-138 / init_heap),
//   (b) no label can occur twice among the lists a query scans (ids are globally
//       unique, i.e. IVF.build(n_probes=1), or positions of one list), so the
//       duplicate test of `insert` can never fire,
// each LANE can replay one query: same blocks, same stale-bound test, same
// unconditional inserts, same sift-down — 64 queries per wave in SIMT.
//   * heap entry = one dword (value8 << 24 | flat position24) in LDS, laid out
//     [slot j][lane]: a lane only touches its own column and bank = lane % 32, so
//     the divergent sift-down addresses are conflict-free by construction;
//   * the distance rows are staged 16 blocks per lane at a time: the next segment is
//     fetched into registers while the current one is replayed and dropped into
//     ST[k][lane] at the switch, so that no lane ever waits on a dependent global
//     load; inside a segment lanes are decoupled: each walks to its next block with
//     a byte below its bound, then all lanes with a pending candidate perform one
//     insert together; a bound is refreshed when its block is done;
//   * pad_fix_kernel has set the rows that pad a list's last chunk to the largest
//     value beforehand (`pos < n`, _fast_pq_256.pyx:111), so there is no row mask
//     and, with distinct labels, no slot cursor in the loop;
//   * the scan kernel also wrote each block's minimum (1 byte per block): a lane
//     tests 16 minima at once and touches only blocks that can contain a hit;
//   * the top three heap levels (nodes 0..6) are kept in registers;
//   * labels are resolved from the flat positions once, at the end.
#ifndef TK_LANES_SEG
#define TK_LANES_SEG 8      // blocks per staged segment of the lane replay's form without a duplicate test: 16 or 8
#endif
template <bool SIGNED>
__device__ __forceinline__ int entry_val(uint32_t e)
{
    return SIGNED ? ((int32_t)e >> 24) : (int)(e >> 24);
}

template <bool SIGNED>
__device__ __forceinline__ uint32_t mask_lt16(const uint4 v, uint32_t bound8)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int b = 0; b < 4; b++)
            m |= (uint32_t)byte_lt<SIGNED>((w[i] >> (8 * b)) & 0xffu, bound8) << (4 * i + b);
    return m;
}

// The same 16-bit mask by byte-parallel arithmetic (22 VALU instead of ~75: the lane replay is a
// dependent-issue chain on a wave that has its SIMD to itself, so its time IS its instruction
// count).  Per dword: bytes as unsigned (signed: top bits flipped), x < b per byte from
// t = (x | H) - (b & ~H) — no borrow crosses a byte — as  (~x7 & b7) | (~(x7 ^ b7) & ~t7)  in bit 7
// of every byte; v_dot4_u32_u8 with weights 1,2,4,8 / 16,..,128 gathers the four 0x80 flags of a
// dword into a nibble.  `bb`: the bound's byte (biased for SIGNED) in all four bytes.
template <bool SIGNED>
__device__ __forceinline__ uint32_t bound_bytes(uint32_t bound8)
{
    const uint32_t b = SIGNED ? (bound8 ^ 0x80u) : bound8;
    return __builtin_amdgcn_perm(b, b, 0u);       // byte 0 in every byte
}

template <bool SIGNED>
__device__ __forceinline__ uint32_t lt4(uint32_t x, uint32_t bb, uint32_t bl)
{
    const uint32_t H = 0x80808080u;
    const uint32_t xu = SIGNED ? (x ^ H) : x;
    const uint32_t t = (xu | H) - bl;
    return ((~xu & bb) | (~(xu ^ bb) & ~t)) & H;
}

template <bool SIGNED>
__device__ __forceinline__ uint32_t mask_lt16_swar(const uint4 v, uint32_t bb)
{
    const uint32_t bl = bb & 0x7f7f7f7fu;
    uint32_t lo = __builtin_amdgcn_udot4(lt4<SIGNED>(v.x, bb, bl), 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(lt4<SIGNED>(v.y, bb, bl), 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(lt4<SIGNED>(v.z, bb, bl), 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(lt4<SIGNED>(v.w, bb, bl), 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);
}

// `pos < n` (_fast_pq_256.pyx:111) once, ahead of the replay: the rows that pad the last chunk
// of a probed list (code of the zero vector, fast_pq.py:165) get the largest distance value —
// nothing is ever below a bound with it, which is exactly what the reference's row test
// achieves — and the chunk's minimum is recomputed.  One thread per (query, slot).
template <bool SIGNED>
__global__ void pad_fix_kernel(uint4 *__restrict__ dist, int64_t cap, int64_t nq,
                               const int *__restrict__ slot_prefix, const int *__restrict__ slot_n,
                               int S, int slots_uniform, uint8_t *__restrict__ mins, int64_t cap_min,
                               int *__restrict__ flag_list)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && flag_list) flag_list[0] = 0;      // the lane replay behind this kernel appends the queries it flags
    if (i >= nq * S) return;
    const int64_t q = i / S;
    const int sl = (int)(i - q * S);
    const int64_t qs = slots_uniform ? 0 : q;
    const int f0 = slot_prefix[qs * (S + 1) + sl], f1 = slot_prefix[qs * (S + 1) + sl + 1];
    int n = slot_n[qs * S + sl];
    n = n < 0 ? 0 : n;
    const uint32_t fill = SIGNED ? 0x7f7f7f7fu : 0xffffffffu;
    for (int c = n >> 4; c < f1 - f0; c++) {
        const int keep = n - 16 * c;                   // valid rows of this chunk: 0..15
        uint4 v = dist[q * cap + f0 + c];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int m = SIGNED ? 127 : 255;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int km = keep - 4 * j;
            km = km < 0 ? 0 : (km > 4 ? 4 : km);
            const uint32_t kmask = km >= 4 ? 0xffffffffu : ((1u << (8 * km)) - 1u);
            w[j] = (w[j] & kmask) | (fill & ~kmask);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t b = (w[j] >> (8 * t)) & 0xffu;
                const int x = SIGNED ? (int)(int8_t)b : (int)b;
                m = x < m ? x : m;
            }
        }
        dist[q * cap + f0 + c] = make_uint4(w[0], w[1], w[2], w[3]);
        mins[q * cap_min + f0 + c] = (uint8_t)m;
    }
}

// DEDUPE (labels may repeat, IVF.build(n_probes >= 2)): the low 24 bits of an entry are
// a SLOT, LAB[slot][lane] holds the slot's 32-bit label, `insert` first scans LAB for
// the candidate's label (the reference's duplicate test, _fast_pq.pyx:284-287) and a
// new entry inherits the slot of the root it evicts.  labels32 = the ids as int32.
// LAZY (rows far longer than the heap: _FastDistanceTable.top over a whole data set, 62 500 blocks per
// query at 1M rows): nothing is staged — a lane reads its row's block minima, 16 at a time and one
// segment ahead, and fetches a block from global memory only when its minimum passes the bound.
// A handful of blocks in 4 000 segments pass once the heap is warm; staging every block cost 16
// row-per-lane loads and 16 LDS writes per segment and lane (55 of the 90 ms per 10 000 queries).
// TWIN (labels may repeat AND every copy of a label carries ONE value: the lists of IVF.build(n_probes >= 2),
// ivf.py:53,77-102 — same code, same table): the duplicate test without a set of labels.  Entries stay
// POSITIONS as with distinct labels; the lane keeps f = the smallest value of any root it has evicted.  Every
// entry but the root is <= f (a root is the maximum when it leaves; a later insert above f stays at the root
// and is the next to leave).  A row that passes its block's bound (captured at block start, never rising from
// block to block) and has an EARLIER copy in the lists replayed so far: that copy passed too (same value, a
// bound no lower) and went in, or found a still earlier one in the heap.  So a copy is in the heap now
//     always                                         where v <  f  (nothing of value v has ever left),
//     iff the root is one of the earlier copies      where v >  f,
//     iff some entry is one of the earlier copies    where v == f  (the one case that scans the heap);
// and a row without an earlier copy never finds its label.  Checked at every passing row of random and
// adversarial streams against the reference's loop: tests/test_twin_dedupe_lemma.py.  Where the earlier
// copies are comes from twins.hip's table (list + offset of a stored row's other copies) and the query's
// probe list: one 4-byte load per candidate, fetched a candidate ahead; a 64-bit mask of the lists replayed
// so far answers "not probed" for most rows at once, the probe list in LDS answers exactly.  The hash set
// of the DEDUPE form (64 KB per 64 queries, its removal and insertion inside every round) is gone: 64
// queries per wave, three workgroups per CU, the LAZY form and pairs of calls apply as with distinct labels.
template <bool SIGNED, bool DEDUPE, int LW, bool LAZY = false, bool TWIN = false>
__global__ __launch_bounds__(256) void heap_replay_lanes_kernel(
    const uint4 *__restrict__ dist, int64_t cap, int64_t nq, const int *__restrict__ slot_prefix,
    const int *__restrict__ slot_n, const int64_t *__restrict__ slot_label_off, int S,
    const int64_t *__restrict__ labels, int64_t *__restrict__ heap_idx,
    int32_t *__restrict__ heap_val, int R, int slots_uniform,
    unsigned char *__restrict__ skip, int nbuf, const uint8_t *__restrict__ mins,
    int64_t cap_min, const int32_t *__restrict__ labels32, unsigned long long *__restrict__ dbg,
    int prio, int wave_lds, const int *__restrict__ plain0_arr, const int *__restrict__ qlim, const TkTwins tw,
    int *__restrict__ flag_list)
{
    static_assert(!(TWIN && DEDUPE), "one form of the duplicate test");
    // the replay is a chain of dependent LDS round trips on 157 waves; when it shares SIMDs
    // with other batches' VALU-bound scan waves, let the arbiter issue its instructions first
    if ((prio & 0xff) >= 3) __builtin_amdgcn_s_setprio(3);
    else if ((prio & 0xff) == 2) __builtin_amdgcn_s_setprio(2);
    else if ((prio & 0xff) == 1) __builtin_amdgcn_s_setprio(1);
    // LDS: H[R+2][64] heap columns (+2 sentinel rows) | DEDUPE: LAB[ceil(R/4)][64][4]
    //      labels by slot, TB[64][64] x 4 hash set of the labels | ST[16][64] staged blocks | slot table
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
    unsigned char *smem = smem_wg + (size_t)(threadIdx.x >> 6) * wave_lds;
    uint32_t *H = (uint32_t *)smem;
    const int R4 = (R + 3) >> 2;
    // LW = queries per wave: 64, or 32 (lanes 32..63 idle) where the per-lane columns of the
    // duplicate test would otherwise leave room for only ONE workgroup per CU
    constexpr size_t CB = (size_t)LW * 4;      // bytes of one row of dword columns
    uint32_t *LAB = (uint32_t *)(smem + (size_t)(R + 2) * CB);
    // DEDUPE: TB[64 buckets][64 lanes] x 4 labels: the labels in the heap as a two-choice hash set
    uint4 *TB = (uint4 *)(smem + (size_t)(R + 2) * CB + (size_t)R4 * CB * 4);
    uint4 *ST = (uint4 *)(smem + (size_t)(R + 2) * CB +
                          (DEDUPE ? (size_t)R4 * CB * 4 + (size_t)64 * CB * 4 : 0));
    // DEDUPE: slot table of the lane's query, SE[s][lane] = first flat chunk past slot s,
    // SB[s][lane] = label offset of slot s - 16 * its first flat chunk (label of row r of flat
    // chunk c in slot s = labels32[SB[s] + 16 c + r])
    // (staging rows: none when LAZY, TK_LANES_SEG blocks without a duplicate test, 16 otherwise)
    int *SE = (int *)(ST + (DEDUPE ? 16 : LAZY ? 0 : TWIN ? 16 : TK_LANES_SEG) * LW);
    int *SB = SE + (size_t)S * LW;
    // TWIN: the probed lists of the lane's query, four to a uint4: PL[t / 4][lane]
    uint4 *PL = (uint4 *)(SB + (size_t)S * LW);
    const int S4 = (S + 3) >> 2;
    // TWIN: one bit per list of the index, set for the lists replayed before slot s: BM[list / 32][lane] (where the
    // index has few enough lists: tw.bm_words > 0; a 64-bit mask of list & 63 in front of a search of PL otherwise)
    uint32_t *BM = (uint32_t *)(PL + (size_t)S4 * LW);
    const int BMW = TWIN ? tw.bm_words : 0;
#define TK_LAB(slot) LAB[(((slot) >> 2) * LW + lane) * 4 + ((slot) & 3)]
    // a workgroup = blockDim.x / 64 independent query-waves, each with its own LDS region
    const int lane = threadIdx.x & 63;
    const int64_t q = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * LW + lane;
    // `skip`: queries whose probe list may repeat a list (left to the wave kernel)
    if (LW < 64 && lane >= LW) return;   // idle half-wave: no barrier follows, ballots see exec only
    const bool valid = q < nq && !(skip && skip[q]);
    const int64_t qc = q < nq ? q : nq - 1;
    const int64_t qs = slots_uniform ? 0 : qc;
    const int *prefix = slot_prefix + qs * (S + 1);
    const uint4 *drow = dist + qc * cap;

    const uint32_t fresh_val = SIGNED ? 0x7f000000u : 0xff000000u;
    const uint32_t fresh = fresh_val | 0x00ffffffu;
    const uint32_t lowest = SIGNED ? 0x80000000u : 0u;    // a value no entry is below
    for (int j = 0; j < R; j++)
        H[j * LW + lane] = DEDUPE ? (fresh_val | (uint32_t)j) : fresh;   // node j owns slot j
    if (DEDUPE) {
        for (int g = 0; g < R4; g++)                                     // every label -1
            ((uint4 *)LAB)[g * LW + lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int b = 0; b < 64; b++) TB[b * LW + lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    // DEDUPE: labels that found both their buckets full wait in a four-entry stash (registers);
    // only a lane whose stash is full too falls back to scanning LAB — with 64 lanes x ~1000
    // inserts per wave even a 1e-4 event per insert would otherwise put every wave on the scan
    uint32_t sh0 = ~0u, sh1 = ~0u, sh2 = ~0u, sh3 = ~0u;
    bool tb_ovf = false;
    H[R * LW + lane] = H[(R + 1) * LW + lane] = lowest;   // sentinel rows: never taken
    // the top three levels (nodes 0..6) live in registers; nodes >= R are sentinels
#define TK_FRESH(j) (R > (j) ? (DEDUPE ? (fresh_val | (uint32_t)(j)) : fresh) : lowest)
    uint32_t h0 = TK_FRESH(0), h1 = TK_FRESH(1), h2 = TK_FRESH(2), h3 = TK_FRESH(3),
             h4 = TK_FRESH(4), h5 = TK_FRESH(5), h6 = TK_FRESH(6);
#undef TK_FRESH
    uint32_t bound = SIGNED ? 0x7fu : 0xffu;
    uint32_t bb = bound_bytes<SIGNED>(bound);     // the bound's byte, biased, in all four bytes (mask_lt16_swar)

    const int total = (valid && S > 0) ? prefix[S] : 0;   // flat chunks of this lane's query
    // plain_scan.hip: the blocks from flat chunk plain0 on carry clamp(plain sums), which equal the
    // reference's values below qlim[q] and are >= qlim[q] elsewhere: the replay over them is the
    // replay over the exact values iff the bound is <= qlim[q] when the first of their blocks is
    // reached.  b_plain follows the bound across the exact blocks; checked at the end.
    const int plain0 = (plain0_arr && valid) ? plain0_arr[qc] : 0x7fffffff;
    uint32_t b_plain = SIGNED ? 0x7fu : 0xffu;
    const uint4 *mrow = (const uint4 *)(mins + qc * cap_min);   // per-block minima, 16 per uint4
    // Blocks per staged segment.  The staging rows are LDS SPACE, which is what the pipelined batch runs out of
    // (DESIGN §3.6): the form without a duplicate test — every index with distinct labels, and the coarse stage of
    // all — stages 8 blocks at a time (8 KB per wave instead of 16, 32 prefetch registers instead of 64, 86 VGPRs
    // instead of 118; the minima still arrive 16 to a load, a pair of segments shares one).  Alone that replay is 9 %
    // slower (twice the segments, a prefetch eight blocks ahead); with the plain kernel on 256 instead of 512
    // workgroups beside it the batch's kernels fit the chip's LDS and the headline batch gains 4-5 % (25.4-25.8
    // against 24.4-24.7 M queries/s, same box).  The TWIN form keeps 16 (replay-bound: 15.5 against 14.9 M).
    constexpr int SEG = (LAZY || DEDUPE || TWIN) ? 16 : TK_LANES_SEG;
    int nseg = (total + SEG - 1) / SEG;
    int max_nseg = nseg;
    for (int o = LW / 2; o > 0; o >>= 1) {
        int other = __shfl_xor(max_nseg, o, 64);
        max_nseg = other > max_nseg ? other : max_nseg;
    }
    max_nseg = __builtin_amdgcn_readfirstlane(max_nseg);

    // DEDUPE / TWIN: slot cursor (monotonic) over the LDS slot table
    int s = 0;
    if (DEDUPE || TWIN) {
        for (int t = 0; t < S; t++) {
            const int e0 = prefix[t], e1 = prefix[t + 1];
            SE[t * LW + lane] = valid ? e1 : 0x7fffffff;
            SB[t * LW + lane] = (int)(slot_label_off[qs * S + t] - 16 * (int64_t)e0);
        }
    }
    // TWIN: f (see the head of the kernel), pm = bit (list & 63) of every list replayed before slot s
    int f = 0x7fffffff;
    uint64_t pm = 0;
    int se_cur = 0x7fffffff, sb_cur = 0;      // SE[s], SB[s] of the slot the cursor is in
    if (TWIN && S > 0) {
        se_cur = valid ? prefix[1] : 0x7fffffff;
        sb_cur = (int)(slot_label_off[qs * S] - 16 * (int64_t)prefix[0]);
    }
    if (TWIN) {
        const int64_t *pq = tw.probes + qc * S;
        for (int g = 0; g < S4; g++) {
            uint4 pl;
            pl.x = (uint32_t)(int)pq[4 * g];                                    // (a probe of -1 flags its query: `skip`)
            pl.y = 4 * g + 1 < S ? (uint32_t)(int)pq[4 * g + 1] : 0x7fffffffu;
            pl.z = 4 * g + 2 < S ? (uint32_t)(int)pq[4 * g + 2] : 0x7fffffffu;
            pl.w = 4 * g + 3 < S ? (uint32_t)(int)pq[4 * g + 3] : 0x7fffffffu;
            PL[g * LW + lane] = pl;
        }
        for (int g = 0; g < BMW; g++) BM[g * LW + lane] = 0;
    }

    // Segment g + 1 (16 blocks per lane, each lane from its own row, + its 16 block minima) is
    // fetched into REGISTERS while segment g is replayed and dropped into the LDS staging rows
    // ST[k][lane] — from where the replay picks blocks by a run-time index — when segment g is
    // done.  (Round 1 staged by LDS-DMA, global_load_lds_dwordx4: the compiler must put
    // s_waitcnt vmcnt(0) in front of every LDS read that follows such an operation — it cannot
    // tell that the read does not alias the DMA's target — so the "next" segment was always
    // waited for before the current one was touched.  Fetching only the blocks whose minimum
    // passes the previous bound, predicated per lane, was measured slower: the exec-mask
    // handling costs more than the addresses it saves; profiles/r02_replay_phases.md.)
    // Blocks past the lane's row are clamped to its last valid address and never looked at.
    const int last_blk = (int)(cap > 0 ? cap - 1 : 0);
    const int last_m = (int)(cap_min / 16) - 1;
    uint4 nx[SEG];
    uint4 mins_nx = make_uint4(0, 0, 0, 0);
#define TK_MINS_ROW16(g_) mrow[(g_) < last_m ? (g_) : (last_m > 0 ? last_m : 0)]
#define TK_MINS_ROW(g_) TK_MINS_ROW16(SEG == 16 ? (g_) : (g_) >> 1)
#define TK_FETCH_BLOCKS(g_)                                                       \
    {                                                                             \
        _Pragma("unroll") for (int k = 0; k < SEG; k++) {                         \
            int blk_ = SEG * (g_) + k;                                            \
            blk_ = blk_ < last_blk ? blk_ : last_blk;                             \
            nx[k] = drow[blk_];                                                   \
        }                                                                         \
    }
#pragma unroll
    for (int k = 0; k < SEG; k++) nx[k] = make_uint4(0, 0, 0, 0);
    // LAZY: the minima run TWO segments ahead, so that at the start of segment g those of g + 1 are in registers
    // and the FIRST block of g + 1 whose minimum passes the bound of now — a superset of what will pass then —
    // can be requested while segment g is replayed: on long lists ~1.3 blocks of a segment with any hit pass
    // (500 of 6 250 per query at 100M x 128), each was a dependent trip to memory of its own.
    uint4 mins_n2 = make_uint4(0, 0, 0, 0), pre = make_uint4(0, 0, 0, 0), pre_n = make_uint4(0, 0, 0, 0);
    int pre_blk = -1, pre_n_blk = -1;
    if (max_nseg > 0) {
        mins_nx = TK_MINS_ROW(0);
        if (LAZY && max_nseg > 1) mins_n2 = TK_MINS_ROW(1);
        if (!LAZY) TK_FETCH_BLOCKS(0)
    }
    int rounds = 0;               // wave-uniform: iterations of the insert loop (each a dependent chain: roofline.replay)
    for (int g = 0; g < max_nseg; g++) {
        if (!LAZY) {
#pragma unroll
            for (int k = 0; k < SEG; k++) ST[k * LW + lane] = nx[k];
        }
        const uint4 mins_cur = mins_nx;
        if (LAZY) {
            mins_nx = mins_n2;                              // segment g + 1: requested an iteration ago
            if (g + 2 < max_nseg) mins_n2 = TK_MINS_ROW(g + 2);
            pre = pre_n;
            pre_blk = pre_n_blk;
            pre_n_blk = -1;
            if (g + 1 < max_nseg) {
                int kn = total - 16 * (g + 1);
                kn = kn < 0 ? 0 : (kn > 16 ? 16 : kn);
                uint32_t hn = mask_lt16_swar<SIGNED>(mins_nx, bb);
                hn &= kn >= 16 ? 0xffffu : ((1u << kn) - 1u);
                if (hn) {
                    pre_n_blk = 16 * (g + 1) + __builtin_ctz(hn);
                    pre_n = drow[pre_n_blk < last_blk ? pre_n_blk : last_blk];
                }
            }
        } else if (g + 1 < max_nseg) {
            mins_nx = TK_MINS_ROW(g + 1);
            TK_FETCH_BLOCKS(g + 1)
        }
        const int buf = 0;
        int kmax = total - SEG * g;
        kmax = kmax < 0 ? 0 : (kmax > SEG ? SEG : kmax);
        // blocks whose minimum is below the bound at segment start: a superset of the
        // blocks the reference enters (the bound only decreases)
        uint32_t hit = mask_lt16_swar<SIGNED>(mins_cur, bb);
        if (SEG == 8) hit = (hit >> (8 * (g & 1))) & 0xffu;       // (this segment's half of the 16 minima)
        hit &= kmax >= SEG ? ((1u << SEG) - 1u) : ((1u << kmax) - 1u);
        uint32_t bits = 0;
        uint4 dd = make_uint4(0, 0, 0, 0);
        int cur = 0;
        uint32_t lab_next = 0;        // DEDUPE: label of the lowest pending row, fetched ahead
        int lab_base = 0;             //         labels32 index of row 0 of the current block
        int tw_next = -1;             // TWIN: list of the lowest pending row's first other copy, fetched ahead
        // next block of this segment with a byte below the live bound (a macro, not a lambda: inline at the head of the
        // round loop the compiler folds this loop into it; as a lambda it stayed a nested loop of its own and the form
        // without a duplicate test lost 7 % alone).  `pos < n` (:111): the rows that pad a list's last chunk were set to the
        // largest value by pad_fix_kernel and can never be below a bound — no row count, and with distinct labels no slot
        // cursor at all, is needed; cmp_mask: _fast_pq_256.pyx:81-90; DEDUPE / TWIN: the slot cursor steps over empty lists,
        // TWIN: the list left joins the set, rows of the first probed list have no earlier copy (nothing to ask)
#define TK_ADVANCE() \
            while (bits == 0 && hit) { \
                const int k = __builtin_ctz(hit); \
                hit &= hit - 1; \
                cur = SEG * g + k; \
                if (LAZY) { \
                    if (cur == pre_blk) dd = pre; \
                    else dd = drow[cur < last_blk ? cur : last_blk]; \
                } else { \
                    dd = ST[(buf * 16 + k) * LW + lane]; \
                } \
                bits = mask_lt16_swar<SIGNED>(dd, bb); \
                if (DEDUPE && bits) { \
                    while (cur >= SE[s * LW + lane]) s++; \
                    lab_base = SB[s * LW + lane] + 16 * cur; \
                    lab_next = (uint32_t)labels32[(int64_t)lab_base + __builtin_ctz(bits)]; \
                } \
                if (TWIN && bits) { \
                    while (cur >= se_cur) { \
                        const int cs = ((const int *)PL)[(((s >> 2) * LW + lane) << 2) + (s & 3)]; \
                        if (BMW) BM[(cs >> 5) * LW + lane] |= 1u << (cs & 31); \
                        else pm |= 1ull << (cs & 63); \
                        s++; \
                        se_cur = SE[s * LW + lane]; \
                        sb_cur = SB[s * LW + lane]; \
                    } \
                    lab_base = sb_cur + 16 * cur; \
                    if (s > 0) tw_next = tw.list[(lab_base + __builtin_ctz(bits)) * tw.w]; \
                } \
            }
        // One round: every lane without a pending candidate looks for its next block (`advance`, one call site); the
        // lanes with one take it through the duplicate test and `insert` — register levels, then LDS levels.  TWIN
        // with staged blocks (EARLY) looks between the two: the new root — the bound behind the block — is known
        // after the register levels, the block is in LDS, and the request for the next candidate's row of the twin
        // table is then in flight while the sift goes through the LDS levels (a trip to memory in every round
        // otherwise: 1.7 us per round against 1.0 of the form without the test; 1.35 this way).  Requesting a LAZY
        // form's next passing block ahead of the insert in the same manner was measured and is not kept (100M x 128:
        // 6.27 M queries/s with, 6.27 without, same box).
        constexpr bool EARLY = TWIN && !LAZY;
        // `insert` below node j (3..6): children in LDS, branch-free per level — rows R and R+1 hold a value no entry
        // exceeds, so children beyond the heap (clamped to R) are never taken
        auto lds_levels = [&](int j, const uint32_t entry, const int v) {
            bool first = true, go = true;
            do {
                const int l = 2 * j + 1;
                const int lc = l < R ? l : R;
                const uint32_t el = H[lc * LW + lane];
                const uint32_t er = H[(lc + 1) * LW + lane];
                const int vl = entry_val<SIGNED>(el), vr = entry_val<SIGNED>(er);
                const bool cl = vl > v;                 // vals[l] > nxt_val
                const int nvv = cl ? vl : v;
                uint32_t ne = cl ? el : entry;
                int nxt = cl ? l : j;
                const bool cr = vr > nvv;               // vals[r] > nxt_val
                ne = cr ? er : ne;
                nxt = cr ? l + 1 : nxt;
                if (first) {
                    h3 = j == 3 ? ne : h3; h4 = j == 4 ? ne : h4;
                    h5 = j == 5 ? ne : h5; h6 = j == 6 ? ne : h6;
                    first = false;
                } else {
                    H[j * LW + lane] = ne;              // entry itself when nxt == j
                }
                go = nxt != j;
                j = nxt;
            } while (go);
        };
        for (;;) {
            if (!EARLY) {
                TK_ADVANCE()
                if (__builtin_amdgcn_ballot_w64(bits != 0) == 0) break;
                rounds++;
            }
            int j = 0;             // > 0: the sift goes on below node j (3..6), whose children are in LDS
            uint32_t entry = 0;
            int v = 0;
            if (bits) {   // one insert per lane with a pending candidate
                const int r = __builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t w = r < 4 ? dd.x : r < 8 ? dd.y : r < 12 ? dd.z : dd.w;
                const uint32_t by = (w >> (8 * (r & 3))) & 0xffu;
                uint32_t low = (uint32_t)(16 * cur + r);
                bool dup = false;
                if (DEDUPE) {
                    const uint32_t label = lab_next;
                    if (bits) lab_next = (uint32_t)labels32[(int64_t)lab_base + __builtin_ctz(bits)];
                    // `if i == indices[j]: return` over every slot, _fast_pq.pyx:284-287, answered by
                    // a hash set of the labels in the heap instead of a scan of all R slots (with
                    // any cheaper pre-filter some lane of 64 passes it in every round, and the
                    // scan then runs for the whole wave: it was 1.0 of the kernel's 2.2 ms,
                    // profiles/r02_replay_phases.md).  Two candidate buckets of four labels per
                    // label, the emptier one takes it; a label that finds both full is only kept in
                    // LAB and switches its lane to the full scan (exactness never depends on the
                    // set: LAB always holds every slot's label).
                    const uint32_t hh = label * 0x9E3779B1u;
                    const int b1 = (int)(hh >> 26);
                    int b2 = (int)((hh >> 18) & 63u);
                    b2 = b2 == b1 ? (b1 ^ 1) : b2;
                    const uint4 x1 = TB[b1 * LW + lane], x2 = TB[b2 * LW + lane];
                    dup = (x1.x == label) | (x1.y == label) | (x1.z == label) | (x1.w == label) |
                          (x2.x == label) | (x2.y == label) | (x2.z == label) | (x2.w == label) |
                          (sh0 == label) | (sh1 == label) | (sh2 == label) | (sh3 == label);
                    if (tb_ovf) {
#pragma unroll 4
                        for (int g = 0; g < R4; g++) {
                            const uint4 lv = ((const uint4 *)LAB)[g * LW + lane];
                            dup |= (lv.x == label) | (lv.y == label) | (lv.z == label) | (lv.w == label);
                        }
                    }
                    low = h0 & 0x00ffffffu;            // the evicted root's slot
                    const uint32_t gone = TK_LAB(low); // label leaving with the root
                    if (!dup) {
                        if (gone != 0xffffffffu) {     // out of its bucket (if it ever got into one)
                            const uint32_t gh = gone * 0x9E3779B1u;
                            const int g1 = (int)(gh >> 26);
                            int g2 = (int)((gh >> 18) & 63u);
                            g2 = g2 == g1 ? (g1 ^ 1) : g2;
                            const uint4 y1 = TB[g1 * LW + lane], y2 = TB[g2 * LW + lane];
                            const int p1 = y1.x == gone ? 0 : y1.y == gone ? 1 : y1.z == gone ? 2 : y1.w == gone ? 3 : -1;
                            const int p2 = y2.x == gone ? 0 : y2.y == gone ? 1 : y2.z == gone ? 2 : y2.w == gone ? 3 : -1;
                            if (p1 >= 0) ((uint32_t *)TB)[(g1 * LW + lane) * 4 + p1] = 0xffffffffu;
                            else if (p2 >= 0) ((uint32_t *)TB)[(g2 * LW + lane) * 4 + p2] = 0xffffffffu;
                            else {
                                sh0 = sh0 == gone ? ~0u : sh0; sh1 = sh1 == gone ? ~0u : sh1;
                                sh2 = sh2 == gone ? ~0u : sh2; sh3 = sh3 == gone ? ~0u : sh3;
                            }
                        }
                        // x1 / x2 were read before that removal: an entry it freed still looks taken,
                        // which can only cost an unnecessary overflow, never a wrong placement
                        const int e1 = (x1.x == ~0u) + (x1.y == ~0u) + (x1.z == ~0u) + (x1.w == ~0u);
                        const int e2 = (x2.x == ~0u) + (x2.y == ~0u) + (x2.z == ~0u) + (x2.w == ~0u);
                        const bool first = e1 >= e2;
                        const uint4 xs = first ? x1 : x2;
                        const int pe = xs.x == ~0u ? 0 : xs.y == ~0u ? 1 : xs.z == ~0u ? 2 : xs.w == ~0u ? 3 : -1;
                        if (pe >= 0) ((uint32_t *)TB)[((first ? b1 : b2) * LW + lane) * 4 + pe] = label;
                        else if (sh0 == ~0u) sh0 = label;
                        else if (sh1 == ~0u) sh1 = label;
                        else if (sh2 == ~0u) sh2 = label;
                        else if (sh3 == ~0u) sh3 = label;
                        else tb_ovf = true;
                        TK_LAB(low) = label;
                    }
                }
                entry = (by << 24) | low;
                v = entry_val<SIGNED>(entry);
                if (TWIN && s > 0) {
                    const int row = lab_base + r;                   // the candidate's row in the twin table
                    int c = tw_next;
                    if (bits) tw_next = tw.list[(lab_base + __builtin_ctz(bits)) * tw.w];
#pragma nounroll
                    for (int u = 0; u < tw.w && !dup; u++) {
                        if (u > 0) c = tw.list[row * tw.w + u];
                        if (c < 0) break;                           // no further copies
                        // was list c replayed before this one?
                        if (BMW) {
                            if (!((BM[(c >> 5) * LW + lane] >> (c & 31)) & 1u)) continue;
                        } else if (!((pm >> (c & 63)) & 1ull)) continue;
                        if (BMW && v < f) { dup = true; break; }    // nothing of this value has ever left
                        int t = -1;                                 // its slot (the mask: exactly; the bitmap: for the position)
#pragma nounroll
                        for (int g = 0; g < S4; g++) {
                            const uint4 pl = PL[g * LW + lane];
                            t = ((int)pl.x == c && 4 * g < s) ? 4 * g : t;
                            t = ((int)pl.y == c && 4 * g + 1 < s) ? 4 * g + 1 : t;
                            t = ((int)pl.z == c && 4 * g + 2 < s) ? 4 * g + 2 : t;
                            t = ((int)pl.w == c && 4 * g + 3 < s) ? 4 * g + 3 : t;
                        }
                        if (t < 0) continue;
                        if (v < f) { dup = true; break; }           // nothing of this value has ever left
                        // the earlier copy as an entry: same value, its position in this query's rows
                        const uint32_t E = (by << 24) | (uint32_t)(16 * prefix[t] + tw.off[row * tw.w + u]);
                        if (v > f) {
                            dup = h0 == E;                          // only the root can be above f
                        } else {
                            bool found = (h0 == E) | (h1 == E) | (h2 == E) | (h3 == E) | (h4 == E) | (h5 == E) | (h6 == E);
#pragma nounroll
                            for (int jj = 7; jj < R; jj++) found |= H[jj * LW + lane] == E;
                            dup = found;
                        }
                    }
                }
                if (TWIN && !dup) {
                    const int root = entry_val<SIGNED>(h0);
                    f = root < f ? root : f;
                }
                // insert, _fast_pq.pyx:291-307.  Levels 0-2 in registers, the rest in
                // LDS, branch-free per level: rows R and R+1 hold a value no entry
                // exceeds, so children beyond the heap (clamped to R) are never taken.
                if (!dup)
                {   // node 0, children 1 and 2
                    const int v1 = entry_val<SIGNED>(h1), v2 = entry_val<SIGNED>(h2);
                    const bool c1 = v1 > v;
                    const int nv = c1 ? v1 : v;
                    const bool c2 = v2 > nv;
                    h0 = c2 ? h2 : (c1 ? h1 : entry);
                    if (c1 | c2) {   // node 1 or 2, children (3,4) or (5,6)
                        const uint32_t a = c2 ? h5 : h3, b = c2 ? h6 : h4;
                        const int va = entry_val<SIGNED>(a), vb = entry_val<SIGNED>(b);
                        const bool ca = va > v;
                        const int nv1 = ca ? va : v;
                        const bool cb = vb > nv1;
                        const uint32_t ne1 = cb ? b : (ca ? a : entry);
                        if (c2) h2 = ne1; else h1 = ne1;
                        if (ca | cb) {
                            j = (c2 ? 5 : 3) + (cb ? 1 : 0);
                            if (!EARLY) lds_levels(j, entry, v);        // (EARLY: behind the look-ahead below)
                        }
                    }
                }
                // refresh after the block, :123 (EARLY: the new root is known behind the register levels — the bound
                // does not wait for the levels below)
                if (bits == 0) {
                    bound = h0 >> 24;
                    bb = bound_bytes<SIGNED>(bound);
                    b_plain = cur < plain0 ? bound : b_plain;
                }
            }
            if (EARLY) {
                TK_ADVANCE()
                if (j) lds_levels(j, entry, v);
                if (__builtin_amdgcn_ballot_w64(bits != 0) == 0) break;
                rounds++;
            }
        }
    }
#undef TK_ADVANCE
#undef TK_FETCH_BLOCKS
#undef TK_MINS_ROW
#undef TK_MINS_ROW16
    if (dbg && lane == 0) {       // TK_OPT_REPLAY_COUNT: [0] += rounds, [1] = max rounds of a wave, [2] += waves, [3] += segments
        atomicAdd(&dbg[0], (unsigned long long)rounds);
        atomicMax(&dbg[1], (unsigned long long)rounds);
        atomicAdd(&dbg[2], 1ull);
        atomicAdd(&dbg[3], (unsigned long long)max_nseg);
    }
    // registers back to their heap rows
    if (R > 0) H[0 * LW + lane] = h0;
    if (R > 1) H[1 * LW + lane] = h1;
    if (R > 2) H[2 * LW + lane] = h2;
    if (R > 3) H[3 * LW + lane] = h3;
    if (R > 4) H[4 * LW + lane] = h4;
    if (R > 5) H[5 * LW + lane] = h5;
    if (R > 6) H[6 * LW + lane] = h6;
    // flag_list: the queries left to the kernels behind this one — flagged before it (probe lists that name a list
    // twice) or by the check below — [0] = their count (zeroed by pad_fix_kernel), then their numbers in any order:
    // what a separate one-workgroup kernel over all nq flags used to compile (30 us of every batch's replay stream)
    if (!valid) {
        if (flag_list && q < nq) flag_list[1 + atomicAdd(&flag_list[0], 1)] = (int)q;
        return;
    }
    // re-scan + replay again: flag 2 with distinct labels (no duplicate test needed then) and from the TWIN form, 1 otherwise
    if (plain0_arr && plain0 < total && (int)(int8_t)b_plain > qlim[qc]) {
        skip[q] = DEDUPE ? 1 : 2;
        if (flag_list) flag_list[1 + atomicAdd(&flag_list[0], 1)] = (int)q;
    }
    // ---- resolve flat positions (or slots) to labels
    const int64_t *loffs = slot_label_off + qs * S;
    for (int j = 0; j < R; j++) {
        const uint32_t e = H[j * LW + lane];
        const uint32_t pos = e & 0x00ffffffu;
        int64_t label = -1;
        if (DEDUPE) {
            label = (int64_t)(int32_t)TK_LAB(pos);
        } else if (pos != 0x00ffffffu) {
            const int f = (int)(pos >> 4);
            int lo = 0, hi = S;  // prefix[lo] <= f < prefix[hi]
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (prefix[mid] <= f) lo = mid; else hi = mid;
            }
            const int64_t inlist = (int64_t)pos - 16 * (int64_t)prefix[lo];
            const int64_t loff = loffs[lo];
            label = loff < 0 ? inlist : labels[loff + inlist];
        }
        heap_idx[q * R + j] = label;
        heap_val[q * R + j] = entry_val<SIGNED>(e);
    }
}

// ---------------------------------------------------------------------------
// Wave-per-query replay on packed entries: same preconditions as the
// lane-per-query kernel (fresh heap, labels cannot repeat), same entry format
// (value8 << 24 | flat position24), but the 64 lanes of a wave cooperate on ONE
// query: coalesced 1 KiB loads of 64 blocks, a ballot vote, the exact per-block
// mask from lanes 0..15, and a scalar-controlled sift-down over a heap that is R
// dwords of LDS (children l, l+1 are adjacent dwords).  Far fewer instructions per
// insert than heap_replay_kernel (no label array, no duplicate scan, one LDS word
// per node), which is what bounds that kernel at 10^4 concurrent queries.
// DEDUPE: labels may repeat among a query's lists (IVF.build(n_probes >= 2), or a
// wrapped probe id).  The low 24 bits of an entry are then a SLOT number instead of
// a position; lab[slot] holds the int64 label, `insert` first runs the reference's
// duplicate test over lab[] with all 64 lanes, and a new entry takes over the slot
// of the root it evicts.  `flags`/`run_if`: process query q iff flags[q] == run_if.
template <bool SIGNED, bool DEDUPE>
__global__ __launch_bounds__(64 * TK_HEAP_WAVES) void heap_replay_packed_kernel(
    const uint4 *__restrict__ dist, int64_t cap, const int *__restrict__ slot_prefix,
    const int *__restrict__ slot_n, const int64_t *__restrict__ slot_label_off, int S,
    const int64_t *__restrict__ labels, int64_t *__restrict__ heap_idx,
    int32_t *__restrict__ heap_val, int R, int slots_uniform,
    const unsigned char *__restrict__ flags, int run_if, int64_t nq, const int *__restrict__ flag_list,
    volatile int *host_count)
{
    // (the count of the flagged queries to the page-locked word the host polls: plain_scan's state machine)
    if (host_count && flag_list && blockIdx.x == 0 && threadIdx.x == 0) *host_count = flag_list[0];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    const size_t wstride = DEDUPE ? (((size_t)R * 12 + 15) & ~(size_t)15) : (size_t)R * 4;
    lds_vi64 *lab = LDS_PTR(lds_vi64, smem + wave * wstride);                       // [R] (DEDUPE)
    lds_vu32 *H = LDS_PTR(lds_vu32, smem + wave * wstride + (DEDUPE ? (size_t)R * 8 : 0));  // [R]
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (q >= nq) return;   // wave-uniform; no workgroup barrier below
    if (flags && (run_if < 0 ? flags[q] == 0 : (int)flags[q] != run_if)) return;      // run_if < 0: every flagged query
    const int64_t qs = slots_uniform ? 0 : q;
    const int *prefix = slot_prefix + qs * (S + 1);
    const uint4 *drow = dist + q * cap;

    const uint32_t fresh_val = SIGNED ? 0x7f000000u : 0xff000000u;
    for (int t = lane; t < R; t += 64) {
        H[t] = fresh_val | (DEDUPE ? (uint32_t)t : 0x00ffffffu);
        if (DEDUPE) lab[t] = -1;
    }
    uint32_t bound = SIGNED ? 0x7fu : 0xffu;
    const uint4 never = SIGNED ? make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu)
                               : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    for (int s = 0; s < S; s++) {
        const int c0 = prefix[s];
        const int nchunks = prefix[s + 1] - c0;
        const int n = slot_n[qs * S + s];
        const int64_t loff = slot_label_off[qs * S + s];
        const int64_t *labp = loff < 0 ? nullptr : labels + loff;
        for (int base = 0; base < nchunks; base += 64) {
            const int b = base + lane;
            const bool have = b < nchunks;
            uint4 dd = never;
            if (have) dd = drow[c0 + b];
            bool vote = have && any_lt16<SIGNED>(dd, bound);
            uint64_t mask = __builtin_amdgcn_ballot_w64(vote);
            // DEDUPE: the 16 labels of a voted block are fetched by lanes 0..15 with one
            // vector load, one block ahead of the block being replayed
            auto load_labels = [&](int jb) -> int64_t {
                const int64_t inl = 16 * (int64_t)(base + jb) + lane;
                if (lane >= 16 || inl >= n) return -2;   // never a label
                return labp ? labp[inl] : inl;
            };
            int j_pref = -1;
            int64_t lab_pref = -2;
            if (DEDUPE && mask) {
                j_pref = __builtin_ctzll(mask);
                lab_pref = load_labels(j_pref);
            }
            while (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1;
                int64_t lab_cur = -2;
                if (DEDUPE) {
                    lab_cur = (j == j_pref) ? lab_pref : load_labels(j);
                    if (mask) {
                        j_pref = __builtin_ctzll(mask);
                        lab_pref = load_labels(j_pref);
                    }
                }
                const uint32_t d0 = __builtin_amdgcn_readlane(dd.x, j);
                const uint32_t d1 = __builtin_amdgcn_readlane(dd.y, j);
                const uint32_t d2 = __builtin_amdgcn_readlane(dd.z, j);
                const uint32_t d3 = __builtin_amdgcn_readlane(dd.w, j);
                // lane r < 16 tests row r against the live bound (= block-start bound)
                const uint32_t w = lane < 4 ? d0 : lane < 8 ? d1 : lane < 12 ? d2 : d3;
                const uint32_t by = (w >> (8 * (lane & 3))) & 0xffu;
                const int rows = n - 16 * (base + j);   // `pos < n`
                const bool lt = lane < 16 && lane < rows && byte_lt<SIGNED>(by, bound);
                uint32_t bits = (uint32_t)__builtin_amdgcn_ballot_w64(lt);
                if (!bits) continue;
                const uint32_t pos0 = (uint32_t)(16 * (c0 + base + j));
                while (bits) {
                    const int r = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const uint32_t byr = __builtin_amdgcn_readlane(by, r);
                    uint32_t low = pos0 + (uint32_t)r;
                    if (DEDUPE) {
                        const uint32_t llo = __builtin_amdgcn_readlane((uint32_t)lab_cur, r);
                        const uint32_t lhi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)lab_cur >> 32), r);
                        const int64_t label = (int64_t)(((uint64_t)lhi << 32) | llo);
                        bool dup = false;   // `if i == indices[j]: return`, _fast_pq.pyx:284-287
                        for (int t = lane; t < R; t += 64) dup |= (lab[t] == label);
                        if (__builtin_amdgcn_ballot_w64(dup)) continue;
                        // the new entry inherits the slot of the root it replaces
                        low = __builtin_amdgcn_readfirstlane(H[0]) & 0x00ffffffu;
                        if (lane == 0) lab[low] = label;
                    }
                    const uint32_t entry = (byr << 24) | low;
                    const int v = entry_val<SIGNED>(entry);
                    int jn = 0;
                    for (;;) {  // insert, _fast_pq.pyx:291-307
                        const int l = 2 * jn + 1;
                        if (l >= R) { if (lane == 0) H[jn] = entry; break; }
                        uint32_t el = H[l];
                        uint32_t er = (l + 1 < R) ? H[l + 1] : 0u;
                        el = __builtin_amdgcn_readfirstlane(el);
                        er = __builtin_amdgcn_readfirstlane(er);
                        int nxt = jn, nv = v;
                        uint32_t ne = entry;
                        if (entry_val<SIGNED>(el) > nv) { nxt = l; nv = entry_val<SIGNED>(el); ne = el; }
                        if (l + 1 < R && entry_val<SIGNED>(er) > nv) { nxt = l + 1; ne = er; }
                        if (nxt == jn) { if (lane == 0) H[jn] = entry; break; }
                        if (lane == 0) H[jn] = ne;  // one lane: same-address stores serialise
                        jn = nxt;
                    }
                }
                bound = __builtin_amdgcn_readfirstlane(H[0]) >> 24;  // :123
                if (mask) {
                    vote = vote && any_lt16<SIGNED>(dd, bound);
                    mask &= __builtin_amdgcn_ballot_w64(vote);
                }
            }
        }
    }
    // heap arrays out
    const int64_t *loffs = slot_label_off + qs * S;
    for (int t = lane; t < R; t += 64) {
        const uint32_t e = H[t];
        const uint32_t pos = e & 0x00ffffffu;
        int64_t label = -1;
        if (DEDUPE) {
            label = lab[pos];
        } else if (pos != 0x00ffffffu) {   // resolve the flat position
            const int f = (int)(pos >> 4);
            int lo = 0, hi = S;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (prefix[mid] <= f) lo = mid; else hi = mid;
            }
            const int64_t inlist = (int64_t)pos - 16 * (int64_t)prefix[lo];
            const int64_t loff = loffs[lo];
            label = loff < 0 ? inlist : labels[loff + inlist];
        }
        heap_idx[q * R + t] = label;
        heap_val[q * R + t] = entry_val<SIGNED>(e);
    }
}

void tk_launch_heap_replay_packed(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                                  const int *slot_n, const int64_t *slot_label_off, int S,
                                  const int64_t *labels, int64_t *heap_idx, int32_t *heap_val,
                                  int R, int signd, int slots_uniform, const unsigned char *flags,
                                  int run_if, int dedupe, hipStream_t s, const int *flag_list, int *host_count)
{
    if (nq == 0 || R == 0) return;
    const size_t wstride = dedupe ? (((size_t)R * 12 + 15) & ~(size_t)15) : (size_t)R * 4;
    int waves = (int)(64 * 1024 / wstride);
    waves = waves < 1 ? 1 : (waves > TK_HEAP_WAVES ? TK_HEAP_WAVES : waves);
    size_t lds = wstride * waves;
    dim3 grid((unsigned)((nq + waves - 1) / waves)), block(64 * waves);
#define TK_LAUNCH(S_, D_)                                                                        \
    hipLaunchKernelGGL((heap_replay_packed_kernel<S_, D_>), grid, block, lds, s, dist, cap,      \
                       slot_prefix, slot_n, slot_label_off, S, labels, heap_idx, heap_val, R,    \
                       slots_uniform, flags, run_if, nq, flag_list, host_count)
    if (signd) { if (dedupe) TK_LAUNCH(true, true); else TK_LAUNCH(true, false); }
    else { if (dedupe) TK_LAUNCH(false, true); else TK_LAUNCH(false, false); }
#undef TK_LAUNCH
}

#undef TK_LAB

// LDS of one query-wave of the lane kernel without the staged segment: heap columns, and with
// the duplicate test the label slots, the label-hash counters and the slot table
static size_t tk_lanes_fixed_lds(int R, int S, int dedupe)
{
    return (size_t)(R + 2) * 256 + (dedupe ? (size_t)((R + 3) / 4) * 1024 + 65536 + (size_t)S * 512 : 0);
}
// ... of the TWIN form: heap columns, slot table, probe list
static size_t tk_lanes_twin_lds(int R, int S, int bm_words)
{
    return (size_t)(R + 2) * 256 + (size_t)S * 512 + (size_t)((S + 3) / 4) * 1024 + (size_t)bm_words * 256;
}
// one bit per list and query where that is at most 32 KB per wave of 64 queries (4096 lists)
int tk_lanes_twin_bm_words(int64_t n_lists) { return n_lists <= 4096 ? (int)((n_lists + 31) / 32) : 0; }
int tk_lanes_twin_fits(int R, int S, int64_t n_lists)
{
    return R <= TK_LANES_MAX_R && tk_lanes_twin_lds(R, S, tk_lanes_twin_bm_words(n_lists)) + 16384 <= 160 * 1024;
}

int tk_lanes_dedupe_fits(int R, int S)
{
    return R <= TK_LANES_MAX_R_DEDUPE && tk_lanes_fixed_lds(R, S, 1) + 16384 <= 160 * 1024;
}

int tk_launch_heap_replay_lanes(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                                const int *slot_n, const int64_t *slot_label_off, int S,
                                const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R,
                                int signd, int slots_uniform, unsigned char *skip,
                                const uint8_t *mins, int64_t cap_min, const int32_t *labels32,
                                hipStream_t s, const int *plain0, const int *qlim, int lazy,
                                unsigned long long *counters, const TkTwins *twins, int *flag_list)
{
    if (nq == 0 || R == 0) return 0;
    if (!plain0 || !qlim || !skip || !signd) plain0 = qlim = nullptr;
    const int dedupe = labels32 != nullptr;
    const bool twin = !dedupe && twins && twins->w > 0 && twins->list && twins->off && twins->probes && !slots_uniform;
    const TkTwins tw = twin ? *twins : TkTwins();
    // Queries per wave: 64, or 32 with the duplicate test — a 64-query wave then needs 140+ KB of
    // LDS at R = 111: one workgroup per CU, and the two replay kernels of the pipelined mode (2 x 157
    // workgroups on 256 CUs) waited for each other's CUs (1.6 ms alone became 2.9 ms in the
    // pipeline); half-filled waves (lanes 32..63 exit at once) halve every per-lane column: two
    // workgroups per CU.  (Measured and dropped, profiles/HISTORY.md: 32- and 16-query waves for
    // distinct labels and for the coarse replay, 2-4 query-waves per workgroup, blocks fetched only
    // where their minimum passes, s_setprio below 3.)
    const int LWr = dedupe ? 32 : 64;
    // heap columns (+ label slots) + one staged segment (16 blocks x LW lanes x 16 B; the next one
    // waits in registers), scaled to the columns in use
    // (staging rows: none when LAZY, TK_LANES_SEG blocks without a duplicate test, 16 otherwise)
    const size_t st_rows = dedupe ? (size_t)16384 : lazy ? 0 : twin ? (size_t)16384 : (size_t)1024 * TK_LANES_SEG;
    const size_t lds = twin ? tk_lanes_twin_lds(R, S, tw.bm_words) + st_rows
                            : tk_lanes_fixed_lds(R, S, dedupe) * LWr / 64 + st_rows * LWr / 64;
    static bool attr_set = false;
    if (!attr_set) {
        const void *fns[] = {(const void *)heap_replay_lanes_kernel<true, false, 64>,
                             (const void *)heap_replay_lanes_kernel<false, false, 64>,
                             (const void *)heap_replay_lanes_kernel<true, true, 32>,
                             (const void *)heap_replay_lanes_kernel<false, true, 32>,
                             (const void *)heap_replay_lanes_kernel<true, false, 64, true>,
                             (const void *)heap_replay_lanes_kernel<false, false, 64, true>,
                             (const void *)heap_replay_lanes_kernel<true, false, 64, false, true>,
                             (const void *)heap_replay_lanes_kernel<false, false, 64, false, true>,
                             (const void *)heap_replay_lanes_kernel<true, false, 64, true, true>,
                             (const void *)heap_replay_lanes_kernel<false, false, 64, true, true>};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
                hipSuccess)
                return -1;
        attr_set = true;
    }
    {   // padding rows out of the way first (the kernels below no longer test `pos < n`)
        const int64_t items = nq * S;
        const unsigned pg = (unsigned)((items + 255) / 256);
        if (items > 0 && signd)
            hipLaunchKernelGGL(pad_fix_kernel<true>, dim3(pg), dim3(256), 0, s, (uint4 *)dist, cap, nq,
                               slot_prefix, slot_n, S, slots_uniform, (uint8_t *)mins, cap_min, flag_list);
        else if (items > 0)
            hipLaunchKernelGGL(pad_fix_kernel<false>, dim3(pg), dim3(256), 0, s, (uint4 *)dist, cap, nq,
                               slot_prefix, slot_n, S, slots_uniform, (uint8_t *)mins, cap_min, flag_list);
    }
    // one query-wave per workgroup (multi-wave workgroups are placed only when a whole CU has room,
    // which next to the persistent scan kernels means at their launch boundaries)
    const int wave_lds = (int)lds;
    const int64_t n_waves = (nq + LWr - 1) / LWr;
    dim3 grid((unsigned)n_waves);
    const int prio = 3;         // s_setprio of the replay waves: they share SIMDs with issue-bound scan waves
#define TK_LAUNCH3(S_, D_, L_)                                                                    \
    hipLaunchKernelGGL((heap_replay_lanes_kernel<S_, D_, L_>), grid, dim3(64), lds, s, dist, cap, nq,  \
                       slot_prefix, slot_n, slot_label_off, S, labels, heap_idx, heap_val, R,     \
                       slots_uniform, skip, 1, mins, cap_min, labels32, counters, prio, wave_lds,  \
                       plain0, qlim, tw, flag_list)
    if (dedupe) { if (signd) TK_LAUNCH3(true, true, 32); else TK_LAUNCH3(false, true, 32); }
    else if (twin) {
#define TK_LAUNCH_TWIN(S_, Z_)                                                                    \
    hipLaunchKernelGGL((heap_replay_lanes_kernel<S_, false, 64, Z_, true>), grid, dim3(64), lds, s, dist, cap, nq, \
                       slot_prefix, slot_n, slot_label_off, S, labels, heap_idx, heap_val, R,     \
                       slots_uniform, skip, 1, mins, cap_min, labels32, counters, prio, wave_lds,  \
                       plain0, qlim, tw, flag_list)
        if (signd) { if (lazy) TK_LAUNCH_TWIN(true, true); else TK_LAUNCH_TWIN(true, false); }
        else { if (lazy) TK_LAUNCH_TWIN(false, true); else TK_LAUNCH_TWIN(false, false); }
#undef TK_LAUNCH_TWIN
    } else if (lazy) {
#define TK_LAUNCH_LAZY(S_)                                                                        \
    hipLaunchKernelGGL((heap_replay_lanes_kernel<S_, false, 64, true>), grid, dim3(64), lds, s, dist, cap, nq, \
                       slot_prefix, slot_n, slot_label_off, S, labels, heap_idx, heap_val, R,     \
                       slots_uniform, skip, 1, mins, cap_min, labels32, counters, prio, wave_lds,  \
                       plain0, qlim, tw, flag_list)
        if (signd) TK_LAUNCH_LAZY(true); else TK_LAUNCH_LAZY(false);
#undef TK_LAUNCH_LAZY
    } else { if (signd) TK_LAUNCH3(true, false, 64); else TK_LAUNCH3(false, false, 64); }
#undef TK_LAUNCH3
    return 0;
}

// ---------------------------------------------------------------------------
// Wave-per-query replay with the heap in REGISTERS, two nodes per lane: heaps of up to 129 entries (IVF.query at the
// reference's bench settings: (n_probes + 1) k + 1 = 111; its coarse `top`: 2 n_probes + 10 = 30).  This is the kernel
// of ONE query per call (examples/bench.py:118-137 times exactly that) and of small batches, where the lane kernel's
// round of ~2 000 cycles serves a single lane: ~760 dependent inserts of one query were 0.47 of the call's 0.54 ms.
//
// Layout (tests/test_pair_heap_lemma.py restates it in numpy and checks it against the reference's loop): the root is
// wave-uniform; lane L holds the two CHILDREN of node L — node 2L+1 in slot 0, node 2L+2 in slot 1.  "Which child of
// node L is larger" (the left one on ties: `vl > v`, then `vr > nxt_val`, _fast_pq.pyx:297-300) is a comparison inside
// the lane; one ballot B of it describes the whole max-child path, which `insert`'s sift follows whatever v is: node t
// is on it iff every ancestor chose the child on the chain down to t — ((B ^ bits) & mask) == 0 with a per-lane constant
// pair.  Values never rise along the path, so with CE[L] = the entry of node L's larger child
//     root                                           <- CE[0]  if CE[0].v > v   else (label, v)
//     larger child c of an on-path node L, CE[L].v > v <- CE[c]  if c < 64 and CE[c].v > v   else (label, v)
// B, the path, CE and the cross-lane fetch of CE[c] (one ds_bpermute per register of an entry) are prepared BEHIND an
// insert, before the next candidate is known; a compare and two selects per lane remain on the candidate's own chain.
// Nodes >= R hold a value no candidate is below and are never taken.
// Entries: POSITIONS (value8 << 24 | flat position24, one register per slot) where no label can repeat among a query's
// lists; (value, label64) — three registers per slot — with the reference's duplicate test (`if i == indices[j]:
// return`, :284-287, one ballot) where labels repeat (IVF.build(n_probes >= 2)) or a probe list names a list twice.
#define TK_PAIR_LOW (-(1 << 20))

struct PairLane {      // constants of lane L = node L: its ancestors (as lanes, all < 32) and the child each must choose
    uint32_t anc_mask, anc_bits;
};
__device__ __forceinline__ PairLane pair_lane(int lane)
{
    PairLane K;
    K.anc_mask = K.anc_bits = 0;
    for (int t = lane; t > 0;) {
        const int par = (t - 1) >> 1;
        K.anc_mask |= 1u << par;
        K.anc_bits |= (uint32_t)((t - 1) & 1) << par;
        t = par;
    }
    return K;
}

// ---- position entries
template <bool SIGNED>
struct PairHeapPos {
    uint32_t e0, e1, er;            // er: wave-uniform
    // prepared behind every change
    bool right, onp;
    uint32_t ce, fe, ce0;
    int cev, fev;
    __device__ __forceinline__ void init(int lane, int R)
    {
        const uint32_t fresh = SIGNED ? 0x7fffffffu : 0xffffffffu;
        const uint32_t lowest = SIGNED ? 0x80ffffffu : 0x00ffffffu;
        er = fresh;
        e0 = 2 * lane + 1 < R ? fresh : lowest;
        e1 = 2 * lane + 2 < R ? fresh : lowest;
    }
    __device__ __forceinline__ void prepare(int lane, const PairLane &K)
    {
        const uint32_t lowest = SIGNED ? 0x80ffffffu : 0x00ffffffu;
        const int v0 = entry_val<SIGNED>(e0), v1 = entry_val<SIGNED>(e1);
        right = v1 > v0;
        const uint32_t ball = (uint32_t)__builtin_amdgcn_ballot_w64(right);
        onp = ((ball ^ K.anc_bits) & K.anc_mask) == 0;
        ce = right ? e1 : e0;
        cev = right ? v1 : v0;
        const int c = 2 * lane + 1 + (int)right;
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((c & 63) << 2, (int)ce);
        fe = c < 64 ? got : lowest;
        fev = entry_val<SIGNED>(fe);
        ce0 = __builtin_amdgcn_readlane(ce, 0);
    }
    __device__ __forceinline__ uint32_t bound() const { return er >> 24; }
    // entries value8 << 24 | label24 (labels below 0xffffff): `if i == indices[j]: return` (:284-287) on the low 24 bits —
    // fresh nodes and nodes >= R hold 0xffffff, which no row carries
    __device__ __forceinline__ bool holds24(uint32_t l24) const
    {
        const bool here = (((e0 ^ l24) & 0x00ffffffu) == 0) | (((e1 ^ l24) & 0x00ffffffu) == 0);
        return __builtin_amdgcn_ballot_w64(here) != 0 || ((er ^ l24) & 0x00ffffffu) == 0;
    }
    __device__ __forceinline__ void insert(uint32_t e, int v, int lane, const PairLane &K)
    {
        const bool upd = onp && cev > v;
        const uint32_t ne = fev > v ? fe : e;
        e0 = (upd && !right) ? ne : e0;
        e1 = (upd && right) ? ne : e1;
        er = entry_val<SIGNED>(ce0) > v ? ce0 : e;
        prepare(lane, K);
    }
};

// ---- (value, label) entries with the duplicate test
template <bool SIGNED>
struct PairHeapLab {
    int v0, v1, vr;
    uint32_t lo0, lo1, lor, hi0, hi1, hir;
    bool right, onp;
    int cev, fev, cev0;
    uint32_t clo, chi, flo, fhi, clo0, chi0;
    __device__ __forceinline__ void init(int lane, int R)
    {
        const int fresh = SIGNED ? 127 : 255;
        vr = fresh;
        v0 = 2 * lane + 1 < R ? fresh : TK_PAIR_LOW;
        v1 = 2 * lane + 2 < R ? fresh : TK_PAIR_LOW;
        lo0 = lo1 = lor = hi0 = hi1 = hir = 0xffffffffu;       // label -1
    }
    __device__ __forceinline__ void prepare(int lane, const PairLane &K)
    {
        right = v1 > v0;
        const uint32_t ball = (uint32_t)__builtin_amdgcn_ballot_w64(right);
        onp = ((ball ^ K.anc_bits) & K.anc_mask) == 0;
        cev = right ? v1 : v0;
        clo = right ? lo1 : lo0;
        chi = right ? hi1 : hi0;
        const int c = 2 * lane + 1 + (int)right;
        const int a = (c & 63) << 2;
        const int gv = __builtin_amdgcn_ds_bpermute(a, cev);
        const uint32_t gl = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)clo);
        const uint32_t gh = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)chi);
        fev = c < 64 ? gv : TK_PAIR_LOW;
        flo = gl;
        fhi = gh;
        cev0 = __builtin_amdgcn_readlane(cev, 0);
        clo0 = __builtin_amdgcn_readlane(clo, 0);
        chi0 = __builtin_amdgcn_readlane(chi, 0);
    }
    __device__ __forceinline__ uint32_t bound() const { return (uint32_t)vr & 0xffu; }
    // `if i == indices[j]: return` (:284-287): nodes >= R and fresh nodes hold -1, which no row carries
    __device__ __forceinline__ bool holds(uint32_t llo, uint32_t lhi) const
    {
        const bool here = ((lo0 == llo) & (hi0 == lhi)) | ((lo1 == llo) & (hi1 == lhi));
        return __builtin_amdgcn_ballot_w64(here) != 0 || (lor == llo && hir == lhi);
    }
    __device__ __forceinline__ void insert(uint32_t llo, uint32_t lhi, int v, int lane, const PairLane &K)
    {
        const bool upd = onp && cev > v;
        const bool deeper = fev > v;
        const int nv = deeper ? fev : v;
        const uint32_t nl = deeper ? flo : llo, nh = deeper ? fhi : lhi;
        const bool u0 = upd && !right, u1 = upd && right;
        v0 = u0 ? nv : v0; lo0 = u0 ? nl : lo0; hi0 = u0 ? nh : hi0;
        v1 = u1 ? nv : v1; lo1 = u1 ? nl : lo1; hi1 = u1 ? nh : hi1;
        const bool rt = cev0 > v;
        vr = rt ? cev0 : v; lor = rt ? clo0 : llo; hir = rt ? chi0 : lhi;
        prepare(lane, K);
    }
};

// ---- G groups of node pairs per lane: heaps of up to 128 G + 1 entries (G = 2: 257 — n_probes <= 24 at k = 10; G = 4: 513 —
// n_probes <= 50).  Lane L, group g holds the children of node t = 64 g + L.  One ballot per group describes the path; the
// ancestors of an internal node are all below 32 G (the lower half of the groups), so a node's on-path test is a mask test
// against the ballots of those groups with per-lane constants; the chosen child c = 2 t + 1 + right of a node of group g
// has its own pair in lane c & 63 of group c >> 6 in {2g, 2g+1, 2g+2} (c < 64 G), else it is a leaf.  Formulation checked
// on the CPU for G = 1, 2, 4 (tests/test_pair_heap_lemma.py: GroupHeap).  G = 1 keeps the structs above (same arithmetic,
// written out: the kernel of the smallest heaps is the one that matters most).
template <int G>
struct GroupLane {
    static constexpr int GA = G >= 2 ? G / 2 : 1;      // groups that hold ancestors
    uint32_t m_lo[G][GA], m_hi[G][GA], b_lo[G][GA], b_hi[G][GA];
};
template <int G>
__device__ __forceinline__ GroupLane<G> group_lane(int lane)
{
    GroupLane<G> K;
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int a = 0; a < GroupLane<G>::GA; a++) K.m_lo[g][a] = K.m_hi[g][a] = K.b_lo[g][a] = K.b_hi[g][a] = 0;
        for (int t = 64 * g + lane; t > 0;) {
            const int par = (t - 1) >> 1;
            const uint32_t bit = (uint32_t)((t - 1) & 1);
            const int a = par >> 6, l = par & 63;       // a < GA by construction
#pragma unroll
            for (int aa = 0; aa < GroupLane<G>::GA; aa++)
                if (aa == a) {
                    if (l < 32) { K.m_lo[g][aa] |= 1u << l; K.b_lo[g][aa] |= bit << l; }
                    else { K.m_hi[g][aa] |= 1u << (l - 32); K.b_hi[g][aa] |= bit << (l - 32); }
                }
            t = par;
        }
    }
    return K;
}

template <bool SIGNED, int G>
struct GroupHeapPos {
    static constexpr int GROUPS = G;
    uint32_t e0[G], e1[G], er;
    bool right[G], onp[G];
    uint32_t ce[G], fe[G], ce0;
    __device__ __forceinline__ void init(int lane, int R)
    {
        const uint32_t fresh = SIGNED ? 0x7fffffffu : 0xffffffffu;
        const uint32_t lowest = SIGNED ? 0x80ffffffu : 0x00ffffffu;
        er = fresh;
#pragma unroll
        for (int g = 0; g < G; g++) {
            e0[g] = 2 * (64 * g + lane) + 1 < R ? fresh : lowest;
            e1[g] = 2 * (64 * g + lane) + 2 < R ? fresh : lowest;
        }
    }
    __device__ __forceinline__ void prepare(int lane, const GroupLane<G> &K)
    {
        const uint32_t lowest = SIGNED ? 0x80ffffffu : 0x00ffffffu;
        uint64_t B[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            right[g] = entry_val<SIGNED>(e1[g]) > entry_val<SIGNED>(e0[g]);
            B[g] = __builtin_amdgcn_ballot_w64(right[g]);
            ce[g] = right[g] ? e1[g] : e0[g];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            uint32_t bad = 0;
#pragma unroll
            for (int a = 0; a < GroupLane<G>::GA; a++)
                bad |= (((uint32_t)B[a] ^ K.b_lo[g][a]) & K.m_lo[g][a]) | (((uint32_t)(B[a] >> 32) ^ K.b_hi[g][a]) & K.m_hi[g][a]);
            onp[g] = bad == 0;
            const int c = 2 * (64 * g + lane) + 1 + (int)right[g];
            const int a4 = (c & 63) << 2;
            uint32_t f = lowest;
#pragma unroll
            for (int sg = 2 * g; sg <= 2 * g + 2; sg++)
                if (sg < G) {
                    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)ce[sg]);
                    f = (c >> 6) == sg ? got : f;
                }
            fe[g] = f;
        }
        ce0 = __builtin_amdgcn_readlane(ce[0], 0);
    }
    __device__ __forceinline__ uint32_t bound() const { return er >> 24; }
    __device__ __forceinline__ bool holds24(uint32_t l24) const
    {
        bool here = false;
#pragma unroll
        for (int g = 0; g < G; g++) here |= (((e0[g] ^ l24) & 0x00ffffffu) == 0) | (((e1[g] ^ l24) & 0x00ffffffu) == 0);
        return __builtin_amdgcn_ballot_w64(here) != 0 || ((er ^ l24) & 0x00ffffffu) == 0;
    }
    __device__ __forceinline__ void insert(uint32_t e, int v, int lane, const GroupLane<G> &K)
    {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const bool upd = onp[g] && entry_val<SIGNED>(ce[g]) > v;
            const uint32_t ne = entry_val<SIGNED>(fe[g]) > v ? fe[g] : e;
            e0[g] = (upd && !right[g]) ? ne : e0[g];
            e1[g] = (upd && right[g]) ? ne : e1[g];
        }
        er = entry_val<SIGNED>(ce0) > v ? ce0 : e;
        prepare(lane, K);
    }
    __device__ __forceinline__ uint32_t slot_entry(int g, int slot) const { return slot ? e1[g] : e0[g]; }
};

template <bool SIGNED, int G>
struct GroupHeapLab {
    static constexpr int GROUPS = G;
    int v0[G], v1[G], vr;
    uint32_t lo0[G], lo1[G], hi0[G], hi1[G], lor, hir;
    bool right[G], onp[G];
    int cev[G], fev[G], cev0;
    uint32_t clo[G], chi[G], flo[G], fhi[G], clo0, chi0;
    __device__ __forceinline__ void init(int lane, int R)
    {
        const int fresh = SIGNED ? 127 : 255;
        vr = fresh;
        lor = hir = 0xffffffffu;
#pragma unroll
        for (int g = 0; g < G; g++) {
            v0[g] = 2 * (64 * g + lane) + 1 < R ? fresh : TK_PAIR_LOW;
            v1[g] = 2 * (64 * g + lane) + 2 < R ? fresh : TK_PAIR_LOW;
            lo0[g] = lo1[g] = hi0[g] = hi1[g] = 0xffffffffu;       // label -1
        }
    }
    __device__ __forceinline__ void prepare(int lane, const GroupLane<G> &K)
    {
        uint64_t B[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            right[g] = v1[g] > v0[g];
            B[g] = __builtin_amdgcn_ballot_w64(right[g]);
            cev[g] = right[g] ? v1[g] : v0[g];
            clo[g] = right[g] ? lo1[g] : lo0[g];
            chi[g] = right[g] ? hi1[g] : hi0[g];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            uint32_t bad = 0;
#pragma unroll
            for (int a = 0; a < GroupLane<G>::GA; a++)
                bad |= (((uint32_t)B[a] ^ K.b_lo[g][a]) & K.m_lo[g][a]) | (((uint32_t)(B[a] >> 32) ^ K.b_hi[g][a]) & K.m_hi[g][a]);
            onp[g] = bad == 0;
            const int c = 2 * (64 * g + lane) + 1 + (int)right[g];
            const int a4 = (c & 63) << 2;
            int fv = TK_PAIR_LOW;
            uint32_t fl = 0xffffffffu, fh = 0xffffffffu;
#pragma unroll
            for (int sg = 2 * g; sg <= 2 * g + 2; sg++)
                if (sg < G) {
                    const int gv = __builtin_amdgcn_ds_bpermute(a4, cev[sg]);
                    const uint32_t gl = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)clo[sg]);
                    const uint32_t gh = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)chi[sg]);
                    const bool here = (c >> 6) == sg;
                    fv = here ? gv : fv; fl = here ? gl : fl; fh = here ? gh : fh;
                }
            fev[g] = fv; flo[g] = fl; fhi[g] = fh;
        }
        cev0 = __builtin_amdgcn_readlane(cev[0], 0);
        clo0 = __builtin_amdgcn_readlane(clo[0], 0);
        chi0 = __builtin_amdgcn_readlane(chi[0], 0);
    }
    __device__ __forceinline__ uint32_t bound() const { return (uint32_t)vr & 0xffu; }
    __device__ __forceinline__ bool holds(uint32_t llo, uint32_t lhi) const
    {
        bool here = false;
#pragma unroll
        for (int g = 0; g < G; g++)
            here |= ((lo0[g] == llo) & (hi0[g] == lhi)) | ((lo1[g] == llo) & (hi1[g] == lhi));
        return __builtin_amdgcn_ballot_w64(here) != 0 || (lor == llo && hir == lhi);
    }
    __device__ __forceinline__ void insert(uint32_t llo, uint32_t lhi, int v, int lane, const GroupLane<G> &K)
    {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const bool upd = onp[g] && cev[g] > v;
            const bool deeper = fev[g] > v;
            const int nv = deeper ? fev[g] : v;
            const uint32_t nl = deeper ? flo[g] : llo, nh = deeper ? fhi[g] : lhi;
            const bool u0 = upd && !right[g], u1 = upd && right[g];
            v0[g] = u0 ? nv : v0[g]; lo0[g] = u0 ? nl : lo0[g]; hi0[g] = u0 ? nh : hi0[g];
            v1[g] = u1 ? nv : v1[g]; lo1[g] = u1 ? nl : lo1[g]; hi1[g] = u1 ? nh : hi1[g];
        }
        const bool rt = cev0 > v;
        vr = rt ? cev0 : v; lor = rt ? clo0 : llo; hir = rt ? chi0 : lhi;
        prepare(lane, K);
    }
};

// The replay of one query by one wave: LABELS = the duplicate test on (value, label) entries, else position entries.
// (Two instantiations called from one wave-uniform branch: as ONE loop with a run-time switch the compiler merged
//  both heaps' registers and masks through every iteration — ~30 scalar moves per insert.)
// KIND 0: position entries (no label can repeat among the query's lists); 1: (value, label64) entries with the duplicate
// test; 2: the duplicate test on value8 << 24 | label24 entries (every label of the index below 0xffffff: one register per
// slot as with positions — the reference's default build, IVF.build(n_probes=2), of anything up to 16.7 M rows)
template <bool SIGNED, int KIND, int G>
__device__ __forceinline__ void pair_replay_body(const uint4 *__restrict__ drow, const uint8_t *__restrict__ mrow,
                                                 const int *__restrict__ prefix, const int *__restrict__ ns,
                                                 const int64_t *__restrict__ loffs, int S,
                                                 const int64_t *__restrict__ labels, int64_t *__restrict__ oi,
                                                 int32_t *__restrict__ ov, int R, int lane, int plain0, uint32_t &b_plain)
{
    // G = 1 (heaps <= 129): the written-out structs; G = 2 / 4: the same arithmetic over G groups of pairs per lane
    typename std::conditional<G == 1, PairLane, GroupLane<G>>::type K;
    if constexpr (G == 1) K = pair_lane(lane); else K = group_lane<G>(lane);
    const int total = S > 0 ? prefix[S] : 0;
    const uint4 never = SIGNED ? make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu)
                               : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    constexpr bool LABELS = KIND == 1;          // three registers per slot
    constexpr bool NEEDS_LABELS = KIND != 0;    // the candidate rows' labels are loaded
    typename std::conditional<G == 1,
                              typename std::conditional<LABELS, PairHeapLab<SIGNED>, PairHeapPos<SIGNED>>::type,
                              typename std::conditional<LABELS, GroupHeapLab<SIGNED, G>, GroupHeapPos<SIGNED, G>>::type>::type H;
    H.init(lane, R);
    H.prepare(lane, K);
    uint32_t bound = SIGNED ? 0x7fu : 0xffu;
    // slot cursor (wave-uniform): flat blocks [c0, c1) belong to slot s, a list of n_s rows with its labels at loff_s
    int s = 0, c0 = 0, c1 = S > 0 ? prefix[1] : 0;
    int n_s = S > 0 ? ns[0] : 0;
    int64_t loff_s = S > 0 ? loffs[0] : -1;
    // 64 blocks per step, each lane one block and its minimum; the next step's are requested before this one is replayed
    uint4 nd = never;
    uint32_t nm = SIGNED ? 0x7fu : 0xffu;
    if (lane < total) { nd = drow[lane]; nm = mrow[lane]; }
    for (int base = 0; base < total; base += 64) {
        const uint4 dd = nd;
        const uint32_t mn = nm;
        nd = never;
        nm = SIGNED ? 0x7fu : 0xffu;
        if (base + 64 + lane < total) { nd = drow[base + 64 + lane]; nm = mrow[base + 64 + lane]; }
        // blocks whose minimum is below the bound of now: a superset of what the reference enters (the bound only falls)
        uint64_t mask = __builtin_amdgcn_ballot_w64(byte_lt<SIGNED>(mn, bound));
        // The labels of a voted block — lanes 0..15, one vector load — are requested one voted block ahead (two ahead was
        // measured: no gain).  The slot's row count and label offset live in registers and change only where the cursor
        // steps into the next list: read through `ns[s]` / `loffs[s]` per voted block they were a scalar load each in
        // front of every block's cmp_mask — the build(n_probes=2) index (labels: three such loads per voted block) spent
        // 0.28 ms in this replay against 0.09 for distinct labels.
        // (a macro, not a lambda: a closure over c0 / n_s / loff_s by reference whose body selects between them and
        //  freshly loaded values went to scratch — 12 bytes per lane, tests/test_kernel_resources.py)
#define TK_LOAD_LABELS(out_, f2_)                                                          \
        {                                                                                  \
            const int f2__ = (f2_);                                                        \
            int a0__ = c0, n2__ = n_s;                                                     \
            int64_t loff2__ = loff_s;                                                      \
            if (f2__ >= c1) {       /* (rare: the next voted block lies in a later list) */ \
                int s2__ = s, a1__ = c1;                                                   \
                while (f2__ >= a1__) { s2__++; a0__ = a1__; a1__ = prefix[s2__ + 1]; }     \
                n2__ = ns[s2__];                                                           \
                loff2__ = loffs[s2__];                                                     \
            }                                                                              \
            const int64_t inl2__ = 16 * (int64_t)(f2__ - a0__) + lane;                     \
            (out_) = -2;                                                                   \
            if (lane < 16 && inl2__ < n2__) (out_) = loff2__ < 0 ? inl2__ : labels[loff2__ + inl2__]; \
        }
        int j_pref = -1;
        int64_t lab_pref = -2;
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int f = base + j;
            while (f >= c1) {            // (empty lists: several steps)
                s++;
                c0 = c1;
                c1 = prefix[s + 1];
                n_s = ns[s];
                loff_s = loffs[s];
            }
            const int rows = n_s - 16 * (f - c0);           // `pos < n`, _fast_pq_256.pyx:111
            int64_t lab_cur = -2;
            if (NEEDS_LABELS) {
                if (j == j_pref) lab_cur = lab_pref;
                else TK_LOAD_LABELS(lab_cur, f)
                // The labels of THIS block are taken into use here, in front of the next request: the compiler waits
                // for a load with `s_waitcnt vmcnt(0)` at its first use (it cannot count loads across the loop's back
                // edge); with that first use behind the next request the "prefetch" would be waited for at once.
                // (Worth 3 % on the build(n_probes=2) index, 0.254 -> 0.245 ms per query: its replay is 2.7 x the one
                // over distinct labels because the lists are twice as long and every near row is met twice — ~1 240
                // candidates and ~800 inserts against 620 — not because of the labels' trip to memory.)
                uint32_t cur_lo = (uint32_t)lab_cur, cur_hi = (uint32_t)((uint64_t)lab_cur >> 32);
                asm volatile("" : "+v"(cur_lo), "+v"(cur_hi));
                __builtin_amdgcn_sched_barrier(0);
                lab_cur = (int64_t)(((uint64_t)cur_hi << 32) | cur_lo);
                j_pref = -1;
                if (mask) {                                 // the next voted block of this step (a superset: the bound may fall)
                    j_pref = __builtin_ctzll(mask);
                    TK_LOAD_LABELS(lab_pref, base + j_pref)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint32_t d0 = __builtin_amdgcn_readlane(dd.x, j);
            const uint32_t d1 = __builtin_amdgcn_readlane(dd.y, j);
            const uint32_t d2 = __builtin_amdgcn_readlane(dd.z, j);
            const uint32_t d3 = __builtin_amdgcn_readlane(dd.w, j);
            // the reference's cmp_mask of this block against the bound captured at its start: row r in lane r
            const uint32_t w = lane < 4 ? d0 : lane < 8 ? d1 : lane < 12 ? d2 : d3;
            const uint32_t by = (w >> (8 * (lane & 3))) & 0xffu;
            uint32_t bits = (uint32_t)__builtin_amdgcn_ballot_w64(lane < 16 && lane < rows && byte_lt<SIGNED>(by, bound));
            if (!bits) continue;
            const uint32_t pos0 = (uint32_t)(16 * f);
            while (bits) {               // every passing row goes in, in row order, without a second look (:113-118)
                const int r = __builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t byr = __builtin_amdgcn_readlane(by, r);
                const int v = SIGNED ? (int)(int8_t)byr : (int)byr;
                if constexpr (LABELS) {
                    const uint32_t llo = __builtin_amdgcn_readlane((uint32_t)lab_cur, r);
                    const uint32_t lhi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)lab_cur >> 32), r);
                    if (H.holds(llo, lhi)) continue;
                    H.insert(llo, lhi, v, lane, K);
                } else if constexpr (KIND == 2) {
                    const uint32_t l24 = __builtin_amdgcn_readlane((uint32_t)lab_cur, r) & 0x00ffffffu;
                    if (H.holds24(l24)) continue;
                    H.insert((byr << 24) | l24, v, lane, K);
                } else {
                    H.insert((byr << 24) | (pos0 + (uint32_t)r), v, lane, K);
                }
            }
            bound = H.bound();                                              // refresh behind the block, :123
            b_plain = f < plain0 ? bound : b_plain;                         // (plain_scan.hip's lemma: the bound at the first plain block)
            if (mask) mask &= __builtin_amdgcn_ballot_w64(byte_lt<SIGNED>(mn, bound));
        }
    }
#undef TK_LOAD_LABELS
    // heap arrays out, in the reference's layout: node 0 = the root, node 2 t + 1 + slot = the slot of lane L, group g (t = 64 g + L)
    auto resolve = [&](uint32_t e) -> int64_t {      // a flat position back to (list, row): its label (KIND 2: the label itself)
        const uint32_t pos = e & 0x00ffffffu;
        if (pos == 0x00ffffffu) return -1;
        if (KIND == 2) return (int64_t)pos;
        const int fb = (int)(pos >> 4);
        int lo = 0, hi = S;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= fb) lo = mid; else hi = mid;
        }
        const int64_t inlist = (int64_t)pos - 16 * (int64_t)prefix[lo];
        const int64_t loff = loffs[lo];
        return loff < 0 ? inlist : labels[loff + inlist];
    };
    if constexpr (G == 1 && LABELS) {
        if (lane == 0) { oi[0] = (int64_t)(((uint64_t)H.hir << 32) | H.lor); ov[0] = H.vr; }
        if (2 * lane + 1 < R) { oi[2 * lane + 1] = (int64_t)(((uint64_t)H.hi0 << 32) | H.lo0); ov[2 * lane + 1] = H.v0; }
        if (2 * lane + 2 < R) { oi[2 * lane + 2] = (int64_t)(((uint64_t)H.hi1 << 32) | H.lo1); ov[2 * lane + 2] = H.v1; }
    } else if constexpr (G == 1) {
        if (lane == 0) { oi[0] = resolve(H.er); ov[0] = entry_val<SIGNED>(H.er); }
        if (2 * lane + 1 < R) { oi[2 * lane + 1] = resolve(H.e0); ov[2 * lane + 1] = entry_val<SIGNED>(H.e0); }
        if (2 * lane + 2 < R) { oi[2 * lane + 2] = resolve(H.e1); ov[2 * lane + 2] = entry_val<SIGNED>(H.e1); }
    } else if constexpr (LABELS) {
        if (lane == 0) { oi[0] = (int64_t)(((uint64_t)H.hir << 32) | H.lor); ov[0] = H.vr; }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int t0 = 2 * (64 * g + lane) + 1;
            if (t0 < R) { oi[t0] = (int64_t)(((uint64_t)H.hi0[g] << 32) | H.lo0[g]); ov[t0] = H.v0[g]; }
            if (t0 + 1 < R) { oi[t0 + 1] = (int64_t)(((uint64_t)H.hi1[g] << 32) | H.lo1[g]); ov[t0 + 1] = H.v1[g]; }
        }
    } else {
        if (lane == 0) { oi[0] = resolve(H.er); ov[0] = entry_val<SIGNED>(H.er); }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int t0 = 2 * (64 * g + lane) + 1;
            if (t0 < R) { oi[t0] = resolve(H.e0[g]); ov[t0] = entry_val<SIGNED>(H.e0[g]); }
            if (t0 + 1 < R) { oi[t0 + 1] = resolve(H.e1[g]); ov[t0 + 1] = entry_val<SIGNED>(H.e1[g]); }
        }
    }
}

// One wave per query.  dist / mins / slot tables as heap_replay_packed_kernel; `flags`: queries that need the duplicate
// test although labels are distinct (a probe list that names a list twice); dedupe_all: labels repeat in the index.
// plain0_arr / qlim / flag_list (plain_scan.hip: the blocks from flat chunk plain0_arr[q] on carry clamp(plain sums)): the
// replay checks the lemma's condition per query as the lane kernel does — bound at the first plain block <= qlim[q] —
// and lists the queries that fail it, and those flagged beforehand, for the exact re-scan behind it (flags[q] = 2).
template <bool SIGNED, int G>
__global__ __launch_bounds__(64) void heap_replay_pair_kernel(
    const uint4 *__restrict__ dist, int64_t cap, const uint8_t *__restrict__ mins, int64_t cap_min,
    const int *__restrict__ slot_prefix, const int *__restrict__ slot_n, const int64_t *__restrict__ slot_label_off,
    int S, const int64_t *__restrict__ labels, int64_t *__restrict__ heap_idx, int32_t *__restrict__ heap_val, int R,
    int slots_uniform, unsigned char *__restrict__ flags, int dedupe_all, int64_t nq,
    const int *__restrict__ plain0_arr, const int *__restrict__ qlim, int *__restrict__ flag_list, int only_flagged,
    const int *__restrict__ count_src, volatile int *host_count)
{
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    // (the count of the flagged queries to the page-locked word the host polls: plain_scan's state machine)
    if (host_count && count_src && q == 0 && lane == 0) *host_count = count_src[0];
    if (q >= nq) return;
    const bool flagged = flags && flags[q];
    if (only_flagged && !flagged) return;      // the tail behind a lane replay: the queries it left (re-scanned exactly)
    __builtin_amdgcn_s_setprio(3);
    const int64_t qs = slots_uniform ? 0 : q;
    if (plain0_arr && flagged) {       // (its rows hold plain sums: left, with the queries that fail the check, to the kernels behind)
        if (lane == 0 && flag_list) flag_list[1 + atomicAdd(&flag_list[0], 1)] = (int)q;
        return;
    }
    const int plain0 = plain0_arr ? plain0_arr[q] : 0x7fffffff;
    uint32_t b_plain = SIGNED ? 0x7fu : 0xffu;
    const int *prefix = slot_prefix + qs * (S + 1);
    // dedupe_all: bit 0 = every query takes the duplicate test (labels repeat in the index), bit 1 = every label is below 0xffffff
    const uint4 *drow = dist + q * cap;
    const uint8_t *mrow = mins + q * cap_min;
    const int *ns = slot_n + qs * S;
    const int64_t *loffs = slot_label_off + qs * S;
    int64_t *oi = heap_idx + q * R;
    int32_t *ov = heap_val + q * R;
    if (!((dedupe_all & 1) || flagged))
        pair_replay_body<SIGNED, 0, G>(drow, mrow, prefix, ns, loffs, S, labels, oi, ov, R, lane, plain0, b_plain);
    else if (dedupe_all & 2)
        pair_replay_body<SIGNED, 2, G>(drow, mrow, prefix, ns, loffs, S, labels, oi, ov, R, lane, plain0, b_plain);
    else
        pair_replay_body<SIGNED, 1, G>(drow, mrow, prefix, ns, loffs, S, labels, oi, ov, R, lane, plain0, b_plain);
    if (SIGNED && plain0_arr && S > 0 && plain0 < prefix[S] && (int)(int8_t)b_plain > qlim[q] && lane == 0) {
        flags[q] = 2;
        if (flag_list) flag_list[1 + atomicAdd(&flag_list[0], 1)] = (int)q;
    }
}

// cap * 16 <= 0xffffff (position entries) is the caller's to check where labels are distinct.  plain0 / qlim (+ flags): both
// or none (signed tables only); flag_list (optional: the failing queries listed for a re-scan) has its count zeroed here,
// on the stream, in front of the kernel.  only_flagged: replay only the queries with flags[q] != 0 (the tail behind a lane
// replay: queries it flagged, re-scanned exactly meanwhile) — with the duplicate test; count_src / host_count: count_src[0]
// is copied to the page-locked *host_count (the flagged count the host polls).
int tk_launch_heap_replay_pair(const uint4 *dist, int64_t cap, int64_t nq, const uint8_t *mins, int64_t cap_min,
                               const int *slot_prefix, const int *slot_n, const int64_t *slot_label_off, int S,
                               const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R, int signd,
                               int slots_uniform, unsigned char *flags, int dedupe_all, hipStream_t s,
                               const int *plain0, const int *qlim, int *flag_list, int only_flagged,
                               const int *count_src, int *host_count)
{
    if (nq == 0 || R == 0) return 0;
    if (!plain0 || !qlim || !flags || !signd) plain0 = qlim = nullptr, flag_list = nullptr;
    if (flag_list && hipMemsetAsync(flag_list, 0, 4, s) != hipSuccess) return -1;
#define TK_LAUNCH_PAIR(S_, G_)                                                                                           \
    hipLaunchKernelGGL((heap_replay_pair_kernel<S_, G_>), dim3((unsigned)nq), dim3(64), 0, s, dist, cap, mins, cap_min,  \
                       slot_prefix, slot_n, slot_label_off, S, labels, heap_idx, heap_val, R, slots_uniform, flags,     \
                       dedupe_all, nq, plain0, qlim, flag_list, only_flagged, count_src, host_count)
    // two nodes per lane up to 129 entries, four up to 257 (n_probes <= 24 at k = 10), eight up to 513 (n_probes <= 50)
    if (R <= 129) { if (signd) TK_LAUNCH_PAIR(true, 1); else TK_LAUNCH_PAIR(false, 1); }
    else if (R <= 257) { if (signd) TK_LAUNCH_PAIR(true, 2); else TK_LAUNCH_PAIR(false, 2); }
    else if (R <= TK_PAIR_MAX_R) { if (signd) TK_LAUNCH_PAIR(true, 4); else TK_LAUNCH_PAIR(false, 4); }
    else return -1;
#undef TK_LAUNCH_PAIR
    return 0;
}

// ---------------------------------------------------------------------------
__global__ void heap_fill_kernel(int64_t *heap_idx, int32_t *heap_val, int64_t count, int32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        heap_idx[i] = -1;
        heap_val[i] = v;
    }
}

void tk_launch_heap_fill(int64_t *heap_idx, int32_t *heap_val, int64_t count, int32_t v,
                         hipStream_t s)
{
    if (count == 0) return;
    hipLaunchKernelGGL(heap_fill_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s,
                       heap_idx, heap_val, count, v);
}

__global__ __launch_bounds__(64) void heap_insert_kernel(int64_t *heap_idx, int32_t *heap_val,
                                                         int R, int64_t i, int32_t v, int is)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_vi64 *hidx = LDS_PTR(lds_vi64, smem);
    lds_vi32 *hval = LDS_PTR(lds_vi32, smem + (size_t)R * 8);
    const int lane = threadIdx.x;
    for (int t = lane; t < R; t += 64) {
        hidx[t] = heap_idx[t];
        hval[t] = heap_val[t];
    }
    if (is)
        heap_insert_is(hidx, hval, R, i, v, lane);
    else
        heap_insert(hidx, hval, R, i, v, lane);
    for (int t = lane; t < R; t += 64) {
        heap_idx[t] = hidx[t];
        heap_val[t] = hval[t];
    }
}

void tk_launch_heap_insert(int64_t *heap_idx, int32_t *heap_val, int R, int64_t i, int32_t v,
                           int is, hipStream_t s)
{
    hipLaunchKernelGGL(heap_insert_kernel, dim3(1), dim3(64), (size_t)R * 12, s, heap_idx,
                       heap_val, R, i, v, is);
}
