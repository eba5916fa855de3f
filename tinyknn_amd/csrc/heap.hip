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
// strictly greater).  This file replays exactly that, one wavefront per query:
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
#include "kernels.h"

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
__device__ __forceinline__ int32_t lds_i32(volatile int32_t *p)
{
    return __builtin_amdgcn_readfirstlane(*p);
}
__device__ __forceinline__ int64_t lds_i64(volatile int64_t *p)
{
    int64_t v = *p;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// insert (_fast_pq.pyx:274-307).  Called by all 64 lanes with wave-uniform
// arguments; every lane performs the same LDS writes so that the heap is
// coherent in each lane's own program order.
__device__ __forceinline__ void heap_insert(volatile int64_t *hidx, volatile int32_t *hval, int R,
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
            hval[j] = v;
            hidx[j] = label;
            break;
        }
        hval[j] = nxt_val;
        hidx[j] = lds_i64(&hidx[nxt]);
        j = nxt;
    }
}

// insert_is (_fast_pq.pyx:256-271)
__device__ __forceinline__ void heap_insert_is(volatile int64_t *hidx, volatile int32_t *hval,
                                               int R, int64_t label, int32_t v, int lane)
{
    bool dup = false;
    for (int t = lane; t < R; t += 64) dup |= (hidx[t] == label);
    if (__builtin_amdgcn_ballot_w64(dup)) return;
    int j = 0;
    while (j + 1 != R) {
        int32_t nv = lds_i32(&hval[j + 1]);
        if (!(nv > v)) break;
        hidx[j] = lds_i64(&hidx[j + 1]);
        hval[j] = nv;
        j++;
    }
    hidx[j] = label;
    hval[j] = v;
}

template <bool SIGNED>
__global__ __launch_bounds__(64) void heap_replay_kernel(
    const uint4 *__restrict__ dist, int64_t cap, const int *__restrict__ slot_prefix,
    const int *__restrict__ slot_n, const int64_t *__restrict__ slot_label_off, int S,
    const int64_t *__restrict__ labels, int64_t *__restrict__ heap_idx,
    int32_t *__restrict__ heap_val, int R, int slots_uniform)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    volatile int64_t *hidx = (volatile int64_t *)smem;
    volatile int32_t *hval = (volatile int32_t *)(smem + (size_t)R * 8);
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;

    for (int t = lane; t < R; t += 64) {
        hidx[t] = heap_idx[q * R + t];
        hval[t] = heap_val[q * R + t];
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
        for (int base = 0; base < nchunks; base += 64) {
            const int b = base + lane;
            const bool have = b < nchunks;
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
    for (int t = lane; t < R; t += 64) {
        heap_idx[q * R + t] = hidx[t];
        heap_val[q * R + t] = hval[t];
    }
}

void tk_launch_heap_replay(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                           const int *slot_n, const int64_t *slot_label_off, int S,
                           const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R,
                           int signd, int slots_uniform, hipStream_t s)
{
    if (nq == 0 || R == 0) return;
    size_t lds = (size_t)R * 12;
    if (signd)
        hipLaunchKernelGGL(heap_replay_kernel<true>, dim3((unsigned)nq), dim3(64), lds, s, dist,
                           cap, slot_prefix, slot_n, slot_label_off, S, labels, heap_idx,
                           heap_val, R, slots_uniform);
    else
        hipLaunchKernelGGL(heap_replay_kernel<false>, dim3((unsigned)nq), dim3(64), lds, s, dist,
                           cap, slot_prefix, slot_n, slot_label_off, S, labels, heap_idx,
                           heap_val, R, slots_uniform);
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
    volatile int64_t *hidx = (volatile int64_t *)smem;
    volatile int32_t *hval = (volatile int32_t *)(smem + (size_t)R * 8);
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
