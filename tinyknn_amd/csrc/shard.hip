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

// blockIdx.x < W: sender role, stream me -> peer.   spos (nq*S): position in the send
//                 buffer of every slot I own (region = peer), -1 otherwise.
// blockIdx.x >= W: receiver role, stream peer -> me.   rpos (qh*S): position in the
//                 receive buffer of the slots of MY queries that peer owns (region = peer).
__global__ __launch_bounds__(1024) void shard_positions_kernel(
    const int64_t *__restrict__ probes, const int *__restrict__ slot_prefix, int S, int64_t nq,
    int64_t n_lists, const int *__restrict__ owner, int me, int W, int64_t qh, int64_t C,
    int *__restrict__ spos, int *__restrict__ rpos, int *__restrict__ flag)
{
    __shared__ int s_v[1024];
    __shared__ int64_t carry;
    const bool sender = (int)blockIdx.x < W;
    const int peer = sender ? (int)blockIdx.x : (int)blockIdx.x - W;
    const int home = sender ? peer : me;
    const int own = sender ? me : peer;
    const int64_t q0 = (int64_t)home * qh;
    int64_t q1 = q0 + qh;
    if (q1 > nq) q1 = nq;
    const int64_t n_e = q1 > q0 ? (q1 - q0) * S : 0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    // 8 consecutive entries per thread: a serial prefix inside the thread, one block-wide
    // scan of the 1024 thread sums per 8192 entries (at W = 1 a stream is nq * S = 10^5
    // entries long: 98 block scans of 1024 became 13)
    constexpr int PER = 8;
    for (int64_t base = 0; base < n_e; base += 1024 * PER) {
        int c[PER];
        bool mine[PER];
        int tot = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int64_t e = base + (int64_t)threadIdx.x * PER + u;
            c[u] = 0;
            mine[u] = false;
            if (e < n_e) {
                const int64_t qi = q0 + e / S;
                const int sl = (int)(e % S);
                int64_t cl = probes[qi * S + sl];
                if (cl < 0) cl += n_lists;
                mine[u] = owner[cl] == own;
                if (mine[u]) c[u] = slot_prefix[qi * (S + 1) + sl + 1] - slot_prefix[qi * (S + 1) + sl];
            }
            tot += c[u];
        }
        s_v[threadIdx.x] = tot;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan of the thread sums
            int v = threadIdx.x >= (unsigned)o ? s_v[threadIdx.x - o] : 0;
            __syncthreads();
            s_v[threadIdx.x] += v;
            __syncthreads();
        }
        int64_t run = carry + s_v[threadIdx.x] - tot;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int64_t e = base + (int64_t)threadIdx.x * PER + u;
            if (e < n_e) {
                const int64_t qi = q0 + e / S;
                const int sl = (int)(e % S);
                const bool fits = run + c[u] <= C;
                const int where = fits ? (int)((int64_t)peer * C + run) : -1;
                if (sender) spos[qi * S + sl] = mine[u] ? where : -1;
                else if (mine[u]) rpos[(qi - q0) * S + sl] = where;
                if (mine[u] && !fits) atomicOr(flag, 1);
            }
            run += c[u];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry += s_v[1023];
        __syncthreads();
    }
}

void tk_launch_shard_positions(const int64_t *probes, const int *slot_prefix, int S, int64_t nq,
                               int64_t n_lists, const int *owner, int me, int W, int64_t qh,
                               int64_t C, int *spos, int *rpos, int *flag, hipStream_t s)
{
    if (nq == 0 || S == 0) return;
    hipLaunchKernelGGL(shard_positions_kernel, dim3(2 * W), dim3(1024), 0, s, probes, slot_prefix,
                       S, nq, n_lists, owner, me, W, qh, C, spos, rpos, flag);
}

// (query, slot) records of the lists this rank owns, grouped by list, for the list-major
// scan; the record's row offset is the segment's position in the send buffer.  Segments
// that did not fit become padding records.
__global__ void shard_pairs_fill_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                        int64_t n_lists, const int *__restrict__ owner, int me,
                                        const int *__restrict__ spos,
                                        const int *__restrict__ pair_off, int *__restrict__ cursor,
                                        int *__restrict__ pair_q, int *__restrict__ pair_f0)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * S) return;
    int64_t cl = probes[i];
    if (cl < 0) cl += n_lists;
    if (owner[cl] != me) return;
    const int pos = atomicAdd(&cursor[cl], 1);
    const int p = spos[i];
    pair_q[pair_off[cl] + pos] = p >= 0 ? (int)(i / S) : -1;
    pair_f0[pair_off[cl] + pos] = p >= 0 ? p : 0;
}

void tk_launch_shard_pairs_fill(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                const int *owner, int me, const int *spos, const int *pair_off,
                                int *cursor, int *pair_q, int *pair_f0, hipStream_t s)
{
    const int64_t np = nq * S;
    if (np == 0) return;
    hipLaunchKernelGGL(shard_pairs_fill_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s,
                       probes, S, nq, n_lists, owner, me, spos, pair_off, cursor, pair_q, pair_f0);
}

// Received segments -> the home queries' distance rows (the layout the replay kernels
// read) + the per-chunk minima the scan kernels would have written.
// One 64-lane workgroup per (home query, slot).
template <bool SIGNED>
__global__ __launch_bounds__(64) void shard_unpack_kernel(
    const uint4 *__restrict__ recv, const int *__restrict__ rpos,
    const int *__restrict__ slot_prefix, int S, uint4 *__restrict__ dist, int64_t cap,
    uint8_t *__restrict__ mins, int64_t min_stride)
{
    const int64_t b = blockIdx.x;
    const int64_t qi = b / S;
    const int s = (int)(b - qi * S);
    const int f0 = slot_prefix[qi * (S + 1) + s];
    const int n = slot_prefix[qi * (S + 1) + s + 1] - f0;
    const int p = rpos[b];
    for (int c = threadIdx.x; c < n; c += 64) {
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
    const unsigned grid = (unsigned)(nq_home * S);
    if (signd)
        hipLaunchKernelGGL(shard_unpack_kernel<true>, dim3(grid), dim3(64), 0, s, recv, rpos,
                           slot_prefix, S, dist, cap, mins, min_stride);
    else
        hipLaunchKernelGGL(shard_unpack_kernel<false>, dim3(grid), dim3(64), 0, s, recv, rpos,
                           slot_prefix, S, dist, cap, mins, min_stride);
}
