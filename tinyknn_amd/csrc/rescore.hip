// rescore.hip — exact second pass on gfx950.
//
// Replaces knn_brute1 + bottom_k (utils.py:89-92, 22-25) as used by
// _FastDistanceTable.top (fast_pq.py:307-312, coarse stage) and IVF.query
// (ivf.py:154-163, final stage), and turns the coarse result into the per-slot
// scan descriptors of the list scan.
//
// Distances restate numpy's einsum("ij,ij->i") rounding (SSE3 baseline: 16-byte
// vectors = 4 float32 or 2 float64 lanes, un-fused multiply-add, 4-vector groups
// folded 3,2,1,0, zero-filled tail vector, pairwise horizontal add); float32 when
// both the vectors and the query are float32, float64 otherwise (numpy's
// promotion of `Y - x`); built with -ffp-contract=off.  One lane owns one
// candidate row and carries the lane-accumulators.  The k best are returned in ascending distance (ties:
// lower candidate position first) — the order numpy's argpartition yields on the
// fixture host for these sizes; a rank-by-counting pass in LDS does the ordering.
#include "kernels.h"

// squared distance of row y to the query xs in numpy's einsum order, computed in T
// (float when both operands are float32, else double as numpy promotes): L = 16 /
// sizeof(T) lane-accumulators, groups of 4 vectors folded 3,2,1,0, zero tail.
template <typename T, typename TY>
__device__ __forceinline__ T sqdist_row(const TY *__restrict__ y, const T *xs, int d)
{
    constexpr int L = 16 / (int)sizeof(T);
    T acc[L];
#pragma unroll
    for (int l = 0; l < L; l++) acc[l] = 0;
    int i = 0;
    for (; d - i >= 4 * L; i += 4 * L) {
        T df[4 * L];
#pragma unroll
        for (int t = 0; t < 4 * L; t++) df[t] = (T)y[i + t] - xs[i + t];
#pragma unroll
        for (int l = 0; l < L; l++) {
            T ab3 = df[3 * L + l] * df[3 * L + l] + acc[l];
            T ab2 = df[2 * L + l] * df[2 * L + l] + ab3;
            T ab1 = df[L + l] * df[L + l] + ab2;
            acc[l] = df[l] * df[l] + ab1;
        }
    }
    for (; i < d; i += L) {
#pragma unroll
        for (int l = 0; l < L; l++) {
            T df = (i + l < d) ? ((T)y[i + l] - xs[i + l]) : (T)0;
            acc[l] = df * df + acc[l];
        }
    }
    if (L == 4) return (acc[0] + acc[1]) + (acc[2 % L] + acc[3 % L]);
    return acc[0] + acc[1 % L];
}

// LDS: cand[R] (int64) | dist[R] (T) | x[d] (T)
template <typename T, typename TY, typename TX>
__global__ __launch_bounds__(128) void rescore_kernel(const TX *__restrict__ q, int d,
                                                      const TY *__restrict__ rows,
                                                      int64_t n_rows,
                                                      const int64_t *__restrict__ cand, int R,
                                                      int k, int strip, int64_t *__restrict__ out,
                                                      int *__restrict__ out_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t *cs = (int64_t *)smem;
    T *ds = (T *)(smem + (size_t)R * 8);
    T *xs = ds + R;
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int64_t *c = cand + qi * R;

    for (int t = tid; t < d; t += blockDim.x) xs[t] = (T)q[qi * d + t];
    // ordered compaction of the candidate ids (ivf.py:154-155 drops -1)
    if (tid < 64) {
        int base = 0;
        for (int t0 = 0; t0 < R; t0 += 64) {
            int t = t0 + tid;
            int64_t id = t < R ? c[t] : -1;
            bool keep = t < R && (!strip || id != -1);
            uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            int before = __builtin_popcountll(m & ((1ull << tid) - 1ull));
            if (keep) cs[base + before] = id;
            base += __builtin_popcountll(m);
        }
        if (tid == 0) s_count = base;
    }
    __syncthreads();
    const int nc = s_count;
    int64_t *o = out + qi * k;
    if (nc <= k) {  // ivf.py:158-159 / fast_pq.py:307-308: heap order, no rescoring
        for (int t = tid; t < k; t += blockDim.x) o[t] = t < nc ? cs[t] : -1;
        if (tid == 0 && out_count) out_count[qi] = nc;
        return;
    }
    for (int t = tid; t < nc; t += blockDim.x) {
        int64_t id = cs[t];
        if (id < 0) id += n_rows;  // numpy fancy indexing with a negative index
        if (strip & 0x100) id = 0;            // timing experiment (wrong results): one row for every lane
        else if (strip & 0x200) id = blockIdx.x * 128 + t;    // ... consecutive rows
        ds[t] = sqdist_row<T, TY>(rows + id * (int64_t)d, xs, d);
    }
    __syncthreads();
    for (int t = tid; t < nc; t += blockDim.x) {
        const T dv = ds[t];
        int rank = 0;
        for (int u = 0; u < nc; u++) {
            const T du = ds[u];
            rank += (du < dv) || (du == dv && u < t);
        }
        if (rank < k) o[rank] = cs[t];
    }
    if (tid == 0 && out_count) out_count[qi] = k;
}

void tk_launch_rescore(const void *q, int q_is_f64, int d, const void *rows, int rows_is_f64,
                       int64_t n_rows, const int64_t *cand, int R, int64_t nq, int k, int strip,
                       int64_t *out, int *out_count, hipStream_t s)
{
    if (nq == 0 || k == 0) return;
    const bool dbl = q_is_f64 || rows_is_f64;   // numpy promotes `Y - x` to float64
    size_t lds = (size_t)R * 8 + ((size_t)R + d) * (dbl ? 8 : 4) + 16;
    dim3 grid((unsigned)nq), block(128);
    if (!dbl)
        hipLaunchKernelGGL((rescore_kernel<float, float, float>), grid, block, lds, s,
                           (const float *)q, d, (const float *)rows, n_rows, cand, R, k, strip, out,
                           out_count);
    else if (rows_is_f64 && q_is_f64)
        hipLaunchKernelGGL((rescore_kernel<double, double, double>), grid, block, lds, s,
                           (const double *)q, d, (const double *)rows, n_rows, cand, R, k, strip,
                           out, out_count);
    else if (rows_is_f64)
        hipLaunchKernelGGL((rescore_kernel<double, double, float>), grid, block, lds, s,
                           (const float *)q, d, (const double *)rows, n_rows, cand, R, k, strip, out,
                           out_count);
    else
        hipLaunchKernelGGL((rescore_kernel<double, float, double>), grid, block, lds, s,
                           (const double *)q, d, (const float *)rows, n_rows, cand, R, k, strip, out,
                           out_count);
}

// ---------------------------------------------------------------------------
// probes (nq, S) -> scan descriptors.  A negative list id (an unfilled coarse heap
// slot, fast_pq.py:307-312 does not filter them) addresses from the end, as the
// reference's Python list indexing does (ivf.py:141).
__global__ void make_slots_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                  int64_t n_lists, const int64_t *__restrict__ list_chunk_off,
                                  const int64_t *__restrict__ list_n,
                                  const int64_t *__restrict__ ids_off, int *__restrict__ slot_prefix,
                                  int64_t *__restrict__ slot_chunk0, int *__restrict__ slot_n,
                                  int64_t *__restrict__ slot_label_off,
                                  unsigned char *__restrict__ repeat_flag,
                                  int *__restrict__ pair_count, const int *__restrict__ owner,
                                  int me)
{
    int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int acc = 0;
    bool wrapped = false;  // a wrapped id may name a list that is probed twice
    slot_prefix[qi * (S + 1)] = 0;
    for (int s = 0; s < S; s++) {
        int64_t cl = probes[qi * S + s];
        if (cl < 0) { cl += n_lists; wrapped = true; }
        int64_t c0 = list_chunk_off[cl];
        acc += (int)(list_chunk_off[cl + 1] - c0);
        slot_prefix[qi * (S + 1) + s + 1] = acc;
        slot_chunk0[qi * S + s] = c0;
        slot_n[qi * S + s] = (int)list_n[cl];
        slot_label_off[qi * S + s] = ids_off[cl];
        // pairs per list, for the list-major scan (sharded index: of the lists I own)
        if (pair_count && (!owner || owner[cl] == me)) atomicAdd(&pair_count[cl], 1);
    }
    if (repeat_flag) repeat_flag[qi] = wrapped;
}

void tk_launch_make_slots(const int64_t *probes, const int *probe_count, int kc, int64_t nq,
                          int64_t n_lists, const int64_t *list_chunk_off, const int64_t *list_n,
                          const int64_t *ids_off, int *slot_prefix, int64_t *slot_chunk0,
                          int *slot_n, int64_t *slot_label_off, unsigned char *repeat_flag,
                          int *pair_count, const int *owner, int me, hipStream_t s)
{
    (void)probe_count;
    if (nq == 0) return;
    hipLaunchKernelGGL(make_slots_kernel, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, s,
                       probes, kc, nq, n_lists, list_chunk_off, list_n, ids_off, slot_prefix,
                       slot_chunk0, slot_n, slot_label_off, repeat_flag, pair_count, owner, me);
}
