// rescore.hip — exact second pass on gfx950.
//
// Replaces knn_brute1 + bottom_k (utils.py:89-92, 22-25) as used by
// _FastDistanceTable.top (fast_pq.py:307-312, coarse stage) and IVF.query
// (ivf.py:154-163, final stage), and turns the coarse result into the per-slot
// scan descriptors of the list scan.
//
// Distances are float32 and restate numpy's einsum("ij,ij->i") rounding (SSE3
// baseline: 4 lanes, un-fused multiply-add, 16-element groups folded 3,2,1,0,
// zero-filled tail vector, (l0+l1)+(l2+l3)); built with -ffp-contract=off.  One
// lane owns one candidate row and carries the 4 lane-accumulators, reading its
// row with 16-byte loads.  The k best are returned in ascending distance (ties:
// lower candidate position first) — the order numpy's argpartition yields on the
// fixture host for these sizes; a rank-by-counting pass in LDS does the ordering.
#include "kernels.h"

__device__ __forceinline__ float sqdist_row(const float *__restrict__ y, const float *xs, int d)
{
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; d - i >= 16; i += 16) {
        float df[16];
#pragma unroll
        for (int t = 0; t < 16; t++) df[t] = y[i + t] - xs[i + t];
#pragma unroll
        for (int l = 0; l < 4; l++) {
            float ab3 = df[12 + l] * df[12 + l] + acc[l];
            float ab2 = df[8 + l] * df[8 + l] + ab3;
            float ab1 = df[4 + l] * df[4 + l] + ab2;
            acc[l] = df[l] * df[l] + ab1;
        }
    }
    for (; i < d; i += 4) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            float df = (i + l < d) ? (y[i + l] - xs[i + l]) : 0.f;
            acc[l] = df * df + acc[l];
        }
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// LDS: x[d] | cand[R] (int64) | dist[R] (float)
__global__ __launch_bounds__(128) void rescore_kernel(const float *__restrict__ q, int d,
                                                      const float *__restrict__ rows,
                                                      int64_t n_rows,
                                                      const int64_t *__restrict__ cand, int R,
                                                      int k, int strip, int64_t *__restrict__ out,
                                                      int *__restrict__ out_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t *cs = (int64_t *)smem;
    float *ds = (float *)(smem + (size_t)R * 8);
    float *xs = ds + R;
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int64_t *c = cand + qi * R;

    for (int t = tid; t < d; t += blockDim.x) xs[t] = q[qi * d + t];
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
        ds[t] = sqdist_row(rows + id * (int64_t)d, xs, d);
    }
    __syncthreads();
    for (int t = tid; t < nc; t += blockDim.x) {
        const float dv = ds[t];
        int rank = 0;
        for (int u = 0; u < nc; u++) {
            const float du = ds[u];
            rank += (du < dv) || (du == dv && u < t);
        }
        if (rank < k) o[rank] = cs[t];
    }
    if (tid == 0 && out_count) out_count[qi] = k;
}

void tk_launch_rescore(const float *q, int d, const float *rows, int64_t n_rows,
                       const int64_t *cand, int R, int64_t nq, int k, int strip, int64_t *out,
                       int *out_count, hipStream_t s)
{
    if (nq == 0 || k == 0) return;
    size_t lds = (size_t)R * 12 + (size_t)d * 4;
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)nq), dim3(128), lds, s, q, d, rows, n_rows,
                       cand, R, k, strip, out, out_count);
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
                                  unsigned char *__restrict__ repeat_flag)
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
    }
    if (repeat_flag) repeat_flag[qi] = wrapped;
}

void tk_launch_make_slots(const int64_t *probes, const int *probe_count, int kc, int64_t nq,
                          int64_t n_lists, const int64_t *list_chunk_off, const int64_t *list_n,
                          const int64_t *ids_off, int *slot_prefix, int64_t *slot_chunk0,
                          int *slot_n, int64_t *slot_label_off, unsigned char *repeat_flag,
                          hipStream_t s)
{
    (void)probe_count;
    if (nq == 0) return;
    hipLaunchKernelGGL(make_slots_kernel, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, s,
                       probes, kc, nq, n_lists, list_chunk_off, list_n, ids_off, slot_prefix,
                       slot_chunk0, slot_n, slot_label_off, repeat_flag);
}
