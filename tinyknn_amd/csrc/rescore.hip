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
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

// (tried: nontemporal loads of the candidate rows, so that they would not evict the scan
// kernels' code and table lines from L2 — 2.5x slower alone, 0.135 -> 0.332 ms per 10 000
// queries, and the scan beside it 0.61 -> 0.79 ms: a lane fetches its row in 25 pieces of 16 B
// and an uncached piece is a memory transaction of its own.  -DTK_NT=1 rebuilds that.)
#ifndef TK_NT
#define TK_NT 0
#endif

// (tried: the 16-byte pieces of three groups requested before the first is used — 3 dependent
// trips to memory per 100-float row instead of 7, 77 VGPRs instead of 44: 0.138 -> 0.150 ms alone
// and the batch 0.650 -> 0.693 ms; the kernel's cost to its neighbours is the NUMBER of
// line visits, and bunching them makes it worse.  `#pragma unroll 4` alone changes nothing: the
// compiler keeps each group's loads behind the previous group's arithmetic.)
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
        for (int t = 0; t < 4 * L; t++) df[t] = (T)(TK_NT ? __builtin_nontemporal_load(y + i + t) : y[i + t]) - xs[i + t];
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
                                                      int *__restrict__ out_count, const TX *__restrict__ q_b,
                                                      int64_t q_na, int64_t *__restrict__ out_b, int64_t out_na)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t *cs = (int64_t *)smem;
    T *ds = (T *)(smem + (size_t)R * 8);
    T *xs = ds + R;
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int64_t *c = cand + qi * R;

    const TX *qrow = (q_b && qi >= q_na) ? q_b + (qi - q_na) * d : q + qi * d;   // second call of a pair
    for (int t = tid; t < d; t += blockDim.x) xs[t] = (T)qrow[t];
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
    int64_t *o = (out_b && qi >= out_na) ? out_b + (qi - out_na) * k : out + qi * k;
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

// sqdist_row<float, float> for a row held in LDS as 16-byte pieces (explicit 128-bit reads: with
// an odd piece stride the 64 lanes' reads are conflict-free; 32-bit reads of the same layout
// would collide four ways).  Same operations in the same order.
__device__ __forceinline__ float sqdist_row_lds(const float4 *__restrict__ y4, const float *xs, int d)
{
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; d - i >= 16; i += 16) {
        const float4 a = y4[(i >> 2)], b = y4[(i >> 2) + 1], c = y4[(i >> 2) + 2], e = y4[(i >> 2) + 3];
        const float yv[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, e.x, e.y, e.z, e.w};
        float df[16];
#pragma unroll
        for (int t = 0; t < 16; t++) df[t] = yv[t] - xs[i + t];
#pragma unroll
        for (int l = 0; l < 4; l++) {
            float ab3 = df[12 + l] * df[12 + l] + acc[l];
            float ab2 = df[8 + l] * df[8 + l] + ab3;
            float ab1 = df[4 + l] * df[4 + l] + ab2;
            acc[l] = df[l] * df[l] + ab1;
        }
    }
    for (; i < d; i += 4) {          // d % 4 == 0: whole vectors only
        const float4 a = y4[i >> 2];
        const float yv[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const float df = yv[l] - xs[i + l];
            acc[l] = df * df + acc[l];
        }
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// make_slots_kernel's work for ONE query by the wave that ranked its probed lists (S <= 64): lane s
// holds the id of slot s.  Same outputs, same atomics.
__device__ __forceinline__ void slots_epilogue(const TkSlotsOut &so, int64_t qi, int S, int lane, int64_t cl)
{
    const bool act = lane < S;
    bool wr = false;
    if (act && cl < 0) { cl += so.n_lists; wr = true; }
    const bool wrapped = __builtin_amdgcn_ballot_w64(wr) != 0;
    int64_t c0 = 0, rows = 0, loff = 0;
    int len = 0, n = 0;
    if (act) {
        c0 = so.list_chunk_off[cl];
        len = (int)(so.list_chunk_off[cl + 1] - c0);
        const int64_t ln = so.list_n[cl];
        n = (int)ln;
        rows = ln;
        loff = so.ids_off[cl];
    }
    int acc = len;              // inclusive scans over the slots
    int64_t racc = rows;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ua = __shfl_up(acc, o, 64);
        const int64_t ur = __shfl_up(racc, o, 64);
        if (lane >= o) { acc += ua; racc += ur; }
    }
    if (lane == 0) so.slot_prefix[qi * (S + 1)] = 0;
    if (act) {
        so.slot_prefix[qi * (S + 1) + lane + 1] = acc;
        so.slot_chunk0[qi * S + lane] = c0;
        so.slot_n[qi * S + lane] = n;
        so.slot_label_off[qi * S + lane] = loff;
    }
    // leading slots that stay exact: until the lists scanned so far hold R rows
    const uint64_t full = __builtin_amdgcn_ballot_w64(act && racc >= (int64_t)so.R);
    int e = full ? __builtin_ctzll(full) + 1 : S;
    const bool plain_ok = so.slot_exact && !wrapped && so.qlim[qi] != TK_PLAIN_NEVER;
    const int e_walk = e;
    if (!plain_ok) e = S;
    bool head = false;
    if (so.slot_exact) {
        const int E = (so.R + 15) >> 4;
        const int first_len = __shfl(acc, 0, 64);
        head = plain_ok && e_walk == 1 && first_len > E;
        const int at_e = e > 0 ? __shfl(acc, e - 1, 64) : 0;
        if (lane == 0) {
            so.slot_exact[qi] = head ? 0 : e;
            so.plain0[qi] = head ? E : at_e;
        }
    }
    if (so.pair_count && act && !(so.owner && so.owner[cl] != so.me)) {
        if (head && lane == 0) {
            atomicAdd(&so.pair_count3[cl], 1);
            atomicAdd(&so.pair_count2[cl], 1);
        } else {
            atomicAdd(lane < e ? &so.pair_count[cl] : &so.pair_count2[cl], 1);
        }
    }
    if (lane == 0 && so.repeat_flag) so.repeat_flag[qi] = wrapped;
}

// The same for float32 vectors and queries with d % 4 == 0, rows staged through LDS: in the
// kernel above a lane walks its own row, so every load instruction of a wave touches 64
// different lines — 2775 line visits per query at R = 111, d = 100 — and that address traffic
// is what this kernel costs the scan kernels it runs beside (profiles/r02_scan_grid.md).  Here
// the wave reads 64 candidate rows as one stream of 16-byte pieces, consecutive lanes =
// consecutive pieces of a row (4 lines per 400-byte row), drops them into a tile whose row
// stride is an odd number of 16-byte pieces (conflict-free ds_read_b128), and each lane then
// runs the identical summation (sqdist_row: numpy's order) on its row from LDS.
// LDS: cand[R] (int64) | dist[R] | x[d] | tile[64 + 1 spare][stride]
template <int TILE>
__global__ __launch_bounds__(64) void rescore_staged_kernel(const float *__restrict__ q, int d,
                                                            const float *__restrict__ rows,
                                                            int64_t n_rows,
                                                            const int64_t *__restrict__ cand, int R,
                                                            int k, int strip, int64_t *__restrict__ out,
                                                            int *__restrict__ out_count, int stride4,
                                                            const float *__restrict__ q_b, int64_t q_na,
                                                            int64_t *__restrict__ out_b, int64_t out_na,
                                                            const TkSlotsOut *__restrict__ so_dev)
{
    const bool have_slots = so_dev != nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t *cs = (int64_t *)smem;
    float *ds = (float *)(smem + (((size_t)R * 8 + 15) & ~(size_t)15));
    float *xs = ds + ((R + 3) & ~3);
    float4 *tile = (float4 *)(xs + ((d + 3) & ~3));      // 16-byte aligned
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int64_t *c = cand + qi * R;
    const float *qrow = (q_b && qi >= q_na) ? q_b + (qi - q_na) * d : q + qi * d;   // second call of a pair
    for (int t = tid; t < d; t += 64) xs[t] = qrow[t];
    // ordered compaction of the candidate ids (ivf.py:154-155 drops -1)
    int nc = 0;
    for (int t0 = 0; t0 < R; t0 += 64) {
        const int t = t0 + tid;
        const int64_t id = t < R ? c[t] : -1;
        const bool keep = t < R && (!strip || id != -1);
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        const int before = __builtin_popcountll(m & ((1ull << tid) - 1ull));
        if (keep) cs[nc + before] = id;
        nc += __builtin_popcountll(m);
    }
    __syncthreads();
    int64_t *o = (out_b && qi >= out_na) ? out_b + (qi - out_na) * k : out + qi * k;
    if (nc <= k) {  // ivf.py:158-159 / fast_pq.py:307-308: heap order, no rescoring
        for (int t = tid; t < k; t += 64) o[t] = t < nc ? cs[t] : -1;
        if (tid == 0 && out_count) out_count[qi] = nc;
        if (have_slots) slots_epilogue(*so_dev, qi, k, tid, tid < nc ? cs[tid < k ? tid : 0] : -1);
        return;
    }
    const int d4 = d >> 2;
    // lanes_per_row = the power of two >= d4 (16, 32 or 64): a load instruction covers
    // 64 / lanes_per_row whole rows, lane (rr, pc) fetches piece pc of its row
    const int lpr_log = d4 <= 16 ? 4 : (d4 <= 32 ? 5 : 6);
    const int rr = tid >> lpr_log, pc = tid & ((1 << lpr_log) - 1), rpi = 64 >> lpr_log;
    for (int t0 = 0; t0 < nc; t0 += TILE) {
        const int nt = nc - t0 < TILE ? nc - t0 : TILE;
        if (pc < d4)
            for (int r0 = rr; r0 < nt; r0 += 16 * rpi) {
                // sixteen loads in flight per lane before the first LDS store waits for one
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int r = r0 + u * rpi;
                    int64_t id = cs[t0 + (r < nt ? r : nt - 1)];
                    if (id < 0) id += n_rows;  // numpy fancy indexing with a negative index
                    v[u] = ((const float4 *)(rows + id * (int64_t)d))[pc];
                }
                // unconditional stores (rows past the tile go to a spare row): a branch here
                // lets the compiler sink each load next to its store, one exposed latency apiece
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int r = r0 + u * rpi;
                    tile[(r < nt ? r : TILE) * stride4 + pc] = v[u];
                }
            }
        __syncthreads();
        if (tid < nt) ds[t0 + tid] = sqdist_row_lds(tile + tid * stride4, xs, d);
        __syncthreads();
    }
    int64_t *ps = (int64_t *)(tile + (TILE + 1) * stride4);      // have_slots: the k results in order
    for (int t = tid; t < nc; t += 64) {
        const float dv = ds[t];
        int rank = 0;
        for (int u = 0; u < nc; u++) {
            const float du = ds[u];
            rank += (du < dv) || (du == dv && u < t);
        }
        if (rank < k) {
            o[rank] = cs[t];
            if (have_slots) ps[rank] = cs[t];
        }
    }
    if (tid == 0 && out_count) out_count[qi] = k;
    if (have_slots) {
        __syncthreads();
        slots_epilogue(*so_dev, qi, k, tid, tid < k ? ps[tid] : -1);
    }
}

int tk_launch_rescore(const void *q, int q_is_f64, int d, const void *rows, int rows_is_f64,
                      int64_t n_rows, const int64_t *cand, int R, int64_t nq, int k, int strip,
                      int64_t *out, int *out_count, hipStream_t s, int form, TkSecond q2, TkSecond out2,
                      const TkSlotsOut *slots)
{
    if (nq == 0 || k == 0) return 0;
    const bool dbl = q_is_f64 || rows_is_f64;   // numpy promotes `Y - x` to float64
    size_t lds = (size_t)R * 8 + ((size_t)R + d) * (dbl ? 8 : 4) + 16;
    // one wave is enough for the coarse stage's 2 * n_probes + 10 candidates: the second wave of
    // a 128-thread workgroup only waited at the barriers and took a wave slot beside the scan
    dim3 grid((unsigned)nq), block(R <= 64 ? 64 : 128);
    // Rows staged through LDS in tiles of 32 (form 2, DEFAULT) / 64 (form 1) rows, or every lane
    // walking its own row (form 0); tk_index_set_option(TK_OPT_RESCORE_FORM).  Round 2
    // (profiles/r02_scan_grid.md): the 32-row form was the fastest alone (0.118 ms per 10 000
    // queries against 0.135) but lost in the pipeline — its workgroups (15-29 KB of LDS) were
    // placed late next to the exact scan's persistent grid: 0.68-0.70 ms per batch against 0.656.
    // Round 3, next to the plain kernel (two 58 KB workgroups per CU leave 44 KB): 0.495 ms per
    // batch against 0.543 (profiles/r03/ab_pipeline_knobs.txt) — the 64 lines a wave touches per
    // load in mode 0 were costing the scans beside it more than its own time.
    const int staged = form < 0 || form > 2 ? 2 : form;
    // (heaps beyond 256 entries: a 64-lane workgroup walks 16+ tiles one after the other — n_probes 50,
    //  R = 511: 3.18 M queries/s staged against 3.85 M with the 128-lane lane-per-row kernel)
    if (!dbl && staged && d % 4 == 0 && d <= 256 && R <= 256 ) {
        const int stride4 = (d / 4) | 1;                      // odd number of 16-byte pieces
        const int tile_rows = staged == 2 ? 32 : 64;
        const bool fuse = slots != nullptr && k <= 64;
        const TkSlotsOut *so = fuse ? slots : nullptr;      // (a DEVICE copy of the structure)
        const size_t slds = (((size_t)R * 8 + 15) & ~(size_t)15) +
                            (size_t)(((R + 3) & ~3) + ((d + 3) & ~3)) * 4 + (size_t)(tile_rows + 1) * stride4 * 16 +
                            (fuse ? (size_t)k * 8 : 0);
        if (slds <= 64 * 1024) {
            if (tile_rows == 32)
                hipLaunchKernelGGL(rescore_staged_kernel<32>, grid, dim3(64), slds, s, (const float *)q, d,
                                   (const float *)rows, n_rows, cand, R, k, strip, out, out_count, stride4,
                                   (const float *)q2.b, q2.n_a, (int64_t *)out2.b, out2.n_a, so);
            else
                hipLaunchKernelGGL(rescore_staged_kernel<64>, grid, dim3(64), slds, s, (const float *)q, d,
                                   (const float *)rows, n_rows, cand, R, k, strip, out, out_count, stride4,
                                   (const float *)q2.b, q2.n_a, (int64_t *)out2.b, out2.n_a, so);
            return fuse ? 1 : 0;
        }
    }
    if (!dbl)
        hipLaunchKernelGGL((rescore_kernel<float, float, float>), grid, block, lds, s,
                           (const float *)q, d, (const float *)rows, n_rows, cand, R, k, strip, out,
                           out_count, (const float *)q2.b, q2.n_a, (int64_t *)out2.b, out2.n_a);
    else if (rows_is_f64 && q_is_f64)
        hipLaunchKernelGGL((rescore_kernel<double, double, double>), grid, block, lds, s,
                           (const double *)q, d, (const double *)rows, n_rows, cand, R, k, strip,
                           out, out_count, (const double *)q2.b, q2.n_a, (int64_t *)out2.b, out2.n_a);
    else if (rows_is_f64)
        hipLaunchKernelGGL((rescore_kernel<double, double, float>), grid, block, lds, s,
                           (const float *)q, d, (const double *)rows, n_rows, cand, R, k, strip, out,
                           out_count, (const float *)q2.b, q2.n_a, (int64_t *)out2.b, out2.n_a);
    else
        hipLaunchKernelGGL((rescore_kernel<double, float, double>), grid, block, lds, s,
                           (const double *)q, d, (const float *)rows, n_rows, cand, R, k, strip, out,
                           out_count, (const double *)q2.b, q2.n_a, (int64_t *)out2.b, out2.n_a);
    return 0;
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
                                  int me, const int *__restrict__ qlim, int R,
                                  int *__restrict__ slot_exact, int *__restrict__ pair_count2,
                                  int *__restrict__ plain0, int *__restrict__ pair_count3)
{
    int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int acc = 0;
    bool wrapped = false;  // a wrapped id may name a list that is probed twice
    slot_prefix[qi * (S + 1)] = 0;
    // leading slots that stay exact: until the lists scanned so far hold 2R rows (the heap is then
    // full of real values and its bound far below the table's limit; the replay checks)
    int e = S;
    int64_t rows = 0;
    for (int s = 0; s < S; s++) {
        int64_t cl = probes[qi * S + s];
        if (cl < 0) { cl += n_lists; wrapped = true; }
        int64_t c0 = list_chunk_off[cl];
        acc += (int)(list_chunk_off[cl + 1] - c0);
        slot_prefix[qi * (S + 1) + s + 1] = acc;
        slot_chunk0[qi * S + s] = c0;
        slot_n[qi * S + s] = (int)list_n[cl];
        slot_label_off[qi * S + s] = ids_off[cl];
        rows += list_n[cl];
        if (e == S && rows >= (int64_t)R) e = s + 1;        // (R here: the rows the exact kernel keeps)
    }
    const bool plain_ok = slot_exact && !wrapped && qlim[qi] != TK_PLAIN_NEVER;
    const int e_walk = e;
    if (!plain_ok) e = S;
    bool head = false;
    if (slot_exact) {
        // head mode: the first list alone holds 2R rows and is longer than the head
        const int E = (R + 15) >> 4;
        head = plain_ok && e_walk == 1 && slot_prefix[qi * (S + 1) + 1] > E;
        slot_exact[qi] = head ? 0 : e;
        plain0[qi] = head ? E : slot_prefix[qi * (S + 1) + e];
    }
    if (pair_count)
        for (int s = 0; s < S; s++) {
            int64_t cl = probes[qi * S + s];
            if (cl < 0) cl += n_lists;
            // pairs per list, for the list-major scan (sharded index: of the lists I own)
            if (owner && owner[cl] != me) continue;
            if (head && s == 0) {
                atomicAdd(&pair_count3[cl], 1);
                atomicAdd(&pair_count2[cl], 1);
            } else {
                atomicAdd(s < e ? &pair_count[cl] : &pair_count2[cl], 1);
            }
        }
    if (repeat_flag) repeat_flag[qi] = wrapped;
}

// The same with one WAVE per query, lane s = slot s (S <= 64): the thread-per-query form walks its S slots one
// dependent load after the other — 76 us for the 80 000 queries x 10 slots of a list-sharded batch at W = 8,
// every rank doing all of it; here the slots of a query are loaded at once and scanned by shuffles.
__global__ __launch_bounds__(256) void make_slots_wave_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                                              const TkSlotsOut so)
{
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;       // (whole waves leave: the ballots below see full waves only)
    slots_epilogue(so, qi, S, lane, lane < S ? probes[qi * S + lane] : -1);
}

void tk_launch_make_slots(const int64_t *probes, const int *probe_count, int kc, int64_t nq,
                          int64_t n_lists, const int64_t *list_chunk_off, const int64_t *list_n,
                          const int64_t *ids_off, int *slot_prefix, int64_t *slot_chunk0,
                          int *slot_n, int64_t *slot_label_off, unsigned char *repeat_flag,
                          int *pair_count, const int *owner, int me, hipStream_t s,
                          const int *qlim, int R, int *slot_exact, int *pair_count2, int *plain0,
                          int *pair_count3)
{
    (void)probe_count;
    if (nq == 0) return;
    if (!qlim || !pair_count2 || !plain0 || !pair_count3) slot_exact = nullptr;
    if (kc <= 64 && nq >= 4096) {
        TkSlotsOut so;
        so.n_lists = n_lists; so.list_chunk_off = list_chunk_off; so.list_n = list_n; so.ids_off = ids_off;
        so.slot_prefix = slot_prefix; so.slot_chunk0 = slot_chunk0; so.slot_n = slot_n; so.slot_label_off = slot_label_off;
        so.repeat_flag = repeat_flag; so.pair_count = pair_count; so.owner = owner; so.me = me; so.qlim = qlim; so.R = R;
        so.slot_exact = slot_exact; so.pair_count2 = pair_count2; so.plain0 = plain0; so.pair_count3 = pair_count3;
        hipLaunchKernelGGL(make_slots_wave_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, probes, kc, nq, so);
        return;
    }
    hipLaunchKernelGGL(make_slots_kernel, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, s,
                       probes, kc, nq, n_lists, list_chunk_off, list_n, ids_off, slot_prefix,
                       slot_chunk0, slot_n, slot_label_off, repeat_flag, pair_count, owner, me,
                       qlim, R, slot_exact, pair_count2, plain0, pair_count3);
}
