// twins.hip — for every stored row, where the OTHER copies of its label are stored.
//
// IVF.build(n_probes = b >= 2) (ivf.py:53, :77-102: knn_brute(data, all_centers, k=n_probes) ->
// group_data_by_indices) puts every row into b lists.  The reference's `insert` finds a label's second
// arrival by scanning the heap's labels (_fast_pq.pyx:284-287); the TWIN form of the lane replay
// (heap.hip) decides the same test from the positions of a row's earlier copies, which it gets from
// this table: twin_list[i * w + u] / twin_off[i * w + u] = list and offset inside that list of the u-th
// other copy of flat row i (flat = the index into the concatenated ids), -1 where a label has fewer
// copies.  w = (largest number of copies of any label) - 1.
//
// Built on the device from the int32 labels (host upload and device build alike): a count per label,
// one slot per copy handed out by an atomic cursor, then every row lists the other slots of its label.
#include "kernels.h"

__global__ void twin_count_kernel(const int32_t *__restrict__ ids32, int64_t T, int *__restrict__ cnt,
                                  int *__restrict__ cnt_max)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const int c = atomicAdd(&cnt[ids32[i]], 1) + 1;
    // (the last arrival of a label sees its full count)
    if (c > 1) atomicMax(cnt_max, c);
}

__global__ void twin_where_kernel(const int32_t *__restrict__ ids32, int64_t T, int *__restrict__ cursor,
                                  int *__restrict__ where, int b)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const int64_t L = ids32[i];
    const int t = atomicAdd(&cursor[L], 1);
    where[L * b + t] = (int)i;
}

__global__ void twin_fill_kernel(const int32_t *__restrict__ ids32, int64_t T, const int *__restrict__ cnt,
                                 const int *__restrict__ where, int b, const int64_t *__restrict__ ids_off,
                                 int n_lists, int32_t *__restrict__ twin_list, int32_t *__restrict__ twin_off)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const int64_t L = ids32[i];
    const int c = cnt[L];
    const int w = b - 1;
    int k = 0;
    for (int t = 0; t < c; t++) {
        const int j = where[L * b + t];
        if (j == (int)i) continue;
        int lo = 0, hi = n_lists;            // ids_off[lo] <= j < ids_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ids_off[mid] <= (int64_t)j) lo = mid; else hi = mid;
        }
        twin_list[i * w + k] = lo;
        twin_off[i * w + k] = (int32_t)((int64_t)j - ids_off[lo]);
        k++;
    }
    for (; k < w; k++) twin_list[i * w + k] = twin_off[i * w + k] = -1;
}

// What the TWIN form rests on, checked instead of assumed (an index may come from anywhere: tk_index_set_lists takes
// any ids and codes): the copies of a label lie in DIFFERENT lists (flag bit 1 otherwise) and carry the SAME code
// (bit 0 otherwise) — what IVF.build's lists have by construction (ivf.py:77-102: a list's codes are the codes of
// data[ids]).  One thread per stored row: its own (list, offset) by a search of ids_off, then M/2 bytes of its
// code against every other copy's.  codes: the tiled array of ALL lists (kernels.h), list_chunk_off its chunks.
__global__ void twin_verify_kernel(const uint4 *__restrict__ codes, int P, const int64_t *__restrict__ list_chunk_off,
                                   const int64_t *__restrict__ ids_off, int n_lists,
                                   const int32_t *__restrict__ twin_list, const int32_t *__restrict__ twin_off, int w,
                                   int64_t T, int *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    int lo = 0, hi = n_lists;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ids_off[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t ga = list_chunk_off[lo] + ((i - ids_off[lo]) >> 4);     // the row's chunk in the whole array
    const int ra = (int)((i - ids_off[lo]) & 15);
    int bad = 0;
    for (int u = 0; u < w; u++) {
        const int l2 = twin_list[i * w + u];
        if (l2 < 0) break;
        if (l2 == lo) { bad |= 2; continue; }
        const int64_t o2 = twin_off[i * w + u];
        const int64_t gb = list_chunk_off[l2] + (o2 >> 4);
        const int rb = (int)(o2 & 15);
        for (int p = 0; p < P; p++) {
            const uint8_t *a = (const uint8_t *)&codes[(ga >> 3) * (int64_t)(8 * P) + p * 8 + (ga & 7)];
            const uint8_t *b = (const uint8_t *)&codes[(gb >> 3) * (int64_t)(8 * P) + p * 8 + (gb & 7)];
            if (a[ra] != b[rb]) bad |= 1;
        }
    }
    if (bad) atomicOr(flag, bad);
}

void tk_launch_twin_verify(const uint4 *codes, int P, const int64_t *list_chunk_off, const int64_t *ids_off, int n_lists,
                           const int32_t *twin_list, const int32_t *twin_off, int w, int64_t T, int *flag, hipStream_t s)
{
    if (T == 0) return;
    hipLaunchKernelGGL(twin_verify_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, codes, P, list_chunk_off,
                       ids_off, n_lists, twin_list, twin_off, w, T, flag);
}

// cnt: label_bound ints, zeroed; returns the largest count through *cnt_max (device int, zeroed)
void tk_launch_twin_count(const int32_t *ids32, int64_t T, int *cnt, int *cnt_max, hipStream_t s)
{
    if (T == 0) return;
    hipLaunchKernelGGL(twin_count_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, ids32, T, cnt, cnt_max);
}

// cursor: label_bound ints, zeroed (it holds the counts again afterwards); where: label_bound * b ints
void tk_launch_twin_fill(const int32_t *ids32, int64_t T, int *cursor, int *where, int b, const int64_t *ids_off,
                         int n_lists, int32_t *twin_list, int32_t *twin_off, hipStream_t s)
{
    if (T == 0) return;
    const dim3 grid((unsigned)((T + 255) / 256)), block(256);
    hipLaunchKernelGGL(twin_where_kernel, grid, block, 0, s, ids32, T, cursor, where, b);
    hipLaunchKernelGGL(twin_fill_kernel, grid, block, 0, s, ids32, T, cursor, where, b, ids_off, n_lists, twin_list,
                       twin_off);
}
