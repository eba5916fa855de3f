// brute.hip — exact k nearest rows, the ground truth of every recall figure
// (knn_brute, utils.py:66-86, as the reference's benches use it: examples/bench.py:85);
// the one dense contraction around the path.
//
//   part[i][j] = (|x_i|^2 + |y_j|^2) - (2 x_i) . y_j          utils.py:84
// on the f32 matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit the f32 FMA chain over k
// ascending (MI355X guide, "FP32-input MFMA"), i.e. what OpenBLAS returns for (2X) @ Y.T, and
// the norms use numpy's einsum order (np_order.h) — the part values are numpy's.
// Selection of the k smallest without data-dependent loops:
//   pass A  part values of `ns` rows of Y taken at a constant stride over the whole matrix
//           (not its first rows: cluster-ordered data would give a useless bound) -> (nq, ns)
//   sort    per query: k-th smallest of them = tau_q           (bitonic sort in LDS)
//   pass B  the rows of Y in segments of 2^20: every (part, j) with part <= tau_q is appended
//           to the query's candidate list (atomic counter, capacity `cap`); after each
//           segment the list is sorted, cut to its k best and tau_q tightened to the k-th of
//           them, so a segment adds about k * 2^20 / max(ns, rows seen) entries whatever N is
//   result  the k best after the last segment, by (part, j) ascending.  Ties: lower j first.
// One workgroup = 4 waves = 128 query rows against 32 rows of Y at a time; the Y tile (32
// consecutive rows = one contiguous block of memory) is fetched once per workgroup into LDS
// in MFMA operand order, double-buffered.
#include "kernels.h"
#include "np_order.h"

typedef float tk_f32x16 __attribute__((ext_vector_type(16)));

// |y_j|^2 in numpy's einsum order
__global__ void row_norms_kernel(const float *__restrict__ Y, int64_t n, int d, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = einsum_selfdot<float>(Y + i * d, d);
}

// MODE 0: write part values of rows [0, ns) of Y to vals (nq, ns).
// MODE 1: append (part, j) with part <= tau[row] to cand (nq, cap) / count (nq,).
template <int MODE>
__global__ __launch_bounds__(256) void brute_tiles_kernel(
    const float *__restrict__ X, int64_t nq, int d, const float *__restrict__ Y,
    const float *__restrict__ ynorm2, int64_t N, float *__restrict__ vals, int64_t ns,
    const float *__restrict__ tau, unsigned long long *__restrict__ cand, int cap,
    int *__restrict__ count, int64_t j_base)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    const int64_t my_row = r0 + col < nq ? r0 + col : nq - 1;
    const int KT = (d + 1) >> 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *a = (float *)smem + (size_t)(threadIdx.x >> 6) * KT * 64;     // A operands [t][lane]
    float *bt = (float *)smem + (size_t)4 * KT * 64;                      // B tiles [2][t][lane]
    for (int t = 0; t < KT; t++) {
        const int kk = 2 * t + half;
        a[t * 64 + lane] = kk < d ? 2.0f * X[my_row * d + kk] : 0.0f;
    }
    float xn[16], tq[16];
#pragma unroll
    for (int v = 0; v < 16; v++) {
        int64_t row = r0 + (v & 3) + 8 * (v >> 2) + 4 * half;
        row = row < nq ? row : nq - 1;
        xn[v] = einsum_selfdot<float>(X + row * d, d);
        tq[v] = MODE == 1 ? tau[row] : 0.0f;
    }
    // blockIdx.y splits the rows of Y: few queries (a recall sample) would otherwise occupy a
    // handful of CUs; appends are atomic, so the split does not change the result
    const int64_t all_cols = MODE == 0 ? ns : N;
    const int64_t per_y = ((all_cols + gridDim.y - 1) / gridDim.y + 31) / 32 * 32;
    const int64_t c_begin = (int64_t)blockIdx.y * per_y;
    const int64_t ncols = c_begin + per_y < all_cols ? c_begin + per_y : all_cols;
    // B tile of rows [j0, j0+32): element (t, l) = Y[j0 + (l & 31)][2t + (l >> 5)].  The 32
    // rows are contiguous in memory: thread e reads float e of the block (coalesced) and
    // drops it at its operand position.
    const int tile_f = 32 * d;
    constexpr int PER = 32 * 128 / 256;        // floats per thread and tile at d <= 128
    float stage[PER];
    auto fetch = [&](int64_t j0) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = u * 256 + (int)threadIdx.x;
            // rows past the range read as zeros: element e belongs to row j0 + e / d, so it is
            // in range iff e < (ncols - j0) * d
            float val = 0.0f;
            if (e < tile_f && (int64_t)e < (ncols - j0) * d) val = Y[j0 * d + e];
            stage[u] = val;
        }
    };
    // operand position of this thread's tile elements (the same for every tile)
    int dst[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int e = u * 256 + (int)threadIdx.x;
        const int jj = e / d, kk = e - jj * d;
        dst[u] = e < tile_f ? (kk >> 1) * 64 + (kk & 1) * 32 + jj : -1;
    }
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (dst[u] >= 0) bt[(size_t)buf * KT * 64 + dst[u]] = stage[u];
    };
    if (d & 1)      // odd d: the k = d operands of the upper half-wave are zero in both buffers
        for (int e = threadIdx.x; e < 2 * 32; e += 256)
            bt[(size_t)(e >> 5) * KT * 64 + (KT - 1) * 64 + 32 + (e & 31)] = 0.0f;
    if (c_begin < ncols) {
        fetch(c_begin);
        commit(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = c_begin; j0 < ncols; j0 += 32, buf ^= 1) {
        const bool more = j0 + 32 < ncols;
        if (more) fetch(j0 + 32);
        const int64_t j = j0 + col;
        tk_f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; v++) acc[v] = 0.0f;
        const float *bb = bt + (size_t)buf * KT * 64;
#pragma unroll 4
        for (int t = 0; t < KT; t++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t * 64 + lane], bb[t * 64 + lane], acc, 0, 0, 0);
        if (j < ncols) {
            const float yn = ynorm2[j];
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int64_t row = r0 + (v & 3) + 8 * (v >> 2) + 4 * half;
                const float part = (xn[v] + yn) - acc[v];
                if (row < nq) {
                    if (MODE == 0) {
                        vals[row * ns + j] = part;
                    } else if (part <= tq[v]) {
                        const int pos = atomicAdd(&count[row], 1);
                        if (pos < cap) {
                            // order-preserving key: float bits made monotone, then the index
                            uint32_t u = __float_as_uint(part);
                            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                            cand[row * (int64_t)cap + pos] = ((unsigned long long)u << 32) | (uint32_t)(j_base + j);
                        }
                    }
                }
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
}

// Bitonic sort of up to TK_BR_SORT 64-bit keys per query in LDS (one workgroup per query).
// MODE 0: keys = part values of the sample (index ignored) -> tau[q] = k-th smallest.
// MODE 1: keys = candidates (part, j) -> out[q][0..k) = j of the k smallest; the k smallest
//         stay at the head of the query's list (count = min(k, n)) and tau[q] becomes the
//         k-th part value once k candidates exist: the next segment of rows appends to that.
#define TK_BR_SORT 8192
template <int MODE>
__global__ __launch_bounds__(1024) void brute_select_kernel(
    const float *__restrict__ vals, int64_t ns, unsigned long long *__restrict__ cand, int cap,
    int *__restrict__ count, int k, float *__restrict__ tau, int64_t *__restrict__ out,
    int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *key = (unsigned long long *)smem;
    const int64_t q = blockIdx.x;
    int n = MODE == 0 ? (int)ns : count[q];
    if (MODE == 1 && n > cap) {
        if (threadIdx.x == 0) atomicOr(overflow, 1);
        n = cap;
    }
    int P = 1;
    while (P < n) P <<= 1;
    for (int e = threadIdx.x; e < P; e += 1024) {
        unsigned long long kk = ~0ull;
        if (e < n) {
            if (MODE == 0) {
                uint32_t u = __float_as_uint(vals[q * ns + e]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                kk = ((unsigned long long)u << 32) | (uint32_t)e;
            } else {
                kk = cand[q * (int64_t)cap + e];
            }
        }
        key[e] = kk;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = threadIdx.x; e < P / 2; e += 1024) {
                const int lo = (e / stride) * 2 * stride + (e % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long x = key[lo], y = key[hi];
                if ((x > y) == up) { key[lo] = y; key[hi] = x; }
            }
            __syncthreads();
        }
    if (MODE == 0) {
        if (threadIdx.x == 0) {
            const uint32_t u = (uint32_t)(key[k - 1 < n ? k - 1 : n - 1] >> 32);
            tau[q] = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
        }
    } else {
        for (int e = threadIdx.x; e < k; e += 1024) {
            out[q * k + e] = e < n ? (int64_t)(uint32_t)key[e] : -1;
            if (e < n) cand[q * (int64_t)cap + e] = key[e];
        }
        if (threadIdx.x == 0) {
            count[q] = n < k ? n : k;
            if (n >= k) {
                const uint32_t u = (uint32_t)(key[k - 1] >> 32);
                tau[q] = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
            }
        }
    }
}

// `ns` rows of Y at stride N / ns (row i * stride) -> a contiguous sample
__global__ void sample_rows_kernel(const float *__restrict__ Y, int d, int64_t ns, int64_t stride,
                                   float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns * d) return;
    const int64_t r = i / d;
    out[i] = Y[r * stride * d + (i - r * d)];
}

// X (nq, d), Y (N, d) float32 on the device, d <= 128, N < 2^31; out (nq, k) int64.
// Work buffers are the caller's: ynorm2 (N), vals (nq * ns), tau (nq), cand (nq * cap) u64,
// count (nq) int, overflow (1) int, sample (ns * (d + 1)) floats.
// ns <= TK_BR_SORT, cap <= TK_BR_SORT, k <= ns <= N, 2k <= cap.
int tk_launch_knn_brute(const float *X, int64_t nq, int d, const float *Y, int64_t N, int k,
                        float *ynorm2, float *vals, int64_t ns, float *tau,
                        unsigned long long *cand, int cap, int *count, int *overflow, int64_t *out,
                        float *sample, hipStream_t s)
{
    if (nq == 0) return 0;
    if (d > 128 || ns > TK_BR_SORT || cap > TK_BR_SORT || k > ns || ns > N || 2 * k > cap) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        const void *fns[] = {(const void *)brute_tiles_kernel<0>, (const void *)brute_tiles_kernel<1>,
                             (const void *)brute_select_kernel<0>, (const void *)brute_select_kernel<1>};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return -1;
        attr_set = true;
    }
    const int KT = (d + 1) / 2;
    const size_t tile_lds = (size_t)(4 + 2) * KT * 64 * 4;
    // enough workgroups for the chip: split the rows of Y when there are few query tiles
    const unsigned qt = (unsigned)((nq + 127) / 128);
    auto splits = [&](int64_t cols) {
        int64_t s = (1024 + qt - 1) / qt;
        const int64_t most = (cols + 1023) / 1024;          // at least 32 tiles per split
        s = s < 1 ? 1 : (s > most ? most : s);
        return (unsigned)(s < 1 ? 1 : s);
    };
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, Y, N, d, ynorm2);
    // tau from a strided sample of the whole matrix (its rows are rows of Y: tau >= the true
    // k-th smallest part, bit for bit the same arithmetic)
    float *snorm = sample + (size_t)ns * d;
    hipLaunchKernelGGL(sample_rows_kernel, dim3((unsigned)((ns * d + 255) / 256)), dim3(256), 0, s, Y, d, ns,
                       N / ns, sample);
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, sample, ns, d, snorm);
    hipLaunchKernelGGL(brute_tiles_kernel<0>, dim3(qt, splits(ns)), dim3(256), tile_lds, s, X, nq, d, sample, snorm,
                       ns, vals, ns, nullptr, nullptr, 0, nullptr, (int64_t)0);
    hipLaunchKernelGGL(brute_select_kernel<0>, dim3((unsigned)nq), dim3(1024), (size_t)TK_BR_SORT * 8, s,
                       vals, ns, nullptr, 0, nullptr, k, tau, nullptr, nullptr);
    (void)hipMemsetAsync(count, 0, (size_t)nq * 4, s);
    (void)hipMemsetAsync(overflow, 0, 4, s);
    const int64_t seg = 1 << 20;
    for (int64_t s0 = 0; s0 < N; s0 += seg) {
        const int64_t m = N - s0 < seg ? N - s0 : seg;
        hipLaunchKernelGGL(brute_tiles_kernel<1>, dim3(qt, splits(m)), dim3(256), tile_lds, s, X, nq, d,
                           Y + s0 * d, ynorm2 + s0, m, nullptr, 0, tau, cand, cap, count, s0);
        hipLaunchKernelGGL(brute_select_kernel<1>, dim3((unsigned)nq), dim3(1024), (size_t)TK_BR_SORT * 8, s,
                           nullptr, 0, cand, cap, count, k, tau, out, overflow);
    }
    return 0;
}
