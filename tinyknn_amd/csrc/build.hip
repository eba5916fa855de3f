// build.hip — the step BEFORE the hot path: PQ encoding
// (FastPQ.transform, fast_pq.py:147-184) and the assignment of every point to its nearest
// coarse centres (IVF.build, ivf.py:85; knn_brute, utils.py:66-86).  On the CPU both are a
// Python loop over 100-row chunks — hours at 100M vectors.
//
// Both are chains of numpy operations whose rounding the kernels restate (-ffp-contract=off;
// every FMA below is explicit):
//   part = (|x|^2 + |y|^2) - (2x) @ y.T          utils.py:84 ("2 * X @ Y.T" = (2X) @ Y.T)
//   |.|^2      np.einsum("ij,ij->i"): numpy's SSE3-baseline kernel (np_order.h)
//   (2x)@y.T   OpenBLAS GEMM: for the shapes the path issues (100-row chunks, K <= 384) every
//              output element is the FMA chain over k ascending from 0 [measured against
//              exact rational arithmetic and numpy, tests/test_build_path.py]
//   argpartition(part, k)[:, :k], k <= 2: numpy's dumb_select (kth < 3): a selection sort
//              by strict "<" that swaps the winners into place
// float32 rows against float32 centres run in float32; a float64 operand (rotated PQ,
// float64 centres) promotes the rest to float64 exactly as numpy does.
#include "kernels.h"
#include "np_order.h"

// ---------------------------------------------------------------------------
// labels[row][b] = argmin_c part(row, b, c), first occurrence of the minimum
template <typename T>
struct Fma;
template <>
struct Fma<float> { static __device__ __forceinline__ float f(float a, float b, float c) { return __builtin_fmaf(a, b, c); } };
template <>
struct Fma<double> { static __device__ __forceinline__ double f(double a, double b, double c) { return __builtin_fma(a, b, c); } };

#define TK_ENC_MAX_DPB 32
#define TK_ENC_STRIP 16    // elements per staged column strip; dims_per_block must divide it

// DPB > 0: dims_per_block known at compile time (the row slice stays in registers);
// DPB == 0: any dims_per_block <= 32.
template <typename T, int DPB>
__global__ __launch_bounds__(256) void encode_pq_kernel(const float *__restrict__ centers, int dq,
                                                        int dpb_rt, const T *__restrict__ data,
                                                        int64_t n, uint8_t *__restrict__ labels)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dpb = DPB > 0 ? DPB : dpb_rt;
    constexpr int CAP = DPB > 0 ? DPB : TK_ENC_MAX_DPB;
    const int M = dq / dpb;
    // per block b: 16 x (dpb centroid coordinates, |centroid|^2), so that one centroid is one
    // contiguous LDS read
    float *cb = (float *)smem;             // (M, 16, dpb + 1)
    for (int e = threadIdx.x; e < 16 * M; e += 256) {
        const int b = e >> 4, c = e & 15;
        float y[CAP];
        for (int k = 0; k < dpb; k++) {
            y[k] = centers[c * dq + b * dpb + k];
            cb[e * (dpb + 1) + k] = y[k];
        }
        cb[e * (dpb + 1) + dpb] = einsum_selfdot<float>(y, dpb);
    }
    __syncthreads();
    // Rows are staged through LDS in column strips of TK_ENC_STRIP elements: the 64 rows of
    // a wave are adjacent in memory, so the strip is read with coalesced loads (a lane
    // walking its own row would touch 64 different lines per load and thrash L1), and the
    // labels leave as one contiguous block per wave.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T *tile = (T *)(cb + 16 * M * (dpb + 1)) + (size_t)wave * 64 * (TK_ENC_STRIP + 1);
    uint8_t *lab = (uint8_t *)((T *)(cb + 16 * M * (dpb + 1)) + (size_t)4 * 64 * (TK_ENC_STRIP + 1)) +
                   (size_t)wave * 64 * M;
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 64;
    for (int s0 = 0; s0 < dq; s0 += TK_ENC_STRIP) {
        const int w = dq - s0 < TK_ENC_STRIP ? dq - s0 : TK_ENC_STRIP;
        __syncthreads();
        for (int e = lane; e < 64 * w; e += 64) {
            const int r = e / w, k = e - r * w;
            tile[r * (TK_ENC_STRIP + 1) + k] = row0 + r < n ? data[(row0 + r) * dq + s0 + k] : (T)0;
        }
        __syncthreads();
        for (int b = s0 / dpb; b < (s0 + w) / dpb; b++) {
            T x[CAP], x2[CAP];
#pragma unroll
            for (int k = 0; k < dpb; k++) {
                x[k] = tile[lane * (TK_ENC_STRIP + 1) + b * dpb - s0 + k];
                x2[k] = (T)2 * x[k];
            }
            const T xn = einsum_selfdot<T>(x, dpb);
            int best = 0;
            T bestv = 0;
            const float *yb = cb + b * 16 * (dpb + 1);
#pragma unroll 4
            for (int c = 0; c < 16; c++) {
                const float *y = yb + c * (dpb + 1);
                T p = 0;
#pragma unroll
                for (int k = 0; k < dpb; k++) p = Fma<T>::f(x2[k], (T)y[k], p);
                const T part = (xn + (T)y[dpb]) - p;
                if (c == 0 || part < bestv) {
                    bestv = part;
                    best = c;
                }
            }
            lab[lane * M + b] = (uint8_t)best;
        }
    }
    __syncthreads();
    const int64_t nrow = n - row0 < 64 ? n - row0 : 64;      // rows of this wave (may be <= 0)
    for (int64_t e = lane; e < nrow * M; e += 64) labels[row0 * M + e] = lab[e];
}

int tk_launch_encode_pq(const float *centers, int dq, int dpb, const void *data, int is_f64,
                        int64_t n, uint8_t *labels, hipStream_t s)
{
    if (n == 0) return 0;
    const int M = dq / dpb;
    // codebook + four strip tiles + four label tiles
    const size_t lds = (size_t)16 * M * (dpb + 1) * 4 +
                       (size_t)4 * 64 * (TK_ENC_STRIP + 1) * (is_f64 ? 8 : 4) + (size_t)4 * 64 * M;
    if (lds > 160 * 1024 || dpb > TK_ENC_MAX_DPB || TK_ENC_STRIP % dpb) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        const void *fns[] = {(const void *)encode_pq_kernel<float, 1>, (const void *)encode_pq_kernel<float, 2>,
                             (const void *)encode_pq_kernel<float, 4>, (const void *)encode_pq_kernel<float, 8>,
                             (const void *)encode_pq_kernel<float, 0>, (const void *)encode_pq_kernel<double, 1>,
                             (const void *)encode_pq_kernel<double, 2>, (const void *)encode_pq_kernel<double, 4>,
                             (const void *)encode_pq_kernel<double, 8>, (const void *)encode_pq_kernel<double, 0>};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return -1;
        attr_set = true;
    }
    dim3 grid((unsigned)((n + 255) / 256));
#define TK_ENC(T_, D_)                                                                          \
    hipLaunchKernelGGL((encode_pq_kernel<T_, D_>), grid, dim3(256), lds, s, centers, dq, dpb,   \
                       (const T_ *)data, n, labels)
#define TK_ENC_T(T_)                                              \
    do {                                                          \
        if (dpb == 1) TK_ENC(T_, 1);                              \
        else if (dpb == 2) TK_ENC(T_, 2);                         \
        else if (dpb == 4) TK_ENC(T_, 4);                         \
        else if (dpb == 8) TK_ENC(T_, 8);                         \
        else TK_ENC(T_, 0);                                       \
    } while (0)
    if (is_f64) TK_ENC_T(double); else TK_ENC_T(float);
#undef TK_ENC_T
#undef TK_ENC
    return 0;
}

// ---------------------------------------------------------------------------
// X / np.linalg.norm(X, axis=1, keepdims=True) for float32 rows of d <= 128 elements:
// sqrt(add.reduce(x*x)) with numpy's pairwise summation = one leaf of 8 accumulators.
__global__ void normalise_rows_kernel(const float *__restrict__ X, int64_t n, int d,
                                      float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *x = X + i * d;
    float res;
    if (d < 8) {
        res = 0.0f;
        for (int t = 0; t < d; t++) res += x[t] * x[t];
    } else {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = x[j] * x[j];
        int t = 8;
        for (; t < d - (d % 8); t += 8)
            for (int j = 0; j < 8; j++) r[j] += x[t + j] * x[t + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; t < d; t++) res += x[t] * x[t];
    }
    const float nr = __builtin_sqrtf(res);     // HIP's default: correctly rounded sqrt and divide
    for (int t = 0; t < d; t++) out[i * d + t] = x[t] / nr;
}

// knn_brute, ROWS rows per workgroup: thread t scores centres t, t+256, ...; the
// rows are wave-uniform (scalar loads feed the FMAs), Yt is (d, L) so that a wave reads
// consecutive centres.  Every thread keeps its three best (value, index) per row, the
// workgroup merges them, and dumb_select's answer follows from the three best overall.


template <typename T>
struct Cand {
    T v;
    int j;
};

template <typename T>
__device__ __forceinline__ bool cand_lt(T v, int j, const Cand<T> &b)
{
    return b.j < 0 || v < b.v || (v == b.v && j < b.j);
}

template <typename T>
__device__ __forceinline__ void cand_push(Cand<T> (&top)[3], T v, int j)
{
    if (cand_lt(v, j, top[0])) {
        top[2] = top[1]; top[1] = top[0]; top[0] = {v, j};
    } else if (cand_lt(v, j, top[1])) {
        top[2] = top[1]; top[1] = {v, j};
    } else if (cand_lt(v, j, top[2])) {
        top[2] = {v, j};
    }
}

// ROWS rows per workgroup, C centres per thread and pass (each staged row value feeds C FMAs,
// each centre value ROWS FMAs), NTOP candidates kept per row: 1 when k == 1, 3 when k == 2
// (dumb_select's second pass needs the three best overall and part[0]).
template <typename T, int ROWS, int C, int NTOP>
__global__ __launch_bounds__(256) void assign_kernel(const float *__restrict__ X, int64_t n, int d,
                                                     const T *__restrict__ Yt,
                                                     const T *__restrict__ ynorm2, int L, int k,
                                                     int64_t *__restrict__ nearest)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *x2s = (T *)smem;                            // (d, ROWS): 2*x as T, row-interleaved
    __shared__ Cand<T> s_top[ROWS][4][NTOP];       // the best of each wave
    __shared__ T s_v0[ROWS];
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
    const int nr = n - r0 < ROWS ? (int)(n - r0) : ROWS;
    for (int e = threadIdx.x; e < d * ROWS; e += 256) {
        const int t = e / ROWS, r = e - t * ROWS;
        x2s[e] = (T)(2.0f * X[(r0 + (r < nr ? r : 0)) * d + t]);
    }
    // |x|^2 in float32 (np.einsum on the float32 rows), promoted when added to float64 |y|^2
    T xn[ROWS];
    for (int r = 0; r < ROWS; r++) {
        const float *x = X + (r0 + (r < nr ? r : 0)) * d;
        xn[r] = (T)einsum_selfdot<float>(x, d);
    }
    __syncthreads();
    Cand<T> top[ROWS][NTOP];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int t = 0; t < NTOP; t++) top[r][t] = {(T)0, -1};
    auto push = [&](Cand<T> (&tp)[NTOP], T v, int j) {
        if (NTOP == 1) {
            if (cand_lt(v, j, tp[0])) tp[0] = {v, j};
        } else {
            if (cand_lt(v, j, tp[0])) {
                tp[NTOP - 1] = tp[NTOP > 2 ? 1 : 0]; tp[NTOP > 1 ? 1 : 0] = tp[0]; tp[0] = {v, j};
            } else if (cand_lt(v, j, tp[NTOP > 1 ? 1 : 0])) {
                tp[NTOP - 1] = tp[NTOP > 1 ? 1 : 0]; tp[NTOP > 1 ? 1 : 0] = {v, j};
            } else if (cand_lt(v, j, tp[NTOP - 1])) {
                tp[NTOP - 1] = {v, j};
            }
        }
    };
    for (int j0 = threadIdx.x; j0 < L; j0 += 256 * C) {
        T p[C][ROWS];
        int jj[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            jj[c] = j0 + c * 256 < L ? j0 + c * 256 : j0;   // out of range: recompute j0, ignored
#pragma unroll
            for (int r = 0; r < ROWS; r++) p[c][r] = 0;
        }
        for (int t = 0; t < d; t++) {
            T y[C];
#pragma unroll
            for (int c = 0; c < C; c++) y[c] = Yt[(int64_t)t * L + jj[c]];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const T x2 = x2s[t * ROWS + r];             // same address in every lane
#pragma unroll
                for (int c = 0; c < C; c++) p[c][r] = Fma<T>::f(x2, y[c], p[c][r]);
            }
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j = j0 + c * 256;
            if (j < L) {
                const T yn = ynorm2[j];
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const T part = (xn[r] + yn) - p[c][r];
                    push(top[r], part, j);
                    if (j == 0) s_v0[r] = part;
                }
            }
        }
    }
    // the best of the wave: butterfly over the lanes, each step merging the partner's list
    // into mine (both lanes end up with the same list)
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        for (int o = 32; o > 0; o >>= 1) {
            Cand<T> other[NTOP];
#pragma unroll
            for (int t = 0; t < NTOP; t++) {
                other[t].v = __shfl_xor(top[r][t].v, o, 64);
                other[t].j = __shfl_xor(top[r][t].j, o, 64);
            }
#pragma unroll
            for (int t = 0; t < NTOP; t++)
                if (other[t].j >= 0) push(top[r], other[t].v, other[t].j);
        }
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int t = 0; t < NTOP; t++) s_top[r][threadIdx.x >> 6][t] = top[r][t];
    }
    __syncthreads();
    if ((int)threadIdx.x < nr) {
        const int r = threadIdx.x;
        Cand<T> best[NTOP];
#pragma unroll
        for (int t = 0; t < NTOP; t++) best[t] = {(T)0, -1};
        for (int t = 0; t < 4; t++)
            for (int u = 0; u < NTOP; u++) {
                const Cand<T> c = s_top[r][t][u];
                if (c.j >= 0) push(best, c.v, c.j);
            }
        const int m0 = best[0].j;          // first occurrence of the minimum
        nearest[(r0 + r) * k] = m0;
        if (NTOP > 1 && k == 2) {
            // dumb_select, second pass: positions 0 and m0 were swapped, so the scan order
            // is 1 .. m0-1, (element 0 at position m0), m0+1 .. L-1, strict "<"
            int second;
            if (m0 == 0) {
                second = best[1].j;
            } else {
                int s = -1;
                T sv = 0;
                for (int u = 1; u < NTOP; u++)
                    if (best[u].j > 0) { s = best[u].j; sv = best[u].v; break; }
                const T v0 = s_v0[r];
                if (s < 0 || v0 < sv || (v0 == sv && m0 < s)) second = 0;
                else second = s;
            }
            nearest[(r0 + r) * k + 1] = second;
        }
    }
}

void tk_launch_normalise_rows(const float *X, int64_t n, int d, float *out, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(normalise_rows_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, X,
                       n, d, out);
}

void tk_launch_assign(const float *X, int64_t n, int d, const void *Yt, const void *ynorm2,
                      int y_is_f64, int L, int k, int64_t *nearest, hipStream_t s)
{
    if (n == 0) return;
    // k == 1: 16 rows per workgroup (the centre matrix is re-read from L2 once per workgroup:
    // twice the rows, half the traffic), one candidate per row; k == 2: 8 rows, three candidates
#define TK_ASSIGN(T_, ROWS_, C_, NTOP_)                                                              \
    hipLaunchKernelGGL((assign_kernel<T_, ROWS_, C_, NTOP_>), dim3((unsigned)((n + ROWS_ - 1) / ROWS_)), \
                       dim3(256), (size_t)d * ROWS_ * sizeof(T_), s, X, n, d, (const T_ *)Yt,         \
                       (const T_ *)ynorm2, L, k, nearest)
    if (k == 1) {
        if (y_is_f64) TK_ASSIGN(double, 16, 2, 1); else TK_ASSIGN(float, 16, 2, 1);
    } else {
        if (y_is_f64) TK_ASSIGN(double, 8, 4, 3); else TK_ASSIGN(float, 8, 4, 3);
    }
#undef TK_ASSIGN
}

// ---------------------------------------------------------------------------
// Device front end ("fast mode"): what IVF.query does on the host before
// the table build (ivf.py:125-128, fast_pq.py:200-204).  NOT bit-identical to the host path:
// numpy normalises a query with a BLAS dot and rotates it with a BLAS GEMV, whose summation
// orders are not restated; here the norm is numpy's pairwise float32 sum (the order of
// np.linalg.norm(axis=1)) and the rotation a float64 FMA chain over k ascending.
// One workgroup of 64 lanes per query; Rt is R transposed (d_pad, dq) so that lane j reads
// consecutive addresses.
__global__ __launch_bounds__(64) void rotate_rows_kernel(const float *__restrict__ X, int64_t n,
                                                         int d, int d_pad,
                                                         const double *__restrict__ Rt, int dq,
                                                         double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *x = (float *)smem;
    const int64_t i = blockIdx.x;
    for (int t = threadIdx.x; t < d_pad; t += 64) x[t] = t < d ? X[i * d + t] : 0.0f;
    __syncthreads();
    for (int j = threadIdx.x; j < dq; j += 64) {
        double acc = 0.0;
        for (int t = 0; t < d_pad; t++) acc = __builtin_fma((double)x[t], Rt[(int64_t)t * dq + j], acc);
        out[i * dq + j] = acc;
    }
}

// zero padding only (no rotation): out (n, dq) float32
__global__ void pad_rows_kernel(const float *__restrict__ X, int64_t n, int d, int dq,
                                float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * dq) return;
    const int64_t i = e / dq;
    const int t = (int)(e - i * dq);
    out[e] = t < d ? X[i * d + t] : 0.0f;
}

void tk_launch_prepare_queries(const float *X, int64_t n, int d, const double *Rt, int dq, int d_pad,
                               void *out, hipStream_t s)
{
    if (n == 0) return;
    if (Rt)
        hipLaunchKernelGGL(rotate_rows_kernel, dim3((unsigned)n), dim3(64), (size_t)d_pad * 4, s, X, n,
                           d, d_pad, Rt, dq, (double *)out);
    else
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n * dq + 255) / 256)), dim3(256), 0, s, X,
                           n, d, dq, (float *)out);
}
