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

// NTOP > 3 (k = 3 .. 9: the k nearest in ascending (value, index) order — numpy's own order there
// is its SIMD quickselect's, ascending on the fixture host, unpinned by the reference): insertion
// into a sorted list held in registers, branch-free below the first test
template <typename T, int NTOP>
__device__ __forceinline__ void cand_insert(Cand<T> (&tp)[NTOP], T v, int j)
{
    if (!cand_lt(v, j, tp[NTOP - 1])) return;
#pragma unroll
    for (int t = NTOP - 1; t >= 0; t--) {
        const bool here = cand_lt(v, j, tp[t]);
        const bool above = t > 0 && cand_lt(v, j, tp[t > 0 ? t - 1 : 0]);
        if (above) tp[t] = tp[t > 0 ? t - 1 : 0];
        else if (here) tp[t] = {v, j};
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
        if (NTOP > 3) {
            cand_insert<T, NTOP>(tp, v, j);
        } else if (NTOP == 1) {
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
        if (NTOP > 3) {                    // k >= 3: ascending (value, index)
#pragma unroll
            for (int t = 1; t < NTOP; t++)
                if (t < k) nearest[(r0 + r) * k + t] = best[t].j;
        } else if (NTOP > 1 && k == 2) {
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

// ---------------------------------------------------------------------------
// The same assignment on the matrix cores, float32 centres (what sklearn's KMeans leaves for
// float32 data).  v_mfma_f32_32x32x2_f32 is bit-for-bit the f32 FMA chain over k ascending
// (MI355X guide, "FP32-input MFMA") — exactly what OpenBLAS returns for (2X) @ Y.T — so this
// is the one GEMM-shaped operation of the package on MFMA with NO change of results.
// One wave = 32 rows against all centres, 32 at a time:
//   A[i = lane & 31][k = lane >> 5] = 2 * x[i][2t + (lane >> 5)]   staged once per wave in LDS
//   B[k = lane >> 5][j = lane & 31] = Yt[2t + (lane >> 5)][j0 + (lane & 31)]   coalesced
//   D: lane holds column j = lane & 31, rows i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// Each lane keeps the running best (NTOP per row) of ITS columns for its 16 rows; one
// butterfly over the 32 lanes of a half at the end.  d <= 2 * TK_MF_KT (LDS: 4 waves x d/2 x 256 B).
#define TK_MF_KT 64
typedef float tk_f32x16 __attribute__((ext_vector_type(16)));

template <int NTOP>
__global__ __launch_bounds__(256) void assign_mfma_kernel(const float *__restrict__ X, int64_t n,
                                                          int d, const float *__restrict__ Yt,
                                                          const float *__restrict__ ynorm2, int L,
                                                          int k, int64_t *__restrict__ nearest)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    // (waves past the last row still take part in the workgroup's barriers)
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    const int64_t my_row = r0 + col < n ? r0 + col : n - 1;     // A-operand row of this lane
    const int KT = (d + 1) >> 1;
    // LDS: the A operands of each wave, [t][lane] (one conflict-free read per MFMA), and the B
    // operands of the workgroup's current column tile, [t][lane], double-buffered: the four
    // waves work on different rows of the SAME 32 centres, so a tile is fetched once per
    // workgroup (a 256-byte operand per MFMA straight from L2 would need 9.7 TB/s at the MFMA
    // peak)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *a = (float *)smem + (size_t)(threadIdx.x >> 6) * KT * 64;
    float *bt = (float *)smem + (size_t)4 * KT * 64;         // [2][KT][64]
    for (int t = 0; t < KT; t++) {
        const int kk = 2 * t + half;
        a[t * 64 + lane] = kk < d ? 2.0f * X[my_row * d + kk] : 0.0f;
    }
    // |x|^2 of the 16 rows this lane sees in D (np.einsum on the float32 rows)
    float xn[16];
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const int64_t row = r0 + (v & 3) + 8 * (v >> 2) + 4 * half;
        xn[v] = einsum_selfdot<float>(X + (row < n ? row : n - 1) * d, d);
    }
    Cand<float> top[16][NTOP];
    float v0[16];
#pragma unroll
    for (int v = 0; v < 16; v++) {
        v0[v] = 0.0f;
#pragma unroll
        for (int t = 0; t < NTOP; t++) top[v][t] = {0.0f, -1};
    }
    auto push = [&](Cand<float> (&tp)[NTOP], float val, int j) {
        if (NTOP == 1) {
            if (cand_lt(val, j, tp[0])) tp[0] = {val, j};
        } else {
            if (cand_lt(val, j, tp[0])) {
                tp[NTOP - 1] = tp[NTOP > 2 ? 1 : 0]; tp[NTOP > 1 ? 1 : 0] = tp[0]; tp[0] = {val, j};
            } else if (cand_lt(val, j, tp[NTOP > 1 ? 1 : 0])) {
                tp[NTOP - 1] = tp[NTOP > 1 ? 1 : 0]; tp[NTOP > 1 ? 1 : 0] = {val, j};
            } else if (cand_lt(val, j, tp[NTOP - 1])) {
                tp[NTOP - 1] = {val, j};
            }
        }
    };
    // element e of a B tile: t = e / 64, lane' = e % 64 -> Yt[2t + (lane' >> 5)][j0 + (lane' & 31)]
    const int per_thread = (KT * 64 + 255) / 256;
    float stage[TK_MF_KT / 4];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < TK_MF_KT / 4; u++) {
            const int e = u * 256 + (int)threadIdx.x;
            float val = 0.0f;
            if (u < per_thread && e < KT * 64) {
                const int t = e >> 6, l2 = e & 63;
                const int kk = 2 * t + (l2 >> 5), j = j0 + (l2 & 31);
                if (kk < d) val = Yt[(int64_t)kk * L + (j < L ? j : L - 1)];
            }
            stage[u] = val;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < TK_MF_KT / 4; u++) {
            const int e = u * 256 + (int)threadIdx.x;
            if (u < per_thread && e < KT * 64) bt[(size_t)buf * KT * 64 + e] = stage[u];
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int j0 = 0; j0 < L; j0 += 32, buf ^= 1) {
        const bool more = j0 + 32 < L;
        if (more) fetch(j0 + 32);                        // global loads in flight over the MFMAs
        const int j = j0 + col;
        tk_f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; v++) acc[v] = 0.0f;
        const float *bb = bt + (size_t)buf * KT * 64;
#pragma unroll 4
        for (int t = 0; t < KT; t++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t * 64 + lane], bb[t * 64 + lane], acc, 0, 0, 0);
        if (j < L) {
            const float yn = ynorm2[j];
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const float part = (xn[v] + yn) - acc[v];
                push(top[v], part, j);
                if (j == 0) v0[v] = part;
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    // merge over the 32 lanes (columns) of the half; part[0] lives in lane `half * 32`
#pragma unroll
    for (int v = 0; v < 16; v++) {
        for (int o = 16; o > 0; o >>= 1) {
            Cand<float> other[NTOP];
#pragma unroll
            for (int t = 0; t < NTOP; t++) {
                other[t].v = __shfl_xor(top[v][t].v, o, 64);
                other[t].j = __shfl_xor(top[v][t].j, o, 64);
            }
#pragma unroll
            for (int t = 0; t < NTOP; t++)
                if (other[t].j >= 0) push(top[v], other[t].v, other[t].j);
        }
    }
    if (col == 0) {
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int64_t row = r0 + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (row >= n) continue;
            const int m0 = top[v][0].j;
            nearest[row * k] = m0;
            if (NTOP > 1 && k == 2) {
                int second;
                if (m0 == 0) {
                    second = top[v][1].j;
                } else {
                    int s = -1;
                    float sv = 0;
                    for (int u = 1; u < NTOP; u++)
                        if (top[v][u].j > 0) { s = top[v][u].j; sv = top[v][u].v; break; }
                    if (s < 0 || v0[v] < sv || (v0[v] == sv && m0 < s)) second = 0;
                    else second = s;
                }
                nearest[row * k + 1] = second;
            }
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
    if (!y_is_f64 && k == 1 && d <= 2 * TK_MF_KT) {
        // float32 centres, nearest centre only: the matrix cores, same bits (see
        // assign_mfma_kernel; the three-candidate form for k == 2 spills and stays on the VALU)
        dim3 grid((unsigned)((n + 127) / 128));
        const size_t mf_lds = (size_t)(4 + 2) * ((d + 1) / 2) * 64 * 4;     // <= 96 KiB
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)assign_mfma_kernel<1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(assign_mfma_kernel<1>, grid, dim3(256), mf_lds, s, X, n, d, (const float *)Yt,
                           (const float *)ynorm2, L, k, nearest);
    } else if (k == 1) {
        if (y_is_f64) TK_ASSIGN(double, 16, 2, 1); else TK_ASSIGN(float, 16, 2, 1);
    } else if (k == 2) {
        if (y_is_f64) TK_ASSIGN(double, 8, 4, 3); else TK_ASSIGN(float, 8, 4, 3);
    } else {
        // k = 3 .. 9: nine sorted candidates per row and thread (examples/bench.py:108-111 sweeps
        // build_probes 1 .. 9)
        if (y_is_f64) TK_ASSIGN(double, 4, 2, 9); else TK_ASSIGN(float, 4, 4, 9);
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
