// tables.hip — per-query distance tables on gfx950, bit-exact with
// FastPQ.distance_table (fast_pq.py:186-222) and udistance_table (:224-252).
//
// The table is a chain of numpy operations whose rounding the kernel restates
// operation by operation (compiled with -ffp-contract=off: no FMA may appear):
//   diff  = centers - q                     element type T (float32, or float64
//                                           when the query is float64 / rotated)
//   dists = einsum("ijk,ijk->ij")           numpy's SSE3-baseline kernel: L = 16/
//                                           sizeof(T) lanes, un-fused multiply-add,
//                                           4-vector groups folded 3,2,1,0, zero
//                                           tail, horizontal add (l0+l1)+(l2+l3)
//   shift = T(mean(dists)) * ln2            mean = numpy PAIRWISE sum in MEMORY
//                                           order (leaves <= 128 with 8
//                                           accumulators) divided by T(count)
//   dists -= shift
//   scale = 128 / (f64(max) * sqrt_n_blocks)           always float64
//   table = uint8(int(rint(f64(dists) * scale)))       half-even, wrap-around
// One 64-lane workgroup per query; the 16*M distances sit in LDS.  The pairwise
// tree is evaluated by lane 0 in numpy's exact order (16*M <= a few thousand adds;
// the whole batch is microseconds and the scan kernels dominate).
#include "kernels.h"

template <typename T>
struct EinsumLanes;
template <>
struct EinsumLanes<float> { static constexpr int L = 4; };
template <>
struct EinsumLanes<double> { static constexpr int L = 2; };

// sum_k a[k]*a[k] for k < n in numpy's einsum order (see header)
template <typename T>
__device__ __forceinline__ T einsum_selfdot(const T *a, int n)
{
    constexpr int L = EinsumLanes<T>::L;
    T acc[L];
#pragma unroll
    for (int l = 0; l < L; l++) acc[l] = 0;
    int i = 0;
    for (; n - i >= 4 * L; i += 4 * L) {
#pragma unroll
        for (int l = 0; l < L; l++) {
            T ab3 = a[i + 3 * L + l] * a[i + 3 * L + l] + acc[l];
            T ab2 = a[i + 2 * L + l] * a[i + 2 * L + l] + ab3;
            T ab1 = a[i + L + l] * a[i + L + l] + ab2;
            acc[l] = a[i + l] * a[i + l] + ab1;
        }
    }
    for (; i < n; i += L) {
#pragma unroll
        for (int l = 0; l < L; l++) {
            T x = (i + l < n) ? a[i + l] : (T)0;
            acc[l] = x * x + acc[l];
        }
    }
    if (L == 4) return (acc[0] + acc[1]) + (acc[2 % L] + acc[3 % L]);
    return acc[0] + acc[1 % L];
}

// numpy pairwise sum of a[0..n) (contiguous), exact order.
template <typename T>
__device__ T pairwise_leaf(const T *a, int n)
{
    if (n < 8) {
        T res = 0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    T r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    }
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

template <typename T>
__device__ T pairwise_sum(const T *a, int n)
{
    // explicit post-order walk of numpy's recursion: n > 128 splits at
    // n2 = n/2 - (n/2)%8 into [0,n2) and [n2,n)
    int st_off[40], st_n[40], st_vis[40];
    T vals[40];
    int sp = 0, vp = 0;
    st_off[0] = 0; st_n[0] = n; st_vis[0] = 0; sp = 1;
    while (sp > 0) {
        int off = st_off[sp - 1], m = st_n[sp - 1];
        if (m <= 128) {
            vals[vp++] = pairwise_leaf(a + off, m);
            sp--;
        } else if (!st_vis[sp - 1]) {
            st_vis[sp - 1] = 1;
            int n2 = m / 2;
            n2 -= n2 % 8;
            st_off[sp] = off + n2; st_n[sp] = m - n2; st_vis[sp] = 0; sp++;
            st_off[sp] = off; st_n[sp] = n2; st_vis[sp] = 0; sp++;
        } else {
            T r = vals[--vp];
            T l = vals[--vp];
            vals[vp++] = l + r;
            sp--;
        }
    }
    return vals[0];
}

template <typename T, bool SIGNED>
__global__ __launch_bounds__(64) void build_tables_kernel(
    const float *__restrict__ centers, int dq, int dpb, int f_order, const T *__restrict__ qs,
    double aux0, double aux1, uint8_t *__restrict__ tables, T *__restrict__ shift_out,
    double *__restrict__ scale_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *dists = (T *)smem;  // 16*M, numpy memory order
    __shared__ T sh_shift;
    const int lane = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int M = dq / dpb;
    const int cnt = 16 * M;
    const T *q = qs + qi * dq;
    constexpr int MAXDPB = 32;
    T diff[MAXDPB];

    for (int e = lane; e < cnt; e += 64) {
        const int i = e / M, m = e - i * M;
        T v;
        if (SIGNED) {
            if (dpb <= MAXDPB) {
                for (int k = 0; k < dpb; k++)
                    diff[k] = (T)centers[(int64_t)i * dq + m * dpb + k] - q[m * dpb + k];
                v = einsum_selfdot<T>(diff, dpb);
            } else {
                v = 0;  // rejected on the host
            }
        } else {
            // np.square(centers - q).reshape(16, nb, dpb).sum(-1): sequential adds
            v = 0;
            for (int k = 0; k < dpb; k++) {
                T df = (T)centers[(int64_t)i * dq + m * dpb + k] - q[m * dpb + k];
                T sq = df * df;
                v += sq;
            }
        }
        dists[f_order ? (m * 16 + i) : e] = v;
    }
    __syncthreads();
    if (lane == 0) {
        T sh;
        if (SIGNED) {
            T mean = pairwise_sum<T>(dists, cnt) / (T)cnt;   // _mean
            sh = mean * (T)0.6931471806;                       // fast_pq.py:214
        } else {
            sh = dists[0];
            for (int e = 1; e < cnt; e++) sh = dists[e] < sh ? dists[e] : sh;  // np.min
        }
        sh_shift = sh;
    }
    __syncthreads();
    const T shift = sh_shift;
    T mx = -INFINITY;
    for (int e = lane; e < cnt; e += 64) {
        T v = dists[e] - shift;  // fast_pq.py:215 / :247
        dists[e] = v;
        mx = v > mx ? v : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        T other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
    }
    double scale;
    if (SIGNED)
        scale = 128.0 / ((double)mx * aux0);           // :216
    else
        scale = 255.0 / (((double)mx * aux0) * aux1);  // :248
    __syncthreads();
    for (int e = lane; e < cnt; e += 64) {
        const int i = e / M, m = e - i * M;
        double v = rint((double)dists[f_order ? (m * 16 + i) : e] * scale);
        tables[qi * (int64_t)cnt + m * 16 + i] = (uint8_t)(int32_t)v;  // :217-221
    }
    if (lane == 0) {
        shift_out[qi] = shift;
        scale_out[qi] = scale;
    }
}

void tk_launch_build_tables(const float *centers, int dq, int dpb, int f_order, const void *q,
                            int q_is_f64, int64_t nq, double aux0, double aux1, int signd,
                            uint8_t *tables, void *shift, double *scale, hipStream_t s)
{
    if (nq == 0) return;
    const int M = dq / dpb;
    size_t lds = (size_t)16 * M * (q_is_f64 ? 8 : 4);
    dim3 grid((unsigned)nq), block(64);
    if (q_is_f64) {
        if (signd)
            hipLaunchKernelGGL((build_tables_kernel<double, true>), grid, block, lds, s, centers, dq,
                               dpb, f_order, (const double *)q, aux0, aux1, tables,
                               (double *)shift, scale);
        else
            hipLaunchKernelGGL((build_tables_kernel<double, false>), grid, block, lds, s, centers,
                               dq, dpb, f_order, (const double *)q, aux0, aux1, tables,
                               (double *)shift, scale);
    } else {
        if (signd)
            hipLaunchKernelGGL((build_tables_kernel<float, true>), grid, block, lds, s, centers, dq,
                               dpb, f_order, (const float *)q, aux0, aux1, tables, (float *)shift,
                               scale);
        else
            hipLaunchKernelGGL((build_tables_kernel<float, false>), grid, block, lds, s, centers,
                               dq, dpb, f_order, (const float *)q, aux0, aux1, tables,
                               (float *)shift, scale);
    }
}
