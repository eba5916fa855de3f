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
// One wave per query (four per workgroup); the 16*M distances sit in LDS.  The
// pairwise mean is evaluated in numpy's exact order with the 8 accumulators of
// every <=128-element leaf spread over the lanes; only the leaf combine is serial.
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

// Leaves of numpy's pairwise recursion over n elements, in order: n > 128 splits at
// n2 = n/2 - (n/2)%8 into [0,n2) and [n2,n).  Returns the leaf count (<= 64 for
// n <= 8192).
__device__ int pairwise_leaves(int n, int *leaf_off, int *leaf_n)
{
    int st_off[32], st_n[32];
    int sp = 0, nl = 0;
    st_off[0] = 0; st_n[0] = n; sp = 1;
    while (sp > 0) {
        sp--;
        const int off = st_off[sp], m = st_n[sp];
        if (m <= 128) {
            leaf_off[nl] = off;
            leaf_n[nl] = m;
            nl++;
        } else {
            int n2 = m / 2;
            n2 -= n2 % 8;
            st_off[sp] = off + n2; st_n[sp] = m - n2; sp++;   // right, popped after
            st_off[sp] = off; st_n[sp] = n2; sp++;            // left, popped first
        }
    }
    return nl;
}

// Combine the leaf sums in numpy's tree order (post-order: left + right).
template <typename T>
__device__ T pairwise_combine(int n, const T *leaf_sum)
{
    int st_n[40], st_vis[40];
    T vals[40];
    int sp = 0, vp = 0, next_leaf = 0;
    st_n[0] = n; st_vis[0] = 0; sp = 1;
    while (sp > 0) {
        const int m = st_n[sp - 1];
        if (m <= 128) {
            vals[vp++] = leaf_sum[next_leaf++];
            sp--;
        } else if (!st_vis[sp - 1]) {
            st_vis[sp - 1] = 1;
            int n2 = m / 2;
            n2 -= n2 % 8;
            st_n[sp] = m - n2; st_vis[sp] = 0; sp++;
            st_n[sp] = n2; st_vis[sp] = 0; sp++;
        } else {
            const T r = vals[--vp];
            const T l = vals[--vp];
            vals[vp++] = l + r;
            sp--;
        }
    }
    return vals[0];
}

// One wave per query, blockDim.x / 64 queries per workgroup.  Per-wave LDS:
//   dists[16M] T | acc[64][8] T | leaf_sum[64] T | leaf_off[64], leaf_n[64] int | misc
template <typename T, bool SIGNED>
__global__ __launch_bounds__(256) void build_tables_kernel(
    const float *__restrict__ centers, int dq, int dpb, int f_order, const T *__restrict__ qs,
    double aux0, double aux1, uint8_t *__restrict__ tables, T *__restrict__ shift_out,
    double *__restrict__ scale_out, int64_t nq, int wave_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const int M = dq / dpb;
    const int cnt = 16 * M;
    unsigned char *base = smem + (size_t)wave * wave_lds;
    T *dists = (T *)base;                 // numpy memory order
    T *acc = dists + cnt;                 // [leaf][8]
    T *leaf_sum = acc + 64 * 8;
    int *leaf_off = (int *)(leaf_sum + 64);
    int *leaf_n = leaf_off + 64;
    T *sh_shift = (T *)(leaf_n + 64);
    int *sh_nleaf = (int *)(sh_shift + 1);
    const int64_t qraw = (int64_t)blockIdx.x * waves + wave;
    const bool valid = qraw < nq;
    const int64_t qi = valid ? qraw : nq - 1;   // surplus waves recompute the last query
    const T *q = qs + qi * dq;
    constexpr int MAXDPB = 32;
    T diff[MAXDPB];

    for (int e = lane; e < cnt; e += 64) {
        const int i = e / M, m = e - i * M;
        T v;
        if (SIGNED) {
            for (int k = 0; k < dpb; k++)
                diff[k] = (T)centers[(int64_t)i * dq + m * dpb + k] - q[m * dpb + k];
            v = einsum_selfdot<T>(diff, dpb);
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
    if (lane == 0) *sh_nleaf = SIGNED ? pairwise_leaves(cnt, leaf_off, leaf_n) : 0;
    __syncthreads();
    T shift;
    if (SIGNED) {
        // numpy's pairwise mean, the 8 accumulators of every <=128-element leaf in
        // parallel (lane = leaf*8 + accumulator), then the exact combine tree
        const int nleaf = *sh_nleaf;
        for (int t = lane; t < nleaf * 8; t += 64) {
            const int L = t >> 3, j = t & 7;
            const int off = leaf_off[L], n = leaf_n[L];
            T r = 0;
            if (n >= 8) {
                r = dists[off + j];
                for (int i = 8; i < n - (n % 8); i += 8) r += dists[off + i + j];
            }
            acc[L * 8 + j] = r;
        }
        __syncthreads();
        for (int L = lane; L < nleaf; L += 64) {
            const int off = leaf_off[L], n = leaf_n[L];
            T res;
            if (n < 8) {
                res = 0;
                for (int i = 0; i < n; i++) res += dists[off + i];
            } else {
                const T *r = acc + L * 8;
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (int i = n - (n % 8); i < n; i++) res += dists[off + i];
            }
            leaf_sum[L] = res;
        }
        __syncthreads();
        if (lane == 0) {
            T mean = pairwise_combine<T>(cnt, leaf_sum) / (T)cnt;   // _mean
            *sh_shift = mean * (T)0.6931471806;                     // fast_pq.py:214
        }
        __syncthreads();
        shift = *sh_shift;
    } else {
        T mn = INFINITY;
        for (int e = lane; e < cnt; e += 64) mn = dists[e] < mn ? dists[e] : mn;   // np.min
        for (int o = 32; o > 0; o >>= 1) {
            T other = __shfl_xor(mn, o, 64);
            mn = other < mn ? other : mn;
        }
        shift = mn;
    }
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
    if (!valid) return;
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
    const size_t esz = q_is_f64 ? 8 : 4;
    // per-wave LDS: dists + 64x8 accumulators + 64 leaf sums + 2x64 ints + shift, nleaf
    int wave_lds = (int)(((size_t)16 * M + 64 * 8 + 64) * esz + 128 * 4 + 32);
    wave_lds = (wave_lds + 15) & ~15;
    int waves = 64 * 1024 / wave_lds;
    waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
    size_t lds = (size_t)waves * wave_lds;
    dim3 grid((unsigned)((nq + waves - 1) / waves)), block(64 * waves);
    const int pdpb = dpb;
    if (q_is_f64) {
        if (signd)
            hipLaunchKernelGGL((build_tables_kernel<double, true>), grid, block, lds, s, centers, dq,
                               pdpb, f_order, (const double *)q, aux0, aux1, tables,
                               (double *)shift, scale, nq, wave_lds);
        else
            hipLaunchKernelGGL((build_tables_kernel<double, false>), grid, block, lds, s, centers,
                               dq, pdpb, f_order, (const double *)q, aux0, aux1, tables,
                               (double *)shift, scale, nq, wave_lds);
    } else {
        if (signd)
            hipLaunchKernelGGL((build_tables_kernel<float, true>), grid, block, lds, s, centers, dq,
                               pdpb, f_order, (const float *)q, aux0, aux1, tables, (float *)shift,
                               scale, nq, wave_lds);
        else
            hipLaunchKernelGGL((build_tables_kernel<float, false>), grid, block, lds, s, centers,
                               dq, pdpb, f_order, (const float *)q, aux0, aux1, tables,
                               (float *)shift, scale, nq, wave_lds);
    }
}
