// np_order.h — numpy's summation orders, shared by the kernels that restate numpy arithmetic.
#pragma once
#include <hip/hip_runtime.h>

// np.einsum("...k,...k->...") inner kernel for contiguous operands and a stride-0 output
// (einsum_sumprod.c.src, *_sum_of_products_contig_contig_outstride0_two) as built for the
// SSE3 baseline: L = 16 / sizeof(T) lanes, un-fused multiply-add, groups of four vectors
// folded in the order 3,2,1,0, zero-filled tail vectors, horizontal add (l0+l1)+(l2+l3).
template <typename T>
struct EinsumLanes;
template <>
struct EinsumLanes<float> { static constexpr int L = 4; };
template <>
struct EinsumLanes<double> { static constexpr int L = 2; };

// sum_k a[k]*a[k] for k < n in that order
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

// the same over a[k] = f(k) computed on the fly (every element is used once): no private array, so a
// run-time n leaves nothing to index dynamically (build_tables_kernel<double, true> kept 272 B per
// lane of scratch for its `diff[32]`)
template <typename T, typename F>
__device__ __forceinline__ T einsum_selfdot_fn(F f, int n)
{
    constexpr int L = EinsumLanes<T>::L;
    T acc[L];
#pragma unroll
    for (int l = 0; l < L; l++) acc[l] = 0;
    int i = 0;
    for (; n - i >= 4 * L; i += 4 * L) {
#pragma unroll
        for (int l = 0; l < L; l++) {
            const T x3 = f(i + 3 * L + l), x2 = f(i + 2 * L + l), x1 = f(i + L + l), x0 = f(i + l);
            T ab3 = x3 * x3 + acc[l];
            T ab2 = x2 * x2 + ab3;
            T ab1 = x1 * x1 + ab2;
            acc[l] = x0 * x0 + ab1;
        }
    }
    for (; i < n; i += L) {
#pragma unroll
        for (int l = 0; l < L; l++) {
            T x = (i + l < n) ? f(i + l) : (T)0;
            acc[l] = x * x + acc[l];
        }
    }
    if (L == 4) return (acc[0] + acc[1]) + (acc[2 % L] + acc[3 % L]);
    return acc[0] + acc[1 % L];
}
