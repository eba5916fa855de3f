// adc_scan.hip — the Quick-ADC 4-bit PQ code scan for gfx950 (CDNA4).
//
// Replaces compute_block_dists (_fast_pq.pyx:209-236) and compute_block_dists_avx
// (_fast_pq_256.pyx:126-156) of the reference and the loops that call them
// (estimate_pq_* and the distance half of query_pq_*).
//
// Mapping.  One lane owns one 16-row chunk (the reference's __m128i), a wave 64
// consecutive chunks.  Per block pair p the lane loads the chunk's 16-byte group
// (global_load_dwordx4; the tiled layout of kernels.h makes 8 neighbouring lanes
// share one 128-byte line) and looks every nibble up in the query's 16-entry
// table with v_perm_b32 — the CDNA byte shuffle, used exactly as the reference
// uses pshufb: the table row is wave-uniform (it arrives through the scalar
// cache), the selector is the per-lane code.  A 16-entry table is two 8-byte
// halves, so a lookup of 4 rows is perm(lo half), perm(hi half), perm(select by
// code bit 3).  No LDS and no MFMA: the path is a gather/reduce.
//
// Exact int8 saturation.  Accumulators are int16 pairs holding value<<8; a
// v_pk_add_i16 with clamp then saturates at 127<<8|0xff / -128<<8, which is
// _mm_adds_epi8 bit for bit in the high byte (SURVEY §7 hard part 2); unsigned
// mode uses v_pk_add_u16 clamp.  One VGPR carries two rows; the AVX order keeps
// two such accumulator sets (block pairs p even / p odd) and merges them with
// one final saturating add, the SSE order keeps one.
#include <string.h>

#include "kernels.h"

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

template <bool SIGNED>
__device__ __forceinline__ uint32_t sat_add2(uint32_t a, uint32_t b)
{
    if (SIGNED) {
        v2s r = __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, a),
                                              __builtin_bit_cast(v2s, b));
        return __builtin_bit_cast(uint32_t, r);
    } else {
        v2u r = __builtin_elementwise_add_sat(__builtin_bit_cast(v2u, a),
                                              __builtin_bit_cast(v2u, b));
        return __builtin_bit_cast(uint32_t, r);
    }
}

// 4 rows (the 4 bytes of `sel3`, each a 3-bit index) looked up in one 16-entry
// byte table t; `pick` holds, per byte i, i or 4+i according to code bit 3.
// Result is added, value<<8, into the two row-pair accumulators.
template <bool SIGNED>
__device__ __forceinline__ void lut4(uint32_t sel3, uint32_t pick, const uint4 t,
                                     uint32_t &acc01, uint32_t &acc23)
{
    uint32_t lo = __builtin_amdgcn_perm(t.y, t.x, sel3);  // entries 0..7
    uint32_t hi = __builtin_amdgcn_perm(t.w, t.z, sel3);  // entries 8..15
    uint32_t r = __builtin_amdgcn_perm(hi, lo, pick);     // byte i = T[code_i]
    uint32_t w01 = __builtin_amdgcn_perm(0u, r, 0x010c000cu);  // [r1,0,r0,0]
    uint32_t w23 = __builtin_amdgcn_perm(0u, r, 0x030c020cu);  // [r3,0,r2,0]
    acc01 = sat_add2<SIGNED>(acc01, w01);
    acc23 = sat_add2<SIGNED>(acc23, w23);
}

// (a & b) | c in one VALU op.  VOP3 on gfx9 takes no literal and one SGPR, so the
// mask rides in an SGPR and the byte-index constant in a VGPR; hipcc otherwise
// splits this into v_and + v_or.
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask_s, uint32_t c_v)
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask_s), "v"(c_v));
    return r;
}

// One block pair: 16 rows x 2 blocks.  Low nibble = block 2p first, then the high
// nibble = block 2p+1, into the same accumulator set (the order of both reference
// kernels inside a 128-bit lane).
template <bool SIGNED>
__device__ __forceinline__ void pair_step(const uint4 x, const uint4 tl, const uint4 th,
                                          uint32_t (&acc)[8])
{
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t v = xs[j];
        uint32_t s_lo = v & 0x07070707u;
        uint32_t k_lo = and_or(v >> 1, 0x04040404u, 0x03020100u);
        uint32_t vh = v >> 4;
        uint32_t s_hi = vh & 0x07070707u;
        uint32_t k_hi = and_or(v >> 5, 0x04040404u, 0x03020100u);
        lut4<SIGNED>(s_lo, k_lo, tl, acc[2 * j], acc[2 * j + 1]);
        lut4<SIGNED>(s_hi, k_hi, th, acc[2 * j], acc[2 * j + 1]);
    }
}

template <bool SIGNED>
__device__ __forceinline__ uint32_t pk_min2(uint32_t a, uint32_t b)
{
    if (SIGNED) {
        v2s r = __builtin_elementwise_min(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b));
        return __builtin_bit_cast(uint32_t, r);
    } else {
        v2u r = __builtin_elementwise_min(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b));
        return __builtin_bit_cast(uint32_t, r);
    }
}

// All M blocks of chunk c against table `tab` (M uint4, wave-uniform address).
// Returns the chunk's 16 int8/uint8 distances and, in `mn`, their minimum (the
// heap replay uses it to skip blocks without reading them).
template <int ORDER, bool SIGNED>
__device__ __forceinline__ uint4 scan_chunk(const uint4 *__restrict__ codes, int64_t c, int P,
                                            const uint4 *__restrict__ tab, uint32_t &mn)
{
    const uint4 *src = codes + ((c >> 3) * (int64_t)P) * 8 + (c & 7);
    uint32_t a0[8], a1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a0[i] = a1[i] = 0;

    if (ORDER == TK_ORDER_AVX) {
        // _fast_pq_256.pyx:135-149: 4 blocks per step, pairs p even -> lane 0 of the
        // ymm (a0), p odd -> lane 1 (a1); a trailing odd pair is not read there.
        const int steps = P >> 1;
#pragma unroll 2
        for (int j = 0; j < steps; j++) {
            uint4 x0 = src[(2 * j) * 8];
            uint4 x1 = src[(2 * j + 1) * 8];
            pair_step<SIGNED>(x0, tab[4 * j], tab[4 * j + 1], a0);
            pair_step<SIGNED>(x1, tab[4 * j + 2], tab[4 * j + 3], a1);
        }
#pragma unroll
        // :152-156.  A clamped accumulator carries 0xff in its low byte; clear it in
        // one operand so the two low bytes cannot carry into the value byte.
        for (int i = 0; i < 8; i++) a0[i] = sat_add2<SIGNED>(a0[i], a1[i] & 0xff00ff00u);
    } else {
#pragma unroll 2
        for (int p = 0; p < P; p++) {
            uint4 x = src[p * 8];
            pair_step<SIGNED>(x, tab[2 * p], tab[2 * p + 1], a0);
        }
    }
    {   // minimum over the 16 rows: the value<<8 halves order like the values
        uint32_t m = pk_min2<SIGNED>(pk_min2<SIGNED>(pk_min2<SIGNED>(a0[0], a0[1]),
                                                     pk_min2<SIGNED>(a0[2], a0[3])),
                                     pk_min2<SIGNED>(pk_min2<SIGNED>(a0[4], a0[5]),
                                                     pk_min2<SIGNED>(a0[6], a0[7])));
        m = pk_min2<SIGNED>(m, m >> 16);
        mn = (m >> 8) & 0xffu;
    }
    uint4 o;
    o.x = __builtin_amdgcn_perm(a0[1], a0[0], 0x07050301u);
    o.y = __builtin_amdgcn_perm(a0[3], a0[2], 0x07050301u);
    o.z = __builtin_amdgcn_perm(a0[5], a0[4], 0x07050301u);
    o.w = __builtin_amdgcn_perm(a0[7], a0[6], 0x07050301u);
    return o;
}

// ---------------------------------------------------------------------------
// reference layout -> tiled layout
__global__ void retile_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                              int64_t chunks, int64_t chunks_pad, int P)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (c_pad, p)
    int64_t total = chunks_pad * P;
    if (i >= total) return;
    // iterate in destination order so that stores are coalesced
    int64_t tile = i / (8 * (int64_t)P);
    int rem = (int)(i - tile * 8 * P);
    int p = rem >> 3, cl = rem & 7;
    int64_t c = tile * 8 + cl;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < chunks) v = src[c * P + p];
    dst[i] = v;
}

void tk_launch_retile(const uint4 *src_ref, uint4 *dst_tiled, int64_t chunks, int P,
                      hipStream_t s)
{
    int64_t chunks_pad = (chunks + 7) / 8 * 8;
    int64_t total = chunks_pad * P;
    if (total == 0) return;
    int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(retile_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src_ref, dst_tiled,
                       chunks, chunks_pad, P);
}

// ---------------------------------------------------------------------------
template <int ORDER, bool SIGNED>
__global__ __launch_bounds__(256) void scan_flat_kernel(const uint4 *__restrict__ codes,
                                                        int64_t chunks, int P,
                                                        const uint4 *__restrict__ tables, int M,
                                                        uint4 *__restrict__ out,
                                                        int64_t out_stride,
                                                        uint8_t *__restrict__ mins,
                                                        int64_t min_stride)
{
    const int q = blockIdx.y;
    const uint4 *tab = tables + (int64_t)q * M;
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool active = c < chunks;
    // inactive lanes of the last wave read chunk 0 (in bounds) and drop the result
    uint32_t mn;
    uint4 o = scan_chunk<ORDER, SIGNED>(codes, active ? c : 0, P, tab, mn);
    if (active) {
        out[(int64_t)q * out_stride + c] = o;
        if (mins) mins[(int64_t)q * min_stride + c] = (uint8_t)mn;
    }
}

void tk_launch_scan_flat(const uint4 *codes, int64_t chunks, int M, const uint4 *tables,
                         int64_t nq, uint4 *out, int64_t out_stride, uint8_t *mins,
                         int64_t min_stride, int signd, int order, hipStream_t s)
{
    if (chunks == 0 || nq == 0) return;
    dim3 grid((unsigned)((chunks + 255) / 256), (unsigned)nq);
    int P = M / 2;
#define TK_LAUNCH(O, S_)                                                                      \
    hipLaunchKernelGGL((scan_flat_kernel<O, S_>), grid, dim3(256), 0, s, codes, chunks, P,    \
                       tables, M, out, out_stride, mins, min_stride)
    if (order == TK_ORDER_AVX) {
        if (signd) TK_LAUNCH(TK_ORDER_AVX, true); else TK_LAUNCH(TK_ORDER_AVX, false);
    } else {
        if (signd) TK_LAUNCH(TK_ORDER_SSE, true); else TK_LAUNCH(TK_ORDER_SSE, false);
    }
#undef TK_LAUNCH
}

// ---------------------------------------------------------------------------
// Probed lists of one query, concatenated in probe order into a flat chunk
// range [0, prefix[S]); lane f of the grid owns flat chunk f.
template <int ORDER, bool SIGNED>
__device__ __forceinline__ void scan_probes_row(
    const uint4 *__restrict__ codes, int P, const uint4 *__restrict__ tables, int M,
    const int *__restrict__ slot_prefix, const int64_t *__restrict__ slot_chunk0, int S,
    uint4 *__restrict__ dist, int64_t cap, uint8_t *__restrict__ mins, int64_t min_stride, int q)
{
    const int *prefix = slot_prefix + (int64_t)q * (S + 1);
    const int total = prefix[S];
    const int f0 = blockIdx.x * 256 + (threadIdx.x & ~63);
    if (f0 >= total) return;  // whole wave beyond this query's work
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool active = f < total;
    const int ff = active ? f : total - 1;
    // largest s with prefix[s] <= ff  (prefix is non-decreasing, prefix[0] = 0)
    int lo = 0, hi = S;  // invariant: prefix[lo] <= ff < prefix[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= ff) lo = mid; else hi = mid;
    }
    const int64_t c = slot_chunk0[(int64_t)q * S + lo] + (ff - prefix[lo]);
    const uint4 *tab = tables + (int64_t)q * M;
    uint32_t mn;
    uint4 o = scan_chunk<ORDER, SIGNED>(codes, c, P, tab, mn);
    if (active) {
        dist[(int64_t)q * cap + f] = o;
        if (mins) mins[(int64_t)q * min_stride + f] = (uint8_t)mn;
    }
}

// only_list (or NULL): [count, q_0, q_1, ...] — score only these queries; the gridDim.y rows of
// blocks stride over them (the re-scan of the queries the lane replay flagged, plain_scan.hip)
template <int ORDER, bool SIGNED>
__global__ __launch_bounds__(256) void scan_probes_kernel(
    const uint4 *__restrict__ codes, int P, const uint4 *__restrict__ tables, int M,
    const int *__restrict__ slot_prefix, const int64_t *__restrict__ slot_chunk0, int S,
    uint4 *__restrict__ dist, int64_t cap, uint8_t *__restrict__ mins, int64_t min_stride,
    const int *__restrict__ only_list)
{
    if (only_list) {
        const int count = only_list[0];
        for (int i = blockIdx.y; i < count; i += gridDim.y)
            scan_probes_row<ORDER, SIGNED>(codes, P, tables, M, slot_prefix, slot_chunk0, S, dist, cap, mins,
                                           min_stride, only_list[1 + i]);
        return;
    }
    scan_probes_row<ORDER, SIGNED>(codes, P, tables, M, slot_prefix, slot_chunk0, S, dist, cap, mins,
                                   min_stride, blockIdx.y);
}

// ids of the flagged queries, in order: list[0] = count, list[1..] = ids (one workgroup)
__global__ __launch_bounds__(1024) void flagged_list_kernel(const unsigned char *__restrict__ flags, int64_t nq,
                                                            int *__restrict__ list, volatile int *host_count)
{
    __shared__ int s_base, s_wave[16];
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b = 0; b < nq; b += 1024) {
        const int64_t q = b + threadIdx.x;
        const bool f = q < nq && flags[q];
        const uint64_t m = __builtin_amdgcn_ballot_w64(f);
        if (lane == 0) s_wave[wave] = __builtin_popcountll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        if (f) list[1 + before + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)q;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        list[0] = s_base;
        if (host_count) *host_count = s_base;      // page-locked host word: read without a synchronisation
    }
}

void tk_launch_flagged_list(const unsigned char *flags, int64_t nq, int *list, hipStream_t s, int *host_count)
{
    if (nq == 0) return;
    hipLaunchKernelGGL(flagged_list_kernel, dim3(1), dim3(1024), 0, s, flags, nq, list, host_count);
}

void tk_launch_scan_probes(const uint4 *codes, int M, const uint4 *tables, int64_t nq,
                           const int *slot_prefix, const int64_t *slot_chunk0, int S,
                           int max_flat_chunks, uint4 *dist, int64_t cap, uint8_t *mins,
                           int64_t min_stride, int signd, int order, hipStream_t s,
                           const int *only)
{
    if (nq == 0 || max_flat_chunks == 0 || S == 0) return;
    dim3 grid((unsigned)((max_flat_chunks + 255) / 256), (unsigned)(only ? (nq < 32 ? nq : 32) : nq));
    int P = M / 2;
#define TK_LAUNCH(O, S_)                                                                      \
    hipLaunchKernelGGL((scan_probes_kernel<O, S_>), grid, dim3(256), 0, s, codes, P, tables,  \
                       M, slot_prefix, slot_chunk0, S, dist, cap, mins, min_stride, only)
    if (order == TK_ORDER_AVX) {
        if (signd) TK_LAUNCH(TK_ORDER_AVX, true); else TK_LAUNCH(TK_ORDER_AVX, false);
    } else {
        if (signd) TK_LAUNCH(TK_ORDER_SSE, true); else TK_LAUNCH(TK_ORDER_SSE, false);
    }
#undef TK_LAUNCH
}

// ===========================================================================
// List-major scan ("units"): the throughput form for large batches.
//
// In a batch most inverted lists are probed by many queries (10 000 queries x 10
// probes over 1087 lists: ~92 queries per list).  Here a lane still owns one
// 16-row chunk, but scores it for FOUR queries per pass: the code dword is loaded
// once and the six selector words derived from it (the VALU work that does not
// depend on the query) are shared, so the cost per query drops from 21 to
// 12 + 11/4 ops per dword, and the code bytes are fetched once per four queries.
//   * work unit = (list l, group of 4 (query, probe-slot) pairs of l, chunk c);
//     units are numbered list by list, so the 64 lanes of a wave hold 64
//     consecutive units: no lane idles at list or query boundaries except at the
//     very end of the grid;
//   * the tables are no longer wave-uniform (a wave can straddle groups), so each
//     lane reads its queries' 16-byte table rows with ordinary vector loads; lanes
//     of one group hit the same address and the rows stay in L1 (832 B per query);
//   * selectors produce the value<<8 pairs directly: with rows (B0,B2) / (B1,B3) of
//     a code dword paired in one accumulator, one shift aligns both "bit 3" flags
//     with byte 1 and byte 3 of a selector, so perm(lo half), perm(hi half) and two
//     selecting perms give the two widened pairs (one perm fewer per query than the
//     lookup-then-widen form of the query-major kernel).
// Exactly the arithmetic of scan_chunk (same saturating chain per row).

template <bool SIGNED>
__device__ __forceinline__ void lut4x(uint32_t sel3, uint32_t kA, uint32_t kB, const uint4 t,
                                      uint32_t &acc02, uint32_t &acc13)
{
    uint32_t lo = __builtin_amdgcn_perm(t.y, t.x, sel3);
    uint32_t hi = __builtin_amdgcn_perm(t.w, t.z, sel3);
    uint32_t wA = __builtin_amdgcn_perm(hi, lo, kA);   // [T[B2],0,T[B0],0]
    uint32_t wB = __builtin_amdgcn_perm(hi, lo, kB);   // [T[B3],0,T[B1],0]
    acc02 = sat_add2<SIGNED>(acc02, wA);
    acc13 = sat_add2<SIGNED>(acc13, wB);
}

struct Sel6 {
    uint32_t s_lo, kA_lo, kB_lo, s_hi, kA_hi, kB_hi;
};

__device__ __forceinline__ Sel6 make_sel(uint32_t v, uint32_t cA, uint32_t cB)
{
    Sel6 r;
    r.s_lo = v & 0x07070707u;
    r.kA_lo = and_or(v << 7, 0x04000400u, cA);
    r.kB_lo = and_or(v >> 1, 0x04000400u, cB);
    r.s_hi = (v >> 4) & 0x07070707u;
    r.kA_hi = and_or(v << 3, 0x04000400u, cA);
    r.kB_hi = and_or(v >> 5, 0x04000400u, cB);
    return r;
}

template <bool SIGNED>
__device__ __forceinline__ void finish_chunk(uint32_t (&a0)[8], uint4 &o, uint32_t &mn)
{
    uint32_t m = pk_min2<SIGNED>(pk_min2<SIGNED>(pk_min2<SIGNED>(a0[0], a0[1]),
                                                 pk_min2<SIGNED>(a0[2], a0[3])),
                                 pk_min2<SIGNED>(pk_min2<SIGNED>(a0[4], a0[5]),
                                                 pk_min2<SIGNED>(a0[6], a0[7])));
    m = pk_min2<SIGNED>(m, m >> 16);
    mn = (m >> 8) & 0xffu;
    // a0[2j] = rows (4j, 4j+2), a0[2j+1] = rows (4j+1, 4j+3)
    o.x = __builtin_amdgcn_perm(a0[1], a0[0], 0x07030501u);
    o.y = __builtin_amdgcn_perm(a0[3], a0[2], 0x07030501u);
    o.z = __builtin_amdgcn_perm(a0[5], a0[4], 0x07030501u);
    o.w = __builtin_amdgcn_perm(a0[7], a0[6], 0x07030501u);
}

#include "tickets.h"

// One scan job of the list-major kernel: everything tk_launch_scan_units takes.
// (kernels.h: struct TkScanJob)

// Block `blk` (64 consecutive units) of a job.
template <int ORDER, bool SIGNED>
__device__ __forceinline__ void scan_units_block(
    const uint4 *__restrict__ codes, int P, const uint4 *__restrict__ tables, int M,
    const int64_t *__restrict__ list_chunk_off, int n_lists, const int *__restrict__ unit_prefix,
    const int *__restrict__ pair_off, const int *__restrict__ pair_q,
    const int *__restrict__ pair_f0, uint4 *__restrict__ dist, int64_t cap,
    uint8_t *__restrict__ mins, int64_t min_stride, int U, int blk, int max_chunks = 0)
{
    const uint32_t cA = 0x020c000cu, cB = 0x030c010cu;
    const int u0 = blk << 6;
    const int u = u0 + (threadIdx.x & 63);
    const bool active = u < U;
    const int uu = active ? u : U - 1;
    int lo = 0, hi = n_lists;   // unit_prefix[lo] <= uu < unit_prefix[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (unit_prefix[mid] <= uu) lo = mid; else hi = mid;
    }
    const int l = lo;
    const int64_t c0 = list_chunk_off[l];
    int C = (int)(list_chunk_off[l + 1] - c0);
    C = (max_chunks > 0 && max_chunks < C) ? max_chunks : C;     // head pairs: the list's first chunks only
    const int local = uu - unit_prefix[l];
    const int qg = local / C, c = local - qg * C;
    const int rec = pair_off[l] + TK_UNIT_Q * qg;
    // table row offsets (in uint4) of the four queries; q and f0 are re-read at the
    // end instead of being held across the loop (register pressure)
    int tq[TK_UNIT_Q];
#pragma unroll
    for (int i = 0; i < TK_UNIT_Q; i++) {
        const int qi = pair_q[rec + i];
        tq[i] = (qi < 0 ? 0 : qi) * M;
    }
    const int64_t gc = c0 + c;
    const uint4 *src = codes + ((gc >> 3) * (int64_t)P) * 8 + (gc & 7);
    uint32_t a0[TK_UNIT_Q][8], a1[TK_UNIT_Q][8];
#pragma unroll
    for (int i = 0; i < TK_UNIT_Q; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) a0[i][j] = a1[i][j] = 0;

    const int steps = (ORDER == TK_ORDER_AVX) ? (P >> 1) : P;
    for (int st = 0; st < steps; st++) {
        // AVX: pairs 2*st (accumulator set 0) and 2*st+1 (set 1); SSE: pair st (set 0)
        const int p0 = (ORDER == TK_ORDER_AVX) ? 2 * st : st;
        const uint4 x0 = src[p0 * 8];
        uint4 x1 = make_uint4(0, 0, 0, 0);
        if (ORDER == TK_ORDER_AVX) x1 = src[(p0 + 1) * 8];
        {   // pair p0 -> accumulator set 0.  Dword-outer / query-inner keeps the six
            // selector words of one dword live instead of those of all eight.
            uint4 tl[TK_UNIT_Q], th[TK_UNIT_Q];
#pragma unroll
            for (int i = 0; i < TK_UNIT_Q; i++) { tl[i] = tables[tq[i] + 2 * p0]; th[i] = tables[tq[i] + 2 * p0 + 1]; }
            const uint32_t xs0[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const Sel6 s = make_sel(xs0[j], cA, cB);
#pragma unroll
                for (int i = 0; i < TK_UNIT_Q; i++) {
                    lut4x<SIGNED>(s.s_lo, s.kA_lo, s.kB_lo, tl[i], a0[i][2 * j], a0[i][2 * j + 1]);
                    lut4x<SIGNED>(s.s_hi, s.kA_hi, s.kB_hi, th[i], a0[i][2 * j], a0[i][2 * j + 1]);
                }
            }
        }
        if (ORDER == TK_ORDER_AVX) {   // pair p0+1 -> accumulator set 1
            uint4 tl[TK_UNIT_Q], th[TK_UNIT_Q];
#pragma unroll
            for (int i = 0; i < TK_UNIT_Q; i++) { tl[i] = tables[tq[i] + 2 * p0 + 2]; th[i] = tables[tq[i] + 2 * p0 + 3]; }
            const uint32_t xs1[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const Sel6 s = make_sel(xs1[j], cA, cB);
#pragma unroll
                for (int i = 0; i < TK_UNIT_Q; i++) {
                    lut4x<SIGNED>(s.s_lo, s.kA_lo, s.kB_lo, tl[i], a1[i][2 * j], a1[i][2 * j + 1]);
                    lut4x<SIGNED>(s.s_hi, s.kA_hi, s.kB_hi, th[i], a1[i][2 * j], a1[i][2 * j + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TK_UNIT_Q; i++) {
        if (ORDER == TK_ORDER_AVX) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                a0[i][j] = sat_add2<SIGNED>(a0[i][j], a1[i][j] & 0xff00ff00u);
        }
        uint4 o;
        uint32_t mn;
        finish_chunk<SIGNED>(a0[i], o, mn);
        const int qi = pair_q[rec + i];
        if (active && qi >= 0) {
            const int f0 = pair_f0[rec + i];
            dist[(int64_t)qi * cap + f0 + c] = o;
            if (mins) mins[(int64_t)qi * min_stride + f0 + c] = (uint8_t)mn;
        }
    }
}

// ---- the same block with the table rows in LDS --------------------------------------------
// In scan_units_block every pair-step issues 8 per-lane global_load_dwordx4 of table rows
// (4 queries x 2 rows of 16 entries): 234 vector-memory instructions per 64-unit block against
// ~6400 VALU, all through the CU's one texture-address path, which 12 waves share.  Here the
// rows a block needs are copied ONCE per block into the wave's own LDS region and read back
// with ds_read_b128 (64 dwords/clk, identical addresses broadcast): the units of a block are
// consecutive, so they belong to a few consecutive groups of 4 queries (`grp` = record / 4 is
// monotone in the unit number), typically one or two.  A wave whose block spans more groups
// than its region holds (short lists) makes several passes over windows of `gmax` groups with
// the lanes of other windows idle — same code, no second path.  Per-wave regions: no
// workgroup barrier, waves keep drawing blocks independently.
template <int ORDER, bool SIGNED, bool TIGHT>
__device__ __forceinline__ void scan_units_block_lds(
    const uint4 *__restrict__ codes, int P, const uint4 *__restrict__ tables, int M,
    const int64_t *__restrict__ list_chunk_off, int n_lists, const int *__restrict__ unit_prefix,
    const int *__restrict__ pair_off, const int *__restrict__ pair_q,
    const int *__restrict__ pair_f0, uint4 *__restrict__ dist, int64_t cap,
    uint8_t *__restrict__ mins, int64_t min_stride, int U, int blk, uint4 *lw, int gmax)
{
    const uint32_t cA = 0x020c000cu, cB = 0x030c010cu;
    const int lane = threadIdx.x & 63;
    const int u = (blk << 6) + lane;
    const bool active = u < U;
    const int uu = active ? u : U - 1;
    int lo = 0, hi = n_lists;   // unit_prefix[lo] <= uu < unit_prefix[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (unit_prefix[mid] <= uu) lo = mid; else hi = mid;
    }
    const int l = lo;
    const int64_t c0 = list_chunk_off[l];
    const int C = (int)(list_chunk_off[l + 1] - c0);
    const int local = uu - unit_prefix[l];
    const int qg = local / C, c = local - qg * C;
    const int rec = pair_off[l] + TK_UNIT_Q * qg;
    const int grp = rec >> 2;
    const int g_first = __builtin_amdgcn_readfirstlane(grp);
    const int g_last = __builtin_amdgcn_readlane(grp, 63);
    const int64_t gc = c0 + c;
    const uint4 *src = codes + ((gc >> 3) * (int64_t)P) * 8 + (gc & 7);
    const int steps = (ORDER == TK_ORDER_AVX) ? (P >> 1) : P;

    for (int gw = g_first; gw <= g_last; gw += gmax) {
        const int gn = g_last - gw + 1 < gmax ? g_last - gw + 1 : gmax;
        // fill: record r of the window -> M rows at lw[r * M]; the query of a record is
        // wave-uniform, lanes copy consecutive rows
        for (int r = 0; r < gn * TK_UNIT_Q; r++) {
            int qi = pair_q[gw * TK_UNIT_Q + r];
            qi = __builtin_amdgcn_readfirstlane(qi < 0 ? 0 : qi);
            const uint4 *trow = tables + (int64_t)qi * M;
            for (int m = lane; m < M; m += 64) lw[r * M + m] = trow[m];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool mine = grp >= gw && grp < gw + gn;
        const uint4 *lt = lw + (mine ? (grp - gw) : 0) * (TK_UNIT_Q * M);

        uint32_t a0[TK_UNIT_Q][8], a1[TK_UNIT_Q][8];
#pragma unroll
        for (int i = 0; i < TK_UNIT_Q; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) a0[i][j] = a1[i][j] = 0;
        for (int st = 0; st < steps; st++) {
            const int p0 = (ORDER == TK_ORDER_AVX) ? 2 * st : st;
            // pair p0 -> accumulator set 0 (and, AVX, pair p0+1 -> set 1).  Pair-outer,
            // query-inner: the 24 selector words of a pair's four dwords are shared by the four
            // queries, whose table rows arrive just in time from LDS (8 VGPRs per query instead
            // of 32 for all four: that is what lets this form run 4 waves per SIMD).
#pragma unroll
            for (int h = 0; h < (ORDER == TK_ORDER_AVX ? 2 : 1); h++) {
                const uint4 x = src[(p0 + h) * 8];
                const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
                Sel6 sl[4];
#pragma unroll
                for (int j = 0; j < 4; j++) sl[j] = make_sel(xs[j], cA, cB);
#pragma unroll
                for (int i = 0; i < TK_UNIT_Q; i++) {
                    const uint4 tl = lt[i * M + 2 * (p0 + h)], th = lt[i * M + 2 * (p0 + h) + 1];
                    uint32_t(&acc)[8] = h == 0 ? a0[i] : a1[i];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        lut4x<SIGNED>(sl[j].s_lo, sl[j].kA_lo, sl[j].kB_lo, tl, acc[2 * j], acc[2 * j + 1]);
                        lut4x<SIGNED>(sl[j].s_hi, sl[j].kA_hi, sl[j].kB_hi, th, acc[2 * j], acc[2 * j + 1]);
                    }
                    // keep the next query's rows from being hoisted above this one's work:
                    // 8 live row registers instead of 32 (the difference is a spill at 128)
                    if (TIGHT) asm volatile("" ::: "memory");
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TK_UNIT_Q; i++) {
            if (ORDER == TK_ORDER_AVX) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    a0[i][j] = sat_add2<SIGNED>(a0[i][j], a1[i][j] & 0xff00ff00u);
            }
            uint4 o;
            uint32_t mn;
            finish_chunk<SIGNED>(a0[i], o, mn);
            const int qi = pair_q[rec + i];
            if (active && mine && qi >= 0) {
                const int f0 = pair_f0[rec + i];
                dist[(int64_t)qi * cap + f0 + c] = o;
                if (mins) mins[(int64_t)qi * min_stride + f0 + c] = (uint8_t)mn;
            }
        }
        // the next window (or block) overwrites the region: every lane's reads come first
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// bytes of LDS per wave for the table rows: 16 waves per CU x 9984 B = 156 KiB of the 160
// (form 1, 4 waves/SIMD); 12 waves x 13312 B (form 2, 3 waves/SIMD)
#define TK_LDS_WAVE_BYTES 9984
#define TK_LDS_WAVE_BYTES_MAX 13312

// COARSE only tags the instantiation used for the coded centres so that profilers
// list the two launches of a batch separately.  LDS_T: table rows through LDS (gmax > 0).
// FORM: 0 = table rows by per-lane global loads (3 waves/SIMD), 1 = rows through LDS, rows of
// one query live at a time (127 VGPRs, 4 waves/SIMD), 2 = rows through LDS, scheduler free
// (3 waves/SIMD)
template <int ORDER, bool SIGNED, int FORM, bool COARSE>
__global__ __launch_bounds__(256, (FORM == 1 ? 4 : 3)) void scan_units_kernel(
    const uint4 *__restrict__ codes, int P, const uint4 *__restrict__ tables, int M,
    const int64_t *__restrict__ list_chunk_off, int n_lists,
    const int *__restrict__ unit_prefix,   // (n_lists+1) units before each list, then tickets
    const int *__restrict__ pair_off,      // (n_lists+1) first record of each list (x4 padded)
    const int *__restrict__ pair_q,        // query of a record, -1 = padding
    const int *__restrict__ pair_f0,       // first flat chunk of that (query, slot) row range
    uint4 *__restrict__ dist, int64_t cap, uint8_t *__restrict__ mins, int64_t min_stride, int gmax)
{
    extern __shared__ uint4 tk_lds_tables[];
    uint4 *lw = tk_lds_tables + (threadIdx.x >> 6) * (gmax * TK_UNIT_Q * M);
    const int U = unit_prefix[n_lists];
    int *ticket = const_cast<int *>(unit_prefix) + TK_TICKET_OFF(n_lists);
    ticketed_blocks((U + 63) >> 6, ticket, [&](int blk) {
        if (FORM != 0)
            scan_units_block_lds<ORDER, SIGNED, FORM == 1>(codes, P, tables, M, list_chunk_off, n_lists,
                                                unit_prefix, pair_off, pair_q, pair_f0, dist, cap,
                                                mins, min_stride, U, blk, lw, gmax);
        else
            scan_units_block<ORDER, SIGNED>(codes, P, tables, M, list_chunk_off, n_lists, unit_prefix,
                                            pair_off, pair_q, pair_f0, dist, cap, mins, min_stride, U,
                                            blk);
    });
}

// Two jobs in one launch, one pool of blocks: the list scan of batch b and the coarse scan
// of batch b+1 (pipelined mode, api.hip).  Either job may be absent (unit_prefix == NULL).
// The work counters are those of the first job present.
// The three jobs arrive as ONE by-value argument, the first one, and are read where they lie — in
// the kernel-argument segment, through scalar loads at a wave-uniform offset.  (Selecting among
// three by-value structs, `which == 0 ? a : ...`, made the compiler copy all of them to private
// memory: 480 B of scratch per lane on every instantiation; tests/test_kernel_resources.py.)
struct TkScanJobs3 { TkScanJob j[3]; };
typedef const __attribute__((address_space(4))) TkScanJob tk_kernarg_job;

template <int ORDER, bool SIGNED, int FORM>
__global__ __launch_bounds__(256, (FORM == 1 ? 4 : 3)) void scan_units2_kernel(TkScanJobs3 jobs_by_value, int P,
                                                                int M, int gmax)
{
    extern __shared__ uint4 tk_lds_tables[];
    uint4 *lw = tk_lds_tables + (threadIdx.x >> 6) * (gmax * TK_UNIT_Q * M);
    (void)jobs_by_value;        // (offset 0 of the kernel-argument segment)
    tk_kernarg_job *jobs = (tk_kernarg_job *)__builtin_amdgcn_kernarg_segment_ptr();
    const int *upa = jobs[0].unit_prefix, *upb = jobs[1].unit_prefix, *upc = jobs[2].unit_prefix;
    const int UA = upa ? upa[jobs[0].n_lists] : 0;
    const int UB = upb ? upb[jobs[1].n_lists] : 0;
    const int UC = upc ? upc[jobs[2].n_lists] : 0;
    const int NBA = (UA + 63) >> 6, NBB = (UB + 63) >> 6, NBC = (UC + 63) >> 6;
    int *ticket = upa ? const_cast<int *>(upa) + TK_TICKET_OFF(jobs[0].n_lists)
                : upb ? const_cast<int *>(upb) + TK_TICKET_OFF(jobs[1].n_lists)
                      : const_cast<int *>(upc) + TK_TICKET_OFF(jobs[2].n_lists);
    // (captures by VALUE: by reference, `which == 0 ? UA : ...` becomes a select between the addresses
    // of the closure's members, and the closure goes to scratch)
    ticketed_blocks(NBA + NBB + NBC, ticket, [=](int blk) {
        // one copy of the block body; `which` is wave-uniform
        const int which = blk < NBA ? 0 : (blk < NBA + NBB ? 1 : 2);
        tk_kernarg_job &j = jobs[which];
        // (masks, not `which == 0 ? UA : ...`: the compiler turns that into a load through a selected
        // ADDRESS of the closure's members, which pins the closure in scratch)
        const int U = (UA & -(int)(which == 0)) | (UB & -(int)(which == 1)) | (UC & -(int)(which == 2));
        const int jb = blk - (NBA & -(int)(which >= 1)) - (NBB & -(int)(which == 2));
        if (FORM != 0)
            scan_units_block_lds<ORDER, SIGNED, FORM == 1>(j.codes, P, j.tables, M, j.list_chunk_off, j.n_lists,
                                                j.unit_prefix, j.pair_off, j.pair_q, j.pair_f0, j.dist,
                                                j.cap, j.mins, j.min_stride, U, jb, lw, gmax);
        else
            scan_units_block<ORDER, SIGNED>(j.codes, P, j.tables, M, j.list_chunk_off, j.n_lists,
                                            j.unit_prefix, j.pair_off, j.pair_q, j.pair_f0, j.dist,
                                            j.cap, j.mins, j.min_stride, U, jb, j.max_chunks);
    });
}

// ---- pair lists: (query, slot) pairs grouped by the list they probe ------------
// (the per-list pair counts come from make_slots_kernel)
// one workgroup: exclusive scans over the lists
__global__ __launch_bounds__(1024) void pairs_scan_kernel(int *__restrict__ count,
                                                          const int64_t *__restrict__ list_chunk_off,
                                                          int n_lists, int *__restrict__ pair_off,
                                                          int *__restrict__ unit_prefix,
                                                          int *__restrict__ cursor,
                                                          int *__restrict__ pair_q)
{
    __shared__ int s_rec[1024], s_unit[1024];
    __shared__ int carry_rec, carry_unit;
    if (threadIdx.x == 0) carry_rec = carry_unit = 0;
    __syncthreads();
    for (int base = 0; base < n_lists; base += 1024) {
        const int l = base + threadIdx.x;
        int rec = 0, unit = 0, cnt = 0;
        if (l < n_lists) {
            cnt = count[l];
            count[l] = 0;   // zero again for the next batch
            const int groups = (cnt + TK_UNIT_Q - 1) / TK_UNIT_Q;
            rec = groups * TK_UNIT_Q;
            unit = groups * (int)(list_chunk_off[l + 1] - list_chunk_off[l]);
        }
        s_rec[threadIdx.x] = rec;
        s_unit[threadIdx.x] = unit;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
            int r = threadIdx.x >= o ? s_rec[threadIdx.x - o] : 0;
            int un = threadIdx.x >= o ? s_unit[threadIdx.x - o] : 0;
            __syncthreads();
            s_rec[threadIdx.x] += r;
            s_unit[threadIdx.x] += un;
            __syncthreads();
        }
        if (l < n_lists) {
            const int off = carry_rec + s_rec[threadIdx.x] - rec;
            pair_off[l] = off;
            unit_prefix[l] = carry_unit + s_unit[threadIdx.x] - unit;
            cursor[l] = 0;
            for (int t = cnt; t < rec; t++) pair_q[off + t] = -1;   // padding records
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry_rec += s_rec[1023];
            carry_unit += s_unit[1023];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pair_off[n_lists] = carry_rec;
        unit_prefix[n_lists] = carry_unit;
    }
    if (threadIdx.x < TK_TICKETS) unit_prefix[TK_TICKET_OFF(n_lists) + threadIdx.x * 32] = 0;
}

__global__ void pairs_fill_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                  int64_t n_lists, const int *__restrict__ slot_prefix,
                                  const int *__restrict__ pair_off, int *__restrict__ cursor,
                                  int *__restrict__ pair_q, int *__restrict__ pair_f0)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * S) return;
    const int64_t qi = i / S;
    const int s = (int)(i - qi * S);
    int64_t cl = probes[i];
    if (cl < 0) cl += n_lists;
    const int pos = atomicAdd(&cursor[cl], 1);
    pair_q[pair_off[cl] + pos] = (int)qi;
    pair_f0[pair_off[cl] + pos] = slot_prefix[qi * (S + 1) + s];
}

void tk_launch_unit_pairs(int64_t nq, const int64_t *probes, int S, int64_t n_lists,
                          const int64_t *list_chunk_off, const int *slot_prefix, int *count,
                          int *pair_off, int *unit_prefix, int *cursor, int *pair_q, int *pair_f0,
                          int64_t max_records, hipStream_t s)
{
    if (nq == 0 || S == 0) return;
    const int64_t np = nq * S;
    (void)max_records;
    hipLaunchKernelGGL(pairs_scan_kernel, dim3(1), dim3(1024), 0, s, count, list_chunk_off,
                       (int)n_lists, pair_off, unit_prefix, cursor, pair_q);
    hipLaunchKernelGGL(pairs_fill_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s,
                       probes, S, nq, n_lists, slot_prefix, pair_off, cursor, pair_q, pair_f0);
}

// ---- three pair sets in one pass (plain_scan.hip): set 0 = slots below slot_exact[q] -> the
// exact kernel's units (records padded to groups of 4); set 1 = the plain kernel's tiles of 32
// pairs (records not padded); set 2 = head pairs (slot 0 of a query in head mode, also in set 1)
// -> units of the exact kernel over the first head_chunks chunks of the list.
struct PairSets {
    int *count[3], *cursor[3], *pair_off[3], *unit_prefix[3], *pair_q[3], *pair_f0[3];
    int *unit_desc;     // set 1: int4 (list, tile, first chunk pair, end chunk pair) per unit
    int plain_k;        // set 1: chunk pairs per unit (a multiple of 4)
};

// Units of the plain kernel (plain_scan.hip: one wave per unit): a tile of 32 of a list's pairs
// times a range of at most K chunk pairs, numbered (list, tile, range) with the range fastest —
// the waves of a workgroup take neighbouring units, i.e. ranges of ONE tile, and read the same
// table rows.  unit_prefix[l] = units before list l.
__device__ __forceinline__ int plain_units_of(int cnt, int C, int K)
{
    if (cnt <= 0 || C <= 0) return 0;
    const int CP = (C + 1) >> 1;
    return ((cnt + 31) >> 5) * ((CP + K - 1) / K);
}

__device__ __forceinline__ void plain_desc_fill(const int *__restrict__ unit_prefix,
                                                const int64_t *__restrict__ list_chunk_off, int n_lists,
                                                int K, int4 *__restrict__ desc, int u)
{
    int lo = 0, hi = n_lists;       // unit_prefix[lo] <= u < unit_prefix[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (unit_prefix[mid] <= u) lo = mid; else hi = mid;
    }
    const int local = u - unit_prefix[lo];
    const int C = (int)(list_chunk_off[lo + 1] - list_chunk_off[lo]);
    const int CP = (C + 1) >> 1;
    const int nsub = (CP + K - 1) / K;
    const int t = local / nsub, sb = local - t * nsub;
    const int a = sb * K, b = a + K < CP ? a + K : CP;
    desc[u] = make_int4(lo, t, a, b);
}

__global__ void plain_desc_kernel(const int *__restrict__ unit_prefix, const int64_t *__restrict__ list_chunk_off,
                                  int n_lists, int K, int4 *__restrict__ desc)
{
    const int U = unit_prefix[n_lists];
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x)
        plain_desc_fill(unit_prefix, list_chunk_off, n_lists, K, desc, u);
}

void tk_launch_plain_desc(const TkPairSet &pl, const int64_t *list_chunk_off, int64_t n_lists, hipStream_t s)
{
    hipLaunchKernelGGL(plain_desc_kernel, dim3(256), dim3(256), 0, s, pl.unit_prefix, list_chunk_off,
                       (int)n_lists, pl.plain_k, (int4 *)pl.unit_desc);
}

__global__ __launch_bounds__(1024) void pairs_scan3_kernel(PairSets ps, const int64_t *__restrict__ list_chunk_off,
                                                           int n_lists, int head_chunks)
{
    __shared__ int s_a[1024], s_b[1024];
    __shared__ int carry_a, carry_b;
    for (int set = 0; set < 3; set++) {
        if (threadIdx.x == 0) carry_a = carry_b = 0;
        __syncthreads();
        int *cnt_p = ps.count[set], *off_p = ps.pair_off[set], *unit_p = ps.unit_prefix[set];
        int *cur_p = ps.cursor[set], *pq = ps.pair_q[set];
        for (int base = 0; base < n_lists; base += 1024) {
            const int l = base + threadIdx.x;
            int rec = 0, unit = 0, cnt = 0;
            if (l < n_lists) {
                cnt = cnt_p[l];
                cnt_p[l] = 0;   // zero again for the next batch
                const int C = (int)(list_chunk_off[l + 1] - list_chunk_off[l]);
                if (set != 1) {
                    const int groups = (cnt + TK_UNIT_Q - 1) / TK_UNIT_Q;
                    rec = groups * TK_UNIT_Q;
                    unit = groups * (set == 2 && head_chunks < C ? head_chunks : C);
                } else {
                    rec = cnt;
                    unit = plain_units_of(cnt, C, ps.plain_k);
                }
            }
            s_a[threadIdx.x] = rec;
            s_b[threadIdx.x] = unit;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
                int ra = threadIdx.x >= o ? s_a[threadIdx.x - o] : 0;
                int rb = threadIdx.x >= o ? s_b[threadIdx.x - o] : 0;
                __syncthreads();
                s_a[threadIdx.x] += ra;
                s_b[threadIdx.x] += rb;
                __syncthreads();
            }
            if (l < n_lists) {
                const int off = carry_a + s_a[threadIdx.x] - rec;
                const int ubase = carry_b + s_b[threadIdx.x] - unit;
                off_p[l] = off;
                unit_p[l] = ubase;
                cur_p[l] = 0;
                if (set != 1)
                    for (int t = cnt; t < rec; t++) pq[off + t] = -1;   // padding records
            }
            __syncthreads();
            if (threadIdx.x == 1023) {
                carry_a += s_a[1023];
                carry_b += s_b[1023];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            off_p[n_lists] = carry_a;
            unit_p[n_lists] = carry_b;
        }
        if (threadIdx.x < TK_TICKETS) unit_p[TK_TICKET_OFF(n_lists) + threadIdx.x * 32] = 0;
        __syncthreads();
    }
}

// pos / owner / me / ovf_pos (list-sharded index, api_shard.hip): only the pairs of the lists `me` owns get
// records (make_slots counted only those), and a record's row offset is the segment's position in the send
// buffer, pos[i].  pos[i] < 0 = the segment did not fit its region (the batch is flagged and repeated): a
// padding record for the exact kernel; the plain kernel has none and scores the pair to ovf_pos, the
// tail of the buffer, where the longest list fits.
__global__ void pairs_fill3_kernel(const int64_t *__restrict__ probes, int S, int64_t nq,
                                   int64_t n_lists, const int *__restrict__ slot_prefix,
                                   const int *__restrict__ slot_exact, PairSets ps,
                                   const int64_t *__restrict__ list_chunk_off,
                                   const int *__restrict__ pos, const int *__restrict__ owner, int me,
                                   int ovf_pos)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {   // the plain kernel's unit descriptors (unit_prefix is complete: pairs_scan3_kernel ran before)
        const int U = ps.unit_prefix[1][n_lists];
        for (int64_t u = i; u < U; u += (int64_t)gridDim.x * blockDim.x)
            plain_desc_fill(ps.unit_prefix[1], list_chunk_off, (int)n_lists, ps.plain_k, (int4 *)ps.unit_desc, (int)u);
    }
    if (i >= nq * S) return;
    const int64_t qi = i / S;
    const int s = (int)(i - qi * S);
    int64_t cl = probes[i];
    if (cl < 0) cl += n_lists;
    if (owner && owner[cl] != me) return;
    const int f0 = pos ? pos[i] : slot_prefix[qi * (S + 1) + s];
    const bool ovf = pos && f0 < 0;
    const int e = slot_exact[qi];
    auto put = [&](int set) {
        const int at = ps.pair_off[set][cl] + atomicAdd(&ps.cursor[set][cl], 1);
        ps.pair_q[set][at] = (ovf && set != 1) ? -1 : (int)qi;
        ps.pair_f0[set][at] = ovf ? (set == 1 ? ovf_pos : 0) : f0;
    };
    if (e == 0 && s == 0) {        // head mode: first chunks exact, the whole list plain (overwritten)
        put(2);
        put(1);
    } else {
        put(s < e ? 0 : 1);
    }
}

static PairSets pair_sets(const TkPairSet &ex, const TkPairSet &pl, const TkPairSet &hd)
{
    PairSets ps;
    const TkPairSet *sets[3] = {&ex, &pl, &hd};
    for (int i = 0; i < 3; i++) {
        ps.count[i] = sets[i]->count; ps.cursor[i] = sets[i]->cursor; ps.pair_off[i] = sets[i]->pair_off;
        ps.unit_prefix[i] = sets[i]->unit_prefix; ps.pair_q[i] = sets[i]->pair_q; ps.pair_f0[i] = sets[i]->pair_f0;
    }
    ps.unit_desc = pl.unit_desc;
    ps.plain_k = pl.plain_k < 4 ? 12 : (pl.plain_k + 3) & ~3;
    return ps;
}

void tk_launch_pairs_scan3(const TkPairSet &ex, const TkPairSet &pl, const TkPairSet &hd,
                           const int64_t *list_chunk_off, int64_t n_lists, int head_chunks,
                           hipStream_t s)
{
    hipLaunchKernelGGL(pairs_scan3_kernel, dim3(1), dim3(1024), 0, s, pair_sets(ex, pl, hd),
                       list_chunk_off, (int)n_lists, head_chunks);
}

void tk_launch_unit_pairs2(int64_t nq, const int64_t *probes, int S, int64_t n_lists,
                           const int64_t *list_chunk_off, const int *slot_prefix,
                           const int *slot_exact, const TkPairSet &ex, const TkPairSet &pl,
                           const TkPairSet &hd, int head_chunks, hipStream_t s, const int *pos,
                           const int *owner, int me, int ovf_pos)
{
    if (nq == 0 || S == 0) return;
    const int64_t np = nq * S;
    PairSets ps = pair_sets(ex, pl, hd);
    hipLaunchKernelGGL(pairs_scan3_kernel, dim3(1), dim3(1024), 0, s, ps, list_chunk_off, (int)n_lists,
                       head_chunks);
    hipLaunchKernelGGL(pairs_fill3_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, probes, S,
                       nq, n_lists, slot_prefix, slot_exact, ps, list_chunk_off, pos, owner, me, ovf_pos);
}

void tk_launch_pairs_scan(int *count, const int64_t *list_chunk_off, int64_t n_lists, int *pair_off,
                          int *unit_prefix, int *cursor, int *pair_q, hipStream_t s)
{
    hipLaunchKernelGGL(pairs_scan_kernel, dim3(1), dim3(1024), 0, s, count, list_chunk_off,
                       (int)n_lists, pair_off, unit_prefix, cursor, pair_q);
}

// Descriptors for "every query scans the one list" (the coarse stage): records are
// the queries themselves, padded to a multiple of 4.
__global__ void pairs_identity_kernel(int64_t nq, int chunks, int *__restrict__ pair_off,
                                      int *__restrict__ unit_prefix, int *__restrict__ pair_q,
                                      int *__restrict__ pair_f0)
{
    const int64_t nrec = (nq + TK_UNIT_Q - 1) / TK_UNIT_Q * TK_UNIT_Q;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrec) {
        pair_q[i] = i < nq ? (int)i : -1;
        pair_f0[i] = 0;
    }
    if (i == 0) {
        pair_off[0] = 0;
        pair_off[1] = (int)nrec;
        unit_prefix[0] = 0;
        unit_prefix[1] = (int)(nrec / TK_UNIT_Q) * chunks;
        for (int t = 0; t < TK_TICKETS; t++) unit_prefix[TK_TICKET_OFF(1) + t * 32] = 0;
    }
}

// The same for the plain kernel (one wave per unit): the one list, pair i = query i, tiles of 32
// consecutive queries x ranges of K chunk pairs (the range fastest).
__global__ void plain_identity_kernel(int64_t nq, int chunks, int K, int *__restrict__ pair_off,
                                      int *__restrict__ unit_prefix, int *__restrict__ pair_q,
                                      int *__restrict__ pair_f0, int4 *__restrict__ desc)
{
    const int CP = (chunks + 1) >> 1;
    const int nsub = (CP + K - 1) / K;
    const int64_t U = ((nq + 31) / 32) * nsub;
    const int64_t nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (U > nq ? U : nq); i += nt) {
        if (i < nq) {
            pair_q[i] = (int)i;
            pair_f0[i] = 0;
        }
        if (i < U) {
            const int t = (int)(i / nsub), sb = (int)(i - (int64_t)t * nsub);
            const int a = sb * K, b = a + K < CP ? a + K : CP;
            desc[i] = make_int4(0, t, a, b);
        }
        if (i == 0) {
            pair_off[0] = 0;
            pair_off[1] = (int)nq;
            unit_prefix[0] = 0;
            unit_prefix[1] = (int)U;
            for (int t = 0; t < TK_TICKETS; t++) unit_prefix[TK_TICKET_OFF(1) + t * 32] = 0;
        }
    }
}

void tk_launch_plain_identity(int64_t nq, int chunks, const TkPairSet &pl, hipStream_t s)
{
    if (nq == 0 || chunks == 0) return;
    const int K = pl.plain_k < 4 ? 12 : (pl.plain_k + 3) & ~3;
    hipLaunchKernelGGL(plain_identity_kernel, dim3(256), dim3(256), 0, s, nq, chunks, K, pl.pair_off,
                       pl.unit_prefix, pl.pair_q, pl.pair_f0, (int4 *)pl.unit_desc);
}

void tk_launch_identity_pairs(int64_t nq, int chunks, int *pair_off, int *unit_prefix, int *pair_q,
                              int *pair_f0, hipStream_t s)
{
    if (nq == 0) return;
    const int64_t nrec = (nq + TK_UNIT_Q - 1) / TK_UNIT_Q * TK_UNIT_Q;
    hipLaunchKernelGGL(pairs_identity_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s,
                       nq, chunks, pair_off, unit_prefix, pair_q, pair_f0);
}

// (form of the list-major kernel — tk_index_set_option(TK_OPT_SCAN_FORM): 0 = per-lane global loads of
//  the table rows, the default: profiles/r02_scan_forms.md; 1 / 2 = rows staged per block in LDS)

template <typename K>
static void lds_attr(K kern)
{
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              4 * TK_LDS_WAVE_BYTES_MAX);
}

// groups of 4 queries whose rows fit a wave's LDS region (0: use the global form)
static int scan_form_gmax(int M, int &form)
{
    if (form == 0) return 0;
    const int bytes = form == 1 ? TK_LDS_WAVE_BYTES : TK_LDS_WAVE_BYTES_MAX;
    int g = bytes / (TK_UNIT_Q * 16 * M);
    g = g > 8 ? 8 : g;
    if (g < 1) form = 0;
    return g;
}

void tk_launch_scan_units(const uint4 *codes, int M, const uint4 *tables, int64_t nq, int S,
                          int64_t n_lists, const int64_t *list_chunk_off, const int *pair_off,
                          const int *unit_prefix, const int *pair_q, const int *pair_f0,
                          uint4 *dist, int64_t cap, uint8_t *mins, int64_t min_stride, int signd,
                          int order, int n_blocks, hipStream_t s, int form)
{
    if (nq == 0 || S == 0) return;
    const int P = M / 2;
    const bool coarse = n_lists == 1 && S == 1;
    form = form < 0 || form > 2 ? 0 : form;
    const int gmax = scan_form_gmax(M, form);
    const size_t lds = (size_t)4 * gmax * TK_UNIT_Q * M * 16;
#define TK_LAUNCH3(O, S_, F_, C_)                                                                \
    do {                                                                                         \
        static bool attr_ = false;                                                               \
        if (F_ != 0 && !attr_) { lds_attr(scan_units_kernel<O, S_, F_, C_>); attr_ = true; }     \
        hipLaunchKernelGGL((scan_units_kernel<O, S_, F_, C_>), dim3(n_blocks), dim3(256), lds, s, \
                           codes, P, tables, M, list_chunk_off, (int)n_lists, unit_prefix, pair_off, \
                           pair_q, pair_f0, dist, cap, mins, min_stride, gmax);                      \
    } while (0)
#define TK_LAUNCH2(O, S_, C_)                                          \
    do {                                                               \
        if (form == 1) TK_LAUNCH3(O, S_, 1, C_);                       \
        else if (form == 2) TK_LAUNCH3(O, S_, 2, C_);                  \
        else TK_LAUNCH3(O, S_, 0, C_);                                 \
    } while (0)
#define TK_LAUNCH(O, S_)                                               \
    do {                                                               \
        if (coarse) TK_LAUNCH2(O, S_, true);                           \
        else TK_LAUNCH2(O, S_, false);                                 \
    } while (0)
    if (order == TK_ORDER_AVX) {
        if (signd) TK_LAUNCH(TK_ORDER_AVX, true); else TK_LAUNCH(TK_ORDER_AVX, false);
    } else {
        if (signd) TK_LAUNCH(TK_ORDER_SSE, true); else TK_LAUNCH(TK_ORDER_SSE, false);
    }
#undef TK_LAUNCH
#undef TK_LAUNCH2
#undef TK_LAUNCH3
}

void tk_launch_scan_units2(const TkScanJob &a, const TkScanJob &b, int M, int order, int n_blocks,
                           hipStream_t s, const TkScanJob *cj, int form)
{
    TkScanJobs3 jobs;
    memset(&jobs, 0, sizeof jobs);
    jobs.j[0] = a;
    jobs.j[1] = b;
    if (cj) jobs.j[2] = *cj;
    const TkScanJob &c = jobs.j[2];
    if (!a.unit_prefix && !b.unit_prefix && !c.unit_prefix) return;
    const int P = M / 2;
    form = form < 0 || form > 2 ? 0 : form;
    if (a.max_chunks || b.max_chunks || c.max_chunks) form = 0;      // (only the global-load form, the default, knows heads)
    const int gmax = scan_form_gmax(M, form);
    const size_t lds = (size_t)4 * gmax * TK_UNIT_Q * M * 16;
#define TK_LAUNCH(O, F_)                                                                          \
    do {                                                                                          \
        static bool attr_ = false;                                                                \
        if (F_ != 0 && !attr_) { lds_attr(scan_units2_kernel<O, true, F_>); attr_ = true; }       \
        hipLaunchKernelGGL((scan_units2_kernel<O, true, F_>), dim3(n_blocks), dim3(256), lds, s,  \
                           jobs, P, M, gmax);                                                     \
    } while (0)
#define TK_LAUNCH1(O)                                  \
    do {                                               \
        if (form == 1) TK_LAUNCH(O, 1);                \
        else if (form == 2) TK_LAUNCH(O, 2);           \
        else TK_LAUNCH(O, 0);                          \
    } while (0)
    if (order == TK_ORDER_AVX) TK_LAUNCH1(TK_ORDER_AVX); else TK_LAUNCH1(TK_ORDER_SSE);
#undef TK_LAUNCH1
#undef TK_LAUNCH
}
