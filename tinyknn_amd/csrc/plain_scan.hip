// plain_scan.hip — the probed lists BEHIND the first ones on the int8 matrix cores (gfx950).
//
// What it replaces: the same compute_block_dists_avx / compute_block_dists chain of
// _fast_pq_256.pyx:126-156 / _fast_pq.pyx:209-236 that adc_scan.hip restates with v_perm_b32 and
// saturating v_pk_add_i16 — for the (query, list) pairs where a PLAIN integer sum provably gives
// the same replay.
//
// Lemma (proved in DESIGN §3.1b, checked row by row by scripts/r03_plain_sum_stats.py and
// tests/test_plain_scan_lemma.py).  Let T be a query's signed table, chain c the blocks one
// saturating accumulator of the reference adds up (AVX order: blocks with (m >> 1) & 1 == c;
// SSE order: all blocks), N_c = sum over the chain's blocks of max(0, -min_code T[m][code]) and
// C = 127 - N_0 - N_1.  If N_0 <= 128 and N_1 <= 128 then for every stored code with plain sum S
//      S <  C  =>  the reference's saturated value v == max(S, -128)
//      S >= C  =>  v >= C
// (a clamp at -128 inside a chain is impossible, and a clamp at +127 leaves v >= 127 - N and needs
// a prefix sum > 127, i.e. S >= 128 - N).  So o = clamp(S, -128, 127) equals v wherever v < C and
// is >= C elsewhere.  The reference inserts a row only if its value is below the bound captured
// at its block's start, and that bound never rises (_fast_pq_256.pyx:73-123): once the bound is
// <= C, a replay over o is the replay over v — same inserts, same values, same heap arrays.
// The heap replay checks exactly that (bound at the first plain block <= C, per query) and flags
// the queries for which it does not hold; those are re-scanned by the exact kernel and replayed
// again (api.hip: stage_back).  Nothing is approximated.
//
// S is a one-hot(code) x table contraction, 16 x M deep: v_mfma_i32_32x32x32_i8, one
// instruction per block pair for 32 rows x 32 queries.
//   B operand: lane (q = lane & 31, h = lane >> 5) holds the 16-byte table row of block 2p + h
//              of its pair's query for every block pair p (registers; staged per unit through
//              LDS with coalesced loads, the next unit's tile fetched while this one computes);
//   A operand: rows = the 32 rows of two consecutive 16-row chunks; lane (r, h) turns nibble h
//              of byte r of the chunks' 16-byte group p into a 16-byte one-hot through a
//              256-byte LDS table (ds_read_b128: distinct entries sit on distinct banks);
//   D: lane (q, h) holds rows {0-3, 8-11, 16-19, 24-27} + 4h of query q; clamped to int8, and
//      after two v_permlane32_swap the lane holds the whole 16-byte block of (chunk 2cp + h,
//      query q): one 16-byte store + the block's minimum byte.
// Work: persistent 256-thread workgroups draw units = (list, tile of 32 pairs of that list)
// from a counter; the four waves take the list's chunk pairs round-robin; the code groups of a
// chunk pair arrive by ONE 16-byte load per lane, two chunk pairs ahead, and are re-sliced
// through a per-wave LDS region.
#include <limits.h>
#include <stdlib.h>

#include "kernels.h"
#include "tickets.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifdef TK_PLAIN_CLOCK       // scripts/micro only: the clock the chip holds inside the kernel (s_memtime / s_memrealtime)
__device__ unsigned long long tk_plain_clock[4];
#define TK_CLOCK_BEGIN() const unsigned long long ck_t0_ = __builtin_readcyclecounter(), ck_r0_ = __builtin_amdgcn_s_memrealtime()
#define TK_CLOCK_END()                                                                          \
    do {                                                                                        \
        if ((threadIdx.x & 63) == 0) {                                                          \
            atomicAdd(&tk_plain_clock[0], (unsigned long long)__builtin_readcyclecounter() - ck_t0_);       \
            atomicAdd(&tk_plain_clock[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - ck_r0_);   \
            atomicAdd(&tk_plain_clock[2], 1ull);                                                \
        }                                                                                       \
    } while (0)
#else
#define TK_CLOCK_BEGIN()
#define TK_CLOCK_END()
#endif

// ---------------------------------------------------------------------------
// C of the lemma per query, or TK_PLAIN_NEVER when a chain's negative mass exceeds 128
__global__ __launch_bounds__(256) void table_limits_kernel(const uint4 *__restrict__ tables, int M_used,
                                                           int M, int avx, int64_t nq, int force,
                                                           int *__restrict__ qlim)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    int n0 = 0, n1 = 0;
    for (int m = lane; m < M_used; m += 64) {
        const uint4 t = tables[q * M + m];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
        int mn = 127;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int v = (int)(int8_t)(w[i] >> (8 * b));
                mn = v < mn ? v : mn;
            }
        const int neg = mn < 0 ? -mn : 0;
        if (avx && ((m >> 1) & 1)) n1 += neg; else n0 += neg;
    }
    for (int o = 32; o > 0; o >>= 1) {
        n0 += __shfl_xor(n0, o, 64);
        n1 += __shfl_xor(n1, o, 64);
    }
    if (lane == 0) {
        int c = (n0 <= 128 && n1 <= 128) ? 127 - n0 - n1 : TK_PLAIN_NEVER;
        if (force != INT_MAX && c > force) c = force;     // debug: provoke the re-scan path
        qlim[q] = c;
    }
}

static int g_plain_force = INT_MAX;
void tk_plain_force_limit(int v) { g_plain_force = v; }
int tk_plain_forced(void) { return g_plain_force != INT_MAX; }

void tk_launch_table_limits(const uint4 *tables, int M, int order, int64_t nq, int *qlim, hipStream_t s)
{
    if (nq == 0) return;
    const int avx = order == TK_ORDER_AVX;
    const int M_used = avx ? (M & ~3) : M;       // the AVX kernels read block pairs two at a time
    static bool env = false;
    if (!env) {
        const char *e = getenv("TINYKNN_PLAIN_FORCE_LIMIT");
        if (e) g_plain_force = atoi(e);
        env = true;
    }
    hipLaunchKernelGGL(table_limits_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, tables,
                       M_used, M, avx, nq, g_plain_force, qlim);
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d)
{
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x04000c0cu);
    return lo | hi;
}

__device__ __forceinline__ int clamp8(int x) { return min(max(x, -128), 127); }

// One-hot operand: entry (nibble) of the 256-byte LDS table, addressed as (rotated code dword & 0xf0) |
// table address — ONE v_and_or_b32 because the table is 256-byte aligned (an add of the dynamic LDS
// base, a relocated literal the compiler cannot fold, cost a third instruction per MFMA: 26 of the
// loop's 129 vector instructions, and the loop is issue-bound: profiles/r03/ab_pipeline_knobs.txt)
typedef __attribute__((address_space(3))) const v4i lds_cv4i;
__device__ __forceinline__ v4i one_hot(uint32_t lut0, uint32_t x, uint32_t rot)
{
    const uint32_t a = (__builtin_amdgcn_alignbit(x, x, rot) & 0xf0u) | lut0;
    return *(lds_cv4i *)(uintptr_t)a;
}

// a: lanes 32..63 <-> b: lanes 0..31
__device__ __forceinline__ void swap_halves(uint32_t &a, uint32_t &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// PT: block pairs the registers are sized for (P <= PT at run time)
template <int PT>
struct PlainShape {
    static constexpr int PS = (PT + 3) & ~3;      // dwords per (chunk, dword) row of the code staging
    static constexpr int TROW = 2 * PT + 1;       // uint4 per query row of a staged table tile
    static constexpr int TL = (32 * 2 * PT + 255) / 256;
    static constexpr size_t lds = 256 + (size_t)4 * 8 * PS * 4 + (size_t)2 * 32 * TROW * 16 + 16 + 2 * 64 * 4;
};

// EXACT: P == PT (the common shapes, M = 52 and M = 32): no guards around the block pairs, and
// the one-hot operands are requested DEPTH block pairs ahead of the MFMA that takes them (an LDS
// round trip is 100+ cycles with eight waves on the CU's LDS, an MFMA 32).
//
// The chunk-pair loop holds NO conditional vector-memory operation: loads and stores count
// together, in issue order, in one counter (vmcnt), and behind a branch the compiler can only wait
// for all of them — the prefetched code groups would then wait for the previous iteration's
// scattered stores (measured: 56 % of the wave cycles in s_waitcnt).  So lanes without work of
// their own CLONE a lane that has some — a pair past the tile's last one the last pair, the second
// chunk of an odd chunk pair the first — and load / store the same bytes at the same addresses.
template <int PT, bool EXACT>
__global__ __launch_bounds__(256, 2) void scan_plain_kernel(TkScanJob j, int P, int M)
{
    if (EXACT) P = PT;
    TK_CLOCK_BEGIN();
    using SH = PlainShape<PT>;
    constexpr int PS = SH::PS, TROW = SH::TROW, TL = SH::TL;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem_plain[];
    uint4 *lut = (uint4 *)smem_plain;                                   // 16 one-hot entries
    uint32_t *stage = (uint32_t *)(smem_plain + 256);                   // [4 waves][8][PS]
    uint4 *tile = (uint4 *)(smem_plain + 256 + 4 * 8 * PS * 4);         // [2][32][TROW]
    int *s_unit = (int *)(smem_plain + 256 + 4 * 8 * PS * 4 + 2 * 32 * TROW * 16);   // [2]
    int *s_q = s_unit + 4;                                              // [2][32] query of a tile's pairs
    int *s_f0 = s_q + 64;                                               // [2][32] ... and their row offset
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        w[threadIdx.x >> 2] = 1u << (8 * (threadIdx.x & 3));
        lut[threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const int n_units = j.unit_prefix[j.n_lists];
    int *counter = const_cast<int *>(j.unit_prefix) + TK_PLAIN_COUNTER_OFF(j.n_lists);
    const int r = lane & 31, h = lane >> 5;
    const int rr = r & 15;
    const uint32_t rot = (uint32_t)(8 * (rr & 3) + 4 * h + 28) & 31u;    // rotate right: nibble -> bits 4..7
    const uint32_t lut0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_plain;   // 256-byte aligned
    const bool loader = lane < 2 * P;
    const int ll = loader ? lane : 2 * P - 1;     // (lanes past the 2P groups clone the last loader)
    const int lch = ll / P, lp = ll - lch * P;
    uint32_t *st = stage + wave * 8 * PS;
    const uint32_t *rd = st + ((r >> 4) * 4 + (rr >> 2)) * PS;
    const int rows_m = 2 * P;                     // table rows (blocks) used per query

    // (a binary search through unit_prefix here was ten dependent trips to L2 per look-up, three
    // look-ups per unit: longer than the unit's arithmetic)
    auto locate = [&](int u, int &l, int &t) {
        const int4 d = ((const int4 *)j.unit_desc4)[u];     // (this form takes whole tiles: plain_k = TK_PLAIN_K_WHOLE)
        l = __builtin_amdgcn_readfirstlane(d.x);
        t = __builtin_amdgcn_readfirstlane(d.y);
    };
    // query and row offset of a unit's 32 pairs -> LDS (pairs past the last one: the last one)
    auto stage_pairs = [&](int b, int l, int t) {
        if (threadIdx.x < 32) {
            const int cnt = j.pair_off[l + 1] - j.pair_off[l] - 32 * t;      // >= 1
            const int rec = j.pair_off[l] + 32 * t + (threadIdx.x < cnt ? threadIdx.x : cnt - 1);
            s_q[b * 32 + threadIdx.x] = j.pair_q[rec];
            s_f0[b * 32 + threadIdx.x] = j.pair_f0[rec];
        }
    };
    // slice k (of TL) of the table rows of a unit's 32 pairs: one 16-byte load per thread,
    // coalesced per query row; kept in registers for the length of ONE chunk pair only.
    // Element i = threadIdx.x + 256 k of the tile is (pair i / rows_m, row i % rows_m): the pair
    // and row of slice 0 are computed once, the next slice's by stepping 256 elements on.
    const int sl_pr0 = threadIdx.x / rows_m, sl_m0 = threadIdx.x - sl_pr0 * rows_m;
    const int step_pr = 256 / rows_m, step_m = 256 - step_pr * rows_m;
    int sl_pr = sl_pr0, sl_m = sl_m0;
    auto slice_next = [&]() {
        sl_pr += step_pr;
        sl_m += step_m;
        if (sl_m >= rows_m) { sl_m -= rows_m; sl_pr++; }
    };
    auto fetch_slice = [&](int b) -> uint4 {       // unconditional (rows past the tile: row 31's)
        const int qs = s_q[b * 32 + (sl_pr < 32 ? sl_pr : 31)];
        return j.tables[(int64_t)qs * M + sl_m];
    };
    auto store_slice = [&](int b, const uint4 v) {
        if (sl_pr < 32) tile[(b * 32 + sl_pr) * TROW + sl_m] = v;
    };
    if (threadIdx.x == 0) s_unit[0] = atomicAdd(counter, 1);
    __syncthreads();
    int u = s_unit[0];
    int buf = 0;
    if (u < n_units) {
        int l0, t0;
        locate(u, l0, t0);
        stage_pairs(0, l0, t0);
        __syncthreads();
        for (int k = 0; k < TL; k++) {
            store_slice(0, fetch_slice(0));
            slice_next();
        }
    }
    while (u < n_units) {     // (workgroup-uniform: every wave reaches the barriers)
        if (threadIdx.x == 0) s_unit[buf ^ 1] = atomicAdd(counter, 1);
        __syncthreads();                   // tile[buf] is complete; the next unit is known
        int un = s_unit[buf ^ 1];
        int l, t;
        locate(u, l, t);
        // the next unit's tile travels to the other LDS buffer one slice per chunk pair (after
        // the last unit: this unit's once more, so that the loop below has one shape)
        {
            int nl = l, nt = t;
            if (un < n_units) locate(un, nl, nt);
            stage_pairs(buf ^ 1, nl, nt);
        }
        sl_pr = sl_pr0;
        sl_m = sl_m0;
        int ks = 0;
        const int64_t c0 = j.list_chunk_off[l];
        const int C = (int)(j.list_chunk_off[l + 1] - c0);
        const int CP = (C + 1) >> 1;
        const int qi = s_q[buf * 32 + r];          // (written a unit ago, or before the first barrier pair)
        const int f0 = s_f0[buf * 32 + r];
        int nvalid = j.pair_off[l + 1] - j.pair_off[l] - 32 * t;
        nvalid = nvalid < 32 ? nvalid : 32;
        const int rc = r < nvalid ? r : nvalid - 1;
        v4i B[PT];
#pragma unroll
        for (int p = 0; p < PT; p++)
            if (EXACT || p < P) B[p] = *(const v4i *)&tile[(buf * 32 + rc) * TROW + 2 * p + h];
        uint4 *drow = j.dist + (int64_t)qi * j.cap + f0;
        uint8_t *mrow = j.mins + (int64_t)qi * j.min_stride + f0;
        auto fetch = [&](int cp) -> uint4 {        // unconditional (chunks past the list: its last)
            int c = 2 * cp + lch;
            c = c < C ? c : C - 1;
            const int64_t gc = c0 + c;
            return j.codes[((gc >> 3) * (int64_t)(M >> 1) + lp) * 8 + (gc & 7)];
        };
        __syncthreads();                   // the next unit's pairs are staged
        // Two code groups in flight in two NAMED registers, iterations in pairs, the first pair peeled:
        // loads and stores retire through one in-order counter, and at a loop header the compiler
        // merges the counter's state on entry (the two prefetched groups are the youngest operations)
        // with the back edge's — a rotation by moves (g0 = g1; g1 = g2) and an un-peeled loop made
        // every iteration wait for the store it had issued a few instructions earlier (vmcnt(1)).
        auto stage_codes = [&](const uint4 g) {
            if (loader) {
                st[(lch * 4 + 0) * PS + lp] = g.x;
                st[(lch * 4 + 1) * PS + lp] = g.y;
                st[(lch * 4 + 2) * PS + lp] = g.z;
                st[(lch * 4 + 3) * PS + lp] = g.w;
            }
        };
        auto chunk_pair = [&](int cp) {
            const uint4 tslice = fetch_slice(buf ^ 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (the staging region is rewritten one iteration later, after this iteration's reads:
            // LDS operations of one wave complete in issue order)
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (EXACT) {
                uint32_t x[PS];
#pragma unroll
                for (int k = 0; k < PS / 4; k++) {
                    const uint4 v = *(const uint4 *)(rd + 4 * k);
                    x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
                }
                v4i A[PT];
#if defined(TK_PLAIN_EXPERIMENT) && TK_PLAIN_EXPERIMENT == 1
                // (scripts/micro only: WRONG results — the one-hot operands without their LDS reads,
                //  to see what those reads cost)
#pragma unroll
                for (int p = 0; p < PT; p++) {
                    const int t = (int)(__builtin_amdgcn_alignbit(x[p], x[p], rot) & 0xf0u);
                    A[p] = v4i{t, t ^ 1, t ^ 2, t ^ 3};
                }
#else
#pragma unroll
                for (int p = 0; p < PT; p++)
                    A[p] = one_hot(lut0, x[p], rot);
#endif
#pragma unroll
                for (int p = 0; p < PT; p++)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], B[p], acc, 0, 0, 0);
                // schedule: the x reads, DEPTH one-hot reads, then one MFMA per further one-hot read
                constexpr int DEPTH = 6 < PT ? 6 : PT;
                __builtin_amdgcn_sched_group_barrier(0x100, PS / 4 + DEPTH, 0);
#pragma unroll
                for (int p = 0; p < PT - DEPTH; p++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, DEPTH, 0);
            } else {
#pragma unroll
                for (int k = 0; k < PS / 4; k++) {
                    const uint4 v = *(const uint4 *)(rd + 4 * k);
                    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int p = 4 * k + i;
                        if (p < PT && p < P) {
                            const v4i A = one_hot(lut0, x[i], rot);
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[p], acc, 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int o[16];
#pragma unroll
            for (int i = 0; i < 16; i++) o[i] = clamp8(acc[i]);
            uint32_t X = pack4(o[0], o[1], o[2], o[3]), Y = pack4(o[4], o[5], o[6], o[7]);
            uint32_t Z = pack4(o[8], o[9], o[10], o[11]), W = pack4(o[12], o[13], o[14], o[15]);
            uint32_t mA = (uint32_t)min(min(min(o[0], o[1]), min(o[2], o[3])), min(min(o[4], o[5]), min(o[6], o[7])));
            uint32_t mB = (uint32_t)min(min(min(o[8], o[9]), min(o[10], o[11])), min(min(o[12], o[13]), min(o[14], o[15])));
            swap_halves(X, Z);
            swap_halves(Y, W);
            swap_halves(mA, mB);
            const int mn = min((int)mA, (int)mB);
            int cc = 2 * cp + h;
            cc = cc < C ? cc : C - 1;      // odd list: the second half holds the first chunk again
#if defined(TK_PLAIN_EXPERIMENT) && TK_PLAIN_EXPERIMENT == 2
            if (mn == 12345) {      // (scripts/micro only: WRONG results — no stores)
                drow[cc] = make_uint4(X, Z, Y, W);
                mrow[cc] = (uint8_t)mn;
            }
#else
            drow[cc] = make_uint4(X, Z, Y, W);
            mrow[cc] = (uint8_t)mn;
#endif
            if (ks < TL) {
                store_slice(buf ^ 1, tslice);
                slice_next();
                ks++;
            }
        };
        int cp = wave;
        uint4 ga = fetch(cp), gb = fetch(cp + 4);
#define TK_PAIR(cp_)                                  \
        {                                             \
            stage_codes(ga);                          \
            ga = fetch((cp_) + 8);                    \
            chunk_pair(cp_);                          \
            stage_codes(gb);                          \
            gb = fetch((cp_) + 12);                   \
            chunk_pair((cp_) + 4);                    \
        }
        if (cp + 4 < CP) {
            TK_PAIR(cp)
            cp += 8;
#pragma nounroll
            while (cp + 4 < CP) {
                TK_PAIR(cp)
                cp += 8;
            }
        }
#undef TK_PAIR
        if (cp < CP) {
            stage_codes(ga);
            chunk_pair(cp);
        }
        for (; ks < TL; ks++) {
            store_slice(buf ^ 1, fetch_slice(buf ^ 1));
            slice_next();
        }
        u = un;
        buf ^= 1;
    }
    TK_CLOCK_END();
}

// ---------------------------------------------------------------------------
// The same kernel with the table operand read from LDS for every MFMA instead of held in
// registers for the length of a unit: 104 fewer registers per lane (M = 52), so four waves per
// SIMD instead of two — the counters of the register form (scripts/r03_pmc_micro.sh) say 42 % of
// its wave cycles wait, and taking ALL of its one-hot LDS reads away (scripts/micro, wrong results)
// did not make it faster: it is bound by latency it has too few waves to cover, not by the LDS
// pipe.  One table tile per workgroup (31 KB with staging: four workgroups per CU), loaded between
// two barriers at the start of a unit — the other workgroups of the CU cover that.  Exact shapes
// only (P == PT).  A/B: TINYKNN_PLAIN_FORM (0 = registers, 1 = this).
template <int PT>
struct PlainShapeL {
    static constexpr int PS = (PT + 3) & ~3;
    static constexpr int TROW = 2 * PT + 1;
    static constexpr int TL = (32 * 2 * PT + 255) / 256;
    static constexpr size_t lds = 256 + (size_t)4 * 8 * PS * 4 + (size_t)32 * TROW * 16 + 16 + 2 * 32 * 4;
};

template <int PT, int WPS>
__global__ __launch_bounds__(256, WPS) void scan_plain_lds_kernel(TkScanJob j, int M)
{
    constexpr int P = PT;
    using SH = PlainShapeL<PT>;
    constexpr int PS = SH::PS, TROW = SH::TROW, TL = SH::TL;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem_plain[];
    uint4 *lut = (uint4 *)smem_plain;                                   // 16 one-hot entries
    uint32_t *stage = (uint32_t *)(smem_plain + 256);                   // [4 waves][8][PS]
    uint4 *tile = (uint4 *)(smem_plain + 256 + 4 * 8 * PS * 4);         // [32][TROW]
    int *s_unit = (int *)(smem_plain + 256 + 4 * 8 * PS * 4 + 32 * TROW * 16);       // [1] (+ pad)
    int *s_q = s_unit + 4;                                              // [32]
    int *s_f0 = s_q + 32;                                               // [32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        w[threadIdx.x >> 2] = 1u << (8 * (threadIdx.x & 3));
        lut[threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const int n_units = j.unit_prefix[j.n_lists];
    int *counter = const_cast<int *>(j.unit_prefix) + TK_PLAIN_COUNTER_OFF(j.n_lists);
    const int r = lane & 31, h = lane >> 5;
    const int rr = r & 15;
    const uint32_t rot = (uint32_t)(8 * (rr & 3) + 4 * h + 28) & 31u;
    const uint32_t lut0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_plain;   // 256-byte aligned
    const bool loader = lane < 2 * P;
    const int ll = loader ? lane : 2 * P - 1;
    const int lch = ll / P, lp = ll - lch * P;
    uint32_t *st = stage + wave * 8 * PS;
    const uint32_t *rd = st + ((r >> 4) * 4 + (rr >> 2)) * PS;
    const int rows_m = 2 * P;
    const int sl_pr0 = threadIdx.x / rows_m, sl_m0 = threadIdx.x - sl_pr0 * rows_m;
    const int step_pr = 256 / rows_m, step_m = 256 - step_pr * rows_m;
    for (;;) {
        if (threadIdx.x == 0) s_unit[0] = atomicAdd(counter, 1);
        __syncthreads();                   // every wave is done with the previous unit's tile
        const int u = s_unit[0];
        if (u >= n_units) break;           // (workgroup-uniform)
        const int4 d = ((const int4 *)j.unit_desc4)[u];
        const int l = __builtin_amdgcn_readfirstlane(d.x), t = __builtin_amdgcn_readfirstlane(d.y);
        int nvalid = j.pair_off[l + 1] - j.pair_off[l] - 32 * t;      // >= 1
        nvalid = nvalid < 32 ? nvalid : 32;
        if (threadIdx.x < 32) {            // pairs past the tile's last one: the last one
            const int rec = j.pair_off[l] + 32 * t + (threadIdx.x < nvalid ? (int)threadIdx.x : nvalid - 1);
            s_q[threadIdx.x] = j.pair_q[rec];
            s_f0[threadIdx.x] = j.pair_f0[rec];
        }
        __syncthreads();
        {   // the table rows of the tile's 32 pairs: element i = threadIdx.x + 256 k is (pair i / rows_m, row i % rows_m)
            int pr = sl_pr0, m = sl_m0;
            uint4 v[TL];
#pragma unroll
            for (int k = 0; k < TL; k++) {
                v[k] = j.tables[(int64_t)s_q[pr < 32 ? pr : 31] * M + m];
                pr += step_pr;
                m += step_m;
                if (m >= rows_m) { m -= rows_m; pr++; }
            }
            pr = sl_pr0;
            m = sl_m0;
#pragma unroll
            for (int k = 0; k < TL; k++) {
                if (pr < 32) tile[pr * TROW + m] = v[k];
                pr += step_pr;
                m += step_m;
                if (m >= rows_m) { m -= rows_m; pr++; }
            }
        }
        const int64_t c0 = j.list_chunk_off[l];
        const int C = (int)(j.list_chunk_off[l + 1] - c0);
        const int CP = (C + 1) >> 1;
        const int qi = s_q[r];
        const int f0 = s_f0[r];
        const int rc = r < nvalid ? r : nvalid - 1;
        const v4i *brow = (const v4i *)&tile[rc * TROW + h];           // block 2p + h: brow[2 * p]
        uint4 *drow = j.dist + (int64_t)qi * j.cap + f0;
        uint8_t *mrow = j.mins + (int64_t)qi * j.min_stride + f0;
        auto fetch = [&](int cp) -> uint4 {        // unconditional (chunks past the list: its last)
            int c = 2 * cp + lch;
            c = c < C ? c : C - 1;
            const int64_t gc = c0 + c;
            return j.codes[((gc >> 3) * (int64_t)(M >> 1) + lp) * 8 + (gc & 7)];
        };
        __syncthreads();                   // the tile is complete
        int cp = wave;
        uint4 g0 = fetch(cp), g1 = fetch(cp + 4);
        for (; cp < CP; cp += 4) {
            const uint4 g2 = fetch(cp + 8);
            if (loader) {
                st[(lch * 4 + 0) * PS + lp] = g0.x;
                st[(lch * 4 + 1) * PS + lp] = g0.y;
                st[(lch * 4 + 2) * PS + lp] = g0.z;
                st[(lch * 4 + 3) * PS + lp] = g0.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t x[PS];
#pragma unroll
            for (int k = 0; k < PS / 4; k++) {
                const uint4 v = *(const uint4 *)(rd + 4 * k);
                x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
            }
            v4i A[PT], B[PT];
#pragma unroll
            for (int p = 0; p < PT; p++) {
                A[p] = one_hot(lut0, x[p], rot);
                B[p] = brow[2 * p];
            }
#pragma unroll
            for (int p = 0; p < PT; p++)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], B[p], acc, 0, 0, 0);
            // schedule: the x reads, DEPTH operand pairs, then one MFMA per further pair
            constexpr int DEPTH = 3 < PT ? 3 : PT;
            __builtin_amdgcn_sched_group_barrier(0x100, PS / 4 + 2 * DEPTH, 0);
#pragma unroll
            for (int p = 0; p < PT - DEPTH; p++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, DEPTH, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int o[16];
#pragma unroll
            for (int i = 0; i < 16; i++) o[i] = clamp8(acc[i]);
            uint32_t X = pack4(o[0], o[1], o[2], o[3]), Y = pack4(o[4], o[5], o[6], o[7]);
            uint32_t Z = pack4(o[8], o[9], o[10], o[11]), W = pack4(o[12], o[13], o[14], o[15]);
            uint32_t mA = (uint32_t)min(min(min(o[0], o[1]), min(o[2], o[3])), min(min(o[4], o[5]), min(o[6], o[7])));
            uint32_t mB = (uint32_t)min(min(min(o[8], o[9]), min(o[10], o[11])), min(min(o[12], o[13]), min(o[14], o[15])));
            swap_halves(X, Z);
            swap_halves(Y, W);
            swap_halves(mA, mB);
            const int mn = min((int)mA, (int)mB);
            int cc = 2 * cp + h;
            cc = cc < C ? cc : C - 1;
            drow[cc] = make_uint4(X, Z, Y, W);
            mrow[cc] = (uint8_t)mn;
            g0 = g1;
            g1 = g2;
        }
    }
}

// ---------------------------------------------------------------------------
// Round 4: ONE WAVE PER UNIT.  A unit is (list, tile of 32 of its pairs, range of chunk pairs):
// int4 descriptors written by plain_desc_fill (adc_scan.hip).  What changed against the
// workgroup-per-tile form above, and why:
//   * no workgroup barrier and no table tile in LDS: a wave loads the 26 table rows of its lane's
//     query straight into the B registers (the sub-units of a tile are consecutive units, so the
//     four waves of a workgroup read the same rows: L1 hits) — 3.8 KB of LDS per workgroup
//     instead of 58 KB, and a unit's 12 chunk pairs amortise that load, where the four waves of
//     the old form met at two barriers every ~9 chunk pairs;
//   * the chunk-pair loop is PEELED twice by hand.  Loads and stores retire through one in-order
//     counter (vmcnt); at a loop header the compiler merges the counter state of the loop's entry
//     with that of its back edge, and on entry the two prefetched code groups are the YOUNGEST
//     operations in flight: the merged wait became vmcnt(1) in every iteration, i.e. each
//     iteration waited for the 16-byte store it had issued a few instructions earlier (a full
//     round trip to L2: the 42 % of wave cycles in s_waitcnt of round 3).  Entering the loop from
//     two peeled copies of its body makes entry and back edge look alike, and the waits count
//     what they should (vmcnt(5): the loads of two iterations ago).
//   * units are handed out like the exact kernel's blocks (tickets.h): one statically per wave,
//     the rest from eight counters.
// Same outputs, byte for byte, as scan_plain_kernel (tests/test_plain_scan_gpu.py,
// scripts/micro/mfma_scan.hip).

template <int PT>
struct PlainWaveShape {
    static constexpr int PS = (PT + 3) & ~3;
    // one-hot table | code staging [4 waves][8][PS] dwords | per wave: output tile 32 x 9 x 16 B,
    // minima 32 x 8 B, row offsets of the tile's pairs 2 x 32 x 8 B
    static constexpr size_t wave_out = 32 * 9 * 16 + 32 * 8 + 2 * 32 * 8;
    static constexpr size_t lds = 256 + (size_t)4 * 8 * PS * 4 + 4 * wave_out;
};

// FLUSH: a lane's 16-byte block of (query, chunk) and its minimum go to an LDS tile [query][chunk
// slot] and leave every fourth chunk pair as four 16-byte stores whose eight neighbouring lanes
// write the eight chunks of ONE query (a whole 128-byte line when the row is aligned) — instead of
// one 16-byte store and one byte store per chunk pair that touch 32 rows each.  The texture path
// (TA/TD 67 % / 77 % busy under the per-row form, rocprofv3 on scripts/micro/mfma_scan) is what
// this kernel saturates first.
template <int PT, bool EXACT, bool FLUSH>
__global__ __launch_bounds__(256, 2) void scan_plain_wave_kernel(TkScanJob j, int P, int M)
{
    if (EXACT) P = PT;          // (P < PT: the block pairs past P get zero table rows)
    TK_CLOCK_BEGIN();
    using SH = PlainWaveShape<PT>;
    constexpr int PS = SH::PS;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem_plain[];
    uint4 *lut = (uint4 *)smem_plain;                                   // 16 one-hot entries
    uint32_t *stage = (uint32_t *)(smem_plain + 256);                   // [4 waves][8][PS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *wout = smem_plain + 256 + 4 * 8 * PS * 4 + wave * SH::wave_out;
    uint4 *otile = (uint4 *)wout;                                       // [32 pairs][9 slots]
    uint8_t *omin = wout + 32 * 9 * 16;                                 // [32 pairs][8 slots]
    long long *orow = (long long *)(wout + 32 * 9 * 16 + 32 * 8);       // [32] uint4 offset of a pair's distance row
    long long *omrow = orow + 32;                                       // [32] byte offset of its minima row
    if (threadIdx.x < 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        w[threadIdx.x >> 2] = 1u << (8 * (threadIdx.x & 3));
        lut[threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const int n_units = j.unit_prefix[j.n_lists];
    int *ticket = const_cast<int *>(j.unit_prefix) + TK_TICKET_OFF(j.n_lists);
    const int r = lane & 31, h = lane >> 5;
    const int rr = r & 15;
    const uint32_t rot = (uint32_t)(8 * (rr & 3) + 4 * h + 28) & 31u;    // rotate right: nibble -> bits 4..7
    const uint32_t lut0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_plain;   // 256-byte aligned
    const bool loader = lane < 2 * P;
    const int ll = loader ? lane : 2 * P - 1;     // (lanes past the 2P groups clone the last loader)
    const int lch = ll / P, lp = ll - lch * P;
    uint32_t *st = stage + wave * 8 * PS;
    uint32_t *stw = st + (lch * 4) * PS + lp;
    const uint32_t *rd = st + ((r >> 4) * 4 + (rr >> 2)) * PS;

    ticketed_blocks(n_units, ticket, [&](int u) {
        const int4 d = ((const int4 *)j.unit_desc4)[u];
        const int l = __builtin_amdgcn_readfirstlane(d.x), t = __builtin_amdgcn_readfirstlane(d.y);
        const int cpa = __builtin_amdgcn_readfirstlane(d.z), cpb = __builtin_amdgcn_readfirstlane(d.w);
        const int po = j.pair_off[l];
        int nvalid = j.pair_off[l + 1] - po - 32 * t;                     // >= 1
        nvalid = nvalid < 32 ? nvalid : 32;
        const int rec = po + 32 * t + (r < nvalid ? r : nvalid - 1);      // (pairs past the last one: the last one)
        const int qi = j.pair_q[rec], f0 = j.pair_f0[rec];
        const int64_t c0 = j.list_chunk_off[l];
        const int C = (int)(j.list_chunk_off[l + 1] - c0);
        // tiled code layout (kernels.h) relative to the 8-chunk row the list starts in
        const uint4 *cbase = j.codes + (c0 >> 3) * (int64_t)(8 * P) + lp * 8;
        const int cin = (int)(c0 & 7);
        auto fetch = [&](int cp) -> uint4 {        // unconditional (chunks past the list: its last)
            int c = 2 * cp + lch;
            c = (c < C ? c : C - 1) + cin;
            return cbase[(c >> 3) * (8 * P) + (c & 7)];
        };
        uint4 ga = fetch(cpa), gb = fetch(cpa + 1);
        v4i B[PT];
        {
            const v4i *brow = (const v4i *)(j.tables + (int64_t)qi * M + h);
#pragma unroll
            for (int p = 0; p < PT; p++) B[p] = (EXACT || p < P) ? brow[2 * p] : v4i{0, 0, 0, 0};
        }
        uint4 *drow = j.dist + (int64_t)qi * j.cap + f0;
        uint8_t *mrow = j.mins + (int64_t)qi * j.min_stride + f0;
        if (FLUSH && h == 0) {
            orow[r] = (long long)qi * j.cap + f0;
            omrow[r] = (long long)qi * j.min_stride + f0;
        }
        const int c_end = 2 * cpb < C ? 2 * cpb : C;      // chunks this unit owns: [2 cpa, c_end)

        auto stage_codes = [&](const uint4 g) {
            if (loader) {
                stw[0 * PS] = g.x;
                stw[1 * PS] = g.y;
                stw[2 * PS] = g.z;
                stw[3 * PS] = g.w;
            }
        };
        // chunks [8 g8, 8 g8 + 8) of the tile's pairs: LDS tile -> rows
        auto flush = [&](int g8) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int s = lane & 7, chunk = 8 * g8 + s;
            const bool cok = chunk >= 2 * cpa && chunk < c_end;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int qq = (lane >> 3) + 8 * k;
                const uint4 v = otile[qq * 9 + s];
                const uint8_t m = omin[qq * 8 + s];
                const long long ro = orow[qq], mo = omrow[qq];
                if (cok && qq < nvalid) {
                    j.dist[ro + chunk] = v;
                    j.mins[mo + chunk] = m;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        auto chunk_pair = [&](int cp) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (the staging region is rewritten one iteration later, after this iteration's reads:
            // LDS operations of one wave complete in issue order)
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t x[PS];
#pragma unroll
            for (int k = 0; k < PS / 4; k++) {
                const uint4 v = *(const uint4 *)(rd + 4 * k);
                x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
            }
            v4i A[PT];
#pragma unroll
            for (int p = 0; p < PT; p++) A[p] = one_hot(lut0, x[p], rot);
#pragma unroll
            for (int p = 0; p < PT; p++)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], B[p], acc, 0, 0, 0);
            // schedule: the x reads, DEPTH one-hot reads, then one MFMA per further one-hot read
            constexpr int DEPTH = 6 < PT ? 6 : PT;
            __builtin_amdgcn_sched_group_barrier(0x100, PS / 4 + DEPTH, 0);
#pragma unroll
            for (int p = 0; p < PT - DEPTH; p++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, DEPTH, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int o[16];
#pragma unroll
            for (int i = 0; i < 16; i++) o[i] = clamp8(acc[i]);
            uint32_t X = pack4(o[0], o[1], o[2], o[3]), Y = pack4(o[4], o[5], o[6], o[7]);
            uint32_t Z = pack4(o[8], o[9], o[10], o[11]), W = pack4(o[12], o[13], o[14], o[15]);
            uint32_t mA = (uint32_t)min(min(min(o[0], o[1]), min(o[2], o[3])), min(min(o[4], o[5]), min(o[6], o[7])));
            uint32_t mB = (uint32_t)min(min(min(o[8], o[9]), min(o[10], o[11])), min(min(o[12], o[13]), min(o[14], o[15])));
            swap_halves(X, Z);
            swap_halves(Y, W);
            swap_halves(mA, mB);
            const int mn = min((int)mA, (int)mB);
            if (FLUSH) {
                const int slot = (2 * cp + h) & 7;
                otile[r * 9 + slot] = make_uint4(X, Z, Y, W);
                omin[r * 8 + slot] = (uint8_t)mn;
                if ((cp & 3) == 3) flush(cp >> 2);
            } else {
                int cc = 2 * cp + h;
                cc = cc < C ? cc : C - 1;      // odd list: the second half holds the first chunk again
                drow[cc] = make_uint4(X, Z, Y, W);
                mrow[cc] = (uint8_t)mn;
            }
        };
        // Two code groups in flight, in two named registers (no rotation by moves: a move of the
        // younger group would wait for it); iterations go in pairs, the first pair peeled (see the
        // header comment), an odd last one on its own.
#define TK_PAIR(cp_)                                  \
        {                                             \
            stage_codes(ga);                          \
            ga = fetch((cp_) + 2);                    \
            chunk_pair(cp_);                          \
            stage_codes(gb);                          \
            gb = fetch((cp_) + 3);                    \
            chunk_pair((cp_) + 1);                    \
        }
        int cp = cpa;
        if (cp + 1 < cpb) {
            TK_PAIR(cp)
            cp += 2;
#pragma nounroll
            while (cp + 1 < cpb) {
                TK_PAIR(cp)
                cp += 2;
            }
        }
#undef TK_PAIR
        if (cp < cpb) {
            stage_codes(ga);
            chunk_pair(cp);
            cp++;
        }
        if (FLUSH && (cp & 3) != 0) flush((cp - 1) >> 2);      // (a unit that ends inside a group of four)
    });
    TK_CLOCK_END();
}

int tk_plain_fits(int M) { return M >= 2 && M % 2 == 0 && M / 2 <= 26; }

static int g_plain_flush = 1;      // wave form: outputs leave through an LDS tile, a whole line per query (A/B: tk_plain_set_flush)
void tk_plain_set_flush(int on) { g_plain_flush = on; }
static int g_plain_form = -1;       // -1: not read yet (environment, default 3)
void tk_plain_set_form(int form) { g_plain_form = form; }
// 3 (default): one wave per unit; 0 / 1 / 2: the round-3 workgroup-per-tile forms (whole tiles: the
// descriptors must then be built with plain_k = TK_PLAIN_K_WHOLE — tk_plain_k says which)
static int plain_form()
{
    if (g_plain_form < 0) g_plain_form = getenv("TINYKNN_PLAIN_FORM") ? atoi(getenv("TINYKNN_PLAIN_FORM")) : 3;
    return g_plain_form;
}
int tk_plain_wave_form(void) { return plain_form() == 3; }

// j.unit_prefix: tiles of 32 pairs before each list (n_lists + 1), then the work counter (zeroed
// by the kernel that wrote the table); P block pairs are summed (AVX order: an odd trailing pair
// is not read by the reference's kernel either, _fast_pq_256.pyx:135-149)
int tk_launch_scan_plain(const TkScanJob &j, int M, int order, int n_blocks, hipStream_t s)
{
    if (!j.unit_prefix) return 0;
    int P = M / 2;
    if (order == TK_ORDER_AVX) P &= ~1;
    if (P < 1 || P > 26) return -1;
    if (tk_plain_wave_form()) { // one wave per unit (descriptors with chunk-pair ranges)
#define TK_LAUNCH_W(PT_, EX_)                                                                       \
    do {                                                                                            \
        if (g_plain_flush) {                                                                        \
            static bool attr_ = false;                                                              \
            if (!attr_) {                                                                           \
                if (hipFuncSetAttribute((const void *)scan_plain_wave_kernel<PT_, EX_, true>,       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,                 \
                                        (int)PlainWaveShape<PT_>::lds) != hipSuccess)               \
                    return -1;                                                                      \
                attr_ = true;                                                                       \
            }                                                                                       \
            hipLaunchKernelGGL((scan_plain_wave_kernel<PT_, EX_, true>), dim3(n_blocks), dim3(256), \
                               PlainWaveShape<PT_>::lds, s, j, P, M);                               \
        } else                                                                                      \
            hipLaunchKernelGGL((scan_plain_wave_kernel<PT_, EX_, false>), dim3(n_blocks), dim3(256), \
                               PlainWaveShape<PT_>::lds, s, j, P, M);                               \
    } while (0)
        if (P == 26) TK_LAUNCH_W(26, true);
        else if (P == 16) TK_LAUNCH_W(16, true);
        else if (P <= 8) TK_LAUNCH_W(8, false);
        else if (P <= 16) TK_LAUNCH_W(16, false);
        else TK_LAUNCH_W(26, false);
#undef TK_LAUNCH_W
        return 0;
    }
#define TK_LAUNCH(PT_, EX_)                                                                         \
    do {                                                                                            \
        static bool attr_ = false;                                                                  \
        if (!attr_) {                                                                               \
            if (hipFuncSetAttribute((const void *)scan_plain_kernel<PT_, EX_>,                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,                     \
                                    (int)PlainShape<PT_>::lds) != hipSuccess)                       \
                return -1;                                                                          \
            attr_ = true;                                                                           \
        }                                                                                           \
        hipLaunchKernelGGL((scan_plain_kernel<PT_, EX_>), dim3(n_blocks), dim3(256),                \
                           PlainShape<PT_>::lds, s, j, P, M);                                       \
    } while (0)
    static int lean = -1;      // A/B: TINYKNN_PLAIN_LEAN=1: the guarded form (fewer registers, shallow prefetch)
    if (lean < 0) lean = getenv("TINYKNN_PLAIN_LEAN") ? atoi(getenv("TINYKNN_PLAIN_LEAN")) : 0;
    // A/B: TINYKNN_PLAIN_FORM / tk_plain_set_form: 1 = table operand from LDS per MFMA, four waves per
    // SIMD; 2 = the same at three
    const int form = plain_form();
#define TK_LAUNCH_L(PT_, WPS_)                                                                      \
    do {                                                                                            \
        static bool attr_ = false;                                                                  \
        if (!attr_) {                                                                               \
            if (hipFuncSetAttribute((const void *)scan_plain_lds_kernel<PT_, WPS_>,                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,                     \
                                    (int)PlainShapeL<PT_>::lds) != hipSuccess)                      \
                return -1;                                                                          \
            attr_ = true;                                                                           \
        }                                                                                           \
        hipLaunchKernelGGL((scan_plain_lds_kernel<PT_, WPS_>), dim3(n_blocks * (form == 2 ? 3 : 4) / 2), dim3(256), \
                           PlainShapeL<PT_>::lds, s, j, M);                                         \
    } while (0)
    if (form == 1 && P == 26) { TK_LAUNCH_L(26, 4); return 0; }
    if (form == 2 && P == 26) { TK_LAUNCH_L(26, 3); return 0; }
    if (form == 1 && P == 16) { TK_LAUNCH_L(16, 4); return 0; }
    if (P == 26 && !lean) TK_LAUNCH(26, true);
    else if (P == 16 && !lean) TK_LAUNCH(16, true);
    else if (P <= 8) TK_LAUNCH(8, false);
    else if (P <= 16) TK_LAUNCH(16, false);
    else TK_LAUNCH(26, false);
#undef TK_LAUNCH
    return 0;
}
