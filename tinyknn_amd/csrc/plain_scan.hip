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
// Work: ONE WAVE PER UNIT.  A unit is (list, tile of 32 of its pairs, range of at most K chunk
// pairs): int4 descriptors written on the device (adc_scan.hip: plain_desc_fill).  A wave loads the
// 26 table rows of its lane's query straight into the B registers, walks the unit's chunk pairs with
// two code groups in flight (ONE 16-byte load per lane and chunk pair, re-sliced through a per-wave
// LDS region), and writes its outputs through an LDS tile as whole lines per query.  Units are
// handed out like the exact kernel's blocks (tickets.h).
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
#ifdef TK_PLAIN_STAMPS      // scripts/micro only: where a wave's cycles go (s_memtime between the phases)
__device__ unsigned long long tk_plain_stamps[8];
#define TK_STAMP(i)                                                   \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        stamp_acc[i] += now_ - stamp_last;                            \
        stamp_last = now_;                                            \
    } while (0)
#else
#define TK_STAMP(i)
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


// force (INT_MAX = none): a cap on every query's limit — the tests' way to provoke the re-scan path
// (tk_index_set_option TK_OPT_PLAIN_LIMIT)
void tk_launch_table_limits(const uint4 *tables, int M, int order, int64_t nq, int *qlim, hipStream_t s, int force)
{
    if (nq == 0) return;
    const int avx = order == TK_ORDER_AVX;
    const int M_used = avx ? (M & ~3) : M;       // the AVX kernels read block pairs two at a time
    hipLaunchKernelGGL(table_limits_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, tables,
                       M_used, M, avx, nq, force, qlim);
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

template <int PT>
struct PlainWaveShape {
    static constexpr int PS = (PT + 3) & ~3;
    // one-hot table | code staging [4 waves][8][PS] dwords | per wave: output tile 32 x 9 x 16 B,
    // minima 32 x 8 B, row offsets of the tile's pairs 2 x 32 x 8 B
    static constexpr size_t wave_out = 32 * 9 * 16 + 32 * 8 + 2 * 32 * 8;
    static constexpr size_t lds = 256 + (size_t)4 * 8 * PS * 4 + 4 * wave_out;
};

// Outputs: a lane's 16-byte block of (query, chunk) and its minimum go to an LDS tile [query][chunk
// slot] and leave every fourth chunk pair as four 16-byte stores whose eight neighbouring lanes
// write the eight chunks of ONE query (a whole 128-byte line when the row is aligned) — instead of
// one 16-byte store and one byte store per chunk pair that touch 32 rows each.  The texture path
// (TA/TD 67 % / 77 % busy under the per-row form, rocprofv3 on scripts/micro/mfma_scan) is what
// this kernel saturates first.
template <int PT, bool EXACT>
__global__ __launch_bounds__(256, 2) void scan_plain_wave_kernel(TkScanJob j, int P, int M)
{
    if (EXACT) P = PT;          // (P < PT: the block pairs past P get zero table rows)
    TK_CLOCK_BEGIN();
#ifdef TK_PLAIN_STAMPS
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_readcyclecounter(), stamp_it = 0;
    const unsigned long long stamp_k0 = stamp_last;
#endif
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
        TK_STAMP(5);        // between units: ticket
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
        if (h == 0) {
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
        TK_STAMP(0);        // unit prologue: descriptor, pair records, table rows, first code groups
        auto chunk_pair = [&](int cp) {
            TK_STAMP(1);    // staging stores + next fetch issued
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
#ifdef TK_PLAIN_STAMPS
            asm volatile("s_nop 0" ::"v"(o[15]));
            TK_STAMP(2);    // LDS reads + MFMA chain + clamp
            stamp_it++;
#endif
            uint32_t X = pack4(o[0], o[1], o[2], o[3]), Y = pack4(o[4], o[5], o[6], o[7]);
            uint32_t Z = pack4(o[8], o[9], o[10], o[11]), W = pack4(o[12], o[13], o[14], o[15]);
            uint32_t mA = (uint32_t)min(min(min(o[0], o[1]), min(o[2], o[3])), min(min(o[4], o[5]), min(o[6], o[7])));
            uint32_t mB = (uint32_t)min(min(min(o[8], o[9]), min(o[10], o[11])), min(min(o[12], o[13]), min(o[14], o[15])));
            swap_halves(X, Z);
            swap_halves(Y, W);
            swap_halves(mA, mB);
            const int mn = min((int)mA, (int)mB);
            // (odd list: the second half of its last chunk pair holds the last chunk again; its slot lies
            //  past the list and is not flushed)
            const int slot = (2 * cp + h) & 7;
            otile[r * 9 + slot] = make_uint4(X, Z, Y, W);
            omin[r * 8 + slot] = (uint8_t)mn;
            TK_STAMP(3);    // pack, swaps, LDS tile
            if ((cp & 3) == 3) flush(cp >> 2);
            TK_STAMP(4);    // flush (every fourth)
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
        if ((cp & 3) != 0) flush((cp - 1) >> 2);      // (a unit that ends inside a group of four)
        TK_STAMP(4);
    });
    TK_CLOCK_END();
#ifdef TK_PLAIN_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 6; i++) atomicAdd(&tk_plain_stamps[i], stamp_acc[i]);
        atomicAdd(&tk_plain_stamps[6], stamp_it);
        atomicAdd(&tk_plain_stamps[7], (unsigned long long)__builtin_readcyclecounter() - stamp_k0);
    }
#endif
}

int tk_plain_fits(int M) { return M >= 2 && M % 2 == 0 && M / 2 <= 26; }

// j.unit_prefix: units before each list (n_lists + 1), then the work counters of tickets.h (zeroed
// by the kernel that wrote the table); P block pairs are summed (AVX order: an odd trailing pair
// is not read by the reference's kernel either, _fast_pq_256.pyx:135-149)
int tk_launch_scan_plain(const TkScanJob &j, int M, int order, int n_blocks, hipStream_t s)
{
    if (!j.unit_prefix) return 0;
    int P = M / 2;
    if (order == TK_ORDER_AVX) P &= ~1;
    if (P < 1 || P > 26) return -1;
#define TK_LAUNCH_W(PT_, EX_)                                                                       \
    do {                                                                                            \
        static bool attr_ = false;                                                                  \
        if (!attr_) {                                                                               \
            if (hipFuncSetAttribute((const void *)scan_plain_wave_kernel<PT_, EX_>,           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,                     \
                                    (int)PlainWaveShape<PT_>::lds) != hipSuccess)                   \
                return -1;                                                                          \
            attr_ = true;                                                                           \
        }                                                                                           \
        hipLaunchKernelGGL((scan_plain_wave_kernel<PT_, EX_>), dim3(n_blocks), dim3(256),     \
                           PlainWaveShape<PT_>::lds, s, j, P, M);                                   \
    } while (0)
    if (P == 26) TK_LAUNCH_W(26, true);
    else if (P == 16) TK_LAUNCH_W(16, true);
    else if (P <= 8) TK_LAUNCH_W(8, false);
    else if (P <= 16) TK_LAUNCH_W(16, false);
    else TK_LAUNCH_W(26, false);
#undef TK_LAUNCH_W
    return 0;
}
