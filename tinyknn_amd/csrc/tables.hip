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
#include <stdlib.h>
#include "kernels.h"
#include "np_order.h"
#include "tickets.h"

// numpy's pairwise recursion over n elements (n > 128 splits at n2 = n/2 - (n/2)%8
// into [0,n2) and [n2,n)) depends only on n, so the host lays it out once per
// launch and passes it by value: the <=64 leaves in order, and the internal nodes
// as (dst, left, right) adds grouped by tree level so that a level runs in parallel.
struct PwProgram {
    unsigned short leaf_off[64], leaf_n[64];
    unsigned char op_dst[64], op_a[64], op_b[64];
    unsigned char level_start[10];   // ops of level v are [level_start[v-1], level_start[v])
    int nleaf, nlevels, root;
};

static int pw_build(PwProgram &P, int off, int n, int &next_internal, int *level_of,
                    unsigned char (*ops)[3], int *op_level, int &nops)
{
    if (n <= 128) {
        int id = P.nleaf++;
        P.leaf_off[id] = (unsigned short)off;
        P.leaf_n[id] = (unsigned short)n;
        level_of[id] = 0;
        return id;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int l = pw_build(P, off, n2, next_internal, level_of, ops, op_level, nops);
    int r = pw_build(P, off + n2, n - n2, next_internal, level_of, ops, op_level, nops);
    int id = next_internal++;
    level_of[id] = 1 + (level_of[l] > level_of[r] ? level_of[l] : level_of[r]);
    ops[nops][0] = (unsigned char)id;
    ops[nops][1] = (unsigned char)l;
    ops[nops][2] = (unsigned char)r;
    op_level[nops] = level_of[id];
    nops++;
    return id;
}

static PwProgram pw_program(int n)
{
    PwProgram P = {};
    int level_of[128];
    unsigned char ops[64][3];
    int op_level[64];
    int nops = 0;
    // leaves get ids 0..nleaf-1 in order, so count them first
    int nleaf = 0;
    {
        int st[64][2], sp = 0;
        st[sp][0] = 0; st[sp][1] = n; sp++;
        while (sp) {
            sp--;
            int m = st[sp][1];
            if (m <= 128) { nleaf++; continue; }
            int n2 = m / 2; n2 -= n2 % 8;
            st[sp][0] = 0; st[sp][1] = m - n2; sp++;
            st[sp][0] = 0; st[sp][1] = n2; sp++;
        }
    }
    int next_internal = nleaf;
    P.root = pw_build(P, 0, n, next_internal, level_of, ops, op_level, nops);
    int maxl = 0;
    for (int i = 0; i < nops; i++) maxl = op_level[i] > maxl ? op_level[i] : maxl;
    P.nlevels = maxl;
    int k = 0;
    for (int lv = 1; lv <= maxl; lv++) {
        for (int i = 0; i < nops; i++)
            if (op_level[i] == lv) {
                P.op_dst[k] = ops[i][0]; P.op_a[k] = ops[i][1]; P.op_b[k] = ops[i][2];
                k++;
            }
        P.level_start[lv] = (unsigned char)k;
    }
    P.level_start[0] = 0;
    return P;
}

// Everything a query needs lives in its own wave and its own LDS region, and LDS operations of
// one wave complete in issue order: a wave-level fence (for the compiler) is all the
// synchronisation the stages need — with __syncthreads() the four waves of a workgroup waited
// for each other seven times per query.
#define TK_WAVE_SYNC()                                              \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)

// One wave per query, blockDim.x / 64 queries per workgroup.  Per-wave LDS:
//   dists[16M] T | acc[64][8] T | node_val[128] T (leaf sums, then inner nodes)
template <typename T, bool SIGNED>
__global__ __launch_bounds__(256) void build_tables_kernel(
    const float *__restrict__ centers, int dq, int dpb, int f_order, const T *__restrict__ qs,
    double aux0, double aux1, uint8_t *__restrict__ tables, T *__restrict__ shift_out,
    double *__restrict__ scale_out, int64_t nq, int wave_lds, const PwProgram pw,
    const T *__restrict__ qs_b, int64_t n_a, const TkTablesExtra ex)
{
    // a wave = one query's chain of dependent LDS round trips; in the pipelined mode it shares its SIMD
    // with the scans of earlier batches and heads the stream the launches wait for: its instructions
    // first (same box: 0.406 -> 0.401 ms per 10 000 queries, profiles/r04/ab_tables_prio.txt; the
    // same on the rescoring kernels: no gain coarse, a loss final)
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const int M = dq / dpb;
    const int cnt = 16 * M;
    unsigned char *base = smem + (size_t)wave * wave_lds;
    T *dists = (T *)base;                 // numpy memory order
    T *acc = dists + cnt;                 // [leaf][8]
    T *node_val = acc + 64 * 8;
    const int64_t qraw = (int64_t)blockIdx.x * waves + wave;
    if (qraw >= nq) return;                     // (no workgroup barrier below)
    const int64_t qi = qraw;
    const T *q = (qs_b && qi >= n_a) ? qs_b + (qi - n_a) * dq : qs + qi * dq;   // second call of a pair
    const float rcpM = 1.0f / (float)M;
    if (SIGNED && dpb == 2 && M <= 256) {
        // FastPQ(2), the configuration every BASELINE index uses: the 13 rounds of a query were
        // 13 dependent trips to L2 (centre pair + query pair per entry); unrolled, four rounds'
        // loads are in flight.  Same operations: einsum_selfdot over the two differences.
#pragma unroll 4
        for (int e = lane; e < cnt; e += 64) {
            const int i = (int)(((float)e + 0.5f) * rcpM);
            const int m = e - i * M;
            const float2 c2 = *reinterpret_cast<const float2 *>(centers + (int64_t)i * dq + 2 * m);
            T d2[2];
            d2[0] = (T)c2.x - q[2 * m];
            d2[1] = (T)c2.y - q[2 * m + 1];
            dists[f_order ? (m * 16 + i) : e] = einsum_selfdot<T>(d2, 2);
        }
    } else
    for (int e = lane; e < cnt; e += 64) {
        // e / M without the integer division (exact for e < 2^16: |error| << 0.5 / M)
        const int i = M <= 256 ? (int)(((float)e + 0.5f) * rcpM) : e / M;
        const int m = e - i * M;
        T v;
        if (SIGNED) {
            const float *cp = centers + (int64_t)i * dq + m * dpb;
            const T *qp = q + m * dpb;
            v = einsum_selfdot_fn<T>([&](int k) { return (T)cp[k] - qp[k]; }, dpb);
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
    TK_WAVE_SYNC();
    T shift;
    if (SIGNED) {
        // numpy's pairwise mean, the 8 accumulators of every <=128-element leaf in
        // parallel (lane = leaf*8 + accumulator), then the exact combine tree level
        // by level
        const int nleaf = pw.nleaf;
        for (int t = lane; t < nleaf * 8; t += 64) {
            const int L = t >> 3, j = t & 7;
            const int off = pw.leaf_off[L], n = pw.leaf_n[L];
            T r = 0;
            if (n >= 8) {
                r = dists[off + j];
                for (int i = 8; i < n - (n % 8); i += 8) r += dists[off + i + j];
            }
            acc[L * 8 + j] = r;
        }
        TK_WAVE_SYNC();
        for (int L = lane; L < nleaf; L += 64) {
            const int off = pw.leaf_off[L], n = pw.leaf_n[L];
            T res;
            if (n < 8) {
                res = 0;
                for (int i = 0; i < n; i++) res += dists[off + i];
            } else {
                const T *r = acc + L * 8;
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (int i = n - (n % 8); i < n; i++) res += dists[off + i];
            }
            node_val[L] = res;
        }
        TK_WAVE_SYNC();
        for (int lv = 1; lv <= pw.nlevels; lv++) {
            const int t = pw.level_start[lv - 1] + lane;
            if (t < pw.level_start[lv])
                node_val[pw.op_dst[t]] = node_val[pw.op_a[t]] + node_val[pw.op_b[t]];
            TK_WAVE_SYNC();
        }
        const T mean = node_val[pw.root] / (T)cnt;   // _mean
        shift = mean * (T)0.6931471806;              // fast_pq.py:214
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
    TK_WAVE_SYNC();
    // in OUTPUT order (entry o = m * 16 + i), four bytes per lane: one coalesced dword store
    // instead of four byte stores 16 bytes apart
    uint32_t *out4 = reinterpret_cast<uint32_t *>(tables + qi * (int64_t)cnt);
    int n0 = 0, n1 = 0;         // ex.qlim: negative mass of the two saturating chains (plain_scan.hip's lemma)
    for (int o4 = lane; o4 < cnt / 4; o4 += 64) {
        uint32_t pack = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int o = 4 * o4 + b;
            const int m = o >> 4, i = o & 15;
            const double v = rint((double)dists[f_order ? o : (i * M + m)] * scale);
            pack |= (uint32_t)(uint8_t)(int32_t)v << (8 * b);  // :217-221
        }
        out4[o4] = pack;
        if (SIGNED && ex.qlim) {
            // the four lanes 4j .. 4j + 3 hold the four dwords of block m = o4 >> 2 (4 M dwords: whole groups
            // are active or idle together): its smallest entry, as the reference's signed table reads it
            int mn = min(min((int)(int8_t)pack, (int)(int8_t)(pack >> 8)), min((int)(int8_t)(pack >> 16), (int)(int8_t)(pack >> 24)));
            mn = min(mn, __shfl_xor(mn, 1, 64));
            mn = min(mn, __shfl_xor(mn, 2, 64));
            const int m = o4 >> 2;
            if ((o4 & 3) == 0 && m < ex.lim_m_used) {
                const int neg = mn < 0 ? -mn : 0;
                if (ex.lim_avx && ((m >> 1) & 1)) n1 += neg; else n0 += neg;
            }
        }
    }
    if (SIGNED && ex.qlim) {    // C of the lemma per query, or TK_PLAIN_NEVER (what table_limits_kernel computes)
        for (int o = 32; o > 0; o >>= 1) {
            n0 += __shfl_xor(n0, o, 64);
            n1 += __shfl_xor(n1, o, 64);
        }
        if (lane == 0) {
            int c = (n0 <= 128 && n1 <= 128) ? 127 - n0 - n1 : TK_PLAIN_NEVER;
            if (ex.lim_force != 0x7fffffff && c > ex.lim_force) c = ex.lim_force;
            ex.qlim[qi] = c;
        }
    }
    if (ex.c_pair_q) {
        // "every query scans the one list of coded centres" (pairs_identity_kernel): pair qi = query qi, the
        // records padded to groups of four, the header and the scan's work counters by the first wave
        const int64_t nrec = (ex.c_nq + TK_UNIT_Q - 1) / TK_UNIT_Q * TK_UNIT_Q;
        if (qi < ex.c_nq) {
            if (lane == 0) {
                ex.c_pair_q[qi] = (int)qi;
                ex.c_pair_f0[qi] = 0;
            }
            if (qi == ex.c_nq - 1 && lane >= 1 && ex.c_nq + lane - 1 < nrec) {
                ex.c_pair_q[ex.c_nq + lane - 1] = -1;
                ex.c_pair_f0[ex.c_nq + lane - 1] = 0;
            }
        }
        if (qi == 0) {
            if (lane == 0) {
                ex.c_pair_off[0] = 0;
                ex.c_pair_off[1] = (int)nrec;
                ex.c_unit_prefix[0] = 0;
                ex.c_unit_prefix[1] = (int)(nrec / TK_UNIT_Q) * ex.c_chunks;
            }
            if (lane < TK_TICKETS) ex.c_unit_prefix[TK_TICKET_OFF(1) + lane * 32] = 0;
        }
    }
    if (lane == 0) {
        shift_out[qi] = shift;
        scale_out[qi] = scale;
    }
}

void tk_launch_build_tables(const float *centers, int dq, int dpb, int f_order, const void *q,
                            int q_is_f64, int64_t nq, double aux0, double aux1, int signd,
                            uint8_t *tables, void *shift, double *scale, hipStream_t s, TkSecond q2,
                            const TkTablesExtra *extra)
{
    if (nq == 0) return;
    TkTablesExtra ex;
    if (extra) ex = *extra;
    const int M = dq / dpb;
    const size_t esz = q_is_f64 ? 8 : 4;
    // per-wave LDS: dists + 64x8 accumulators + 128 tree nodes
    const PwProgram pw = pw_program(16 * M);
    int wave_lds = (int)(((size_t)16 * M + 64 * 8 + 128) * esz);
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
                               (double *)shift, scale, nq, wave_lds, pw,
                               (const double *)q2.b, q2.n_a, ex);
        else
            hipLaunchKernelGGL((build_tables_kernel<double, false>), grid, block, lds, s, centers,
                               dq, pdpb, f_order, (const double *)q, aux0, aux1, tables,
                               (double *)shift, scale, nq, wave_lds, pw,
                               (const double *)q2.b, q2.n_a, ex);
    } else {
        if (signd)
            hipLaunchKernelGGL((build_tables_kernel<float, true>), grid, block, lds, s, centers, dq,
                               pdpb, f_order, (const float *)q, aux0, aux1, tables, (float *)shift, scale, nq, wave_lds, pw,
                               (const float *)q2.b, q2.n_a, ex);
        else
            hipLaunchKernelGGL((build_tables_kernel<float, false>), grid, block, lds, s, centers,
                               dq, pdpb, f_order, (const float *)q, aux0, aux1, tables,
                               (float *)shift, scale, nq, wave_lds, pw,
                               (const float *)q2.b, q2.n_a, ex);
    }
}
