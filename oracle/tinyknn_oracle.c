/*
 * tinyknn_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the ONE hot path of thomasahle/tinyknn (the
 * Quick-ADC 4-bit PQ scan + bounded order-dependent top-R heap behind
 * FastPQ / _FastDistanceTable / IVF.query).  It exists only so that
 *   - tests/ can check the HIP kernels bit-for-bit,
 *   - __graft_entry__.smoke() can check one small GPU invocation,
 *   - bench.py can time a CPU baseline ("cpu_baseline.kind": "port").
 * Nothing under tinyknn_amd/ may import, link or call this file.
 *
 * Parity pin: every function here is checked against golden vectors that were
 * produced by the *compiled, unmodified* reference in the build container
 * (tests/golden/make_golden.py; fixtures: the .npz files in tests/golden) and against the
 * known-answer tests the reference's own test-suite holds (tests/test_oracle_*).
 * The reference's native code is Cython (needs generated C++), so no
 * oracle/_ref build exists; see DESIGN.md.
 *
 * Citations are file:line relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

typedef int64_t i64;

#define TKO_ORDER_SSE 0
#define TKO_ORDER_AVX 1

/* ------------------------------------------------------------------ */
/* Layout transforms (tinyknn/_transform.py)                           */
/* ------------------------------------------------------------------ */

/* transform_data, _transform.py:4-77.  codes: (n, M) uint8 4-bit values, n%16==0,
 * M even.  packed: uint64 (n/16, M).  Viewed as bytes, chunk c / block pair p /
 * row r lives at byte c*8M + 16p + r and holds code[16c+r][2p] (low nibble) and
 * code[16c+r][2p+1] (high nibble). */
void tko_pack_codes(const uint8_t *codes, i64 n, int M, uint64_t *packed)
{
    uint8_t *out = (uint8_t *)packed;
    i64 chunks = n / 16;
    for (i64 c = 0; c < chunks; c++)
        for (int p = 0; p < M / 2; p++)
            for (int r = 0; r < 16; r++) {
                const uint8_t *row = codes + (16 * c + r) * (i64)M;
                out[c * 8 * (i64)M + 16 * p + r] =
                    (uint8_t)((row[2 * p] & 15) | ((row[2 * p + 1] & 15) << 4));
            }
}

/* unpack, _transform.py:80-111 */
void tko_unpack_codes(const uint64_t *packed, i64 chunks, int M, uint8_t *codes)
{
    const uint8_t *in = (const uint8_t *)packed;
    for (i64 c = 0; c < chunks; c++)
        for (int p = 0; p < M / 2; p++)
            for (int r = 0; r < 16; r++) {
                uint8_t b = in[c * 8 * (i64)M + 16 * p + r];
                uint8_t *row = codes + (16 * c + r) * (i64)M;
                row[2 * p] = b & 15;
                row[2 * p + 1] = b >> 4;
            }
}

/* transform_tables, _transform.py:114-138: (M,16) uint8 row-major viewed as
 * uint64[2M] on a little-endian host: a byte copy. */
void tko_transform_tables(const uint8_t *table, int M, uint64_t *out)
{
    memcpy(out, table, (size_t)M * 16);
}

/* ------------------------------------------------------------------ */
/* Quick-ADC block distances                                           */
/* ------------------------------------------------------------------ */

static inline int sat_add(int a, int b, int signd)
{
    int s = a + b;
    if (signd) {
        if (s > 127) s = 127;
        if (s < -128) s = -128;
    } else {
        if (s > 255) s = 255;
    }
    return s;
}

/* compute_block_dists (_fast_pq.pyx:209-236, single accumulator, blocks in
 * order) and compute_block_dists_avx (_fast_pq_256.pyx:126-156: accumulator 0
 * takes blocks m%4 in {0,1}, accumulator 1 blocks m%4 in {2,3}, then one final
 * saturating add :152-156).  The AVX loop runs M/4 steps, the SSE loop M/2, so
 * trailing blocks beyond 4*(M/4) resp. 2*(M/2) are ignored, as there. */
static void block_dists_scalar(const uint8_t *chunk, int M, const uint8_t *T,
                               int signd, int order, uint8_t out[16])
{
    int Muse = (order == TKO_ORDER_AVX) ? 4 * (M / 4) : 2 * (M / 2);
    for (int r = 0; r < 16; r++) {
        int acc0 = 0, acc1 = 0;
        for (int m = 0; m < Muse; m++) {
            uint8_t b = chunk[16 * (m >> 1) + r];
            int code = (m & 1) ? (b >> 4) : (b & 15);
            int t = signd ? (int)(int8_t)T[16 * m + code] : (int)T[16 * m + code];
            if (order == TKO_ORDER_AVX && ((m >> 1) & 1))
                acc1 = sat_add(acc1, t, signd);
            else
                acc0 = sat_add(acc0, t, signd);
        }
        int res = (order == TKO_ORDER_AVX) ? sat_add(acc0, acc1, signd) : acc0;
        out[r] = (uint8_t)res;
    }
}

#if defined(__x86_64__)
/* Same arithmetic with pshufb / (v)paddsb, used only to make the CPU baseline
 * an honest SIMD one; checked against the scalar version in tests. */
__attribute__((target("avx2"))) static void
block_dists_avx2(const uint8_t *chunk, int M, const uint8_t *T, int signd,
                 int order, uint8_t out[16])
{
    const __m128i low = _mm_set1_epi8(0x0f);
    if (order == TKO_ORDER_AVX) {
        const __m256i low2 = _mm256_set1_epi8(0x0f);
        __m256i acc = _mm256_setzero_si256();
        for (int j = 0; j < M / 4; j++) {
            __m256i blk = _mm256_loadu_si256((const __m256i *)(chunk + 32 * j));
            __m256i tlo = _mm256_set_m128i(
                _mm_loadu_si128((const __m128i *)(T + 64 * j + 32)),
                _mm_loadu_si128((const __m128i *)(T + 64 * j)));
            __m256i thi = _mm256_set_m128i(
                _mm_loadu_si128((const __m128i *)(T + 64 * j + 48)),
                _mm_loadu_si128((const __m128i *)(T + 64 * j + 16)));
            __m256i dl = _mm256_shuffle_epi8(tlo, _mm256_and_si256(blk, low2));
            __m256i dh = _mm256_shuffle_epi8(
                thi, _mm256_and_si256(_mm256_srli_epi64(blk, 4), low2));
            if (signd) {
                acc = _mm256_adds_epi8(acc, dl);
                acc = _mm256_adds_epi8(acc, dh);
            } else {
                acc = _mm256_adds_epu8(acc, dl);
                acc = _mm256_adds_epu8(acc, dh);
            }
        }
        __m128i lo = _mm256_extracti128_si256(acc, 0);
        __m128i hi = _mm256_extracti128_si256(acc, 1);
        __m128i res = signd ? _mm_adds_epi8(lo, hi) : _mm_adds_epu8(lo, hi);
        _mm_storeu_si128((__m128i *)out, res);
    } else {
        __m128i acc = _mm_setzero_si128();
        for (int j = 0; j < M / 2; j++) {
            __m128i blk = _mm_loadu_si128((const __m128i *)(chunk + 16 * j));
            __m128i t0 = _mm_loadu_si128((const __m128i *)(T + 32 * j));
            __m128i t1 = _mm_loadu_si128((const __m128i *)(T + 32 * j + 16));
            __m128i dl = _mm_shuffle_epi8(t0, _mm_and_si128(blk, low));
            __m128i dh =
                _mm_shuffle_epi8(t1, _mm_and_si128(_mm_srli_epi64(blk, 4), low));
            if (signd) {
                acc = _mm_adds_epi8(acc, dl);
                acc = _mm_adds_epi8(acc, dh);
            } else {
                acc = _mm_adds_epu8(acc, dl);
                acc = _mm_adds_epu8(acc, dh);
            }
        }
        _mm_storeu_si128((__m128i *)out, acc);
    }
}
#endif

static int g_force_scalar = 0;
void tko_force_scalar(int on) { g_force_scalar = on; }

static int have_avx2(void)
{
#if defined(__x86_64__)
    static int cached = -1;
    if (cached < 0) cached = __builtin_cpu_supports("avx2") ? 1 : 0;
    return cached && !g_force_scalar;
#else
    return 0;
#endif
}
int tko_simd(void) { return have_avx2(); }

static inline void block_dists(const uint8_t *chunk, int M, const uint8_t *T,
                               int signd, int order, uint8_t out[16])
{
#if defined(__x86_64__)
    if (have_avx2()) {
        block_dists_avx2(chunk, M, T, signd, order, out);
        return;
    }
#endif
    block_dists_scalar(chunk, M, T, signd, order, out);
}

/* estimate_pq_sse _fast_pq.pyx:101-111 / estimate_pq_avx _fast_pq_256.pyx:52-62 */
void tko_estimate_pq(const uint64_t *data, i64 chunks, int M, const uint64_t *tables,
                     uint64_t *out, int signd, int order)
{
    const uint8_t *d8 = (const uint8_t *)data;
    const uint8_t *T = (const uint8_t *)tables;
    uint8_t *o8 = (uint8_t *)out;
    for (i64 c = 0; c < chunks; c++)
        block_dists(d8 + c * 8 * (i64)M, M, T, signd, order, o8 + 16 * c);
}

/* ------------------------------------------------------------------ */
/* Heap primitives                                                     */
/* ------------------------------------------------------------------ */

/* init_heap _fast_pq.pyx:240-252 */
void tko_init_heap(i64 *indices, int32_t *vals, int R, int signd)
{
    for (int i = 0; i < R; i++) {
        indices[i] = -1;
        vals[i] = signd ? 127 : 255;
    }
}

/* insert _fast_pq.pyx:274-307 (copy _fast_pq_256.pyx:188-210): return if the
 * label is already anywhere in the array, else replace the root and sift down,
 * taking the left child unless the right one is strictly greater. */
void tko_heap_insert(i64 *indices, int32_t *vals, int R, i64 i, int32_t v)
{
    for (int j = 0; j < R; j++)
        if (indices[j] == i) return;
    int j = 0;
    for (;;) {
        int nxt = j;
        int32_t nxt_val = v;
        int l = 2 * j + 1, r = 2 * j + 2;
        if (l < R && vals[l] > nxt_val) { nxt = l; nxt_val = vals[l]; }
        if (r < R && vals[r] > nxt_val) { nxt = r; nxt_val = vals[r]; }
        if (nxt == j) { vals[j] = v; indices[j] = i; break; }
        vals[j] = vals[nxt];
        indices[j] = indices[nxt];
        j = nxt;
    }
}

/* insert_is _fast_pq.pyx:256-271 */
void tko_heap_insert_is(i64 *indices, int32_t *vals, int R, i64 i, int32_t v)
{
    for (int j = 0; j < R; j++)
        if (indices[j] == i) return;
    int j = 0;
    while (j + 1 != R && vals[j + 1] > v) {
        indices[j] = indices[j + 1];
        vals[j] = vals[j + 1];
        j++;
    }
    indices[j] = i;
    vals[j] = v;
}

/* query_pq_sse _fast_pq.pyx:114-206 / query_pq_avx _fast_pq_256.pyx:65-123.
 * The bound is the low 8 bits of vals[0] captured at block start (:73 / :153);
 * every lane that passes is inserted without re-checking (:111-118); the bound
 * is refreshed once per block that had a hit (:123).
 * Optional counters: n_ins[0] += inserts attempted, n_ins[1] += blocks with hit. */
static int32_t *g_block_trace = NULL;   /* optional: inserts attempted per block */
void tko_set_block_trace(int32_t *buf) { g_block_trace = buf; }

void tko_query_pq_stats(const uint64_t *data, i64 chunks, int M, i64 n,
                        const uint64_t *tables, i64 *indices, int32_t *vals, int R,
                        int signd, const i64 *labels, int order, i64 *n_ins)
{
    const uint8_t *d8 = (const uint8_t *)data;
    const uint8_t *T = (const uint8_t *)tables;
    uint8_t bound = (uint8_t)(vals[0] & 0xff);
    uint8_t dist[16];
    for (i64 c = 0; c < chunks; c++) {
        block_dists(d8 + c * 8 * (i64)M, M, T, signd, order, dist);
        unsigned bits = 0;
        for (int r = 0; r < 16; r++) {
            int lt = signd ? ((int8_t)dist[r] < (int8_t)bound) : (dist[r] < bound);
            bits |= (unsigned)lt << r;
        }
        if (!bits) continue;
        if (n_ins) n_ins[1]++;
        for (int r = 0; r < 16; r++) {
            if (!((bits >> r) & 1)) continue;
            i64 pos = 16 * c + r;
            if (pos < n) {
                i64 label = labels ? labels[pos] : pos;
                int32_t v = signd ? (int32_t)(int8_t)dist[r] : (int32_t)dist[r];
                if (n_ins) n_ins[0]++;
                if (g_block_trace) g_block_trace[c]++;
                tko_heap_insert(indices, vals, R, label, v);
            }
        }
        bound = (uint8_t)(vals[0] & 0xff);
    }
}

void tko_query_pq(const uint64_t *data, i64 chunks, int M, i64 n,
                  const uint64_t *tables, i64 *indices, int32_t *vals, int R,
                  int signd, const i64 *labels, int order)
{
    tko_query_pq_stats(data, chunks, M, n, tables, indices, vals, R, signd, labels,
                       order, NULL);
}

/* ------------------------------------------------------------------ */
/* numpy arithmetic the path depends on (numpy 2.2.6, SSE3 baseline)    */
/* ------------------------------------------------------------------ */

/* numpy pairwise summation (loops_utils.h.src @TYPE@_pairwise_sum): leaves of
 * <=128 elements with 8 accumulators, halves rounded down to a multiple of 8.
 * Used by np.add.reduce, hence numpy.core._methods._mean (fast_pq.py:15,214). */
#define PW_BLOCK 128
#define DEF_PAIRWISE(NAME, T)                                                   \
    static T NAME(const T *a, i64 n)                                            \
    {                                                                           \
        if (n < 8) {                                                            \
            T res = 0;                                                          \
            for (i64 i = 0; i < n; i++) res += a[i];                            \
            return res;                                                         \
        } else if (n <= PW_BLOCK) {                                             \
            T r[8];                                                             \
            i64 i;                                                              \
            for (i = 0; i < 8; i++) r[i] = a[i];                                \
            for (i = 8; i < n - (n % 8); i += 8)                                \
                for (int j = 0; j < 8; j++) r[j] += a[i + j];                   \
            T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])); \
            for (; i < n; i++) res += a[i];                                     \
            return res;                                                         \
        } else {                                                                \
            i64 n2 = n / 2;                                                     \
            n2 -= n2 % 8;                                                       \
            return NAME(a, n2) + NAME(a + n2, n - n2);                          \
        }                                                                       \
    }
DEF_PAIRWISE(pairwise_f32, float)
DEF_PAIRWISE(pairwise_f64, double)

float tko_pairwise_sum_f32(const float *a, i64 n) { return pairwise_f32(a, n); }
double tko_pairwise_sum_f64(const double *a, i64 n) { return pairwise_f64(a, n); }

/* np.einsum("...k,...k->...") inner kernel for two contiguous operands and a
 * stride-0 output (einsum_sumprod.c.src, *_sum_of_products_contig_contig_outstride0_two)
 * as built for the SSE3 baseline: L lanes (4 for f32, 2 for f64), un-fused
 * multiply-add, groups of 4 vectors folded in the order 3,2,1,0, zero-filled
 * tail vectors, then a horizontal add ((l0+l1)+(l2+l3)).  Used by
 * fast_pq.py:207 (count = dims_per_block) and utils.py:91 (count = d). */
static float einsum_dot_f32(const float *a, const float *b, i64 count)
{
    float acc[4] = {0, 0, 0, 0};
    i64 i = 0;
    for (; count - i >= 16; i += 16) {
        for (int l = 0; l < 4; l++) {
            float ab3 = a[i + 12 + l] * b[i + 12 + l] + acc[l];
            float ab2 = a[i + 8 + l] * b[i + 8 + l] + ab3;
            float ab1 = a[i + 4 + l] * b[i + 4 + l] + ab2;
            acc[l] = a[i + l] * b[i + l] + ab1;
        }
    }
    for (; i < count; i += 4)
        for (int l = 0; l < 4; l++) {
            float x = (i + l < count) ? a[i + l] : 0.0f;
            float y = (i + l < count) ? b[i + l] : 0.0f;
            acc[l] = x * y + acc[l];
        }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

static double einsum_dot_f64(const double *a, const double *b, i64 count)
{
    double acc[2] = {0, 0};
    i64 i = 0;
    for (; count - i >= 8; i += 8) {
        for (int l = 0; l < 2; l++) {
            double ab3 = a[i + 6 + l] * b[i + 6 + l] + acc[l];
            double ab2 = a[i + 4 + l] * b[i + 4 + l] + ab3;
            double ab1 = a[i + 2 + l] * b[i + 2 + l] + ab2;
            acc[l] = a[i + l] * b[i + l] + ab1;
        }
    }
    for (; i < count; i += 2)
        for (int l = 0; l < 2; l++) {
            double x = (i + l < count) ? a[i + l] : 0.0;
            double y = (i + l < count) ? b[i + l] : 0.0;
            acc[l] = x * y + acc[l];
        }
    return acc[0] + acc[1];
}

float tko_einsum_dot_f32(const float *a, const float *b, i64 n) { return einsum_dot_f32(a, b, n); }
double tko_einsum_dot_f64(const double *a, const double *b, i64 n) { return einsum_dot_f64(a, b, n); }

/* ------------------------------------------------------------------ */
/* Distance tables (fast_pq.py:186-252)                                */
/* ------------------------------------------------------------------ */

/* `dists` keeps numpy's MEMORY order, because np.add.reduce (the mean) sums in
 * memory order: FastPQ.centers is C-contiguous for dims_per_block >= 2 but an
 * F-ordered view for dims_per_block == 1 (fast_pq.py:99-101: the reshape after
 * the transpose is then copy-free), and `centers - q` / einsum keep that order. */
#define DIDX(i, m) (f_order ? ((m) * 16 + (i)) : ((i) * M + (m)))

/* (uint8)(int) truncation = what x86 numpy's astype(uint8) gives for the small
 * negative table entries (fast_pq.py:220). */
static inline uint8_t f64_to_u8_wrap(double v) { return (uint8_t)(int32_t)v; }

/* FastPQ.distance_table, float32 path (no rotation, float32 query):
 * fast_pq.py:206-221.  centers (16, dq) f32 row-major, dq = M*dpb; q padded to dq.
 * table out: (M,16) uint8 (= the transform_tables byte image). */
int tko_distance_table_f32(const float *centers, int dq, int dpb, const float *q,
                           double sqrt_n_blocks, int f_order, uint8_t *table, float *shift_out,
                           double *scale_out)
{
    int M = dq / dpb;
    float *dists = (float *)malloc(sizeof(float) * 16 * (size_t)M);
    float *diff = (float *)malloc(sizeof(float) * (size_t)dpb);
    if (!dists || !diff) return -1;
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++) {
            for (int k = 0; k < dpb; k++)
                diff[k] = centers[i * (i64)dq + m * dpb + k] - q[m * dpb + k];
            dists[DIDX(i, m)] = einsum_dot_f32(diff, diff, dpb); /* :206-207 */
        }
    i64 cnt = 16 * (i64)M;
    float mean = pairwise_f32(dists, cnt) / (float)cnt;          /* _mean */
    float shift = mean * 0.6931471806f;                          /* :214 */
    float mx = -INFINITY;
    for (i64 i = 0; i < cnt; i++) {
        dists[i] -= shift;                                       /* :215 */
        if (dists[i] > mx) mx = dists[i];
    }
    double scale = 128.0 / ((double)mx * sqrt_n_blocks);         /* :216 */
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++)
            table[m * 16 + i] = f64_to_u8_wrap(rint((double)dists[DIDX(i, m)] * scale)); /* :217-221 */
    *shift_out = shift;
    *scale_out = scale;
    free(dists);
    free(diff);
    return 0;
}

/* Same, float64 path: taken when the query is float64 after `q @ R.T`
 * (fast_pq.py:203-204; R is float64) or was float64 to begin with. */
int tko_distance_table_f64(const float *centers, int dq, int dpb, const double *q,
                           double sqrt_n_blocks, int f_order, uint8_t *table, double *shift_out,
                           double *scale_out)
{
    int M = dq / dpb;
    double *dists = (double *)malloc(sizeof(double) * 16 * (size_t)M);
    double *diff = (double *)malloc(sizeof(double) * (size_t)dpb);
    if (!dists || !diff) return -1;
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++) {
            for (int k = 0; k < dpb; k++)
                diff[k] = (double)centers[i * (i64)dq + m * dpb + k] - q[m * dpb + k];
            dists[DIDX(i, m)] = einsum_dot_f64(diff, diff, dpb);
        }
    i64 cnt = 16 * (i64)M;
    double mean = pairwise_f64(dists, cnt) / (double)cnt;
    double shift = mean * 0.6931471806;
    double mx = -INFINITY;
    for (i64 i = 0; i < cnt; i++) {
        dists[i] -= shift;
        if (dists[i] > mx) mx = dists[i];
    }
    double scale = 128.0 / (mx * sqrt_n_blocks);
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++)
            table[m * 16 + i] = f64_to_u8_wrap(rint(dists[DIDX(i, m)] * scale));
    *shift_out = shift;
    *scale_out = scale;
    free(dists);
    free(diff);
    return 0;
}

/* FastPQ.udistance_table (experimental, fast_pq.py:224-252).  np.square then a
 * sum over the last axis (sequential for dims_per_block < 8); shift = min;
 * scale = 255 / ((max * log(nb)) * sqrt(nb)), left to right as Python evaluates
 * fast_pq.py:248; the caller passes np.log(nb) and np.sqrt(nb) (float64).  The f32 path multiplies in f64 because
 * `scale` is a numpy float64 scalar. */
int tko_udistance_table_f32(const float *centers, int dq, int dpb, const float *q,
                            double lognb, double sqrtnb, uint8_t *table, float *shift_out,
                            double *scale_out)
{
    int M = dq / dpb;
    float *dists = (float *)malloc(sizeof(float) * 16 * (size_t)M);
    if (!dists) return -1;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++) {
            float s = 0.0f;
            for (int k = 0; k < dpb; k++) {
                float df = centers[i * (i64)dq + m * dpb + k] - q[m * dpb + k];
                float sq = df * df;
                s += sq;
            }
            dists[i * M + m] = s;
            if (s < mn) mn = s;
        }
    for (i64 i = 0; i < 16 * (i64)M; i++) {
        dists[i] -= mn;
        if (dists[i] > mx) mx = dists[i];
    }
    double scale = 255.0 / (((double)mx * lognb) * sqrtnb);
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++)
            table[m * 16 + i] = f64_to_u8_wrap(rint((double)dists[i * M + m] * scale));
    *shift_out = mn;
    *scale_out = scale;
    free(dists);
    return 0;
}

int tko_udistance_table_f64(const float *centers, int dq, int dpb, const double *q,
                            double lognb, double sqrtnb, uint8_t *table, double *shift_out,
                            double *scale_out)
{
    int M = dq / dpb;
    double *dists = (double *)malloc(sizeof(double) * 16 * (size_t)M);
    if (!dists) return -1;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++) {
            double s = 0.0;
            for (int k = 0; k < dpb; k++) {
                double df = (double)centers[i * (i64)dq + m * dpb + k] - q[m * dpb + k];
                double sq = df * df;
                s += sq;
            }
            dists[i * M + m] = s;
            if (s < mn) mn = s;
        }
    for (i64 i = 0; i < 16 * (i64)M; i++) {
        dists[i] -= mn;
        if (dists[i] > mx) mx = dists[i];
    }
    double scale = 255.0 / ((mx * lognb) * sqrtnb);
    for (int i = 0; i < 16; i++)
        for (int m = 0; m < M; m++)
            table[m * 16 + i] = f64_to_u8_wrap(rint(dists[i * M + m] * scale));
    *shift_out = mn;
    *scale_out = scale;
    free(dists);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Exact rescoring (utils.py:22-25, 89-92)                             */
/* ------------------------------------------------------------------ */

/* knn_brute1's distances: diff = Y[idx] - x, einsum("ij,ij->i") — float32 when both
 * operands are float32, float64 otherwise (numpy's promotion).  Results are stored
 * as double (exact for the float32 case).  A negative index addresses from the end,
 * as numpy fancy indexing does (fast_pq.py:311 can see -1 sentinels). */
void tko_sqdist_gather(const void *x, int x_is_f64, const void *Y, int y_is_f64, i64 nY, int d,
                       const i64 *idx, i64 n, double *out)
{
    if (!x_is_f64 && !y_is_f64) {
        const float *xf = (const float *)x, *Yf = (const float *)Y;
        float *diff = (float *)malloc(sizeof(float) * (size_t)d);
        for (i64 i = 0; i < n; i++) {
            i64 r = idx[i] < 0 ? idx[i] + nY : idx[i];
            const float *y = Yf + r * (i64)d;
            for (int j = 0; j < d; j++) diff[j] = y[j] - xf[j];
            out[i] = (double)einsum_dot_f32(diff, diff, d);
        }
        free(diff);
        return;
    }
    double *diff = (double *)malloc(sizeof(double) * (size_t)d);
    for (i64 i = 0; i < n; i++) {
        i64 r = idx[i] < 0 ? idx[i] + nY : idx[i];
        for (int j = 0; j < d; j++) {
            double yv = y_is_f64 ? ((const double *)Y)[r * (i64)d + j] : (double)((const float *)Y)[r * (i64)d + j];
            double xv = x_is_f64 ? ((const double *)x)[j] : (double)((const float *)x)[j];
            diff[j] = yv - xv;
        }
        out[i] = einsum_dot_f64(diff, diff, d);
    }
    free(diff);
}

/* bottom_k (utils.py:22-25): if k >= n, arange(n); else the positions of the k
 * smallest.  numpy's argpartition order is an implementation detail; on the
 * fixture host (numpy 2.2.6, AVX-512 argselect) the first k come back ascending
 * for the sizes this path produces, so ascending (ties: lower position first) is
 * the canonical order here.  Returns the number of positions written. */
i64 tko_bottom_k(const double *dists, i64 n, i64 k, i64 *out)
{
    if (k >= n) {
        for (i64 i = 0; i < n; i++) out[i] = i;
        return n;
    }
    /* stable insertion sort of positions by value; n <= R is small */
    i64 *pos = (i64 *)malloc(sizeof(i64) * (size_t)n);
    for (i64 i = 0; i < n; i++) {
        i64 j = i;
        while (j > 0 && dists[pos[j - 1]] > dists[i]) {
            pos[j] = pos[j - 1];
            j--;
        }
        pos[j] = i;
    }
    for (i64 i = 0; i < k; i++) out[i] = pos[i];
    free(pos);
    return k;
}

/* ------------------------------------------------------------------ */
/* IVF.query orchestration (ivf.py:106-163, fast_pq.py:284-312)        */
/* ------------------------------------------------------------------ */

typedef struct {
    int d;               /* raw vector dimension                       */
    int dq;              /* PQ dimension = M * dpb                     */
    int dpb;             /* dims per block                             */
    int M;               /* blocks                                     */
    int order;           /* TKO_ORDER_*                                */
    int rotated;         /* 1: caller supplies q_pq as float64 (dq,)   */
    double sqrt_n_blocks;
    const float *pq_centers;      /* (16, dq)                           */
    i64 n_lists;                  /* active centres                     */
    const uint64_t *center_codes; /* (center_chunks, M)                 */
    i64 center_chunks;
    const float *active_centers;  /* (n_lists, d)                       */
    const i64 *list_chunk_off;    /* (n_lists+1,) in chunks             */
    const i64 *list_n;            /* (n_lists,) true rows               */
    const uint64_t *codes;        /* all lists, reference chunk layout  */
    const i64 *ids_off;           /* (n_lists+1,)                       */
    const i64 *ids;               /* labels, list-major                 */
    const void *data;             /* (N, d) rescoring vectors           */
    i64 N;
    int data_is_f64;              /* dtype of `data` (ivf.py:77 keeps X's) */
} tko_index;

/* One query.  `q` is the float32 query AFTER the caller applied the metric's
 * normalisation (ivf.py:125-127 uses BLAS, which is not restated).  `q_pq` is the
 * padded (and, when index->rotated, rotated float64) query fed to the table
 * build.  Outputs: out_ids (<= max(k, pass_1) entries), optional probe order and
 * final heap arrays for the parity tests.  Returns the number of ids. */
i64 tko_ivf_query(const tko_index *ix, const float *q, const void *q_pq, int k,
                  int n_probes, int pass_1, i64 *out_ids, i64 *out_probes,
                  i64 *out_heap_idx, int32_t *out_heap_val, uint8_t *out_table)
{
    int M = ix->M;
    uint8_t *table = (uint8_t *)malloc((size_t)M * 16);
    if (ix->rotated) {
        double sh, sc;
        tko_distance_table_f64(ix->pq_centers, ix->dq, ix->dpb, (const double *)q_pq,
                               ix->sqrt_n_blocks, ix->dpb == 1, table, &sh, &sc);
    } else {
        float sh;
        double sc;
        tko_distance_table_f32(ix->pq_centers, ix->dq, ix->dpb, (const float *)q_pq,
                               ix->sqrt_n_blocks, ix->dpb == 1, table, &sh, &sc);
    }
    if (out_table) memcpy(out_table, table, (size_t)M * 16);
    const uint64_t *tables = (const uint64_t *)table;

    /* coarse stage: dtable.top(centers, active_centers, k=n_probes)  ivf.py:131 */
    i64 nC = ix->n_lists;
    i64 kc = n_probes < nC ? n_probes : nC;                 /* fast_pq.py:291 */
    i64 rescore = 2 * kc + 10 < nC ? 2 * kc + 10 : nC;      /* :293-294 */
    i64 *cidx = (i64 *)malloc(sizeof(i64) * (size_t)rescore);
    int32_t *cval = (int32_t *)malloc(sizeof(int32_t) * (size_t)rescore);
    tko_init_heap(cidx, cval, (int)rescore, 1);
    tko_query_pq(ix->center_codes, ix->center_chunks, M, nC, tables, cidx, cval,
                 (int)rescore, 1, NULL, ix->order);
    i64 *probes = (i64 *)malloc(sizeof(i64) * (size_t)rescore);
    i64 n_top;
    if (rescore <= kc) {                                    /* :307-308 */
        for (i64 i = 0; i < rescore; i++) probes[i] = cidx[i];
        n_top = rescore;
    } else {
        double *cd = (double *)malloc(sizeof(double) * (size_t)rescore);
        i64 *best = (i64 *)malloc(sizeof(i64) * (size_t)rescore);
        tko_sqdist_gather(q, 0, ix->active_centers, 0, nC, ix->d, cidx, rescore, cd);
        n_top = tko_bottom_k(cd, rescore, kc, best);        /* :311-312 */
        for (i64 i = 0; i < n_top; i++) probes[i] = cidx[best[i]];
        free(cd);
        free(best);
    }
    if (out_probes)
        for (i64 i = 0; i < n_top; i++) out_probes[i] = probes[i];

    /* list scans through ONE heap, in probe order  ivf.py:135-150 */
    if (pass_1 <= 0) pass_1 = (n_probes + 1) * k + 1;
    i64 *hidx = (i64 *)malloc(sizeof(i64) * (size_t)pass_1);
    int32_t *hval = (int32_t *)malloc(sizeof(int32_t) * (size_t)pass_1);
    for (int i = 0; i < pass_1; i++) { hidx[i] = -1; hval[i] = 127; }
    for (i64 t = 0; t < n_top; t++) {
        i64 cl = probes[t] < 0 ? probes[t] + nC : probes[t];
        i64 c0 = ix->list_chunk_off[cl];
        tko_query_pq(ix->codes + c0 * M, ix->list_chunk_off[cl + 1] - c0, M,
                     ix->list_n[cl], tables, hidx, hval, pass_1, 1,
                     ix->ids + ix->ids_off[cl], ix->order);
    }
    if (out_heap_idx) memcpy(out_heap_idx, hidx, sizeof(i64) * (size_t)pass_1);
    if (out_heap_val) memcpy(out_heap_val, hval, sizeof(int32_t) * (size_t)pass_1);

    /* strip sentinels, rescore  ivf.py:154-163 */
    i64 nc = 0;
    for (int i = 0; i < pass_1; i++)
        if (hidx[i] != -1) hidx[nc++] = hidx[i];
    i64 n_out;
    if (nc <= k) {
        for (i64 i = 0; i < nc; i++) out_ids[i] = hidx[i];
        n_out = nc;
    } else {
        double *fd = (double *)malloc(sizeof(double) * (size_t)nc);
        i64 *best = (i64 *)malloc(sizeof(i64) * (size_t)nc);
        tko_sqdist_gather(q, 0, ix->data, ix->data_is_f64, ix->N, ix->d, hidx, nc, fd);
        n_out = tko_bottom_k(fd, nc, k, best);
        for (i64 i = 0; i < n_out; i++) out_ids[i] = hidx[best[i]];
        free(fd);
        free(best);
    }
    free(table); free(cidx); free(cval); free(probes); free(hidx); free(hval);
    return n_out;
}

/* Batch loop used only by bench.py's cpu_baseline leg (one thread, one query at
 * a time, exactly the reference's protocol examples/bench.py:118-137 minus the
 * Python interpreter).  out_ids is (nq, k) padded with -1. */
void tko_ivf_query_batch(const tko_index *ix, const float *qs, const void *qs_pq,
                         i64 nq, int k, int n_probes, int pass_1, i64 *out_ids)
{
    int R = pass_1 > 0 ? pass_1 : (n_probes + 1) * k + 1;
    i64 *tmp = (i64 *)malloc(sizeof(i64) * (size_t)(R > k ? R : k));
    size_t qpq_stride = (size_t)ix->dq * (ix->rotated ? sizeof(double) : sizeof(float));
    for (i64 i = 0; i < nq; i++) {
        i64 n = tko_ivf_query(ix, qs + i * (i64)ix->d,
                              (const char *)qs_pq + i * qpq_stride, k, n_probes,
                              pass_1, tmp, NULL, NULL, NULL, NULL);
        for (int j = 0; j < k; j++) out_ids[i * k + j] = j < n ? tmp[j] : -1;
    }
    free(tmp);
}

/* ------------------------------------------------------------------ */
/* Offline build path: the step BEFORE the hot path   */
/* FastPQ.transform fast_pq.py:147-184, IVF.build ivf.py:85-102,       */
/* knn_brute utils.py:66-86                                            */
/* ------------------------------------------------------------------ */

/* numpy's `A @ B.T` here is an OpenBLAS GEMM.  On the fixture host (OpenBLAS 0.3.29,
 * AVX-512 kernels) every GEMM the build path issues — K = dims_per_block, K = d <= 128,
 * float32 and float64, >= 32 rows — returns, for every output element, the FMA chain
 * over k ascending starting from 0 (one accumulator per element, no K split)
 * [measured against exact rational arithmetic and numpy, tests/test_build_path.py].  NOT
 * restated, and kept in numpy by every caller: the rotation `data @ R.T` (fast_pq.py:168;
 * its DGEMM order changes with the row count and the thread split), operands of a few rows
 * (small-matrix kernels) and 1-row operands (GEMV).
 * argpartition(part, k)[:, :k] for k <= 2 is restated as numpy's generic path (dumb_select:
 * first occurrence of the minimum by strict `<`); with EXACT ties numpy's AVX-512 argselect
 * network may pick another of the tied entries — unpinned by the reference, absent from its
 * fixtures. */
static inline float chain_f32(const float *a2, const float *b, int K)
{
    float acc = 0.0f;
    for (int k = 0; k < K; k++) acc = fmaf(a2[k], b[k], acc);
    return acc;
}

/* labels[i][b] = knn_brute(col_b, code_b, 1)[i]   fast_pq.py:174-181: per block the
 * squared distance in the expanded form  (|x|^2 + |c|^2) - (2x).c   (utils.py:84; note
 * `2 * Xchunk @ Y.T` parses as (2*Xchunk) @ Y.T), norms by einsum, the product by GEMM,
 * argpartition(.., 1)[:, :1] = first occurrence of the minimum (numpy's dumb_select for
 * kth < 3).  data: (n, dq) rows already padded (and rotated: then float64). */
void tko_encode_pq(const float *centers, int dq, int dpb, const void *data, int is_f64, i64 n,
                   uint8_t *labels)
{
    const int M = dq / dpb;
    float yn[16];
    float x2f[32];
    double x2d[32], yd[32];
    for (int b = 0; b < M; b++) {
        for (int c = 0; c < 16; c++) {
            const float *y = centers + (i64)c * dq + b * dpb;
            yn[c] = einsum_dot_f32(y, y, dpb);
        }
        for (i64 i = 0; i < n; i++) {
            int best = 0;
            if (!is_f64) {
                const float *x = (const float *)data + i * dq + b * dpb;
                const float xn = einsum_dot_f32(x, x, dpb);
                for (int k = 0; k < dpb; k++) x2f[k] = 2.0f * x[k];
                float bestv = 0;
                for (int c = 0; c < 16; c++) {
                    const float p = chain_f32(x2f, centers + (i64)c * dq + b * dpb, dpb);
                    const float part = (xn + yn[c]) - p;
                    if (c == 0 || part < bestv) { bestv = part; best = c; }
                }
            } else {
                const double *x = (const double *)data + i * dq + b * dpb;
                const double xn = einsum_dot_f64(x, x, dpb);
                for (int k = 0; k < dpb; k++) x2d[k] = 2.0 * x[k];
                double bestv = 0;
                for (int c = 0; c < 16; c++) {
                    const float *y = centers + (i64)c * dq + b * dpb;
                    for (int k = 0; k < dpb; k++) yd[k] = (double)y[k];
                    double p = 0.0;
                    for (int k = 0; k < dpb; k++) p = fma(x2d[k], yd[k], p);
                    const double part = (xn + (double)yn[c]) - p;
                    if (c == 0 || part < bestv) { bestv = part; best = c; }
                }
            }
            labels[i * M + b] = (uint8_t)best;
        }
    }
}

/* np.linalg.norm(X, axis=1) for float32 rows: sqrt(add.reduce(x*x)) with numpy's
 * pairwise summation (ivf.py:79, utils.py:74-75) */
static float row_norm_f32(const float *x, int d, float *tmp)
{
    for (int k = 0; k < d; k++) tmp[k] = x[k] * x[k];
    return sqrtf(pairwise_f32(tmp, d));
}
static double row_norm_f64(const double *x, int d, double *tmp)
{
    for (int k = 0; k < d; k++) tmp[k] = x[k] * x[k];
    return sqrt(pairwise_f64(tmp, d));
}

/* np.argpartition(part, k, axis=1)[:, :k] for k <= 2 (kth < 3: numpy's dumb_select, a
 * selection sort of the first kth+1 positions by strict `<`, swapping into place) */
static void dumb_select_first(const double *v, i64 n, int k, i64 *out)
{
    /* position 0: first occurrence of the minimum */
    i64 m0 = 0;
    for (i64 j = 1; j < n; j++) if (v[j] < v[m0]) m0 = j;
    out[0] = m0;
    if (k < 2) return;
    /* after swapping positions 0 and m0 the scan order of the rest is
     * 1 .. m0-1, (original 0 at position m0), m0+1 .. n-1 */
    i64 best = -1;
    for (i64 pos = 1; pos < n; pos++) {
        const i64 j = (pos == m0) ? 0 : pos;
        if (best < 0 || v[j] < v[best]) best = j;
    }
    out[1] = best;
}

/* np.argpartition(part, k, axis=1)[:, :k] for 3 <= k <= 16.  numpy takes its introselect there
 * (or, where the CPU has AVX-512 / AVX2, x86-simd-sort's argselect): the SET of the first k is the k
 * smallest; their ORDER is an implementation detail the reference does not pin.  On the fixture
 * host (numpy 2.2.6, AVX-512) it is ascending for every k <= 9 and every row length tried
 * (31 .. 10 000); ascending by (value, position) is the canonical order here. */
static void select_ascending(const double *v, i64 n, int k, i64 *out)
{
    for (int t = 0; t < k; t++) {
        i64 best = -1;
        for (i64 j = 0; j < n; j++) {
            int taken = 0;
            for (int u = 0; u < t; u++) taken |= out[u] == j;
            if (!taken && (best < 0 || v[j] < v[best])) best = j;
        }
        out[t] = best;
    }
}

/* knn_brute(X, Y, k, metric) utils.py:66-86, k <= 16.  X: (n, d) float32 (IVF.data after
 * ivf.py:77-79); Y: (L, d) all_centers, float32 or float64; angular: both are divided by
 * their row norms first (utils.py:73-75).  out: (n, k).  Chunks of one row (n % 100 == 1)
 * are a GEMV in numpy and not restated: the caller keeps numpy for that row. */
int tko_assign(const float *X, i64 n, int d, const void *Y, int y_is_f64, i64 L, int k,
               int angular, i64 *out)
{
    if (k < 1 || k > 16 || k > L || d > 4096) return -1;
    float *Yf = NULL, *ynf = NULL, *xf = (float *)malloc(sizeof(float) * (size_t)d * 3);
    double *Yd = NULL, *ynd = NULL, *xd = (double *)malloc(sizeof(double) * (size_t)d * 2);
    double *part = (double *)malloc(sizeof(double) * (size_t)L);
    float *tmpf = xf + d, *x2f = xf + 2 * d;
    double *x2d = xd + d;
    if (y_is_f64) {
        Yd = (double *)malloc(sizeof(double) * (size_t)L * d);
        ynd = (double *)malloc(sizeof(double) * (size_t)L);
        double *tmpd = (double *)malloc(sizeof(double) * (size_t)d);
        for (i64 j = 0; j < L; j++) {
            const double *y = (const double *)Y + j * d;
            double nr = angular ? row_norm_f64(y, d, tmpd) : 1.0;
            for (int t = 0; t < d; t++) Yd[j * d + t] = angular ? y[t] / nr : y[t];
            ynd[j] = einsum_dot_f64(Yd + j * d, Yd + j * d, d);
        }
        free(tmpd);
    } else {
        Yf = (float *)malloc(sizeof(float) * (size_t)L * d);
        ynf = (float *)malloc(sizeof(float) * (size_t)L);
        for (i64 j = 0; j < L; j++) {
            const float *y = (const float *)Y + j * d;
            float nr = angular ? row_norm_f32(y, d, tmpf) : 1.0f;
            for (int t = 0; t < d; t++) Yf[j * d + t] = angular ? y[t] / nr : y[t];
            ynf[j] = einsum_dot_f32(Yf + j * d, Yf + j * d, d);
        }
    }
    for (i64 i = 0; i < n; i++) {
        const float *x0 = X + i * d;
        float nr = angular ? row_norm_f32(x0, d, tmpf) : 1.0f;
        for (int t = 0; t < d; t++) xf[t] = angular ? x0[t] / nr : x0[t];
        const float xn = einsum_dot_f32(xf, xf, d);
        if (y_is_f64) {
            /* float32 X against float64 Y: numpy promotes (2*X) to float64, DGEMM */
            for (int t = 0; t < d; t++) x2d[t] = (double)(2.0f * xf[t]);
            for (i64 j = 0; j < L; j++) {
                double p = 0.0;
                for (int t = 0; t < d; t++) p = fma(x2d[t], Yd[j * d + t], p);
                part[j] = ((double)xn + ynd[j]) - p;
            }
        } else {
            for (int t = 0; t < d; t++) x2f[t] = 2.0f * xf[t];
            for (i64 j = 0; j < L; j++) {
                const float p = chain_f32(x2f, Yf + j * d, d);
                part[j] = (double)((xn + ynf[j]) - p);
            }
        }
        if (k <= 2) dumb_select_first(part, L, k, out + i * k);
        else select_ascending(part, L, k, out + i * k);
    }
    free(xf); free(xd); free(part); free(Yf); free(ynf); free(Yd); free(ynd);
    return 0;
}
